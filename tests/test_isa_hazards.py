"""Static check of the generated gfx950 code (no GPU needed): 128-bit buffer stores and their data registers.

Round 2 measured on MI355X that a VALU write to the data registers of a `buffer_store_dwordx4` in the instruction right
after it corrupts lanes of the stored data, also in the SGPR-`soffset` form the compiler's hazard recogniser exempts
(falcon-r1cs_amd/csrc/frw_kernels.hip::tile_store).  The kernels therefore keep the data registers allocated for two
wait states after every such store; this test re-derives that property from the assembly hipcc emits for the
committed source, so a refactor that drops it fails here and not as sporadic wrong witnesses on the GPU.
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"
WAIT_STATES = 2


def _written_vgprs(line):
    """VGPR numbers an instruction writes (first operand of vector ALU / LDS-return / load instructions)."""
    m = re.match(r"\s*(v_\w+|ds_read\w*|ds_bpermute\w*|ds_swizzle\w*|global_load\w*|buffer_load\w*|flat_load\w*|scratch_load\w*)\s+([^,\s]+)", line)
    if not m:
        return set()
    op, dst = m.group(1), m.group(2)
    if op.startswith(("v_cmp", "v_readlane", "v_readfirstlane")):
        return set()
    r = re.match(r"v\[(\d+):(\d+)\]", dst)
    if r:
        return set(range(int(r.group(1)), int(r.group(2)) + 1))
    r = re.match(r"v(\d+)$", dst)
    return {int(r.group(1))} if r else set()


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_buffer_store_data_registers_survive_two_wait_states(tmp_path):
    asm = tmp_path / "frw_kernels.s"
    src = os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_kernels.hip")
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", str(asm), src],
                          stderr=subprocess.DEVNULL)
    lines = [l for l in open(asm).read().splitlines()
             if l.strip() and not l.strip().startswith((";", ".", "//")) and not l.rstrip().endswith(":")]
    stores = 0
    for i, line in enumerate(lines):
        m = re.match(r"\s*buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\]", line)
        if not m:
            continue
        stores += 1
        data = set(range(int(m.group(1)), int(m.group(2)) + 1))
        slots, j = 0, i + 1
        while slots < WAIT_STATES and j < len(lines):
            nxt = lines[j]
            nop = re.match(r"\s*s_nop\s+(\d+)", nxt)
            if nop:
                slots += int(nop.group(1)) + 1
            else:
                hit = _written_vgprs(nxt) & data
                assert not hit, "data register(s) v%s of\n  %s\nrewritten %d wait state(s) later by\n  %s" % (
                    sorted(hit), line.strip(), slots + 1, nxt.strip())
                if re.match(r"\s*(s_endpgm|s_branch|s_cbranch)", nxt):
                    break
                slots += 1
            j += 1
    assert stores > 100          # the tile writer of every kernel instantiation is in there


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_field_product_instruction_count_is_what_the_roofline_prices(tmp_path):
    """bench.py prices the QAP transforms' VALU-issue roofline with the instructions of ONE f29_mul (frw_fr29.h): 153
    v_mad_u64_u32 and 77 other vector instructions.  Re-derived here from the assembly hipcc emits for a kernel that is
    nothing but one product (loads and stores are not vector-ALU instructions), so the constant cannot go stale."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    src = tmp_path / "one_mul.hip"
    src.write_text('#include "frw_fr29.h"\n'
                   'using namespace frw;\n'
                   '__global__ void one_mul(const uint32_t *a, const uint32_t *b, uint32_t *out)\n'
                   '{\n'
                   '    F29 x, y;\n'
                   '    for (int k = 0; k < NL29; k++) { x.l[k] = a[threadIdx.x * NL29 + k]; y.l[k] = b[threadIdx.x * NL29 + k]; }\n'
                   '    const F29 r = f29_mul(x, y);\n'
                   '    for (int k = 0; k < NL29; k++) out[threadIdx.x * NL29 + k] = r.l[k];\n'
                   '}\n')
    asm = tmp_path / "one_mul.s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                           "-I", os.path.join(ROOT, "falcon-r1cs_amd", "csrc"), "-o", str(asm), str(src)], stderr=subprocess.DEVNULL)
    body = open(asm).read().split("one_mul")
    text = open(asm).read()
    start = text.index("_Z7one_mulPKjS0_Pj:")
    code = text[start:text.index("s_endpgm", start)]
    valu = [l.split()[0] for l in code.splitlines() if re.match(r"\s*v_", l)]
    mads = sum(1 for op in valu if op.startswith("v_mad_u64_u32"))
    other = len(valu) - mads
    assert mads == bench.F29_MUL_MAD64, mads
    # address arithmetic of the three pointers is in `other` too: allow a handful beyond the product's own
    assert bench.F29_MUL_OTHER <= other <= bench.F29_MUL_OTHER + 16, other


@pytest.mark.skipif(not os.path.exists(HIPCC), reason="hipcc not installed")
def test_fq_product_instruction_count_is_what_the_msm_roofline_prices(tmp_path):
    """The same for the 381-bit field of the multi-scalar multiplication (frw_fq29.h): fq_mul = 392 v_mad_u64_u32 + 14 v_mul_lo_u32
    and 90 other vector instructions; fq_sqr and fq_mul_sub (the other two operations the point formulas are made of) likewise."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    def count(name, body):
        src = tmp_path / ("one_%s.hip" % name)
        src.write_text('#include "frw_fq29.h"\n'
                       'using namespace frw;\n'
                       '__global__ void one_op(const uint32_t *a, const uint32_t *b, uint32_t *out)\n'
                       '{\n'
                       '    Fq29 x, y;\n'
                       '    for (int k = 0; k < NLQ; k++) { x.l[k] = a[threadIdx.x * NLQ + k]; y.l[k] = b[threadIdx.x * NLQ + k]; }\n'
                       '    %s\n'
                       '    for (int k = 0; k < NLQ; k++) out[threadIdx.x * NLQ + k] = r.l[k];\n'
                       '}\n' % body)
        asm = tmp_path / ("one_%s.s" % name)
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
                               "-I", os.path.join(ROOT, "falcon-r1cs_amd", "csrc"), "-o", str(asm), str(src)], stderr=subprocess.DEVNULL)
        text = open(asm).read()
        start = text.index("_Z6one_opPKjS0_Pj:")
        code = text[start:text.index("s_endpgm", start)]
        valu = [l.split()[0] for l in code.splitlines() if re.match(r"\s*v_", l)]
        mult = sum(1 for op in valu if op.startswith(("v_mad_u64_u32", "v_mul_lo_u32")))
        return mult, len(valu) - mult

    mult, other = count("mul", "const Fq29 r = fq_mul(x, y);")
    assert mult == bench.FQ_MUL_MULTIPLY, mult
    assert bench.FQ_MUL_OTHER <= other <= bench.FQ_MUL_OTHER + 16, other
    # the dedicated square: 105 + 14 + 196
    mult, other = count("sqr", "const Fq29 r = fq_sqr(x);")
    assert mult == bench.FQ_SQR_MULTIPLY == 105 + 14 + 196, mult
    assert bench.FQ_SQR_OTHER <= other <= bench.FQ_SQR_OTHER + 16, other
    # a b - c d with one reduction: 2 x 196 + 14 + 196
    mult, other = count("pair", "Fq29 z, w; for (int k = 0; k < NLQ; k++) { z.l[k] = a[(threadIdx.x + 64) * NLQ + k]; "
                                "w.l[k] = b[(threadIdx.x + 64) * NLQ + k]; } const Fq29 r = fq_mul_sub<16>(x, y, z, w);")
    assert mult == bench.FQ_PAIR_MULTIPLY == 2 * 196 + 14 + 196, mult
    assert bench.FQ_PAIR_OTHER <= other <= bench.FQ_PAIR_OTHER + 16, other
