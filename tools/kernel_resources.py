#!/usr/bin/env python3
"""What the compiler gave every kernel: VGPRs, AGPRs, SGPRs, scratch (bytes per lane), LDS, waves per SIMD -- hipcc
-Rpass-analysis=kernel-resource-usage, gfx950.  No GPU needed.

    python tools/kernel_resources.py --full > profiles/rNN_kernel_resources.txt     every kernel of every .hip (five minutes)
    python tools/kernel_resources.py --hot                                          the hot list only, through probe units (a minute)

The hot list = the kernels the benchmark's lines spend their time in; tests/test_kernel_resources.py compiles it and fails when one of
them reports ScratchSize > 0 or fewer waves per SIMD than it is written for (round 4's VERDICT: scratch had crept back into the
transform passes -- 28 to 112 bytes per lane -- and nothing in the tree noticed)."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "falcon-r1cs_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FILES = ("frw_kernels.hip", "frw_prepare.hip", "frw_r1cs_check.hip", "frw_qap.hip", "frw_setup.hip", "frw_msm.hip")

# (probe source, [(explicit instantiation, scratch allowed, waves per SIMD at least)])
QAP_PASSES = [("PASS_FIRST", 6, "LOAD_PLAIN", "STORE_FACTOR"), ("PASS_FIRST", 6, "LOAD_PRODUCTS_AB", "STORE_FACTOR"),
              ("PASS_DIT_SH0", 6, "LOAD_AB", "STORE_FACTOR"), ("PASS_DIT_SH0", 6, "LOAD_AB_MINUS_C", "STORE_FACTOR"),
              ("PASS_DIF_SH0", 6, "LOAD_PLAIN", "STORE_PLAIN")] + [
    (mode, t, "LOAD_PLAIN", store) for t in (6, 5, 4) for mode, store in (
        ("PASS_DIT", "STORE_FACTOR"), ("PASS_DIF", "STORE_FACTOR"), ("PASS_DIT_DIF", "STORE_FACTOR"), ("PASS_DIT_DIF", "STORE_FACTOR_A"),
        ("PASS_DIT", "STORE_FACTOR_CANONICAL"), ("PASS_DIT", "STORE_CONST_ADD_CANONICAL"))]
HOT = {
    "frw_qap.hip": ("FRW_QAP_PROBE", ["template __global__ void ntt_pass_kernel<%s, %d, %s, %s, false>(const NttPass);" % p for p in QAP_PASSES], 0, 3),
    "frw_msm.hip": ("FRW_MSM_PROBE", ["template __global__ void msm_bucket_kernel<FqField, true>(MsmDev, const uint32_t *, const uint32_t *, const uint32_t *, "
                                      "const uint32_t *, const uint32_t *, uint32_t *, uint32_t, uint32_t, size_t, const unsigned long long *);",
                                      "template __global__ void nmsm_bucket_kernel<FqField, true>(NmsmTables, const uint32_t *, const uint32_t *, const uint32_t *, "
                                      "const uint32_t *, const uint32_t *, uint32_t *, uint32_t, uint32_t, const unsigned long long *);"], 0, 2),
}
# kernels of the files compiled whole (no templates to pick from): name prefix -> (scratch allowed, waves at least)
HOT_WHOLE = {"frw_kernels.hip": {"frw::witness_ntt_verify_kernel<10, 1>": (0, 1), "frw::witness_ntt_verify_kernel<9, 1>": (0, 1),
                                 "frw::ntt_modq_kernel<9, 1>": (0, 1)},
             # (the flattened rows' kernel wants 131 registers and is written for FOUR waves per SIMD, 128 each: three of them live in scratch
             # memory -- a latency-bound kernel, the fourth wave is worth more than the twelve bytes; more than that is a regression)
             "frw_r1cs_check.hip": {"frw::r1cs_eval_flat_kernel<true>": (12, 4), "frw::r1cs_long_rows_small_kernel": (0, 2)}}

FIELDS = (("VGPRs", "vgpr"), ("AGPRs", "agpr"), ("SGPRs", "sgpr"), ("ScratchSize [bytes/lane]", "scratch"), ("Occupancy [waves/SIMD]", "waves"),
          ("LDS Size [bytes/block]", "lds"))


def compile_report(path, extra=()):
    """[{name, vgpr, agpr, sgpr, scratch, waves, lds}] for every kernel the unit at `path` emits"""
    cmd = [HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-w", "-I", CSRC, "-c", path, "-o", os.devnull,
           "-Rpass-analysis=kernel-resource-usage"] + list(extra)
    err = subprocess.run(cmd, capture_output=True, text=True, cwd=CSRC).stderr
    kernels, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark: .*?Function Name: (\S+)", line)
        if m:
            cur = {"name": m.group(1)}
            kernels.append(cur)
            continue
        for label, key in FIELDS:
            m = re.search(re.escape(label) + r": (\d+)", line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
    if kernels:
        names = subprocess.run(["c++filt"], input="\n".join(k["name"] for k in kernels), capture_output=True, text=True).stdout.splitlines()
        for k, nm in zip(kernels, names):
            k["name"] = re.sub(r"^void ", "", nm)
    if not kernels:
        raise RuntimeError("no resource remarks from %s:\n%s" % (path, err[-2000:]))
    return kernels


def hot_report():
    """(rows, violations) for the hot list"""
    rows, bad = [], []
    with tempfile.TemporaryDirectory() as tmp:
        for src, (macro, insts, scratch_ok, waves_min) in HOT.items():
            probe = os.path.join(tmp, "probe_" + src)
            with open(probe, "w") as f:
                f.write("#define %s 1\n#include \"%s\"\nnamespace frw {\n%s\n}\n" % (macro, src, "\n".join(insts)))
            for k in compile_report(probe):
                if not any(t in k["name"] for t in ("ntt_pass_kernel<", "msm_bucket_kernel<")):
                    continue
                rows.append((src, k))
                if k["scratch"] > scratch_ok or k["waves"] < waves_min:
                    bad.append((src, k))
    for src, wanted in HOT_WHOLE.items():
        found = set()
        for k in compile_report(os.path.join(CSRC, src)):
            for prefix, (scratch_ok, waves_min) in wanted.items():
                if k["name"].startswith(prefix):
                    found.add(prefix)
                    rows.append((src, k))
                    if k["scratch"] > scratch_ok or k["waves"] < waves_min:
                        bad.append((src, k))
        missing = set(wanted) - found
        if missing:
            raise RuntimeError("%s: hot kernels not found: %s" % (src, sorted(missing)))
    return rows, bad


def fmt(src, k):
    return "%-20s %5d %5d %5d %8d %6d %7d  %s" % (src, k.get("vgpr", -1), k.get("agpr", -1), k.get("sgpr", -1), k.get("scratch", -1),
                                                   k.get("waves", -1), k.get("lds", -1), k["name"])


HEADER = "%-20s %5s %5s %5s %8s %6s %7s  %s" % ("file", "VGPR", "AGPR", "SGPR", "scratch", "waves", "LDS", "kernel")


def main():
    if "--hot" in sys.argv:
        rows, bad = hot_report()
        print(HEADER)
        for src, k in rows:
            print(fmt(src, k))
        if bad:
            print("\nHOT KERNELS WITH SCRATCH OR TOO FEW WAVES:")
            for src, k in bad:
                print(fmt(src, k))
            sys.exit(1)
        return
    print("# hipcc -Rpass-analysis=kernel-resource-usage, gfx950, every kernel of falcon-r1cs_amd/csrc/*.hip (tools/kernel_resources.py --full)")
    print("# scratch: bytes per lane; waves: per SIMD; LDS: bytes per workgroup")
    print(HEADER)
    for src in FILES:
        for k in sorted(compile_report(os.path.join(CSRC, src)), key=lambda k: k["name"]):
            print(fmt(src, k))


if __name__ == "__main__":
    main()
