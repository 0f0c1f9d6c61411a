"""CPU tests of the drop-in boundary: libfrw.so loads, exports every symbol include/frw.h declares, answers the
structural questions, and refuses to compute without a GPU (no CPU fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import falcon_r1cs_amd as frw
from falcon_r1cs_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "frw.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(frw_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = frw.load_library()
    names = declared_symbols()
    assert len(names) >= 17
    for name in names:
        assert hasattr(lib, name), name
    assert sorted(_lib.PROTOTYPES) == names          # the Python binding covers the whole header


def test_layout_matches_reference_counts():
    for logn, (i, w, c) in {9: (1025, 78386, 81460), 10: (2049, 156724, 162870)}.items():   # README.md:44,55
        L = frw.layout(logn)
        assert (L.num_instance, L.num_witness, L.num_constraints) == (i, w, c)
        n = L.n
        assert L.seg_len == (n, n, 27 * n, 29 * n, 29 * n, 30 * n, 36 * n, 50 if logn == 9 else 52)
        assert L.seg_off[0] == 0 and all(L.seg_off[k + 1] == L.seg_off[k] + L.seg_len[k] for k in range(7))
        assert L.seg_off[7] + L.seg_len[7] == w
    with pytest.raises(frw.FrwError):
        frw.layout(8)


def test_product_layout_equals_oracle_layout(oracle):
    for logn in (9, 10):
        a, b = frw.layout(logn), oracle.layout(logn)
        assert a.seg_off == tuple(b.seg_off) and a.seg_len == tuple(b.seg_len)


def test_no_device_means_error_not_fallback():
    lib = frw.load_library()
    if lib.frw_device_count() > 0:
        pytest.skip("a GPU is present; the refusal path is exercised on the CPU box")
    h = C.c_void_p()
    assert lib.frw_ctx_create(0, C.byref(h)) == -2                       # FRW_E_NO_DEVICE
    assert not h.value
    assert b"no CPU path" in lib.frw_strerror(-2)
    with pytest.raises(frw.FrwError) as ei:
        frw.WitnessEngine(0)
    assert ei.value.code == -2
    # compute entry points reject a null context instead of computing anything
    assert lib.frw_witness_ntt_verify(None, 10, 1, None, None, None, 1, None, None, None, 1) == -1
    assert lib.frw_ntt_modq_dev(None, 9, 1, None, 1, None, None, None, None) == -1


def test_product_does_not_reference_the_oracle():
    """The product sources must not include, import, link or call anything under oracle/."""
    pkg = os.path.join(ROOT, "falcon-r1cs_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    needed = os.popen("readelf -d %s" % frw.lib_path()).read()
    assert "frw_oracle" not in needed


def test_synth_rejects_bad_arguments():
    lib = frw.load_library()
    buf = np.zeros(512, dtype=np.uint16)
    p = buf.ctypes.data_as(C.c_void_p)
    assert lib.frw_synth_triples(8, 1, 0, 0, p, p, p) == -1
    assert lib.frw_synth_triples(9, 1, 0, 0, None, p, p) == -1


def test_header_is_plain_c99(tmp_path):
    """include/frw.h compiles with gcc -std=c99 -pedantic and the library links from C."""
    import subprocess
    exe = str(tmp_path / "hdr")
    subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-o", exe,
                           os.path.join(ROOT, "tests", "c", "test_header_c99.c"),
                           "-L" + os.path.join(ROOT, "falcon-r1cs_amd"), "-lfrw",
                           "-Wl,-rpath," + os.path.join(ROOT, "falcon-r1cs_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.returncode, out.stdout, out.stderr)


def test_expand_host_inverts_the_compact_relayout(oracle):
    """frw_expand_host (host-side, no device): compact bytes built by re-laying out the oracle's witness
    (tests/frw_testlib.compact_from_witness, written from include/frw.h's description) expand back to exactly that
    witness / instance -- two independent statements of FRW_ENC_COMPACT agree."""
    import ctypes as C
    import numpy as np
    import falcon_r1cs_amd as frw
    import frw_testlib as T
    lib = frw.load_library()
    for logn in (9, 10):
        CL = frw.compact_layout(logn)
        assert CL.bytes_per_signature % 128 == 0 and CL.num_small == 11 << logn and CL.num_t == 2 << logn
        py = T.CompactLayoutPy(logn)
        for f in ("bytes_per_signature", "small_off", "num_small", "t_off", "num_t", "bits_off", "num_bit_words",
                  "bit_seg_off", "instance_off", "num_instance_values", "status_off"):
            assert getattr(CL, f) == getattr(py, f), f
        sig, pk, hm = frw.synth_triples(logn, 3, seed=77)
        wit, inst, st = oracle.witness_ntt_verify(logn, sig, pk, hm, 1)
        comp = np.stack([np.frombuffer(T.compact_from_witness(logn, wit[k], inst[k], CL), dtype=np.uint8) for k in range(3)])
        w2, i2 = np.empty_like(wit), np.empty_like(inst)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert lib.frw_expand_host(logn, 3, p(comp), p(w2), p(i2)) == 0
        assert np.array_equal(w2, wit) and np.array_equal(i2, inst)
    assert lib.frw_expand_host(11, 1, None, None, None) == -1


def test_expand_host_montgomery_conversion_on_random_integers():
    """frw_expand_host on a compact buffer filled with random integers of the full documented ranges (32-bit values,
    160-bit quotients, random booleans): every output element == x * 2^256 mod p computed with Python integers."""
    import ctypes as C
    import random
    import numpy as np
    import falcon_r1cs_amd as frw
    import frw_testlib as T
    lib = frw.load_library()
    rng = random.Random(2026)
    logn = 9
    CL, L = T.CompactLayoutPy(logn), frw.layout(logn)
    n = 1 << logn
    buf = np.zeros(CL.bytes_per_signature, dtype=np.uint8)
    small = [rng.choice([0, 1, 12288, (1 << 28) - 1, (1 << 32) - 1, rng.getrandbits(32)]) for _ in range(CL.num_small)]
    tq = [rng.choice([0, 1, (1 << 160) - 1, (1 << 146) - 1, rng.getrandbits(160)]) for _ in range(CL.num_t)]
    bits = [rng.getrandbits(1) for _ in range(140 * n)] + [rng.getrandbits(1) for _ in range(50)]
    inst = [rng.getrandbits(32) for _ in range(2 * n)]
    buf[:4 * len(small)] = np.array(small, dtype=np.uint32).view(np.uint8)
    buf[CL.t_off:CL.t_off + 20 * len(tq)] = np.array([[(t >> (32 * j)) & 0xFFFFFFFF for j in range(5)] for t in tq], dtype=np.uint32).view(np.uint8).ravel()
    words = np.packbits(np.array(bits[:140 * n], dtype=np.uint8), bitorder="little").view(np.uint32)
    tail = np.packbits(np.array(bits[140 * n:] + [0] * 14, dtype=np.uint8), bitorder="little").view(np.uint32)
    allw = np.concatenate([words, tail])
    buf[CL.bits_off:CL.bits_off + 4 * len(allw)] = allw.view(np.uint8)
    buf[CL.instance_off:CL.instance_off + 8 * n] = np.array(inst, dtype=np.uint32).view(np.uint8)
    wit = np.empty((1, L.num_witness, 4), dtype=np.uint64)
    ins = np.empty((1, L.num_instance, 4), dtype=np.uint64)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    assert lib.frw_expand_host(logn, 1, p(buf), p(wit), p(ins)) == 0
    R = (1 << 256) % T.P_FR
    to_int = lambda l: int(l[0]) | int(l[1]) << 64 | int(l[2]) << 128 | int(l[3]) << 192
    mont = lambda x: x * R % T.P_FR
    # expected element list in witness order
    it_small, it_t, it_bit = iter(small), iter(tq), iter(bits)
    want = [mont(next(it_small)) for _ in range(2 * n)]
    want += [mont(next(it_bit)) for _ in range(27 * n)]
    seg = []
    b_s3 = [next(it_small) for _ in range(n)]
    b_s4 = [next(it_small) for _ in range(n)]
    for bs in (b_s3, b_s4):
        for k in range(n):
            seg += [mont(next(it_t)), mont(bs[k])] + [mont(next(it_bit)) for _ in range(27)]
    want += seg
    for k in range(n):
        want += [mont(next(it_small)) for _ in range(3)] + [mont(next(it_bit)) for _ in range(27)]
    for k in range(2 * n):
        want += [mont(next(it_bit)) for _ in range(16)] + [mont(next(it_small)) for _ in range(2)]
    want += [mont(next(it_bit)) for _ in range(50)]
    got = [to_int(wit[0, i]) for i in range(L.num_witness)]
    assert got == want
    assert [to_int(ins[0, i]) for i in range(L.num_instance)] == [R] + [mont(x) for x in inst]
