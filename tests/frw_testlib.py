"""Shared helpers of the test-suite: loading the ORACLE (oracle/) and drawing inputs.

The oracle is the checker.  Nothing here is imported by the product package.
"""
import ctypes as C
import os
import random
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
Q = 12289
SIGMA = {9: 165.7366, 10: 168.3886}
SIG_L2_BOUND = {9: 34034726, 10: 70265242}


class OracleLayout(C.Structure):
    _fields_ = [("logn", C.c_int32), ("n", C.c_int32), ("num_witness", C.c_int32), ("num_instance", C.c_int32),
                ("num_constraints", C.c_int32), ("seg_off", C.c_int32 * 8), ("seg_len", C.c_int32 * 8)]


class Oracle:
    """ctypes face of oracle/libfrw_oracle.so (plain-C closed-form restatement)."""

    def __init__(self, path):
        lib = C.CDLL(path)
        lib.frw_oracle_layout.argtypes = [C.c_int, C.POINTER(OracleLayout)]
        lib.frw_oracle_witness_ntt_verify.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 3 + [C.c_int]
        lib.frw_oracle_ntt_modq.argtypes = [C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]
        lib.frw_oracle_ntt_clear.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.frw_oracle_witness_dual_ntt_verify.argtypes = [C.c_int, C.c_size_t] + [C.c_void_p] * 3 + [C.c_int] + \
            [C.c_void_p] * 3
        lib.frw_oracle_digest.argtypes = [C.c_void_p, C.c_size_t]
        lib.frw_oracle_digest.restype = C.c_uint64
        lib.frw_oracle_qap_domain_log.argtypes = [C.c_uint64, C.c_uint64]
        lib.frw_oracle_qap_matvec.argtypes = [C.c_uint64] + [C.c_void_p] * 4 + [C.c_uint64, C.c_void_p]
        lib.frw_oracle_qap_matvec.restype = None
        lib.frw_oracle_qap_witness_map.argtypes = [C.c_void_p] * 3 + [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        lib.frw_oracle_qap_product_high_half.argtypes = [C.c_void_p] * 2 + [C.c_uint64, C.c_uint64, C.c_void_p, C.c_void_p]
        lib.frw_oracle_g1_generator.argtypes = [C.c_void_p]
        lib.frw_oracle_g1_on_curve.argtypes = [C.c_void_p]
        lib.frw_oracle_g1_scalar_mul.argtypes = [C.c_void_p] * 3
        lib.frw_oracle_g1_add.argtypes = [C.c_void_p] * 3
        lib.frw_oracle_g1_fixed_base.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_int]
        lib.frw_oracle_g1_msm.argtypes = [C.c_size_t, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        for f in (lib.frw_oracle_g1_generator, lib.frw_oracle_g1_scalar_mul, lib.frw_oracle_g1_add, lib.frw_oracle_g1_fixed_base,
                  lib.frw_oracle_g1_msm):
            f.restype = None
        self.lib = lib

    # ---- BLS12-381 G1 (oracle/bls12_381.c): points are uint64[12] = ark-ff's Montgomery limbs of x then y, zeros = infinity
    def g1_generator(self):
        out = np.zeros(12, dtype=np.uint64)
        self.lib.frw_oracle_g1_generator(out.ctypes.data_as(C.c_void_p))
        return out

    def g1_on_curve(self, p):
        p = np.ascontiguousarray(p, dtype=np.uint64)
        return bool(self.lib.frw_oracle_g1_on_curve(p.ctypes.data_as(C.c_void_p)))

    def g1_scalar_mul(self, base, k):
        base = np.ascontiguousarray(base, dtype=np.uint64)
        kk = ints_to_limbs([int(k)])
        out = np.zeros(12, dtype=np.uint64)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.frw_oracle_g1_scalar_mul(P(base), P(kk), P(out))
        return out

    def g1_add(self, a, b):
        a, b = (np.ascontiguousarray(x, dtype=np.uint64) for x in (a, b))
        out = np.zeros(12, dtype=np.uint64)
        P = lambda x: x.ctypes.data_as(C.c_void_p)
        self.lib.frw_oracle_g1_add(P(a), P(b), P(out))
        return out

    def g1_fixed_base(self, scalars, threads=8):
        """k_i G1 for canonical scalars uint64[count, 4] -> uint64[count, 12]."""
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        out = np.zeros((scalars.shape[0], 12), dtype=np.uint64)
        P = lambda x: x.ctypes.data_as(C.c_void_p)
        self.lib.frw_oracle_g1_fixed_base(scalars.shape[0], P(scalars), P(out), threads)
        return out

    def g1_msm(self, bases, scalars, window_bits=13, threads=8):
        """sum k_i P_i (bucket method on the CPU): bases uint64[count, 12], canonical scalars uint64[count, 4] -> uint64[12]."""
        bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 12)
        scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
        assert bases.shape[0] == scalars.shape[0]
        out = np.zeros(12, dtype=np.uint64)
        P = lambda x: x.ctypes.data_as(C.c_void_p)
        self.lib.frw_oracle_g1_msm(bases.shape[0], P(bases), P(scalars), P(out), window_bits, threads)
        return out

    def qap_product_high_half(self, az, bz, num_inputs, z):
        """hi of a(X) b(X) = lo + X^n hi (what frw_qap_quotient_dev returns); uint64[., 4] canonical."""
        az, bz, z = (np.ascontiguousarray(a, dtype=np.uint64) for a in (az, bz, z))
        nc = az.shape[0]
        lg = self.lib.frw_oracle_qap_domain_log(nc, num_inputs)
        h = np.zeros((1 << lg, 4), dtype=np.uint64)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        assert self.lib.frw_oracle_qap_product_high_half(P(az), P(bz), nc, num_inputs, P(z), P(h)) == 0
        return h

    def qap_matvec(self, ptr, col, val, z):
        """One CSR matrix (canonical values) times z (uint64[vars, 4], canonical) -> uint64[rows, 4] canonical."""
        ptr = np.ascontiguousarray(ptr, dtype=np.uint64)
        col = np.ascontiguousarray(col, dtype=np.uint32)
        val = np.ascontiguousarray(val, dtype=np.uint64)
        z = np.ascontiguousarray(z, dtype=np.uint64)
        out = np.zeros((len(ptr) - 1, 4), dtype=np.uint64)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        self.lib.frw_oracle_qap_matvec(len(ptr) - 1, P(ptr), P(col), P(val), P(z), z.shape[0], P(out))
        return out

    def qap_witness_map(self, az, bz, cz, num_inputs, z):
        """ark-groth16's R1CStoQAP::witness_map from the three products on; everything uint64[., 4] canonical."""
        az, bz, cz, z = (np.ascontiguousarray(a, dtype=np.uint64) for a in (az, bz, cz, z))
        nc = az.shape[0]
        lg = self.lib.frw_oracle_qap_domain_log(nc, num_inputs)
        h = np.zeros((1 << lg, 4), dtype=np.uint64)
        P = lambda a: a.ctypes.data_as(C.c_void_p)
        assert self.lib.frw_oracle_qap_witness_map(P(az), P(bz), P(cz), nc, num_inputs, P(z), P(h)) == 0
        return h

    def layout(self, logn):
        L = OracleLayout()
        assert self.lib.frw_oracle_layout(logn, C.byref(L)) == 0
        return L

    def witness_ntt_verify(self, logn, sig, pk, hm, encoding=1, threads=1, out=None):
        """out = (wit, inst, st) reuses caller-owned (already touched) buffers of at least the batch size; slots of
        rejected signatures are then left as they were."""
        L = self.layout(logn)
        sig, pk, hm = (np.ascontiguousarray(a, dtype=np.uint16).reshape(-1, L.n) for a in (sig, pk, hm))
        batch = sig.shape[0]
        if out is not None:
            wit, inst, st = (a[:batch] for a in out)
        else:
            wit = np.zeros((batch, L.num_witness, 4), dtype=np.uint64)
            inst = np.zeros((batch, L.num_instance, 4), dtype=np.uint64)
            st = np.zeros(batch, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        rc = self.lib.frw_oracle_witness_ntt_verify(logn, batch, p(sig), p(pk), p(hm), encoding, p(wit), p(inst),
                                                    p(st), threads)
        assert rc == 0
        return wit, inst, st

    def witness_dual_ntt_verify(self, logn, sig, pk, hm, encoding=1):
        n = 1 << logn
        sig, pk, hm = (np.ascontiguousarray(a, dtype=np.uint16).reshape(-1, n) for a in (sig, pk, hm))
        batch = sig.shape[0]
        wit = np.zeros((batch, self.lib.frw_oracle_dual_num_witness(logn), 4), dtype=np.uint64)
        inst = np.zeros((batch, 2 * n + 1, 4), dtype=np.uint64)
        st = np.zeros(batch, dtype=np.int32)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert self.lib.frw_oracle_witness_dual_ntt_verify(logn, batch, p(sig), p(pk), p(hm), encoding, p(wit), p(inst),
                                                           p(st)) == 0
        return wit, inst, st

    def ntt_modq(self, logn, poly, encoding=1):
        n = 1 << logn
        poly = np.ascontiguousarray(poly, dtype=np.uint16).reshape(-1, n)
        batch = poly.shape[0]
        wit = np.zeros((batch, 29 * n, 4), dtype=np.uint64)
        out = np.zeros((batch, n), dtype=np.uint16)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        assert self.lib.frw_oracle_ntt_modq(logn, batch, p(poly), encoding, p(wit), p(out)) == 0
        return wit, out

    def ntt_clear(self, logn, poly, inverse=False):
        poly = np.ascontiguousarray(poly, dtype=np.uint16)
        out = np.zeros_like(poly)
        assert self.lib.frw_oracle_ntt_clear(logn, poly.ctypes.data_as(C.c_void_p), out.ctypes.data_as(C.c_void_p),
                                             1 if inverse else 0) == 0
        return out

    def digest(self, words):
        words = np.ascontiguousarray(words, dtype=np.uint64)
        return int(self.lib.frw_oracle_digest(words.ctypes.data_as(C.c_void_p), words.size))


def load_oracle():
    so = os.path.join(ORACLE_DIR, "libfrw_oracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("frw_oracle.c", "qap_oracle.c", "bls12_381.c")]
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "libfrw_oracle.so"])
    return Oracle(so)


def limbs_to_ints(a):
    """uint64[..., 4] little-endian limbs -> list of Python ints (row-major)."""
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int.from_bytes(r.tobytes(), "little") for r in a]


def ints_to_limbs(vals):
    return np.frombuffer(b"".join(int(v).to_bytes(32, "little") for v in vals), dtype=np.uint64).reshape(-1, 4).copy()


def negacyclic_mul(a, b):
    """Schoolbook product mod (x^N + 1, q) with numpy -- independent of every NTT in the repo."""
    n = len(a)
    a = np.asarray(a, dtype=np.int64)
    b = np.asarray(b, dtype=np.int64)
    full = np.convolve(a, b)                       # < 1024 * q^2 < 2^38
    res = full[:n].copy()
    res[: n - 1] -= full[n:]
    return np.mod(res, Q)


def random_triple(logn, rng: random.Random, scale=1.0):
    """Python-side synthetic (sig, pk, hm); scale > 1 inflates the norm (for the bound tests)."""
    n = 1 << logn
    s = SIGMA[logn] * scale
    sig = [round(rng.gauss(0, s)) % Q for _ in range(n)]
    v = [round(rng.gauss(0, s)) % Q for _ in range(n)]
    pk = [rng.randrange(Q) for _ in range(n)]
    hm = (np.array(v, dtype=np.int64) + negacyclic_mul(sig, pk)) % Q
    return (np.array(sig, dtype=np.uint16), np.array(pk, dtype=np.uint16), hm.astype(np.uint16),
            np.array(v, dtype=np.uint16))


def centred_norm(*polys):
    tot = 0
    for p in polys:
        x = np.asarray(p, dtype=np.int64)
        x = np.where(x < 6144, x, Q - x)
        tot += int((x * x).sum())
    return tot


class CompactLayoutPy:
    """include/frw.h's FRW_ENC_COMPACT layout restated (independent of the library; tests/test_capi.py compares the two)."""

    def __init__(self, logn):
        n = 1 << logn
        self.logn, self.n = logn, n
        self.small_off, self.num_small = 0, 11 * n
        self.t_off, self.num_t = 4 * self.num_small, 2 * n
        self.bits_off = self.t_off + 20 * self.num_t
        self.num_bit_words = 4 * (27 * n // 32) + n + 2
        self.bit_seg_off = (0, 27 * n, 54 * n, 81 * n, 108 * n, 140 * n)
        self.instance_off = self.bits_off + (4 * self.num_bit_words + 15) // 16 * 16
        self.num_instance_values = 2 * n
        self.status_off = self.instance_off + 4 * self.num_instance_values
        self.bytes_per_signature = (self.status_off + 4 + 127) // 128 * 128


P_FR = 0x73eda753299d7d483339d80809a1d80553bda402fffe5bfeffffffff00000001
R_INV = pow(1 << 256, -1, P_FR)


def _from_montgomery(limbs4):
    x = int(limbs4[0]) | int(limbs4[1]) << 64 | int(limbs4[2]) << 128 | int(limbs4[3]) << 192
    return x * R_INV % P_FR


def compact_from_witness(logn, wit, inst, CL=None, status=0):
    """FRW_ENC_COMPACT restated as a re-layout of one arkworks witness/instance pair (uint64 [W,4] / [I,4], Montgomery):
    the non-boolean elements as plain integers (11 N of them as uint32, the 2 N mod_q quotients as 5 x uint32), the boolean
    elements as a bit array in witness order, the instance values without the leading one as uint32, the status word
    (include/frw.h); everything else zero.
    CL = falcon_r1cs_amd.compact_layout(logn), or None for the restated layout above.  Returns bytes."""
    CL = CL or CompactLayoutPy(logn)
    n = 1 << logn
    nb = 50 if logn == 9 else 52
    off = np.cumsum([0, n, n, 27 * n, 29 * n, 29 * n, 30 * n, 36 * n])
    S2, S3, S4, S5, S6, S7 = (int(off[i]) for i in (2, 3, 4, 5, 6, 7))
    k = np.arange(n)
    k2 = np.arange(2 * n)
    small = np.concatenate([
        np.arange(0, 2 * n),
        S3 + 29 * k + 1,
        S4 + 29 * k + 1,
        (S5 + 30 * k[:, None] + np.arange(3)).ravel(),
        (S6 + 18 * k2[:, None] + 16 + np.arange(2)).ravel()])
    tq = np.concatenate([S3 + 29 * k, S4 + 29 * k])
    bools = np.concatenate([
        np.arange(S2, S2 + 27 * n),
        (S3 + 29 * k[:, None] + 2 + np.arange(27)).ravel(),
        (S4 + 29 * k[:, None] + 2 + np.arange(27)).ravel(),
        (S5 + 30 * k[:, None] + 3 + np.arange(27)).ravel(),
        (S6 + 18 * k2[:, None] + np.arange(16)).ravel()])
    assert len(small) == CL.num_small and len(tq) == CL.num_t and len(bools) == 140 * n
    out = np.zeros(CL.bytes_per_signature, dtype=np.uint8)
    sv = [_from_montgomery(wit[i]) for i in small]
    assert max(sv) < 1 << 32
    out[: 4 * len(sv)] = np.array(sv, dtype=np.uint32).view(np.uint8)
    tv = np.array([[(t >> (32 * j)) & 0xFFFFFFFF for j in range(5)] for t in (_from_montgomery(wit[i]) for i in tq)], dtype=np.uint32)
    out[CL.t_off: CL.t_off + tv.nbytes] = tv.view(np.uint8).ravel()
    bits = (wit[bools] != 0).any(axis=1)
    words = np.packbits(bits.astype(np.uint8), bitorder="little").view(np.uint32)
    tail = (wit[S7:S7 + nb] != 0).any(axis=1)
    tail = np.packbits(np.concatenate([tail, np.zeros(64 - nb, dtype=bool)]).astype(np.uint8), bitorder="little").view(np.uint32)
    allw = np.concatenate([words, tail])
    assert len(allw) == CL.num_bit_words
    out[CL.bits_off: CL.bits_off + 4 * len(allw)] = allw.view(np.uint8)
    iv = np.array([_from_montgomery(inst[i]) for i in range(1, 2 * n + 1)], dtype=np.uint32)
    out[CL.instance_off: CL.instance_off + iv.nbytes] = iv.view(np.uint8)
    out[CL.status_off: CL.status_off + 4] = np.array([status], dtype=np.uint32).view(np.uint8)
    return out.tobytes()
