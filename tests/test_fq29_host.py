"""The fourteen-limb Fq arithmetic and the XYZZ point formulas of the MSM kernels (falcon-r1cs_amd/csrc/frw_fq29.h),
compiled for the host through the test-only shim (tests/cpp/hip_host) and checked against Python integers and the G1 oracle
(oracle/bls12_381.py): products at the lazy bounds the formulas rely on, differences with their K q offsets, the constants
of the header, conversion from and to ark-ff's bytes, inversion, the zero test, and every branch of the point additions."""
import ctypes as C
import os
import random
import subprocess

import pytest

from oracle import bls12_381 as E

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
Q = E.Q
RQ = 1 << 406
M29 = (1 << 29) - 1


@pytest.fixture(scope="module")
def lib():
    if os.environ.get("FRW_TEST_FQ29_SO"):           # tools/sanitize_cpu.sh: its own ASan / UBSan build, at a path of its own
        return C.CDLL(os.environ["FRW_TEST_FQ29_SO"])
    out = os.path.join(HERE, "cpp", "build")
    os.makedirs(out, exist_ok=True)
    so = os.path.join(out, "libtest_fq29.so")
    src = os.path.join(HERE, "cpp", "test_fq29.cpp")
    hdr = os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_fq29.h")
    hdr2 = os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_quad.h")
    if not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in (src, hdr, hdr2)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-Wno-unknown-pragmas", "-fno-strict-aliasing", "-I", os.path.join(HERE, "cpp", "hip_host"),
                               "-I", os.path.join(ROOT, "falcon-r1cs_amd", "csrc"), "-o", so, src])
    return C.CDLL(so)


def limbs(v, n=14):
    assert v >> (29 * (n - 1) + 32) == 0
    return (C.c_uint32 * n)(*([(v >> (29 * i)) & M29 for i in range(n - 1)] + [v >> (29 * (n - 1))]))


def words(v, n=12):
    return (C.c_uint32 * n)(*[(v >> (32 * i)) & 0xFFFFFFFF for i in range(n)])


def value(arr, bits=29):
    return sum(int(x) << (bits * i) for i, x in enumerate(arr))


def fq(lib, op, a, b=0, out_words=14):
    out = (C.c_uint32 * 24)()
    lib.t_fq(op, a, b if not isinstance(b, int) else limbs(b), out)
    return out[:out_words]


def test_constants_of_the_header():
    src = open(os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_fq29.h")).read()
    import re

    def table(name):
        body = re.search(name + r" = \{\{([^}]*)\}\}", src).group(1)
        return value([int(x.strip().rstrip("u"), 16) for x in body.split(",")])
    assert table("FQ29_ONE") == RQ % Q
    assert table("FQ29_C_IN") == pow(2, 428, Q) and table("FQ29_C_OUT") == pow(2, 384, Q) and table("FQ29_R3") == pow(2, 3 * 406, Q)
    assert (table("G1_GEN_X29"), table("G1_GEN_Y29")) == (E.G1[0] * RQ % Q, E.G1[1] * RQ % Q)
    # the endomorphism constant: a primitive cube root of unity, and the one that belongs to lambda = z^2 - 1
    beta = table("G1_ENDO_BETA29") * pow(RQ, -1, Q) % Q
    lam = E.Z_BLS ** 2 - 1
    assert beta != 1 and pow(beta, 3, Q) == 1 and E.R == lam * lam + lam + 1 and lam.bit_length() == 128
    assert E.mul(E.G1, lam) == (beta * E.G1[0] % Q, E.G1[1])
    assert int(re.search(r"QINV29 = (0x[0-9a-f]+)u", src).group(1), 16) == (-pow(Q, -1, 1 << 29)) % (1 << 29)
    q_words = [int(x.rstrip("u"), 16) for x in re.search(r"Q32_\[12\] = \{([^}]*)\}", src, re.S).group(1).replace("\n", " ").replace(" ", "").split(",")]
    assert value(q_words, 32) == Q


def test_field_operations_at_their_bounds(lib):
    rng = random.Random(41)
    inv = pow(RQ, -1, Q)
    edge = [0, 1, Q - 1, Q, 2 * Q - 1]
    for _ in range(300):
        a = rng.choice(edge + [rng.randrange(64 * Q)] * 5)
        b = rng.choice(edge + [rng.randrange(32 * Q)] * 5)
        got = fq(lib, 0, limbs(a), limbs(b))
        v = value(got)
        assert all(x <= M29 for x in got[:13]) and v % Q == a * b * inv % Q and v < 2 * Q and v < (a * b >> 406) + Q + 1
    # the dedicated square (105 multiplications) and a b - c d with one reduction, incl. all limbs at their maximum
    top = [M29] * 13 + [(16 * Q) >> (29 * 13)]
    for k in range(300):
        a = value(top) if k == 0 else rng.choice(edge + [rng.randrange(16 * Q)] * 5)
        got = fq(lib, 14, limbs(a))
        v = value(got)
        assert all(x <= M29 for x in got[:13]) and v % Q == a * a * inv % Q and v < (a * a >> 406) + Q + 1
    out = (C.c_uint32 * 14)()
    for k in range(300):
        which = k & 1
        kq = (16 if which else 4) * Q
        a, b, d = (value(top) if k < 2 else rng.randrange(16 * Q) for _ in range(3))
        c = kq - 1 if k < 4 else rng.randrange(kq)
        lib.t_fq_mul_sub(which, limbs(a), limbs(b), limbs(c), limbs(d), out)
        v = value(list(out))
        assert all(x <= M29 for x in list(out)[:13]) and v % Q == (a * b - c * d) * inv % Q and v < ((a * b + kq * d) >> 406) + Q + 1 and v < 2 * Q
    # an un-normalised left operand (limbs up to 2^30 - 1): the header allows it for one operand of one product
    for k in range(50):
        la = [(1 << 30) - 1] * 13 + [(1 << 12) - 1] if k == 0 else [rng.randrange(1 << 30) for _ in range(13)] + [rng.randrange(1 << 12)]
        a = value(la)
        b = value([M29] * 13 + [(2 * Q) >> (29 * 13)]) if k == 0 else rng.randrange(2 * Q)
        got = fq(lib, 0, (C.c_uint32 * 14)(*la), limbs(b))
        assert value(got) % Q == a * b * inv % Q and all(x <= M29 for x in got[:13])
    for _ in range(200):
        a, b = rng.randrange(18 * Q), rng.randrange(18 * Q)
        assert value(fq(lib, 1, limbs(a), limbs(b))) == a + b
        assert value(fq(lib, 2, limbs(a), limbs(b % (4 * Q)))) == a - b % (4 * Q) + 4 * Q
        assert value(fq(lib, 3, limbs(a), limbs(b % (16 * Q)))) == a - b % (16 * Q) + 16 * Q
        assert value(fq(lib, 4, limbs(a), limbs(b))) == a - b + 64 * Q
        assert value(fq(lib, 11, limbs(b % (4 * Q)))) == 4 * Q - b % (4 * Q)
        assert value(fq(lib, 6, limbs(a))) == a % Q
    for a in (0, 1, Q - 1, Q, Q + 1, 2 * Q - 1):
        assert value(fq(lib, 5, limbs(a))) == a % Q
    assert value(fq(lib, 2, limbs(0), limbs(4 * Q - 1))) == 1 and value(fq(lib, 3, limbs(0), limbs(16 * Q - 1))) == 1
    # zero test: every multiple of q below 64 q is zero, neighbours and random values are not
    for k in range(64):
        assert fq(lib, 10, limbs(k * Q))[0] == 1
        assert fq(lib, 10, limbs(k * Q + 1))[0] == 0 and (k == 0 or fq(lib, 10, limbs(k * Q - 1))[0] == 0)
    assert all(fq(lib, 10, limbs(rng.randrange(1, Q)))[0] == 0 for _ in range(200))
    # a value that passes the two-instruction filter without being a multiple of q
    fake = 5 * Q + (1 << 29) * 12345
    assert (fake & M29) == (5 * Q) & M29 and fq(lib, 10, limbs(fake))[0] == 0


def test_conversions_and_inverse(lib):
    rng = random.Random(43)
    for x in [0, 1, Q - 1] + [rng.randrange(Q) for _ in range(40)]:
        ark = x * E.FQ_R % Q
        assert value(fq(lib, 12, words(ark))) == ark                              # unpack
        assert value(fq(lib, 13, limbs(ark), out_words=12), 32) == ark            # pack
        mont = fq(lib, 7, words(ark))
        assert value(mont) % Q == x * RQ % Q and value(mont) < 2 * Q
        back = fq(lib, 8, (C.c_uint32 * 14)(*mont), out_words=12)
        assert value(back, 32) == ark
        if x:
            got = value(fq(lib, 9, (C.c_uint32 * 14)(*mont)))
            assert got % Q == pow(x, -1, Q) * RQ % Q and got < 2 * Q                  # binary extended Euclid (fq_inv)
            assert value(fq(lib, 15, (C.c_uint32 * 14)(*mont))) % Q == got % Q        # Fermat
    # lazily reduced operands (a sum leaves ZZ ZZZ < 2 q; anything below 2^12 q must do), the extremes of the Euclidean rounds
    for x in [2, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 1 << 380, (1 << 380) - 1, 3]:
        for k in (0, 1, 100, 4000):
            a = x * RQ % Q + k * Q
            got = value(fq(lib, 9, limbs(a)))
            assert got % Q == pow(x, -1, Q) * RQ % Q and got < 2 * Q, (x, k)


def test_inverse_thirty_one_rounds_at_a_time(lib):
    """fq_inv's fast path (frw_fq29.h fq_inv_words31: Pornin's batched binary GCD) on canonical words: the inverse, and that the path
    FINISHES (b == 1 after 25 x 31 rounds) -- for the values whose Euclidean rounds are extreme, for values around the word boundaries of
    the stand-ins, and for a few thousand random ones; zero is refused (the caller falls back and returns zero); its constant."""
    assert fq(lib, 18, limbs(0))[0] == (-pow(Q, -1, 1 << 31)) % (1 << 31)
    rng = random.Random(4343)
    special = [1, 2, 3, Q - 1, Q - 2, (Q - 1) // 2, (Q + 1) // 2, 1 << 380, (1 << 380) - 1, (1 << 64) - 1, 1 << 64, (1 << 64) + 1,
               (1 << 31) - 1, 1 << 31, (1 << 32) + 1, (1 << 96) - 1, Q >> 1, Q >> 33, (Q >> 64) + 1]
    special += [(1 << k) % Q for k in range(1, 381, 7)] + [Q - (1 << k) for k in range(0, 380, 11)]
    for x in special + [rng.randrange(1, Q) for _ in range(3000)] + [rng.randrange(1, 1 << rng.randrange(1, 381)) for _ in range(1000)]:
        out = (C.c_uint32 * 14)()
        lib.t_fq(17, words(x), None, out)
        assert out[12] == 1, "the batched rounds did not end in b == 1 for %x" % x
        assert value(list(out)[:12], 32) == pow(x, -1, Q), hex(x)
    out = (C.c_uint32 * 14)()
    lib.t_fq(17, words(0), None, out)
    assert out[12] == 0
    # the fallback is still the inverse, and fq_inv agrees with it whatever path it took
    for x in [1, Q - 1] + [rng.randrange(1, Q) for _ in range(20)]:
        mont = limbs(x * RQ % Q)
        assert value(fq(lib, 16, mont)) % Q == value(fq(lib, 9, mont)) % Q == pow(x, -1, Q) * RQ % Q
    assert value(fq(lib, 9, limbs(0))) % Q == 0


def _pt(p):
    return (C.c_uint32 * 24)(*[(l >> (32 * h)) & 0xFFFFFFFF for l in E.to_limbs(p) for h in range(2)])


def _g1(lib, op, p, q=None):
    out = (C.c_uint32 * 24)()
    lib.t_g1(op, _pt(p), _pt(q), out)
    return E.from_limbs([out[2 * i] | out[2 * i + 1] << 32 for i in range(12)])


def test_point_formulas_every_branch(lib):
    rng = random.Random(47)
    for _ in range(4):
        p, q = E.mul(E.G1, rng.randrange(1, E.R)), E.mul(E.G1, rng.randrange(1, E.R))
        assert _g1(lib, 0, p, q) == E.add(p, q)
        assert _g1(lib, 1, p) == E.add(p, p)
        assert _g1(lib, 2, p, q) == E.add(E.add(p, p), E.add(q, q))
        assert _g1(lib, 3, p, q) == E.add(E.add(p, q), E.add(p, q))
        assert _g1(lib, 4, p, q) == E.add(p, p)
        assert _g1(lib, 5, p) == E.mul(p, 4) and _g1(lib, 6, p) is None
        # the branches a bucket can run into: the accumulator meets its own value, its negative, the point at infinity
        assert _g1(lib, 0, p, p) == E.add(p, p) and _g1(lib, 0, p, E.neg(p)) is None
        assert _g1(lib, 0, p, None) == p and _g1(lib, 0, None, q) == q and _g1(lib, 0, None, None) is None
        assert _g1(lib, 3, p, E.neg(p)) is None and _g1(lib, 3, p, None) == E.add(p, p) and _g1(lib, 3, None, q) == E.add(q, q)
        assert _g1(lib, 2, p, E.neg(p)) is None and _g1(lib, 2, p, p) == E.mul(p, 4) and _g1(lib, 2, None, q) == E.add(q, q)
    assert _g1(lib, 1, None) is None and _g1(lib, 5, None) is None


def test_scalar_multiplication_on_four_lanes(lib):
    """frw_quad.h: k P, k = k0 + lambda k1, by the level programmes run for four lanes one after the other -- against the oracle's
    double-and-add; incl. the scalars that leave the point as one of the three addends, that make the running point meet the
    addend (the complete one-lane formula takes over) or its negative, zero, and a point that arrives with ZZ != 1."""
    rng = random.Random(59)
    lam = E.Z_BLS ** 2 - 1
    degenerate = [0]

    def run(p, pre, k0, k1):
        out = (C.c_uint32 * 24)()
        k = (C.c_uint32 * 8)(*([(k0 >> (32 * i)) & 0xFFFFFFFF for i in range(4)] + [(k1 >> (32 * i)) & 0xFFFFFFFF for i in range(4)]))
        rc = lib.t_g1_quad_scale(_pt(p), pre, k, out)
        assert rc & 0xFF <= 32
        degenerate[0] += rc >> 8
        return E.from_limbs([out[2 * i] | out[2 * i + 1] << 32 for i in range(12)])

    p = E.mul(E.G1, rng.randrange(1, E.R))
    cases = [(0, 0), (1, 0), (0, 1), (1, 1), (2, 0), (3, 0), (0, 2), (3, 3), (2, 3), (5, 4), ((1 << 128) - 1, (1 << 128) - 1), (lam - 1, lam + 1)]
    cases += [(rng.randrange(1 << 128), rng.randrange(1 << 128)) for _ in range(6)]
    cases += [(rng.randrange(1 << 20), 0), (0, rng.randrange(1 << 20)), (1 << 127, 0), (0, 1 << 127)]
    for k0, k1 in cases:
        for pre in (0, 2):
            base = E.mul(p, 1 << pre)
            assert run(p, pre, k0, k1) == E.mul(base, (k0 + lam * k1) % E.R), (hex(k0), hex(k1), pre)
    for k0 in range(8):
        for k1 in range(8):
            assert run(p, 1, k0, k1) == E.mul(E.mul(p, 2), (k0 + lam * k1) % E.R)
    assert degenerate[0] == 0
    # the running point meets the addend (or its negative) when 2 v = +-a mod r for the prefix v and a in {1, lambda, 1 + lambda}:
    # v = +-a / 2 mod r, split like any scalar, then one more bit pair that selects a
    met = 0
    for a_sel, a in ((1, 1), (2, lam), (3, 1 + lam)):
        for sign in (1, -1):
            v = sign * a * pow(2, -1, E.R) % E.R
            v0, v1 = v % lam, v // lam
            if v0 >> 127 or v1 >> 127:
                continue
            k0, k1 = v0 << 1 | (a_sel & 1), v1 << 1 | (a_sel >> 1)
            before = degenerate[0]
            assert run(p, 1, k0, k1) == (E.mul(E.mul(p, 2), 2 * a % E.R) if sign == 1 else None), (a_sel, sign)
            assert degenerate[0] == before + 1
            # and the chain goes on from there: two more bit pairs
            if not (k0 >> 126 or k1 >> 126):
                assert run(p, 0, k0 << 2 | 1, k1 << 2 | 2) == E.mul(p, (((k0 << 2) | 1) + lam * ((k1 << 2) | 2)) % E.R)
            met += 1
    assert met >= 2
    assert run(None, 0, 5, 7) is None


def test_level_programmes_are_well_formed(lib):
    """frw_quad.h's level programmes as data: no two lanes of a level write the same slot (but the dump), a level never writes a
    constant or the addend, a temporary is written before it is read within its operation, and the bounds the header states hold
    when every slot carries the bound of what was last written to it (a difference must never subtract more than the K q it adds)."""
    buf = (C.c_uint8 * (8 * 4 * 12))()
    assert lib.t_quad_programmes(buf) == 8
    names = ["ZERO", "K4", "K8", "X", "AY", "BY", "ZZ", "ZZZ", "PX", "PEX", "PBX", "PY", "PNY", "PZZ", "PZZZ"] + ["T%d" % i for i in range(11)] + ["DUMP"]
    idx = {n: i for i, n in enumerate(names)}
    QX, QY = 62, 63
    steps = [[dict(ap=list(buf[(s * 4 + l) * 12:(s * 4 + l) * 12 + 4]), am=list(buf[(s * 4 + l) * 12 + 4:(s * 4 + l) * 12 + 7]),
                   bp=list(buf[(s * 4 + l) * 12 + 7:(s * 4 + l) * 12 + 9]), bm=buf[(s * 4 + l) * 12 + 9], dst=buf[(s * 4 + l) * 12 + 10],
                   kind=buf[(s * 4 + l) * 12 + 11]) for l in range(4)] for s in range(8)]
    state = {idx[n] for n in ("X", "AY", "BY", "ZZ", "ZZZ")}
    constant = {idx[n] for n in ("ZERO", "K4", "K8", "PX", "PEX", "PBX", "PY", "PNY", "PZZ", "PZZZ")}
    # bounds in units of q: what setup() and the programmes leave in the slots
    for first, last in ((0, 3), (3, 7), (7, 8)):                 # doubling, addition, copy
        bound = {idx["ZERO"]: 0, idx["K4"]: 4, idx["K8"]: 8, idx["X"]: 10, idx["AY"]: 2, idx["BY"]: 2, idx["ZZ"]: 2, idx["ZZZ"]: 2,
                 idx["PX"]: 10, idx["PEX"]: 2, idx["PBX"]: 2, idx["PY"]: 2, idx["PNY"]: 2, idx["PZZ"]: 2, idx["PZZZ"]: 2, QX: 10, QY: 2}
        written = set()
        for step in steps[first:last]:
            dsts = [l["dst"] for l in step if l["dst"] != idx["DUMP"]]
            assert len(dsts) == len(set(dsts)) and not (set(dsts) & constant)
            new_bounds = {}
            for l in step:
                reads = [x for x in l["ap"] + l["am"] + l["bp"] + [l["bm"]] if x != idx["ZERO"]]
                for x in reads:
                    assert x in bound, "slot %s is read before anything wrote it" % (names[x] if x < len(names) else x)
                plus_a, minus_a = sum(bound[x] for x in l["ap"]), sum(bound[x] for x in l["am"])
                plus_b, minus_b = sum(bound[x] for x in l["bp"]), bound[l["bm"]]
                k_a = sum(bound[x] for x in l["ap"] if x in (idx["K4"], idx["K8"]))
                k_b = sum(bound[x] for x in l["bp"] if x in (idx["K4"], idx["K8"]))
                assert minus_a <= k_a and minus_b <= k_b, "a difference may go negative"
                assert plus_a < 4096 and plus_b < 4096                          # operands of a product: anything below 2^12 q
                if l["dst"] != idx["DUMP"]:
                    new_bounds[l["dst"]] = plus_a if l["kind"] == 1 else 2       # a linear form keeps its bound, a product returns < 2 q
            bound.update(new_bounds)
            written |= set(new_bounds)
        assert state <= written | {idx["BY"]}                                   # (the copy leaves BY as it is: zero for the identity)
        assert bound[idx["X"]] <= 10 and bound[idx["AY"]] <= 2 and bound[idx["BY"]] <= 2 and bound[idx["ZZ"]] <= 2 and bound[idx["ZZZ"]] <= 2


def limbs2(v):
    return (C.c_uint32 * 28)(*(list(limbs(v[0])) + list(limbs(v[1]))))


def value2(arr):
    return value(arr[:14]), value(arr[14:28])


def test_fq2_operations_at_their_bounds(lib):
    rng = random.Random(53)
    inv = pow(RQ, -1, Q)
    out = (C.c_uint32 * 28)()
    for _ in range(200):
        a = (rng.randrange(300 * Q), rng.randrange(300 * Q))                 # the largest operands a G2 formula multiplies
        b = (rng.randrange(300 * Q), rng.randrange(300 * Q))
        lib.t_fq2(0, limbs2(a), limbs2(b), out)
        c0, c1 = value2(out)
        want = E.f2_mul((a[0] % Q, a[1] % Q), (b[0] % Q, b[1] % Q))
        assert (c0 % Q, c1 % Q) == (want[0] * inv % Q, want[1] * inv % Q) and c0 < 6 * Q and c1 < 10 * Q
        lib.t_fq2(1, limbs2(a), limbs2(a), out)
        c0, c1 = value2(out)
        want = E.f2_mul((a[0] % Q, a[1] % Q), (a[0] % Q, a[1] % Q))
        assert (c0 % Q, c1 % Q) == (want[0] * inv % Q, want[1] * inv % Q) and c0 < 2 * Q and c1 < 4 * Q
    for _ in range(10):
        x = (rng.randrange(Q), rng.randrange(Q))
        lib.t_fq2(2, limbs2((x[0] * RQ % Q, x[1] * RQ % Q)), limbs2((0, 0)), out)
        c0, c1 = value2(out)
        xi = E.f2_inv(x)
        assert (c0 % Q, c1 % Q) == (xi[0] * RQ % Q, xi[1] * RQ % Q)
    for a, z in (((0, 0), 1), ((5 * Q, 17 * Q), 1), ((Q, 1), 0), ((1, 3 * Q), 0), ((2047 * Q, 0), 1)):
        lib.t_fq2(3, limbs2(a), limbs2((0, 0)), out)
        assert out[0] == z


def _pt2(p):
    return (C.c_uint32 * 48)(*[(l >> (32 * h)) & 0xFFFFFFFF for l in E.g2_to_limbs(p) for h in range(2)])


def _g2(lib, op, p, q=None):
    out = (C.c_uint32 * 48)()
    lib.t_g2(op, _pt2(p), _pt2(q), out)
    return E.g2_from_limbs([out[2 * i] | out[2 * i + 1] << 32 for i in range(24)])


def test_g2_point_formulas_every_branch_and_a_long_chain(lib):
    rng = random.Random(59)
    assert E.g2_on_curve(E.G2) and E.g2_mul(E.G2, E.R) is None
    for _ in range(3):
        p, q = E.g2_mul(E.G2, rng.randrange(1, E.R)), E.g2_mul(E.G2, rng.randrange(1, E.R))
        assert _g2(lib, 0, p, q) == E.g2_add(p, q) and _g2(lib, 1, p) == E.g2_add(p, p)
        assert _g2(lib, 2, p, q) == E.g2_add(E.g2_add(p, p), E.g2_add(q, q))
        assert _g2(lib, 3, p, q) == E.g2_add(E.g2_add(p, q), E.g2_add(p, q))
        assert _g2(lib, 4, p, q) == E.g2_add(p, p)
        assert _g2(lib, 5, p) == E.g2_mul(p, 4) and _g2(lib, 6, p) is None
        assert _g2(lib, 0, p, p) == E.g2_add(p, p) and _g2(lib, 0, p, E.g2_neg(p)) is None
        assert _g2(lib, 0, p, None) == p and _g2(lib, 0, None, q) == q and _g2(lib, 0, None, None) is None
        assert _g2(lib, 2, p, E.g2_neg(p)) is None and _g2(lib, 2, p, p) == E.g2_mul(p, 4)
        # forty formulas in a row without any reduction in between: the stated bounds (X < 90 q, Y < 26 q) are invariants
        want = p
        for k in range(40):
            want = E.g2_add(want, want) if k % 7 == 6 else E.g2_add(want, q)
        assert _g2(lib, 7, p, q) == want
        p1, q1 = E.mul(E.G1, rng.randrange(1, E.R)), E.mul(E.G1, rng.randrange(1, E.R))
        out = (C.c_uint32 * 24)()
        lib.t_g1_chain(_pt(p1), _pt(q1), out)
        want = p1
        for k in range(40):
            want = E.add(want, want) if k % 7 == 6 else E.add(want, q1)
        assert E.from_limbs([out[2 * i] | out[2 * i + 1] << 32 for i in range(12)]) == want
    src = open(os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_fq29.h")).read()
    import re
    for name, v in zip(("G2_GEN_X0_29", "G2_GEN_X1_29", "G2_GEN_Y0_29", "G2_GEN_Y1_29"), (E.G2[0][0], E.G2[0][1], E.G2[1][0], E.G2[1][1])):
        body = re.search(name + r" = \{\{([^}]*)\}\}", src).group(1)
        assert value([int(x.strip().rstrip("u"), 16) for x in body.replace("\n", " ").split(",")]) == v * RQ % Q
