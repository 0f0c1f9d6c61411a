"""The nine-limb field arithmetic of the QAP kernels (falcon-r1cs_amd/csrc/frw_fr29.h), compiled for the host through a
test-only shim (tests/cpp/hip_host) and checked against Python integers: products, lazy additions and subtractions with
their K p offsets, the conditional reductions, packing, and the wide Montgomery reduction -- at the bounds the kernels rely on."""
import os
import random
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
RP = 1 << 261


@pytest.fixture(scope="module")
def harness():
    out = os.path.join(HERE, "cpp", "build")
    os.makedirs(out, exist_ok=True)
    exe = os.path.join(out, "test_fr29")
    src = os.path.join(HERE, "cpp", "test_fr29.cpp")
    hdrs = [os.path.join(ROOT, "falcon-r1cs_amd", "csrc", h) for h in ("frw_fr29.h", "frw_fr.h")]
    if not os.path.exists(exe) or os.path.getmtime(exe) < max(os.path.getmtime(f) for f in [src] + hdrs):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-Wno-unknown-pragmas", "-I", os.path.join(HERE, "cpp", "hip_host"),
                               "-I", os.path.join(ROOT, "falcon-r1cs_amd", "csrc"), "-o", exe, src])

    def run(ops):
        text = "\n".join("%s %s" % (op, " ".join("%x" % v for v in args)) for op, *args in ops) + "\n"
        res = subprocess.run([exe], input=text, capture_output=True, text=True, timeout=120, check=True).stdout.split("\n")
        return [(int(l.split()[0], 16), l.split()[1] == "1") for l in res if l]
    return run


def test_products_and_reductions(harness):
    rng = random.Random(29)
    edge = [0, 1, P - 1, P, 2 * P - 1, (1 << 255) - 1]
    ops, want = [], []
    for _ in range(400):
        a = rng.choice(edge + [rng.randrange(60 * P)] * 6)          # lazily reduced left operand, up to 60 p
        b = rng.choice(edge[:5] + [rng.randrange(2 * P)] * 6)       # normalised right operand < 2 p
        ops.append(("mul", a, b)); want.append(("mul", a * b))
    got = harness(ops)
    inv = pow(RP, -1, P)
    for (op, a, b), (v, norm), (_, prod) in zip(ops, got, want):
        assert norm and v % P == prod * inv % P
        assert v < (prod >> 261) + P + 1                             # < a b / R' + p, so < 2 p when a b < R' p
    ops = []
    for _ in range(300):
        a, b = rng.randrange(2 * P), rng.randrange(2 * P)
        ops += [("add", a, b), ("sub2", a, b), ("sub4", rng.randrange(8 * P), rng.randrange(4 * P)),
                ("sub8", rng.randrange(16 * P), rng.randrange(8 * P)), ("red4", rng.randrange(4 * P)), ("canon", rng.randrange(2 * P)),
                ("packunpack", rng.randrange(1 << 256))]
    ops += [("red4", v) for v in (0, 2 * P - 1, 2 * P, 4 * P - 1)] + [("canon", v) for v in (0, P - 1, P, 2 * P - 1)]
    ops += [("sub2", 0, 2 * P - 1), ("sub4", 0, 4 * P - 1), ("sub8", 0, 8 * P - 1)]
    for (op, *args), (v, norm) in zip(ops, harness(ops)):
        assert norm, op
        if op == "add": assert v == args[0] + args[1]
        if op == "sub2": assert v == args[0] - args[1] + 2 * P
        if op == "sub4": assert v == args[0] - args[1] + 4 * P
        if op == "sub8": assert v == args[0] - args[1] + 8 * P
        if op == "red4": assert v == (args[0] - 2 * P if args[0] >= 2 * P else args[0])
        if op == "canon": assert v == (args[0] - P if args[0] >= P else args[0])
        if op == "packunpack": assert v == args[0]


def test_lazy_decimation_in_frequency_bounds(harness):
    """The butterflies of the factor passes without conditional subtractions (frw_qap.hip dif_mul_lazy): differences with
    16 p and 32 p added, the one conditional subtraction of 8 p between the rounds, and the worst case of the whole chain --
    a sum just under 64 p entering a product with a factor just under p -- must come back below 2 p (it is 2^261 = 70.4 p
    that allows it), exactly as an integer computation says."""
    rng = random.Random(37)
    ops = []
    for _ in range(200):
        ops += [("sub16", rng.randrange(16 * P), rng.randrange(16 * P)), ("sub32", rng.randrange(32 * P), rng.randrange(32 * P)),
                ("csub8", rng.randrange(16 * P))]
    ops += [("sub16", 0, 16 * P - 1), ("sub16", 16 * P - 1, 0), ("sub32", 0, 32 * P - 1), ("sub32", 32 * P - 1, 0),
            ("csub8", 0), ("csub8", 8 * P - 1), ("csub8", 8 * P), ("csub8", 16 * P - 1)]
    worst = [(64 * P - 1, P - 1), (64 * P - 1, 1), (63 * P + 12345, P - 2)] + [(rng.randrange(32 * P, 64 * P), rng.randrange(P)) for _ in range(200)]
    ops += [("mul", a, b) for a, b in worst]
    inv = pow(RP, -1, P)
    for (op, *args), (v, norm) in zip(ops, harness(ops)):
        assert norm, op
        if op == "sub16": assert v == args[0] - args[1] + 16 * P
        if op == "sub32": assert v == args[0] - args[1] + 32 * P and v < 64 * P
        if op == "csub8": assert v == (args[0] - 8 * P if args[0] >= 8 * P else args[0]) and v < 8 * P
        if op == "mul": assert v % P == args[0] * args[1] * inv % P and v < 2 * P and v < (1 << 256)


def test_wide_montgomery_reduction(harness):
    """f29_redc_wide: x / 2^261 mod p for x up to 2^300 (what 64 lanes of a long row add up to), result < 2 p."""
    rng = random.Random(31)
    xs = [0, 1, RP, (1 << 300) - 1] + [rng.randrange(1 << 300) for _ in range(200)]
    inv = pow(RP, -1, P)
    for x, (v, norm) in zip(xs, harness([("redc", x) for x in xs])):
        assert norm and v < 2 * P and v % P == x * inv % P
