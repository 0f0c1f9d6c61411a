"""The product's Groth16 verifier (falcon-r1cs_amd/csrc/frw_verify.cpp, frw_pairing.h: host code, runs without a GPU)
against the oracle's pairing (oracle/bls12_381.py: Fq12 as Fq[w], affine Miller loop, one big power) -- two constructions
that share nothing but the curve.  What examples/pok_sig.rs:47 calls: Groth16::verify -> ark-groth16 verifier.rs."""
import random

import numpy as np
import pytest

import falcon_r1cs_amd as frw
from oracle import bls12_381 as E


def u64(limbs):
    return np.array(limbs, dtype=np.uint64)


def gt_from_product(out):
    inv = pow(E.FQ_R, -1, E.Q)
    return tuple(sum(int(l) << (64 * i) for i, l in enumerate(row)) * inv % E.Q for row in out)


def fr_limbs(values, montgomery):
    r2 = 1 << 256
    return np.frombuffer(b"".join(((v * r2 % E.R) if montgomery else v).to_bytes(32, "little") for v in values), dtype=np.uint64).reshape(-1, 4)


def test_hard_part_decomposition_is_an_identity_of_integers():
    """frw_pairing.h's final exponentiation: l0 + l1 q + l2 q^2 + l3 q^3 = 3 (q^4 - q^2 + 1) / r."""
    z, q, r = E.Z_BLS, E.Q, E.R
    assert z == -0xD201000000010000
    l3 = (z - 1) ** 2
    l2 = l3 * z
    l1 = l2 * z - l3
    l0 = l1 * z + 3
    assert (q ** 4 - q ** 2 + 1) % r == 0
    assert l0 + l1 * q + l2 * q ** 2 + l3 * q ** 3 == 3 * (q ** 4 - q ** 2 + 1) // r
    assert (q - 1) % 6 == 0 and r % 3 != 0


def test_pairing_equals_the_oracles_value():
    """product = (reduced ate pairing)^3 with the loop over z < 0; the oracle runs over |z| and takes the plain power:
    product == (oracle^-1)^3, coefficient for coefficient in the oracle's basis; and bilinearity through the product alone."""
    rng = random.Random(11)
    a, b = rng.randrange(1, E.R), rng.randrange(1, E.R)
    p, q = E.mul(E.G1, a), E.g2_mul(E.G2, b)
    got = gt_from_product(frw.diag_pairing(u64(E.to_limbs(p)), u64(E.g2_to_limbs(q))))
    assert got == E.f12_pow(E.f12_inv(E.pairing(p, q)), 3)
    base = gt_from_product(frw.diag_pairing(u64(E.to_limbs(E.G1)), u64(E.g2_to_limbs(E.G2))))
    assert base != E.F12_ONE and E.f12_pow(base, E.R) == E.F12_ONE
    assert got == E.f12_pow(base, a * b % E.R)
    # a point at infinity on either side: one
    zero1, zero2 = np.zeros(12, dtype=np.uint64), np.zeros(24, dtype=np.uint64)
    assert gt_from_product(frw.diag_pairing(zero1, u64(E.g2_to_limbs(q)))) == E.F12_ONE
    assert gt_from_product(frw.diag_pairing(u64(E.to_limbs(p)), zero2)) == E.F12_ONE
    # a point off the curve is refused
    bad = u64(E.to_limbs(p)).copy()
    bad[0] ^= 1
    with pytest.raises(frw.FrwError):
        frw.diag_pairing(bad, u64(E.g2_to_limbs(q)))


def make_statement(rng, num_public, input_bits):
    """A verifying key, public inputs and a proof that satisfy Groth16's equation by construction (in the exponent):
    a b = alpha beta + gamma sum x_i g_i + c delta."""
    alpha, beta, gamma, delta = (rng.randrange(1, E.R) for _ in range(4))
    g = [rng.randrange(1, E.R) for _ in range(num_public + 1)]
    x = [1] + [rng.randrange(1 << input_bits) % E.R for _ in range(num_public)]
    a, b = rng.randrange(1, E.R), rng.randrange(1, E.R)
    acc = sum(xi * gi for xi, gi in zip(x, g)) % E.R
    c = (a * b - alpha * beta - gamma * acc) * pow(delta, -1, E.R) % E.R
    vk = {"alpha_g1": E.mul(E.G1, alpha), "beta_g2": E.g2_mul(E.G2, beta), "gamma_g2": E.g2_mul(E.G2, gamma),
          "delta_g2": E.g2_mul(E.G2, delta), "gamma_abc_g1": [E.mul(E.G1, gi) for gi in g]}
    proof = (E.mul(E.G1, a), E.g2_mul(E.G2, b), E.mul(E.G1, c))
    return vk, x, proof


def vk_limbs(vk):
    return {"alpha_g1": u64(E.to_limbs(vk["alpha_g1"])), "beta_g2": u64(E.g2_to_limbs(vk["beta_g2"])),
            "gamma_g2": u64(E.g2_to_limbs(vk["gamma_g2"])), "delta_g2": u64(E.g2_to_limbs(vk["delta_g2"])),
            "gamma_abc_g1": u64([E.to_limbs(p) for p in vk["gamma_abc_g1"]])}


def proof_limbs(proof):
    return u64(E.to_limbs(proof[0]) + E.g2_to_limbs(proof[1]) + E.to_limbs(proof[2]))


@pytest.mark.parametrize("input_bits,montgomery", [(14, True), (255, False)])
def test_verifier_accepts_and_rejects_like_the_oracle(input_bits, montgomery):
    rng = random.Random(20 + input_bits)
    vk, x, proof = make_statement(rng, 5, input_bits)
    assert E.verify_proof(vk, x[1:], proof)
    ver = frw.Groth16Verifier(vk_limbs(vk))
    enc = frw.ENC_MONTGOMERY if montgomery else frw.ENC_CANONICAL
    cases, want = [], []
    cases.append((x, proof)); want.append(1)
    x2 = list(x); x2[3] = (x2[3] + 1) % E.R
    cases.append((x2, proof)); want.append(0)                                          # another statement
    cases.append((x, (E.mul(proof[0], 2), proof[1], proof[2]))); want.append(0)         # A tampered with
    cases.append((x, (proof[0], E.g2_mul(proof[1], 3), proof[2]))); want.append(0)      # B
    cases.append((x, (proof[0], proof[1], E.add(proof[2], E.G1)))); want.append(0)      # C
    cases.append((x, (None, proof[1], proof[2]))); want.append(0)                       # A at infinity: a point, not a proof
    x3 = list(x); x3[0] = 2
    cases.append((x3, proof)); want.append(-1)                                          # the constant one is not one
    for (xi, pi), w in zip(cases, want):
        if w >= 0:
            assert E.verify_proof(vk, xi[1:], pi) == bool(w)
    inst = np.stack([fr_limbs(xi, montgomery) for xi, _ in cases])
    proofs = np.stack([proof_limbs(pi) for _, pi in cases])
    assert ver.verify(inst, proofs, enc).tolist() == want
    # malformed encodings: a value >= r (canonical only: every 256-bit pattern below 2^256 is some Montgomery residue), a point off its curve
    if not montgomery:
        big = fr_limbs(x, False).copy()
        big[2] = np.frombuffer(E.R.to_bytes(32, "little"), dtype=np.uint64)
        assert ver.verify(big[None], proofs[:1], enc).tolist() == [-1]
    off = proofs[0].copy()
    off[12] ^= 1
    assert ver.verify(inst[:1], off[None], enc).tolist() == [-1]
    ver.close()


def test_points_outside_the_subgroup_are_malformed_unless_vouched_for():
    """ark checks the subgroup when it deserialises a proof; the C ABI takes raw limbs, so the verifier does."""
    rng = random.Random(31)
    vk, x, proof = make_statement(rng, 2, 14)
    ver = frw.Groth16Verifier(vk_limbs(vk))
    # a point of E(Fq) outside G1: x = 1, 2, ... until x^3 + 4 is a square and r P != O
    xx = 0
    while True:
        xx += 1
        y2 = (xx ** 3 + 4) % E.Q
        y = pow(y2, (E.Q + 1) // 4, E.Q)
        # (E.mul reduces its scalar mod r: r P is (r - 1) P + P)
        if y * y % E.Q == y2 and E.add(E.mul((xx, y), E.R - 1), (xx, y)) is not None:
            break
    stray = (xx, y)
    assert E.on_curve(stray)
    inst = fr_limbs(x, True)[None]
    bad = proof_limbs((stray, proof[1], proof[2]))[None]
    assert ver.verify(inst, bad).tolist() == [-1]
    assert ver.verify(inst, bad, flags=frw.VERIFY_POINTS_ARE_CHECKED).tolist() == [0]
    assert ver.verify(inst, proof_limbs(proof)[None]).tolist() == [1]
    # ark's deserialiser rejects representations that are not below the modulus; the raw limbs are held to the same.
    # (1) a coordinate given as x + q (the same residue, another encoding): malformed, not "the same point"
    def plus_q(limbs6):
        v = int.from_bytes(np.asarray(limbs6, dtype=np.uint64).tobytes(), "little") + E.Q
        assert v < 1 << 384
        return np.frombuffer(v.to_bytes(48, "little"), dtype=np.uint64)
    good = proof_limbs(proof)
    for first in (0, 12, 18, 36, 42):                               # A.x, B.x.c0, B.x.c1, C.x, C.y
        alias = good.copy()
        alias[first:first + 6] = plus_q(good[first:first + 6])
        assert ver.verify(inst, alias[None]).tolist() == [-1], first
        assert ver.verify(inst, alias[None], flags=frw.VERIFY_POINTS_ARE_CHECKED).tolist() == [-1], first
    # (2) an instance value whose limbs are >= r, in Montgomery form too (the conversion would reduce it silently)
    big = fr_limbs(x, True).copy()
    v = int.from_bytes(big[1].tobytes(), "little") + E.R
    assert v < 1 << 256
    big[1] = np.frombuffer(v.to_bytes(32, "little"), dtype=np.uint64)
    assert ver.verify(big[None], good[None]).tolist() == [-1]
    # (3) a verifying key whose gamma_abc holds a point outside the subgroup, or a non-canonical coordinate, does not load
    k = vk_limbs(vk)
    k["gamma_abc_g1"] = k["gamma_abc_g1"].copy()
    k["gamma_abc_g1"][1] = u64(E.to_limbs(stray))
    with pytest.raises(frw.FrwError):
        frw.Groth16Verifier(k)
    k = vk_limbs(vk)
    k["gamma_abc_g1"] = k["gamma_abc_g1"].copy()
    k["gamma_abc_g1"][2][:6] = plus_q(k["gamma_abc_g1"][2][:6])
    with pytest.raises(frw.FrwError):
        frw.Groth16Verifier(k)
    k = vk_limbs(vk)
    k["alpha_g1"] = k["alpha_g1"].copy()
    k["alpha_g1"][6:] = plus_q(k["alpha_g1"][6:])
    with pytest.raises(frw.FrwError):
        frw.Groth16Verifier(k)
    # a verifying key with a point off its curve does not load
    k = vk_limbs(vk)
    k["gamma_g2"] = k["gamma_g2"].copy()
    k["gamma_g2"][0] ^= 1
    with pytest.raises(frw.FrwError):
        frw.Groth16Verifier(k)
    ver.close()
