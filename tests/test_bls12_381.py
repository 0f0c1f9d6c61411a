"""The G1 / Groth16 oracle (oracle/bls12_381.py, oracle/bls12_381.c) pinned to what can be pinned without arkworks:
the curve's published parametrisation, group laws, C == Python integers, the Groth16 verification equation in the
exponent, and the MSM-free value of the h_query multi-scalar multiplication (examples/pok_sig.rs:30-47)."""
import random

import numpy as np
import pytest

import frw_testlib as T
from oracle import bls12_381 as E
from oracle import qap as Q


def test_published_parameters_and_generator():
    E.check_parameters()
    # cofactor of G1: (z - 1)^2 / 3; #E(Fq) = h r
    h = (E.Z_BLS - 1) ** 2 // 3
    assert h == 0x396C8C005555E1568C00AAAB0000AAAB
    rng = random.Random(1)
    for _ in range(3):
        k = rng.randrange(E.R)
        p = E.mul(E.G1, k)
        assert E.on_curve(p) and E.add(E.mul(p, E.R - 1), p) is None      # (mul reduces its scalar mod r)
        assert E.add(p, E.neg(p)) is None and E.add(p, p) == E.mul(E.G1, 2 * k)


def test_c_oracle_equals_python_integers(oracle):
    rng = random.Random(2)
    g = oracle.g1_generator()
    assert g.tolist() == E.to_limbs(E.G1) and oracle.g1_on_curve(g)
    assert E.from_limbs(g) == E.G1
    for k in (0, 1, 2, E.R - 1, E.R, rng.randrange(E.R), rng.randrange(1 << 255)):
        got = oracle.g1_scalar_mul(g, k % (1 << 256))
        assert got.tolist() == E.to_limbs(E.mul(E.G1, k)), k
    a, b = E.mul(E.G1, 1234567), E.mul(E.G1, 7654321)
    for x, y in ((a, b), (a, a), (a, E.neg(a)), (a, None), (None, b), (None, None)):
        assert oracle.g1_add(E.to_limbs(x), E.to_limbs(y)).tolist() == E.to_limbs(E.add(x, y))
    ks = [0, 1, E.R - 1] + [rng.randrange(E.R) for _ in range(13)]
    fb = oracle.g1_fixed_base(T.ints_to_limbs(ks), threads=3)
    assert [row.tolist() for row in fb] == [E.to_limbs(E.mul(E.G1, k)) for k in ks]


@pytest.mark.parametrize("window_bits", [3, 8, 13, 16])
def test_c_bucket_method_equals_the_sum_of_scalar_multiplications(oracle, window_bits):
    """Edge cases the bucket method must survive: zero and one scalars, r - 1 (all digits negative-carry), the same base
    twice (an accumulator meets its own value: doubling), a base and its negative (cancellation), the point at infinity."""
    rng = random.Random(100 + window_bits)
    base_k = [rng.randrange(1, E.R) for _ in range(20)]
    bases = [E.mul(E.G1, k) for k in base_k]
    bases += [bases[0], bases[1], E.neg(bases[2]), None]
    scalars = [0, 1, E.R - 1, 2, (1 << 255) - 19] + [rng.randrange(E.R) for _ in range(15)]
    scalars += [scalars[0] + 5, scalars[1], scalars[2], 77]
    want = E.msm_naive(bases, scalars)
    got = oracle.g1_msm(np.array([E.to_limbs(b) for b in bases], dtype=np.uint64), T.ints_to_limbs(scalars), window_bits, threads=4)
    assert got.tolist() == E.to_limbs(want)
    # same digit everywhere: every point lands in one bucket, window after window
    same = [0x0101010101010101010101010101010101010101010101010101010101010101 % E.R] * len(bases)
    got = oracle.g1_msm(np.array([E.to_limbs(b) for b in bases], dtype=np.uint64), T.ints_to_limbs(same), window_bits, threads=2)
    assert got.tolist() == E.to_limbs(E.msm_naive(bases, same))


def _toy_system():
    """x^3 + x + 5 = out with x = 3, out = 35 public: z = (1, out | x, x^2, x^3);  rows: x x = x2;  x2 x = x3;
    (x3 + x + 5) 1 = out -- plus a second public input tied to it (2 out = dbl) so that two instance rows are appended."""
    # columns: 0 one, 1 out, 2 dbl | 3 x, 4 x2, 5 x3
    a = [[(1, 3)], [(1, 4)], [(1, 5), (1, 3), (5, 0)], [(2, 1)]]
    b = [[(1, 3)], [(1, 3)], [(1, 0)], [(1, 0)]]
    c = [[(1, 4)], [(1, 5)], [(1, 1)], [(1, 2)]]
    z = [1, 35, 70, 3, 9, 27]
    return (a, b, c), 3, 3, z


def test_groth16_verification_equation_holds_in_the_exponent():
    mats, ni, nw, z = _toy_system()
    d = Q.Domain(len(mats[0]) + ni)
    assert all(x * y % E.R == w for x, y, w in zip(*Q.matvec(mats, z)))
    rng = random.Random(7)
    toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
    pk = E.setup_exponents(mats, ni, nw, d, toxic)
    assert len(pk["h"]) == d.size - 1 and len(pk["l"]) == nw and len(pk["gamma_abc"]) == ni
    h = Q.witness_map(mats, ni, z)
    assert h[-1] == 0
    for r, s in ((0, 0), (rng.randrange(E.R), rng.randrange(E.R))):
        proof = E.prove_exponents(pk, z, h, r, s)
        assert E.verify_exponents(pk, z[1:ni], proof)
        assert not E.verify_exponents(pk, [36, 70], proof)                       # another statement
    zbad = list(z)
    zbad[4] = 10                                                                 # x2 != x x: h is not a quotient any more
    hbad = Q.witness_map(mats, ni, zbad)
    assert not E.verify_exponents(pk, z[1:ni], E.prove_exponents(pk, zbad, hbad, 5, 6))


def test_prover_from_products_equals_the_prover_from_the_matrices():
    """prove_exponents_from_products (the form that scales to aggregate statements: only A z, B z, C z and the matrix entries
    in instance columns) gives the same exponents as prove_exponents on the toy system and on a random one with public
    inputs inside long rows."""
    from test_qap import small_system
    rng = random.Random(11)
    toy = _toy_system()
    random_mats, random_z = small_system(rng, 40, 6, 20)
    for mats, ni, z in [(toy[0], 3, toy[3]), (random_mats, 6, random_z)]:
        nw = len(z) - ni
        d = Q.Domain(len(mats[0]) + ni)
        toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
        pk = E.setup_exponents(mats, ni, nw, d, toxic)
        h = Q.witness_map(mats, ni, z)
        az, bz, cz = Q.matvec(mats, z)
        terms = [(m, row, col, coeff) for m in range(3) for row, r_ in enumerate(mats[m]) for coeff, col in r_ if col < ni]
        for r, s in ((0, 0), (rng.randrange(E.R), rng.randrange(E.R))):
            a, b, c, h_acc = E.prove_exponents(pk, z, h, r, s)
            a2, b2, c2, h2, gamma_abc = E.prove_exponents_from_products(toxic, d, az, bz, cz, z[:ni], terms, h, r, s)
            assert (a, b, c, h_acc) == (a2, b2, c2, h2)
            assert gamma_abc == pk["gamma_abc"]
            assert E.verify_exponents({"toxic": toxic, "gamma_abc": gamma_abc}, z[1:ni], (a2, b2, c2))


def test_h_query_msm_has_an_msm_free_value(oracle):
    """sum h_i h_query[i] = (h(t) zt / delta) G1: what the GPU's multi-scalar multiplication is compared with at full size,
    checked here with real points on the toy system -- bases from the C oracle's fixed-base routine, sum by Python integers,
    by the C bucket method and by ONE scalar multiplication."""
    mats, ni, nw, z = _toy_system()
    d = Q.Domain(len(mats[0]) + ni)
    rng = random.Random(8)
    toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
    pk = E.setup_exponents(mats, ni, nw, d, toxic)
    h = Q.witness_map(mats, ni, z)
    bases = oracle.g1_fixed_base(T.ints_to_limbs(pk["h"]))
    assert [E.from_limbs(b) for b in bases] == [E.mul(E.G1, k) for k in pk["h"]]
    h_at_t = sum(c * pow(toxic["t"], i, E.R) for i, c in enumerate(h)) % E.R
    expected = E.mul(E.G1, h_at_t * pk["zt"] % E.R * pow(toxic["delta"], -1, E.R) % E.R)
    assert E.msm_naive([E.from_limbs(b) for b in bases], h[:d.size - 1]) == expected
    assert oracle.g1_msm(bases, T.ints_to_limbs(h[:d.size - 1]), 5).tolist() == E.to_limbs(expected)
    assert E.prove_exponents(pk, z, h, 0, 0)[3] == h_at_t * pk["zt"] % E.R * pow(toxic["delta"], -1, E.R) % E.R


def test_pairing_is_bilinear_and_verifies_a_real_proof():
    """The plain restatement of the pairing (Fq12 as Fq[w] / (w^12 - 2 w^6 + 2), affine ate Miller loop, one big final power):
    bilinear, of order r, non-degenerate -- and with it ark-groth16's verify_proof on ACTUAL points: the proof of the toy system
    made by the restated prover (exponents -> points) verifies, another statement and a tampered C do not."""
    rng = random.Random(12)
    e1 = E.pairing(E.G1, E.G2)
    assert e1 != E.F12_ONE and E.f12_pow(e1, E.R) == E.F12_ONE
    a, b = rng.randrange(1, E.R), rng.randrange(1, E.R)
    assert E.pairing(E.mul(E.G1, a), E.g2_mul(E.G2, b)) == E.f12_pow(e1, a * b % E.R)
    assert E.pairing(E.add(E.mul(E.G1, a), E.G1), E.G2) == E.f12_mul(E.f12_pow(e1, a), e1)         # additive in the first argument
    assert E.pairing(None, E.G2) == E.F12_ONE and E.pairing(E.G1, None) == E.F12_ONE
    x = tuple(rng.randrange(E.Q) for _ in range(12))
    assert E.f12_mul(x, E.f12_inv(x)) == E.F12_ONE
    mats, ni, nw, z = _toy_system()
    d = Q.Domain(len(mats[0]) + ni)
    toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
    pk = E.setup_exponents(mats, ni, nw, d, toxic)
    h = Q.witness_map(mats, ni, z)
    pa, pb, pc, _ = E.prove_exponents(pk, z, h, rng.randrange(E.R), rng.randrange(E.R))
    vk = {"alpha_g1": E.mul(E.G1, toxic["alpha"]), "beta_g2": E.g2_mul(E.G2, toxic["beta"]), "gamma_g2": E.g2_mul(E.G2, toxic["gamma"]),
          "delta_g2": E.g2_mul(E.G2, toxic["delta"]), "gamma_abc_g1": [E.mul(E.G1, g) for g in pk["gamma_abc"]]}
    proof = (E.mul(E.G1, pa), E.g2_mul(E.G2, pb), E.mul(E.G1, pc))
    assert E.verify_proof(vk, z[1:ni], proof)
    assert not E.verify_proof(vk, [36, 70], proof)
    assert not E.verify_proof(vk, z[1:ni], (proof[0], proof[1], E.add(proof[2], E.G1)))
