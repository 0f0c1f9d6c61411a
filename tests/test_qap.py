"""The R1CS -> QAP witness map ("a GPU Groth16 prover (MSM/FFT)": the FFT half), CPU side.

The oracle (oracle/qap.py, oracle/qap_oracle.c) restates ark-groth16 0.3.0's R1CStoQAP::witness_map over ark-poly
0.3.0's radix-2 domain -- crates that are not in /root/reference, so parity with them is unpinned; what pins the
oracle here: (1) the field constants against ark-bls12-381's published limbs, (2) C == Python step for step on small
systems, (3) the defining identity A(tau) B(tau) - C(tau) = h(tau) (tau^n - 1), evaluated without any FFT, on small
systems and on the real Falcon-512 circuit with the oracle's witness."""
import random

import numpy as np
import pytest

import frw_testlib as T
from oracle import qap

P = qap.P


def small_system(rng, rows, num_inputs, nvars0):
    z = [1] + [rng.randrange(P) for _ in range(nvars0 - 1)]
    A, B, Cm = [], [], []
    for _ in range(rows):
        ra = [(rng.randrange(P), rng.randrange(len(z))) for _ in range(rng.randrange(1, 5))]
        rb = [(rng.randrange(P), rng.randrange(len(z))) for _ in range(rng.randrange(1, 4))]
        z.append(qap.evaluate_constraint(ra, z) * qap.evaluate_constraint(rb, z) % P)
        A.append(ra); B.append(rb); Cm.append([(1, len(z) - 1)])
    return (A, B, Cm), z


def test_field_constants_match_the_published_limbs():
    qap.check_constants()


@pytest.mark.parametrize("rows,num_inputs", [(1, 1), (5, 3), (29, 3), (100, 28), (250, 6)])
def test_python_witness_map_satisfies_the_identity(rows, num_inputs):
    rng = random.Random(rows)
    mats, z = small_system(rng, rows, num_inputs, 12 + num_inputs)
    az, bz, cz = qap.matvec(mats, z)
    h = qap.witness_map(mats, num_inputs, z)
    assert len(h) == qap.Domain(rows + num_inputs).size
    for _ in range(3):
        lhs, rhs = qap.check_identity(az, bz, cz, num_inputs, z, h, rng.randrange(P))
        assert lhs == rhs
    assert h[-1] == 0                                            # deg h <= n - 2
    z[-1] = (z[-1] + 1) % P                                      # break one constraint
    az, bz, cz = qap.matvec(mats, z)
    h = qap.witness_map(mats, num_inputs, z)
    lhs, rhs = qap.check_identity(az, bz, cz, num_inputs, z, h, 0x1234567)
    assert lhs != rhs


@pytest.mark.parametrize("rows,num_inputs", [(1, 1), (7, 2), (100, 28), (1000, 25)])
def test_c_oracle_equals_python_oracle(oracle, rows, num_inputs):
    rng = random.Random(1000 + rows)
    mats, z = small_system(rng, rows, num_inputs, 12 + num_inputs)
    want_abc = qap.matvec(mats, z)
    zl = T.ints_to_limbs(z)
    got_abc = []
    for m in mats:
        ptr = np.cumsum([0] + [len(r) for r in m]).astype(np.uint64)
        col = np.array([c for r in m for _, c in r], dtype=np.uint32)
        val = T.ints_to_limbs([v for r in m for v, _ in r])
        got_abc.append(oracle.qap_matvec(ptr, col, val, zl))
    for g, w in zip(got_abc, want_abc):
        assert T.limbs_to_ints(g) == w
    h = oracle.qap_witness_map(*got_abc, num_inputs, zl)
    assert T.limbs_to_ints(h) == qap.witness_map(mats, num_inputs, z)


def falcon_products(oracle, tmp_path, logn, seed):
    from test_r1cs_export import export, read_r1cs
    rng = random.Random(seed)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    wit, inst, st = oracle.witness_ntt_verify(logn, sig, pk, hm, 0)
    assert st[0] == 0
    z = np.concatenate([inst[0], wit[0]])
    path = tmp_path / "c.r1cs"
    export(0, logn, path)
    ni, nw, nc, mats = read_r1cs(path)
    assert ni + nw == z.shape[0]
    return ni, nc, z, [oracle.qap_matvec(*m, z) for m in mats]


def test_falcon512_witness_map_satisfies_the_identity(oracle, tmp_path):
    ni, nc, z, (az, bz, cz) = falcon_products(oracle, tmp_path, 9, 5)
    h = oracle.qap_witness_map(az, bz, cz, ni, z)
    assert h.shape[0] == 1 << 17 and not h[-1].any()
    lhs, rhs = qap.check_identity(*(T.limbs_to_ints(a) for a in (az, bz, cz)), ni, T.limbs_to_ints(z[:ni]),
                                  T.limbs_to_ints(h), random.Random(3).randrange(P))
    assert lhs == rhs
