"""The aggregate statement in the oracle's own terms (BASELINE configs[4]): FalconNTTVerificationCircuit run once
per statement on ONE constraint system of the arkworks front-end simulation (oracle/ark_sim.py).  What the product relies on -- and
never materialises -- is that this system is the block arrangement of the per-signature systems: rows, witness variables and public
inputs of the statements end to end, one shared constant column.  Here that is checked on the oracle's independently built system,
entry by entry; tests/test_gpu_aggregate.py then holds the device to the same system's products, h and proof."""
import random

import pytest

import frw_testlib as T
from oracle import falcon_gadgets as G

P = G.P_BLS12_381_FR


def shifted_blocks(singles):
    """The aggregate's (A, B, C) predicted from the statements' own matrices [(ni, nw, (A, B, C)), ...]."""
    ni_tot = 1 + sum(ni - 1 for ni, _, _ in singles)
    out = ([], [], [])
    pub = wit = 0
    for ni, nw, mats in singles:
        def col(c):
            return 0 if c == 0 else (pub + c if c < ni else ni_tot + wit + (c - ni))
        for k in range(3):
            out[k].extend(sorted((col(c), v) for c, v in row) for row in mats[k])
        pub += ni - 1
        wit += nw
    return out


@pytest.mark.parametrize("logns", [(9, 9), (10, 9)])
def test_aggregate_system_is_the_block_arrangement_of_its_statements(logns):
    rng = random.Random(100 + sum(logns))
    triples = [T.random_triple(l, rng)[:3] for l in logns]
    statements = [(s.tolist(), p.tolist(), h.tolist(), l) for (s, p, h), l in zip(triples, logns)]
    cs = G.run_reference_flow_aggregate(statements, strict=True)
    singles = [G.run_reference_flow(*st, strict=True) for st in statements]
    assert cs.num_instance_variables() == 1 + sum(2 << l for l in logns)
    assert cs.num_witness_variables() == sum(c.num_witness_variables() for c in singles)
    assert cs.num_constraints() == sum(c.num_constraints() for c in singles)
    assert cs.is_satisfied()
    assert cs.instance_assignment == [1] + [v for c in singles for v in c.instance_assignment[1:]]
    assert cs.witness_assignment == [v for c in singles for v in c.witness_assignment]
    if logns == (9, 9):                                              # the inlining of 2.5 M terms per Falcon-1024 statement: once is enough
        want = shifted_blocks([(c.num_instance_variables(), c.num_witness_variables(), c.to_matrices()) for c in singles])
        got = cs.to_matrices()
        for k in range(3):
            assert got[k] == want[k], "matrix %d of the aggregate is not the block arrangement" % k
    # one statement's witness spoilt: the whole statement is unsatisfied, in that statement's rows
    cs.witness_assignment[singles[0].num_witness_variables() + 7] ^= 1
    bad = cs.which_is_unsatisfied()
    assert bad is not None and bad >= singles[0].num_constraints()
