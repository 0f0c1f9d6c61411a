"""Golden vectors of the Groth16 h_query multi-scalar multiplication for the committed witness fixtures.

All from the ORACLE side (the product is not involved): the witness map's h from oracle/qap_oracle.c for the fixture's
witness (as tests/golden/make_qap.py; its digest must equal the one committed in qap.json), a proving key's h_query from
KNOWN toxic waste (t, delta) -- h_query[i] = (zt / delta) t^i G1, ark-groth16 0.3.0 generator.rs, restated in
oracle/bls12_381.py -- and the expected sum WITHOUT any multi-scalar multiplication: sum h_i h_query[i] =
(h(t) zt / delta) G1, one scalar multiplication by Python integers.  As a cross-check before anything is written the C
bucket method (oracle/bls12_381.c) must reach the same point from the 2^17 - 1 / 2^18 - 1 actual bases.
    python tests/golden/make_msm.py      (a few minutes)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import frw_testlib as T  # noqa: E402
import make_qap  # noqa: E402
from oracle import bls12_381 as E  # noqa: E402
from oracle import falcon_gadgets as G  # noqa: E402

TOXIC = {"t": 0x1F3D5B79A8C6E4021F3D5B79A8C6E4021F3D5B79A8C6E4021F3D5B79A8C6E402 % E.R,
         "delta": 0x2468ACE013579BDF2468ACE013579BDF2468ACE013579BDF2468ACE013579BDF % E.R}


def h_query_scalars(n, toxic=TOXIC):
    """(zt / delta) t^i for i < n - 1 (generator.rs: h_query), as Python integers."""
    zt = (pow(toxic["t"], n, E.R) - 1) % E.R
    c = zt * pow(toxic["delta"], -1, E.R) % E.R
    out, x = [], c
    for _ in range(n - 1):
        out.append(x)
        x = x * toxic["t"] % E.R
    return out, c


def expected_point(h_int, n, toxic=TOXIC):
    _, c = h_query_scalars(2, toxic)            # c depends on n through zt: recompute properly below
    zt = (pow(toxic["t"], n, E.R) - 1) % E.R
    c = zt * pow(toxic["delta"], -1, E.R) % E.R
    acc = 0
    for v in reversed(h_int[:n - 1]):
        acc = (acc * toxic["t"] + v) % E.R
    return E.mul(E.G1, acc * c % E.R)


def fixture_h(oracle, name):
    fx = json.load(open(os.path.join(HERE, name)))
    logn = fx["logn"]
    sig, pk, hm = (np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16) for k in ("sig", "pk", "hm"))
    cs = G.run_reference_flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    ni = cs.num_instance_variables()
    z = T.ints_to_limbs(cs.instance_assignment + cs.witness_assignment)
    prods = []
    for rows in cs.to_matrices():
        ptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
        col = np.fromiter((c for r in rows for c, _ in r), dtype=np.uint32)
        val = T.ints_to_limbs([v for r in rows for _, v in r])
        prods.append(oracle.qap_matvec(ptr, col, val, z))
    return logn, oracle.qap_witness_map(*prods, ni, z)


def main():
    oracle = T.load_oracle()
    committed = {c["witness_fixture"]: c for c in json.load(open(os.path.join(HERE, "qap.json")))["cases"]}
    out = {"description": "sum_i h_i h_query[i] over BLS12-381 G1 for the committed witness fixtures, h_query from known toxic "
                          "waste: the affine result as ark-ff's bytes (12 x uint64 little-endian limbs, hex) and the digest of "
                          "the bases the sum was cross-checked with",
           "toxic": {k: hex(v) for k, v in TOXIC.items()}, "cases": []}
    for name in ("witness_logn9_seed101.json", "witness_logn10_seed201.json"):
        logn, h = fixture_h(oracle, name)
        assert make_qap.sha(h.tobytes()) == committed[name]["h_sha256"]["canonical"]
        n = h.shape[0]
        h_int = T.limbs_to_ints(h)
        want = expected_point(h_int, n)
        scalars, _ = h_query_scalars(n)
        bases = oracle.g1_fixed_base(T.ints_to_limbs(scalars), threads=os.cpu_count() or 1)
        got = oracle.g1_msm(bases, h[:n - 1], 13, threads=os.cpu_count() or 1)
        assert got.tolist() == E.to_limbs(want), "the bucket method over the actual bases disagrees with the MSM-free value"
        out["cases"].append({"witness_fixture": name, "logn": logn, "domain_size": n, "num_points": n - 1,
                             "bases_sha256": hashlib.sha256(bases.tobytes()).hexdigest(),
                             "h_acc": ["%016x" % v for v in E.to_limbs(want)]})
        print("wrote", name, out["cases"][-1]["h_acc"][0], flush=True)
    json.dump(out, open(os.path.join(HERE, "msm.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
