"""Golden vectors of the R1CS -> QAP witness map for the committed witness fixtures (tests/golden/witness_*.json).

Everything here comes from the ORACLE side: the constraint system and its witness from oracle/falcon_gadgets.py on
oracle/ark_sim.py (matrices by the oracle's own LC inlining), the map from oracle/qap_oracle.c, and the FFT-free identity
A(tau) B(tau) - C(tau) = h(tau) (tau^n - 1) evaluated by oracle/qap.py as the cross-check before anything is written.
The product is not involved.   python tests/golden/make_qap.py   (about two minutes)"""
import hashlib
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, ".."))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
import frw_testlib as T  # noqa: E402
from oracle import falcon_gadgets as G  # noqa: E402
from oracle import qap  # noqa: E402

P = qap.P
TAU = 0x1234567890ABCDEF1234567890ABCDEF1234567890ABCDEF


def sha(b):
    return hashlib.sha256(b).hexdigest()


def fixture(oracle, path):
    fx = json.load(open(path))
    logn = fx["logn"]
    sig, pk, hm = (np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16) for k in ("sig", "pk", "hm"))
    cs = G.run_reference_flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    assert cs.is_satisfied()
    ni = cs.num_instance_variables()
    z_int = cs.instance_assignment + cs.witness_assignment
    z = T.ints_to_limbs(z_int)
    prods = []
    for rows in cs.to_matrices():
        ptr = np.cumsum([0] + [len(r) for r in rows]).astype(np.uint64)
        col = np.fromiter((c for r in rows for c, _ in r), dtype=np.uint32)
        val = T.ints_to_limbs([v for r in rows for _, v in r])
        prods.append(oracle.qap_matvec(ptr, col, val, z))
    h = oracle.qap_witness_map(*prods, ni, z)
    h_int = T.limbs_to_ints(h)
    lhs, rhs = qap.check_identity(*(T.limbs_to_ints(a) for a in prods), ni, z_int[:ni], h_int, TAU)
    assert lhs == rhs and h_int[-1] == 0
    mont = T.ints_to_limbs([v * qap.R_MONT % P for v in h_int])
    return {"witness_fixture": os.path.basename(path), "logn": logn, "log_domain_size": int(np.log2(h.shape[0])),
            "num_constraints": cs.num_constraints(), "num_instance": ni,
            "h_sha256": {"canonical": sha(h.tobytes()), "montgomery": sha(mont.tobytes())},
            "h_first4": [str(v) for v in h_int[:4]], "h_at_n_minus_2": str(h_int[-2]),
            "identity_tau": hex(TAU), "identity_value": str(lhs)}


def main():
    oracle = T.load_oracle()
    out = {"description": "h = (A B - C) / (X^n - 1) (ark-groth16 0.3.0 R1CStoQAP::witness_map, restated in oracle/qap.py) "
                          "for the committed witness fixtures; h as uint64[n][4] little-endian, canonical / Montgomery",
           "cases": []}
    for name in ("witness_logn9_seed101.json", "witness_logn10_seed201.json"):
        fx = fixture(oracle, os.path.join(HERE, name))
        out["cases"].append(fx)
        print("wrote", name, fx["h_sha256"]["montgomery"][:16], flush=True)
    json.dump(out, open(os.path.join(HERE, "qap.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
