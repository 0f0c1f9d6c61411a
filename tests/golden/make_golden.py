#!/usr/bin/env python3
"""Regenerates the fixtures in tests/golden/.  Run from the repo root: python tests/golden/make_golden.py

The reference (Rust) cannot run in this environment and holds no golden vectors, so these fixtures are produced
by the ORACLE: oracle/falcon_gadgets.py executed against oracle/ark_sim.py (gadget-by-gadget restatement on a
simulation of the arkworks front end).  They pin the closed-form C oracle and the HIP path to that execution.

* witness_*.json  -- one seeded (sig, pk, hm) triple each: the inputs (hex, u16 LE), SHA-256 of the witness and
                     instance assignment bytes in both encodings, the counts, and a few sampled elements
                     (the first mod_q quotients `t`, 132-/146-bit integers, as decimal strings).
* dual_*.json     -- the same for FalconDualNTTVerificationCircuit (falcon_dual_ntt.rs).
* prepare.json    -- input preparation: (nonce, msg) -> hash_to_point digests, encoded key / signature -> coefficient
                     digests (oracle/falcon_codec.py; SHAKE256 from hashlib).
* ntt_table.json  -- SHA-256 of falcon-rust's NTT_TABLE as the oracle regenerates it (7^bitrev10(i) mod q).
                     When /root/reference is present the script ALSO parses script/ntt_param.sage:3-132
                     (Falcon's GMb table, the reference's own data) and checks forward[i]/4091 mod q against the
                     regenerated table before writing -- that is the one place the reference pins this data.
"""
import hashlib
import json
import os
import random
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import falcon_gadgets as G  # noqa: E402
import frw_testlib as T  # noqa: E402
import numpy as np  # noqa: E402


def sha(b):
    return hashlib.sha256(b).hexdigest()


def ntt_table_fixture():
    table = G.NTT_TABLE
    checked = False
    sage = "/root/reference/script/ntt_param.sage"
    if os.path.exists(sage):
        text = open(sage).read()
        m = re.search(r"forward\s*=\s*\[(.*?)\]", text, re.S)
        fwd = [int(x) for x in re.findall(r"\d+", m.group(1))]
        assert len(fwd) == 1024
        inv = pow(4091, -1, G.MODULUS)
        assert [f * inv % G.MODULUS for f in fwd] == table, "NTT_TABLE != reference forward/4091"
        m = re.search(r"reverse\s*=\s*\[(.*?)\]", text, re.S)
        rev = [int(x) for x in re.findall(r"\d+", m.group(1))]
        assert [r * inv % G.MODULUS for r in rev] == [pow(7, (2048 - G.bitrev10(i)) % 2048, G.MODULUS) for i in range(1024)]
        checked = True
    blob = np.array(table, dtype=np.uint16).tobytes()
    return {"description": "falcon-rust NTT_TABLE = 7^bitrev10(i) mod 12289, u16 LE", "sha256": sha(blob),
            "first8": table[:8], "checked_against_reference_sage": checked}


def witness_fixture(logn, seed):
    rng = random.Random(seed)
    sig, pk, hm, v = T.random_triple(logn, rng)
    cs = G.run_reference_flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    assert cs.is_satisfied()
    n = 1 << logn
    t_off = 29 * n          # first mod_q block of S3: [t, b, ...]
    return {
        "logn": logn, "seed": seed,
        "sig": sig.tobytes().hex(), "pk": pk.tobytes().hex(), "hm": hm.tobytes().hex(),
        "num_instance": cs.num_instance_variables(), "num_witness": cs.num_witness_variables(),
        "num_constraints": cs.num_constraints(),
        "witness_sha256": {"canonical": sha(G.encode_elements(cs.witness_assignment, False)),
                           "montgomery": sha(G.encode_elements(cs.witness_assignment, True))},
        "instance_sha256": {"canonical": sha(G.encode_elements(cs.instance_assignment, False)),
                            "montgomery": sha(G.encode_elements(cs.instance_assignment, True))},
        "sample_t": [str(cs.witness_assignment[t_off + 29 * k]) for k in range(4)],
        "sample_b": [cs.witness_assignment[t_off + 29 * k + 1] for k in range(4)],
        "l2_norm": T.centred_norm(sig, v),
        # FRW_ENC_COMPACT of the same assignment (integers + boolean bit array, include/frw.h), re-laid out on the host
        **compact_fixture(logn, cs),
    }


def compact_fixture(logn, cs):
    as_u64 = lambda vals: np.frombuffer(G.encode_elements(vals, True), dtype=np.uint64).reshape(-1, 4)
    comp = T.compact_from_witness(logn, as_u64(cs.witness_assignment), as_u64(cs.instance_assignment))
    return {"compact_sha256": sha(comp), "compact_bytes": len(comp)}


def dual_fixture(logn, seed):
    """FalconDualNTTVerificationCircuit (falcon_dual_ntt.rs) on a seeded triple."""
    rng = random.Random(seed)
    sig, pk, hm, v = T.random_triple(logn, rng)
    cs = G.run_reference_flow_dual(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    assert cs.is_satisfied()
    return {
        "circuit": "dual", "logn": logn, "seed": seed,
        "sig": sig.tobytes().hex(), "pk": pk.tobytes().hex(), "hm": hm.tobytes().hex(),
        "num_instance": cs.num_instance_variables(), "num_witness": cs.num_witness_variables(),
        "num_constraints": cs.num_constraints(),
        "witness_sha256": {"canonical": sha(G.encode_elements(cs.witness_assignment, False)),
                           "montgomery": sha(G.encode_elements(cs.witness_assignment, True))},
        "instance_sha256": {"canonical": sha(G.encode_elements(cs.instance_assignment, False)),
                            "montgomery": sha(G.encode_elements(cs.instance_assignment, True))},
    }


def prepare_fixture():
    """Input preparation (oracle/falcon_codec.py): fixed (nonce, msg) pairs -> hash_to_point; encoded key/signature."""
    from oracle import falcon_codec as K
    rng = random.Random(77)
    cases = []
    for logn in (9, 10):
        n = 1 << logn
        for mlen in (0, 15, 96, 97, 300):
            nonce = bytes(rng.randrange(256) for _ in range(40))
            msg = bytes(rng.randrange(256) for _ in range(mlen))
            hm = K.hash_to_point(nonce, msg, logn)
            cases.append({"logn": logn, "nonce": nonce.hex(), "msg": msg.hex(), "hm_first8": hm[:8],
                          "hm_sha256": sha(np.array(hm, dtype=np.uint16).tobytes())})
        pk = [rng.randrange(G.MODULUS) for _ in range(n)]
        s2 = [max(-2047, min(2047, round(rng.gauss(0, T.SIGMA[logn])))) for _ in range(n)]
        nonce = bytes(rng.randrange(256) for _ in range(40))
        cases.append({"logn": logn, "pk_bytes": K.modq_encode(pk, logn).hex(), "sig_bytes": K.comp_encode(s2, logn, nonce).hex(),
                      "pk_sha256": sha(np.array(pk, dtype=np.uint16).tobytes()),
                      "sig_sha256": sha(np.array([x % G.MODULUS for x in s2], dtype=np.uint16).tobytes())})
    return {"description": "SHAKE256 hash-to-point and Falcon codecs (Falcon spec; hashlib SHAKE256)", "cases": cases}


def main():
    json.dump(prepare_fixture(), open(os.path.join(HERE, "prepare.json"), "w"), indent=1)
    for logn, seed in [(9, 301), (10, 302)]:
        fx = dual_fixture(logn, seed)
        json.dump(fx, open(os.path.join(HERE, "dual_logn%d_seed%d.json" % (logn, seed)), "w"), indent=1)
        print("wrote dual", logn, seed, fx["witness_sha256"]["montgomery"][:16])
    json.dump(ntt_table_fixture(), open(os.path.join(HERE, "ntt_table.json"), "w"), indent=1)
    for logn, seed in [(9, 101), (9, 102), (10, 201)]:
        fx = witness_fixture(logn, seed)
        json.dump(fx, open(os.path.join(HERE, "witness_logn%d_seed%d.json" % (logn, seed)), "w"), indent=1)
        print("wrote", logn, seed, fx["witness_sha256"]["montgomery"][:16])


if __name__ == "__main__":
    main()
