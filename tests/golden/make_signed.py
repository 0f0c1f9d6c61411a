#!/usr/bin/env python3
"""Regenerates tests/golden/falcon_signed.json: genuine Falcon (pk, msg, sig) triples.
Run from the repo root: python tests/golden/make_signed.py   (a few seconds)

The reference's end-to-end test signs with falcon-rust (KeyPair::keygen, sign_with_seed("test seed", "testing message"),
falcon_ntt.rs:133-160); falcon-rust is not under /root/reference and cannot be built here, and the reference holds no
signature vector.  The triples below come from oracle/falcon_sign.py, a restatement of the Falcon specification's key
generation and signing; each is checked with the specification's Verify before it is written.  Besides the encoded key,
message and signature the fixture records what the input-preparation step must produce from them (SHA-256 of the three
coefficient vectors) and what the hot path must produce (SHA-256 of the witness / instance assignment, Montgomery form,
from oracle/falcon_gadgets.py run strictly: the norm bound holds, the system is satisfied).
"""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import falcon_codec as K  # noqa: E402
from oracle import falcon_gadgets as G  # noqa: E402
from oracle import falcon_sign as S  # noqa: E402
import numpy as np  # noqa: E402

CASES = [(9, b"test seed", b"testing message"), (9, b"second key", b""),
         (10, b"test seed", b"testing message"), (10, b"second key", bytes(range(200)))]


def sha(b):
    return hashlib.sha256(b).hexdigest()


def make(logn, seed, msg):
    sk = S.keygen(logn, seed)
    pkb = sk.public_key_bytes()
    sgb = S.sign(sk, msg, seed)
    assert S.verify(pkb, msg, sgb, logn)
    nonce, sig = K.comp_decode(sgb, logn)
    pk = K.modq_decode(pkb, logn)
    hm = K.hash_to_point(nonce, msg, logn)
    cs = G.run_reference_flow(sig, pk, hm, logn, strict=True)
    assert cs.is_satisfied()
    u16 = lambda v: np.array(v, dtype=np.uint16).tobytes()
    return {"logn": logn, "key_seed": seed.hex(), "msg": msg.hex(), "pk_bytes": pkb.hex(), "sig_bytes": sgb.hex(),
            "sig_sha256": sha(u16(sig)), "pk_sha256": sha(u16(pk)), "hm_sha256": sha(u16(hm)),
            "witness_sha256_montgomery": sha(G.encode_elements(cs.witness_assignment, True)),
            "instance_sha256_montgomery": sha(G.encode_elements(cs.instance_assignment, True)),
            "num_witness": len(cs.witness_assignment)}


if __name__ == "__main__":
    out = {"description": "genuine Falcon signatures from oracle/falcon_sign.py (Falcon spec v1.2 keygen + sign), each "
                          "accepted by the spec's Verify; see tests/golden/make_signed.py",
           "cases": [make(*c) for c in CASES]}
    json.dump(out, open(os.path.join(HERE, "falcon_signed.json"), "w"), indent=1)
    print("wrote", len(out["cases"]), "cases")
