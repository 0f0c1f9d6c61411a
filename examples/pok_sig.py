#!/usr/bin/env python3
"""Counterpart of the reference's examples/pok_sig.rs (a Groth16 proof of knowledge of a Falcon signature), on the engine.

    pok_sig.rs:13-22   keygen, sign "testing message", verify        -> a genuine (pk, msg, sig), read from a JSON file here
                                                                        (the product has no Falcon signer; tests/golden/falcon_signed.json
                                                                        holds triples made by the specification's algorithms)
    pok_sig.rs:24-31   build_circuit; circuit_specific_setup          -> frw_prepare_inputs; frw_groth16_setup (toxic waste from --seed)
    pok_sig.rs:32      create_random_proof                            -> frw_witness_ntt_verify_dev + frw_groth16_prove_dev
    pok_sig.rs:34-47   public inputs pk_ntt || hm_ntt; verify_proof   -> frw_groth16_verify (host pairing) on the instance buffer and the
                                                                        proof as the device wrote them; and again for a wrong statement

    python examples/pok_sig.py tests/golden/falcon_signed.json [--case 0] [--seed 1] [--json]
"""
import argparse
import json
import os
import random
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import falcon_r1cs_amd as frw

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("signed", help="JSON with cases of {logn, pk_bytes, msg, sig_bytes} (hex)")
    ap.add_argument("--case", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1, help="seed of the toxic waste and the blinding factors (a key whose toxic waste is known proves nothing: demonstration only)")
    ap.add_argument("--json", action="store_true", help="print the verifying key, public inputs and proof as JSON (hex limbs)")
    args = ap.parse_args()
    case = json.load(open(args.signed))["cases"][args.case]
    logn = case["logn"]
    dev = torch.device("cuda:0")
    eng = frw.WitnessEngine(0)
    L = frw.layout(logn)
    # Polynomial::from(&sig), Polynomial::from(&pk), from_hash_of_message(msg, nonce)        pok_sig.rs:24-28 -> falcon_ntt.rs:27-28,44
    sig, pk, hm, st = eng.prepare_inputs(logn, [bytes.fromhex(case["pk_bytes"])], [bytes.fromhex(case["msg"])], [bytes.fromhex(case["sig_bytes"])])
    if st.any():
        raise SystemExit("malformed public key or signature encoding")
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((1, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((1, L.num_instance, 4), dtype=torch.int64, device=dev)
    status = torch.empty(1, dtype=torch.int32, device=dev)
    eng.witness_ntt_verify_dev(logn, 1, d[0], d[1], d[2], wit, inst, status, frw.ENC_MONTGOMERY, 0)
    torch.cuda.synchronize()
    if int(status[0]) != 0:
        raise SystemExit("Invalid input: the signature fails its range checks (status %d)" % int(status[0]))
    rng = random.Random(args.seed)
    alpha, beta, gamma, delta, t = (rng.randrange(2, R) for _ in range(5))
    key, vk = eng.groth16_setup(0, logn, alpha, beta, gamma, delta, t)                       # circuit_specific_setup
    r1cs = eng.r1cs_load(0, logn)
    ws_bytes = eng.groth16_workspace_bytes(key, r1cs, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    proof = torch.empty((1, 48), dtype=torch.int64, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    rs = np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(2)), dtype=np.uint64).reshape(1, 2, 4)
    eng.groth16_prove_dev(key, r1cs, 1, wit, inst, rs, proof, ws, ws_bytes, bad, 0)           # create_random_proof
    torch.cuda.synchronize()
    if int(bad[0]) != 0:
        raise SystemExit("the witness violates %d constraints" % int(bad[0]))
    p = proof.cpu().numpy().view(np.uint64)[0]
    # Groth16::verify(&vk, &public_inputs, &proof)                                             pok_sig.rs:47
    verifier = frw.Groth16Verifier(vk)
    inst_h = inst.cpu().numpy().view(np.uint64)
    accepted = int(verifier.verify(inst_h, p[None])[0])
    other = inst_h.copy()
    other[0, 1, 0] ^= np.uint64(1)                                                           # another public key
    accepted_other = int(verifier.verify(other, p[None])[0])
    verifier.close()
    if accepted != 1 or accepted_other != 0:
        raise SystemExit("verify_proof: %d for the statement proved, %d for another one" % (accepted, accepted_other))
    # public inputs: pk_ntt || hm_ntt                                                        pok_sig.rs:38-45
    r_inv = pow(1 << 256, -1, R)
    public = [int.from_bytes(row.tobytes(), "little") * r_inv % R for row in inst[0, 1:].cpu().numpy().view(np.uint64)]
    hexl = lambda a: ["%016x" % int(v) for v in a]
    if args.json:
        print(json.dumps({"logn": logn, "verified": accepted == 1, "public_inputs": [str(x) for x in public], "proof": {"a": hexl(p[:12]), "b": hexl(p[12:36]), "c": hexl(p[36:])},
                          "vk": {k: (hexl(v) if v.ndim == 1 else [hexl(r) for r in v]) for k, v in vk.items()}}))
    else:
        print("Falcon-%d signature on %r: Groth16 proof made on the device" % (L.n, bytes.fromhex(case["msg"])))
        print("  %d constraints, %d witnesses, %d public inputs (pk_ntt || hm_ntt)" % (L.num_constraints, L.num_witness, len(public)))
        print("  A.x = 0x%s..." % "".join(reversed(hexl(p[:6])))[:48])
        print("  verify_proof: accepted; for another public key: rejected")
    eng.r1cs_free(r1cs)
    eng.groth16_pk_free(key)
    eng.close()


if __name__ == "__main__":
    main()
