#!/usr/bin/env python3
"""What the reference's falcon-aggregate-sig crate would do (upstream it is a stub: falcon-aggregate-sig/src/main.rs:1-3), on the engine:
ONE Groth16 proof that ALL of several genuine Falcon signatures verify -- Falcon-512 and Falcon-1024 mixed.

    per statement      Polynomial::from(&sig), ::from(&pk), from_hash_of_message(msg, nonce)     -> frw_prepare_inputs per parameter set
                       (falcon_ntt.rs:27-28,44)
    the statement      FalconNTTVerificationCircuit::generate_constraints once per (pk, msg, sig)  -> frw_r1cs_load_aggregate (never synthesised)
                       on one constraint system (falcon_ntt.rs:26-123)
    setup, proof       as examples/pok_sig.rs:30-32                                               -> frw_groth16_setup_r1cs; frw_witness_ntt_verify_dev per
                                                                                                     parameter set, frw_aggregate_assign_dev, frw_groth16_prove_dev
    verification       public inputs pk_ntt_0 || hm_ntt_0 || pk_ntt_1 || ... (pok_sig.rs:38-45    -> frw_groth16_verify on the aggregate's instance vector;
                       per statement), Groth16::verify                                               and again with one statement's message hash changed

    python examples/aggregate_sig.py tests/golden/falcon_signed.json [--cases 0,2,3,1] [--seed 1] [--json]
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
    ap.add_argument("--cases", default="0,2,3,1", help="the statements, in order (indices into the file's cases)")
    ap.add_argument("--seed", type=int, default=1, help="seed of the toxic waste and the blinding factors (demonstration only)")
    ap.add_argument("--json", action="store_true", help="print the verifying key, public inputs and proof as JSON (hex limbs)")
    args = ap.parse_args()
    cases = json.load(open(args.signed))["cases"]
    statements = [cases[int(i)] for i in args.cases.split(",")]
    logns = [c["logn"] for c in statements]
    dev = torch.device("cuda:0")
    eng = frw.WitnessEngine(0)
    batches = {}
    for g in (9, 10):
        mine = [c for c in statements if c["logn"] == g]
        if not mine:
            continue
        L = frw.layout(g)
        sig, pk, hm, st = eng.prepare_inputs(g, [bytes.fromhex(c["pk_bytes"]) for c in mine], [bytes.fromhex(c["msg"]) for c in mine],
                                             [bytes.fromhex(c["sig_bytes"]) for c in mine])
        if st.any():
            raise SystemExit("malformed public key or signature encoding among the Falcon-%d statements" % L.n)
        d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
        wit = torch.empty((len(mine), L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((len(mine), L.num_instance, 4), dtype=torch.int64, device=dev)
        status = torch.empty(len(mine), dtype=torch.int32, device=dev)
        eng.witness_ntt_verify_dev(g, len(mine), d[0], d[1], d[2], wit, inst, status, frw.ENC_MONTGOMERY, 0)
        torch.cuda.synchronize()
        if status.any():
            raise SystemExit("Invalid input: a Falcon-%d signature fails its range checks (%s)" % (L.n, status.tolist()))
        batches[g] = (wit, inst)
    agg = eng.r1cs_load_aggregate(logns)
    info = eng.r1cs_info(agg)
    ni, nw = int(info.num_instance), int(info.num_witness)
    wit = torch.empty((1, nw, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((1, ni, 4), dtype=torch.int64, device=dev)
    b9, b10 = batches.get(9, (None, None)), batches.get(10, (None, None))
    eng.aggregate_assign_dev(agg, b9[0], b9[1], b10[0], b10[1], wit, inst, 0)
    rng = random.Random(args.seed)
    key, vk = eng.groth16_setup_r1cs(agg, *(rng.randrange(2, R) for _ in range(5)))           # circuit_specific_setup, for this statement shape
    ws_bytes = eng.groth16_workspace_bytes(key, agg, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    proof = torch.empty((1, 48), dtype=torch.int64, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    rs = np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(2)), dtype=np.uint64).reshape(1, 2, 4)
    eng.groth16_prove_dev(key, agg, 1, wit, inst, rs, proof, ws, ws_bytes, bad, 0)             # create_random_proof: ONE proof
    torch.cuda.synchronize()
    if int(bad[0]) != 0:
        raise SystemExit("the aggregate's witness violates %d constraints" % int(bad[0]))
    p = proof.cpu().numpy().view(np.uint64)[0]
    verifier = frw.Groth16Verifier(vk)
    inst_h = inst.cpu().numpy().view(np.uint64)
    accepted = int(verifier.verify(inst_h, p[None])[0])
    other = inst_h.copy()
    other[0, ni - 1, 0] ^= np.uint64(1)                                                      # the last statement's message hash, one coefficient
    accepted_other = int(verifier.verify(other, p[None])[0])
    verifier.close()
    if accepted != 1 or accepted_other != 0:
        raise SystemExit("verify_proof: %d for the statement proved, %d for another one" % (accepted, accepted_other))
    r_inv = pow(1 << 256, -1, R)
    public = [int.from_bytes(row.tobytes(), "little") * r_inv % R for row in inst_h[0, 1:]]
    hexl = lambda a: ["%016x" % int(v) for v in a]
    if args.json:
        print(json.dumps({"logn": logns, "verified": accepted == 1, "public_inputs": [str(x) for x in public],
                          "proof": {"a": hexl(p[:12]), "b": hexl(p[12:36]), "c": hexl(p[36:])},
                          "vk": {k: (hexl(v) if v.ndim == 1 else [hexl(r) for r in v]) for k, v in vk.items()}}))
    else:
        print("%d genuine Falcon signatures (%s): ONE Groth16 proof made on the device" % (len(logns), ", ".join("Falcon-%d" % (1 << g) for g in logns)))
        print("  %d constraints, %d witnesses, %d public inputs, QAP domain 2^%d" % (int(info.num_constraints), nw, len(public), int(info.log_domain_size)))
        print("  proof: %d bytes (A, B, C as ark-ff limbs); verify_proof: accepted; with one coefficient of the last message hash changed: rejected" % (48 * 8))
    eng.groth16_pk_free(key)
    eng.r1cs_free(agg)
    eng.close()


if __name__ == "__main__":
    main()
