#!/usr/bin/env python3
"""Soak of the prover: many back-to-back frw_groth16_prove_dev calls on fixed witnesses and fixed blinding factors, from two
host threads at once on two streams (one key: the calls take turns on the key's four side streams; each has its own
workspace and output), every call's proofs compared with the first call's byte for byte, and at the end every proof put to
the product's verifier.  A race between the side streams, the events that fork and join them, or the workspaces of the five
sums would show up as a differing or rejected proof.   python tools/soak_groth16.py [calls=100] [logn=10] [batch=16]"""
import os
import random
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import falcon_r1cs_amd as frw  # noqa: E402

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 100
logn = int(sys.argv[2]) if len(sys.argv) > 2 else 10
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 16
dev = torch.device("cuda:0")
eng = frw.WitnessEngine(0)
L = frw.layout(logn)
sig, pk, hm = frw.synth_triples(logn, batch, seed=271828)
d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
st = torch.empty(batch, dtype=torch.int32, device=dev)
eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, 0)
torch.cuda.synchronize()
assert int(st.abs().sum()) == 0
rng = random.Random(99)
key, vk = eng.groth16_setup(0, logn, *(rng.randrange(2, R) for _ in range(5)))
r1cs = eng.r1cs_load(0, logn)
ws_bytes = eng.groth16_workspace_bytes(key, r1cs, batch)
rs = np.frombuffer(b"".join(rng.randrange(R).to_bytes(32, "little") for _ in range(2 * batch)), dtype=np.uint64).reshape(batch, 2, 4)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
ws = [torch.empty(ws_bytes, dtype=torch.uint8, device=dev) for _ in streams]
out = [torch.zeros((batch, 48), dtype=torch.int64, device=dev) for _ in streams]
bad = [torch.zeros(batch, dtype=torch.int32, device=dev) for _ in streams]
eng.groth16_prove_dev(key, r1cs, batch, wit, inst, rs, out[0], ws[0], ws_bytes, bad[0], streams[0].cuda_stream)
torch.cuda.synchronize()
first = out[0].cpu().numpy().copy()
assert int(bad[0].abs().sum()) == 0
mismatches = [0, 0]
errors = []


def worker(k):
    try:
        for _ in range(calls):
            with torch.cuda.stream(streams[k]):
                out[k].zero_()
            eng.groth16_prove_dev(key, r1cs, batch, wit, inst, rs, out[k], ws[k], ws_bytes, bad[k], streams[k].cuda_stream)
            streams[k].synchronize()
            if not np.array_equal(out[k].cpu().numpy(), first) or int(bad[k].abs().sum()):
                mismatches[k] += 1
    except Exception as exc:                                                     # noqa: BLE001
        errors.append(repr(exc))


t0 = time.time()
threads = [threading.Thread(target=worker, args=(k,)) for k in range(2)]
for t in threads:
    t.start()
for t in threads:
    t.join()
dt = time.time() - t0
ver = frw.Groth16Verifier(vk)
accepted = ver.verify(inst.cpu().numpy().view(np.uint64), first.view(np.uint64))
ver.close()
print("Falcon-%d, %d proofs per call, 2 host threads x %d calls on two streams with one key: %d + %d calls differ from the first call, "
      "%d errors; %d / %d proofs accepted by frw_groth16_verify; %.1f s (%.0f proofs/s with both threads)"
      % (L.n, batch, calls, mismatches[0], mismatches[1], len(errors), int((accepted == 1).sum()), batch, dt, 2 * calls * batch / dt))
for e in errors:
    print("  ", e)
eng.r1cs_free(r1cs)
eng.groth16_pk_free(key)
sys.exit(1 if (sum(mismatches) or errors or int((accepted == 1).sum()) != batch) else 0)
