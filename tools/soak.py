#!/usr/bin/env python3
"""Soak: many back-to-back launches of both circuits on fixed inputs; every launch's per-signature digests must equal
the first launch's (a rare race in the work queue, the LDS aliasing or the tile writer would show up as a mismatch)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import falcon_r1cs_amd as frw  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
eng = frw.WitnessEngine(0)
dev = torch.device("cuda:0")
s0 = torch.cuda.current_stream().cuda_stream
for logn, batch, dual in ((10, 3000, False), (9, 5000, False), (10, 1500, True)):
    L = frw.layout_dual(logn) if dual else frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=123)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    ref = None
    t0 = time.time()
    fn = eng.witness_dual_ntt_verify_dev if dual else eng.witness_ntt_verify_dev
    for i in range(launches):
        dig = torch.zeros(batch, dtype=torch.int64, device=dev)
        wit.fill_(0x5a5a5a5a) if i % 50 == 0 else None          # stale data must never survive
        fn(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
        eng.digest_dev(wit, L.num_witness * 4, batch, dig, s0)
        if ref is None:
            ref = dig
        elif not torch.equal(ref, dig):
            bad = torch.nonzero(ref != dig).flatten().tolist()
            raise SystemExit("MISMATCH in launch %d, signatures %s" % (i, bad[:10]))
    torch.cuda.synchronize()
    print("falcon-%d%s: %d launches x %d signatures identical (%.1f s)" % (1 << logn, " dual" if dual else "", launches, batch, time.time() - t0))
