#!/usr/bin/env python3
"""Soak: many back-to-back launches of every kernel on fixed inputs, two streams at a time (the store-data hazard of
round 2 only showed under contention); every launch's per-signature digests must equal the oracle's for a sample and the
first launch's for all (a rare race in the LDS aliasing, the tile writer or a hazard would show up as a mismatch)."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import falcon_r1cs_amd as frw  # noqa: E402
import frw_testlib  # noqa: E402

launches = int(sys.argv[1]) if len(sys.argv) > 1 else 300
FULL = len(sys.argv) > 2 and sys.argv[2] == "full"       # also: the benchmark's launch shape, one buffer per stream would not fit
oracle = frw_testlib.load_oracle()
eng = frw.WitnessEngine(0)
dev = torch.device("cuda:0")
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
CASES = [(10, 3000, "ntt"), (9, 5000, "ntt"), (10, 1500, "dual"), (9, 2500, "dual"), (10, 3000, "compact"), (9, 4096, "ntt_modq")]
if FULL:
    CASES = [(10, 16384, "ntt"), (10, 16384, "compact")]     # two 82 GB buffers, 21.33 rounds + split tail per launch
for logn, batch, mode in CASES:
    dual = mode == "dual"
    L = frw.layout_dual(logn) if dual else frw.layout(logn)
    CL = frw.compact_layout(logn)
    n = 1 << logn
    words = 29 * n * 4 if mode == "ntt_modq" else L.num_witness * 4
    sig, pk, hm = frw.synth_triples(logn, batch, seed=123)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    bufs = []
    for s in streams:
        bufs.append({"wit": torch.empty((batch, words // 4, 4), dtype=torch.int64, device=dev),
                     "inst": torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev),
                     "st": torch.empty(batch, dtype=torch.int32, device=dev),
                     "comp": torch.empty((batch, CL.bytes_per_signature), dtype=torch.uint8, device=dev) if mode == "compact" else None,
                     "out": torch.empty((batch, n), dtype=torch.int16, device=dev)})
    # oracle digests of a sample
    idx = list(range(0, batch, max(1, batch // 48)))
    if mode == "ntt_modq":
        ow, _ = oracle.ntt_modq(logn, pk[idx], 1)
    elif dual:
        ow, _, _ = oracle.witness_dual_ntt_verify(logn, sig[idx], pk[idx], hm[idx], 1)
    else:
        ow, _, _ = oracle.witness_ntt_verify(logn, sig[idx], pk[idx], hm[idx], 1, threads=8)
    want = torch.tensor([oracle.digest(ow[j]) for j in range(len(idx))], dtype=torch.uint64).view(torch.int64).to(dev)
    tidx = torch.tensor(idx, device=dev)
    ref = None
    t0 = time.time()
    for i in range(launches):
        b, s = bufs[i & 1], streams[i & 1]
        dig = torch.zeros(batch, dtype=torch.int64, device=dev)
        with torch.cuda.stream(s):
            if i % 50 < 2:
                b["wit"].fill_(0x5a5a5a5a)          # stale data must never survive
            if mode == "ntt":
                eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], b["wit"], b["inst"], b["st"], 1, s.cuda_stream)
            elif dual:
                eng.witness_dual_ntt_verify_dev(logn, batch, d[0], d[1], d[2], b["wit"], b["inst"], b["st"], 1, s.cuda_stream)
            elif mode == "compact":
                eng.witness_ntt_verify_compact_dev(logn, batch, d[0], d[1], d[2], b["comp"], b["st"], s.cuda_stream)
                eng.expand_dev(logn, batch, b["comp"], b["wit"], b["inst"], s.cuda_stream)
            else:
                eng.ntt_modq_dev(logn, batch, d[1], b["wit"], b["out"], b["st"], 1, s.cuda_stream)
            eng.digest_dev(b["wit"], words, batch, dig, s.cuda_stream)
            if ref is None:
                s.synchronize()
                ref = dig
                assert torch.equal(dig[tidx], want), "first launch differs from the oracle"
            else:
                ok = torch.equal(ref, dig)            # synchronises this stream; the other keeps running
                if not ok:
                    bad = torch.nonzero(ref != dig).flatten().tolist()
                    raise SystemExit("MISMATCH in %s launch %d, signatures %s" % (mode, i, bad[:10]))
    torch.cuda.synchronize()
    print("falcon-%d %s: %d launches x %d items on two streams, all identical and equal to the oracle on %d samples (%.1f s)"
          % (n, mode, launches, batch, len(idx), time.time() - t0), flush=True)
