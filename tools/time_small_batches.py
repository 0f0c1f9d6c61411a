#!/usr/bin/env python3
"""Latency of small batches through the device entry point (the reference's own use, examples/pok_sig.rs, proves ONE
signature): one workgroup owns a whole signature, so batch < resident workgroups leaves CUs idle by design."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import falcon_r1cs_amd as frw  # noqa: E402

eng = frw.WitnessEngine(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
for logn in (9, 10):
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, 1024, seed=1)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((1024, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((1024, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(1024, dtype=torch.int32, device=dev)
    for batch in (1, 8, 64, 256, 768, 1024):
        ts = []
        for rep in range(12):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, stream.cuda_stream)
            e1.record(stream)
            e1.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        us = ts[len(ts) // 2]
        print("falcon-%d batch %5d: %8.1f us  (%.0f signatures/s, %.0f GB/s)" %
              (1 << logn, batch, us, batch / us * 1e6, batch * L.num_witness * 32 / us / 1e3))
