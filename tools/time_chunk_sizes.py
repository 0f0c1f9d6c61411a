#!/usr/bin/env python3
"""Witness-kernel bandwidth vs signatures per launch, interleaved in one process (one 16,384-signature buffer)."""
import os
import statistics
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import falcon_r1cs_amd as frw  # noqa: E402

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
eng = frw.WitnessEngine(0)
dev = torch.device("cuda:0")
stream = torch.cuda.current_stream()
L = frw.layout(logn)
cap = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
sig, pk, hm = frw.synth_triples(logn, cap, seed=1)
d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
wit = torch.empty((cap, L.num_witness, 4), dtype=torch.int64, device=dev)
inst = torch.empty((cap, L.num_instance, 4), dtype=torch.int64, device=dev)
st = torch.empty(cap, dtype=torch.int32, device=dev)
sizes = [b for b in (1024, 2048, 3072, 4096, 4608, 6144, 8192, 12288, 16384, 24576, 32768, 49152) if b <= cap]
res = {b: [] for b in sizes}
ws = {b: [] for b in sizes}
bytes_per = 32 * (L.num_witness + 2 * L.n) + 6 * L.n
for rnd in range(6):
    for b in sizes:
        reps = max(1, cap // b)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            eng.witness_ntt_verify_dev(logn, b, d[0], d[1], d[2], wit, inst, st, 1, stream.cuda_stream)
        e1.record(stream)
        e1.synchronize()
        if rnd:
            res[b].append(e0.elapsed_time(e1) / reps)
        # compute-free write stream over the same footprint
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(reps):
            eng.diag_write_stream_dev(wit, b * L.num_witness * 32, L.num_witness * 32, stream.cuda_stream)
        e1.record(stream)
        e1.synchronize()
        if rnd:
            ws[b].append(e0.elapsed_time(e1) / reps)
for b in sizes:
    ms = statistics.median(res[b])
    wms = statistics.median(ws[b])
    print("falcon-%d  %6d signatures/launch: %.4f ms  %.1f GB/s  %.0f signatures/s   | write stream, same footprint: %.1f GB/s" %
          (1 << logn, b, ms, b * bytes_per / ms / 1e6, b / ms * 1e3, b * L.num_witness * 32 / wms / 1e6))
