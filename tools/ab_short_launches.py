"""A/B for launches that are too short to reach the sustained write rate (BASELINE configs[1]: 4,096 Falcon-512 polynomials =
1.95 GB in ~0.33 ms; Falcon-512 full verify at 8,192 signatures per launch): back-to-back launches on ONE stream against the
same launches dealt over TWO streams, so that the tail of launch k (the last workgroups draining) overlaps the ramp of launch
k + 1.  Independent batches -- what a consumer with more than one batch in flight does anyway.
usage: python tools/ab_short_launches.py [launches=200]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import falcon_r1cs_amd as frw

HBM = 8000.0


def run(name, make_call, nbytes, launches, nstreams):
    streams = [torch.cuda.Stream() for _ in range(nstreams)]
    calls = [make_call(s) for s in streams]
    for c in calls:
        for _ in range(10):
            c()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(launches):
        calls[i % nstreams]()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = dt / launches * 1e3
    print("%-44s %d stream(s): %.4f ms/launch  %.0f GB/s  %.3f of spec" % (name, nstreams, ms, nbytes / ms / 1e6, nbytes / ms / 1e6 / HBM), flush=True)


def main():
    launches = int(sys.argv[1]) if len(sys.argv) > 1 else 200
    dev = torch.device("cuda:0")
    eng = frw.WitnessEngine(0)
    rng = np.random.default_rng(1)
    # configs[1]
    logn, batch = 9, 4096
    n = 1 << logn
    poly = torch.from_numpy(rng.integers(0, 12289, size=(batch, n), dtype=np.uint16).view(np.int16)).to(dev)

    def modq_call(stream):
        wit = torch.empty((batch, 29 * n, 4), dtype=torch.int64, device=dev)
        out = torch.empty((batch, n), dtype=torch.int16, device=dev)
        st = torch.empty(batch, dtype=torch.int32, device=dev)
        return lambda: eng.ntt_modq_dev(logn, batch, poly, wit, out, st, frw.ENC_MONTGOMERY, stream.cuda_stream)
    for ns in (1, 2, 3):
        run("ntt_modq falcon-512 x 4096", modq_call, batch * (32 * 29 * n + 2 * n), launches, ns)
    # Falcon-512 full verify, 8,192 per launch
    batch2 = 8192
    L = frw.layout(9)
    sig, pk, hm = frw.synth_triples(9, batch2, seed=5)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]

    def verify_call(stream):
        wit = torch.empty((batch2, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((batch2, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.empty(batch2, dtype=torch.int32, device=dev)
        return lambda: eng.witness_ntt_verify_dev(9, batch2, d[0], d[1], d[2], wit, inst, st, frw.ENC_MONTGOMERY, stream.cuda_stream)
    for ns in (1, 2):
        run("verify falcon-512 x 8192", verify_call, batch2 * (32 * (L.num_witness + 2 * L.n) + 6 * L.n), max(20, launches // 8), ns)
    # the dual circuit, Falcon-1024, 2,048 per launch (a ragged last round without a split tail)
    LD = frw.layout_dual(10)
    for b3 in (2048, 2500):
        s3, p3, h3 = frw.synth_triples(10, b3, seed=6)
        d3 = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (s3, p3, h3)]

        def dual_call(stream, b3=b3, d3=d3):
            wit = torch.empty((b3, LD.num_witness, 4), dtype=torch.int64, device=dev)
            inst = torch.empty((b3, LD.num_instance, 4), dtype=torch.int64, device=dev)
            st = torch.empty(b3, dtype=torch.int32, device=dev)
            return lambda: eng.witness_dual_ntt_verify_dev(10, b3, d3[0], d3[1], d3[2], wit, inst, st, frw.ENC_MONTGOMERY, stream.cuda_stream)
        for ns in (1, 2):
            run("dual falcon-1024 x %d" % b3, dual_call, b3 * (32 * (LD.num_witness + 2 * LD.n) + 6 * LD.n), max(20, launches // 8), ns)


if __name__ == "__main__":
    main()
