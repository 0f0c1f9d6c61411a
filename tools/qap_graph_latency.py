"""Latency of ONE witness map (one signature): plain stream launches vs the same call captured in a HIP graph
(torch.cuda.CUDAGraph) and replayed.  The QAP entry points make no allocation when they can lend their workspace to the
products, so the whole call is capturable.   python tools/qap_graph_latency.py [logn=10] [reps=200]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import falcon_r1cs_amd as frw

logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda:0")
eng = frw.WitnessEngine(0)
L = frw.layout(logn)
sig, pk, hm = frw.synth_triples(logn, 1, seed=5)
d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
wit = torch.empty((1, L.num_witness, 4), dtype=torch.int64, device=dev)
inst = torch.empty((1, L.num_instance, 4), dtype=torch.int64, device=dev)
st = torch.empty(1, dtype=torch.int32, device=dev)
eng.witness_ntt_verify_dev(logn, 1, d[0], d[1], d[2], wit, inst, st, 1, 0)
r = eng.r1cs_load(0, logn)
q = eng.qap_info(r)
n, per = int(q.domain_size), int(q.workspace_bytes_per_signature)
ws = torch.empty(per, dtype=torch.uint8, device=dev)
h = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
h2 = torch.empty_like(h)
bad = torch.zeros(1, dtype=torch.int32, device=dev)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


for name, call in (("witness map (7 transforms)", eng.qap_witness_map_dev), ("quotient (6 transforms)", eng.qap_quotient_dev)):
    plain = timed(lambda: call(r, 1, wit, inst, h, ws, per, bad, torch.cuda.current_stream().cuda_stream))
    g = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        call(r, 1, wit, inst, h2, ws, per, bad, side.cuda_stream)          # warm the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=side):
            call(r, 1, wit, inst, h2, ws, per, bad, torch.cuda.current_stream().cuda_stream)
    graph = timed(g.replay)
    torch.cuda.synchronize()
    assert torch.equal(h, h2) and int(bad.item()) == 0
    print("Falcon-%d, one signature, %-28s %7.1f us per call on a stream, %7.1f us as a replayed graph (same h)"
          % (1 << logn, name + ":", plain, graph))
eng.r1cs_free(r)
