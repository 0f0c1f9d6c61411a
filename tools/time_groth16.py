"""Time frw_groth16_prove_dev (whole Groth16 proofs of resident witnesses) with a proving key made on the device from
exponents drawn here.  usage: python tools/time_groth16.py [logn=10] [batch=32] [reps=3]
(rocprofv3 --kernel-trace --stats on it for the split between the witness map, the five MSMs and the assembly)"""
import os
import random
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import falcon_r1cs_amd as frw

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    dev = torch.device("cuda:0")
    eng = frw.WitnessEngine(0)
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=1)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    r1cs = eng.r1cs_load(0, logn)
    n = int(eng.qap_info(r1cs).domain_size)
    ni, nw = L.num_instance, L.num_witness
    nv = ni + nw
    rng = random.Random(3)
    lim = lambda ks: np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    draw = lambda c: [rng.randrange(1, R) for _ in range(c)]
    t0 = time.time()
    f1, f2 = eng.g1_fixed_base(lim(draw(3))), eng.g2_fixed_base(lim(draw(2)))
    v = draw(nv)
    key = eng.groth16_pk_load(ni, nw, n, f1[0], f1[1], f1[2], f2[0], f2[1], eng.g1_fixed_base(lim(draw(nv))), eng.g1_fixed_base(lim(v)),
                              eng.g2_fixed_base(lim(v)), eng.g1_fixed_base(lim(draw(n - 1))), eng.g1_fixed_base(lim(draw(nw))))
    print("proving key (%d G1 + %d G2 points) made and loaded in %.2f s" % (2 * nv + nw + n + 2, nv + 2, time.time() - t0), flush=True)
    ws_bytes = eng.groth16_workspace_bytes(key, r1cs, batch)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    proofs = torch.empty((batch, 48), dtype=torch.int64, device=dev)
    rs = np.stack([lim(draw(2)) for _ in range(batch)])
    run = lambda: eng.groth16_prove_dev(key, r1cs, batch, wit, inst, rs, proofs, ws, ws_bytes, None, s0)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    print("Falcon-%d, %d proofs per call: %.2f ms/call = %.2f ms/proof = %.1f proofs/s" % (1 << logn, batch, ms, ms / batch, batch / ms * 1e3))
    if batch <= 4 and os.environ.get("FRW_TIME_GRAPH", "1") != "0":
        # the same call with the blinding factors in device memory (frw_groth16_prove_rs_dev: nothing waits on the host), on a
        # stream and as a captured HIP graph replayed
        d_rs = torch.from_numpy(rs.view(np.int64)).to(dev)
        proofs2 = torch.empty_like(proofs)
        run_dev = lambda s: eng.groth16_prove_rs_dev(key, r1cs, batch, wit, inst, d_rs, proofs2, ws, ws_bytes, None, s)
        run_dev(s0)
        torch.cuda.synchronize()
        assert torch.equal(proofs2, proofs)
        e0.record()
        for _ in range(reps):
            run_dev(s0)
        e1.record()
        torch.cuda.synchronize()
        ms_dev = e0.elapsed_time(e1) / reps
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            run_dev(side.cuda_stream)
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                run_dev(torch.cuda.current_stream().cuda_stream)
        proofs2.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(proofs2, proofs)
        e0.record()
        for _ in range(reps):
            graph.replay()
        e1.record()
        torch.cuda.synchronize()
        ms_graph = e0.elapsed_time(e1) / reps
        print("  blinding factors in device memory: %.2f ms/call on a stream, %.2f ms/call as a captured HIP graph replayed (same proofs)" % (ms_dev, ms_graph))


if __name__ == "__main__":
    main()
