"""Time the Groth16 h_acc multi-scalar multiplication (frw_groth16_msm_h_dev) on resident witness maps: a proving key's
h_query made on the device from toy toxic waste (frw_g1_fixed_base), h from the witness map of synthetic signatures.

usage: python tools/time_msm.py [logn=10] [batch=16] [reps=5]      (rocprofv3 --kernel-trace --stats on it for the split)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import falcon_r1cs_amd as frw

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def h_query(eng, n, t=0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % R, delta=987654321):
    """h_query[i] = (zt / delta) t^i G1 (ark-groth16 generator.rs), on the device."""
    c = (pow(t, n, R) - 1) * pow(delta, -1, R) % R
    ks, x = [], c
    for _ in range(n - 1):
        ks.append(x)
        x = x * t % R
    lim = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    return eng.g1_fixed_base(lim)


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
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
    r = eng.r1cs_load(0, logn)
    q = eng.qap_info(r)
    n = int(q.domain_size)
    ws = torch.empty(batch * int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
    h = torch.empty((batch, n, 4), dtype=torch.int64, device=dev)
    eng.qap_witness_map_dev(r, batch, wit, inst, h, ws, ws.numel(), None, s0)
    torch.cuda.synchronize()
    eng.r1cs_free(r)
    del ws
    t0 = time.time()
    bases = h_query(eng, n)
    t1 = time.time()
    m = eng.msm_g1_load(bases)
    torch.cuda.synchronize()
    info = eng.msm_info(m)
    print("h_query on the device: %.2f s (%d points); window table %.0f MB in %.2f s" % (t1 - t0, n - 1, info.table_bytes / 1e6, time.time() - t1), flush=True)
    mws = torch.empty(batch * int(info.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
    out = torch.empty((batch, 12), dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    run = lambda: eng.groth16_msm_h_dev(m, batch, h, n, out, mws, mws.numel(), s0)
    run()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    adds = 16 * (n - 1)
    print("Falcon-%d, %d signatures per call, %d points: %.3f ms/call = %.1f us/signature = %.1f h_acc/s; %.2f G mixed additions/s"
          % (1 << logn, batch, n - 1, ms, 1e3 * ms / batch, batch / ms * 1e3, adds * batch / ms / 1e6))
    eng.msm_free(m)


if __name__ == "__main__":
    main()
