"""Time the QAP witness map (frw_qap_witness_map_dev) on resident witnesses: signatures/s, ms per signature, and the split
between the sparse products and the transforms (the latter by timing a second call pattern is not possible from outside;
use rocprofv3 --kernel-trace --stats on this script for the per-kernel split).

usage: python tools/time_qap.py [logn=10] [batch=64] [reps=5]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import falcon_r1cs_amd as frw


def main():
    logn = int(sys.argv[1]) if len(sys.argv) > 1 else 10
    batch = int(sys.argv[2]) if len(sys.argv) > 2 else 64
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
    t0 = time.time()
    r = eng.r1cs_load(0, logn)
    print("r1cs_load (matrices + domain tables): %.2f s" % (time.time() - t0), flush=True)
    q = eng.qap_info(r)
    n, per = int(q.domain_size), int(q.workspace_bytes_per_signature)
    ws = torch.empty(batch * per, dtype=torch.uint8, device=dev)
    h = torch.empty((batch, n, 4), dtype=torch.int64, device=dev)
    bad = torch.empty(batch, dtype=torch.int32, device=dev)
    abc = torch.empty((batch, 3, L.num_constraints, 4), dtype=torch.int64, device=dev)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

    def timed(fn):
        fn()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps

    ms_all = timed(lambda: eng.qap_witness_map_dev(r, batch, wit, inst, h, ws, batch * per, bad, s0))
    ms_q = timed(lambda: eng.qap_quotient_dev(r, batch, wit, inst, h, ws, batch * per, bad, s0))
    ms_mv = timed(lambda: eng.r1cs_eval_dev(r, batch, wit, inst, bad, abc, s0))
    assert int(bad.abs().sum()) == 0
    print("Falcon-%d, %d signatures per call, domain 2^%d" % (1 << logn, batch, int(q.log_domain_size)))
    print("  witness map      %9.3f ms/call  %8.1f us/signature  %9.1f signatures/s" % (ms_all, 1e3 * ms_all / batch, batch / ms_all * 1e3))
    print("  six-transform quotient (h of a satisfied system) %9.3f ms/call  %8.1f us/signature  %9.1f signatures/s" % (ms_q, 1e3 * ms_q / batch, batch / ms_q * 1e3))
    print("  of which A z, B z, C z (frw_r1cs_eval_dev alone) %9.3f ms/call  %8.1f us/signature" % (ms_mv, 1e3 * ms_mv / batch))
    mul = 6 * (n // 2) * int(q.log_domain_size) + 5 * n            # textbook count of six radix-2 transforms (what both entry points run for these witnesses)
    print("  transforms: %.1f M Montgomery products per signature -> %.1f G products/s" % (mul / 1e6, mul * batch / ((ms_all - ms_mv) * 1e-3) / 1e9))
    eng.r1cs_free(r)


if __name__ == "__main__":
    main()
