#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (frw_witness_ntt_verify): pageable host memory in, host memory
out.  Reported in DESIGN.md next to the HBM-resident `value` of bench.py; never used as `value`."""
import sys
import time
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import falcon_r1cs_amd as frw  # noqa: E402

eng = frw.WitnessEngine(0)
for logn, batch in ((10, 1024), (9, 2048)):
    sig, pk, hm = frw.synth_triples(logn, batch, seed=3)
    eng.witness_ntt_verify(logn, sig[:8], pk[:8], hm[:8])          # warm-up
    t0 = time.perf_counter()
    wit, inst, st = eng.witness_ntt_verify(logn, sig, pk, hm)
    dt = time.perf_counter() - t0
    gb = (wit.nbytes + inst.nbytes) / 1e9
    print("falcon-%d host path, pageable outputs: %d signatures in %.3f s = %.0f signatures/s, %.1f GB/s of D2H (includes "
          "the caller's numpy.zeros of the outputs)" % (1 << logn, batch, dt, batch / dt, gb / dt))
    eng.witness_ntt_verify(logn, sig[:8], pk[:8], hm[:8], pinned=True)
    L = frw.layout(logn)
    import ctypes as C
    import numpy as np
    wit_p = eng.pinned_empty((batch, L.num_witness, 4), np.uint64)
    inst_p = eng.pinned_empty((batch, L.num_instance, 4), np.uint64)
    st_p = eng.pinned_empty((batch,), np.int32)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = eng._lib.frw_witness_ntt_verify(eng._ctx, logn, batch, P(sig), P(pk), P(hm), 1, P(wit_p), P(inst_p), P(st_p), 1)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    assert (wit_p == wit).all() and (inst_p == inst).all()
    print("falcon-%d host path, pinned outputs (frw_host_alloc): %d signatures in %.3f s = %.0f signatures/s, %.1f GB/s of D2H"
          % (1 << logn, batch, best, batch / best, gb / best))
