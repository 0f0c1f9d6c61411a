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
    # FRW_ENC_COMPACT over PCIe (0.51 MB instead of 5.08 MB per Falcon-1024 signature), pinned outputs, and its host-side
    # expansion (frw_expand_host, single thread and all cores)
    from concurrent.futures import ThreadPoolExecutor
    cb = 8 * batch
    s8, p8, h8 = frw.synth_triples(logn, cb, seed=4)
    CL = frw.compact_layout(logn)
    comp_p = eng.pinned_empty((cb, CL.bytes_per_signature), np.uint8)
    st8 = eng.pinned_empty((cb,), np.int32)
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        rc = eng._lib.frw_witness_ntt_verify(eng._ctx, logn, cb, P(s8), P(p8), P(h8), 2, P(comp_p), None, P(st8), 1)
        best = min(best, time.perf_counter() - t0)
        assert rc == 0
    print("falcon-%d host path, FRW_ENC_COMPACT, pinned outputs: %d signatures in %.3f s = %.0f signatures/s, %.1f GB/s of D2H"
          % (1 << logn, cb, best, cb / best, comp_p.nbytes / 1e9 / best))
    w2, i2 = np.empty_like(wit), np.empty_like(inst)
    t0 = time.perf_counter()
    assert eng._lib.frw_expand_host(logn, batch, P(comp_p), P(w2), P(i2)) == 0
    dt1 = time.perf_counter() - t0
    nthreads = os.cpu_count() or 1
    step = max(1, batch // nthreads)

    def part(lo):
        hi = min(batch, lo + step)
        assert eng._lib.frw_expand_host(logn, hi - lo, P(comp_p[lo:hi]), P(w2[lo:hi]), P(i2[lo:hi])) == 0
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthreads) as ex:
        list(ex.map(part, range(0, batch, step)))
    dtn = time.perf_counter() - t0
    w_ref, i_ref, _ = eng.witness_ntt_verify(logn, s8[:batch], p8[:batch], h8[:batch])
    assert (w2 == w_ref).all() and (i2 == i_ref).all()
    print("falcon-%d frw_expand_host: %.0f signatures/s on one thread, %.0f signatures/s on %d threads (%.1f GB/s written)"
          % (1 << logn, batch / dt1, batch / dtn, nthreads, gb / dtn))
