#!/usr/bin/env python3
"""Latency of the call the reference's own consumers make: ONE signature per generate_constraints
(examples/constraint_counts.rs:61-63, examples/pok_sig.rs:24-32 -> rust/falcon-r1cs-accel/src/lib.rs), i.e.
frw_witness_ntt_verify with host buffers and batch = 1 (and 8, 64), for pageable outputs, page-locked outputs
(frw_host_alloc) and FRW_ENC_COMPACT; then frw_ntt_modq and frw_qap_witness_map at batch 1.  Prints median / best
microseconds per call and the allocation counters (the arena must not grow after the first call of a shape)."""
import ctypes as C
import os
import statistics
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import falcon_r1cs_amd as frw  # noqa: E402

P = lambda a: a.ctypes.data_as(C.c_void_p)
eng = frw.WitnessEngine(0)
lib, ctx = eng._lib, eng._ctx


def clock(fn, reps):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        ts.append((time.perf_counter() - t0) * 1e6)
    return statistics.median(ts), min(ts)


for logn in (10, 9):
    L, CL = frw.layout(logn), frw.compact_layout(logn)
    n = L.n
    for batch in (1, 8, 64):
        sig, pk, hm = frw.synth_triples(logn, batch, seed=11 + batch)
        ref_w, ref_i, _ = eng.witness_ntt_verify(logn, sig, pk, hm)
        for name, alloc in (("pageable", lambda shape, dt: np.zeros(shape, dtype=dt)), ("pinned", eng.pinned_empty)):
            wit, inst, st = alloc((batch, L.num_witness, 4), np.uint64), alloc((batch, L.num_instance, 4), np.uint64), alloc((batch,), np.int32)
            call = lambda: lib.frw_witness_ntt_verify(ctx, logn, batch, P(sig), P(pk), P(hm), 1, P(wit), P(inst), P(st), 1)
            assert call() == 0
            a0 = eng.host_allocations()
            med, best = clock(call, 200 if batch == 1 else 50)
            assert eng.host_allocations() == a0, "the arena grew after the first call"
            assert (wit == ref_w).all() and (inst == ref_i).all() and not st.any()
            print("falcon-%d frw_witness_ntt_verify batch=%-2d %-8s outputs: median %8.1f us, best %8.1f us per call = %8.1f us per "
                  "signature (%.1f GB/s out)" % (n, batch, name, med, best, med / batch, (wit.nbytes + inst.nbytes) / med / 1e3))
        comp, st = eng.pinned_empty((batch, CL.bytes_per_signature), np.uint8), eng.pinned_empty((batch,), np.int32)
        call = lambda: lib.frw_witness_ntt_verify(ctx, logn, batch, P(sig), P(pk), P(hm), 2, P(comp), None, P(st), 1)
        assert call() == 0
        a0 = eng.host_allocations()
        med, best = clock(call, 200 if batch == 1 else 50)
        assert eng.host_allocations() == a0
        print("falcon-%d frw_witness_ntt_verify batch=%-2d compact  (pinned): median %8.1f us, best %8.1f us per call = %8.1f us per "
              "signature" % (n, batch, med, best, med / batch))
    # NTTPolyVar::ntt_circuit alone, one polynomial
    poly = np.random.default_rng(5).integers(0, 12289, size=(1, n), dtype=np.uint16)
    wit, out, st = eng.pinned_empty((1, 29 * n, 4), np.uint64), eng.pinned_empty((1, n), np.uint16), eng.pinned_empty((1,), np.int32)
    call = lambda: lib.frw_ntt_modq(ctx, logn, 1, P(poly), 1, P(wit), P(out), P(st))
    assert call() == 0
    a0 = eng.host_allocations()
    med, best = clock(call, 200)
    assert eng.host_allocations() == a0
    print("falcon-%d frw_ntt_modq           batch=1  pinned   outputs: median %8.1f us, best %8.1f us" % (n, med, best))
    # the step after: frw_qap_witness_map for one witness held in host memory
    r = eng.r1cs_load(0, logn)
    q = eng.qap_info(r)
    sig, pk, hm = frw.synth_triples(logn, 1, seed=77)
    w, i, _ = eng.witness_ntt_verify(logn, sig, pk, hm, pinned=True)
    h = eng.pinned_empty((1, int(q.domain_size), 4), np.uint64)
    bad = np.zeros(1, dtype=np.uint32)
    call = lambda: lib.frw_qap_witness_map(r, 1, P(w), P(i), P(h), P(bad))
    assert call() == 0 and bad[0] == 0
    cnt = C.c_uint64()
    lib.frw_r1cs_diag_host_allocations(r, C.byref(cnt))
    a0 = cnt.value
    med, best = clock(call, 50)
    lib.frw_r1cs_diag_host_allocations(r, C.byref(cnt))
    assert cnt.value == a0 and not h[0, -1].any()
    print("falcon-%d frw_qap_witness_map    batch=1  pinned   in/out : median %8.1f us, best %8.1f us (5 MB in, %d MB out)"
          % (n, med, best, h.nbytes >> 20))
    eng.r1cs_free(r)
print("host allocations made by the context over the whole run: %d" % eng.host_allocations())
