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
    print("falcon-%d host path: %d signatures in %.3f s = %.0f signatures/s, %.1f GB/s of D2H (includes numpy output "
          "allocation by the caller: no; pageable memcpy: yes)" % (1 << logn, batch, dt, batch / dt, gb / dt))
