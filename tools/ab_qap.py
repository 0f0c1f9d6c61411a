#!/usr/bin/env python3
"""Interleaved A/B timing of build variants of the QAP witness map in ONE process.

    python tools/ab_qap.py NAME=FLAGS [NAME=FLAGS ...] [--logn 10] [--batch 32] [--rounds 5]
    e.g. python tools/ab_qap.py plain= nomul=-DFRW_QAP_NO_MUL copy=-DFRW_QAP_NO_STAGES nocarry=-DFRW_QAP_AB_NO_NORMALISE

Each variant is compiled to its own shared object under gpurun_out/variants/, loaded with ctypes next to the others;
frw_qap_witness_map_dev and frw_r1cs_eval_dev (the sparse products alone) are timed with HIP events, round-robin."""
import argparse
import ctypes as C
import os
import statistics
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import falcon_r1cs_amd as frw  # noqa: E402
from falcon_r1cs_amd._lib import QapInfoStruct  # noqa: E402

CSRC = os.path.join(ROOT, "falcon-r1cs_amd", "csrc")
SRC = ("frw_kernels.hip", "frw_prepare.hip", "frw_r1cs_check.hip", "frw_qap.hip", "frw_setup.hip", "frw_msm.hip", "frw_capi.cpp", "frw_synth.cpp", "frw_r1cs.cpp", "frw_verify.cpp")


def build(name, flags, logn):
    out_dir = os.path.join(ROOT, "gpurun_out", "variants")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libfrw_qap_%s.so" % name)
    prebuilt = os.path.join(ROOT, "tools", "variants", "libfrw_qap_%s.so" % name)
    if os.path.exists(prebuilt):
        # built where the sources are (tools/build_qap_variant.sh NAME FLAGS: only frw_qap.hip is compiled again, the other objects are
        # the library's own) and sent along: no minutes of the GPU box go into compiling
        so = prebuilt
    else:
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-w", "-o", so] +
                              flags.split() + [os.path.join(CSRC, f) for f in SRC])
    lib = C.CDLL(so)
    lib.frw_r1cs_load.argtypes = [C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]
    lib.frw_qap_info.argtypes = [C.c_void_p, C.POINTER(QapInfoStruct)]
    lib.frw_qap_witness_map_dev.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 5 + [C.c_size_t, C.c_void_p]
    lib.frw_r1cs_eval_dev.argtypes = [C.c_void_p, C.c_size_t] + [C.c_void_p] * 5
    r = C.c_void_p()
    assert lib.frw_r1cs_load(0, 0, logn, C.byref(r)) == 0
    return lib, r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--logn", type=int, default=10)
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    eng = frw.WitnessEngine(0)
    L = frw.layout(a.logn)
    sig, pk, hm = frw.synth_triples(a.logn, a.batch, seed=99)
    d = [torch.from_numpy(x.view(np.int16)).to(dev) for x in (sig, pk, hm)]
    wit = torch.empty((a.batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((a.batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(a.batch, dtype=torch.int32, device=dev)
    eng.witness_ntt_verify_dev(a.logn, a.batch, d[0], d[1], d[2], wit, inst, st, 1, 0)
    torch.cuda.synchronize()
    libs = []
    for v in a.variants:
        name, _, flags = v.partition("=")
        libs.append((name, flags) + build(name, flags, a.logn))
    q = QapInfoStruct()
    assert libs[0][2].frw_qap_info(libs[0][3], C.byref(q)) == 0
    n = int(q.domain_size)
    per = max(int(q.workspace_bytes_per_signature), 3 * L.num_constraints * 32)
    ws = torch.empty(a.batch * per, dtype=torch.uint8, device=dev)
    h = torch.empty((a.batch, n, 4), dtype=torch.int64, device=dev)
    bad = torch.empty(a.batch, dtype=torch.int32, device=dev)
    P = lambda t: C.c_void_p(t.data_ptr())
    times = {name: ([], []) for name, *_ in libs}
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    for rnd in range(a.rounds + 1):
        for name, flags, lib, r in libs:
            e0.record()
            assert lib.frw_qap_witness_map_dev(r, a.batch, P(wit), P(inst), P(h), P(bad), P(ws), a.batch * per, None) == 0
            e1.record()
            assert lib.frw_r1cs_eval_dev(r, a.batch, P(wit), P(inst), P(bad), P(ws), None) == 0
            e2.record()
            torch.cuda.synchronize()
            if rnd:
                times[name][0].append(e0.elapsed_time(e1))
                times[name][1].append(e1.elapsed_time(e2))
    print("# Falcon-%d, %d signatures per call, domain 2^%d" % (1 << a.logn, a.batch, int(q.log_domain_size)))
    for name, flags, *_ in libs:
        full, mv = statistics.median(times[name][0]), statistics.median(times[name][1])
        print("%-10s witness map %8.3f ms  (%7.1f us/signature)  sparse products %8.3f ms  transforms %7.1f us/signature   [%s]"
              % (name, full, 1e3 * full / a.batch, mv, 1e3 * (full - mv) / a.batch, flags))


if __name__ == "__main__":
    main()
