#!/usr/bin/env python3
"""Interleaved A/B timing of kernel build variants in ONE process (methodology rule 24 of the CDNA guide).

    python tools/ab_variants.py NAME=FLAGS [NAME=FLAGS ...] [--logn 10] [--chunk 4096] [--rounds 6]
    e.g. python tools/ab_variants.py plain= nt=-DFRW_STORE_AUX=2 nostore=-DFRW_NO_STORE g512=-DFRW_FORCE_GRID=512

Each variant is compiled to its own shared object under gpurun_out/variants/, loaded with ctypes next to the
others, and the full verify-with-ntt launch is timed with HIP events, round-robin over the variants.
"""
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
import falcon_r1cs_amd as frw  # noqa: E402  (inputs + layout only; loads torch's HIP runtime first)

CSRC = os.path.join(ROOT, "falcon-r1cs_amd", "csrc")


def build(name, flags):
    out_dir = os.path.join(ROOT, "gpurun_out", "variants")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libfrw_%s.so" % name)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", so] + \
        flags.split() + [os.path.join(CSRC, f) for f in ("frw_kernels.hip", "frw_prepare.hip", "frw_r1cs_check.hip", "frw_qap.hip", "frw_capi.cpp", "frw_synth.cpp", "frw_r1cs.cpp")]
    subprocess.check_call(cmd)
    lib = C.CDLL(so)
    lib.frw_ctx_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
    lib.frw_witness_ntt_verify_dev.argtypes = [C.c_void_p, C.c_int, C.c_size_t] + [C.c_void_p] * 3 + [C.c_int] + \
        [C.c_void_p] * 4
    lib.frw_diag_write_stream_dev.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t, C.c_void_p]
    lib.frw_ntt_modq_dev.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int] + [C.c_void_p] * 4
    lib.frw_witness_ntt_verify_compact_dev.argtypes = [C.c_void_p, C.c_int, C.c_size_t] + [C.c_void_p] * 6
    lib.frw_expand_dev.argtypes = [C.c_void_p, C.c_int, C.c_size_t] + [C.c_void_p] * 4
    ctx = C.c_void_p()
    assert lib.frw_ctx_create(0, C.byref(ctx)) == 0
    return lib, ctx


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("variants", nargs="+")
    ap.add_argument("--logn", type=int, default=10)
    ap.add_argument("--chunk", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=6)
    ap.add_argument("--workload", default="verify", choices=["verify", "ntt_modq", "compact", "expand"])
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    torch.cuda.set_device(0)
    L = frw.layout(a.logn)
    sig, pk, hm = frw.synth_triples(a.logn, a.chunk, seed=99)
    d = [torch.from_numpy(x.view(np.int16)).to(dev) for x in (sig, pk, hm)]
    wit = torch.empty((a.chunk, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((a.chunk, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.zeros(a.chunk, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    libs = []
    for v in a.variants:
        name, _, flags = v.partition("=")
        libs.append((name, flags) + build(name, flags))
    bytes_per = a.chunk * (32 * (L.num_witness + 2 * L.n) + 6 * L.n)
    if a.workload == "ntt_modq":
        bytes_per = a.chunk * (32 * 29 * L.n + 2 * L.n)
    times = {name: [] for name, *_ in libs}
    reps = 50 if a.workload == "ntt_modq" else 4
    comp = None
    if a.workload in ("compact", "expand"):
        CL = frw.compact_layout(a.logn)
        comp = torch.empty((a.chunk, CL.bytes_per_signature), dtype=torch.uint8, device=dev)
        if a.workload == "compact":
            bytes_per = a.chunk * CL.bytes_per_signature

    def launch(lib, ctx):
        if a.workload == "compact" or (a.workload == "expand" and not launch.primed):
            rc = lib.frw_witness_ntt_verify_compact_dev(ctx, a.logn, a.chunk, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(),
                                                        comp.data_ptr(), st.data_ptr(), stream.cuda_stream)
            launch.primed = True
            if a.workload == "expand":
                assert rc == 0
                rc = lib.frw_expand_dev(ctx, a.logn, a.chunk, comp.data_ptr(), wit.data_ptr(), inst.data_ptr(), stream.cuda_stream)
        elif a.workload == "expand":
            rc = lib.frw_expand_dev(ctx, a.logn, a.chunk, comp.data_ptr(), wit.data_ptr(), inst.data_ptr(), stream.cuda_stream)
        elif a.workload == "ntt_modq":      # d[1] doubles as the input polynomial, d[2] as the reduced-NTT output
            rc = lib.frw_ntt_modq_dev(ctx, a.logn, a.chunk, d[1].data_ptr(), 1, wit.data_ptr(), d[2].data_ptr(), st.data_ptr(),
                                      stream.cuda_stream)
        else:
            rc = lib.frw_witness_ntt_verify_dev(ctx, a.logn, a.chunk, d[0].data_ptr(), d[1].data_ptr(), d[2].data_ptr(), 1,
                                                wit.data_ptr(), inst.data_ptr(), st.data_ptr(), stream.cuda_stream)
        assert rc == 0
    launch.primed = False
    for name, _, lib, ctx in libs:
        launch(lib, ctx)
    torch.cuda.synchronize()
    for _ in range(a.rounds):
        for name, _, lib, ctx in libs:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            for _ in range(reps):
                launch(lib, ctx)
            e1.record(stream)
            e1.synchronize()
            times[name].append(e0.elapsed_time(e1) / reps)
    # compute-free write stream over the same buffer, interleaved the same way (first variant's library)
    lib0, ctx0 = libs[0][2], libs[0][3]
    wbytes = wit.numel() * 8
    ceil_t = []
    for _ in range(a.rounds):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(4):
            assert lib0.frw_diag_write_stream_dev(ctx0, wit.data_ptr(), wbytes, L.num_witness * 32, stream.cuda_stream) == 0
        e1.record(stream)
        e1.synchronize()
        ceil_t.append(e0.elapsed_time(e1) / 4)
        for name, _, lib, ctx in libs:           # keep the device in the same thermal/clock state
            launch(lib, ctx)
    print("%-12s median %.4f ms  min %.4f ms  -> %.1f GB/s (median)   [compute-free write stream, same buffer]"
          % ("write-only", statistics.median(ceil_t), min(ceil_t), wbytes / statistics.median(ceil_t) / 1e6))
    for name, flags, _, _ in libs:
        t = times[name]
        print("%-12s median %.4f ms  min %.4f ms  -> %.1f GB/s (median)   [%s]"
              % (name, statistics.median(t), min(t), bytes_per / statistics.median(t) / 1e6, flags))


if __name__ == "__main__":
    main()
