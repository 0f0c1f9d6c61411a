#!/usr/bin/env python3
"""bench.py -- Falcon-1024 verify-with-ntt R1CS witnesses per second on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: either under a launcher -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...
     -- or plainly: with no WORLD_SIZE in the environment this file starts its own N ranks as child processes, see
     launch_ranks)

A step = one pass of the hot path (falcon-r1cs/src/circuits/falcon_ntt.rs:26-123 of the reference) over one batch
of synthetic signatures per GPU: BASELINE.json configs[2], "Falcon-1024 batch=65536 sigs, full verify-with-ntt
witness" -- the SAME 65,536 signatures per GPU per step for every N, so that N = 1, 2, 4, 8 are one weak-scaling series
(N = 8: two steps are BASELINE configs[3]'s 1 M signatures over the node).  65,536 witnesses are 329 GB, more than
one GPU's 288 GB, so a step streams the batch through one reused HBM witness buffer in launches of `--chunk`
signatures; inputs are resident in HBM before the timed region and outputs stay in HBM (the boundary a GPU prover or a
peer would consume them from).  Multi-GPU: signatures shard by index, every rank processes its own batch (weak scaling),
no data-path collective inside `value`; the north-star's all-gather of the witness vectors is measured as a second,
separately reported curve (compact encoding over RCCL, expanded on every receiver).

Prints ONE JSON line (rank 0).  What is checked is what was timed: after the timed region the witnesses the LAST timed
launch left in HBM are digested, checked against the independently emitted constraint system on the device
(falcon_ntt.rs:159 `assert!(cs.is_satisfied())`), and a strided sample of them is compared with the oracle.
`roofline` is measured live with HIP events on the launch stream; `cpu_baseline` times the oracle (oracle/frw_oracle.c, a
restatement -- the Rust reference cannot run here) on this host's cores over a bounded sample of the same inputs.
"""
import argparse
import hashlib
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor


def _requested_gpus(argv):
    """--gpus N / --gpus=N from a raw argument list (1 when absent or malformed: argparse reports that later)."""
    n = 1
    for i, a in enumerate(argv):
        try:
            if a == "--gpus" and i + 1 < len(argv):
                n = int(argv[i + 1])
            elif a.startswith("--gpus="):
                n = int(a.split("=", 1)[1])
        except ValueError:
            return 1
    return n


def launch_ranks(argv, n, grace_s=30.0):
    """`python bench.py --gpus N ...` started plainly (no WORLD_SIZE in the environment): start the N ranks as FRESH CHILD
    PROCESSES of this one -- which has imported neither torch nor the HIP library and never will -- with the environment
    torch.distributed's env:// rendezvous reads (RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free
    MASTER_PORT), let them write to this process's stdout / stderr (rank 0 prints the one JSON line), and return the
    largest exit code.  Nothing is exec'ed.  When a rank fails the others get `grace_s` seconds to leave by themselves
    (their own deadlines end a collective whose peer is gone) and are then terminated by PID.
    The ranks do not outlive this process: SIGTERM / SIGINT arriving here are passed on to them, whatever ends the wait (an
    exception, a signal) terminates those still running, and every rank asks the kernel for SIGTERM on its parent's death
    (PR_SET_PDEATHSIG) -- a harness that times this process out does not leave GPUs and a rendezvous port held."""
    import ctypes
    import signal
    import socket
    import subprocess
    if any(a == "--plan" for a in argv):
        return None
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["MASTER_ADDR"] = "127.0.0.1"
    if "MASTER_PORT" not in env:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            env["MASTER_PORT"] = str(s.getsockname()[1])
    env["WORLD_SIZE"] = env["LOCAL_WORLD_SIZE"] = str(n)
    # the host's cores are shared by the ranks (what torch.distributed.run's OMP_NUM_THREADS=1 default is about; the
    # synthetic-input threads are sized by bench.py itself)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n)))
    def die_with_parent():
        # in the child, before its program starts: SIGTERM when the launcher goes (prctl(PR_SET_PDEATHSIG = 1, SIGTERM))
        try:
            ctypes.CDLL(None).prctl(1, int(signal.SIGTERM), 0, 0, 0)
        except Exception:                        # noqa: BLE001 -- not Linux: the finally below still covers the orderly ways out
            pass
    procs = []

    def pass_on(signum, _frame):
        for p in procs:
            if p.poll() is None:
                p.send_signal(signum)
    previous = {sg: signal.signal(sg, pass_on) for sg in (signal.SIGTERM, signal.SIGINT)}
    codes = [None] * n
    first_failure = None
    try:
        for r in range(n):
            e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=e, preexec_fn=die_with_parent))
        codes = _wait_for_ranks(procs, codes, first_failure, grace_s)
    finally:
        for sg, h in previous.items():
            signal.signal(sg, h)
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(10)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    # a rank killed by a signal reports -SIG: any non-zero code is a failed run
    return max(abs(c) for c in codes)


def _wait_for_ranks(procs, codes, first_failure, grace_s):
    import subprocess
    while any(c is None for c in codes):
        for r, p in enumerate(procs):
            if codes[r] is None:
                codes[r] = p.poll()
                if codes[r] not in (None, 0):
                    first_failure = first_failure or time.monotonic()
                    sys.stderr.write("bench.py launcher: rank %d exited with code %d\n" % (r, codes[r]))
                    sys.stderr.flush()
        if first_failure is not None and time.monotonic() - first_failure > grace_s:
            for r, p in enumerate(procs):
                if codes[r] is None:
                    sys.stderr.write("bench.py launcher: terminating rank %d (pid %d) after a peer failed\n" % (r, p.pid))
                    p.terminate()
                    try:
                        codes[r] = p.wait(10)
                    except subprocess.TimeoutExpired:
                        p.kill()
                        codes[r] = p.wait()
            break
        time.sleep(0.05)
    return codes


if __name__ == "__main__" and "WORLD_SIZE" not in os.environ and _requested_gpus(sys.argv[1:]) > 1:
    # before `import torch` and before libfrw.so is mapped: this process must never touch a GPU
    _rc = launch_ranks(sys.argv[1:], _requested_gpus(sys.argv[1:]))
    if _rc is not None:
        sys.exit(_rc)

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import falcon_r1cs_amd as frw  # noqa: E402
from falcon_r1cs_amd import sharding  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
SEED = 0x46414C434F4E31        # recorded in the output
KERNEL_SRC = os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_kernels.hip")


def kernel_source_sha():
    return hashlib.sha256(open(KERNEL_SRC, "rb").read()).hexdigest()[:16]


def measured_traffic(logn, chunk, kernel_prefix="witness_ntt_verify_kernel"):
    """HBM bytes per launch from a committed rocprofv3 PMC summary (profiles/*_hbm_traffic.json, written by
    tools/summarize_profiles.py from separate --pmc WRITE_SIZE / FETCH_SIZE passes of this same command).  Only a
    summary taken from the kernel source that is being run counts: each records the sha-256 of frw_kernels.hip it was
    profiled with, and a mismatch (the kernel changed since) yields None rather than a stale figure."""
    import glob
    sha = kernel_source_sha()
    best = None
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "*_hbm_traffic.json"))):
        try:
            j = json.load(open(path))
        except (OSError, ValueError):
            continue
        if (j.get("logn") == logn and j.get("signatures_per_launch") == chunk and j.get("kernel_source_sha256_16") == sha
                and str(j.get("kernel", "")).startswith(kernel_prefix)):
            best = (j["hbm_bytes_per_launch"], os.path.basename(path))
    return best


def synth(logn, count, first_index, threads):
    """count synthetic triples starting at global index first_index (bit-identical on every host/rank)."""
    n = 1 << logn
    out = [np.empty((count, n), dtype=np.uint16) for _ in range(3)]
    step = max(1, (count + threads - 1) // threads)

    def work(lo):
        s, p, h = frw.synth_triples(logn, min(step, count - lo), SEED, first_index + lo)
        out[0][lo:lo + len(s)], out[1][lo:lo + len(s)], out[2][lo:lo + len(s)] = s, p, h

    with ThreadPoolExecutor(threads) as ex:
        list(ex.map(work, range(0, count, step)))
    return out


def load_oracle():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import frw_testlib                      # the ONLY place bench.py touches oracle/: as the timed CPU baseline + checker
    return frw_testlib.load_oracle()


def cpu_baseline(logn, sig, pk, hm, sample_idx, gpu_digest_of, budget_s=9.0):
    """Oracle ("port") on the host cores over a bounded sample of the signatures of the last timed launch; also the
    parity check of that launch: the first >= 256 sampled witnesses are compared with the GPU's by digest.
    Output buffers are allocated and touched once, outside the timed calls."""
    oracle = load_oracle()
    L = frw.layout(logn)
    nproc = os.cpu_count() or 1
    sub = 256                               # 1.3 GB of host witness at a time
    out = (np.zeros((sub, L.num_witness, 4), dtype=np.uint64), np.zeros((sub, L.num_instance, 4), dtype=np.uint64),
           np.zeros(sub, dtype=np.int32))
    for a in out:
        a.fill(1)                            # first touch happens here
    # single thread: the analogue of the reference's one-threaded generate_constraints (Cargo.toml:32 `parallel = []`)
    n1 = min(len(sample_idx), 2048)
    t1 = 0.0
    for lo in range(0, n1, sub):
        idx = sample_idx[lo:lo + sub]
        t0 = time.perf_counter()
        oracle.witness_ntt_verify(logn, sig[idx], pk[idx], hm[idx], 1, threads=1, out=out)
        t1 += time.perf_counter() - t0
    # all cores, time-bounded; digest parity with the GPU on the same signatures
    checked, t_all, nall = 0, 0.0, 0
    for lo in range(0, len(sample_idx), sub):
        idx = sample_idx[lo:lo + sub]
        t0 = time.perf_counter()
        wit, _, st = oracle.witness_ntt_verify(logn, sig[idx], pk[idx], hm[idx], 1, threads=nproc, out=out)
        t_all += time.perf_counter() - t0
        nall += len(idx)
        if checked < 256:                   # digest a few hundred on the host (python loop over ctypes calls)
            for j, i in enumerate(idx):
                assert st[j] == 0
                assert oracle.digest(wit[j]) == gpu_digest_of(i), "GPU witness %d differs from the oracle" % i
                checked += 1
        if t_all > budget_s:
            break
    return {"value": round(nall / t_all, 1), "unit": "signatures/s", "cores": nproc, "kind": "port",
            "sample": "%d Falcon-%d signatures, a strided sample of the buffer the timed launches left, %d threads on %d host cores "
                      "(oracle/frw_oracle.c, a C restatement; the Rust reference cannot be built here); output buffers "
                      "pre-touched and reused" % (nall, 1 << logn, nproc, nproc),
            "nproc": nproc, "threads": nproc, "single_thread": round(n1 / t1, 1), "single_thread_sample": n1,
            "gpu_witnesses_checked_by_digest": checked, "checked_witnesses_are_from": "the buffer the timed launches left"}


def time_ntt_modq(eng, dev, logn, batch, launches, warm, three_streams=True):
    """NTTPolyVar::ntt_circuit alone (poly.rs:104-159): `launches` back-to-back launches between two HIP events.
    three_streams=False (profile runs: --no-aggregate) leaves the overlapped leg out, so that rocprofv3's per-kernel average
    is over the one-stream launches `avg_launch_ms` is quoted for."""
    n = 1 << logn
    rng = np.random.default_rng(SEED & 0xFFFFFFFF)
    poly = torch.from_numpy(rng.integers(0, 12289, size=(batch, n), dtype=np.uint16).view(np.int16)).to(dev)
    wit = torch.empty((batch, 29 * n, 4), dtype=torch.int64, device=dev)
    out = torch.empty((batch, n), dtype=torch.int16, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    run = lambda: eng.ntt_modq_dev(logn, batch, poly, wit, out, st, frw.ENC_MONTGOMERY, stream.cuda_stream)
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(launches):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / launches
    assert int((st != 0).sum().item()) == 0
    bytes_per = 32 * 29 * n + 2 * n                               # 476,160 / 952,320 B per polynomial
    achieved = batch * bytes_per / (ms * 1e-3) / 1e9
    # The same launches dealt over three streams (independent batches, each with its own output buffer): a launch this short
    # -- 0.3 ms -- spends a tenth of its time filling and draining the chip, and on one stream the next launch waits for the
    # last workgroup of this one; on three the ramp of launch k + 1 runs under the tail of launch k
    # (tools/ab_short_launches.py, profiles/r03_short_launches.txt).  Reported beside `frac`, which stays the per-launch figure
    # rocprofv3's per-kernel average can be compared with.
    res = {"kernel": "ntt_modq_kernel<%d,1>" % logn, "workload": "falcon-%d NTT + mod_q witness kernel, batch=%d" % (n, batch),
           "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
           "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4), "launches_timed": launches,
           "algorithmic_bytes_per_launch": batch * bytes_per, "polynomials_per_s": round(batch * launches / wall, 1)}
    if not three_streams:
        return res
    streams = [torch.cuda.Stream() for _ in range(3)]
    bufs = [(torch.empty_like(wit), torch.empty_like(out), torch.empty_like(st)) for _ in streams]
    runs = [(lambda s=s_, b=b_: eng.ntt_modq_dev(logn, batch, poly, b[0], b[1], b[2], frw.ENC_MONTGOMERY, s.cuda_stream))
            for s_, b_ in zip(streams, bufs)]
    for r_ in runs:
        for _ in range(warm):
            r_()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(launches):
        runs[i % 3]()
    torch.cuda.synchronize()
    ms3 = (time.perf_counter() - t0) / launches * 1e3
    assert all(int((b[2] != 0).sum().item()) == 0 for b in bufs) and all(torch.equal(b[1], out) for b in bufs)
    achieved3 = batch * bytes_per / (ms3 * 1e-3) / 1e9
    res["three_streams"] = {"ms_per_launch": round(ms3, 4), "achieved": round(achieved3, 1), "frac": round(achieved3 / HBM_PEAK_GBS, 4),
                            "what": "the same %d launches dealt over three streams (wall clock over all of them)" % launches}
    return res


def time_verify(eng, dev, logn, batch, launches, warm, threads):
    """Full verify-with-ntt witness at another parameter set / launch size: `launches` launches between two HIP events."""
    L = frw.layout(logn)
    sig, pk, hm = synth(logn, batch, 1 << 40, threads)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    run = lambda: eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, frw.ENC_MONTGOMERY, stream.cuda_stream)
    for _ in range(warm):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(launches):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / launches
    assert int((st != 0).sum().item()) == 0
    bytes_per = 32 * (L.num_witness + 2 * L.n) + 3 * 2 * L.n
    achieved = batch * bytes_per / (ms * 1e-3) / 1e9
    return {"kernel": "witness_ntt_verify_kernel<%d,1>" % logn,
            "workload": "falcon-%d full verify-with-ntt witness, %d signatures per launch" % (L.n, batch),
            "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4), "avg_launch_ms": round(ms, 4), "launches_timed": launches,
            "algorithmic_bytes_per_launch": batch * bytes_per, "signatures_per_s": round(batch / (ms * 1e-3), 1),
            "launch_shape": eng.launch_shape(logn, batch)}


def time_compact(eng, dev, logn, batch, launches, warm, d_in, d_wit):
    """FRW_ENC_COMPACT producer and frw_expand_dev on a resident batch (reusing the benchmark's inputs and, for the
    expansion, its witness buffer): the two halves of the multi-GPU gather path, timed on one GPU."""
    L, CL = frw.layout(logn), frw.compact_layout(logn)
    d_sig, d_pk, d_hm = (t[:batch] for t in d_in)
    comp = torch.empty((batch, CL.bytes_per_signature), dtype=torch.uint8, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    out = {}
    for name, run, nbytes in (
            ("generate_compact", lambda: eng.witness_ntt_verify_compact_dev(logn, batch, d_sig, d_pk, d_hm, comp, st, stream.cuda_stream),
             batch * CL.bytes_per_signature),
            ("expand", lambda: eng.expand_dev(logn, batch, comp, d_wit[:batch], inst, stream.cuda_stream),
             batch * 32 * (L.num_witness + L.num_instance))):
        for _ in range(warm):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(launches):
            run()
        e1.record(stream)
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / launches
        out[name] = {"avg_launch_ms": round(ms, 4), "signatures_per_s": round(batch / (ms * 1e-3), 1),
                     "bytes_written_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1), "signatures_per_launch": batch}
    assert int((st != 0).sum().item()) == 0
    out["bytes_per_signature"] = {"compact": CL.bytes_per_signature, "arkworks": 32 * (L.num_witness + L.num_instance)}
    return out


def qap_cpu_port(logn, triple, gpu_h0):
    """The oracle's witness map (oracle/qap_oracle.c, one thread) for one signature: the CPU figure beside the GPU's, and
    the check of the GPU's h for that signature, element for element.  Matrices: the file frw_r1cs_export writes."""
    import tempfile
    oracle = load_oracle()
    sig, pk, hm = triple
    wit, inst, st = oracle.witness_ntt_verify(logn, sig, pk, hm, 0)                  # canonical limbs
    assert st[0] == 0
    z = np.concatenate([inst[0], wit[0]])
    with tempfile.TemporaryDirectory() as tmp:
        path = os.path.join(tmp, "c.r1cs")
        assert frw.load_library().frw_r1cs_export(0, logn, path.encode(), None) == 0
        raw = open(path, "rb").read()
    ni, nw, nc, *nnz = np.frombuffer(raw, dtype=np.uint64, count=6, offset=8).tolist()
    off, mats = 56, []
    for k in range(3):
        ptr = np.frombuffer(raw, dtype=np.uint64, count=nc + 1, offset=off); off += 8 * (nc + 1)
        col = np.frombuffer(raw, dtype=np.uint32, count=nnz[k], offset=off); off += 4 * nnz[k]
        val = np.frombuffer(raw, dtype=np.uint64, count=4 * nnz[k], offset=off).reshape(-1, 4); off += 32 * nnz[k]
        mats.append((ptr, col, val))
    t0 = time.perf_counter()
    prods = [oracle.qap_matvec(*m, z) for m in mats]
    h = oracle.qap_witness_map(*prods, ni, z)
    seconds = time.perf_counter() - t0
    p_fr = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    r_inv = pow(1 << 256, -1, p_fr)
    got = [int.from_bytes(row.tobytes(), "little") * r_inv % p_fr for row in gpu_h0]
    want = [int.from_bytes(row.tobytes(), "little") for row in h]
    assert got == want, "GPU witness map differs from the oracle"
    return {"seconds_per_map": round(seconds, 3), "threads": 1, "kind": "port",
            "what": "oracle/qap_oracle.c (plain radix-2 transforms, 64-bit-limb Montgomery arithmetic); the GPU's h of the same "
                    "signature compared with it element for element (%d coefficients)" % len(want)}


def qap_products_per_map(log_n):
    """Field products one witness map executes in its transform passes for a witness that satisfies the system (frw_qap.hip,
    counted from the pass schedule): a six-stage pass does 17 twiddle products per 8 elements (a five-stage one 13) and, unless
    it stores plainly, 8 products by the per-index factor.  The map runs SIX transforms for such a witness (two inverse and two
    forward ones to reach the coset psi H with a and b, one inverse there with a b formed at its load, one inverse on the domain
    with (A z)(B z) formed at its load and a constant at its store) -- and the seven of ark-groth16's own schedule only for a
    witness that violates the system, which the benchmark's never do (`qap_products_per_map_seven` is that count)."""
    n = 1 << log_n
    t3 = 17 if log_n - 12 == 6 else 13
    with_factor = lambda tw: (tw + 8) / 8.0
    ifft = with_factor(17) + with_factor(17) + with_factor(t3)
    fft = with_factor(t3) + with_factor(17) + 17 / 8.0                 # its last pass stores without a factor
    return int(n * (2 * ifft + 2 * fft + 2 * (ifft + 1)))


def qap_pass_stages(log_n):
    """frw_device.h qap_pass_schedule: the stages of the passes of a 2^log_n-point transform, lowest bits first."""
    k = (log_n + 5) // 6
    t = [6] * k
    deficit = 6 * k - log_n
    for _ in range(2):
        for i in range(k - 1, 0, -1):
            if deficit:
                t[i] -= 1
                deficit -= 1
    assert deficit == 0
    return t


def qap_products_per_map_any(log_n):
    """qap_products_per_map for any domain (aggregate statements: four or five passes of six, five or four stages): a pass of T stages
    does 5 twiddle products per 8 elements in its three low stages and 4 per further stage (17 / 13 / 9 for T = 6 / 5 / 4), plus 8 by
    the per-index factor unless it stores plainly (the last pass of a forward transform)."""
    n = 1 << log_n
    tw = {6: 17, 5: 13, 4: 9}
    t = qap_pass_stages(log_n)
    ifft = sum((tw[x] + 8) / 8.0 for x in t)
    fft = sum((tw[x] + 8) / 8.0 for x in t[1:]) + tw[t[0]] / 8.0
    return int(n * (2 * ifft + 2 * fft + 2 * (ifft + 1)))


def qap_products_per_map_seven(log_n):
    """ark-groth16's schedule as written: seven transforms (3 + 3 + 1 arrays), plus a(X) b(X) - c(X) once per index."""
    n = 1 << log_n
    t3 = 17 if log_n - 12 == 6 else 13
    with_factor = lambda tw: (tw + 8) / 8.0
    ifft = with_factor(17) + with_factor(17) + with_factor(t3)
    fft = with_factor(t3) + with_factor(17) + 17 / 8.0
    return int(n * (3 * ifft + 3 * fft + (1 + ifft)))


# instructions of one f29_mul as hipcc compiles it for gfx950 (tests/test_isa_hazards.py re-derives them from the assembly)
F29_MUL_MAD64, F29_MUL_OTHER = 153, 77


def qap_roofline(eng, log_n, maps_per_s, kernel):
    """VALU-issue roofline of the transform passes: peak = what the chip could do if it issued nothing but the instructions
    of the field products, priced with the issue rates measured in this process (frw_diag_valu_rates)."""
    r = eng.valu_rates()
    per_wave_product_us = F29_MUL_MAD64 / r["v_mad_u64_u32"] + F29_MUL_OTHER / r["v_add_u32"]
    peak = r["simds"] * 64 / per_wave_product_us / 1e3                 # G products/s
    achieved = qap_products_per_map(log_n) * maps_per_s / 1e9
    return {"bound": "valu_issue", "unit": "G field products/s", "achieved": round(achieved, 1), "peak": round(peak, 1),
            "frac": round(achieved / peak, 4), "kernel": kernel,
            "peak_is": "%d SIMDs x 64 lanes / (%d v_mad_u64_u32 at %.1f + %d other at %.1f wave-instructions/SIMD/us), rates "
                       "measured in this process" % (r["simds"], F29_MUL_MAD64, r["v_mad_u64_u32"], F29_MUL_OTHER, r["v_add_u32"]),
            "products_per_map": qap_products_per_map(log_n),
            "multiplier_loop_G_products_per_s": round(r["f29_mul_products_per_s"] / 1e9, 1),
            "frac_of_multiplier_loop": round(achieved / (r["f29_mul_products_per_s"] / 1e9), 4),
            "note": "achieved counts the transform passes' products over the WHOLE map's time (sparse products included, ~11 % "
                    "of it); the multiplier loop is f29_mul alone at four waves per SIMD, no loads, no butterflies: what the "
                    "same instruction stream sustains when nothing else is issued"}


def time_qap(eng, handle, dev, d_wit, d_inst, nsig, reps, logn=None, cpu_triple=None):
    """The step after the hot path in a Groth16 prover: ark-groth16's R1CStoQAP::witness_map -- A z, B z,
    C z, three inverse + three coset-forward + one coset-inverse transform over the 2^17 / 2^18 domain -- for `nsig` of the
    witnesses the timed launches left in HBM.  Bound by VALU issue (field multiplications), not by HBM."""
    q = eng.qap_info(handle)
    n, per = int(q.domain_size), int(q.workspace_bytes_per_signature)
    ws = torch.empty(nsig * per, dtype=torch.uint8, device=dev)
    h = torch.empty((nsig, n, 4), dtype=torch.int64, device=dev)
    bad = torch.empty(nsig, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    run = lambda: eng.qap_witness_map_dev(handle, nsig, d_wit, d_inst, h, ws, nsig * per, bad, stream.cuda_stream)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    assert int(bad.abs().sum()) == 0, "witness map: unsatisfied rows"
    assert int(h[:, -1].abs().sum()) == 0, "witness map: deg h must be <= n - 2 for satisfied systems"
    # the six-transform route to the same h (valid because every witness here satisfies the system): must be identical
    h6 = torch.empty_like(h)
    run6 = lambda: eng.qap_quotient_dev(handle, nsig, d_wit, d_inst, h6, ws, nsig * per, bad, stream.cuda_stream)
    run6()
    torch.cuda.synchronize()
    e0.record(stream)
    for _ in range(reps):
        run6()
    e1.record(stream)
    torch.cuda.synchronize()
    ms6 = e0.elapsed_time(e1) / reps
    assert torch.equal(h6, h), "six-transform quotient differs from the witness map on satisfied witnesses"
    products = qap_products_per_map(int(q.log_domain_size))
    out = {"workload": "R1CS->QAP witness map (h = (A B - C) / Z, domain 2^%d) of %d resident witnesses per call" %
                       (int(q.log_domain_size), nsig),
           "ms_per_call": round(ms, 3), "signatures_per_s": round(nsig / (ms * 1e-3), 1), "calls_timed": reps,
           "field_products_per_signature_transforms": products,
           "bytes_out_per_signature": n * 32, "checked": "no unsatisfied rows; top coefficient of every h is zero",
           "roofline": qap_roofline(eng, int(q.log_domain_size), nsig / (ms * 1e-3), "ntt_pass_kernel<...> (six transforms = 11 launches per map of satisfied witnesses, + 8 that find an empty list: ark-groth16's seven run only for a witness that violates the system)"),
           "six_transform_quotient": {"ms_per_call": round(ms6, 3), "signatures_per_s": round(nsig / (ms6 * 1e-3), 1),
                                      "what": "frw_qap_quotient_dev: h as the high half of a(X) b(X), six transforms, C z not "
                                              "transformed; bit-identical to the witness map on these (satisfied) witnesses"}}
    if cpu_triple is not None:
        out["cpu_port"] = qap_cpu_port(logn, cpu_triple, h[0].cpu().numpy().view(np.uint64))
        out["cpu_port"]["gpu_maps_per_cpu_core_map"] = round(out["signatures_per_s"] * out["cpu_port"]["seconds_per_map"], 1)
    del h6, ws
    return out, h


# instructions of one fq_mul (frw_fq29.h) as hipcc compiles it for gfx950: 392 v_mad_u64_u32 + 14 v_mul_lo_u32, 90 others;
# of one fq_sqr (105 + 14 + 196 multiplies) and of one fq_mul_sub (a b - c d: two products, one reduction)
# (tests/test_isa_hazards.py re-derives all three from the assembly)
FQ_MUL_MULTIPLY, FQ_MUL_OTHER = 406, 90
FQ_SQR_MULTIPLY, FQ_SQR_OTHER = 315, 101
FQ_PAIR_MULTIPLY, FQ_PAIR_OTHER = 602, 135
# the point formulas in those units (frw_fq29.h): mixed addition = 8 M + 2 S of which one M pair shares a reduction; full addition 12 M + 2 S
MSM_VALU_PER_MADD = 5803          # all vector instructions msm_bucket_kernel<FqField, true> issues per wavefront per mixed addition (PMC, round 4)
MADD_OPS = {"mul": 6, "sqr": 2, "pair": 1}
ADD_OPS = {"mul": 10, "sqr": 2, "pair": 1}


def point_op_instructions(ops):
    return (ops["mul"] * FQ_MUL_MULTIPLY + ops["sqr"] * FQ_SQR_MULTIPLY + ops["pair"] * FQ_PAIR_MULTIPLY,
            ops["mul"] * FQ_MUL_OTHER + ops["sqr"] * FQ_SQR_OTHER + ops["pair"] * FQ_PAIR_OTHER)
R_FR = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def groth16_h_query(eng, n, t=0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % R_FR, delta=987654321):
    """A proving key's h_query from toy toxic waste, on the device: h_query[i] = (zt / delta) t^i G1 (ark-groth16 0.3.0
    generator.rs).  Any key does for timing; a known one lets the result be checked without a second MSM."""
    c = (pow(t, n, R_FR) - 1) * pow(delta, -1, R_FR) % R_FR
    ks, x = [], c
    for _ in range(n - 1):
        ks.append(x)
        x = x * t % R_FR
    lim = np.frombuffer(b"".join(k.to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    return eng.g1_fixed_base(lim), (t, c)


def witness_side_additions(inst_row, wit_row, first_index=0):
    """Point additions the four witness-side sums of ONE proof perform in the narrow pipeline (frw_msm.hip: 8-bit signed
    windows, scalars that are one summed apart, zeros skipped), counted on the host from one signature's assignment
    z = instance ++ witness (Montgomery limbs): (scalars equal to one, non-zero digits of all the others).  The three
    blinding scalars appended to z add at most 3 x 32 digits and are left out.  Third value: the groups of eight consecutive points
    (index = first_index + position) with at least one such scalar -- what the ones cost through the byte-pattern tables."""
    r_inv = pow(1 << 256, -1, R_FR)
    ones = digits = 0
    groups = set()                                   # groups of eight consecutive points holding at least one scalar equal to one
    for idx, row in enumerate(np.concatenate([inst_row, wit_row]).reshape(-1, 4)):
        if not row.any():
            continue
        v = int.from_bytes(row.tobytes(), "little") * r_inv % R_FR
        if v == 1:
            ones += 1
            groups.add((idx + first_index) >> 3)
            continue
        carry = 0
        for j in range(32):
            d = ((v >> (8 * j)) & 0xFF) + carry
            carry = 0
            if j < 31 and d > 128:
                d -= 256
                carry = 1
            digits += d != 0
    return ones, digits, len(groups)


def groth16_roofline(eng, log_n, n, ni, nw, inst_row, wit_row, proofs_per_s):
    """VALU-issue roofline of a whole proof: the instructions of its field products -- the witness map's Fr products (as
    `qap_witness_map_*` counts them) and the Fq products inside the point additions of the five sums (h_query through the 16-bit
    pipeline as `groth16_msm_h_*` counts it; a_query, b_g1_query, l_query, b_g2_query through the 8-bit one, counted from one
    witness) -- each class of instruction at the issue rate measured in this process.  peak = proofs/s if the chip issued
    nothing else; frac = achieved / peak."""
    rates = eng.valu_rates()
    (mm, mo), (am, ao) = point_op_instructions(MADD_OPS), point_op_instructions(ADD_OPS)
    ones, digits, one_groups = witness_side_additions(inst_row, wit_row)
    ones_w, digits_w, one_groups_w = witness_side_additions(inst_row[:0], wit_row, first_index=ni)     # l_query: the witness part alone
    # mixed additions: every non-zero digit, and ONE per group of eight points that holds a scalar equal to one (the byte-pattern
    # tables of frw_msm.hip); full additions: the items of the 128 buckets, the suffix scan and the tree of the one-wavefront fold
    # (~2 per bucket + one per ones' partial sum), all small beside the former
    g1_madds = 2 * (one_groups + digits) + (one_groups_w + digits_w)
    g1_adds = 3 * (2 * 128 + 64)
    g2_madds, g2_adds = one_groups + digits, 2 * 128 + 64
    G2_FACTOR = 3                                                             # an Fq2 product = three Fq products (Karatsuba); frw_fq29.h
    h_madds, h_adds = 16 * (n - 1), 2 * 32768
    madds = h_madds + g1_madds + G2_FACTOR * g2_madds
    adds = h_adds + g1_adds + G2_FACTOR * g2_adds
    fq_products = madds * 10 + adds * 14
    fr_products = qap_products_per_map(log_n)
    wave_us = ((madds * mm + adds * am) / rates["v_mad_u64_u32"] + (madds * mo + adds * ao) / rates["v_add_u32"]
               + fr_products * (F29_MUL_MAD64 / rates["v_mad_u64_u32"] + F29_MUL_OTHER / rates["v_add_u32"]))
    peak_proofs = rates["simds"] * 64 / wave_us * 1e6
    total = fq_products + fr_products
    return {"bound": "valu_issue", "unit": "G field products/s (Fq products of the five sums + Fr products of the witness map)",
            "achieved": round(total * proofs_per_s / 1e9, 2), "peak": round(total * peak_proofs / 1e9, 2),
            "frac": round(proofs_per_s / peak_proofs, 4),
            "per_proof": {"fq_products": fq_products, "fr_products": fr_products,
                          "mixed_additions": {"h_query": h_madds, "a_query + b_g1_query + l_query": g1_madds, "b_g2_query (Fq2)": g2_madds},
                          "scalars_equal_to_one": ones, "groups_of_eight_points_holding_one": one_groups, "non_zero_8_bit_digits_of_the_others": digits,
                          "g2_priced_as_g1_times": G2_FACTOR},
            "peak_is": "%d SIMDs x 64 lanes issuing only the instructions of these products: an Fq product %d multiplies + %d others, "
                       "an Fr product %d + %d; multiplies at %.1f, the others at %.1f wave-instructions/SIMD/us (measured in this "
                       "process)" % (rates["simds"], FQ_MUL_MULTIPLY, FQ_MUL_OTHER, F29_MUL_MAD64, F29_MUL_OTHER,
                                     rates["v_mad_u64_u32"], rates["v_add_u32"]),
            "kernel": "frw_groth16_prove_dev: ntt_pass_kernel + r1cs_eval (witness map), msm_bucket_kernel (h_query), "
                      "nmsm_bucket_kernel / nmsm_ones_kernel x 4 (witness-side sums), folds, tails"}


def time_groth16(eng, handle, dev, d_wit, d_inst, nsig, reps, L, logn, before_timed=None, with_one_proof=True):
    """The whole of examples/pok_sig.rs:30-47, per signature: circuit_specific_setup -> frw_groth16_setup (toxic waste drawn
    here; the QAP at t on the host, the queries as fixed-base multiples on the device), create_random_proof ->
    frw_groth16_prove_dev = witness map + five multi-scalar multiplications (G1: h_query, a_query, b_g1_query, l_query; G2:
    b_g2_query) + blinding + assembly of (A, B, C) for `nsig` of the witnesses the timed launches left in HBM (this is what
    is timed), and Groth16::verify -> frw_groth16_verify (host pairing) on EVERY proof of the last timed call, plus the same
    proofs against a statement with one public input changed (all must be rejected).  Bit-exactness of (A, B, C) against the
    prover restated in the exponent is tests/test_gpu_groth16.py's business."""
    import random
    q = eng.qap_info(handle)
    n, ni, nw = int(q.domain_size), L.num_instance, L.num_witness
    rng = random.Random(SEED)
    lim = lambda ks: np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    t0 = time.perf_counter()
    pk, vk = eng.groth16_setup(0, logn, *(rng.randrange(2, R_FR) for _ in range(5)))
    key_s = time.perf_counter() - t0
    ws_bytes = eng.groth16_workspace_bytes(pk, handle, nsig)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    proofs = torch.empty((nsig, 48), dtype=torch.int64, device=dev)
    bad = torch.empty(nsig, dtype=torch.int32, device=dev)
    rs_arr = np.stack([lim([rng.randrange(R_FR), rng.randrange(R_FR)]) for _ in range(nsig)])
    stream = torch.cuda.current_stream()
    run = lambda: eng.groth16_prove_dev(pk, handle, nsig, d_wit, d_inst, rs_arr, proofs, ws, ws_bytes, bad, stream.cuda_stream)
    run()
    torch.cuda.synchronize()
    if before_timed is not None:
        before_timed()                       # N > 1: every rank starts its timed calls together (control plane only)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    assert int(bad.abs().sum()) == 0
    one_ms = None
    if with_one_proof:
        # the reference's own call pattern (examples/pok_sig.rs: ONE proof): the same call with a batch of one, ten times
        one_proof = torch.empty((1, 48), dtype=torch.int64, device=dev)
        eng.groth16_prove_dev(pk, handle, 1, d_wit, d_inst, rs_arr[:1], one_proof, ws, ws_bytes, bad[:1], stream.cuda_stream)
        torch.cuda.synchronize()
        e0.record(stream)
        for _ in range(10):
            eng.groth16_prove_dev(pk, handle, 1, d_wit, d_inst, rs_arr[:1], one_proof, ws, ws_bytes, bad[:1], stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        one_ms = e0.elapsed_time(e1) / 10
        assert torch.equal(one_proof[0], proofs[0]), "the proof of signature 0 made alone differs from the one made in the batch"
    eng.groth16_pk_free(pk)
    verifier = frw.Groth16Verifier(vk)
    inst_h = d_inst[:nsig].cpu().numpy().view(np.uint64)
    proofs_h = proofs.cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    accepted = verifier.verify(inst_h, proofs_h)
    verify_s = time.perf_counter() - t0
    assert accepted.tolist() == [1] * nsig, "frw_groth16_verify rejects a proof the device made: %s" % accepted.tolist()
    other = inst_h.copy()
    other[np.arange(nsig), 1 + (np.arange(nsig) % (ni - 1)), 0] ^= np.uint64(1)     # one public input of every statement
    assert verifier.verify(other, proofs_h).tolist() == [0] * nsig, "a proof verifies for a statement it was not made for"
    verifier.close()
    return {"workload": "Groth16 proofs of resident Falcon-%d witnesses (ark-groth16 create_proof: witness map + 5 MSMs + assembly), "
                        "%d per call" % (L.n, nsig),
            "ms_per_call": round(ms, 3), "proofs_per_s": round(nsig / (ms * 1e-3), 1), "calls_timed": reps,
            "one_proof_per_call_ms": None if one_ms is None else round(one_ms, 3),
            "roofline": groth16_roofline(eng, int(q.log_domain_size), n, ni, nw, inst_h[0], d_wit[0].cpu().numpy().view(np.uint64),
                                         nsig / (ms * 1e-3)),
            "proving_key": {"points_g1": 2 * (ni + nw) + nw + n - 1 + 3, "points_g2": ni + nw + 2,
                            "frw_groth16_setup_s": round(key_s, 2)},
            "workspace_bytes_per_signature": ws_bytes // nsig,
            "verify": {"proofs": nsig, "host_threads": min(nsig, os.cpu_count() or 1, 32), "seconds": round(verify_s, 3),
                       "proofs_per_s": round(nsig / verify_s, 1)},
            "checked": "no unsatisfied rows; every proof of the last timed call accepted by frw_groth16_verify (pairing check "
                       "e(A,B) = e(alpha,beta) e(sum x_i gamma_abc_i, gamma) e(C, delta) on the host) for its own public "
                       "inputs and rejected with one public input changed"}


def prove_leg(eng, handle, dev, cdev, d_wit, d_inst, held, L, logn, world, rank, reps=10):
    """N > 1: what the reference does with the witness after generate_constraints (examples/pok_sig.rs:30-32), on every GPU of
    the node.  Proofs shard by signature exactly like witnesses -- every rank loads the (replicated) proving key and proves 64
    of the witnesses its own timed launches left in HBM; no collective on the data path.  The ranks start their timed calls
    together (a barrier), every rank verifies its own proofs on its host cores, and the rate is what all ranks proved
    over the slowest rank's time.  A rank that fails still reaches both collectives, so nobody waits for it."""
    import traceback
    nsig = min(64, held)
    res, err, reached = None, None, {"barrier": False}

    def together():
        reached["barrier"] = True
        sharding.barrier()
    try:
        res = time_groth16(eng, handle, dev, d_wit, d_inst, nsig, reps, L, logn, before_timed=together, with_one_proof=False)
    except Exception as ex:                      # noqa: BLE001 -- reported in the line and on stderr
        err = repr(ex)[:300]
        sys.stderr.write("bench.py rank %d: the proof leg raised %r\n%s" % (rank, ex, traceback.format_exc()))
        sys.stderr.flush()
        if not reached["barrier"]:
            sharding.barrier()
    ms = torch.tensor([res["ms_per_call"] if res else -1.0], dtype=torch.float64, device=cdev)
    all_ms = [float(x) for x in sharding.gather_per_signature(ms, world, rank, world).tolist()]
    out = {"ranks": world, "proofs_per_call_per_rank": nsig, "calls_timed": reps, "ms_per_call_per_rank": [round(x, 3) for x in all_ms],
           "sharding": "by signature index; proving key replicated; no data-path collective"}
    if min(all_ms) <= 0:
        out["error"] = err or "rank(s) %s failed" % [r for r, x in enumerate(all_ms) if x <= 0]
        return out
    out["proofs_per_s_all_gpus"] = round(world * nsig / (max(all_ms) * 1e-3), 1)
    out["all_proofs_verified"] = True            # time_groth16 asserts it on every rank (frw_groth16_verify, every proof of the last call)
    if res:
        out["rank0"] = {k: res[k] for k in ("proofs_per_s", "proving_key", "workspace_bytes_per_signature", "verify", "roofline")}
    return out


def aggregate_statements(eng, dev, logns, first_index=0):
    """The statements of an aggregate on the device: per parameter set ONE launch of the witness kernel (synthetic triples, counter-based:
    the same on every rank), then the aggregate's handle and its own assignment vectors (frw_aggregate_assign_dev)."""
    s0 = torch.cuda.current_stream().cuda_stream
    batches = {}
    for g in (9, 10):
        cnt = list(logns).count(g)
        if not cnt:
            continue
        L = frw.layout(g)
        sig, pk, hm = frw.synth_triples(g, cnt, SEED, (1 << 43) + (g << 20) + first_index)
        d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
        wit = torch.empty((cnt, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((cnt, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.empty(cnt, dtype=torch.int32, device=dev)
        eng.witness_ntt_verify_dev(g, cnt, d[0], d[1], d[2], wit, inst, st, frw.ENC_MONTGOMERY, s0)
        torch.cuda.synchronize()
        assert not st.any()
        batches[g] = (wit, inst)
    return batches


def aggregate_roofline(eng, logns, log_n, batches, key_mode, proofs_per_s, h_windows=16, side_counts=None, h_points=None):
    """groth16_roofline for one proof of an aggregate statement: the point additions of the four witness-side sums counted from ONE
    statement of each parameter set (every statement of a set has the same structure: a few hundred additions either way) times the
    number of such statements; a key of window tables sums the ones of eight points in one addition (byte-pattern tables) up to 2^18
    points, a key of bare handles adds every one; the sum over h_query is one mixed addition per point and window -- 16 (n - 1), or
    13 (n - 1) when the bare handle runs 20-bit windows (frw_msm_info of the key's h_query says which: `h_windows`) -- its buckets' folds
    once per call (tables) or once per 32,768-bucket row (bare: 16 rows; wide windows: 13 x 16 rows, each bucket in a weighted and a
    plain running sum as ever), plus Horner's 255 operations per sum.
    side_counts (a key of bare handles: frw_diag_groth16_side_counts on the proof's own scalars): the sums over b_g1_query / b_g2_query
    run over the rows that hold a point there -- their additions are what the device counted, not an estimate.
    h_points: the rows of h_query this key holds (a key in slices: the line is then THIS RANK's work -- its slices of the five sums and
    the whole witness map -- against one chip's peak)."""
    rates = eng.valu_rates()
    (mm, mo), (am, ao) = point_op_instructions(MADD_OPS), point_op_instructions(ADD_OPS)
    n = 1 << log_n
    g1_madds = g2_madds = l_madds = 0
    for g in (9, 10):
        cnt = list(logns).count(g)
        if not cnt:
            continue
        wit, inst = batches[g]
        inst_row, wit_row = inst[0, 1:].cpu().numpy().view(np.uint64), wit[0].cpu().numpy().view(np.uint64)
        ones, digits, _ = witness_side_additions(inst_row, wit_row)
        ones_w, digits_w, _ = witness_side_additions(inst_row[:0], wit_row)
        g1_madds += cnt * (2 * (ones + digits) + (ones_w + digits_w))
        g2_madds += cnt * (ones + digits)
        l_madds += cnt * (ones_w + digits_w)
    bare = key_mode == frw.KEY_BARE
    if bare and side_counts is not None:
        _, digits_all, ones_all, digits_b, ones_b = side_counts
        g1_madds = (digits_all + ones_all) + l_madds + (digits_b + ones_b)          # a_query, l_query (no instance rows), b_g1_query
        g2_madds = digits_b + ones_b
    rows_h, windows_w = ((16 if h_windows == 16 else h_windows * 16), 32) if bare else (1, 1)
    G2_FACTOR = 3
    h_madds, h_adds = h_windows * (n - 1 if h_points is None else h_points), rows_h * 2 * 32768 + (255 if bare else 0)
    g1_adds = 3 * (windows_w * (2 * 128 + 64) + (255 if bare else 0))
    g2_adds = windows_w * (2 * 128 + 64) + (255 if bare else 0)
    madds = h_madds + g1_madds + G2_FACTOR * g2_madds
    adds = h_adds + g1_adds + G2_FACTOR * g2_adds
    fq_products = madds * 10 + adds * 14
    fr_products = qap_products_per_map_any(log_n)
    wave_us = ((madds * mm + adds * am) / rates["v_mad_u64_u32"] + (madds * mo + adds * ao) / rates["v_add_u32"]
               + fr_products * (F29_MUL_MAD64 / rates["v_mad_u64_u32"] + F29_MUL_OTHER / rates["v_add_u32"]))
    peak_proofs = rates["simds"] * 64 / wave_us * 1e6
    total = fq_products + fr_products
    return {"bound": "valu_issue", "unit": "G field products/s (Fq products of the five sums + Fr products of the witness map)",
            "achieved": round(total * proofs_per_s / 1e9, 2), "peak": round(total * peak_proofs / 1e9, 2),
            "frac": round(proofs_per_s / peak_proofs, 4),
            "per_proof": {"fq_products": fq_products, "fr_products": fr_products,
                          "mixed_additions": {"h_query": h_madds, "a_query + b_g1_query + l_query": g1_madds, "b_g2_query (Fq2)": g2_madds,
                                              "rows_of_b_queries_holding_a_point": None if side_counts is None else side_counts[0]},
                          "g2_priced_as_g1_times": G2_FACTOR, "transform_passes": qap_pass_stages(log_n)},
            "peak_is": "%d SIMDs x 64 lanes issuing only the instructions of these products (an Fq product %d multiplies + %d others, an Fr "
                       "product %d + %d; multiplies at %.1f, the others at %.1f wave-instructions/SIMD/us, measured in this process)"
                       % (rates["simds"], FQ_MUL_MULTIPLY, FQ_MUL_OTHER, F29_MUL_MAD64, F29_MUL_OTHER, rates["v_mad_u64_u32"], rates["v_add_u32"]),
            "kernel": "frw_groth16_prove_dev on the aggregate handle: ntt_pass_kernel + r1cs_eval (witness map), msm_bucket_kernel (h_query, %s), "
                      "nmsm_bucket_kernel / nmsm_ones_kernel (witness-side sums), folds"
                      % (("%d window rows over the bare points" % rows_h) if bare else "window tables")}


def time_aggregate_proof(eng, dev, logns, reps, separate=None, check_h=True, world=1, rank=0, cdev=None):
    """BASELINE configs[4] as written: ONE Groth16 proof for an aggregate statement -- FalconNTTVerificationCircuit once per
    (pk, msg, sig) on one constraint system (falcon_ntt.rs:26-123 per statement; the flow of examples/pok_sig.rs:30-47 on the
    whole).  The statements' witnesses come from the witness kernel (one launch per parameter set), frw_aggregate_assign_dev lays
    them out as the aggregate's assignment, frw_groth16_setup_r1cs_opts makes the key ON THE DEVICE (window tables while they fit,
    bare handles -- the points only -- beyond: the 1,024 statements of configs[4] are 121.9 M variables on the 2^27 domain, 83 GB of
    points), and what is timed is the proof: the witness map over the aggregate's domain, five sums, one proof.
    world > 1: the key in `world` slices, one per rank -- every rank proves the WHOLE statement's witness map and its slices of the five
    sums (frw_groth16_prove_partial_dev), ONE all-gather of 576 bytes per rank over the process group, frw_groth16_prove_combine_dev
    on every rank; the time of a proof is then the slowest rank's (strong scaling: the statement is fixed).
    The proof is verified (frw_groth16_verify, host pairing; also against a statement with one public input of the LAST statement
    changed); h_acc is checked against the MSM-free value (h(t) zt / delta) G1 (h(t) by Horner's rule on the device).  `separate`: the
    per-signature prover's line (time_groth16) for the break-even against k separate proofs."""
    import random
    rng = random.Random(SEED ^ len(logns))
    stream = torch.cuda.current_stream()
    s0 = stream.cuda_stream
    t_wit = time.perf_counter()
    batches = aggregate_statements(eng, dev, logns)
    handle = eng.r1cs_load_aggregate(list(logns))
    info = eng.r1cs_info(handle)
    ni, nw, nc, n = int(info.num_instance), int(info.num_witness), int(info.num_constraints), 1 << int(info.log_domain_size)
    d_wit = torch.empty((1, nw, 4), dtype=torch.int64, device=dev)
    d_inst = torch.empty((1, ni, 4), dtype=torch.int64, device=dev)
    b9, b10 = batches.get(9, (None, None)), batches.get(10, (None, None))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    eng.aggregate_assign_dev(handle, b9[0], b9[1], b10[0], b10[1], d_wit, d_inst, s0)
    e1.record(stream)
    torch.cuda.synchronize()
    assign_ms = e0.elapsed_time(e1)
    load_s = time.perf_counter() - t_wit
    toxic = [rng.randrange(2, R_FR) for _ in range(5)]
    t0 = time.perf_counter()
    pk_h, vk = eng.groth16_setup_r1cs(handle, *toxic, mode=frw.KEY_BARE if world > 1 else frw.KEY_AUTO, rank=rank, world=world, want_vk=(rank == 0))
    torch.cuda.synchronize()
    key_s = time.perf_counter() - t0
    pinfo = eng.groth16_pk_info(pk_h)
    key_mode = int(pinfo.mode)
    ws_bytes = eng.groth16_workspace_bytes(pk_h, handle, 1)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
    proof = torch.empty((1, 48), dtype=torch.int64, device=dev)
    bad = torch.empty(1, dtype=torch.int32, device=dev)
    lim = lambda ks: np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    rs = np.stack([lim([rng.randrange(R_FR), rng.randrange(R_FR)])])
    if world > 1:
        part = torch.empty((1, frw.GROTH16_PARTIAL_WORDS), dtype=torch.int64, device=dev)
        cws = torch.empty(frw.GROTH16_COMBINE_WORKSPACE, dtype=torch.uint8, device=dev)

        def run():
            eng.groth16_prove_partial_dev(pk_h, handle, 1, d_wit, d_inst, rs, part, ws, ws_bytes, bad, s0)
            if dist.get_backend() != "nccl":
                torch.cuda.synchronize()
            parts = sharding.all_gather_bytes(part[0], world, rank)           # 576 bytes per rank: RCCL over xGMI (gloo in the rehearsals)
            eng.groth16_prove_combine_dev(pk_h, world, parts, rs[0], proof, cws, cws.numel(), s0)
    else:
        run = lambda: eng.groth16_prove_dev(pk_h, handle, 1, d_wit, d_inst, rs, proof, ws, ws_bytes, bad, s0)
    run()
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    sharding.barrier()
    wall_ms = (time.perf_counter() - t0) * 1e3 / reps
    ms = e0.elapsed_time(e1) / reps if world == 1 else sharding.max_over_ranks(wall_ms, cdev)
    assert bad.tolist() == [0], "the aggregate's witness violates its constraint system"
    # the witness map on its own (timed), and h_acc against the value no MSM is needed for
    q = eng.qap_info(handle)
    assert int(q.workspace_bytes_per_signature) <= ws_bytes
    h = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
    eng.qap_witness_map_dev(handle, 1, d_wit, d_inst, h, ws, ws_bytes, None, s0)
    torch.cuda.synchronize()
    map_reps = max(1, min(reps, 5))
    e0.record(stream)
    for _ in range(map_reps):
        eng.qap_witness_map_dev(handle, 1, d_wit, d_inst, h, ws, ws_bytes, None, s0)
    e1.record(stream)
    torch.cuda.synchronize()
    map_ms = e0.elapsed_time(e1) / map_reps
    assert int(h[0, -1].abs().sum()) == 0, "deg h must be <= n - 2 for a satisfied system"
    h_sum_ms = None
    if check_h and world == 1:
        hq = eng.groth16_pk_query(pk_h, 0)
        assert int(eng.msm_info(hq).workspace_bytes_per_signature) <= ws_bytes
        hacc = torch.empty((1, 12), dtype=torch.int64, device=dev)
        eng.groth16_msm_h_dev(hq, 1, h, n, hacc, ws, ws_bytes, s0)
        torch.cuda.synchronize()
        e0.record(stream)
        eng.groth16_msm_h_dev(hq, 1, h, n, hacc, ws, ws_bytes, s0)
        e1.record(stream)
        torch.cuda.synchronize()
        h_sum_ms = e0.elapsed_time(e1)
        t, delta = toxic[4], toxic[3]
        h_t = eng.diag_poly_eval_dev(h, n, t)
        c = (pow(t, n, R_FR) - 1) * pow(delta, -1, R_FR) % R_FR
        want = eng.g1_fixed_base(np.frombuffer((h_t * c % R_FR).to_bytes(32, "little"), dtype=np.uint64).reshape(1, 4))[0]
        assert np.array_equal(hacc[0].cpu().numpy().view(np.uint64), want), "h_acc differs from (h(t) zt / delta) G1"
    h_windows = int(eng.msm_info(eng.groth16_pk_query(pk_h, 0)).num_windows)       # 16 windows of 16 bits, or 13 of 20 (bare handles from 2^26 points)
    side_counts = None
    if rank == 0 and key_mode == frw.KEY_BARE:
        # z ++ [1, r, s] of this key's slice (the three tail scalars as zeros: three additions in 10^8)
        z = torch.cat([d_inst[0], d_wit[0], torch.zeros((3, 4), dtype=torch.int64, device=dev)])[int(pinfo.z_lo):int(pinfo.z_hi)].contiguous()
        side_counts = eng.diag_groth16_side_counts(pk_h, z, ws, ws_bytes, s0)
        del z
    rates = aggregate_roofline(eng, logns, int(info.log_domain_size), batches, key_mode, 1e3 / ms, h_windows, side_counts,
                               int(pinfo.h_hi) - int(pinfo.h_lo)) if rank == 0 else None
    if rates is not None and world > 1:
        rates["of"] = "rank 0's share of the proof (its slices of the five sums, the whole witness map) against one chip's peak"
    hbm_in_use = torch.cuda.mem_get_info(dev)
    eng.groth16_pk_free(pk_h)
    del ws, h
    k = len(logns)
    inst_h = d_inst.cpu().numpy().view(np.uint64)
    proof_h = proof.cpu().numpy().view(np.uint64)
    verify_s = None
    if world > 1:
        # every rank must hold the same 384 bytes
        dig = torch.from_numpy(proof_h.view(np.int64).copy()).reshape(-1).to(cdev)
        every = sharding.gather_per_signature(dig, 48 * world, rank, world).reshape(world, 48)
        assert bool((every == every[0]).all()), "the ranks' combined proofs differ"
    if rank == 0:
        verifier = frw.Groth16Verifier(vk, points_are_checked=key_mode == frw.KEY_BARE)      # (a key made here a moment ago: 1.57 M points of gamma_abc_g1 for 1,024 statements)
        t0 = time.perf_counter()
        ok = verifier.verify(inst_h, proof_h).tolist()
        verify_s = time.perf_counter() - t0
        assert ok == [1], "frw_groth16_verify rejects the aggregate proof"
        other = inst_h.copy()
        other[0, ni - 3, 0] ^= np.uint64(1)
        assert verifier.verify(other, proof_h).tolist() == [0], "the aggregate proof verifies for another statement"
        verifier.close()
    eng.r1cs_free(handle)
    if rank != 0:
        return {"ms_per_proof": round(ms, 3)}
    plan = sharding.sharded_aggregate_plan(world, rank, list(logns))
    out = {"workload": "ONE Groth16 proof for %d Falcon statements (%d x Falcon-512, %d x Falcon-1024) on one constraint system: "
                       "I = %d, W = %d, C = %d, QAP domain 2^%d" % (k, list(logns).count(9), list(logns).count(10), ni, nw, nc,
                                                                  int(info.log_domain_size)),
           "ms_per_proof": round(ms, 3), "signatures_per_s": round(k / (ms * 1e-3), 1), "calls_timed": reps,
           "witness_map_ms": round(map_ms, 3), "assignment_ms": round(assign_ms, 3),
           "h_query_sum_alone_ms": None if h_sum_ms is None else round(h_sum_ms, 3),
           "proof_bytes": {"this_aggregate": 48 * 8, "k_separate_proofs": k * 48 * 8,
                           "note": "uncompressed ark-ff limbs (A 96 B, B 192 B, C 96 B); 192 B compressed either way"},
           "proving_key": {"kind": "bare handles (the points only)" if key_mode == frw.KEY_BARE else "window tables",
                           "made": "on the device (frw_groth16_setup_r1cs_opts)", "slices": world, "bytes_this_rank": int(pinfo.key_bytes),
                           "rows_this_rank": {"witness_side_tables": [int(pinfo.z_lo), int(pinfo.z_hi)], "h_query": [int(pinfo.h_lo), int(pinfo.h_hi)]},
                           "points_g1": 3 * (ni + nw) - ni + n - 1 + 3, "points_g2": ni + nw + 2, "setup_s": round(key_s, 2)},
           "statements_and_handle_s": round(load_s, 2),
           "workspace_bytes": ws_bytes,
           "hbm": {"plan_bytes_leg": plan["hbm_plan_bytes"] if key_mode == frw.KEY_BARE else None, "plan_limit_bytes": plan["hbm_limit_bytes"],
                   "device_bytes_in_use_while_proving": int(hbm_in_use[1] - hbm_in_use[0])},
           "verify": {"seconds": round(verify_s, 4), "public_inputs": ni - 1, "pairings": 3,
                      "what": "frw_groth16_verify: prepare_inputs over %d public inputs (a few host threads beyond 16,384) + 3 Miller loops + "
                              "1 final exponentiation" % (ni - 1)},
           "proof_sha256": __import__("hashlib").sha256(proof_h.tobytes()).hexdigest(),
           "roofline": rates,
           "checked": "constraint system satisfied (0 violated rows of %d); deg h <= n - 2; %sproof accepted by frw_groth16_verify for its "
                      "%d public inputs and rejected with one input of the last statement changed; bit-exactness of (A, B, C) against the "
                      "prover restated in the exponent (2^18 - 2^22, window tables == bare handles == keys in slices): tests/test_gpu_aggregate.py"
                      % (nc, "h_acc == (h(t) zt / delta) G1; " if h_sum_ms is not None else "", ni - 1)}
    if world > 1:
        out["sharding"] = ("the key in %d slices by row range (every query); every rank: the whole witness map + its slices of the five sums; ONE "
                           "all-gather of 576 bytes per rank (%s); the combination on every rank; all ranks' proofs byte-identical" % (world, dist.get_backend()))
        out["scaling"] = "strong (one statement whatever the number of ranks)"
    if separate is not None and k and all(g == 10 for g in logns):
        per = separate["proofs_per_s"]
        out["against_k_separate_proofs"] = {
            "separate_ms_for_k_proofs_at_64_per_call": round(k / per * 1e3, 3),
            "separate_ms_for_k_proofs_one_per_call": None if separate.get("one_proof_per_call_ms") is None else round(k * separate["one_proof_per_call_ms"], 3),
            "aggregate_over_separate_time": round(ms / (k / per * 1e3), 3),
            "separate_verify_s_for_k_proofs_one_thread": round(k * separate["verify"]["seconds"] * separate["verify"]["host_threads"] / separate["verify"]["proofs"], 4),
            "verifier_pairings": {"aggregate": 3, "separate": 3 * k},
            "break_even": "one proof instead of %d: %dx fewer proof bytes and pairings for %.2fx the proving time of %d separate proofs made 64 "
                          "per call" % (k, k, ms / (k / per * 1e3), k)}
    return out


def time_configs4_as_aggregates_of_16(eng, dev):
    """BASELINE configs[4]'s 1,024 mixed signatures proved FOR REAL as 64 proofs of 16 statements each, one after the other on one GPU --
    what round 4's bench line gave as a multiplication.  The same mix as `aggregate_proof_1024_mixed` (513 Falcon-512 + 511 Falcon-1024),
    grouped by parameter set so that three keys serve all 64 aggregates: 32 x (16 Falcon-512), 31 x (16 Falcon-1024) and the 16 left over
    (1 Falcon-512 + 15 Falcon-1024).  Timed: assignment + proof of all 64, back to back (keys made before; witnesses resident); every
    proof verified afterwards (frw_groth16_verify, one proof per host thread)."""
    import random
    rng = random.Random(SEED ^ 0xC4)
    stream = torch.cuda.current_stream()
    s0 = stream.cuda_stream
    mix = sharding.aggregate_mix(1024)
    n9, n10 = mix.count(9), mix.count(10)
    batches = aggregate_statements(eng, dev, mix)
    shapes = [((9,) * 16, n9 // 16), ((10,) * 16, n10 // 16), ((9,) * (n9 % 16) + (10,) * (n10 % 16), 1 if (n9 % 16 + n10 % 16) else 0)]
    assert sum(len(l) * c for l, c in shapes) == 1024 and len(shapes[2][0]) in (0, 16)
    lim = lambda ks: np.frombuffer(b"".join(int(k).to_bytes(32, "little") for k in ks), dtype=np.uint64).reshape(-1, 4)
    keys, t_keys = [], time.perf_counter()
    for logns, count in shapes:
        if not count:
            continue
        handle = eng.r1cs_load_aggregate(list(logns))
        info = eng.r1cs_info(handle)
        pk_h, vk = eng.groth16_setup_r1cs(handle, *[rng.randrange(2, R_FR) for _ in range(5)])
        ws_bytes = eng.groth16_workspace_bytes(pk_h, handle, 1)
        keys.append({"logns": logns, "count": count, "handle": handle, "pk": pk_h, "vk": vk, "ws_bytes": ws_bytes,
                     "ni": int(info.num_instance), "nw": int(info.num_witness)})
    torch.cuda.synchronize()
    key_s = time.perf_counter() - t_keys
    ws = torch.empty(max(k["ws_bytes"] for k in keys), dtype=torch.uint8, device=dev)
    total = sum(k["count"] for k in keys)
    proofs = torch.empty((total, 48), dtype=torch.int64, device=dev)
    bad = torch.full((total,), -1, dtype=torch.int32, device=dev)
    d_wit = [torch.empty((k["count"], k["nw"], 4), dtype=torch.int64, device=dev) for k in keys]
    d_inst = [torch.empty((k["count"], k["ni"], 4), dtype=torch.int64, device=dev) for k in keys]
    rs = np.stack([lim([rng.randrange(R_FR), rng.randrange(R_FR)]) for _ in range(total)])
    L9, L10 = frw.layout(9), frw.layout(10)

    def all_of_them():
        used, at = {9: 0, 10: 0}, 0
        for ki, k in enumerate(keys):
            c9, c10 = k["logns"].count(9), k["logns"].count(10)
            for j in range(k["count"]):
                w9 = batches[9][0][used[9]:] if c9 else None
                i9 = batches[9][1][used[9]:] if c9 else None
                w10 = batches[10][0][used[10]:] if c10 else None
                i10 = batches[10][1][used[10]:] if c10 else None
                eng.aggregate_assign_dev(k["handle"], w9, i9, w10, i10, d_wit[ki][j:j + 1], d_inst[ki][j:j + 1], s0)
                eng.groth16_prove_dev(k["pk"], k["handle"], 1, d_wit[ki][j:j + 1], d_inst[ki][j:j + 1], rs[at:at + 1], proofs[at:at + 1], ws,
                                      k["ws_bytes"], bad[at:at + 1], s0)
                used[9] += c9
                used[10] += c10
                at += 1
        assert used == {9: n9, 10: n10} and at == total
    all_of_them()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    all_of_them()
    torch.cuda.synchronize()
    seconds = time.perf_counter() - t0
    assert bad.tolist() == [0] * total, "an aggregate's witness violates its constraint system"
    proofs_h = proofs.cpu().numpy().view(np.uint64)
    t0 = time.perf_counter()
    at = 0
    for ki, k in enumerate(keys):
        ver = frw.Groth16Verifier(k["vk"])
        inst_h = d_inst[ki].cpu().numpy().view(np.uint64)
        ok = ver.verify(inst_h, proofs_h[at:at + k["count"]]).tolist()
        assert ok == [1] * k["count"], "frw_groth16_verify rejects an aggregate proof"
        other = inst_h.copy()
        other[:, k["ni"] - 3, 0] ^= np.uint64(1)
        assert ver.verify(other, proofs_h[at:at + k["count"]]).tolist() == [0] * k["count"]
        ver.close()
        at += k["count"]
    verify_s = time.perf_counter() - t0
    for k in keys:
        eng.groth16_pk_free(k["pk"])
        eng.r1cs_free(k["handle"])
    return {"workload": "BASELINE configs[4]'s 1,024 mixed signatures (%d Falcon-512 + %d Falcon-1024) as %d proofs of 16 statements each, one after "
                        "the other on ONE GPU: %s" % (n9, n10, total, ", ".join("%d x (%d Falcon-512 + %d Falcon-1024)" % (k["count"], k["logns"].count(9),
                                                                                                                      k["logns"].count(10)) for k in keys)),
            "run": "timed, not extrapolated: assignment + frw_groth16_prove_dev of all %d aggregates back to back (wall clock, second of two rounds)" % total,
            "seconds": round(seconds, 3), "signatures_per_s": round(1024 / seconds, 1), "proofs": total, "ms_per_proof": round(seconds / total * 1e3, 3),
            "keys": len(keys), "keys_setup_s": round(key_s, 2), "all_proofs_verified": True, "verify_s_all": round(verify_s, 3),
            "against_one_proof_for_all_1024": "secondary.aggregate_proof_1024_mixed"}


def aggregate_leg(eng, dev, cdev, k, world, rank, reps=5):
    """N > 1, BASELINE configs[4] on the node: aggregates shard like signatures -- every rank makes ONE proof for k Falcon-1024
    statements of its own (time_aggregate_proof: witnesses, assignment, key, proof, verification), no collective on the data path; the
    node's rate is all ranks' signatures over the slowest rank's proof time (1,024 signatures = 64 such aggregates of 16, eight per GPU).
    A rank that fails still reaches the two control-plane collectives."""
    import traceback
    res, err = None, None
    try:
        res = time_aggregate_proof(eng, dev, (10,) * k, reps, None)
    except Exception as ex:                      # noqa: BLE001 -- reported in the line and on stderr
        err = repr(ex)[:300]
        sys.stderr.write("bench.py rank %d: the aggregate leg raised %r\n%s" % (rank, ex, traceback.format_exc()))
        sys.stderr.flush()
    sharding.barrier()
    ms = torch.tensor([res["ms_per_proof"] if res else -1.0], dtype=torch.float64, device=cdev)
    all_ms = [float(x) for x in sharding.gather_per_signature(ms, world, rank, world).tolist()]
    out = {"ranks": world, "statements_per_proof": k, "ms_per_proof_per_rank": [round(x, 3) for x in all_ms],
           "sharding": "by aggregate; every rank holds the aggregate's key; no data-path collective"}
    if min(all_ms) <= 0:
        out["error"] = err or "rank(s) %s failed" % [r for r, x in enumerate(all_ms) if x <= 0]
        return out
    out["signatures_per_s_all_gpus"] = round(world * k / (max(all_ms) * 1e-3), 1)
    out["proofs_per_s_all_gpus"] = round(world / (max(all_ms) * 1e-3), 2)
    out["seconds_for_1024_signatures_on_this_node"] = round(1024.0 / (world * k) * max(all_ms) * 1e-3, 3)
    if res:
        out["rank0"] = {key: res[key] for key in ("workload", "proving_key", "verify", "witness_map_ms", "workspace_bytes")}
    return out


def time_msm(eng, dev, d_h, reps, with_cpu):
    """The step after the witness map in the reference's consumer (examples/pok_sig.rs:30-47 -> ark-groth16 prover.rs):
    h_acc = VariableBaseMSM(pk.h_query, h) over BLS12-381 G1, for the h vectors the witness map left in HBM."""
    nsig, n = d_h.shape[0], d_h.shape[1]
    bases, (t, c) = groth16_h_query(eng, n)
    m = eng.msm_g1_load(bases)
    info = eng.msm_info(m)
    ws = torch.empty(nsig * int(info.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
    out = torch.empty((nsig, 12), dtype=torch.int64, device=dev)
    stream = torch.cuda.current_stream()
    run = lambda: eng.groth16_msm_h_dev(m, nsig, d_h, n, out, ws, ws.numel(), stream.cuda_stream)
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    eng.msm_free(m)
    per_s = nsig / (ms * 1e-3)
    # MSM-free check of signature 0: sum h_i h_query[i] = (h(t) zt / delta) G1 -- one fixed-base multiple, on the device too
    r_inv = pow(1 << 256, -1, R_FR)
    h0 = d_h[0].cpu().numpy().view(np.uint64)
    acc = 0
    for row in h0[n - 2::-1]:
        acc = (acc * t + int.from_bytes(row.tobytes(), "little") * r_inv) % R_FR
    want = eng.g1_fixed_base(np.frombuffer((acc * c % R_FR).to_bytes(32, "little"), dtype=np.uint64).reshape(1, 4))[0]
    got = out[0].cpu().numpy().view(np.uint64)
    assert np.array_equal(got, want), "h_acc differs from (h(t) zt / delta) G1"
    rates = eng.valu_rates()
    # priced by the instructions of the field operations inside the point formulas, each class at its measured issue rate:
    # mixed additions into the buckets (16 per point) + the fold's full additions (2 per bucket)
    madds, adds = 16 * (n - 1), 2 * 32768
    products = madds * 10 + adds * 14
    (mm, mo), (am, ao) = point_op_instructions(MADD_OPS), point_op_instructions(ADD_OPS)
    wave_us = (madds * mm + adds * am) / rates["v_mad_u64_u32"] + (madds * mo + adds * ao) / rates["v_add_u32"]   # one lane's work, in SIMD-microseconds per wave-instruction
    peak_sig_per_s = rates["simds"] * 64 / wave_us * 1e6
    peak = products * peak_sig_per_s / 1e9
    res = {"workload": "Groth16 h_acc = sum h_i h_query[i] over BLS12-381 G1, %d points, %d resident h vectors per call" % (n - 1, nsig),
           "ms_per_call": round(ms, 3), "signatures_per_s": round(per_s, 1), "calls_timed": reps,
           "mixed_additions_per_signature": madds, "window_table_bytes": int(info.table_bytes),
           "checked": "h_acc of one signature == (h(t) zt / delta) G1 for the known toxic waste, bit for bit (affine, ark-ff's bytes)",
           "roofline": {"bound": "valu_issue", "unit": "G Fq products/s", "achieved": round(products * per_s / 1e9, 2),
                        "peak": round(peak, 2), "frac": round(per_s / peak_sig_per_s, 4),
                        "kernel": "msm_bucket_kernel (+ count / scan / scatter / fold)", "fq_products_per_signature": products,
                        "all_instructions": {
                            "valu_instructions_per_wavefront_addition": MSM_VALU_PER_MADD, "of_which_multiplies": mm,
                            "from": "SQ_INSTS_VALU of msm_bucket_kernel<FqField, true>, profiles/r04_msm_counters.txt (tools/pmc_msm.sh)",
                            "frac_of_issue_bound": round(per_s / (rates["simds"] * 64 / (madds * (mm / rates["v_mad_u64_u32"] + (MSM_VALU_PER_MADD - mm) / rates["v_add_u32"])) * 1e6), 4),
                            "note": "`frac` above prices the field products' instructions only; with EVERY vector instruction the bucket kernel issues per "
                                    "addition (carry passes of the sums and differences, zero tests, sign select, unpacking) the same time is this "
                                    "fraction of what the vector ALU can issue: the kernel is issue-bound, what is left is instruction count"},
                        "peak_is": "%d SIMDs x 64 lanes; a mixed addition = 6 products (%d multiplies + %d other instructions) + 2 squares "
                                   "(%d + %d) + one a b - c d with a shared reduction (%d + %d), a full addition 10 + 2 + 1 of the same; "
                                   "multiplies at %.1f, the others at %.1f wave-instructions/SIMD/us"
                                   % (rates["simds"], FQ_MUL_MULTIPLY, FQ_MUL_OTHER, FQ_SQR_MULTIPLY, FQ_SQR_OTHER, FQ_PAIR_MULTIPLY,
                                      FQ_PAIR_OTHER, rates["v_mad_u64_u32"], rates["v_add_u32"])}}
    if with_cpu:
        oracle = load_oracle()
        h_can = np.frombuffer(b"".join((int.from_bytes(row.tobytes(), "little") * r_inv % R_FR).to_bytes(32, "little") for row in h0[:n - 1]),
                              dtype=np.uint64).reshape(-1, 4)
        nproc = os.cpu_count() or 1
        t0 = time.perf_counter()
        cpu = oracle.g1_msm(bases, h_can, 16, threads=nproc)
        sec = time.perf_counter() - t0
        assert np.array_equal(cpu, got), "GPU h_acc differs from the oracle's bucket method"
        res["cpu_port"] = {"seconds_per_msm": round(sec, 3), "threads": min(nproc, 17), "kind": "port",
                           "what": "oracle/bls12_381.c: bucket method, 16-bit signed windows, one thread per window (17), Jacobian "
                                   "mixed additions, 64-bit-limb Montgomery arithmetic; same point as the GPU's, bit for bit",
                           "gpu_msms_per_cpu_msm": round(per_s * sec, 1)}
    return res


def time_host_call(eng, logn, reps=200):
    """The call the reference's consumers make: ONE signature per generate_constraints (examples/constraint_counts.rs:61-63,
    pok_sig.rs:24-32), host buffers in and out (frw_witness_ntt_verify, batch = 1).  Median wall time per call, PCIe
    included; the context's working memory must not grow while it runs."""
    import ctypes as C
    import statistics
    L, CL = frw.layout(logn), frw.compact_layout(logn)
    sig, pk, hm = frw.synth_triples(logn, 1, SEED, 1 << 42)
    P = lambda a: a.ctypes.data_as(C.c_void_p)
    lib, ctx = eng._lib, eng._ctx
    wit, inst = eng.pinned_empty((1, L.num_witness, 4), np.uint64), eng.pinned_empty((1, L.num_instance, 4), np.uint64)
    comp, st = eng.pinned_empty((1, CL.bytes_per_signature), np.uint8), eng.pinned_empty((1,), np.int32)
    out = {"workload": "frw_witness_ntt_verify, batch = 1, Falcon-%d, host buffers (page-locked outputs), strict" % L.n}
    for name, call in (("arkworks_layout", lambda: lib.frw_witness_ntt_verify(ctx, logn, 1, P(sig), P(pk), P(hm), 1, P(wit), P(inst), P(st), 1)),
                       ("compact", lambda: lib.frw_witness_ntt_verify(ctx, logn, 1, P(sig), P(pk), P(hm), 2, P(comp), None, P(st), 1))):
        for _ in range(5):
            assert call() == 0
        a0, ts = eng.host_allocations(), []
        for _ in range(reps):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        assert eng.host_allocations() == a0, "the host-buffer entry point allocated after its first call"
        out[name] = {"median_us_per_call": round(statistics.median(ts) * 1e6, 1), "best_us_per_call": round(min(ts) * 1e6, 1),
                     "calls_timed": reps, "allocations_during_the_timed_calls": 0}
    return out


def time_aggregate(eng, dev, total, reps, threads):
    """BASELINE configs[4] shape on one GPU: one aggregate statement of `total` signatures, Falcon-512 and Falcon-1024
    mixed (parameter set drawn from the seed), = one engine launch per parameter set (the reference's
    falcon-aggregate-sig is an empty stub).  Latency of the whole aggregate witness."""
    import random
    rng = random.Random(SEED)
    logns = [rng.choice([9, 10]) for _ in range(total)]
    stream = torch.cuda.current_stream()
    groups = []
    for logn in (9, 10):
        cnt = sum(1 for l in logns if l == logn)
        L = frw.layout(logn)
        sig, pk, hm = synth(logn, cnt, 1 << 41, threads)
        d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
        groups.append((logn, cnt, d, torch.empty((cnt, L.num_witness, 4), dtype=torch.int64, device=dev),
                       torch.empty((cnt, L.num_instance, 4), dtype=torch.int64, device=dev),
                       torch.empty(cnt, dtype=torch.int32, device=dev), L))

    def run():
        for logn, cnt, d, wit, inst, st, _ in groups:
            eng.witness_ntt_verify_dev(logn, cnt, d[0], d[1], d[2], wit, inst, st, frw.ENC_MONTGOMERY, stream.cuda_stream)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / reps
    assert all(int((g[5] != 0).sum().item()) == 0 for g in groups)
    nbytes = sum(g[1] * 32 * (g[6].num_witness + g[6].num_instance) for g in groups)
    return {"workload": "one aggregate statement of %d mixed signatures (%d Falcon-512 + %d Falcon-1024), two launches"
                        % (total, groups[0][1], groups[1][1]),
            "ms_per_aggregate": round(ms, 4), "signatures_per_s": round(total / (ms * 1e-3), 1),
            "bytes_written_GBs": round(nbytes / (ms * 1e-3) / 1e9, 1), "aggregates_timed": reps}


def bench_ntt_modq(args, world, rank, dev):
    """--workload ntt_modq: BASELINE configs[1] as the primary line."""
    eng = frw.WitnessEngine(dev.index)
    sharding.barrier()
    r = time_ntt_modq(eng, dev, args.logn, args.batch, args.steps, max(1, args.warmup))
    n = 1 << args.logn
    if rank == 0:
        emit({"metric": "falcon%d_ntt_modq_witness_polys_per_sec" % n, "value": r["polynomials_per_s"] * world,
              "unit": "polynomials/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
              "ms_per_step": r["avg_launch_ms"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
              "dtype": "u32", "data": "synthetic",
              "config": {"workload": r["workload"], "logn": args.logn, "batch_per_gpu": args.batch},
              "roofline": dict(r, traffic=None)})


def time_prepare(eng, dev, logn, batch, reps, warm, seed=0):
    """Input preparation: decode(pk) + decode(sig) + SHAKE256 hash-to-point for a resident batch of
    encoded (pk, msg, sig); 64-byte messages.  ALU/latency work, two orders of magnitude below the witness kernel."""
    n = 1 << logn
    rng = np.random.default_rng(SEED + seed)
    pk_len, sig_len, mlen = frw.PK_LEN[logn], frw.SIG_LEN[logn], 64
    # random bytes are fine for timing: pk fields are 14 random bits (some >= q -> status 3), signatures decode or not
    pkb = torch.from_numpy(rng.integers(0, 256, size=(batch, pk_len), dtype=np.uint8)).to(dev)
    sgb_h = np.zeros((batch, sig_len), dtype=np.uint8)
    sgb_h[:, 0] = 0x30 + logn
    sgb_h[:, 1:41] = rng.integers(0, 256, size=(batch, 40), dtype=np.uint8)
    sgb_h[:, 41:41 + n * 9 // 8] = 0x81          # every coefficient = "+1" then terminator pattern: decodes
    sgb = torch.from_numpy(sgb_h).to(dev)
    msgs = torch.from_numpy(rng.integers(0, 256, size=batch * mlen, dtype=np.uint8)).to(dev)
    off = torch.arange(0, (batch + 1) * mlen, mlen, dtype=torch.int64, device=dev)
    out = [torch.empty((batch, n), dtype=torch.int16, device=dev) for _ in range(3)]
    nonce = torch.empty((batch, 40), dtype=torch.uint8, device=dev)
    st = [torch.empty(batch, dtype=torch.int32, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream()
    lib, ctx = eng._lib, eng._ctx
    import ctypes as C
    P = lambda t: C.c_void_p(t.data_ptr())

    def run():
        assert lib.frw_decode_public_keys_dev(ctx, logn, batch, P(pkb), P(out[1]), P(st[0]), C.c_void_p(stream.cuda_stream)) == 0
        assert lib.frw_decode_signatures_dev(ctx, logn, batch, P(sgb), sig_len, P(out[0]), P(nonce), P(st[1]),
                                             C.c_void_p(stream.cuda_stream)) == 0
        assert lib.frw_hash_to_point_dev(ctx, logn, batch, P(nonce), P(msgs), P(off), P(out[2]),
                                         C.c_void_p(stream.cuda_stream)) == 0
    for _ in range(max(1, warm)):
        run()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ms = e0.elapsed_time(e1) / reps
    return {"workload": "decode pk + decode sig + SHAKE256 hash-to-point, %d-byte messages, %d Falcon-%d triples per pass"
                        % (mlen, batch, n),
            "ms_per_pass": round(ms, 4), "signatures_per_s": round(batch / (ms * 1e-3), 1), "passes_timed": reps,
            "wall_signatures_per_s": round(batch * reps / elapsed, 1)}


def bench_prepare(args, world, rank, dev):
    """--workload prepare: the input-preparation step as the primary line."""
    eng = frw.WitnessEngine(dev.index)
    r = time_prepare(eng, dev, args.logn, args.batch, args.steps, args.warmup, rank)
    n = 1 << args.logn
    emit({"metric": "falcon%d_input_preparation_signatures_per_sec" % n, "value": r["wall_signatures_per_s"],
          "unit": "signatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
          "ms_per_step": r["ms_per_pass"], "higher_is_better": True, "dtype": "u64 (Keccak lanes)", "data": "synthetic",
          "config": {"workload": r["workload"], "logn": args.logn, "batch_per_gpu": args.batch}})


def bench_qap(args, world, rank, dev):
    """--workload qap: witnesses of `--batch` signatures generated on the device, then the R1CS->QAP witness map of all of
    them per step (the step after the hot path in a Groth16 prover); signatures are independent, ranks too."""
    logn, batch = args.logn, args.batch
    eng = frw.WitnessEngine(dev.index)
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=SEED + rank)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    eng.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, frw.ENC_MONTGOMERY, stream.cuda_stream)
    handle = eng.r1cs_load(0, logn)
    q = eng.qap_info(handle)
    n, per = int(q.domain_size), int(q.workspace_bytes_per_signature)
    chunk = min(batch, 128)                                      # signatures in flight: 128 x 41 MB of workspace
    ws = torch.empty(chunk * per, dtype=torch.uint8, device=dev)
    h = torch.empty((batch, n, 4), dtype=torch.int64, device=dev)
    bad = torch.empty(batch, dtype=torch.int32, device=dev)
    run = lambda: eng.qap_witness_map_dev(handle, batch, wit, inst, h, ws, chunk * per, bad, stream.cuda_stream)
    for _ in range(max(1, args.warmup)):
        run()
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.steps):
        run()
    e1.record(stream)
    torch.cuda.synchronize()
    sharding.barrier()
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, dev if dist.is_initialized() and dist.get_backend() == "nccl" else torch.device("cpu"))
    ms = e0.elapsed_time(e1) / args.steps
    assert int(bad.abs().sum()) == 0 and int(h[:, -1].abs().sum()) == 0
    eng.r1cs_free(handle)
    emit({"metric": "falcon%d_qap_witness_maps_per_sec" % (1 << logn), "value": round(batch * world * args.steps / elapsed, 1),
          "unit": "signatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u32 (BLS12-381 Fr in 29-bit limbs)",
          "data": "synthetic",
          "config": {"workload": "R1CS->QAP witness map of resident witnesses (ark-groth16 witness_map), domain 2^%d" %
                                 int(q.log_domain_size), "logn": logn, "batch_per_gpu": batch, "signatures_in_flight": chunk},
          "roofline": qap_roofline(eng, int(q.log_domain_size), batch * args.steps / (ms * 1e-3 * args.steps),
                                   "ntt_pass_kernel<...> (six transforms = 11 launches per map of satisfied witnesses, + 8 that find an empty list: ark-groth16's seven run only for a witness that violates the system)"),
          "checked": "no unsatisfied rows; top coefficient of every h is zero"})


_REAL_STDOUT = None


def quiet_stdout():
    """The contract is ONE JSON line on stdout.  Libraries write there too (RCCL prints a version banner at
    communicator creation), so everything that goes to fd 1 during the run is sent to stderr and only emit() writes to
    the real stdout."""
    global _REAL_STDOUT
    if _REAL_STDOUT is None:
        sys.stdout.flush()
        _REAL_STDOUT = os.dup(1)
        os.dup2(2, 1)


def emit(obj):
    line = (json.dumps(obj) + "\n").encode()
    sys.stdout.flush()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, line)


def gather_leg(args, plan, eng, dev, cdev, world, rank, logn, batch, chunk, L, d_in, d_wit, d_st, stream):
    """Second curve (north_star / BASELINE configs[3]): generate + all-gather of the per-signature witness vectors for a
    FULL step, every chunk.  MI355X-first form: each rank generates its chunk in FRW_ENC_COMPACT (0.51 MB per signature
    instead of 5.08 MB), one all_gather_into_tensor (RCCL over xGMI) moves the compact chunks, and every receiver
    expands all `world` shards locally into the arkworks layout (frw_expand_dev) -- so every GPU ends up holding every
    witness of the chunk, byte-identical to the direct output.  Double-buffered: the expansion of chunk k-1 and the
    generation of chunk k+1 run while chunk k is on the fabric.  The expansion writes world x 5 MB per signature on
    every GPU, so this curve is bounded by HBM write bandwidth at ~1/world of `value` per GPU -- not by xGMI."""
    CL = frw.compact_layout(logn)
    d_sig, d_pk, d_hm = d_in
    # signatures per rank per collective: 4,096 (0.46 GB of compact witnesses per rank, world x that gathered, twice for
    # the double buffer) unless asked otherwise; the expansion target must fit the benchmark's witness buffer
    # (sharding.step_plan: the same arithmetic the CPU tests evaluate for world = 8)
    gc, nk = plan["gather_chunk_per_rank"], plan["gather_chunks"]
    loc = [torch.empty((gc, CL.bytes_per_signature), dtype=torch.uint8, device=dev) for _ in range(2)]
    gathered = [torch.empty((world, gc, CL.bytes_per_signature), dtype=torch.uint8, device=dev) for _ in range(2)]
    exp_wit = d_wit[:world * gc]
    exp_inst = torch.empty((world * gc, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(gc, dtype=torch.int32, device=dev)
    nccl = args.backend == "nccl"

    def run(k_count):
        works = [None, None]
        for k in range(k_count):
            b = k & 1
            a = k * gc
            eng.witness_ntt_verify_compact_dev(logn, gc, d_sig[a:a + gc], d_pk[a:a + gc], d_hm[a:a + gc], loc[b], st,
                                               stream.cuda_stream)
            works[b] = sharding.all_gather_chunks(loc[b], gathered[b], async_op=nccl)
            if k >= 1:
                if works[1 - b] is not None:
                    works[1 - b].wait()
                eng.expand_dev(logn, world * gc, gathered[1 - b], exp_wit, exp_inst, stream.cuda_stream)
        b = (k_count - 1) & 1
        if works[b] is not None:
            works[b].wait()
        eng.expand_dev(logn, world * gc, gathered[b], exp_wit, exp_inst, stream.cuda_stream)

    run(min(2, nk))                                       # warm-up: communicator, kernels
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    run(nk)
    torch.cuda.synchronize()
    sharding.barrier()
    t = sharding.max_over_ranks(time.perf_counter() - t0, cdev)
    # parity of what was gathered: the expanded buffer now holds the LAST chunk of every rank; its digests must be
    # identical on every rank and, for this rank's own shard, equal to a direct FRW_ENC_MONTGOMERY launch
    dig = torch.zeros(world * gc, dtype=torch.int64, device=dev)
    eng.digest_dev(exp_wit, L.num_witness * 4, world * gc, dig, stream.cuda_stream)
    torch.cuda.synchronize()
    a = (nk - 1) * gc
    oc = plan["own_shard_signatures_checked"]             # all gc when the witness buffer has room next to the expansion
    own_dig = torch.zeros(oc, dtype=torch.int64, device=dev)
    own_inst = torch.empty((oc, L.num_instance, 4), dtype=torch.int64, device=dev)
    own_wit = d_wit[world * gc:world * gc + oc] if plan["own_shard_checked_in_place"] else \
        torch.empty((oc, L.num_witness, 4), dtype=torch.int64, device=dev)
    eng.witness_ntt_verify_dev(logn, oc, d_sig[a:a + oc], d_pk[a:a + oc], d_hm[a:a + oc], own_wit, own_inst, st,
                               frw.ENC_MONTGOMERY, stream.cuda_stream)
    eng.digest_dev(own_wit, L.num_witness * 4, oc, own_dig, stream.cuda_stream)
    torch.cuda.synchronize()
    own_ok = bool(torch.equal(own_dig, dig[rank * gc:rank * gc + oc])) and \
        bool(torch.equal(own_inst, exp_inst[rank * gc:rank * gc + oc]))
    all_dig = sharding.gather_per_signature(dig.to(cdev), plan["all_digests_gathered"], rank, world) if world > 1 else dig.to(cdev)
    same = all(bool(torch.equal(all_dig[r * world * gc:(r + 1) * world * gc], all_dig[:world * gc])) for r in range(world))
    del own_wit
    return {"signatures_per_s_node": round(world * nk * gc / t, 1), "seconds": round(t, 4),
            "signatures_per_rank": nk * gc, "chunk_per_rank": gc, "chunks": nk,
            "wire_format": "FRW_ENC_COMPACT, %d bytes per signature (arkworks layout: %d)"
                           % (CL.bytes_per_signature, 32 * (L.num_witness + L.num_instance)),
            "fabric_ingest_GBs_per_gpu": round((world - 1) * nk * gc * CL.bytes_per_signature / t / 1e9, 1),
            "expanded_bytes_written_GBs_per_gpu": round(world * nk * gc * 32 * (L.num_witness + L.num_instance) / t / 1e9, 1),
            "collective": "all_gather_into_tensor (RCCL)" if nccl else "gloo rehearsal (staged through the host)",
            "overlap": "expand(k-1) and generate(k+1) run while chunk k is gathered (double buffer)",
            "expanded_own_shard_equals_direct_output": own_ok, "own_shard_signatures_compared": oc,
            "expanded_digests_identical_on_all_ranks": same}


def regenerate_leg(args, plan, eng, dev, cdev, world, rank, logn, batch, chunk, L, d_in, d_wit, d_inst, stream):
    """Third curve, the recompute-instead-of-communicate form of 'every GPU holds every witness': all-gather the 6 KB
    INPUTS of a chunk and let every GPU run the witness kernel over all `world` shards itself.  The fabric carries
    0.1 % of the bytes; each GPU writes world x 5 MB per signature of its shard size, so like the compact curve this is
    bounded by one GPU's HBM write rate for the whole node -- with the generator's rate instead of the expander's."""
    d_sig, d_pk, d_hm = d_in
    n = L.n
    gc, nk = plan["gather_chunk_per_rank"], plan["gather_chunks"]
    loc = [torch.empty((3, gc, n), dtype=torch.int16, device=dev) for _ in range(2)]
    gathered = [torch.empty((world, 3, gc, n), dtype=torch.int16, device=dev) for _ in range(2)]
    st = torch.empty(world * gc, dtype=torch.int32, device=dev)
    nccl = args.backend == "nccl"

    def generate(g):
        for r in range(world):
            eng.witness_ntt_verify_dev(logn, gc, g[r, 0], g[r, 1], g[r, 2], d_wit[r * gc:(r + 1) * gc], d_inst[r * gc:(r + 1) * gc],
                                       st[r * gc:(r + 1) * gc], frw.ENC_MONTGOMERY, stream.cuda_stream)

    def run(k_count):
        works = [None, None]
        for k in range(k_count):
            b = k & 1
            a = k * gc
            loc[b][0].copy_(d_sig[a:a + gc])          # the gather that read loc[b] two iterations ago has been waited for
            loc[b][1].copy_(d_pk[a:a + gc])
            loc[b][2].copy_(d_hm[a:a + gc])
            works[b] = sharding.all_gather_chunks(loc[b], gathered[b], async_op=nccl)
            if k >= 1:
                if works[1 - b] is not None:
                    works[1 - b].wait()
                generate(gathered[1 - b])
        b = (k_count - 1) & 1
        if works[b] is not None:
            works[b].wait()
        generate(gathered[b])

    run(min(2, nk))
    torch.cuda.synchronize()
    sharding.barrier()
    t0 = time.perf_counter()
    run(nk)
    torch.cuda.synchronize()
    sharding.barrier()
    t = sharding.max_over_ranks(time.perf_counter() - t0, cdev)
    ok = int((st != 0).sum().item()) == 0
    return {"signatures_per_s_node": round(world * nk * gc / t, 1), "seconds": round(t, 4), "chunk_per_rank": gc, "chunks": nk,
            "wire_format": "the (sig, pk, hm) coefficient vectors, %d bytes per signature" % (6 * n),
            "witness_bytes_written_GBs_per_gpu": round(world * nk * gc * 32 * (L.num_witness + L.num_instance) / t / 1e9, 1),
            "all_statuses_ok": ok}


def arkworks_gather_probe(args, plan, eng, dev, cdev, world, rank, logn, L, d_in, d_wit, d_inst, d_st, stream, chunk):
    """Bounded probe of the naive form (32-byte elements over the wire), for the xGMI ceiling it runs into."""
    d_sig, d_pk, d_hm = d_in
    gc, iters = plan["probe_chunk_per_rank"], 6
    loc = [d_wit[:gc], d_wit[gc:2 * gc]]
    gathered = torch.empty((world,) + tuple(loc[0].shape), dtype=torch.int64, device=dev)
    works = [None, None]
    torch.cuda.synchronize()
    sharding.barrier()
    tg = time.perf_counter()
    for it in range(iters):
        b = it & 1
        if works[b] is not None:
            works[b].wait()
        a = (it * gc) % max(1, d_sig.shape[0] - gc + 1)
        eng.witness_ntt_verify_dev(logn, gc, d_sig[a:a + gc], d_pk[a:a + gc], d_hm[a:a + gc], loc[b], d_inst, d_st[a:a + gc],
                                   frw.ENC_MONTGOMERY, stream.cuda_stream)
        works[b] = sharding.all_gather_chunks(loc[b], gathered, async_op=(args.backend == "nccl"))
    for w_ in works:
        if w_ is not None:
            w_.wait()
    torch.cuda.synchronize()
    sharding.barrier()
    tg = sharding.max_over_ranks(time.perf_counter() - tg, cdev)
    return {"signatures_per_s_node": round(world * gc * iters / tg, 1), "chunk_per_rank": gc, "iterations": iters,
            "fabric_ingest_GBs_per_gpu": round((world - 1) * gc * L.num_witness * 32 * iters / tg / 1e9, 1)}


def main():
    quiet_stdout()
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--logn", type=int, default=10, choices=[9, 10])
    ap.add_argument("--batch", type=int, default=0,
                    help="signatures per GPU per step (default: 65,536 = BASELINE configs[2], for every N: one weak-scaling "
                         "series; at N = 8 two steps are configs[3]'s 1 M signatures over the node)")
    ap.add_argument("--plan", action="store_true",
                    help="touch no GPU: print, for N = 1, 2, 4, 8 (or --gpus N alone if given), every rank's global index "
                         "range, launches, gather chunking and HBM bytes of the run these arguments describe, assert that "
                         "each fits 0.9 x 288 GB, and exit")
    ap.add_argument("--chunk", type=int, default=0,
                    help="signatures per kernel launch = size of the reused HBM witness buffer.  Default: 32,768 Falcon-1024 "
                         "witnesses = 164 GB of the 288 GB (a step = two launches): sized for the HBM, and measured -- the "
                         "write stream itself sustains 6.5 TB/s over a buffer of this size, 6.0 TB/s over 82 GB "
                         "(profiles/r02_chunk_sizes.txt)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-r1cs-check", action="store_true",
                    help="skip the untimed on-device check that every witness of the last timed launch satisfies the "
                         "independently emitted constraint system (builds the matrices on the host, ~6 s)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N = 1 only: skip the untimed secondary rooflines (NTT + mod_q kernel at BASELINE configs[1], "
                         "Falcon-512 full verify)")
    ap.add_argument("--no-aggregate", action="store_true",
                    help="N = 1 only: leave the 1,024-signature mixed aggregate (BASELINE configs[4] shape) and the one-signature host "
                         "calls out of `secondary` "
                         "(the profiling passes do: its small launches of the same kernels would blur the per-kernel averages)")
    ap.add_argument("--no-allgather", action="store_true",
                    help="N > 1 only: skip the second curve, 'generate + RCCL all-gather of the witness vectors'")
    ap.add_argument("--no-prove", action="store_true",
                    help="N > 1 only: skip the proof leg (every rank proves 64 of its own resident witnesses: Groth16 proofs per "
                         "second over all GPUs, sharded by signature like the witnesses)")
    ap.add_argument("--aggregate-leg", type=int, default=16, metavar="K",
                    help="N > 1 only (with the proof leg): every rank also makes ONE proof for an aggregate of K Falcon-1024 statements "
                         "(BASELINE configs[4] on the node: 1,024 signatures = 64 aggregates of 16, eight per GPU); 0 = skip")
    ap.add_argument("--aggregate-sharded", type=int, default=1024, metavar="K",
                    help="BASELINE configs[4] as written: ONE proof for K mixed statements (default 1,024: the 2^27 domain) -- N = 1: the last "
                         "`secondary` line; N > 1: the key in N slices, one per rank (scaling_curves.aggregate_proof_sharded); 0: skip")
    ap.add_argument("--allgather-chunk", type=int, default=0, help="signatures per rank per all-gather (default 4,096)")
    ap.add_argument("--allgather-deadline", type=int, default=240,
                    help="seconds the N > 1 gather legs may take before the run reports `value` without them")
    ap.add_argument("--inject-leg-failure", type=int, default=-1, metavar="RANK",
                    help="testing only: the N > 1 gather leg raises on this rank before its first collective (the run must "
                         "print its primary line, say why on stderr and exit 3 through the launcher)")
    ap.add_argument("--force-pg", action="store_true",
                    help="initialise the process group and run the N > 1 legs even with one rank (rehearses the RCCL "
                         "calls on a single GPU)")
    ap.add_argument("--circuit", default="ntt", choices=["ntt", "dual"],
                    help="ntt = FalconNTTVerificationCircuit (default, the BASELINE metric); dual = the signed-split "
                         "FalconDualNTTVerificationCircuit (SURVEY 8-f row 2; not the headline metric)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="process-group backend for N > 1: nccl (= RCCL over xGMI, the default) or gloo (rehearsal of "
                         "the N > 1 code path with several ranks sharing one GPU)")
    ap.add_argument("--aggregate", default="10x16", help="--workload aggregate: the statement as <logn>x<count>[,<logn>x<count>...] in order "
                                                         "(10x16 = sixteen Falcon-1024 statements; 10x1,9x1,10x2 = a mixed four)")
    ap.add_argument("--workload", default="verify", choices=["verify", "ntt_modq", "prepare", "qap", "aggregate"],
                    help="verify = full verify-with-ntt witness (default, BASELINE configs[2]); ntt_modq = the "
                         "NTT + mod_q witness kernel alone (BASELINE configs[1]: --logn 9 --batch 4096)")
    ap.add_argument("--dump-digests", default="",
                    help="after the run, write this rank's per-signature (global index, status, witness digest) of one "
                         "extra, untimed pass to PATH.rank<r>.npy (multi-rank consistency tests)")
    args = ap.parse_args()
    if args.plan:
        return print_plans(args, ap.get_default("gpus") != args.gpus)

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d: launch with torch.distributed.run --nproc-per-node %d"
                         % (args.gpus, world, args.gpus))
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("no HIP device visible: this benchmark has no CPU path")
    # One rank per GPU.  If the launcher narrows visibility to one device per rank, that device is index 0; the gloo
    # rehearsal deliberately lets ranks share a GPU.
    dev_index = local_rank % ndev
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    use_pg = world > 1 or args.force_pg
    if use_pg:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group("gloo")

    if args.workload == "ntt_modq":
        if not args.batch:
            args.batch = 4096
        return bench_ntt_modq(args, world, rank, dev)
    if args.workload == "prepare":
        if not args.batch:
            args.batch = 65536
        return bench_prepare(args, world, rank, dev)
    if args.workload == "qap":
        if not args.batch:
            args.batch = 256
        return bench_qap(args, world, rank, dev)
    if args.workload == "aggregate":
        logns = [int(part.split("x")[0]) for part in args.aggregate.split(",") for _ in range(int(part.split("x")[1]))]
        eng = frw.WitnessEngine(dev_index)
        r = time_aggregate_proof(eng, dev, logns, max(1, args.steps), None)
        return emit({"metric": "aggregate_proof_signatures_per_sec", "value": r["signatures_per_s"], "unit": "signatures/s", "n_gpus": world,
                     "steps": args.steps, "warmup": args.warmup, "ms_per_step": r["ms_per_proof"], "higher_is_better": True,
                     "scaling": "weak", "vs_baseline": None, "dtype": "u32", "data": "synthetic",
                     "config": {"workload": r["workload"]}, "aggregate": r})
    logn = args.logn
    dual = args.circuit == "dual"
    L = frw.layout_dual(logn) if dual else frw.layout(logn)
    # the whole shape of the run -- index range, launches, gather chunking, HBM bytes -- comes from one piece of plain
    # arithmetic that the CPU tests evaluate for world = 8 as well (falcon-r1cs_amd/sharding.py::step_plan)
    plan = make_plan(args, world, rank)
    assert plan["fits"], "this run plans %.1f GB of HBM per GPU (limit %.1f GB): reduce --chunk" % (
        plan["hbm_plan_bytes"] / 1e9, plan["hbm_limit_bytes"] / 1e9)
    batch, chunk = plan["batch_per_gpu"], plan["chunk"]
    eng = frw.WitnessEngine(dev_index)
    launch = eng.witness_dual_ntt_verify_dev if dual else eng.witness_ntt_verify_dev
    threads = max(1, (os.cpu_count() or 1) // world)

    # ---- inputs resident in HBM -------------------------------------------------------------
    lo, hi = plan["global_lo"], plan["global_hi"]                  # this rank's global signature indices
    sig, pk, hm = synth(logn, hi - lo, lo, threads)
    d_sig, d_pk, d_hm = (torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm))
    d_wit = torch.empty((chunk, L.num_witness, 4), dtype=torch.int64, device=dev)
    d_inst = torch.empty((chunk, L.num_instance, 4), dtype=torch.int64, device=dev)
    d_st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    stream = torch.cuda.current_stream()
    nchunks = plan["launches_per_step"]
    n = L.n

    def step(events=None):
        for c in range(nchunks):
            a = c * chunk
            cnt = min(chunk, batch - a)
            if events is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(stream)
            launch(logn, cnt, d_sig[a:a + cnt], d_pk[a:a + cnt], d_hm[a:a + cnt], d_wit, d_inst,
                   d_st[a:a + cnt], frw.ENC_MONTGOMERY, stream.cuda_stream)
            if events is not None:
                e1.record(stream)
                events.append((e0, e1, cnt))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    events = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step(events)
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    cdev = dev if args.backend == "nccl" else torch.device("cpu")      # where control-plane tensors live
    elapsed = sharding.max_over_ranks(time.perf_counter() - t0, cdev)

    # ---- control-plane exchange (untimed): global status vector ---------------------------------
    status = sharding.gather_per_signature(d_st.to(cdev), batch * world, rank, world)
    n_bad = int((status != 0).sum().item())
    assert n_bad == 0, "%d of the synthetic signatures failed their range checks: the throughput would not be that of " \
                       "valid witnesses" % n_bad

    # ---- what the timed launches left in HBM: digest all of it, check all of it (untimed) ----------------------------
    # The buffer is reused by every launch of a step, so after the timed region slots [0, last_cnt) hold the witnesses of
    # the LAST timed launch and, when that one was the step's ragged remainder, slots [last_cnt, chunk) still hold what
    # the launch before it (a full-size one) wrote.  slot_sig = rank-local signature index behind every slot.
    last_a = (nchunks - 1) * chunk
    last_cnt = batch - last_a
    held = chunk if nchunks >= 2 else last_cnt
    slot_sig = np.arange(held, dtype=np.int64) + last_a
    if held > last_cnt:
        slot_sig[last_cnt:] = np.arange(last_cnt, held) + (nchunks - 2) * chunk
    d_dig = torch.zeros(held, dtype=torch.int64, device=dev)
    eng.digest_dev(d_wit, L.num_witness * 4, held, d_dig, stream.cuda_stream)
    torch.cuda.synchronize()
    held_dig = d_dig.cpu().numpy().view(np.uint64)
    checked = {"signatures": held, "from_the_last_timed_launch": last_cnt,
               "from_the_full_size_launch_before_it": held - last_cnt, "distinct_digests": int(len(np.unique(held_dig)))}
    if not dual:
        for name, cnt in (("full_size_launch", chunk), ("remainder_launch", last_cnt if last_cnt != chunk else 0)):
            if cnt:
                sh = eng.launch_shape(logn, cnt)
                want = frw.MI355X_BENCH_LAUNCH_SHAPES.get((logn, cnt))
                if want is not None and sh["cus"] == 256:
                    # the shape tests/test_gpu_parity.py::test_benchmark_launch_shape_is_checked covers, not a look-alike
                    assert (sh["grid"], sh["split_signatures"]) == want, (sh, want)
                checked[name] = {"signatures": cnt, "grid": sh["grid"], "resident_workgroups_per_cu": sh["resident_per_cu"],
                                 "rounds": round(cnt / max(1, sh["grid"]), 2), "split_signatures": sh["split_signatures"]}
    r1cs = None
    qap_result = msm_result = groth16_result = prove_info = None
    if not args.no_r1cs_check:
        # the reference's assert!(cs.is_satisfied()) (falcon_ntt.rs:159) for every witness in the buffer, on the device,
        # in place, against matrices emitted from the gadget definitions by the host mirror (not the kernels' closed form)
        h = eng.r1cs_load(1 if dual else 0, logn)
        badrows = torch.zeros(held, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        eng.r1cs_check_dev(h, held, d_wit, d_inst, badrows, stream.cuda_stream)
        torch.cuda.synchronize()
        tc = time.perf_counter() - tc
        if world == 1 and not args.no_secondary and not dual:
            s0 = int(slot_sig[0])
            qap_result, d_hvec = time_qap(eng, h, dev, d_wit, d_inst, min(64, held), 20, logn,
                                          None if args.no_cpu_baseline else (sig[s0:s0 + 1], pk[s0:s0 + 1], hm[s0:s0 + 1]))
            msm_result = time_msm(eng, dev, d_hvec, 10, not args.no_cpu_baseline)
            del d_hvec
            groth16_result = time_groth16(eng, h, dev, d_wit, d_inst, min(64, held), 10, L, logn)
        if use_pg and not dual and not args.no_prove:
            prove_info = prove_leg(eng, h, dev, cdev, d_wit, d_inst, held, L, logn, world, rank)
            torch.cuda.empty_cache()             # the leg's workspace goes back to the device before the gather legs allocate
            if args.aggregate_leg and logn == 10:
                prove_info["aggregate"] = aggregate_leg(eng, dev, cdev, args.aggregate_leg, world, rank)
                torch.cuda.empty_cache()
        eng.r1cs_free(h)
        n_unsat = sharding.sum_over_ranks(int((badrows != 0).sum().item()), cdev)
        assert n_unsat == 0, "%d witnesses left by the timed launches violate the constraint system" % n_unsat
        r1cs = {"witnesses_checked": held * world, "per_rank": held, "unsatisfied": n_unsat,
                "constraints_each": L.num_constraints, "seconds": round(tc, 3), "buffer": "as left by the timed launches"}

    # ---- roofline of the dominant kernel, from HIP events on the launch stream ------------------
    full = [(e0.elapsed_time(e1), cnt) for e0, e1, cnt in events if cnt == chunk]
    launch_ms = sum(t for t, _ in full) / max(1, len(full))
    bytes_per_sig = 32 * (L.num_witness + 2 * n) + 3 * 2 * n          # 5,086,848 B for Falcon-1024
    achieved = chunk * bytes_per_sig / (launch_ms * 1e-3) / 1e9

    # ---- calibration (untimed): what a compute-free write stream of the same shape SUSTAINS on this device now -----
    # Same regime as the timed region: one step's worth of launches back to back, second of two rounds (a single
    # launch squeezed between witness kernels reads 5-10 % high: the device boosts after every change of load).
    # Runs after the checks above: it overwrites the witness buffer.
    wbytes = d_wit.numel() * 8
    cal_ms = 0.0
    cal_reps = max(2, nchunks)
    for _ in range(2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(cal_reps):
            eng.diag_write_stream_dev(d_wit, wbytes, L.num_witness * 32, stream.cuda_stream)
        e1.record(stream)
        torch.cuda.synchronize()
        cal_ms = e0.elapsed_time(e1) / cal_reps
    write_stream_gbs = wbytes / (cal_ms * 1e-3) / 1e9

    ranks_seen = dist.get_world_size() if use_pg else 1
    devs = sharding.gather_per_signature(torch.tensor([dev_index], dtype=torch.int64, device=cdev), world, rank, world)

    # ---- second curve (N > 1): generate + all-gather of the witness vectors, full step ------------------------------
    # Every collective the primary result needs has happened by now.  The legs run in a worker thread with a deadline:
    # should a collective of theirs ever hang (they cannot be rehearsed on more than one RCCL rank here), rank 0 still
    # emits `value` and every rank leaves through os._exit instead of waiting for peers.
    gather_info, legs_hung = None, False
    if use_pg and not args.no_allgather and not dual:
        import threading
        box = {}

        def legs():
            torch.cuda.set_device(dev_index)
            try:
                if args.inject_leg_failure >= 0:
                    # every rank skips the leg (no collective is left half-entered); the named rank reports the failure
                    if rank == args.inject_leg_failure:
                        raise RuntimeError("injected gather-leg failure on rank %d" % rank)
                    box["r"] = {"skipped": "--inject-leg-failure"}
                    return
                box["r"] = gather_leg(args, plan, eng, dev, cdev, world, rank, logn, batch, chunk, L, (d_sig, d_pk, d_hm), d_wit,
                                      d_st, stream)
                box["r"]["allgather_inputs_and_regenerate"] = regenerate_leg(
                    args, plan, eng, dev, cdev, world, rank, logn, batch, chunk, L, (d_sig, d_pk, d_hm), d_wit, d_inst, stream)
                box["r"]["naive_32_byte_elements_probe"] = arkworks_gather_probe(
                    args, plan, eng, dev, cdev, world, rank, logn, L, (d_sig, d_pk, d_hm), d_wit, d_inst, d_st, stream, chunk)
            except Exception as ex:      # the primary metric must not depend on these legs
                import traceback
                box["e"] = repr(ex)[:300]
                sys.stderr.write("bench.py rank %d: a gather leg raised %r\n%s" % (rank, ex, traceback.format_exc()))
                sys.stderr.flush()
        th = threading.Thread(target=legs, daemon=True)
        th.start()
        th.join(args.allgather_deadline)
        legs_hung = th.is_alive()
        gather_info = dict(box.get("r") or {})
        if legs_hung:
            gather_info["error"] = "no result within %d s (a collective did not complete)" % args.allgather_deadline
        elif "e" in box:
            gather_info["error"] = box["e"]

    if args.dump_digests:
        # one extra, untimed pass with every launch digested: per-signature (global index, status, digest) of this rank
        dd = torch.zeros(batch, dtype=torch.int64, device=dev)
        for c in range(nchunks):
            a = c * chunk
            cnt = min(chunk, batch - a)
            launch(logn, cnt, d_sig[a:a + cnt], d_pk[a:a + cnt], d_hm[a:a + cnt], d_wit, d_inst, d_st[a:a + cnt],
                   frw.ENC_MONTGOMERY, stream.cuda_stream)
            eng.digest_dev(d_wit, L.num_witness * 4, cnt, dd[a:a + cnt], stream.cuda_stream)
        torch.cuda.synchronize()
        np.save("%s.rank%d.npy" % (args.dump_digests, rank),
                np.stack([np.arange(lo, hi, dtype=np.uint64), d_st.cpu().numpy().astype(np.uint64),
                          dd.cpu().numpy().view(np.uint64)]))

    # ---- third curve (N > 1): BASELINE configs[4] as written -- ONE proof for the 1,024 mixed statements, its key in N slices ----------
    # The witness buffer has served its purpose (digests, checks, gather legs): its 164 GB go back first -- the leg keeps the transform tables
    # of the 2^27 domain, this rank's slices of the key and one proof's workspace (sharding.sharded_aggregate_plan).  One collective on its
    # data path (an all-gather of 576 bytes per rank), so it runs under a deadline like the gather legs.
    sharded_info = None
    if use_pg and not dual and args.aggregate_sharded and logn == 10 and not legs_hung:
        import threading
        mix = sharding.aggregate_mix(args.aggregate_sharded)
        splan = sharding.sharded_aggregate_plan(world, rank, mix)
        assert splan["fits"], "the sharded aggregate leg plans %.1f GB of HBM per GPU" % (splan["hbm_plan_bytes"] / 1e9)
        d_wit = d_inst = None
        torch.cuda.empty_cache()
        sbox = {}

        def sharded_leg():
            torch.cuda.set_device(dev_index)
            try:
                sbox["r"] = time_aggregate_proof(eng, dev, mix, 3, None, check_h=False, world=world, rank=rank, cdev=cdev)
            except Exception as ex:      # noqa: BLE001 -- the primary metric must not depend on this leg
                import traceback
                sbox["e"] = repr(ex)[:300]
                sys.stderr.write("bench.py rank %d: the sharded aggregate leg raised %r\n%s" % (rank, ex, traceback.format_exc()))
                sys.stderr.flush()
        th = threading.Thread(target=sharded_leg, daemon=True)
        th.start()
        th.join(args.allgather_deadline)
        sharded_info = dict(sbox.get("r") or {})
        sharded_info["hbm_plan_bytes_leg"] = splan["hbm_plan_bytes"]
        if th.is_alive():
            legs_hung = True
            sharded_info["error"] = "no result within %d s (a collective did not complete)" % args.allgather_deadline
        elif "e" in sbox:
            sharded_info["error"] = sbox["e"]

    result = None
    traffic = None if dual else measured_traffic(logn, chunk)
    if rank == 0:
        value = world * batch * args.steps / elapsed
        cfg_name = "BASELINE configs[2] on every GPU" + (
            "; two steps = BASELINE configs[3], 1 M signatures sharded over 8 GPUs" if world == 8 and batch == 65536 else "")
        result = {
            "metric": "falcon%d_verify_with_%sntt_r1cs_witnesses_per_sec" % (n, "dual_" if dual else ""), "value": round(value, 1),
            "unit": "signatures/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "u32", "data": "synthetic",
            "config": {"workload": "falcon-%d full verify-with-%sntt witness (NTT ladder + mod_q + pointwise + l2-norm), "
                                   "batch=%d signatures per GPU per step (%s)" % (n, "dual-" if dual else "", batch, cfg_name),
                       "logn": logn, "batch_per_gpu": batch, "signatures_per_step_all_gpus": batch * world,
                       "chunk": chunk, "launches_per_step": nchunks,
                       "encoding": "bls12-381-fr montgomery (arkworks witness_assignment bytes)",
                       "seed": hex(SEED), "sharding": "by signature index, no data-path collective",
                       "signatures_failing_range_checks": n_bad,
                       "ranks_seen": ranks_seen, "device_index_per_rank": [int(x) for x in devs.tolist()],
                       "backend": (args.backend if use_pg else None),
                       "hbm_plan_bytes": plan["hbm_plan_bytes"], "hbm_plan_limit_bytes": plan["hbm_limit_bytes"],
                       "hbm_peak_allocated_bytes": int(torch.cuda.max_memory_allocated(dev)),
                       "global_index_range_rank0": [plan["global_lo"], plan["global_hi"]]},
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic[0] if traffic else None,
                         "traffic_from_committed_profile": traffic[1] if traffic else None,
                         "kernel": "witness_%sntt_verify_kernel<%d,1>" % ("dual_" if dual else "", logn),
                         "kernel_source_sha256_16": kernel_source_sha(),
                         "algorithmic_bytes_per_launch": chunk * bytes_per_sig,
                         "avg_launch_ms": round(launch_ms, 4), "launches_timed": len(full),
                         "device_write_stream_GBs": round(write_stream_gbs, 1),
                         "frac_of_device_write_stream": round(achieved / write_stream_gbs, 4)},
            "launch_shape_checked": checked,
        }
        if r1cs is not None:
            result["r1cs_check"] = r1cs
        if gather_info is not None or prove_info is not None or sharded_info is not None:
            result["scaling_curves"] = {"generate_only_signatures_per_s": round(value, 1)}
            if gather_info is not None:
                result["scaling_curves"]["generate_plus_allgather"] = gather_info
            if prove_info is not None:
                result["scaling_curves"]["prove"] = prove_info
                result["scaling_curves"]["prove_proofs_per_s"] = prove_info.get("proofs_per_s_all_gpus")
            if sharded_info is not None:
                result["scaling_curves"]["aggregate_proof_sharded"] = sharded_info
        if world == 1 and not args.no_secondary and not dual:
            # untimed w.r.t. `value`: the other two rooflines BASELINE / north_star name, measured in this process
            result["secondary"] = {
                "ntt_modq_falcon512_batch4096": time_ntt_modq(eng, dev, 9, 4096, 200, 20, three_streams=not args.no_aggregate),
                "verify_falcon512_8192_per_launch": time_verify(eng, dev, 9, 8192, 12, 3, threads),
                "compact_encoding_falcon%d" % n: time_compact(eng, dev, logn, chunk, 4, 1, (d_sig, d_pk, d_hm), d_wit)}
            if qap_result is not None:
                result["secondary"]["qap_witness_map_falcon%d" % n] = qap_result
                result["secondary"]["groth16_msm_h_falcon%d" % n] = msm_result
                result["secondary"]["groth16_prove_falcon%d" % n] = groth16_result
            if not args.no_aggregate and logn == 10:
                torch.cuda.empty_cache()
                result["secondary"]["aggregate_proof_4_mixed"] = time_aggregate_proof(eng, dev, (10, 9, 10, 10), 10, None)
                torch.cuda.empty_cache()
                result["secondary"]["aggregate_proof_16_falcon1024"] = time_aggregate_proof(eng, dev, (10,) * 16, 10, groth16_result)
                torch.cuda.empty_cache()
            if not args.no_aggregate:
                result["secondary"]["host_call_one_signature_falcon%d" % n] = time_host_call(eng, logn)
                result["secondary"]["aggregate_1024_mixed"] = time_aggregate(eng, dev, 1024, 50, threads)
                result["secondary"]["input_preparation_falcon%d" % n] = time_prepare(eng, dev, logn, 65536, 5, 1)
            if not args.no_aggregate and logn == 10 and args.aggregate_sharded:
                # BASELINE configs[4] as written: ONE proof for the 1,024 mixed statements.  Last: the witness buffer (164 GB) goes back to
                # the device first -- the proof keeps 208 GB of its own (sharding.sharded_aggregate_plan).
                d_wit = d_inst = None
                torch.cuda.empty_cache()
                if args.aggregate_sharded == 1024:
                    # ... and the same 1,024 signatures as 64 proofs of 16 (three keys of window tables: 130 GB)
                    result["secondary"]["configs4_as_64_proofs_of_16"] = time_configs4_as_aggregates_of_16(eng, dev)
                    torch.cuda.empty_cache()
                mix = sharding.aggregate_mix(args.aggregate_sharded)
                assert sharding.sharded_aggregate_plan(1, 0, mix)["fits"]
                result["secondary"]["aggregate_proof_%d_mixed" % len(mix)] = time_aggregate_proof(eng, dev, mix, 3, None)
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline and not dual:
            slots = np.arange(0, held, max(1, held // 4096))[:4096]        # strided over the whole buffer
            digest_of = {int(slot_sig[j]): int(held_dig[j]) for j in slots}
            result["cpu_baseline"] = cpu_baseline(logn, sig, pk, hm, slot_sig[slots], lambda i: digest_of[int(i)])
        emit(result)
    if use_pg:
        leave(legs_hung, bool((gather_info and "error" in gather_info) or (prove_info and ("error" in prove_info or "error" in prove_info.get("aggregate", {})))
                               or (sharded_info and "error" in sharded_info)),
              rank)


def leave(legs_hung, legs_failed, rank, teardown_timeout=10.0):
    """End of a run with a process group.  A gather leg that hung (a collective never completed) or raised is a FAILED run:
    the primary line has been printed, and the process exits non-zero so that no harness records it as healthy.  Otherwise
    the group is torn down -- under a timer, because a teardown that waits for a peer must not outlive the run -- and only
    a teardown that does not return falls back to os._exit, saying so on stderr.  Nothing here re-executes anything."""
    import threading
    sys.stdout.flush()
    if legs_hung:
        # peers may be stuck inside the same collective: no further rendezvous, leave at once
        sys.stderr.write("bench.py rank %d: a gather leg did not complete within its deadline -> exit 3\n" % rank)
        sys.stderr.flush()
        os._exit(3)
    done = {}

    def teardown():
        try:
            dist.destroy_process_group()
            done["ok"] = True
        except Exception as ex:                      # noqa: BLE001 -- reported below
            done["error"] = repr(ex)[:300]
    th = threading.Thread(target=teardown, daemon=True)
    th.start()
    th.join(teardown_timeout)
    rc = 3 if legs_failed else 0
    if th.is_alive() or "error" in done:
        sys.stderr.write("bench.py rank %d: destroy_process_group %s\n" % (
            rank, "did not return within %.0f s" % teardown_timeout if th.is_alive() else "raised " + done["error"]))
        sys.stderr.flush()
        os._exit(rc or 4)
    if rc:
        sys.stderr.write("bench.py rank %d: a gather leg raised -> exit %d\n" % (rank, rc))
        sys.stderr.flush()
        sys.exit(rc)


def make_plan(args, world, rank):
    dual = args.circuit == "dual"
    L = frw.layout_dual(args.logn) if dual else frw.layout(args.logn)
    CL = frw.compact_layout(args.logn)
    batch = args.batch or 65536
    chunk = min(args.chunk or (32768 if args.logn == 10 else 65536), batch)
    legs = (world > 1 or args.force_pg) and not args.no_allgather and not dual
    prove = (world > 1 or args.force_pg) and not args.no_prove and not args.no_r1cs_check and not dual
    agg = args.aggregate_leg if prove and args.logn == 10 else 0
    return sharding.step_plan(world, rank, batch, chunk, args.allgather_chunk, L.n, L.num_witness, L.num_instance,
                              int(CL.bytes_per_signature), with_gather_legs=legs, with_prove_leg=prove, aggregate_statements=agg)


def print_plans(args, only_this_world):
    """--plan: no GPU, no process group.  One JSON object on stdout; AssertionError (non-zero exit) if a plan does not fit."""
    out = {}
    for world in ([args.gpus] if only_this_world else [1, 2, 4, 8]):
        plans = [make_plan(args, world, r) for r in range(world)]
        sharding.check_plans(plans)
        p0 = plans[0]
        out["gpus_%d" % world] = {
            "signatures_per_step_all_gpus": p0["signatures_per_step_all_gpus"], "batch_per_gpu": p0["batch_per_gpu"],
            "chunk": p0["chunk"], "launches_per_step": p0["launches_per_step"],
            "global_index_range_per_rank": [[p["global_lo"], p["global_hi"]] for p in plans],
            "gather": {k: p0[k] for k in ("gather_chunk_per_rank", "gather_chunks", "gathered_signatures_per_collective",
                                          "own_shard_checked_in_place", "all_digests_gathered") if k in p0},
            "hbm_plan_bytes_per_rank": p0["hbm_plan_bytes"], "hbm_plan_GB_per_rank": round(p0["hbm_plan_bytes"] / 1e9, 2),
            "hbm_limit_bytes": p0["hbm_limit_bytes"], "fits": p0["fits"],
            "buffers_GB": {k: round(v / 1e9, 3) for k, v in p0["buffers"].items()}}
        if args.aggregate_sharded and args.logn == 10 and args.circuit != "dual":
            # the last leg, after the witness buffer has gone back to the device: ONE proof for K mixed statements, the key in `world` slices
            sp = [sharding.sharded_aggregate_plan(world, r, sharding.aggregate_mix(args.aggregate_sharded)) for r in range(world)]
            sharding.check_sharded_aggregate_plans(sp)
            out["gpus_%d" % world]["aggregate_proof_sharded (held alone, the witness buffer released)"] = {
                "statements": sp[0]["statements"], "log_domain_size": sp[0]["log_domain_size"],
                "rows_of_the_witness_side_tables_per_rank": [[p["z_lo"], p["z_hi"]] for p in sp],
                "rows_of_h_query_per_rank": [[p["h_lo"], p["h_hi"]] for p in sp],
                "hbm_plan_GB_per_rank": [round(p["hbm_plan_bytes"] / 1e9, 2) for p in sp],
                "buffers_GB_rank0": {k: round(v / 1e9, 3) for k, v in sp[0]["buffers"].items()}}
    emit(out)


if __name__ == "__main__":
    main()
