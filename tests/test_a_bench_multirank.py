"""N > 1 path of bench.py on real hardware, as far as one GPU allows: two and four ranks (gloo process group) sharing the
card.  (Eight cannot: the pool allows at most 6 processes on a card; the world-8 plan runs on CPU tensors in
tests/test_sharding_gloo.py.)

Runs first in the -m gpu suite (file name), i.e. before this process has touched the GPU itself: everything happens in
child processes.  The two ranks shard the global signature range exactly as RCCL ranks would (falcon-r1cs_amd/sharding.py),
so their per-signature digests must equal those of a one-rank run over the same global indices, and the second curve
(compact all-gather + local expansion) must reproduce the direct output on every rank.
"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--aggregate-sharded", "0", "--aggregate-leg", "0"]


def _run(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_and_four_ranks_reproduce_the_one_rank_run(tmp_path):
    one = str(tmp_path / "one")
    j1 = _run([sys.executable, "bench.py", "--gpus", "1", "--batch", "512", "--chunk", "256", "--dump-digests", one] + COMMON)
    a = np.load(one + ".rank0.npy")
    assert a.shape == (3, 512) and np.array_equal(a[0], np.arange(512)) and not a[1].any()
    assert j1["launch_shape_checked"]["signatures"] == 256 and j1["r1cs_check"]["witnesses_checked"] == 256
    assert j1["launch_shape_checked"]["from_the_last_timed_launch"] == 256
    assert j1["config"]["hbm_plan_bytes"] <= j1["config"]["hbm_plan_limit_bytes"]
    for world, port in ((2, 29617), (4, 29641)):
        per = 512 // world
        out = str(tmp_path / ("w%d" % world))
        jw = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                   "--master-addr", "127.0.0.1", "--master-port", str(port), "bench.py", "--gpus", str(world), "--backend", "gloo",
                   "--batch", str(per), "--chunk", str(per // 2), "--dump-digests", out]
                  # world 2: own shard checked inside the witness buffer (the default N = 2, 4 shape); world 4: in a buffer
                  # of its own (the default N = 8 shape)
                  + (["--allgather-chunk", "32"] if world == 2 else []) + COMMON)
        b = np.concatenate([np.load("%s.rank%d.npy" % (out, r)) for r in range(world)], axis=1)
        assert np.array_equal(a, b), "world %d: indices / statuses / digests differ from the one-rank run" % world
        assert jw["n_gpus"] == world and jw["config"]["ranks_seen"] == world
        assert jw["config"]["signatures_per_step_all_gpus"] == 512 and jw["config"]["batch_per_gpu"] == per
        assert jw["r1cs_check"]["witnesses_checked"] == world * (per // 2) and jw["r1cs_check"]["unsatisfied"] == 0
        g = jw["scaling_curves"]["generate_plus_allgather"]
        assert "error" not in g, g
        assert g["expanded_own_shard_equals_direct_output"] and g["expanded_digests_identical_on_all_ranks"]
        assert g["signatures_per_rank"] == per and g["chunk_per_rank"] == (32 if world == 2 else (per // 2) // world)
        assert g["own_shard_signatures_compared"] == g["chunk_per_rank"]
        assert g["allgather_inputs_and_regenerate"]["all_statuses_ok"]
        assert g["naive_32_byte_elements_probe"]["signatures_per_s_node"] > 0


def _leave(code, timeout=120):
    """bench.leave() in a child with a one-rank gloo group (no GPU involved): returns the exit code."""
    prog = ("import os, sys, torch.distributed as dist; sys.path.insert(0, %r); import bench; "
            "os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT='29655', RANK='0', WORLD_SIZE='1'); "
            "dist.init_process_group('gloo'); %s" % (ROOT, code))
    return subprocess.run([sys.executable, "-c", prog], cwd=ROOT, capture_output=True, text=True, timeout=timeout)


def test_failed_or_hung_gather_legs_exit_non_zero():
    """a collective that never completed, or a leg that raised, must not be recorded as a healthy run."""
    ok = _leave("bench.leave(False, False, 0)")
    assert ok.returncode == 0, ok.stderr[-2000:]
    failed = _leave("bench.leave(False, True, 0)")
    assert failed.returncode == 3 and "gather leg raised" in failed.stderr
    hung = _leave("bench.leave(True, False, 0)")
    assert hung.returncode == 3 and "did not complete" in hung.stderr
    stuck = _leave("dist.destroy_process_group = lambda: __import__('time').sleep(60); bench.leave(False, False, 0, 1.0)")
    assert stuck.returncode == 4 and "did not return" in stuck.stderr


def _plain_env():
    return {k: v for k, v in dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0").items()
            if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}


@pytest.mark.gpu
def test_plain_invocation_with_gpus_2_starts_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus N ...` in the form the driver records for N = 1 (no launcher, no WORLD_SIZE) must
    run: the parent starts the ranks as child processes before it touches any GPU and forwards rank 0's one line."""
    out = str(tmp_path / "plain")
    common = COMMON[:-2] + ["--aggregate-leg", "2"]                 # one proof for two statements (2^19) per rank
    run = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--batch", "256", "--chunk", "128",
                          "--dump-digests", out] + common, cwd=ROOT, capture_output=True, text=True, timeout=900, env=_plain_env())
    assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, run.stdout
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["ranks_seen"] == 2 and j["config"]["signatures_per_step_all_gpus"] == 512
    g = j["scaling_curves"]["generate_plus_allgather"]
    assert "error" not in g and g["expanded_own_shard_equals_direct_output"] and g["expanded_digests_identical_on_all_ranks"]
    b = np.concatenate([np.load("%s.rank%d.npy" % (out, r)) for r in range(2)], axis=1)
    assert np.array_equal(b[0], np.arange(512)) and not b[1].any()
    # proofs in the N > 1 run: sharded by signature like the witnesses, no collective on the data path
    p = j["scaling_curves"]["prove"]
    assert p["ranks"] == 2 and p["proofs_per_s_all_gpus"] > 0 and p["all_proofs_verified"]
    # ... and BASELINE configs[4] on the "node": every rank makes ONE proof for an aggregate of its own (time_aggregate_proof verifies it)
    a = p["aggregate"]
    assert "error" not in a and a["ranks"] == 2 and a["statements_per_proof"] == 2 and a["signatures_per_s_all_gpus"] > 0
    assert "QAP domain 2^19" in a["rank0"]["workload"]


@pytest.mark.gpu
def test_a_failed_gather_leg_reaches_the_caller_as_exit_code_3():
    run = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--batch", "256", "--chunk", "128",
                          "--inject-leg-failure", "1", "--no-r1cs-check"] + COMMON, cwd=ROOT, capture_output=True, text=True,
                         timeout=600, env=_plain_env())
    assert run.returncode == 3, (run.returncode, run.stderr[-3000:])
    assert "injected gather-leg failure on rank 1" in run.stderr and "Traceback" in run.stderr
    lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and json.loads(lines[0])["value"] > 0           # the primary line is still printed


@pytest.mark.gpu
def test_one_proof_with_its_key_in_slices_over_two_and_four_ranks():
    """BASELINE configs[4]'s shape on the "node", as far as one card allows: ONE proof for a mixed aggregate (the first 4 statements of the
    benchmark's mix, two of each parameter set: the 2^19 domain)
    whose key lies in slices over the ranks -- every rank: the whole witness map and its slices of the five sums
    (frw_groth16_prove_partial_dev), one all-gather of 576 bytes per rank over the process group (gloo here, RCCL on a node),
    frw_groth16_prove_combine_dev on every rank.  The leg itself asserts that all ranks hold the same 384 bytes and that the product's
    pairing verifier accepts them (and rejects them for another statement); here: two and four ranks reproduce the ONE-rank proof
    (the whole key, frw_groth16_prove_dev) byte for byte."""
    common = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary", "--no-allgather", "--no-prove", "--aggregate-leg", "0",
              "--batch", "256", "--chunk", "128", "--backend", "gloo", "--aggregate-sharded", "4"]
    shas = {}
    for world in (1, 2, 4):
        run = subprocess.run([sys.executable, "bench.py", "--gpus", str(world)] + (["--force-pg"] if world == 1 else []) + common,
                             cwd=ROOT, capture_output=True, text=True, timeout=900, env=_plain_env())
        assert run.returncode == 0, run.stdout[-2000:] + run.stderr[-4000:]
        lines = [l for l in run.stdout.splitlines() if l.startswith("{")]
        assert len(lines) == 1, run.stdout
        a = json.loads(lines[0])["scaling_curves"]["aggregate_proof_sharded"]
        assert "error" not in a, a
        assert "QAP domain 2^19" in a["workload"] and a["proving_key"]["slices"] == world and a["signatures_per_s"] > 0
        assert a["proving_key"]["kind"].startswith("bare" if world > 1 else "window")
        lo, hi = a["proving_key"]["rows_this_rank"]["h_query"]
        assert lo == 0 and hi == ((1 << 19) - 1 + world - 1) // world
        shas[world] = a["proof_sha256"]
    assert shas[2] == shas[1] and shas[4] == shas[1], shas
