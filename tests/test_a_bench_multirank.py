"""N > 1 path of bench.py on real hardware, as far as one GPU allows: two ranks (gloo process group) sharing the card.

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
COMMON = ["--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"]


def _run(cmd, timeout=600):
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", MASTER_ADDR="127.0.0.1")
    out = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=timeout, env=env)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


@pytest.mark.gpu
def test_two_ranks_reproduce_the_one_rank_run(tmp_path):
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    j1 = _run([sys.executable, "bench.py", "--gpus", "1", "--batch", "512", "--chunk", "256", "--dump-digests", one] + COMMON)
    j2 = _run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", "29617", "bench.py", "--gpus", "2", "--backend", "gloo", "--batch", "256", "--chunk", "128",
               "--dump-digests", two] + COMMON)
    a = np.load(one + ".rank0.npy")
    b = np.concatenate([np.load(two + ".rank0.npy"), np.load(two + ".rank1.npy")], axis=1)
    assert a.shape == b.shape == (3, 512)
    assert np.array_equal(a[0], np.arange(512)) and np.array_equal(a, b)          # indices, statuses, digests
    assert not a[1].any()
    assert j2["n_gpus"] == 2 and j2["config"]["ranks_seen"] == 2 and j2["config"]["signatures_per_step_all_gpus"] == 512
    assert j2["r1cs_check"]["witnesses_checked"] == 2 * 128 and j2["r1cs_check"]["unsatisfied"] == 0
    g = j2["scaling_curves"]["generate_plus_allgather"]
    assert "error" not in g, g
    assert g["expanded_own_shard_equals_direct_output"] and g["expanded_digests_identical_on_all_ranks"]
    assert g["signatures_per_rank"] == 256
    assert g["allgather_inputs_and_regenerate"]["all_statuses_ok"] and g["naive_32_byte_elements_probe"]["signatures_per_s_node"] > 0
    assert j1["launch_shape_checked"]["signatures"] == 256 and j1["r1cs_check"]["witnesses_checked"] == 256
    assert j1["launch_shape_checked"]["from_the_last_timed_launch"] == 256
