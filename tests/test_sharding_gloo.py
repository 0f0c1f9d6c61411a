"""The N > 1 path on CPU: world_size-2 gloo process group exercising the same sharding / reduction helpers
bench.py uses over RCCL (falcon-r1cs_amd/sharding.py)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from falcon_r1cs_amd import sharding


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 65536, 1048576 + 3):
        for world in (1, 2, 3, 8):
            ranges = [sharding.shard_range(total, r, world) for r in range(world)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in ranges]
            assert max(sizes) - min(sizes) <= 1


def _worker(rank, world, port, total, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    lo, hi = sharding.shard_range(total, rank, world)
    # per-signature "status" = global index mod 3, "digest" = index * 2654435761
    idx = torch.arange(lo, hi, dtype=torch.int64)
    status = (idx % 3).to(torch.int32)
    digest = idx * 2654435761
    sharding.barrier()
    g_status = sharding.gather_per_signature(status, total, rank, world)
    g_digest = sharding.gather_per_signature(digest, total, rank, world)
    local = torch.full((4, 3), rank + 10, dtype=torch.int64)
    gathered = torch.empty((world, 4, 3), dtype=torch.int64)
    sharding.all_gather_chunks(local, gathered)
    chunks_ok = all(bool((gathered[r] == r + 10).all()) for r in range(world))
    slow = sharding.max_over_ranks(1.0 + rank, dev)
    n = sharding.sum_over_ranks(hi - lo, dev)
    ok = (g_status.tolist() == [i % 3 for i in range(total)]
          and g_digest.tolist() == [i * 2654435761 for i in range(total)]
          and slow == float(world) and n == total and chunks_ok)
    q.put((rank, ok))
    dist.destroy_process_group()


def test_gather_and_reduce_world2_gloo():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    world, total = 2, 37                       # ragged: 19 + 18
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(60)
    assert results == [(0, True), (1, True)]
