"""Multi-GPU partitioning of a batch of signatures (one process per GPU).

Signatures are independent (each reference ``generate_constraints`` call builds its own constraint system,
falcon-r1cs/src/circuits/falcon_ntt.rs:143-151), so the path shards by signature index with no data-path
collective: rank r owns a contiguous block of the global index range and keeps its witnesses in its own HBM.
The only cross-rank traffic is control plane: a barrier around the timed region, a MAX-reduce of the elapsed time
and an all-gather of the per-signature status words / digests (a few bytes per signature), over whatever backend
the process group uses (``nccl`` == RCCL over xGMI on the GPU box, ``gloo`` in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(total, rank, world):
    """Contiguous block of [0, total) owned by ``rank``; blocks differ by at most one item."""
    base, rem = divmod(total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def is_dist():
    return dist.is_available() and dist.is_initialized()


def barrier():
    if is_dist():
        dist.barrier()


def max_over_ranks(seconds, device):
    """Every rank gets the slowest rank's time."""
    t = torch.tensor([seconds], dtype=torch.float64, device=device)
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.int64, device=device)
    if is_dist():
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return int(t.item())


def gather_per_signature(local, total, rank, world):
    """All-gather a per-signature 1-D tensor (status words, digests) sharded by ``shard_range`` into the global
    vector, on every rank.  Shards may be ragged, so they are padded to the largest shard for the collective."""
    if not is_dist() or world == 1:
        return local.clone()
    sizes = [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]
    width = max(sizes)
    padded = torch.zeros(width, dtype=local.dtype, device=local.device)
    padded[: local.numel()] = local
    out = [torch.empty_like(padded) for _ in range(world)]
    dist.all_gather(out, padded)
    return torch.cat([o[:n] for o, n in zip(out, sizes)])


def all_gather_chunks(local, gathered, async_op=False):
    """All-gather one witness chunk (same shape on every rank) into ``gathered`` ([world, *local.shape]).

    ``nccl`` (= RCCL): one ``all_gather_into_tensor`` straight between HBM buffers over xGMI; every rank ingests
    world-1 shards over its point-to-point links, so the time is ~ local_bytes * (world-1) / per-GPU ingest bandwidth.
    ``gloo`` (CPU rehearsal of the code path): staged through host memory.  Returns the work handle when async."""
    if not is_dist():
        gathered[0].copy_(local)
        return None
    # the collectives move bytes: 16-bit tensors (the input polynomials) are not a type every backend accepts
    lb, gb = local.reshape(-1).view(torch.uint8), gathered.reshape(-1).view(torch.uint8)
    if dist.get_backend() == "nccl":
        return dist.all_gather_into_tensor(gb, lb, async_op=async_op)
    host = [torch.empty(lb.shape, dtype=torch.uint8) for _ in range(dist.get_world_size())]
    dist.all_gather(host, lb.cpu())
    n = lb.numel()
    for r, h in enumerate(host):
        gb[r * n:(r + 1) * n].copy_(h)
    return None
