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


# ---- the shape of one benchmark / production step on `world` GPUs, as plain arithmetic ---------------------------------
# bench.py allocates from this plan and nothing else, and the CPU tests evaluate it for world = 1, 2, 4, 8: the rank
# arithmetic of an 8-GPU run is executed (and its HBM budget asserted) before the first 8-GPU node ever sees it.

HBM_BYTES_PER_GPU = 288 * 10**9          # MI355X: 288 GB of HBM3E (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_PLAN_FRACTION = 0.9                  # a plan may claim at most this much of it


def prove_leg_bytes(n, num_witness, num_instance, proofs_in_flight=64):
    """HBM the N > 1 run's proof leg holds while it runs (bench.py::prove_leg; released before the gather legs start): the
    proving key's five window tables (frw_msm.hip: 3,584 bytes per G1 point and 7,168 per G2 point for the witness-side
    queries -- and as much again for their ones tables --, 1,792 per point of h_query), the constraint matrices and transform tables of frw_r1cs_load (~1.3 KB per
    constraint), and frw_groth16_workspace_bytes for `proofs_in_flight` proofs (per proof: the products, h and working
    arrays of the witness map ~ 5 x 32 x domain + 96 C, the sort keys of the five sums ~ 64 x domain + 4 x 132 x (I + W), the
    buckets, work items and partial sums of the 16-bit pipeline ~ 168,000 x 240).  Falcon-1024: 6.2 + 0.3 + 64 x 0.2 = 19.3 GB."""
    nv = num_witness + num_instance
    constraints = num_witness + 6 * n + 2
    domain = 1
    while domain < constraints + num_instance:
        domain *= 2
    key = 2 * (3 * 3584 + 7168) * (nv + 3) + 1792 * domain          # window tables + the subset sums of every group of eight, as many bytes again
    matrices = 1300 * constraints + 9 * 32 * domain
    per_proof = 5 * 32 * domain + 96 * constraints + 64 * domain + 4 * 132 * nv + (32768 + 131072 + 4096) * 240 + (1 << 20)
    return int(key + matrices + proofs_in_flight * per_proof)


def aggregate_leg_bytes(statements, n, num_witness, num_instance):
    """HBM of the N > 1 run's aggregate leg (bench.py::aggregate_leg: ONE proof for `statements` signatures of this parameter set;
    released before the gather legs start, and never held together with the proof leg): the aggregate key's window tables (one point per
    variable of the whole statement), the transform tables of its domain (11 x 32 x domain), the per-signature matrices, the statements'
    own witnesses and the aggregate's assignment (2 x 32 bytes per variable), and the workspace of one proof."""
    nv = statements * (num_witness + num_instance - 1) + 1
    constraints = statements * (num_witness + 6 * n + 2)
    domain = 1
    while domain < constraints + statements * (num_instance - 1) + 1:
        domain *= 2
    key = 3 * 3584 * (nv + 3) + 7168 * (nv + 3) + 1792 * domain
    tables = 11 * 32 * domain + 1300 * (num_witness + 6 * n + 2) + 9 * 32 * (1 << 18)
    proof = 5 * 32 * domain + 96 * constraints + 64 * domain + 4 * 132 * nv + (32768 + 294912 + 4096) * 240 + (1 << 20)
    return int(key + tables + 2 * 32 * nv + proof)


# ---- BASELINE configs[4] as written: ONE proof for an aggregate of mixed statements, its key in slices over the ranks --------------------
BENCH_SEED = 0x46414C434F4E31            # bench.py SEED


def aggregate_mix(total=1024, seed=BENCH_SEED):
    """The parameter sets of the statements of the benchmark's mixed aggregate, in order (drawn from the seed: 513 Falcon-512 and
    511 Falcon-1024 for the 1,024 of BASELINE configs[4])."""
    import random
    rng = random.Random(seed)
    return [rng.choice([9, 10]) for _ in range(total)]


def circuit_counts(logn):
    """(I, W, C) of FalconNTTVerificationCircuit (README.md:44,55 of the reference; frw_layout)."""
    n = 1 << logn
    nb = 50 if logn == 9 else 52
    return 2 * n + 1, 153 * n + nb, 159 * n + nb + 2


def qap_pass_count(log_n):
    return (log_n + 5) // 6


# variables (other than the constant one) that some constraint of FalconNTTVerificationCircuit has on its B side: the columns of B in the
# matrices frw_r1cs_export writes (tests/test_gpu_aggregate.py holds the device's count to them)
B_SIDE_VARIABLES = {9: 46630, 10: 93222}


def sharded_aggregate_plan(world, rank, logns):
    """One proof for the aggregate of `logns`, the key of bare handles in `world` slices (frw_groth16_setup_r1cs_opts): what rank `rank`
    holds and sums, as plain arithmetic -- the statement's sizes, this rank's rows of the witness-side tables ([z_lo, z_hi) of nv + 3) and
    of h_query ([h_lo, h_hi) of domain - 1: frw.h's split, equal counts with the first `total mod world` slices one longer), and the
    bytes of what it keeps in HBM while the leg runs: the transform tables of the domain (every rank runs the whole witness map), its
    slices of the five tables, the statements' witnesses and the aggregate's assignment, and the workspace of one proof."""
    if world < 1 or not 0 <= rank < world or not logns:
        raise ValueError("sharded_aggregate_plan: bad arguments")
    ni, nw, nc = 1, 0, 0
    for g in logns:
        i, w, c = circuit_counts(g)
        ni, nw, nc = ni + i - 1, nw + w, nc + c
    log_n = 14
    while (1 << log_n) < nc + ni:
        log_n += 1
    n, nv = 1 << log_n, ni + nw
    z_lo, z_hi = shard_range(nv + 3, rank, world)
    h_lo, h_hi = shard_range(n - 1, rank, world)
    nz, nh = z_hi - z_lo, h_hi - h_lo
    passes = qap_pass_count(log_n)
    # the rows in which b_g1_query / b_g2_query hold a point: the variables some constraint has on its B side (B_SIDE_VARIABLES per
    # statement, the constant one, beta_2 and delta_2) -- the sums over those two tables run over them only (frw_groth16_pk.b_index);
    # a rank's share of them taken as its share of the rows
    nb_all = 1 + sum(B_SIDE_VARIABLES[g] for g in logns) + 2
    nb = max(1, -(-nb_all * nz // (nv + 3)))

    def narrow(n_):                                                                  # frw_msm.hip nmsm_max_items, nmsm_ones_max, nmsm_slices
        return (min(65536, max(2048, (n_ // 128 + 2047) // 2048 * 2048)) + 256, 1024 if n_ > (1 << 18) else 64,
                128 if n_ > (1 << 18) else 16)

    def sort_bytes(n_):                                                              # nmsm_carve_bare: 32 n entries at worst + the list of ones
        items, _, slices = narrow(n_)
        return 32 * slices * 128 * 4 + 32 * (3 * 128 + items + 8) * 4 + 4 * 33 * n_ + 4096

    def own_bytes(n_, tables, bucket_bytes):                                         # a table's partial sums: work items, buckets, ones
        items, ones_groups, _ = narrow(n_)
        return tables * (32 * (items + 128 + 1) + ones_groups + 64) * bucket_bytes + 4096
    # the sum over this rank's rows of h_query: thirteen 20-bit windows from 2^26 - 1 points (208 rows of 32,768 buckets, the two-level
    # sort's 8-byte and 4-byte entries), sixteen 16-bit windows below (entries 4 B and digits 2 B per point and window): frw_msm.hip
    # msm_carve_wide / msm_carve.  Its workspace and the witness map's are ONE region (same stream, one after the other: groth16_sizes).
    wide = nh >= (1 << 26) - 1
    if wide:
        h_windows = 13
        h_sum = 208 * (4 * 32768 * 4 + 65536 * 8 + 32768 * 240 + 2 * 4096 * 240 + 65536 * 240 + 3 * 240) + 13 * (8 + 4) * nh + 13 * 128 * 512 * 4 + 4096
    else:
        h_windows = 16
        h_sum = 16 * (4 * 32768 * 4 + 131072 * (8 + 240) + (32768 + 4096 + 1) * 240 + 6 * nh) + 4096
    qap = 3 * 32 * nc + 3 * 32 * n + 64
    buffers = {
        "transform tables of the domain (2 (K - 1) twists + 5 scales, 32 n bytes each)": (2 * (passes - 1) + 5) * 32 * n,
        "per-signature matrices (flattened rows, long rows, CSR)": 1300 * sum(circuit_counts(g)[2] for g in set(logns)),
        "key: rows of a_query, b_g1_query, l_query (112 B) and b_g2_query (224 B)": (3 * 112 + 224) * nz,
        "key: rows of h_query (112 B)": 112 * nh,
        "statements' witnesses and instances as the witness kernel wrote them": 32 * (nv + len(logns)),
        "the aggregate's assignment": 32 * nv,
        "proof: A z, B z, C z and three working arrays of the witness map | the sum over h_query, %d windows (one region: the larger)" % h_windows: max(qap, h_sum),
        "proof: h": 32 * n,
        "proof: z ++ [1, r, s]": 32 * (nv + 3),
        "proof: the sort of the slice's scalars (32 n entries at worst) + the own arrays of a_query and l_query": sort_bytes(nz) + own_bytes(nz, 2, 240),
        "proof: the sort over the rows b_g1_query / b_g2_query hold a point in + b_g1_query's own arrays": sort_bytes(nb) + own_bytes(nb, 1, 240),
        "proof: b_g2_query's own arrays": own_bytes(nb, 1, 464),
    }
    return {"world": world, "rank": rank, "statements": len(logns), "falcon512": list(logns).count(9), "falcon1024": list(logns).count(10),
            "num_instance": ni, "num_witness": nw, "num_constraints": nc, "log_domain_size": log_n,
            "z_lo": z_lo, "z_hi": z_hi, "h_lo": h_lo, "h_hi": h_hi,
            "windows_h_query": h_windows, "bucket_additions_h_query": h_windows * nh, "rows_of_b_queries_holding_a_point": nb,
            "partial_sum_bytes_per_rank": 72 * 8,
            "buffers": buffers, "hbm_plan_bytes": int(sum(buffers.values())),
            "hbm_limit_bytes": int(HBM_PLAN_FRACTION * HBM_BYTES_PER_GPU),
            "fits": int(sum(buffers.values())) <= int(HBM_PLAN_FRACTION * HBM_BYTES_PER_GPU)}


def check_sharded_aggregate_plans(plans):
    """The slices of the ranks tile both row ranges in rank order, every rank proves the same statement, and every plan fits HBM."""
    world = plans[0]["world"]
    assert len(plans) == world and [p["rank"] for p in plans] == list(range(world))
    nv3 = plans[0]["num_instance"] + plans[0]["num_witness"] + 3
    assert plans[0]["z_lo"] == 0 and plans[-1]["z_hi"] == nv3 and plans[0]["h_lo"] == 0 and plans[-1]["h_hi"] == (1 << plans[0]["log_domain_size"]) - 1
    for a, b in zip(plans, plans[1:]):
        assert a["z_hi"] == b["z_lo"] and a["h_hi"] == b["h_lo"], "slices must be contiguous"
    for p in plans:
        assert p["z_hi"] > p["z_lo"] and p["h_hi"] > p["h_lo"], "more ranks than rows"
        for k in ("statements", "num_instance", "num_witness", "num_constraints", "log_domain_size"):
            assert p[k] == plans[0][k]
        assert p["fits"], "rank %d plans %.1f GB of HBM for the sharded aggregate, limit %.1f GB" % (p["rank"], p["hbm_plan_bytes"] / 1e9, p["hbm_limit_bytes"] / 1e9)
    return True


def all_gather_bytes(local, world, rank):
    """All-gather a small 1-D uint8/int64 tensor of the same shape on every rank (a rank's partial sums of a sharded proof: 576 bytes):
    returns [world, *local.shape] on local's device.  RCCL when the group is `nccl` (device tensors), staged through the host for gloo."""
    if not is_dist() or world == 1:
        return local.clone().unsqueeze(0)
    if dist.get_backend() == "nccl":
        out = torch.empty((world,) + tuple(local.shape), dtype=local.dtype, device=local.device)
        dist.all_gather_into_tensor(out, local.contiguous())
        return out
    host = [torch.empty(local.shape, dtype=local.dtype) for _ in range(world)]
    dist.all_gather(host, local.cpu())
    return torch.stack(host).to(local.device)


def step_plan(world, rank, batch_per_gpu, chunk, allgather_chunk, n, num_witness, num_instance, compact_bytes,
              with_gather_legs=True, with_prove_leg=False, aggregate_statements=0):
    """Everything bench.py derives from (world, rank, per-GPU batch, launch size): this rank's global index range, the
    launches of a step, the chunking of the gather legs and the bytes of every HBM buffer the run allocates.

    Independent signatures (falcon_ntt.rs:143-151) => rank r owns global indices [r * batch_per_gpu, (r+1) * batch_per_gpu)
    (``shard_range`` of an evenly divisible total).  All sizes in bytes; ``hbm_plan_bytes`` is their sum."""
    if world < 1 or not 0 <= rank < world or batch_per_gpu < 1 or chunk < 1:
        raise ValueError("step_plan: bad arguments")
    chunk = min(chunk, batch_per_gpu)
    lo, hi = shard_range(batch_per_gpu * world, rank, world)
    nchunks = (batch_per_gpu + chunk - 1) // chunk
    wit_row, inst_row = 32 * num_witness, 32 * num_instance
    buffers = {
        "inputs (sig, pk, hm as u16)": 3 * 2 * n * batch_per_gpu,
        "witness buffer (one launch, reused)": chunk * wit_row,
        "instance buffer (one launch, reused)": chunk * inst_row,
        "status words": 4 * batch_per_gpu,
        "digests of the checked launch": 8 * chunk + 4 * chunk,
    }
    plan = {"world": world, "rank": rank, "global_lo": lo, "global_hi": hi, "batch_per_gpu": batch_per_gpu,
            "signatures_per_step_all_gpus": batch_per_gpu * world, "chunk": chunk, "launches_per_step": nchunks}
    if with_gather_legs and world >= 1:
        # signatures per rank per collective; the expansion of all `world` shards must fit the witness buffer
        gc = max(1, min(allgather_chunk or 4096, chunk // world))
        nk = batch_per_gpu // gc
        own_in_place = (world + 1) * gc <= chunk           # room for the direct launch next to the expansion?
        own_checked = gc if own_in_place else min(gc, 1024)  # else a buffer of its own: the first 1,024 of the shard
        probe_gc = max(1, min(512, chunk // 2, 2048 // world))
        plan.update({"gather_chunk_per_rank": gc, "gather_chunks": nk, "gather_signatures_per_rank": nk * gc,
                     "gathered_signatures_per_collective": world * gc, "own_shard_checked_in_place": own_in_place,
                     "own_shard_signatures_checked": own_checked,
                     "all_digests_gathered": world * world * gc, "probe_chunk_per_rank": probe_gc})
        if world * gc > chunk:
            raise ValueError("step_plan: %d ranks x %d signatures do not fit a witness buffer of %d" % (world, gc, chunk))
        buffers.update({
            "compact chunks, local (double buffer)": 2 * gc * compact_bytes,
            "compact chunks, gathered (double buffer)": 2 * world * gc * compact_bytes,
            "instances of the expanded chunk": world * gc * inst_row,
            "direct launch of the own shard (parity of the gather)": (0 if own_in_place else own_checked * wit_row)
                                                                     + own_checked * inst_row,
            "digests of the expanded chunk": 8 * (world * gc + gc),
            "input chunks, local + gathered (regenerate leg, double buffer)": 2 * (1 + world) * 3 * 2 * n * gc,
            "status of the regenerated chunk": 4 * world * gc,
            "naive probe: gathered 32-byte witnesses": world * probe_gc * wit_row,
        })
    if with_prove_leg or aggregate_statements:
        # the two legs run one after the other and give their memory back in between: the plan holds the larger
        p = prove_leg_bytes(n, num_witness, num_instance) if with_prove_leg else 0
        a = aggregate_leg_bytes(aggregate_statements, n, num_witness, num_instance) if aggregate_statements else 0
        buffers["proof legs (transient, the larger of the two): 64 per-signature proofs | one proof for %d statements" % aggregate_statements] = max(p, a)
    plan["buffers"] = buffers
    plan["hbm_plan_bytes"] = int(sum(buffers.values()))
    plan["hbm_limit_bytes"] = int(HBM_PLAN_FRACTION * HBM_BYTES_PER_GPU)
    plan["fits"] = plan["hbm_plan_bytes"] <= plan["hbm_limit_bytes"]
    return plan


def check_plans(plans):
    """Cross-rank invariants of the plans of one world (raises AssertionError with the reason): the shards tile the global
    index range in rank order, every rank has the same shape, the gather legs cover whole steps, and the plan fits HBM."""
    world = plans[0]["world"]
    assert len(plans) == world and [p["rank"] for p in plans] == list(range(world)), "one plan per rank, in rank order"
    assert plans[0]["global_lo"] == 0 and plans[-1]["global_hi"] == plans[0]["signatures_per_step_all_gpus"]
    for a, b in zip(plans, plans[1:]):
        assert a["global_hi"] == b["global_lo"], "shards must be contiguous"
    for p in plans:
        assert p["global_hi"] - p["global_lo"] == p["batch_per_gpu"], "every rank owns exactly batch_per_gpu signatures"
        assert p["launches_per_step"] * p["chunk"] >= p["batch_per_gpu"] > (p["launches_per_step"] - 1) * p["chunk"]
        for k in ("chunk", "launches_per_step", "hbm_plan_bytes") + (("gather_chunk_per_rank", "gather_chunks")
                                                                      if "gather_chunks" in p else ()):
            assert p[k] == plans[0][k], "rank %d differs from rank 0 in %s" % (p["rank"], k)
        if "gather_chunks" in p:
            assert p["gather_chunks"] >= 1 and p["gathered_signatures_per_collective"] <= p["chunk"]
            assert p["all_digests_gathered"] == world * p["gathered_signatures_per_collective"]
        assert p["fits"], "rank %d plans %.1f GB of HBM, limit %.1f GB" % (p["rank"], p["hbm_plan_bytes"] / 1e9,
                                                                            p["hbm_limit_bytes"] / 1e9)
    return True
