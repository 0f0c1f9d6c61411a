"""GPU parity of the multi-scalar multiplication over BLS12-381 G1 (frw_msm_g1_dev / frw_groth16_msm_h_dev) against the
oracle: oracle/bls12_381.c's bucket method (pinned to Python integers by tests/test_bls12_381.py) on small and adversarial
inputs, the MSM-free value (h(t) zt / delta) G1 for a proving key made from known toxic waste at full size, and the committed
golden points of tests/golden/msm.json.  Bit-exact: the result is ONE affine point, in ark-ff's bytes."""
import hashlib
import json
import os
import random
import sys

import numpy as np
import pytest

import frw_testlib as T
from oracle import bls12_381 as E

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
FR_R = (1 << 256) % E.R


def _run(engine, handle, scalars, montgomery, chunk=None):
    """scalars: uint64[batch, stride, 4] on the host -> uint64[batch, 12]."""
    import torch
    dev = torch.device("cuda:0")
    info = engine.msm_info(handle)
    batch, stride = scalars.shape[0], scalars.shape[1]
    d_sc = torch.from_numpy(scalars.view(np.int64)).to(dev)
    out = torch.full((batch, 12), -1, dtype=torch.int64, device=dev)
    per = int(info.workspace_bytes_per_signature)
    ws = torch.empty((chunk or batch) * per, dtype=torch.uint8, device=dev)
    engine.msm_g1_dev(handle, batch, d_sc, stride, montgomery, out, ws, ws.numel(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    return out.cpu().numpy().view(np.uint64)


# (narrow, bare); bare == "wide": a dense bare handle on thirteen 20-bit windows (208 rows: the two-level sort, weighted and plain folds)
HANDLES = [(False, False), (True, False), (False, True), (True, True), (False, "wide")]
HANDLE_IDS = ["tables-16bit", "tables-8bit", "bare-16bit", "bare-8bit", "bare-20bit"]

@pytest.mark.parametrize("narrow,bare", HANDLES, ids=HANDLE_IDS)
def test_small_and_adversarial_inputs_equal_the_cpu_bucket_method(engine, oracle, narrow, bare):
    """Both pipelines (16-bit windows / 32,768 buckets; 8-bit windows / 128 buckets: frw_msm_g1_load_narrow), over window tables and over
    bare handles (the points only: frw_msm_g1_load_bare -- the sums then run window by window and end in Horner's rule)."""
    rng = random.Random(2026)
    n = 300
    ks = [rng.randrange(1, E.R) for _ in range(n)]
    bases = oracle.g1_fixed_base(T.ints_to_limbs(ks))
    # bases that meet each other inside a bucket: a duplicate, a negative, the point at infinity
    bases[10] = bases[3]
    bases[11] = np.array(E.to_limbs(E.neg(E.from_limbs(bases[4]))), dtype=np.uint64)
    bases[12] = 0
    vectors = [
        [rng.randrange(E.R) for _ in range(n)],
        [0] * n,
        [1] * n,                                                   # every point into bucket 0 of window 0
        [E.R - 1] * n,                                             # negative digits and carries all the way up
        [0x0001000100010001000100010001000100010001000100010001000100010001] * n,      # one bucket, sixteen windows
        [0x0101010101010101010101010101010101010101010101010101010101010101] * n,      # one bucket, thirty-two windows (9,600 entries in it: cut into items)
        [(1 << 255) - 19 if i % 2 else 0x8000 for i in range(n)],  # digit exactly 2^15 (kept positive); top-window carry
        [0x80 if i % 3 else 0x7f81 for i in range(n)],             # digit exactly 2^7 (kept positive), 0x81 -> -0x7f with a carry into 0x7f -> 0x80
        [0x80000 if i % 2 else (1 << 255) - (1 << 19) + 1 for i in range(n)],    # digit exactly 2^19 (kept positive) / twenty-bit windows full of carries
        [sum(((j * 7 + i) % 16 * 32768 + 5) << (20 * j) for j in range(12)) for i in range(n)],     # every coarse class of the wide windows, bucket 5 of each
        [rng.randrange(1 << 16) for _ in range(n)],
        # out of contract for "canonical" scalars, but harmless: integers >= r are taken mod r (k P = (k mod r) P), never past the buckets
        [rng.choice([E.R, E.R + 5, (1 << 256) - 1, 2 * E.R + 7, (1 << 255) + rng.randrange(1 << 200)]) for _ in range(n)],
    ]
    vectors[0][3] = vectors[0][10]                                 # the duplicated base with the same scalar: doubling
    vectors[0][11] = vectors[0][4]                                 # the negated base with the same scalar: cancellation
    handle = engine.msm_g1_load(bases, narrow=narrow, bare=bool(bare), wide=bare == "wide")
    try:
        info = engine.msm_info(handle)
        assert info.num_points == n and (info.window_bits, info.num_windows) == ((20, 13) if bare == "wide" else (8, 32) if narrow else (16, 16))
        assert info.table_bytes == n * 112 * (1 if bare else info.num_windows) + (0 if bare or not narrow else (n + 7) // 8 * 255 * 112)
        canon = np.stack([T.ints_to_limbs(v) for v in vectors])
        mont = np.stack([T.ints_to_limbs([x * FR_R % E.R for x in v]) for v in vectors])
        want = [oracle.g1_msm(bases, T.ints_to_limbs([x % E.R for x in v]), 11).tolist() for v in vectors]
        assert want[1] == [0] * 12                                 # all-zero scalars: the point at infinity
        for got in (_run(engine, handle, canon, 0), _run(engine, handle, mont, 1), _run(engine, handle, mont, 1, chunk=3)):
            assert [g.tolist() for g in got] == want
        # more than 16 signatures in one chunk: the narrow pipeline then adds a bucket's items up in the fold itself instead of
        # giving every bucket a workgroup (both forms must agree with the CPU)
        many = np.concatenate([canon, canon, canon])[:20]
        assert [g.tolist() for g in _run(engine, handle, many, 0)] == (want * 3)[:20]
        # a longer stride than points: only the first n scalars of each vector count (how h is laid out: n + 1 per signature)
        padded = np.concatenate([canon, np.full((len(vectors), 5, 4), 0xFFFFFFFF, dtype=np.uint64)], axis=1)
        assert [g.tolist() for g in _run(engine, handle, padded, 0)] == want
    finally:
        engine.msm_free(handle)


def _h_query(oracle, n, toxic):
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_msm
    scalars, _ = make_msm.h_query_scalars(n, toxic)
    return oracle.g1_fixed_base(T.ints_to_limbs(scalars), threads=min(16, os.cpu_count() or 1)), make_msm


@pytest.mark.parametrize("logn", [9, 10])
def test_h_acc_of_gpu_witnesses_equals_the_msm_free_value(engine, oracle, logn):
    """The prover's flow on the device, end to end: synthetic signatures -> witness (the hot path) -> witness map -> h_acc over
    a proving key's h_query (2^17 - 1 / 2^18 - 1 points) made from known toxic waste.  Expected value per signature, with no
    multi-scalar multiplication anywhere: (h(t) zt / delta) G1 from the GPU's own h read back as integers."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    batch = 3
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=777 + logn)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    r = engine.r1cs_load(0, logn)
    q = engine.qap_info(r)
    n = int(q.domain_size)
    ws = torch.empty(batch * int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
    h = torch.empty((batch, n, 4), dtype=torch.int64, device=dev)
    bad = torch.empty(batch, dtype=torch.int32, device=dev)
    engine.qap_witness_map_dev(r, batch, wit, inst, h, ws, ws.numel(), bad, s0)
    torch.cuda.synchronize()
    engine.r1cs_free(r)
    del ws
    assert not bad.any()
    toxic = {"t": 0x0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF0123456789ABCDEF % E.R, "delta": 987654321987654321 + logn}
    bases, make_msm = _h_query(oracle, n, toxic)
    handle = engine.msm_g1_load(bases)
    try:
        info = engine.msm_info(handle)
        assert info.num_points == n - 1
        out = torch.full((batch, 12), -1, dtype=torch.int64, device=dev)
        mws = torch.empty(2 * int(info.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)      # chunks of 2 + 1
        engine.groth16_msm_h_dev(handle, batch, h, n, out, mws, mws.numel(), s0)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        r_inv = pow(FR_R, -1, E.R)
        for i in range(batch):
            h_int = [v * r_inv % E.R for v in T.limbs_to_ints(h[i].cpu().numpy().view(np.uint64))]
            assert got[i].tolist() == E.to_limbs(make_msm.expected_point(h_int, n, toxic)), i
        with pytest.raises(frw.FrwError):                          # a key of another size
            engine.groth16_msm_h_dev(handle, batch, h, n // 2, out, mws, mws.numel(), s0)
    finally:
        engine.msm_free(handle)


def test_h_acc_equals_the_committed_golden_points(engine, oracle):
    """tests/golden/msm.json (made by tests/golden/make_msm.py from the oracle alone): witness fixture -> GPU witness -> GPU
    witness map -> GPU h_acc == the committed affine point, for both parameter sets; the bases the oracle regenerates here are
    the ones the golden file was cross-checked with (digest)."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    gold = json.load(open(os.path.join(HERE, "golden", "msm.json")))
    toxic = {k: int(v, 16) for k, v in gold["toxic"].items()}
    s0 = torch.cuda.current_stream().cuda_stream
    for case in gold["cases"]:
        logn, n = case["logn"], case["domain_size"]
        fx = json.load(open(os.path.join(HERE, "golden", case["witness_fixture"])))
        L = frw.layout(logn)
        d = [torch.from_numpy(np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16).copy().view(np.int16)).to(dev) for k in ("sig", "pk", "hm")]
        wit = torch.empty((1, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((1, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.empty(1, dtype=torch.int32, device=dev)
        engine.witness_ntt_verify_dev(logn, 1, d[0], d[1], d[2], wit, inst, st, 1, s0)
        r = engine.r1cs_load(0, logn)
        q = engine.qap_info(r)
        assert int(q.domain_size) == n
        ws = torch.empty(int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
        h = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
        engine.qap_witness_map_dev(r, 1, wit, inst, h, ws, ws.numel(), None, s0)
        torch.cuda.synchronize()
        engine.r1cs_free(r)
        bases, _ = _h_query(oracle, n, toxic)
        assert hashlib.sha256(bases.tobytes()).hexdigest() == case["bases_sha256"]
        handle = engine.msm_g1_load(bases)
        try:
            info = engine.msm_info(handle)
            out = torch.empty((1, 12), dtype=torch.int64, device=dev)
            mws = torch.empty(int(info.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
            engine.groth16_msm_h_dev(handle, 1, h, n, out, mws, mws.numel(), s0)
            torch.cuda.synchronize()
            assert ["%016x" % v for v in out.cpu().numpy().view(np.uint64)[0].tolist()] == case["h_acc"], case["witness_fixture"]
        finally:
            engine.msm_free(handle)


def test_fixed_base_multiples_of_the_generator_equal_the_oracle(engine, oracle):
    """frw_g1_fixed_base (the FixedBaseMSM ark-groth16's generator builds the queries with) == oracle/bls12_381.c, bit for bit."""
    rng = random.Random(99)
    ks = [0, 1, 2, 255, 256, E.R - 1, (1 << 255) - 1] + [rng.randrange(E.R) for _ in range(200)]
    got = engine.g1_fixed_base(T.ints_to_limbs(ks))
    want = oracle.g1_fixed_base(T.ints_to_limbs(ks))
    assert np.array_equal(got, want)
    assert got[0].tolist() == [0] * 12 and got[1].tolist() == E.to_limbs(E.G1)


@pytest.mark.parametrize("narrow,bare", HANDLES, ids=HANDLE_IDS)
def test_witness_scalars_mostly_zero_and_one_equal_the_cpu_bucket_method(engine, oracle, narrow, bare):
    """The witness-side sums of the prover (prover.rs calculate_coeff / l_aux_acc: MSM(a_query | b_g1_query | l_query,
    assignment)): the scalars are a Falcon-512 witness -- 91 % of them 0 or 1 -- over as many bases as the circuit has
    variables.  Ones are summed apart from the buckets; the result must be the CPU's, bit for bit, for genuine witnesses and
    for the degenerate vectors (all ones, a single one)."""
    import torch
    import falcon_r1cs_amd as frw
    logn, batch = 9, 3
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=31337)
    wit, inst, st = engine.witness_ntt_verify(logn, sig, pk, hm, frw.ENC_CANONICAL)
    z = np.concatenate([inst[:, 1:], wit], axis=1)                # calculate_coeff: query[1..] x (inputs ++ aux)
    n = z.shape[1]
    assert n == L.num_instance - 1 + L.num_witness
    ones = (z == np.array([1, 0, 0, 0], dtype=np.uint64)).all(axis=2).mean()
    assert 0.3 < ones < 0.6
    rng = random.Random(5)
    bases = oracle.g1_fixed_base(T.ints_to_limbs([rng.randrange(1, E.R) for _ in range(n)]), threads=min(16, os.cpu_count() or 1))
    z = np.concatenate([z, np.zeros((2, n, 4), dtype=np.uint64)])
    z[batch, :, 0] = 1                                            # all ones
    z[batch + 1, 12345, 0] = 1                                    # a single one
    handle = engine.msm_g1_load(bases, narrow=narrow, bare=bool(bare), wide=bare == "wide")
    try:
        got = _run(engine, handle, z, 0)
        for i in range(z.shape[0]):
            assert got[i].tolist() == oracle.g1_msm(bases, z[i], 13, threads=min(16, os.cpu_count() or 1)).tolist(), i
        assert got[batch + 1].tolist() == bases[12345].tolist()
    finally:
        engine.msm_free(handle)


@pytest.mark.parametrize("narrow,bare", HANDLES, ids=HANDLE_IDS)
def test_g2_fixed_base_and_msm_equal_python_integers(engine, narrow, bare):
    """G2 (prover.rs: g2_b = MSM(b_g2_query, assignment) + ...): the generator's fixed-base multiples and a multi-scalar
    multiplication against oracle/bls12_381.py's Fq2 arithmetic in Python integers -- bases as multiples k_i G2 of the
    published generator, so that the expected sum is (sum s_i k_i) G2, one scalar multiplication; with the same degenerate
    inputs as in G1 (a base twice, a base and its negative, infinity, zeros, ones, r - 1)."""
    rng = random.Random(4242)
    n = 200
    ks = [rng.randrange(1, E.R) for _ in range(n)]
    ks[10] = ks[3]
    ks[11] = E.R - ks[4]
    ks[12] = 0                                                     # the point at infinity as a base
    bases = engine.g2_fixed_base(T.ints_to_limbs(ks))
    for i in (0, 3, 11, 12, 57):
        assert bases[i].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, ks[i])), i
    assert engine.g2_fixed_base(T.ints_to_limbs([1]))[0].tolist() == E.g2_to_limbs(E.G2)
    vectors = [[rng.randrange(E.R) for _ in range(n)], [0] * n, [1] * n, [E.R - 1] * n,
               [rng.choice([0, 1, 1, rng.randrange(1 << 14), rng.randrange(1 << 146)]) for _ in range(n)]]
    vectors[0][3] = vectors[0][10]
    vectors[0][11] = vectors[0][4]
    handle = engine.msm_g2_load(bases, narrow=narrow, bare=bool(bare), wide=bare == "wide")
    try:
        import torch
        dev = torch.device("cuda:0")
        info = engine.msm_info(handle)
        assert info.num_points == n
        sc = np.stack([T.ints_to_limbs(v) for v in vectors])
        d_sc = torch.from_numpy(sc.view(np.int64)).to(dev)
        out = torch.full((len(vectors), 24), -1, dtype=torch.int64, device=dev)
        ws = torch.empty(2 * int(info.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
        engine.msm_g2_dev(handle, len(vectors), d_sc, n, 0, out, ws, ws.numel(), 0)
        torch.cuda.synchronize()
        got = out.cpu().numpy().view(np.uint64)
        for i, v in enumerate(vectors):
            want = E.g2_mul(E.G2, sum(s * k for s, k in zip(v, ks)) % E.R)
            assert got[i].tolist() == E.g2_to_limbs(want), i
        import falcon_r1cs_amd as frw
        with pytest.raises(frw.FrwError):                          # a G2 table through the G1 entry point
            engine.msm_g1_dev(handle, 1, d_sc, n, 0, out, ws, ws.numel(), 0)
    finally:
        engine.msm_free(handle)


@pytest.mark.parametrize("bare", [False, True, "wide"], ids=["tables", "bare", "bare-20bit"])
def test_one_digit_in_every_window_of_a_large_sum(engine, oracle, bare):
    """More than 2^18 points (the finest split of the dense pipeline: a bucket of the mean size cut 16 ways) whose scalars are ALL the same
    value with one digit in every window: ONE bucket takes all 16 n entries and is cut into 2^18 work items -- which a work item packed
    as bucket | part << 15 could not number (round 4's ADVICE).  The bases are multiples k_i G of the generator, so the expected sum is
    (k sum k_i) G: one scalar multiplication."""
    rng = random.Random(99)
    n = (1 << 18) + 1234
    ks = [rng.randrange(1, E.R) for _ in range(n)]
    bases = engine.g1_fixed_base(T.ints_to_limbs(ks))
    for i in (0, 1, n - 1):
        assert bases[i].tolist() == oracle.g1_scalar_mul(oracle.g1_generator(), ks[i]).tolist()
    k = int("0002" * 16, 16)
    ksum = sum(ks) % E.R
    vectors = np.stack([T.ints_to_limbs([k] * n), T.ints_to_limbs([rng.randrange(E.R) if i % 4096 == 0 else k for i in range(n)])])
    extra = sum((int.from_bytes(vectors[1][i].tobytes(), "little") - k) * ks[i] for i in range(0, n, 4096))
    want = [oracle.g1_scalar_mul(oracle.g1_generator(), k * ksum % E.R).tolist(),
            oracle.g1_scalar_mul(oracle.g1_generator(), (k * ksum + extra) % E.R).tolist()]
    handle = engine.msm_g1_load(bases, bare=bool(bare), wide=bare == "wide")
    try:
        # one signature at a time (the finest split: a chunk of one or two), then both in one chunk
        got1 = [_run(engine, handle, vectors[i:i + 1], 0)[0].tolist() for i in range(2)]
        assert got1 == want
        assert [g.tolist() for g in _run(engine, handle, vectors, 0)] == want
    finally:
        engine.msm_free(handle)
