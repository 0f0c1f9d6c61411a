"""GPU parity tests: the HIP path, called through the C ABI, against the oracle -- bit exact.

All inputs are seeded; sizes are chosen so the oracle finishes in seconds.
"""
import random

import numpy as np
import pytest

import frw_testlib as T

pytestmark = pytest.mark.gpu


def _first_diff(a, b):
    bad = np.nonzero((a != b).reshape(a.shape[0], a.shape[1], -1).any(axis=2))
    return list(zip(bad[0][:5].tolist(), bad[1][:5].tolist()))


@pytest.mark.parametrize("logn", [9, 10])
@pytest.mark.parametrize("enc", [0, 1])
def test_full_witness_matches_oracle(engine, oracle, logn, enc):
    """falcon_ntt.rs:26-123 end to end, host-buffer entry point, batch of 6 synthetic signatures."""
    import falcon_r1cs_amd as frw
    sig, pk, hm = frw.synth_triples(logn, 6, seed=1234 + logn)
    wit, inst, st = engine.witness_ntt_verify(logn, sig, pk, hm, enc, strict=True)
    owit, oinst, ost = oracle.witness_ntt_verify(logn, sig, pk, hm, enc)
    assert st.tolist() == ost.tolist() == [0] * 6
    assert np.array_equal(inst, oinst), _first_diff(inst, oinst)
    assert np.array_equal(wit, owit), _first_diff(wit, owit)


@pytest.mark.parametrize("logn", [9, 10])
@pytest.mark.parametrize("enc", [0, 1])
def test_ntt_modq_matches_oracle(engine, oracle, logn, enc):
    """poly.rs:104-159 alone on uniform random polynomials (as Polynomial::rand, poly.rs:268)."""
    rng = np.random.default_rng(99 + logn)
    poly = rng.integers(0, T.Q, size=(16, 1 << logn), dtype=np.uint16)
    wit, out, st = engine.ntt_modq(logn, poly, enc)
    owit, oout = oracle.ntt_modq(logn, poly, enc)
    assert not st.any()
    assert np.array_equal(out, oout)
    assert np.array_equal(wit, owit), _first_diff(wit, owit)
    # poly.rs:292-297: the reduced outputs are the Falcon NTT of the input
    for i in range(poly.shape[0]):
        assert np.array_equal(out[i], oracle.ntt_clear(logn, poly[i]))


@pytest.mark.parametrize("logn", [9, 10])
def test_edge_polynomials(engine, oracle, logn):
    """All-zero, all-(q-1), single spikes: extreme ladder values (maximum t) and empty norms."""
    n = 1 << logn
    rows = [np.zeros(n), np.full(n, T.Q - 1), np.eye(1, n, 0)[0] * (T.Q - 1), np.eye(1, n, n - 1)[0],
            np.arange(n) % T.Q, (np.arange(n) * 7919) % T.Q]
    poly = np.array(rows, dtype=np.uint16)
    wit, out, st = engine.ntt_modq(logn, poly, 1)
    owit, oout = oracle.ntt_modq(logn, poly, 1)
    assert np.array_equal(out, oout) and np.array_equal(wit, owit)
    # zero signature: v = hm, sig = 0
    zero = np.zeros((1, n), dtype=np.uint16)
    small = np.array([[(i % 5) for i in range(n)]], dtype=np.uint16)
    for sig, pk, hm in [(zero, zero, zero), (zero, poly[4:5], small), (small, poly[5:6], zero)]:
        w, i_, s = engine.witness_ntt_verify(logn, sig, pk, hm, 1, strict=False)
        ow, oi, os_ = oracle.witness_ntt_verify(logn, sig, pk, hm, 1)
        assert s.tolist() == os_.tolist()
        assert np.array_equal(i_, oi) and np.array_equal(w, ow)


@pytest.mark.parametrize("logn", [9, 10])
def test_status_and_strict_mode(engine, oracle, logn):
    """range_proofs.rs:57-60,114-117,205-208: strict = the non-test build (error), permissive = cfg(test)
    (truncated-bit witness, unsatisfied system)."""
    import falcon_r1cs_amd as frw
    rng = random.Random(5)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    big_sig, big_pk, big_hm, big_v = T.random_triple(logn, rng, scale=1.6)       # norm well above the bound
    assert T.centred_norm(big_sig, big_v) >= T.SIG_L2_BOUND[logn]
    bad = sig.copy()
    bad[3] = T.Q                                                                   # coefficient out of range
    S = np.stack([sig, big_sig, bad])
    P = np.stack([pk, big_pk, pk])
    H = np.stack([hm, big_hm, hm])
    wit, inst, st = engine.witness_ntt_verify(logn, S, P, H, 1, strict=False)
    owit, oinst, ost = oracle.witness_ntt_verify(logn, S, P, H, 1)
    assert st.tolist() == ost.tolist() == [frw.ST_OK, frw.ST_NORM_BOUND, frw.ST_COEFF_RANGE]
    assert np.array_equal(wit[:2], owit[:2]) and np.array_equal(inst[:2], oinst[:2])
    with pytest.raises(frw.FrwError) as ei:
        engine.witness_ntt_verify(logn, S, P, H, 1, strict=True)
    assert ei.value.code == -5
    with pytest.raises(ValueError):
        engine.witness_ntt_verify(logn, sig[:-1], pk[:-1], hm[:-1])               # poly.rs:110-112


def test_device_path_digest_and_ragged_batches(engine, oracle):
    """Device-pointer entry point on batches that do not divide the persistent grid, checked per signature by
    digest (frw_digest_dev == oracle digest of the oracle's witness)."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    for logn, batch in [(9, 1), (9, 37), (10, 3), (10, 130)]:
        L = frw.layout(logn)
        sig, pk, hm = frw.synth_triples(logn, batch, seed=42, first_index=1000)
        d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
        wit = torch.zeros((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.zeros((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        dig = torch.zeros(batch, dtype=torch.int64, device=dev)
        stream = torch.cuda.current_stream().cuda_stream
        engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, stream)
        engine.digest_dev(wit, L.num_witness * 4, batch, dig, stream)
        torch.cuda.synchronize()
        assert not st.cpu().numpy().any()
        owit, oinst, _ = oracle.witness_ntt_verify(logn, sig, pk, hm, 1, threads=8)
        want = [oracle.digest(owit[i]) for i in range(batch)]
        got = [int(x) & (2 ** 64 - 1) for x in dig.cpu().numpy()]
        assert got == want
        assert np.array_equal(inst.cpu().numpy().view(np.uint64), oinst)


def test_hip_matches_golden_fixtures(engine):
    """Committed fixtures (tests/golden/, produced by the gadget-by-gadget oracle): SHA-256 of both assignment
    vectors in both encodings."""
    import glob
    import hashlib
    import json
    import os
    paths = sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "witness_*.json")))
    assert len(paths) >= 3
    for path in paths:
        fx = json.load(open(path))
        sig, pk, hm = (np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16) for k in ("sig", "pk", "hm"))
        for enc, name in ((0, "canonical"), (1, "montgomery")):
            wit, inst, st = engine.witness_ntt_verify(fx["logn"], sig, pk, hm, enc, strict=True)
            assert wit.shape[1] == fx["num_witness"] and inst.shape[1] == fx["num_instance"]
            assert hashlib.sha256(wit.tobytes()).hexdigest() == fx["witness_sha256"][name]
            assert hashlib.sha256(inst.tobytes()).hexdigest() == fx["instance_sha256"][name]


@pytest.mark.parametrize("logn", [9, 10])
def test_full_size_properties(engine, logn):
    """BASELINE.json's sizes (4096 Falcon-512 / one 4096-signature launch of Falcon-1024) through size-independent
    properties, computed on the device: (1) every reduced NTT output b in S3 equals the Falcon NTT of sig, i.e.
    the instance relation hm_ntt = v_ntt + sig_ntt*pk_ntt holds on the emitted values; (2) determinism: two runs
    give identical per-signature digests; (3) digests of a strided sample equal the oracle's is covered by
    test_device_path_digest_and_ragged_batches; here (3') all statuses are OK and digests are pairwise distinct."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    batch = 4096
    L = frw.layout(logn)
    n = L.n
    sig, pk, hm = frw.synth_triples(logn, batch, seed=2024)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    dig = [torch.zeros(batch, dtype=torch.int64, device=dev) for _ in range(2)]
    stream = torch.cuda.current_stream().cuda_stream
    for k in range(2):
        engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 0, stream)   # canonical encoding
        engine.digest_dev(wit, L.num_witness * 4, batch, dig[k], stream)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0
    assert torch.equal(dig[0], dig[1])
    assert torch.unique(dig[0]).numel() == batch
    # canonical encoding: limb 0 carries every value < 2^64
    w0 = wit[:, :, 0]
    assert int(wit[:, : 2 * n, 1:].abs().sum()) == 0
    S = L.seg_off
    b_sig = w0[:, S[3] + 1: S[3] + 29 * n: 29]          # b of mod_q blocks of NTT(sig)
    b_v = w0[:, S[4] + 1: S[4] + 29 * n: 29]
    prod = w0[:, S[5]: S[5] + 30 * n: 30]
    tq = w0[:, S[5] + 1: S[5] + 30 * n: 30]
    c = w0[:, S[5] + 2: S[5] + 30 * n: 30]
    pk_ntt = inst[:, 1: 1 + n, 0]
    hm_ntt = inst[:, 1 + n: 1 + 2 * n, 0]
    assert torch.equal(prod, b_sig * pk_ntt)                                   # falcon_ntt.rs:107
    assert torch.equal(b_v + prod, tq * 12289 + c)                             # arithmetics.rs:252-256
    assert torch.equal(c, hm_ntt)                                              # falcon_ntt.rs:105
    assert int(b_sig.max()) < 12289 and int(b_v.max()) < 12289
    # l2 blocks: sq = r*r, and the bits of S7 spell the sum (misc.rs:40-47, range_proofs.rs:119-129)
    r = w0[:, S[6] + 16: S[6] + 36 * n: 18]
    sq = w0[:, S[6] + 17: S[6] + 36 * n: 18]
    assert torch.equal(sq, r * r)
    nbits = 26 if logn == 9 else 27
    weights = (2 ** torch.arange(nbits, device=dev, dtype=torch.int64))
    assert torch.equal((w0[:, S[7]: S[7] + nbits] * weights).sum(dim=1), sq.sum(dim=1))
    # boolean segments really are 0/1
    assert int(w0[:, S[2]: S[3]].max()) == 1 and int(w0[:, S[2]: S[3]].min()) == 0


def test_standalone_gadget_blocks_match_gadget_oracle(engine):
    """frw_gadget: the witness block of a gadget called on its own == what the restated gadget assigns on the
    arkworks front-end simulation (oracle/falcon_gadgets.py), including out-of-range inputs (cfg(test) behaviour)."""
    import falcon_r1cs_amd as frw
    from oracle import falcon_gadgets as G
    from oracle.ark_sim import ConstraintSystem, FpVar
    rng = random.Random(8)
    q = T.Q

    def sim(fn, *vals):
        cs = ConstraintSystem()
        vs = [FpVar.new_witness(cs, v) for v in vals]
        w0 = cs.num_witness_variables()
        fn(cs, *vs)
        return cs.witness_assignment[w0:]

    qv = lambda cs: FpVar.constant(cs, q)
    cases = {
        frw.G_LESS_THAN_Q: ([42, 0, 1 << 12, 1 << 13, q - 1, q, q + 1, q * 10000] + [rng.randrange(1 << 15) for _ in range(150)],
                            lambda cs, a: G.enforce_less_than_q(cs, a)),
        frw.G_MOD_Q: ([6, 0, q, q + 1, (1 << 160) - 1, sum((1 << k) * q ** (k + 1) for k in range(11)) - 1]
                      + [rng.randrange(1 << 160) for _ in range(150)] + [rng.randrange(1 << 30) for _ in range(50)],
                      lambda cs, a: G.mod_q(cs, a, qv(cs))),
        frw.G_L2_ELEM: ([42, 0, 6143, 6144, 6145, q - 1, q] + [rng.randrange(q) for _ in range(150)],
                        lambda cs, a: G.l2_norm_var(cs, [a], qv(cs))),
        frw.G_NORM_BOUND_512: ([42, 0, 1 << 25, 34034725, 34034726, 34034727, 1 << 26, 1 << 27]
                               + [rng.randrange(1 << 27) for _ in range(150)],
                               lambda cs, a: G.enforce_less_than_norm_bound(cs, a, 9)),
        frw.G_NORM_BOUND_1024: ([42, 0, 1 << 26, 70265241, 70265242, 70265243, 1 << 27]
                                + [rng.randrange(1 << 27) for _ in range(150)],
                                lambda cs, a: G.enforce_less_than_norm_bound(cs, a, 10)),
    }
    for kind, (vals, fn) in cases.items():
        for enc in (0, 1):
            blocks, st = engine.gadget(kind, vals, encoding=enc)
            assert not st.any()
            for i, v in enumerate(vals):
                want = G.encode_elements(sim(fn, v), enc == 1)
                assert blocks[i].tobytes() == want, (kind, enc, v)
    # add_mod: [t, c, ltq(c)]
    pairs = [(6, 36), (0, 100), (100, 0), (5, q - 1)] + [(rng.randrange(1 << 30), rng.randrange(1 << 30)) for _ in range(150)] \
        + [(q - 1, (q - 1) ** 2)]
    blocks, st = engine.gadget(frw.G_ADD_MOD, [a for a, _ in pairs], [b for _, b in pairs], encoding=1)
    assert not st.any()
    for i, (a, b) in enumerate(pairs):
        want = G.encode_elements(sim(lambda cs, x, y: G.add_mod(cs, x, y, qv(cs)), a, b), True)
        assert blocks[i].tobytes() == want, (a, b)
    # ragged counts around the 64-block tile and the 256-block workgroup
    for count in (1, 63, 64, 65, 255, 257, 1000):
        vals = [rng.randrange(1 << 160) for _ in range(count)]
        blocks, st = engine.gadget(frw.G_MOD_Q, vals, encoding=0)
        for i in (0, count // 2, count - 1):
            t = sum(int(x) << (64 * k) for k, x in enumerate(blocks[i, 0]))
            assert t == vals[i] // q and int(blocks[i, 1, 0]) == vals[i] % q
    # documented domain limits are reported, not silently wrapped
    _, st = engine.gadget(frw.G_L2_ELEM, [q + 1])
    assert st.tolist() == [frw.ST_COEFF_RANGE]
    _, st = engine.gadget(frw.G_ADD_MOD, [(1 << 64) - 1], [5])
    assert st.tolist() == [frw.ST_COEFF_RANGE]


@pytest.mark.parametrize("logn", [9, 10])
def test_input_preparation_matches_codec_oracle(engine, oracle, logn):
    """decode(pk), decode(sig), hash_to_point(nonce || msg) on the GPU == the restated Falcon codec
    (SHAKE256 from hashlib), including message lengths around the SHAKE rate (136), malformed encodings, and then the
    whole chain (pk, msg, sig) -> witness == oracle witness of the decoded vectors."""
    import falcon_r1cs_amd as frw
    from oracle import falcon_codec as K
    rng = random.Random(40 + logn)
    n = 1 << logn
    lens = [0, 1, 15, 95, 96, 97, 135, 136, 137, 231, 232, 233, 1000, 5000]      # 40 + len crosses 136, 272 ...
    pks, msgs, sigs, want = [], [], [], []
    for L in lens:
        pk = [rng.randrange(T.Q) for _ in range(n)]
        s2 = [max(-2047, min(2047, round(rng.gauss(0, T.SIGMA[logn])))) for _ in range(n)]
        nonce = bytes(rng.randrange(256) for _ in range(40))
        msg = bytes(rng.randrange(256) for _ in range(L))
        pks.append(K.modq_encode(pk, logn))
        sigs.append(K.comp_encode(s2, logn, nonce))
        msgs.append(msg)
        want.append(([x % T.Q for x in s2], pk, K.hash_to_point(nonce, msg, logn)))
    # malformed encodings appended
    bad_sig = bytearray(sigs[0]); bad_sig[-1] |= 1
    bad_pk = bytearray(pks[0]); bad_pk[0] ^= 3
    big = list(want[1][1]); big[7] = 0x3FFF
    acc = 0
    for c in big:
        acc = (acc << 14) | c
    cases_pk = pks + [pks[0], bytes(bad_pk), bytes([logn]) + acc.to_bytes(14 * n // 8, "big")]
    cases_sig = sigs + [bytes(bad_sig), sigs[0], sigs[1]]
    cases_msg = msgs + [b"x", b"y", b"z"]
    sig, pk, hm, st = engine.prepare_inputs(logn, cases_pk, cases_msg, cases_sig)
    assert st.tolist() == [0] * len(lens) + [frw.ST_DECODE] * 3
    for i, (ws, wp, wh) in enumerate(want):
        assert sig[i].tolist() == ws and pk[i].tolist() == wp and hm[i].tolist() == wh, i


def test_genuine_falcon_signatures_end_to_end(engine, oracle):
    """The reference's end-to-end test (falcon_ntt.rs:133-160: keygen -> sign -> verify -> generate_constraints ->
    is_satisfied) on genuine Falcon signatures (tests/golden/falcon_signed.json, produced by oracle/falcon_sign.py and
    accepted by the specification's Verify): encoded (pk, msg, sig) -> frw_prepare_inputs on the GPU -> strict witness call
    (a genuine signature passes the norm bound, as in the reference's non-test build) -> bytes equal the oracle's and the
    fixture's -> every constraint of the independently emitted system holds on the device."""
    import hashlib
    import json
    import os
    import torch
    import falcon_r1cs_amd as frw
    cases = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "falcon_signed.json")))["cases"]
    dev = torch.device("cuda:0")
    for logn in (9, 10):
        cs = [c for c in cases if c["logn"] == logn]
        assert len(cs) == 2
        pks, msgs, sigs = ([bytes.fromhex(c[k]) for c in cs] for k in ("pk_bytes", "msg", "sig_bytes"))
        sig, pk, hm, st = engine.prepare_inputs(logn, pks, msgs, sigs)
        assert not st.any()
        for i, c in enumerate(cs):
            assert hashlib.sha256(sig[i].tobytes()).hexdigest() == c["sig_sha256"]
            assert hashlib.sha256(pk[i].tobytes()).hexdigest() == c["pk_sha256"]
            assert hashlib.sha256(hm[i].tobytes()).hexdigest() == c["hm_sha256"]
        wit, inst, stw = engine.witness_ntt_verify(logn, sig, pk, hm, frw.ENC_MONTGOMERY, strict=True)
        owit, oinst, ost = oracle.witness_ntt_verify(logn, sig, pk, hm, 1)
        assert stw.tolist() == ost.tolist() == [0, 0]
        assert np.array_equal(wit, owit) and np.array_equal(inst, oinst)
        for i, c in enumerate(cs):
            assert hashlib.sha256(wit[i].tobytes()).hexdigest() == c["witness_sha256_montgomery"]
            assert hashlib.sha256(inst[i].tobytes()).hexdigest() == c["instance_sha256_montgomery"]
        d_wit = torch.from_numpy(wit.view(np.int64)).to(dev)
        d_inst = torch.from_numpy(inst.view(np.int64)).to(dev)
        assert _r1cs_all_satisfied(engine, 0, logn, d_wit, d_inst) == 0
        # the dual circuit on the same genuine statements (falcon_dual_ntt.rs:142-169)
        dwit, dinst, dst = engine.witness_dual_ntt_verify(logn, sig, pk, hm, frw.ENC_MONTGOMERY, strict=True)
        odw, odi, ods = oracle.witness_dual_ntt_verify(logn, sig, pk, hm, 1)
        assert dst.tolist() == ods.tolist() == [0, 0] and np.array_equal(dwit, odw) and np.array_equal(dinst, odi)
        assert _r1cs_all_satisfied(engine, 1, logn, torch.from_numpy(dwit.view(np.int64)).to(dev),
                                   torch.from_numpy(dinst.view(np.int64)).to(dev)) == 0


@pytest.mark.parametrize("logn", [9, 10])
@pytest.mark.parametrize("enc", [0, 1])
def test_dual_witness_matches_oracle(engine, oracle, logn, enc):
    """falcon_dual_ntt.rs:26-132 end to end."""
    import falcon_r1cs_amd as frw
    sig, pk, hm = frw.synth_triples(logn, 5, seed=777 + logn)
    wit, inst, st = engine.witness_dual_ntt_verify(logn, sig, pk, hm, enc, strict=True)
    owit, oinst, ost = oracle.witness_dual_ntt_verify(logn, sig, pk, hm, enc)
    assert st.tolist() == ost.tolist() == [0] * 5
    assert np.array_equal(inst, oinst), _first_diff(inst, oinst)
    assert np.array_equal(wit, owit), _first_diff(wit, owit)
    # status paths: norm above the bound (permissive), coefficient out of range
    rng = random.Random(3)
    bs, bp, bh, bv = T.random_triple(logn, rng, scale=1.6)
    bad = sig[0].copy()
    bad[0] = T.Q
    S, P, H = np.stack([bs, bad]), np.stack([bp, pk[0]]), np.stack([bh, hm[0]])
    wit, inst, st = engine.witness_dual_ntt_verify(logn, S, P, H, enc, strict=False)
    owit, oinst, ost = oracle.witness_dual_ntt_verify(logn, S, P, H, enc)
    assert st.tolist() == ost.tolist() == [frw.ST_NORM_BOUND, frw.ST_COEFF_RANGE]
    assert np.array_equal(wit[0], owit[0]) and np.array_equal(inst[0], oinst[0])


def test_aggregate_mixed_batch(engine, oracle):
    """BASELINE configs[4] shape: a mixed Falcon-512 / Falcon-1024 batch; every member == its oracle witness."""
    import falcon_r1cs_amd as frw
    rng = random.Random(6)
    items = []
    for i in range(11):
        logn = rng.choice([9, 10])
        s, p, h = frw.synth_triples(logn, 1, seed=900 + i)
        items.append((logn, s[0], p[0], h[0]))
    res = engine.aggregate(items)
    assert len(res) == len(items)
    for (logn, s, p, h), (wit, inst, st) in zip(items, res):
        ow, oi, ost = oracle.witness_ntt_verify(logn, s, p, h, 1)
        assert st == 0 and np.array_equal(wit, ow[0]) and np.array_equal(inst, oi[0])


def test_full_launch_sample_digests_and_concurrent_streams(engine, oracle):
    """One full-size launch (4096 Falcon-1024 signatures, work-queue scheduling under load): every 64th witness is
    compared with the oracle by digest; and two launches running concurrently on two streams of one context (each
    stream has its own queue head) produce the same digests as when run alone."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    logn, batch = 10, 4096
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=31337)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    dig = torch.zeros(batch, dtype=torch.int64, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    engine.digest_dev(wit, L.num_witness * 4, batch, dig, s0)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0
    got = dig.cpu().numpy().view(np.uint64)
    idx = list(range(0, batch, 64)) + [batch - 1]
    owit, _, _ = oracle.witness_ntt_verify(logn, sig[idx], pk[idx], hm[idx], 1, threads=8)
    assert [int(got[i]) for i in idx] == [oracle.digest(owit[j]) for j in range(len(idx))]
    # two halves concurrently on two side streams
    half = batch // 2
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    wit.zero_()
    dig2 = torch.zeros(batch, dtype=torch.int64, device=dev)
    torch.cuda.synchronize()
    engine.witness_ntt_verify_dev(logn, half, d[0][:half], d[1][:half], d[2][:half], wit[:half], inst[:half], st[:half], 1,
                                  sa.cuda_stream)
    engine.witness_ntt_verify_dev(logn, half, d[0][half:], d[1][half:], d[2][half:], wit[half:], inst[half:], st[half:], 1,
                                  sb.cuda_stream)
    torch.cuda.synchronize()
    engine.digest_dev(wit, L.num_witness * 4, batch, dig2, s0)
    torch.cuda.synchronize()
    assert torch.equal(dig, dig2)


def test_device_entry_point_is_graph_capturable(engine, oracle):
    """The _dev entry point is a memset node + a kernel node: captured into a hipGraph and replayed on fresh inputs."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    logn, batch = 9, 96
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=555)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.zeros((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.zeros((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    dig = torch.zeros(batch, dtype=torch.int64, device=dev)
    side = torch.cuda.Stream()
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, stream=side):
        engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, side.cuda_stream)
    # new inputs in the same buffers, then replay
    sig2, pk2, hm2 = frw.synth_triples(logn, batch, seed=556)
    for t, a in zip(d, (sig2, pk2, hm2)):
        t.copy_(torch.from_numpy(a.view(np.int16)))
    graph.replay()
    torch.cuda.synchronize()
    engine.digest_dev(wit, L.num_witness * 4, batch, dig, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0
    owit, _, _ = oracle.witness_ntt_verify(logn, sig2, pk2, hm2, 1, threads=8)
    assert [int(x) for x in dig.cpu().numpy().view(np.uint64)] == [oracle.digest(owit[i]) for i in range(batch)]


def test_empty_batches_and_bad_arguments(engine):
    """Empty input is a no-op (FRW_OK), malformed arguments are rejected before anything is launched."""
    import ctypes as C
    import falcon_r1cs_amd as frw
    lib, ctx = engine._lib, engine._ctx
    for logn in (9, 10):
        n = 1 << logn
        empty = np.zeros((0, n), dtype=np.uint16)
        wit, inst, st = engine.witness_ntt_verify(logn, empty, empty, empty)
        assert wit.shape[0] == inst.shape[0] == st.shape[0] == 0
        w, o, s_ = engine.ntt_modq(logn, empty)
        assert w.shape[0] == 0
    assert lib.frw_witness_ntt_verify_dev(ctx, 10, 0, None, None, None, 1, None, None, None, None) == 0
    assert lib.frw_witness_ntt_verify_dev(ctx, 10, 4, None, None, None, 1, None, None, None, None) == -1
    assert lib.frw_witness_ntt_verify_dev(ctx, 11, 4, None, None, None, 1, None, None, None, None) == -1
    assert lib.frw_witness_ntt_verify_dev(ctx, 10, 4, None, None, None, 2, None, None, None, None) == -1
    assert lib.frw_gadget(ctx, 99, 1, None, None, 1, None, None) == -1
    blocks, st = engine.gadget(frw.G_MOD_Q, [])
    assert blocks.shape[0] == 0
    with pytest.raises(ValueError):
        engine.gadget(frw.G_MOD_Q, [1 << 161])


@pytest.mark.parametrize("circuit,logn,batch", [(0, 9, 512), (0, 10, 4096), (1, 10, 256)])
def test_whole_launch_satisfies_independent_r1cs_on_device(engine, circuit, logn, batch):
    """falcon_ntt.rs:159 / falcon_dual_ntt.rs:168 `assert!(cs.is_satisfied())` for EVERY signature of a launch, on the
    device, against matrices emitted by the C++ host mirror from the gadget definitions (not from the kernels' closed
    form).  Then targeted corruptions must be caught, and only in the signatures that were touched."""
    import time
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    L = frw.layout_dual(logn) if circuit else frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=4242 + circuit)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    bad = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    launch = engine.witness_dual_ntt_verify_dev if circuit else engine.witness_ntt_verify_dev
    launch(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    h = engine.r1cs_load(circuit, logn)
    try:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        engine.r1cs_check_dev(h, batch, wit, inst, bad, s0)
        torch.cuda.synchronize()
        print("r1cs check of %d witnesses: %.3f s" % (batch, time.perf_counter() - t0))
        assert int((st != 0).sum()) == 0
        assert int(bad.abs().sum()) == 0
        # corruptions: a boolean in S2/first segment area, a quotient t, a public input
        n = L.n
        targets = {3: (2 * n + 40, 0), 7: (L.seg_off[8 if circuit else 3], 0), batch - 1: (5, 1)}
        for s_, (elem, limb) in targets.items():
            wit[s_, elem, limb] += 1
        inst[11, 4, 0] += 1
        engine.r1cs_check_dev(h, batch, wit, inst, bad, s0)
        torch.cuda.synchronize()
        flagged = set(torch.nonzero(bad).flatten().tolist())
        assert flagged == set(targets) | {11}
    finally:
        engine.r1cs_free(h)


def test_config5_shape_mixed_1024_signatures(engine, oracle):
    """BASELINE configs[4] shape: 1,024 signatures, Falcon-512 and Falcon-1024 mixed (parameter set drawn from the
    seed), grouped by logn into two device launches; every witness must satisfy its circuit's constraint system
    (checked on the device) and a sample must equal the oracle's witness by digest."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    rng = random.Random(2025)
    logns = [rng.choice([9, 10]) for _ in range(1024)]
    s0 = torch.cuda.current_stream().cuda_stream
    for logn in (9, 10):
        idx = [i for i, l in enumerate(logns) if l == logn]
        batch = len(idx)
        assert batch > 400
        L = frw.layout(logn)
        sig = np.empty((batch, L.n), dtype=np.uint16); pk = np.empty_like(sig); hm = np.empty_like(sig)
        for j, i in enumerate(idx):                                  # stream = global signature index
            s, p, h = frw.synth_triples(logn, 1, seed=77, first_index=i)
            sig[j], pk[j], hm[j] = s[0], p[0], h[0]
        d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
        wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        bad = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        dig = torch.zeros(batch, dtype=torch.int64, device=dev)
        engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
        engine.digest_dev(wit, L.num_witness * 4, batch, dig, s0)
        h = engine.r1cs_load(0, logn)
        try:
            engine.r1cs_check_dev(h, batch, wit, inst, bad, s0)
            torch.cuda.synchronize()
        finally:
            engine.r1cs_free(h)
        assert int((st != 0).sum()) == 0 and int(bad.abs().sum()) == 0
        sample = list(range(0, batch, 37))
        owit, _, _ = oracle.witness_ntt_verify(logn, sig[sample], pk[sample], hm[sample], 1, threads=8)
        got = dig.cpu().numpy().view(np.uint64)
        assert [int(got[j]) for j in sample] == [oracle.digest(owit[k]) for k in range(len(sample))]


def test_r1cs_matrix_vector_products_match_oracle(engine, oracle, tmp_path):
    """frw_r1cs_eval_dev: A z, B z, C z per signature on the device == the oracle's matrices (independent inlining in
    oracle/ark_sim.py) applied to the oracle's witness with Python integers."""
    import torch
    import falcon_r1cs_amd as frw
    from oracle import falcon_gadgets as G
    dev = torch.device("cuda:0")
    logn, batch = 9, 3
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=8080)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    bad = torch.empty(batch, dtype=torch.int32, device=dev)
    abc = torch.zeros((batch, 3, L.num_constraints, 4), dtype=torch.int64, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    h = engine.r1cs_load(0, logn)
    try:
        engine.r1cs_eval_dev(h, batch, wit, inst, bad, abc, s0)
        torch.cuda.synchronize()
    finally:
        engine.r1cs_free(h)
    assert int(bad.abs().sum()) == 0
    P = G.P_BLS12_381_FR
    rinv = pow(G.R_MONT, -1, P)
    k = 1                                                         # check the middle signature completely
    cs = G.run_reference_flow(sig[k].tolist(), pk[k].tolist(), hm[k].tolist(), logn, strict=True)
    z = cs.instance_assignment + cs.witness_assignment
    got = abc[k].cpu().numpy().view(np.uint64)
    to_int = lambda limbs: (int(limbs[0]) | int(limbs[1]) << 64 | int(limbs[2]) << 128 | int(limbs[3]) << 192) * rinv % P
    for m, rows in enumerate(cs.to_matrices()):
        for i in list(range(0, len(rows), 997)) + [29 * 512 * 0 + 27 * 512 + 29 * 5, len(rows) - 1]:   # sample incl. a dense ladder row
            want = sum(c * z[col] for col, c in rows[i]) % P
            assert to_int(got[m, i]) == want, (m, i)


def test_hip_matches_dual_and_prepare_golden(engine):
    """Committed fixtures of the widened rows: dual-NTT witness digests (tests/golden/dual_*.json) and input
    preparation digests (tests/golden/prepare.json)."""
    import glob
    import hashlib
    import json
    import os
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for path in sorted(glob.glob(os.path.join(gold, "dual_*.json"))):
        fx = json.load(open(path))
        sig, pk, hm = (np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16) for k in ("sig", "pk", "hm"))
        for enc, name in ((0, "canonical"), (1, "montgomery")):
            wit, inst, st = engine.witness_dual_ntt_verify(fx["logn"], sig, pk, hm, enc, strict=True)
            assert hashlib.sha256(wit.tobytes()).hexdigest() == fx["witness_sha256"][name]
            assert hashlib.sha256(inst.tobytes()).hexdigest() == fx["instance_sha256"][name]
    fx = json.load(open(os.path.join(gold, "prepare.json")))
    import falcon_r1cs_amd as frw
    for logn in (9, 10):
        hashes = [c for c in fx["cases"] if c["logn"] == logn and "nonce" in c]
        codec = [c for c in fx["cases"] if c["logn"] == logn and "pk_bytes" in c][0]
        pkb, sgb = bytes.fromhex(codec["pk_bytes"]), bytes.fromhex(codec["sig_bytes"])
        # the hash cases reuse the one encoded key/signature but need their own nonce: re-stamp the signature's nonce
        sigs = [sgb[:1] + bytes.fromhex(c["nonce"]) + sgb[41:] for c in hashes]
        s, p, h, st = engine.prepare_inputs(logn, [pkb] * len(hashes), [bytes.fromhex(c["msg"]) for c in hashes], sigs)
        assert not st.any()
        for i, c in enumerate(hashes):
            assert h[i, :8].tolist() == c["hm_first8"]
            assert hashlib.sha256(h[i].tobytes()).hexdigest() == c["hm_sha256"]
        assert hashlib.sha256(p[0].tobytes()).hexdigest() == codec["pk_sha256"]
        assert hashlib.sha256(s[0].tobytes()).hexdigest() == codec["sig_sha256"]


# ---------------------------------------------------------------------------------------------------------------------
# round 2: the launch shapes bench.py times, and the rejection path under load
# ---------------------------------------------------------------------------------------------------------------------
def _launch_and_digest(engine, logn, sig, pk, hm, dual=False, enc=1):
    """One device launch over the whole batch; returns (wit, inst, status, digests as python ints)."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    L = frw.layout_dual(logn) if dual else frw.layout(logn)
    batch = sig.shape[0]
    d = [torch.from_numpy(np.ascontiguousarray(a).view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    wit.fill_(0x5A5A5A5A5A5A5A5A)                      # stale memory must never survive a launch
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    dig = torch.zeros(batch, dtype=torch.int64, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    launch = engine.witness_dual_ntt_verify_dev if dual else engine.witness_ntt_verify_dev
    launch(logn, batch, d[0], d[1], d[2], wit, inst, st, enc, s0)
    engine.digest_dev(wit, L.num_witness * 4, batch, dig, s0)
    torch.cuda.synchronize()
    return wit, inst, st, [int(x) for x in dig.cpu().numpy().view(np.uint64)]


def _r1cs_all_satisfied(engine, circuit, logn, wit, inst):
    import torch
    batch = wit.shape[0]
    bad = torch.full((batch,), -1, dtype=torch.int32, device=wit.device)
    h = engine.r1cs_load(circuit, logn)
    try:
        engine.r1cs_check_dev(h, batch, wit, inst, bad, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
    finally:
        engine.r1cs_free(h)
    return int((bad != 0).sum().item())


@pytest.mark.parametrize("logn,batch", [(10, 32768), (9, 8192)])
def test_benchmark_launch_shape_is_checked(engine, oracle, logn, batch):
    """The launch shapes bench.py times (BASELINE configs[2]: 32,768 Falcon-1024 signatures per launch = 3,072 workgroups
    -- 4 x the 768 resident ones -- x 10 full rounds of static striding + 2,048 signatures cut into five work items each;
    and the Falcon-512 launch of its `secondary` block, 8,192 signatures = 2,048 workgroups x 4 rounds, no split tail;
    frw.MI355X_BENCH_LAUNCH_SHAPES, which bench.py asserts its own launches against):
    every witness of the launch satisfies the independently emitted constraint system on the device
    (falcon_ntt.rs:159), a strided sample of 256 + the last one equals the oracle's witness by digest, all statuses OK."""
    import falcon_r1cs_amd as frw
    shape = engine.launch_shape(logn, batch)
    assert (shape["grid"], shape["split_signatures"]) == frw.MI355X_BENCH_LAUNCH_SHAPES[(logn, batch)]
    assert (shape["cus"], shape["resident_per_cu"]) == (256, 3)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=0xBE7C4 + logn)
    wit, inst, st, dig = _launch_and_digest(engine, logn, sig, pk, hm)
    assert int((st != 0).sum()) == 0
    assert len(set(dig)) == batch
    assert _r1cs_all_satisfied(engine, 0, logn, wit, inst) == 0
    idx = list(range(0, batch, batch // 256)) + [batch - 1]
    for lo in range(0, len(idx), 64):
        part = idx[lo:lo + 64]
        owit, oinst, _ = oracle.witness_ntt_verify(logn, sig[part], pk[part], hm[part], 1, threads=8)
        assert [dig[i] for i in part] == [oracle.digest(owit[j]) for j in range(len(part))]
        got_inst = inst[part].cpu().numpy().view(np.uint64)
        assert np.array_equal(got_inst, oinst)


def test_config2_full_size_ntt_modq_all_digests(engine, oracle):
    """BASELINE configs[1] at its stated size: 4,096 Falcon-512 polynomials through frw_ntt_modq_dev in ONE launch,
    EVERY witness block compared with the oracle by digest and every reduced NTT output compared exactly."""
    import torch
    dev = torch.device("cuda:0")
    logn, batch = 9, 4096
    n = 1 << logn
    rng = np.random.default_rng(20262)
    poly = rng.integers(0, T.Q, size=(batch, n), dtype=np.uint16)
    d_poly = torch.from_numpy(poly.view(np.int16)).to(dev)
    wit = torch.empty((batch, 29 * n, 4), dtype=torch.int64, device=dev)
    out = torch.empty((batch, n), dtype=torch.int16, device=dev)
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    dig = torch.zeros(batch, dtype=torch.int64, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    engine.ntt_modq_dev(logn, batch, d_poly, wit, out, st, 1, s0)
    engine.digest_dev(wit, 29 * n * 4, batch, dig, s0)
    torch.cuda.synchronize()
    assert int((st != 0).sum()) == 0
    got = [int(x) for x in dig.cpu().numpy().view(np.uint64)]
    got_out = out.cpu().numpy().view(np.uint16)
    for lo in range(0, batch, 512):
        owit, oout = oracle.ntt_modq(logn, poly[lo:lo + 512], 1)
        assert np.array_equal(got_out[lo:lo + 512], oout)
        assert got[lo:lo + 512] == [oracle.digest(owit[i]) for i in range(owit.shape[0])]


@pytest.mark.parametrize("dual", [False, True])
@pytest.mark.parametrize("logn", [9, 10])
def test_rejected_signatures_scattered_through_a_large_batch(engine, oracle, logn, dual):
    """range_proofs.rs:57-60 under load: ~30 % of a batch much larger than the persistent grid carries a coefficient >= q
    at a random place in sig, pk or hm, so every workgroup meets rejected items followed by good ones (static striding
    over whole signatures, then the split tail).  Statuses equal the oracle's, every accepted witness and instance is
    digest-equal to the oracle's, every rejected slot is zero-filled."""
    import falcon_r1cs_amd as frw
    batch = 2400 if not (dual and logn == 10) else 1536
    n = 1 << logn
    sig, pk, hm = frw.synth_triples(logn, batch, seed=0xBAD + logn + 7 * dual)
    rng = np.random.default_rng(1000 + logn + dual)
    bad = rng.random(batch) < 0.3
    bad[:3] = [True, False, True]
    bad[-2:] = [True, False]
    for i in np.nonzero(bad)[0]:
        arr = (sig, pk, hm)[rng.integers(3)]
        arr[i, rng.integers(n)] = rng.choice([T.Q, T.Q + 1, 0x3FFF, 0xFFFF])
    wit, inst, st, dig = _launch_and_digest(engine, logn, sig, pk, hm, dual=dual)
    st = st.cpu().numpy()
    assert np.array_equal(st != 0, bad) and set(st[bad].tolist()) == {frw.ST_COEFF_RANGE}
    L = frw.layout_dual(logn) if dual else frw.layout(logn)
    zero_digest = oracle.digest(np.zeros((L.num_witness, 4), dtype=np.uint64))
    fn = oracle.witness_dual_ntt_verify if dual else oracle.witness_ntt_verify
    step = 128
    for lo in range(0, batch, step):
        sl = slice(lo, lo + step)
        if dual:
            owit, oinst, ost = fn(logn, sig[sl], pk[sl], hm[sl], 1)
        else:
            owit, oinst, ost = fn(logn, sig[sl], pk[sl], hm[sl], 1, threads=8)
        assert np.array_equal(ost, st[sl])
        want = [oracle.digest(owit[j]) for j in range(owit.shape[0])]
        assert dig[lo:lo + step] == want
        assert all(want[j] == zero_digest for j in np.nonzero(bad[sl])[0])
        assert np.array_equal(inst[sl].cpu().numpy().view(np.uint64), oinst)


@pytest.mark.parametrize("logn", [9, 10])
def test_compact_encoding_matches_relayout_of_oracle_witness(engine, oracle, logn):
    """FRW_ENC_COMPACT straight from the kernel == the oracle's arkworks witness re-laid out on the host (values as plain
    integers in witness order, booleans as a bit array, instance without the leading one, status word, zero padding), byte for
    byte; a rejected signature is all zeros but for its status word; and frw_expand_dev / frw_expand_host turn the compact buffer
    back into exactly the bytes of the direct launch -- the oracle's witness / instance, zeros for the rejected signature."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    CL, L = frw.compact_layout(logn), frw.layout(logn)
    batch = 9
    sig, pk, hm = frw.synth_triples(logn, batch, seed=0xC0 + logn)
    big = T.random_triple(logn, random.Random(12), scale=1.6)            # norm above the bound: truncated bits, status 2
    sig[4], pk[4], hm[4] = big[0], big[1], big[2]
    sig[7, 5] = T.Q                                                       # rejected
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    comp = torch.full((batch, CL.bytes_per_signature), 0x5A, dtype=torch.uint8, device=dev)
    st = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    engine.witness_ntt_verify_compact_dev(logn, batch, d[0], d[1], d[2], comp, st, s0)
    engine.expand_dev(logn, batch, comp, wit, inst, s0)
    torch.cuda.synchronize()
    owit, oinst, ost = oracle.witness_ntt_verify(logn, sig, pk, hm, 1)
    assert st.cpu().numpy().tolist() == ost.tolist() == [0, 0, 0, 0, 2, 0, 0, 1, 0]
    got = comp.cpu().numpy()
    # the direct (FRW_ENC_MONTGOMERY) launch over the same batch: what an expansion must reproduce, rejected slot included
    dwit, dinst = torch.full_like(wit, 0x77), torch.full_like(inst, 0x77)
    dst = torch.full((batch,), -1, dtype=torch.int32, device=dev)
    engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], dwit, dinst, dst, 1, s0)
    torch.cuda.synchronize()
    assert torch.equal(dst, st)
    for i in range(batch):
        if ost[i] == 1:
            # all zeros but for the status word (the record itself says "rejected"); the expansion is all zeros too --
            # no leading one in the instance vector -- exactly like the direct output
            assert np.frombuffer(got[i, CL.status_off:CL.status_off + 4].tobytes(), dtype=np.uint32)[0] == 1
            rest = got[i].copy()
            rest[CL.status_off:CL.status_off + 4] = 0
            assert not rest.any()
            assert not wit[i].any() and not inst[i].any()
            assert torch.equal(wit[i], dwit[i]) and torch.equal(inst[i], dinst[i])
            continue
        # the buffer was pre-filled with 0x5A: the producer writes EVERY byte of a record, padding included
        want = np.frombuffer(T.compact_from_witness(logn, owit[i], oinst[i], CL, status=int(ost[i])), dtype=np.uint8)
        assert np.array_equal(got[i], want), i
        assert np.array_equal(wit[i].cpu().numpy().view(np.uint64), owit[i]), i
        assert np.array_equal(inst[i].cpu().numpy().view(np.uint64), oinst[i]), i
    assert torch.equal(wit, dwit) and torch.equal(inst, dinst)
    # the same through host memory: frw_expand_host of the records == the direct output, rejected slot included
    hw, hi = engine.expand_host(logn, got)
    assert np.array_equal(hw, dwit.cpu().numpy().view(np.uint64)) and np.array_equal(hi, dinst.cpu().numpy().view(np.uint64))


def test_expand_of_compact_equals_direct_output_full_launch(engine):
    """expand(compact(x)) == the FRW_ENC_MONTGOMERY launch, digest by digest, for a full 4,096-signature Falcon-1024
    launch and a 2,048-signature Falcon-512 one (plus the committed golden fixtures through the compact path)."""
    import glob
    import hashlib
    import json
    import os
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    s0 = torch.cuda.current_stream().cuda_stream
    for logn, batch in ((10, 4096), (9, 2048)):
        CL, L = frw.compact_layout(logn), frw.layout(logn)
        sig, pk, hm = frw.synth_triples(logn, batch, seed=0xE0 + logn)
        wit, inst, st, dig = _launch_and_digest(engine, logn, sig, pk, hm)
        d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
        comp = torch.empty((batch, CL.bytes_per_signature), dtype=torch.uint8, device=dev)
        st2 = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        wit2 = torch.full_like(wit, 0x33)
        inst2 = torch.full_like(inst, 0x33)
        dig2 = torch.zeros(batch, dtype=torch.int64, device=dev)
        engine.witness_ntt_verify_compact_dev(logn, batch, d[0], d[1], d[2], comp, st2, s0)
        engine.expand_dev(logn, batch, comp, wit2, inst2, s0)
        engine.digest_dev(wit2, L.num_witness * 4, batch, dig2, s0)
        torch.cuda.synchronize()
        assert int((st2 != 0).sum()) == 0
        assert [int(x) for x in dig2.cpu().numpy().view(np.uint64)] == dig
        assert torch.equal(inst, inst2)
        assert torch.equal(wit[::257], wit2[::257])
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for path in sorted(glob.glob(os.path.join(gold, "witness_*.json"))):
        fx = json.load(open(path))
        logn = fx["logn"]
        CL, L = frw.compact_layout(logn), frw.layout(logn)
        arrs = [np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16).reshape(1, -1) for k in ("sig", "pk", "hm")]
        d = [torch.from_numpy(a.view(np.int16).copy()).to(dev) for a in arrs]
        comp = torch.full((1, CL.bytes_per_signature), 0xA5, dtype=torch.uint8, device=dev)
        st = torch.zeros(1, dtype=torch.int32, device=dev)
        wit = torch.empty((1, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((1, L.num_instance, 4), dtype=torch.int64, device=dev)
        engine.witness_ntt_verify_compact_dev(logn, 1, d[0], d[1], d[2], comp, st, s0)
        engine.expand_dev(logn, 1, comp, wit, inst, s0)
        torch.cuda.synchronize()
        assert hashlib.sha256(wit.cpu().numpy().tobytes()).hexdigest() == fx["witness_sha256"]["montgomery"]
        assert hashlib.sha256(inst.cpu().numpy().tobytes()).hexdigest() == fx["instance_sha256"]["montgomery"]
        # the compact bytes themselves, as the kernel wrote them (every byte of a record is written)
        got = comp[0].cpu().numpy().copy()
        assert len(got) == fx["compact_bytes"] and hashlib.sha256(got.tobytes()).hexdigest() == fx["compact_sha256"]


@pytest.mark.parametrize("logn", [9, 10])
def test_compact_host_path_and_host_expansion(engine, oracle, logn):
    """frw_witness_ntt_verify(..., FRW_ENC_COMPACT, ...) through host buffers (more signatures than one pipeline chunk,
    so both device buffers and both streams are in play), expanded on the host: == the oracle's witness for a sample,
    == the direct host path for all statuses."""
    import falcon_r1cs_amd as frw
    batch = 2300
    sig, pk, hm = frw.synth_triples(logn, batch, seed=0x40 + logn)
    sig[1234, 7] = T.Q
    comp, st = engine.witness_ntt_verify_compact(logn, sig, pk, hm, strict=False)
    CL = frw.compact_layout(logn)
    assert st[1234] == frw.ST_COEFF_RANGE and int((st != 0).sum()) == 1
    rejected = comp[1234].copy()                                      # all zeros but for the status word it carries
    assert rejected[CL.status_off:CL.status_off + 4].view(np.uint32)[0] == frw.ST_COEFF_RANGE
    rejected[CL.status_off:CL.status_off + 4] = 0
    assert not rejected.any()
    assert not comp[:, CL.status_off:CL.status_off + 4].view(np.uint32)[np.arange(batch) != 1234].any()
    idx = [0, 1, 255, 256, 2047, 2048, 2049, 2299]
    wit, inst = engine.expand_host(logn, comp[idx])
    owit, oinst, ost = oracle.witness_ntt_verify(logn, sig[idx], pk[idx], hm[idx], 1)
    assert not ost.any() and np.array_equal(wit, owit) and np.array_equal(inst, oinst)
    wz, iz = engine.expand_host(logn, comp[1234:1235])                # expands to zeros, instance vector included
    assert not wz.any() and not iz.any()
    with pytest.raises(frw.FrwError):
        engine.witness_ntt_verify_compact(logn, sig, pk, hm, strict=True)


@pytest.mark.parametrize("logn", [9, 10])
def test_expand_dev_equals_expand_host_on_random_compact_buffers(engine, logn):
    """frw_expand_dev vs frw_expand_host (itself checked against Python integers in tests/test_capi.py) on compact buffers
    filled with random integers of the full documented ranges -- 32-bit values, 160-bit quotients, random bits -- i.e. far
    outside what a Falcon witness ever holds: the two expansions are the same function."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    CL, L = frw.compact_layout(logn), frw.layout(logn)
    batch = 5
    rng = np.random.default_rng(77 + logn)
    comp = rng.integers(0, 256, size=(batch, CL.bytes_per_signature), dtype=np.uint8)
    nb = 50 if logn == 9 else 52
    for i in range(batch):                                   # the two words of the norm-bound block carry nb bits only
        w = comp[i, CL.bits_off: CL.bits_off + 4 * CL.num_bit_words].view(np.uint32)
        tail = (int(w[-2]) | int(w[-1]) << 32) & ((1 << nb) - 1)
        w[-2], w[-1] = tail & 0xFFFFFFFF, tail >> 32
    hw, hi = engine.expand_host(logn, comp)
    d_comp = torch.from_numpy(comp).to(dev)
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    engine.expand_dev(logn, batch, d_comp, wit, inst, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(wit.cpu().numpy().view(np.uint64), hw)
    assert np.array_equal(inst.cpu().numpy().view(np.uint64), hi)


@pytest.mark.gpu
def test_one_signature_per_call_like_the_reference_and_no_allocation_after_the_first(oracle):
    """The reference's consumers call generate_constraints once per signature (examples/constraint_counts.rs:61-63,
    pok_sig.rs:24-32): the host-buffer entry point with batch = 1, again and again, interleaved with other shapes (a
    long two-slot batch, the dual circuit, ntt_circuit alone, the compact form).  Every call == the oracle; the context's
    working memory grows while new shapes arrive and then never again."""
    import falcon_r1cs_amd as frw
    eng = frw.WitnessEngine(0)                     # a context of its own: the counter starts at zero
    assert eng.host_allocations() == 0
    sig, pk, hm = frw.synth_triples(10, 300, seed=4242)
    want = oracle.witness_ntt_verify(10, sig, pk, hm, 1)

    def one_round():
        for i in (0, 1, 299):
            w, ins, st = eng.witness_ntt_verify(10, sig[i:i + 1], pk[i:i + 1], hm[i:i + 1])
            assert np.array_equal(w[0], want[0][i]) and np.array_equal(ins[0], want[1][i]) and st[0] == 0
        w, ins, st = eng.witness_ntt_verify(10, sig, pk, hm)                 # 300 > 256: two slots, two streams
        assert np.array_equal(w, want[0]) and np.array_equal(ins, want[1]) and not st.any()
        comp, st = eng.witness_ntt_verify_compact(10, sig[:3], pk[:3], hm[:3])
        w2, i2 = eng.expand_host(10, comp)
        assert np.array_equal(w2, want[0][:3]) and np.array_equal(i2, want[1][:3])
        wd, _, st = eng.witness_dual_ntt_verify(9, sig[:2, :512], pk[:2, :512], hm[:2, :512], strict=False)
        od = oracle.witness_dual_ntt_verify(9, sig[:2, :512], pk[:2, :512], hm[:2, :512], 1)
        assert np.array_equal(wd, od[0]) and np.array_equal(st, od[2])
        wn, on, st = eng.ntt_modq(9, sig[:5, :512])
        ow, oo = oracle.ntt_modq(9, sig[:5, :512], 1)[:2]
        assert np.array_equal(wn, ow) and np.array_equal(on, oo)
    one_round()
    grown = eng.host_allocations()
    assert 1 <= grown <= 8
    for _ in range(3):
        one_round()
    assert eng.host_allocations() == grown, "a host-buffer call allocated after every shape had been seen"
    eng.trim()
    w, ins, st = eng.witness_ntt_verify(10, sig[:1], pk[:1], hm[:1])         # usable after a trim
    assert np.array_equal(w[0], want[0][0]) and eng.host_allocations() > grown
    eng.close()


@pytest.mark.gpu
def test_r1cs_eval_with_caller_scratch_equals_the_allocating_entry_points_and_is_capturable(engine):
    """frw_r1cs_eval_scratch_dev: the same counts and the same A z, B z, C z as frw_r1cs_check_dev / frw_r1cs_eval_dev, from
    the caller's scratch -- nothing allocated, so the call can sit inside a captured HIP graph and be replayed."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    logn, batch = 9, 96
    L = frw.layout(logn)
    sig, pk, hm = frw.synth_triples(logn, batch, seed=909)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, 0)
    wit[5, 2 * L.n + 3, 0] += 1                                   # one violated signature
    wit[77, L.seg_off[3], 0] += 1
    h = engine.r1cs_load(0, logn)
    try:
        want_bad = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        want_abc = torch.zeros((batch, 3, L.num_constraints, 4), dtype=torch.int64, device=dev)
        engine.r1cs_eval_dev(h, batch, wit, inst, want_bad, want_abc, 0)
        torch.cuda.synchronize()
        assert set(torch.nonzero(want_bad).flatten().tolist()) == {5, 77}
        for with_abc in (False, True):
            need = engine.r1cs_eval_scratch_bytes(h, batch, with_abc)
            assert need > 0
            scratch = torch.empty(need, dtype=torch.uint8, device=dev)
            bad = torch.full((batch,), -1, dtype=torch.int32, device=dev)
            abc = torch.zeros_like(want_abc) if with_abc else None
            with pytest.raises(frw.FrwError):
                engine.r1cs_eval_scratch_dev(h, batch, wit, inst, bad, abc, scratch, need - 1, 0)
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                engine.r1cs_eval_scratch_dev(h, batch, wit, inst, bad, abc, scratch, need, side.cuda_stream)   # warm-up
            side.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                engine.r1cs_eval_scratch_dev(h, batch, wit, inst, bad, abc, scratch, need, side.cuda_stream)
            bad.fill_(-1)
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(bad, want_bad)
            if with_abc:
                assert torch.equal(abc, want_abc)
    finally:
        engine.r1cs_free(h)
