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
