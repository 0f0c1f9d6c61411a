"""GPU parity of the R1CS -> QAP witness map (frw_qap_witness_map_dev) against the oracle (oracle/qap_oracle.c, which
tests/test_qap.py pins to oracle/qap.py and to the FFT-free identity).  Bit-exact: h is a vector of field elements."""
import random

import numpy as np
import pytest

import frw_testlib as T
from oracle import qap

pytestmark = pytest.mark.gpu
P = qap.P
R_INV = pow(qap.R_MONT, -1, P)


def from_montgomery(limbs):
    return [v * R_INV % P for v in T.limbs_to_ints(limbs)]


def oracle_h(oracle, mats, ni, z_limbs):
    az, bz, cz = (oracle.qap_matvec(*m, z_limbs) for m in mats)
    return (az, bz, cz), oracle.qap_witness_map(az, bz, cz, ni, z_limbs)


@pytest.mark.parametrize("circuit,logn", [(0, 9), (0, 10), (1, 9), (1, 10)])
def test_witness_map_equals_oracle(engine, oracle, tmp_path, circuit, logn):
    import torch
    import falcon_r1cs_amd as frw
    from test_r1cs_export import export, read_r1cs
    dev = torch.device("cuda:0")
    batch = 3
    sig, pk, hm = frw.synth_triples(logn, batch, seed=4242 + logn)
    L = frw.layout_dual(logn) if circuit else frw.layout(logn)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    gen = engine.witness_dual_ntt_verify_dev if circuit else engine.witness_ntt_verify_dev
    gen(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    # signature 2: break the witness (one S1 element + 1) -- h is still a deterministic function of (matrices, z)
    tamper = L.n + 7
    one_m = torch.tensor(np.array(T.ints_to_limbs([qap.R_MONT])[0]).view(np.int64), device=dev)
    w2 = from_montgomery(wit[2, tamper].cpu().numpy().view(np.uint64))[0]
    wit[2, tamper] = torch.from_numpy(T.ints_to_limbs([(w2 + 1) * qap.R_MONT % P])[0].view(np.int64)).to(dev)
    del one_m
    # signature 1: a ladder input (S0 element 3, read by every long row) replaced by a full-size field element: the long
    # rows' small-integer path must fall back to field products for that term
    big = 0x5A5A1234_9E3779B9_7F4A7C15_F39CC060_5CEDC834_1082276B_F3A27251_F86C6A11 % P
    wit[1, 3] = torch.from_numpy(T.ints_to_limbs([big * qap.R_MONT % P])[0].view(np.int64)).to(dev)
    h_r1cs = engine.r1cs_load(circuit, logn)
    try:
        q = engine.qap_info(h_r1cs)
        n = int(q.domain_size)
        assert n == 1 << int(q.log_domain_size) and n >= q.num_constraints + q.num_instance > n // 2
        assert q.num_constraints == L.num_constraints and q.num_instance == L.num_instance
        ws_bytes = 2 * int(q.workspace_bytes_per_signature)              # two signatures per chunk: 3 = 2 + 1
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        h = torch.full((batch, n, 4), -1, dtype=torch.int64, device=dev)
        bad = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        engine.qap_witness_map_dev(h_r1cs, batch, wit, inst, h, ws, ws_bytes, bad, s0)
        # the six-transform quotient on the same batch
        hq = torch.full((batch, n, 4), -1, dtype=torch.int64, device=dev)
        badq = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        engine.qap_quotient_dev(h_r1cs, batch, wit, inst, hq, ws, ws_bytes, badq, s0)
        # the witness map without the caller's array for the counts: it keeps its own (it needs them to know which signatures go
        # through the seven-transform map after the six-transform one)
        h_again = torch.full((batch, n, 4), -1, dtype=torch.int64, device=dev)
        engine.qap_witness_map_dev(h_r1cs, batch, wit, inst, h_again, ws, ws_bytes, None, s0)
        torch.cuda.synchronize()
        assert torch.equal(h_again, h)
    finally:
        engine.r1cs_free(h_r1cs)
    assert bad.tolist()[0] == 0 and bad.tolist()[1] > 0 and bad.tolist()[2] > 0
    assert badq.tolist() == bad.tolist()
    assert torch.equal(hq[0], h[0]), "satisfied witness: the six-transform quotient must equal the witness map"
    assert not torch.equal(hq[1], h[1]) and not torch.equal(hq[2], h[2])
    path = tmp_path / "c.r1cs"
    export(circuit, logn, path)
    ni, nw, nc, mats = read_r1cs(path)
    ofn = oracle.witness_dual_ntt_verify if circuit else oracle.witness_ntt_verify
    owit, oinst, ost = ofn(logn, sig, pk, hm, 0)
    for k in (0, 1, 2):
        z = np.concatenate([oinst[k], owit[k]])
        if k == 1:
            z[ni + 3] = T.ints_to_limbs([big])[0]
        if k == 2:
            z[ni + tamper] = T.ints_to_limbs([(T.limbs_to_ints(z[ni + tamper])[0] + 1) % P])[0]
        (az, bz, cz), want = oracle_h(oracle, mats, ni, z)
        got = from_montgomery(h[k].cpu().numpy().view(np.uint64))
        assert got == T.limbs_to_ints(want), "signature %d" % k
        # ... and the quotient entry point: hi of a(X) b(X), whatever the witness (for k = 0 that is `want` again)
        want_hi = oracle.qap_product_high_half(az, bz, ni, z)
        assert from_montgomery(hq[k].cpu().numpy().view(np.uint64)) == T.limbs_to_ints(want_hi), "quotient, signature %d" % k
        if k == 0:
            assert got[-1] == 0                                          # deg h <= n - 2 for a satisfied system


def test_witness_map_full_batch_identity_and_chunking(engine, oracle, tmp_path):
    """Falcon-1024, 40 signatures through a 16-signature workspace (chunks of 16, 16, 8): every signature's h equals the h
    of a one-signature call, and a sampled one satisfies A(tau) B(tau) - C(tau) = h(tau) (tau^n - 1) -- evaluated in
    Python without any FFT from the oracle's A z, B z, C z."""
    import torch
    import falcon_r1cs_amd as frw
    from test_r1cs_export import export, read_r1cs
    dev = torch.device("cuda:0")
    logn, batch = 10, 40
    sig, pk, hm = frw.synth_triples(logn, batch, seed=99)
    L = frw.layout(logn)
    d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
    wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
    inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
    st = torch.empty(batch, dtype=torch.int32, device=dev)
    s0 = torch.cuda.current_stream().cuda_stream
    engine.witness_ntt_verify_dev(logn, batch, d[0], d[1], d[2], wit, inst, st, 1, s0)
    h_r1cs = engine.r1cs_load(0, logn)
    try:
        q = engine.qap_info(h_r1cs)
        n, per = int(q.domain_size), int(q.workspace_bytes_per_signature)
        ws = torch.empty(16 * per, dtype=torch.uint8, device=dev)
        h = torch.empty((batch, n, 4), dtype=torch.int64, device=dev)
        bad = torch.empty(batch, dtype=torch.int32, device=dev)
        engine.qap_witness_map_dev(h_r1cs, batch, wit, inst, h, ws, 16 * per, bad, s0)
        torch.cuda.synchronize()
        assert int(bad.abs().sum()) == 0
        assert int(h[:, -1].abs().sum()) == 0
        h1 = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
        for k in (0, 15, 16, 33, 39):
            engine.qap_witness_map_dev(h_r1cs, 1, wit[k], inst[k], h1, ws, per, None, s0)
            torch.cuda.synchronize()
            assert torch.equal(h1[0], h[k]), k
    finally:
        engine.r1cs_free(h_r1cs)
    k = 33
    path = tmp_path / "c.r1cs"
    export(0, logn, path)
    ni, nw, nc, mats = read_r1cs(path)
    owit, oinst, _ = oracle.witness_ntt_verify(logn, sig[k:k + 1], pk[k:k + 1], hm[k:k + 1], 0)
    z = np.concatenate([oinst[0], owit[0]])
    az, bz, cz = (T.limbs_to_ints(oracle.qap_matvec(*m, z)) for m in mats)
    got = from_montgomery(h[k].cpu().numpy().view(np.uint64))
    lhs, rhs = qap.check_identity(az, bz, cz, ni, T.limbs_to_ints(z[:ni]), got, random.Random(7).randrange(P))
    assert lhs == rhs


@pytest.mark.parametrize("walk", ["flattened rows", "CSR walk"])
def test_witness_map_matches_committed_golden(engine, walk, monkeypatch):
    """tests/golden/qap.json: the digest of h (ark-ff's Montgomery bytes, as frw_qap_witness_map_dev writes them) for the
    committed witness fixtures, Falcon-512 and Falcon-1024 -- through the flattened short rows (r1cs_eval_flat_kernel) and
    through the CSR walk a circuit that does not fit them takes (r1cs_eval_kernel; FRW_R1CS_NO_FLAT at load time), incl. a
    tampered witness, which both must flag."""
    if walk == "CSR walk":
        monkeypatch.setenv("FRW_R1CS_NO_FLAT", "1")
    import hashlib
    import json
    import os
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    gold = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for fx in json.load(open(os.path.join(gold, "qap.json")))["cases"]:
        wfx = json.load(open(os.path.join(gold, fx["witness_fixture"])))
        logn = fx["logn"]
        L = frw.layout(logn)
        d = [torch.from_numpy(np.frombuffer(bytes.fromhex(wfx[k]), dtype=np.uint16).copy().view(np.int16)).to(dev)
             for k in ("sig", "pk", "hm")]
        wit = torch.empty((1, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((1, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.empty(1, dtype=torch.int32, device=dev)
        engine.witness_ntt_verify_dev(logn, 1, d[0], d[1], d[2], wit, inst, st, 1, 0)
        r = engine.r1cs_load(0, logn)
        try:
            q = engine.qap_info(r)
            assert int(q.log_domain_size) == fx["log_domain_size"]
            per = int(q.workspace_bytes_per_signature)
            ws = torch.empty(per, dtype=torch.uint8, device=dev)
            h = torch.empty((1, int(q.domain_size), 4), dtype=torch.int64, device=dev)
            bad = torch.empty(1, dtype=torch.int32, device=dev)
            engine.qap_witness_map_dev(r, 1, wit, inst, h, ws, per, bad, 0)
            torch.cuda.synchronize()
            assert bad.tolist() == [0]
            assert hashlib.sha256(h[0].cpu().numpy().tobytes()).hexdigest() == fx["h_sha256"]["montgomery"]
            # one boolean of a range proof flipped: rows of the flattened / CSR evaluation no longer hold, and the count says so
            wbad = wit.clone()
            wbad[0, 2 * L.n + 5, 0] ^= 1
            h2 = torch.empty_like(h)
            engine.qap_witness_map_dev(r, 1, wbad, inst, h2, ws, per, bad, 0)
            torch.cuda.synchronize()
            assert int(bad[0]) > 0
        finally:
            engine.r1cs_free(r)


def test_host_entry_point_equals_device_entry_point(engine):
    """frw_qap_witness_map (host buffers, what a holder of arkworks' Vec<Fr>s calls) == frw_qap_witness_map_dev, for a
    batch that spans two internal chunks (70 = 64 + 6 Falcon-512 signatures)."""
    import torch
    import falcon_r1cs_amd as frw
    logn, batch = 9, 70
    sig, pk, hm = frw.synth_triples(logn, batch, seed=31337)
    wit, inst, st = engine.witness_ntt_verify(logn, sig, pk, hm, frw.ENC_MONTGOMERY, strict=True)
    r = engine.r1cs_load(0, logn)
    try:
        h, bad = engine.qap_witness_map(r, wit, inst)
        assert not bad.any() and not h[:, -1].any()
        dev = torch.device("cuda:0")
        q = engine.qap_info(r)
        per = int(q.workspace_bytes_per_signature)
        dw = torch.from_numpy(wit.view(np.int64)).to(dev)
        di = torch.from_numpy(inst.view(np.int64)).to(dev)
        dh = torch.empty((batch, int(q.domain_size), 4), dtype=torch.int64, device=dev)
        ws = torch.empty(8 * per, dtype=torch.uint8, device=dev)
        engine.qap_witness_map_dev(r, batch, dw, di, dh, ws, 8 * per, None, 0)
        torch.cuda.synchronize()
        assert np.array_equal(dh.cpu().numpy().view(np.uint64), h)
    finally:
        engine.r1cs_free(r)


def test_argument_checks(engine):
    """FRW_E_INVALID_ARG for a workspace smaller than one signature's and for null buffers; an empty batch is a no-op."""
    import ctypes as C
    import torch
    import falcon_r1cs_amd as frw
    lib = frw.load_library()
    r = engine.r1cs_load(0, 9)
    try:
        q = engine.qap_info(r)
        per = int(q.workspace_bytes_per_signature)
        dev = torch.device("cuda:0")
        buf = torch.zeros(1024, dtype=torch.int64, device=dev)
        P = C.c_void_p(buf.data_ptr())
        for fn in (lib.frw_qap_witness_map_dev, lib.frw_qap_quotient_dev):
            assert fn(r, 0, None, None, None, None, None, 0, None) == 0                     # empty batch
            assert fn(r, 1, P, P, P, None, P, per - 1, None) == -1                         # workspace too small
            assert fn(r, 1, None, P, P, None, P, per, None) == -1                          # null witness
            assert fn(None, 1, P, P, P, None, P, per, None) == -1                          # no matrices
        assert lib.frw_qap_witness_map(r, 0, None, None, None, None) == 0
        assert lib.frw_qap_witness_map(r, 1, None, None, None, None) == -1
        assert lib.frw_qap_info(r, None) == -1
    finally:
        engine.r1cs_free(r)
