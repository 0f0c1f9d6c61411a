"""A whole Groth16 proof on the device (frw_groth16_prove_dev) for Falcon-512 and Falcon-1024 signatures, against the prover restated in the
exponent (oracle/bls12_381.py: ark-groth16 0.3.0 generator.rs / prover.rs / verifier.rs with the toxic waste known).

The proving key is made from known toxic waste: every query element is (a known Fr value) x generator, built with the engine's
fixed-base routines (checked against the oracle in tests/test_gpu_msm.py).  Expected proof = (a G1, b G2, c G1) with (a, b, c)
the discrete logarithms prove_exponents() gives for the same witness, h, r and s -- A and C by ONE scalar multiplication of the
C oracle each, B by Python integers over Fq2 -- and those exponents satisfy the verification equation
a b = alpha beta + (sum x_i gamma_abc_i) gamma + c delta, i.e. the proof the GPU wrote is one verify_proof accepts."""
import random

import numpy as np
import pytest

import frw_testlib as T
from oracle import bls12_381 as E
from oracle import qap

pytestmark = pytest.mark.gpu
FR_R = (1 << 256) % E.R


def _rows(ptr, col, val):
    """CSR with canonical limbs -> the oracle's list of rows [(coeff, column)]."""
    vals = T.limbs_to_ints(val)
    ptr = [int(x) for x in ptr]
    col = col.tolist()
    return [[(vals[k], col[k]) for k in range(ptr[r], ptr[r + 1])] for r in range(len(ptr) - 1)]


@pytest.mark.parametrize("logn", [9, 10])
def test_proofs_equal_the_prover_restated_in_the_exponent(engine, oracle, tmp_path, logn):
    import torch
    import falcon_r1cs_amd as frw
    from test_r1cs_export import export, read_r1cs
    dev = torch.device("cuda:0")
    batch = 3
    L = frw.layout(logn)
    export(0, logn, tmp_path / "c.r1cs")
    ni, nw, nc, mats = read_r1cs(tmp_path / "c.r1cs")
    assert (ni, nw, nc) == (L.num_instance, L.num_witness, L.num_constraints)
    d = qap.Domain(nc + ni)
    n = d.size
    rng = random.Random(16)
    toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
    pk = E.setup_exponents(tuple(_rows(*m) for m in mats), ni, nw, d, toxic)
    lim = T.ints_to_limbs
    g1 = lambda ks: engine.g1_fixed_base(lim(ks))
    g2 = lambda ks: engine.g2_fixed_base(lim(ks))
    a_query, b_g1_query, b_g2_query = g1(pk["u"]), g1(pk["v"]), g2(pk["v"])
    h_query, l_query = g1(pk["h"]), g1(pk["l"])
    fixed = g1([toxic["alpha"], toxic["beta"], toxic["delta"]])
    fixed2 = g2([toxic["beta"], toxic["delta"]])
    # spot checks of the key against the oracle (the whole arrays are products of routines tested elsewhere)
    for arr, ks in ((a_query, pk["u"]), (l_query, pk["l"]), (h_query, pk["h"])):
        for i in (0, 1, len(ks) // 2, len(ks) - 1):
            assert arr[i].tolist() == oracle.g1_scalar_mul(oracle.g1_generator(), ks[i]).tolist()
    assert b_g2_query[ni + 5].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, pk["v"][ni + 5]))
    handle = engine.groth16_pk_load(ni, nw, n, fixed[0], fixed[1], fixed[2], fixed2[0], fixed2[1], a_query, b_g1_query, b_g2_query,
                                    h_query, l_query)
    r1cs = engine.r1cs_load(0, logn)
    try:
        sig, pk_, hm = frw.synth_triples(logn, batch, seed=2718)
        dd = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk_, hm)]
        wit = torch.empty((batch, nw, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((batch, ni, 4), dtype=torch.int64, device=dev)
        st = torch.empty(batch, dtype=torch.int32, device=dev)
        s0 = torch.cuda.current_stream().cuda_stream
        engine.witness_ntt_verify_dev(logn, batch, dd[0], dd[1], dd[2], wit, inst, st, 1, s0)
        rs = [[rng.randrange(E.R), rng.randrange(E.R)] for _ in range(batch)]
        lam = E.Z_BLS ** 2 - 1                                      # the endomorphism's eigenvalue: the scalars are split by it
        rs[0] = [E.R - 1, lam]                                      # k1 at its maximum (r - 1 = lambda (lambda + 1)); k0 = 0, k1 = 1
        rs[1] = [0, 0]                                              # create_proof_no_zk
        rs[2] = [E.R + 12345, (1 << 256) - 1]                       # out of contract (>= the group order): taken mod it
        ws_bytes = engine.groth16_workspace_bytes(handle, r1cs, 2)  # chunks of 2 + 1
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        proofs = torch.full((batch, 48), -1, dtype=torch.int64, device=dev)
        bad = torch.full((batch,), -1, dtype=torch.int32, device=dev)
        engine.groth16_prove_dev(handle, r1cs, batch, wit, inst, np.array([lim(x) for x in rs]), proofs, ws, ws_bytes, bad, s0)
        torch.cuda.synchronize()
        assert not bad.any()
        got = proofs.cpu().numpy().view(np.uint64)
        # a witness that violates the system is flagged (its "proof" is still three points, worth nothing)
        wbad = wit.clone()
        wbad[0, L.n + 3, 0] += 1
        engine.groth16_prove_dev(handle, r1cs, 1, wbad, inst, np.array([lim(rs[0])]), torch.empty((1, 48), dtype=torch.int64, device=dev),
                                 ws, ws_bytes, bad, s0)
        torch.cuda.synchronize()
        assert int(bad[0]) > 0
        # the same h, independently: the witness map on its own
        q = engine.qap_info(r1cs)
        qws = torch.empty(batch * int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
        h = torch.empty((batch, n, 4), dtype=torch.int64, device=dev)
        engine.qap_witness_map_dev(r1cs, batch, wit, inst, h, qws, qws.numel(), None, s0)
        torch.cuda.synchronize()
        r_inv = pow(FR_R, -1, E.R)
        gen = oracle.g1_generator()
        for i in range(batch):
            z = [v * r_inv % E.R for v in T.limbs_to_ints(inst[i].cpu().numpy().view(np.uint64))] + \
                [v * r_inv % E.R for v in T.limbs_to_ints(wit[i].cpu().numpy().view(np.uint64))]
            h_int = [v * r_inv % E.R for v in T.limbs_to_ints(h[i].cpu().numpy().view(np.uint64))]
            a, b, c, _ = E.prove_exponents(pk, z, h_int, rs[i][0] % E.R, rs[i][1] % E.R)
            assert E.verify_exponents(pk, z[1:ni], (a, b, c)), "the restated prover's own proof does not verify"
            assert got[i, :12].tolist() == oracle.g1_scalar_mul(gen, a).tolist(), "A of signature %d" % i
            assert got[i, 12:36].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, b)), "B of signature %d" % i
            assert got[i, 36:].tolist() == oracle.g1_scalar_mul(gen, c).tolist(), "C of signature %d" % i
    finally:
        engine.r1cs_free(r1cs)
        engine.groth16_pk_free(handle)


def test_setup_on_the_product_side_makes_the_same_key(engine, oracle, tmp_path):
    """frw_groth16_setup (generator.rs generate_parameters with the toxic waste given: QAP at t on the host, queries as
    fixed-base multiples on the device) against oracle/bls12_381.py::setup_exponents for Falcon-512: the verifying key's
    elements are the generator multiples the oracle's exponents give, and proofs made with the returned proving key equal the
    prover restated in the exponent -- which could not be if any query element differed."""
    import torch
    import falcon_r1cs_amd as frw
    from test_r1cs_export import export, read_r1cs
    dev = torch.device("cuda:0")
    logn, batch = 9, 2
    L = frw.layout(logn)
    export(0, logn, tmp_path / "c.r1cs")
    ni, nw, nc, mats = read_r1cs(tmp_path / "c.r1cs")
    d = qap.Domain(nc + ni)
    n = d.size
    rng = random.Random(61)
    toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
    pk = E.setup_exponents(tuple(_rows(*m) for m in mats), ni, nw, d, toxic)
    handle, vk = engine.groth16_setup(0, logn, toxic["alpha"], toxic["beta"], toxic["gamma"], toxic["delta"], toxic["t"])
    gen = oracle.g1_generator()
    assert vk["alpha_g1"].tolist() == oracle.g1_scalar_mul(gen, toxic["alpha"]).tolist()
    for name in ("beta", "gamma", "delta"):
        assert vk[name + "_g2"].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, toxic[name])), name
    assert vk["gamma_abc_g1"].shape == (ni, 12)
    for i in (0, 1, ni // 2, ni - 1):
        assert vk["gamma_abc_g1"][i].tolist() == oracle.g1_scalar_mul(gen, pk["gamma_abc"][i]).tolist(), i
    r1cs = engine.r1cs_load(0, logn)
    try:
        sig, pk_, hm = frw.synth_triples(logn, batch, seed=1618)
        dd = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk_, hm)]
        wit = torch.empty((batch, nw, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((batch, ni, 4), dtype=torch.int64, device=dev)
        st = torch.empty(batch, dtype=torch.int32, device=dev)
        s0 = torch.cuda.current_stream().cuda_stream
        engine.witness_ntt_verify_dev(logn, batch, dd[0], dd[1], dd[2], wit, inst, st, 1, s0)
        rs = [[rng.randrange(E.R), rng.randrange(E.R)] for _ in range(batch)]
        rs[1] = [E.Z_BLS ** 2 - 2, 1]                               # lambda - 1: all of it in the low half; and one
        ws_bytes = engine.groth16_workspace_bytes(handle, r1cs, batch)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        proofs = torch.empty((batch, 48), dtype=torch.int64, device=dev)
        engine.groth16_prove_dev(handle, r1cs, batch, wit, inst, np.array([T.ints_to_limbs(x) for x in rs]), proofs, ws, ws_bytes, None, s0)
        q = engine.qap_info(r1cs)
        qws = torch.empty(batch * int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
        h = torch.empty((batch, n, 4), dtype=torch.int64, device=dev)
        engine.qap_witness_map_dev(r1cs, batch, wit, inst, h, qws, qws.numel(), None, s0)
        torch.cuda.synchronize()
        got = proofs.cpu().numpy().view(np.uint64)
        r_inv = pow(FR_R, -1, E.R)
        for i in range(batch):
            z = [v * r_inv % E.R for v in T.limbs_to_ints(inst[i].cpu().numpy().view(np.uint64))] + \
                [v * r_inv % E.R for v in T.limbs_to_ints(wit[i].cpu().numpy().view(np.uint64))]
            h_int = [v * r_inv % E.R for v in T.limbs_to_ints(h[i].cpu().numpy().view(np.uint64))]
            a, b, c, _ = E.prove_exponents(pk, z, h_int, rs[i][0], rs[i][1])
            assert E.verify_exponents(pk, z[1:ni], (a, b, c))
            assert got[i, :12].tolist() == oracle.g1_scalar_mul(gen, a).tolist()
            assert got[i, 12:36].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, b))
            assert got[i, 36:].tolist() == oracle.g1_scalar_mul(gen, c).tolist()
        # ... and the flow of examples/pok_sig.rs:30-47 end to end, with nothing known in the exponent: the key from
        # frw_groth16_setup, the proof from frw_groth16_prove_dev, and ark-groth16's verify_proof restated with a real pairing
        # (oracle/bls12_381.py): e(A, B) = e(alpha, beta) e(sum x_i gamma_abc_i, gamma) e(C, delta)
        vk_pts = {"alpha_g1": E.from_limbs(vk["alpha_g1"]), "beta_g2": E.g2_from_limbs(vk["beta_g2"]),
                  "gamma_g2": E.g2_from_limbs(vk["gamma_g2"]), "delta_g2": E.g2_from_limbs(vk["delta_g2"]),
                  "gamma_abc_g1": [E.from_limbs(row) for row in vk["gamma_abc_g1"]]}
        z0 = [v * r_inv % E.R for v in T.limbs_to_ints(inst[0].cpu().numpy().view(np.uint64))]
        proof0 = (E.from_limbs(got[0, :12]), E.g2_from_limbs(got[0, 12:36]), E.from_limbs(got[0, 36:]))
        assert all(E.on_curve(p_) for p_ in (proof0[0], proof0[2])) and E.g2_on_curve(proof0[1])
        assert E.verify_proof(vk_pts, z0[1:], proof0), "verify_proof rejects the device's proof"
        wrong = list(z0[1:])
        wrong[7] = (wrong[7] + 1) % E.R
        assert not E.verify_proof(vk_pts, wrong, proof0)
        # the product's own verifier (frw_groth16_verify: host pairing, frw_pairing.h) says the same about every proof of the
        # batch, taking the instance buffer and the proofs exactly as the device wrote them
        ver = frw.Groth16Verifier(vk)
        inst_h = inst.cpu().numpy().view(np.uint64)
        assert ver.verify(inst_h, got).tolist() == [1] * batch
        tampered = inst_h.copy()
        tampered[1, 8, 0] ^= np.uint64(2)
        swapped = got[::-1].copy()                                  # proof i with statement 1 - i
        assert ver.verify(tampered, got).tolist() == [1, 0]
        assert ver.verify(inst_h, swapped).tolist() == [0, 0]
        ver.close()
        with pytest.raises(frw.FrwError):                          # t inside the domain: zt = 0
            engine.groth16_setup(0, logn, 3, 5, 7, 11, d.group_gen)
    finally:
        engine.r1cs_free(r1cs)
        engine.groth16_pk_free(handle)


@pytest.mark.parametrize("circuit,logn", [(1, 9), (0, 10)])
def test_setup_prove_verify_with_nothing_known_in_the_exponent(engine, circuit, logn):
    """examples/pok_sig.rs:30-47 on the product alone, for the circuits the other tests leave out (the dual-NTT circuit;
    Falcon-1024 with a key from frw_groth16_setup): circuit_specific_setup -> frw_groth16_setup, create_random_proof ->
    frw_groth16_prove_dev, Groth16::verify -> frw_groth16_verify.  Every proof must verify for its own statement, none for its
    neighbour's, and a proof made from a witness that violates the system (one value flipped after the witness kernel ran) must
    be reported (num_unsatisfied) and must not verify."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    batch = 3
    L = frw.layout_dual(logn) if circuit else frw.layout(logn)
    rng = random.Random(700 + logn + circuit)
    key, vk = engine.groth16_setup(circuit, logn, *(rng.randrange(2, E.R) for _ in range(5)))
    r1cs = engine.r1cs_load(circuit, logn)
    try:
        sig, pk_, hm = frw.synth_triples(logn, batch, seed=2718 + circuit)
        dd = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk_, hm)]
        wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.empty(batch, dtype=torch.int32, device=dev)
        s0 = torch.cuda.current_stream().cuda_stream
        if circuit:
            engine.witness_dual_ntt_verify_dev(logn, batch, dd[0], dd[1], dd[2], wit, inst, st, 1, s0)
        else:
            engine.witness_ntt_verify_dev(logn, batch, dd[0], dd[1], dd[2], wit, inst, st, 1, s0)
        torch.cuda.synchronize()
        assert int(st.abs().sum()) == 0
        # signature 2's witness stops satisfying the system: a bit of its first range proof flipped (Montgomery 0 <-> 1)
        w2 = wit[2, L.seg_off[2]].clone()
        one = torch.from_numpy(np.frombuffer(((1 << 256) % E.R).to_bytes(32, "little"), dtype=np.int64).copy()).to(dev)
        wit[2, L.seg_off[2]] = torch.where(w2.abs().sum() == 0, one, torch.zeros_like(one))
        rs = np.array([T.ints_to_limbs([rng.randrange(E.R), rng.randrange(E.R)]) for _ in range(batch)])
        ws_bytes = engine.groth16_workspace_bytes(key, r1cs, batch)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        proofs = torch.empty((batch, 48), dtype=torch.int64, device=dev)
        bad = torch.zeros(batch, dtype=torch.int32, device=dev)
        engine.groth16_prove_dev(key, r1cs, batch, wit, inst, rs, proofs, ws, ws_bytes, bad, s0)
        torch.cuda.synchronize()
        assert bad[:2].tolist() == [0, 0] and int(bad[2]) > 0
        ver = frw.Groth16Verifier(vk)
        inst_h, proofs_h = inst.cpu().numpy().view(np.uint64), proofs.cpu().numpy().view(np.uint64)
        assert ver.verify(inst_h, proofs_h).tolist() == [1, 1, 0]
        assert ver.verify(inst_h[[1, 0, 2]], proofs_h).tolist() == [0, 0, 0]
        ver.close()
        # a workspace for two signatures in flight: the batch runs as 2 + 1 (the key's side streams and events used twice in one
        # call, the workspace reused) and must give the same bytes
        ws2_bytes = engine.groth16_workspace_bytes(key, r1cs, 2)
        ws2 = torch.empty(ws2_bytes, dtype=torch.uint8, device=dev)
        proofs2 = torch.zeros((batch, 48), dtype=torch.int64, device=dev)
        bad2 = torch.zeros(batch, dtype=torch.int32, device=dev)
        engine.groth16_prove_dev(key, r1cs, batch, wit, inst, rs, proofs2, ws2, ws2_bytes, bad2, s0)
        torch.cuda.synchronize()
        assert torch.equal(proofs2, proofs) and torch.equal(bad2, bad)
    finally:
        engine.r1cs_free(r1cs)
        engine.groth16_pk_free(key)


@pytest.mark.gpu
def test_blinding_factors_in_device_memory_and_a_captured_call(engine):
    """frw_groth16_prove_rs_dev: the blinding factors in device memory, split by the endomorphism on the device -- the same proofs,
    byte for byte, as frw_groth16_prove_dev with the same factors in host memory (incl. factors beyond the group order and the
    extremes of the split); and the call captured into a HIP graph (torch.cuda.CUDAGraph on a stream of its own: the key's streams
    join the capture through the call's events) replays to the same proofs, and to the right ones after the factors in d_rs change."""
    import torch
    import falcon_r1cs_amd as frw
    dev = torch.device("cuda:0")
    logn, batch = 9, 3
    L = frw.layout(logn)
    rng = random.Random(811)
    key, vk = engine.groth16_setup(0, logn, *(rng.randrange(2, E.R) for _ in range(5)))
    r1cs = engine.r1cs_load(0, logn)
    try:
        sig, pk_, hm = frw.synth_triples(logn, batch, seed=99)
        dd = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk_, hm)]
        wit = torch.empty((batch, L.num_witness, 4), dtype=torch.int64, device=dev)
        inst = torch.empty((batch, L.num_instance, 4), dtype=torch.int64, device=dev)
        st = torch.empty(batch, dtype=torch.int32, device=dev)
        engine.witness_ntt_verify_dev(logn, batch, dd[0], dd[1], dd[2], wit, inst, st, 1, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        lam = E.Z_BLS ** 2 - 1
        lim = lambda pairs: np.array([T.ints_to_limbs([a % (1 << 256), b % (1 << 256)]) for a, b in pairs])
        sets = [[(E.R - 1, lam), (E.R + 12345, (1 << 256) - 1), (rng.randrange(E.R), 0)],
                [(rng.randrange(E.R), rng.randrange(E.R)) for _ in range(batch)]]
        ws_bytes = engine.groth16_workspace_bytes(key, r1cs, batch)
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        want = []
        for pairs in sets:
            p = torch.empty((batch, 48), dtype=torch.int64, device=dev)
            engine.groth16_prove_dev(key, r1cs, batch, wit, inst, lim(pairs), p, ws, ws_bytes, None, torch.cuda.current_stream().cuda_stream)
            torch.cuda.synchronize()
            want.append(p)
        ver = frw.Groth16Verifier(vk)
        assert ver.verify(inst.cpu().numpy().view(np.uint64), want[1].cpu().numpy().view(np.uint64)).tolist() == [1] * batch
        ver.close()
        d_rs = torch.from_numpy(lim(sets[0]).view(np.int64)).to(dev)
        got = torch.zeros((batch, 48), dtype=torch.int64, device=dev)
        engine.groth16_prove_rs_dev(key, r1cs, batch, wit, inst, d_rs, got, ws, ws_bytes, None, torch.cuda.current_stream().cuda_stream)
        torch.cuda.synchronize()
        assert torch.equal(got, want[0])
        # captured once, replayed with either set of factors
        side = torch.cuda.Stream()
        graph = torch.cuda.CUDAGraph()
        got.zero_()
        with torch.cuda.stream(side):
            engine.groth16_prove_rs_dev(key, r1cs, batch, wit, inst, d_rs, got, ws, ws_bytes, None, side.cuda_stream)      # warm the capture stream
            torch.cuda.synchronize()
            with torch.cuda.graph(graph, stream=side, capture_error_mode="thread_local"):
                engine.groth16_prove_rs_dev(key, r1cs, batch, wit, inst, d_rs, got, ws, ws_bytes, None, torch.cuda.current_stream().cuda_stream)
        for k in (0, 1, 0):
            d_rs.copy_(torch.from_numpy(lim(sets[k]).view(np.int64)).to(dev))
            got.zero_()
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            assert torch.equal(got, want[k]), "replay %d" % k
    finally:
        engine.r1cs_free(r1cs)
        engine.groth16_pk_free(key)
