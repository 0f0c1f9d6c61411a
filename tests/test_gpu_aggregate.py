"""ONE Groth16 proof for an aggregate statement (BASELINE configs[4]: "aggregate proof, mixed Falcon-512/1024").

The statement: FalconNTTVerificationCircuit::generate_constraints (falcon_ntt.rs:26-123) once per (pk, msg, sig) on one constraint
system; the proof: the flow of examples/pok_sig.rs:30-47 on that system.  Checked here, on the device, against
  (1) the oracle's own aggregate system (oracle/ark_sim.py runs the restated gadgets statement after statement: tests/
      test_oracle_aggregate.py): assignment bytes, A z / B z / C z, h, and (A, B, C) against the prover restated in the exponent
      with the oracle's matrices -- for a mixed two-statement aggregate;
  (2) for larger aggregates (whose matrices Python cannot hold as rows) the same quantities from the statements' own products
      laid end to end -- which (1) and the CPU test pin as the right system: h against the oracle's witness map (oracle/
      qap_oracle.c) and the FFT-free identity, (A, B, C) against oracle/bls12_381.py::prove_exponents_from_products, and the
      product's and the oracle's pairing verifiers on the result."""
import os
import random

import numpy as np
import pytest

import frw_testlib as T
from oracle import bls12_381 as E
from oracle import falcon_gadgets as G
from oracle import qap

pytestmark = pytest.mark.gpu
P = qap.P
R_MONT = qap.R_MONT
R_INV = pow(R_MONT, -1, P)


def from_montgomery(limbs):
    return [v * R_INV % P for v in T.limbs_to_ints(limbs)]


class Aggregate:
    """The statements of one aggregate on the device: per-parameter-set batches from the witness kernel, the aggregate handle,
    and the aggregate's own assignment vectors."""

    def __init__(self, engine, logns, seed):
        import torch
        import falcon_r1cs_amd as frw
        self.engine, self.logns = engine, list(logns)
        dev = self.dev = torch.device("cuda:0")
        self.s0 = torch.cuda.current_stream().cuda_stream
        self.triples, self.batches = {}, {}
        for g in (9, 10):
            cnt = self.logns.count(g)
            if not cnt:
                continue
            L = frw.layout(g)
            sig, pk, hm = frw.synth_triples(g, cnt, seed=seed + g)
            self.triples[g] = (sig, pk, hm)
            d = [torch.from_numpy(a.view(np.int16)).to(dev) for a in (sig, pk, hm)]
            wit = torch.empty((cnt, L.num_witness, 4), dtype=torch.int64, device=dev)
            inst = torch.empty((cnt, L.num_instance, 4), dtype=torch.int64, device=dev)
            st = torch.empty(cnt, dtype=torch.int32, device=dev)
            engine.witness_ntt_verify_dev(g, cnt, d[0], d[1], d[2], wit, inst, st, 1, self.s0)
            torch.cuda.synchronize()
            assert not st.any()
            self.batches[g] = (wit, inst)
        self.handle = engine.r1cs_load_aggregate(self.logns)
        self.info = engine.r1cs_info(self.handle)
        self.ni, self.nw, self.nc = int(self.info.num_instance), int(self.info.num_witness), int(self.info.num_constraints)
        self.wit = torch.full((1, self.nw, 4), -1, dtype=torch.int64, device=dev)
        self.inst = torch.full((1, self.ni, 4), -1, dtype=torch.int64, device=dev)
        b9, b10 = self.batches.get(9, (None, None)), self.batches.get(10, (None, None))
        engine.aggregate_assign_dev(self.handle, b9[0], b9[1], b10[0], b10[1], self.wit, self.inst, self.s0)
        torch.cuda.synchronize()

    def statements(self):
        """(sig, pk, hm, logn) of every statement in order: statement i takes the next unused triple of its parameter set."""
        used, out = {9: 0, 10: 0}, []
        for g in self.logns:
            sig, pk, hm = self.triples[g]
            k = used[g]
            used[g] += 1
            out.append((sig[k], pk[k], hm[k], g))
        return out

    def close(self):
        self.engine.r1cs_free(self.handle)


def statement_products(oracle, tmp_path, agg):
    """A z, B z, C z of the whole statement from the statements' own products (the C oracle on the matrices the product exports
    per parameter set -- tests/test_r1cs_export.py holds those to the oracle's inlining), the full assignment, and the matrix
    entries in instance columns, all in the aggregate's numbering.  Canonical integers / limbs."""
    from test_r1cs_export import export, read_r1cs
    mats = {}
    for g in set(agg.logns):
        export(0, g, tmp_path / ("c%d.r1cs" % g))
        mats[g] = read_r1cs(tmp_path / ("c%d.r1cs" % g))
    abc = [[], [], []]
    inst_parts, wit_parts, terms = [np.array(T.ints_to_limbs([1]))], [], []
    rows = pub = 0
    for sig, pk, hm, g in agg.statements():
        ni, nw, nc, m = mats[g]
        owit, oinst, ost = oracle.witness_ntt_verify(g, sig, pk, hm, 0)
        assert ost[0] == 0
        z = np.concatenate([oinst[0], owit[0]])
        for k in range(3):
            abc[k].append(oracle.qap_matvec(*m[k], z))
            ptr, col, val = m[k]
            sel = np.nonzero(col < ni)[0]
            if len(sel):
                row_of = np.searchsorted(ptr.astype(np.int64), sel, side="right") - 1
                coeffs = T.limbs_to_ints(val[sel])
                for r_, c_, v_ in zip(row_of.tolist(), col[sel].tolist(), coeffs):
                    terms.append((k, rows + r_, 0 if c_ == 0 else pub + c_, v_))
        inst_parts.append(oinst[0][1:])
        wit_parts.append(owit[0])
        rows += nc
        pub += ni - 1
    az, bz, cz = (np.concatenate(x) for x in abc)
    return az, bz, cz, np.concatenate(inst_parts), np.concatenate(wit_parts), terms


def test_two_mixed_statements_against_the_oracles_own_aggregate_system(engine, oracle):
    import torch
    import falcon_r1cs_amd as frw
    agg = Aggregate(engine, (9, 10), seed=505)
    try:
        dev, s0 = agg.dev, agg.s0
        L9, L10 = frw.layout(9), frw.layout(10)
        assert (agg.ni, agg.nw, agg.nc) == (1 + 2 * 512 + 2 * 1024, L9.num_witness + L10.num_witness, L9.num_constraints + L10.num_constraints)
        assert int(agg.info.num_statements) == 2 and int(agg.info.log_domain_size) == 18 and agg.info.witness_map_on_device
        # (1) the oracle's aggregate system and assignment
        cs = G.run_reference_flow_aggregate([(s.tolist(), p.tolist(), h.tolist(), g) for s, p, h, g in agg.statements()], strict=True)
        assert (cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints()) == (agg.ni, agg.nw, agg.nc)
        assert agg.inst.cpu().numpy().tobytes() == G.encode_elements(cs.instance_assignment, True), "aggregate instance_assignment"
        assert agg.wit.cpu().numpy().tobytes() == G.encode_elements(cs.witness_assignment, True), "aggregate witness_assignment"
        z = cs.instance_assignment + cs.witness_assignment
        mats = tuple([[(v, c) for c, v in row] for row in m] for m in cs.to_matrices())
        az, bz, cz = qap.matvec(mats, z)
        # (2) the three products and the satisfaction count, with and without storing them
        abc = torch.full((1, 3, agg.nc, 4), -1, dtype=torch.int64, device=dev)
        bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
        engine.r1cs_eval_dev(agg.handle, 1, agg.wit, agg.inst, bad, abc, s0)
        bad2 = torch.full((1,), -1, dtype=torch.int32, device=dev)
        engine.r1cs_check_dev(agg.handle, 1, agg.wit, agg.inst, bad2, s0)
        torch.cuda.synchronize()
        assert bad.tolist() == [0] and bad2.tolist() == [0]
        got = abc.cpu().numpy().view(np.uint64)[0]
        for k, want in enumerate((az, bz, cz)):
            assert from_montgomery(got[k]) == want, "product %d of the aggregate" % k
        # (3) h = the witness map of the aggregate's 2^18 domain
        q = engine.qap_info(agg.handle)
        n = int(q.domain_size)
        assert n == 1 << 18
        ws = torch.empty(int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
        h = torch.full((1, n, 4), -1, dtype=torch.int64, device=dev)
        engine.qap_witness_map_dev(agg.handle, 1, agg.wit, agg.inst, h, ws, ws.numel(), bad, s0)
        torch.cuda.synchronize()
        h_int = from_montgomery(h[0].cpu().numpy().view(np.uint64))
        lim = T.ints_to_limbs
        want_h = T.limbs_to_ints(oracle.qap_witness_map(lim(az), lim(bz), lim(cz), agg.ni, lim(z)))
        assert h_int == want_h and h_int[-1] == 0
        # (4) one proof for the two statements, against the prover restated in the exponent on the ORACLE's matrices
        rng = random.Random(4)
        toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
        d = qap.Domain(agg.nc + agg.ni)
        pk_exp = E.setup_exponents(mats, agg.ni, agg.nw, d, toxic)
        key, vk = engine.groth16_setup_r1cs(agg.handle, toxic["alpha"], toxic["beta"], toxic["gamma"], toxic["delta"], toxic["t"])
        try:
            gen = oracle.g1_generator()
            for i in (0, 1, 1024, 1025, agg.ni - 1):
                assert vk["gamma_abc_g1"][i].tolist() == oracle.g1_scalar_mul(gen, pk_exp["gamma_abc"][i]).tolist(), i
            r, s = rng.randrange(E.R), rng.randrange(E.R)
            ws_bytes = engine.groth16_workspace_bytes(key, agg.handle, 1)
            pws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            proof = torch.full((1, 48), -1, dtype=torch.int64, device=dev)
            engine.groth16_prove_dev(key, agg.handle, 1, agg.wit, agg.inst, np.array([lim([r, s])]), proof, pws, ws_bytes, bad, s0)
            torch.cuda.synchronize()
            assert bad.tolist() == [0]
            got_p = proof.cpu().numpy().view(np.uint64)[0]
            a, b, c, _ = E.prove_exponents(pk_exp, z, h_int, r, s)
            assert E.verify_exponents(pk_exp, z[1:agg.ni], (a, b, c))
            assert got_p[:12].tolist() == oracle.g1_scalar_mul(gen, a).tolist(), "A"
            assert got_p[12:36].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, b)), "B"
            assert got_p[36:].tolist() == oracle.g1_scalar_mul(gen, c).tolist(), "C"
            ver = frw.Groth16Verifier(vk)
            inst_h = agg.inst.cpu().numpy().view(np.uint64)
            assert ver.verify(inst_h, got_p[None]).tolist() == [1]
            other = inst_h.copy()
            other[0, 1 + 1024 + 5, 0] ^= np.uint64(1)                  # a public input of the SECOND statement
            assert ver.verify(other, got_p[None]).tolist() == [0]
            ver.close()
            # a witness value of the second statement spoilt: the whole statement is flagged and its proof worthless
            wbad = agg.wit.clone()
            wbad[0, L9.num_witness + L10.n + 3, 0] += 1
            engine.groth16_prove_dev(key, agg.handle, 1, wbad, agg.inst, np.array([lim([r, s])]), proof, pws, ws_bytes, bad, s0)
            torch.cuda.synchronize()
            assert int(bad[0]) > 0
            ver = frw.Groth16Verifier(vk)
            assert ver.verify(inst_h, proof.cpu().numpy().view(np.uint64)).tolist() == [0]
            ver.close()
        finally:
            engine.groth16_pk_free(key)
    finally:
        agg.close()


@pytest.mark.parametrize("logns", [(10, 10), (9, 9, 9), (10,) * 8, (10,) * 32], ids=["2^19", "2^18-three-512", "2^21", "2^23"])
def test_witness_map_on_the_other_domains(engine, oracle, tmp_path, logns):
    """The transform schedules beyond the two per-signature domains (2^19 = 6 + 5 + 4 + 4 stages, 2^21 = 6 + 5 + 5 + 5, 2^23 = 6 + 6 + 6 + 5:
    thirty-two Falcon-1024 statements, 5.2 M constraints; 2^20 and 2^22 are the proof tests below): h of an aggregate against the oracle's witness map on the statements' products end to end, the
    six-transform quotient against the seven-transform map, and a spoilt witness through the exact (list-mode) route."""
    import torch
    agg = Aggregate(engine, logns, seed=77 + len(logns))
    try:
        dev, s0 = agg.dev, agg.s0
        q = engine.qap_info(agg.handle)
        n = int(q.domain_size)
        assert n >= agg.nc + agg.ni > n // 2
        ws = torch.empty(int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
        h = torch.full((1, n, 4), -1, dtype=torch.int64, device=dev)
        hq = torch.full((1, n, 4), -1, dtype=torch.int64, device=dev)
        bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
        engine.qap_witness_map_dev(agg.handle, 1, agg.wit, agg.inst, h, ws, ws.numel(), bad, s0)
        engine.qap_quotient_dev(agg.handle, 1, agg.wit, agg.inst, hq, ws, ws.numel(), None, s0)
        torch.cuda.synchronize()
        assert bad.tolist() == [0] and torch.equal(h, hq)
        az, bz, cz, z_inst, z_wit, _ = statement_products(oracle, tmp_path, agg)
        assert len(az) == agg.nc and len(z_inst) == agg.ni and len(z_wit) == agg.nw
        want = oracle.qap_witness_map(az, bz, cz, agg.ni, z_inst)
        got = h[0].cpu().numpy().view(np.uint64)
        want_m = T.ints_to_limbs([v * R_MONT % P for v in T.limbs_to_ints(want)])
        assert np.array_equal(got, want_m)
        assert not got[-1].any()
        # the last statement's witness spoilt: ark-groth16's seven-transform map runs for it (list mode) -- h is still its function
        # of (matrices, z), and the identity fails
        wbad = agg.wit.clone()
        last_w = agg.nw - 1000
        v = from_montgomery(wbad[0, last_w].cpu().numpy().view(np.uint64))[0]
        wbad[0, last_w] = torch.from_numpy(T.ints_to_limbs([(v + 1) * R_MONT % P])[0].view(np.int64)).to(dev)
        engine.qap_witness_map_dev(agg.handle, 1, wbad, agg.inst, h, ws, ws.numel(), bad, s0)
        torch.cuda.synchronize()
        assert int(bad[0]) > 0 and not torch.equal(h, hq)
    finally:
        agg.close()


def _digits8(v):
    """non-zero signed 8-bit digits of a scalar that is not one (frw_msm.hip scalar_digits8: a byte above 128 borrows from the next)"""
    cnt = carry = 0
    for j in range(32):
        d = ((v >> (8 * j)) & 0xff) + carry
        carry = 0
        if j < 31 and d > 128:
            d -= 256
            carry = 1
        cnt += d != 0
    return cnt


def _witness_side_counts(tmp_path, agg, z_ints):
    """What frw_diag_groth16_side_counts must say for the scalars z_ints = z ++ [1, r, s]: the rows in which b_g2_query holds a point are
    the variables with a non-zero column in B (the matrices the product exports per parameter set: tests/test_r1cs_export.py holds them to
    the oracle's) plus beta_2's and delta_2's; digits and ones over all rows and over those."""
    from test_r1cs_export import export, read_r1cs
    cols = {}
    for g in set(agg.logns):
        export(0, g, tmp_path / ("c%d.r1cs" % g))
        ni, nw, nc, m = read_r1cs(tmp_path / ("c%d.r1cs" % g))
        cols[g] = (ni, nw, np.unique(m[1][1]))
    live = np.zeros(agg.ni + agg.nw + 3, dtype=bool)
    pub, wit = 0, 0
    for g in agg.logns:
        ni, nw, c = cols[g]
        live[0] |= bool((c == 0).any())
        inst_c, wit_c = c[(c > 0) & (c < ni)], c[c >= ni]
        live[pub + inst_c] = True
        live[agg.ni + wit + (wit_c - ni)] = True
        pub += ni - 1
        wit += nw
    live[agg.ni + agg.nw] = live[agg.ni + agg.nw + 2] = True              # beta_2 (scalar 1), delta_2 (scalar s); row nv + 1 holds no point
    zs = np.array(z_ints, dtype=object)
    count = lambda sel: (sum(_digits8(int(v)) for v in zs[sel] if v not in (0, 1)), int(sum(1 for v in zs[sel] if v == 1)))
    everything = np.ones(len(zs), dtype=bool)
    return [int(live.sum())] + list(count(everything)) + list(count(live))


def _bare_and_sliced_keys_give_the_same_bytes(engine, agg, toxic, vk, r, s, want_proof, slices, tmp_path=None, z_ints=None):
    """The key of BARE handles (the points only, made on the device end to end: frw_groth16_setup_r1cs_opts) must give the proof of the
    window tables byte for byte -- and so must a key in slices: every rank's partial sums over its slices of the five queries, put
    together by frw_groth16_prove_combine_dev (here the ranks are handles on one card, one after the other)."""
    import torch
    import falcon_r1cs_amd as frw
    from falcon_r1cs_amd import engine as EN
    dev, s0 = agg.dev, agg.s0
    lim = T.ints_to_limbs
    tox = [toxic[k] for k in ("alpha", "beta", "gamma", "delta", "t")]
    rs = np.array([lim([r, s])])
    # window tables grown on the device from device-made rows: the host-made key's proof, byte for byte
    key, vk_dev = engine.groth16_setup_r1cs(agg.handle, *tox, mode=EN.KEY_TABLES)
    try:
        assert engine.groth16_pk_info(key).mode == EN.KEY_TABLES
        for k in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1"):
            assert np.array_equal(vk_dev[k], vk[k]), "the device-made verifying key differs from the host-made one: " + k
        ws_bytes = engine.groth16_workspace_bytes(key, agg.handle, 1)
        pws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        proof = torch.full((1, 48), -1, dtype=torch.int64, device=dev)
        bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
        engine.groth16_prove_dev(key, agg.handle, 1, agg.wit, agg.inst, rs, proof, pws, ws_bytes, bad, s0)
        torch.cuda.synchronize()
        assert bad.tolist() == [0]
        assert proof.cpu().numpy().view(np.uint64)[0].tolist() == want_proof.tolist(), "device-made window tables: another proof than the host-made key's"
        del pws
    finally:
        engine.groth16_pk_free(key)
    key, vk_bare = engine.groth16_setup_r1cs(agg.handle, *tox, mode=EN.KEY_BARE)
    try:
        info = engine.groth16_pk_info(key)
        assert (info.mode, info.rank, info.world) == (EN.KEY_BARE, 0, 1) and (info.z_lo, info.z_hi) == (0, agg.ni + agg.nw + 3)
        n = 1 << int(agg.info.log_domain_size)
        assert (info.h_lo, info.h_hi) == (0, n - 1) and info.key_bytes == 3 * (agg.ni + agg.nw + 3) * 112 + (agg.ni + agg.nw + 3) * 224 + (n - 1) * 112
        for k in ("alpha_g1", "beta_g2", "gamma_g2", "delta_g2", "gamma_abc_g1"):
            assert np.array_equal(vk_bare[k], vk[k]), "the device-made verifying key differs from the host-made one: " + k
        ws_bytes = engine.groth16_workspace_bytes(key, agg.handle, 1)
        pws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
        proof = torch.full((1, 48), -1, dtype=torch.int64, device=dev)
        bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
        engine.groth16_prove_dev(key, agg.handle, 1, agg.wit, agg.inst, rs, proof, pws, ws_bytes, bad, s0)
        torch.cuda.synchronize()
        assert bad.tolist() == [0]
        assert proof.cpu().numpy().view(np.uint64)[0].tolist() == want_proof.tolist(), "bare key: another proof than the window tables'"
        if z_ints is not None:
            # what the sums add up: the sums over b_g1_query / b_g2_query run over the rows that hold a point -- the variables of B's columns
            R = 1 << 256
            zext = z_ints + [1, r, s]
            d_z = torch.from_numpy(np.array(lim([v * R % E.R for v in zext])).view(np.int64)).to(dev)
            got = engine.diag_groth16_side_counts(key, d_z, pws, ws_bytes, s0)
            assert got == _witness_side_counts(tmp_path, agg, zext), got
            assert got[0] < 0.7 * len(zext) and got[3] + got[4] < 0.5 * (got[1] + got[2])     # (why the index exists)
        # a whole key through the partial / combine pair
        part = torch.full((1, EN.GROTH16_PARTIAL_WORDS), -1, dtype=torch.int64, device=dev)
        cws = torch.empty(EN.GROTH16_COMBINE_WORKSPACE, dtype=torch.uint8, device=dev)
        engine.groth16_prove_partial_dev(key, agg.handle, 1, agg.wit, agg.inst, rs, part, pws, ws_bytes, bad, s0)
        engine.groth16_prove_combine_dev(key, 1, part, rs[0], proof, cws, cws.numel(), s0)
        torch.cuda.synchronize()
        assert proof.cpu().numpy().view(np.uint64)[0].tolist() == want_proof.tolist(), "partial + combine of a whole key"
        del pws
    finally:
        engine.groth16_pk_free(key)
    for world in slices:
        parts = torch.full((world, EN.GROTH16_PARTIAL_WORDS), -1, dtype=torch.int64, device=dev)
        covered_z, covered_h = 0, 0
        last = None
        for rank in range(world):
            key, _ = engine.groth16_setup_r1cs(agg.handle, *tox, mode=EN.KEY_BARE, rank=rank, world=world, want_vk=False)
            try:
                info = engine.groth16_pk_info(key)
                assert (info.rank, info.world) == (rank, world) and info.z_lo == covered_z and info.h_lo == covered_h
                covered_z, covered_h = int(info.z_hi), int(info.h_hi)
                ws_bytes = engine.groth16_workspace_bytes(key, agg.handle, 1)
                pws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
                bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
                with pytest.raises(frw.FrwError):                  # a slice proves nothing by itself
                    engine.groth16_prove_dev(key, agg.handle, 1, agg.wit, agg.inst, rs, proof, pws, ws_bytes, bad, s0)
                engine.groth16_prove_partial_dev(key, agg.handle, 1, agg.wit, agg.inst, rs, parts[rank:rank + 1], pws, ws_bytes, bad, s0)
                torch.cuda.synchronize()
                assert bad.tolist() == [0]
                del pws
                if rank == world - 1:
                    proof = torch.full((1, 48), -1, dtype=torch.int64, device=dev)
                    cws = torch.empty(EN.GROTH16_COMBINE_WORKSPACE, dtype=torch.uint8, device=dev)
                    engine.groth16_prove_combine_dev(key, world, parts, rs[0], proof, cws, cws.numel(), s0)
                    torch.cuda.synchronize()
                    last = proof.cpu().numpy().view(np.uint64)[0].tolist()
            finally:
                engine.groth16_pk_free(key)
        assert covered_z == agg.ni + agg.nw + 3 and covered_h == (1 << int(agg.info.log_domain_size)) - 1
        assert last == want_proof.tolist(), "a key in %d slices: another proof than the whole key's" % world


def _prove_and_check(engine, oracle, tmp_path, logns, seed, python_pairing, slices=()):
    import torch
    import falcon_r1cs_amd as frw
    from falcon_r1cs_amd import engine as EN
    agg = Aggregate(engine, logns, seed=seed)
    try:
        dev, s0 = agg.dev, agg.s0
        rng = random.Random(seed)
        toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
        # this key through round 4's HOST-side evaluation of the QAP at t (FRW_SETUP_ON_HOST); the keys below are made on the device
        os.environ["FRW_SETUP_ON_HOST"] = "1"
        try:
            key, vk = engine.groth16_setup_r1cs(agg.handle, toxic["alpha"], toxic["beta"], toxic["gamma"], toxic["delta"], toxic["t"], mode=EN.KEY_TABLES)
        finally:
            del os.environ["FRW_SETUP_ON_HOST"]
        try:
            q = engine.qap_info(agg.handle)
            n = int(q.domain_size)
            r, s = rng.randrange(E.R), rng.randrange(E.R)
            ws_bytes = engine.groth16_workspace_bytes(key, agg.handle, 1)
            pws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            proof = torch.full((1, 48), -1, dtype=torch.int64, device=dev)
            bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
            lim = T.ints_to_limbs
            engine.groth16_prove_dev(key, agg.handle, 1, agg.wit, agg.inst, np.array([lim([r, s])]), proof, pws, ws_bytes, bad, s0)
            # the same h on its own
            ws = torch.empty(int(q.workspace_bytes_per_signature), dtype=torch.uint8, device=dev)
            h = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
            engine.qap_witness_map_dev(agg.handle, 1, agg.wit, agg.inst, h, ws, ws.numel(), None, s0)
            torch.cuda.synchronize()
            assert bad.tolist() == [0]
            got_p = proof.cpu().numpy().view(np.uint64)[0]
            az, bz, cz, z_inst, z_wit, terms = statement_products(oracle, tmp_path, agg)
            want_h = oracle.qap_witness_map(az, bz, cz, agg.ni, z_inst)
            h_int = from_montgomery(h[0].cpu().numpy().view(np.uint64))
            assert h_int == T.limbs_to_ints(want_h), "h of the aggregate differs from the oracle's witness map"
            d = qap.Domain(agg.nc + agg.ni)
            assert d.size == n
            zi = T.limbs_to_ints(z_inst)
            a, b, c, _, gamma_abc = E.prove_exponents_from_products(toxic, d, T.limbs_to_ints(az), T.limbs_to_ints(bz), T.limbs_to_ints(cz),
                                                                   zi, terms, h_int, r, s)
            assert E.verify_exponents({"toxic": toxic, "gamma_abc": gamma_abc}, zi[1:], (a, b, c))
            gen = oracle.g1_generator()
            assert got_p[:12].tolist() == oracle.g1_scalar_mul(gen, a).tolist(), "A"
            assert got_p[12:36].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, b)), "B"
            assert got_p[36:].tolist() == oracle.g1_scalar_mul(gen, c).tolist(), "C"
            for i in (0, 1, agg.ni // 2, agg.ni - 1):
                assert vk["gamma_abc_g1"][i].tolist() == oracle.g1_scalar_mul(gen, gamma_abc[i]).tolist(), i
            ver = frw.Groth16Verifier(vk)
            inst_h = agg.inst.cpu().numpy().view(np.uint64)
            assert ver.verify(inst_h, got_p[None]).tolist() == [1]
            other = inst_h.copy()
            other[0, agg.ni - 3, 0] ^= np.uint64(1)                      # a public input of the LAST statement
            assert ver.verify(other, got_p[None]).tolist() == [0]
            ver.close()
            if python_pairing:
                # ark-groth16's verify_proof restated with a real pairing (oracle/bls12_381.py), nothing known in the exponent
                vk_pts = {"alpha_g1": E.from_limbs(vk["alpha_g1"]), "beta_g2": E.g2_from_limbs(vk["beta_g2"]),
                          "gamma_g2": E.g2_from_limbs(vk["gamma_g2"]), "delta_g2": E.g2_from_limbs(vk["delta_g2"]),
                          "gamma_abc_g1": [E.from_limbs(row) for row in vk["gamma_abc_g1"]]}
                pr = (E.from_limbs(got_p[:12]), E.g2_from_limbs(got_p[12:36]), E.from_limbs(got_p[36:]))
                assert E.verify_proof(vk_pts, zi[1:], pr), "verify_proof rejects the aggregate proof"
        finally:
            engine.groth16_pk_free(key)
        del pws, ws, h
        _bare_and_sliced_keys_give_the_same_bytes(engine, agg, toxic, vk, r, s, got_p, slices, tmp_path,
                                                  zi + T.limbs_to_ints(z_wit) if len(agg.logns) <= 4 else None)
    finally:
        agg.close()


def test_one_proof_for_four_mixed_statements(engine, oracle, tmp_path):
    """(1024, 512, 1024, 1024): three runs, the 2^20 domain (6 + 5 + 5 + 4 stages)."""
    _prove_and_check(engine, oracle, tmp_path, (10, 9, 10, 10), seed=20, python_pairing=True, slices=(2, 3))


def test_one_proof_for_sixteen_falcon1024_statements(engine, oracle, tmp_path):
    """The 2^22 domain (6 + 6 + 5 + 5 stages): 2.5 M variables, 4.2 M points of h_query, 53 GB of window tables."""
    _prove_and_check(engine, oracle, tmp_path, (10,) * 16, seed=22, python_pairing=False, slices=(4,))


# ---- BASELINE configs[4] as written: ONE proof for 1,024 mixed statements (and the sizes on the way there) -------------------------------
BENCH_SEED = 0x46414C434F4E31                                      # bench.py SEED: the mix its aggregate_1024_mixed line draws


def configs4_mix(total=1024):
    rng = random.Random(BENCH_SEED)
    return [rng.choice([9, 10]) for _ in range(total)]


def _prove_large_and_check_by_properties(engine, oracle, logns, seed, log_domain, vouch_for_vk):
    """Statements whose vectors no Python integer arithmetic can visit (10^7 .. 10^8 elements): what the domain offers instead of an
    element-by-element oracle --
      * the witness satisfies the system row by row (0 violated rows, decided on the device) and deg h <= n - 2;
      * h_acc = sum h_i h_query[i], the 2^24 / 2^27-point sum through the bare handle, equals (h(t) zt / delta) G1: ONE fixed-base
        multiple of a field element -- h(t) by Horner's rule on the device (frw_diag_poly_eval_dev), the multiple by the oracle on the CPU;
      * the proof is accepted by the product's pairing verifier for the statement's public inputs (e(A, B) = e(alpha, beta) e(x, gamma)
        e(C, delta) holds only if ALL five sums are right) and rejected with one input of the LAST statement changed.
    The same code paths are held to the exponent prover byte for byte at 2^20 and 2^22 above."""
    import torch
    import falcon_r1cs_amd as frw
    from falcon_r1cs_amd import engine as EN
    torch.cuda.empty_cache()
    agg = Aggregate(engine, logns, seed=seed)
    try:
        dev, s0 = agg.dev, agg.s0
        assert int(agg.info.log_domain_size) == log_domain and agg.info.witness_map_on_device
        n = 1 << log_domain
        assert n >= agg.nc + agg.ni > n // 2
        rng = random.Random(seed)
        toxic = {k: rng.randrange(2, E.R) for k in ("alpha", "beta", "gamma", "delta", "t")}
        key, vk = engine.groth16_setup_r1cs(agg.handle, toxic["alpha"], toxic["beta"], toxic["gamma"], toxic["delta"], toxic["t"])
        try:
            info = engine.groth16_pk_info(key)
            assert info.mode == EN.KEY_BARE, "FRW_KEY_AUTO keeps window tables for a statement of this size"
            nv = agg.ni + agg.nw
            assert info.key_bytes == 3 * (nv + 3) * 112 + (nv + 3) * 224 + (n - 1) * 112
            r, s = rng.randrange(E.R), rng.randrange(E.R)
            lim = T.ints_to_limbs
            ws_bytes = engine.groth16_workspace_bytes(key, agg.handle, 1)
            pws = torch.empty(ws_bytes, dtype=torch.uint8, device=dev)
            proof = torch.full((1, 48), -1, dtype=torch.int64, device=dev)
            bad = torch.full((1,), -1, dtype=torch.int32, device=dev)
            engine.groth16_prove_dev(key, agg.handle, 1, agg.wit, agg.inst, np.array([lim([r, s])]), proof, pws, ws_bytes, bad, s0)
            torch.cuda.synchronize()
            assert bad.tolist() == [0], "violated rows"
            got_p = proof.cpu().numpy().view(np.uint64)[0]
            # h on its own (the prover's workspace lends the memory), its top coefficient, h(t), and the sum over h_query
            q = engine.qap_info(agg.handle)
            assert int(q.domain_size) == n and int(q.workspace_bytes_per_signature) <= ws_bytes
            h = torch.empty((1, n, 4), dtype=torch.int64, device=dev)
            engine.qap_witness_map_dev(agg.handle, 1, agg.wit, agg.inst, h, pws, ws_bytes, bad, s0)
            torch.cuda.synchronize()
            assert bad.tolist() == [0] and int(h[0, -1].abs().sum()) == 0, "deg h must be <= n - 2"
            h_t = engine.diag_poly_eval_dev(h, n, toxic["t"])
            hq = engine.groth16_pk_query(key, 0)
            mi = engine.msm_info(hq)
            assert mi.num_points == n - 1 and mi.table_bytes == (n - 1) * 112
            assert int(mi.workspace_bytes_per_signature) <= ws_bytes
            hacc = torch.full((1, 12), -1, dtype=torch.int64, device=dev)
            engine.groth16_msm_h_dev(hq, 1, h, n, hacc, pws, ws_bytes, s0)
            torch.cuda.synchronize()
            zt = (pow(toxic["t"], n, E.R) - 1) % E.R
            want = oracle.g1_scalar_mul(oracle.g1_generator(), h_t * zt * pow(toxic["delta"], -1, E.R) % E.R)
            assert hacc.cpu().numpy().view(np.uint64)[0].tolist() == want.tolist(), "h_acc differs from (h(t) zt / delta) G1"
            del h, pws
        finally:
            engine.groth16_pk_free(key)
        gen = oracle.g1_generator()
        assert vk["alpha_g1"].tolist() == oracle.g1_scalar_mul(gen, toxic["alpha"]).tolist()
        assert vk["delta_g2"].tolist() == E.g2_to_limbs(E.g2_mul(E.G2, toxic["delta"]))
        ver = frw.Groth16Verifier(vk, points_are_checked=vouch_for_vk)
        inst_h = agg.inst.cpu().numpy().view(np.uint64)
        assert ver.verify(inst_h, got_p[None]).tolist() == [1], "frw_groth16_verify rejects the aggregate proof"
        other = inst_h.copy()
        other[0, agg.ni - 3, 0] ^= np.uint64(1)                          # a public input of the LAST statement
        assert ver.verify(other, got_p[None]).tolist() == [0]
        ver.close()
    finally:
        agg.close()
        torch.cuda.empty_cache()


def test_one_proof_for_sixty_four_falcon1024_statements(engine, oracle):
    """The 2^24 domain (6 + 6 + 6 + 6 stages): 10.2 M variables -- past what window tables hold (FRW_KEY_AUTO: bare handles, 7 GB of points
    instead of 200 GB of tables), the key made on the device end to end."""
    _prove_large_and_check_by_properties(engine, oracle, (10,) * 64, seed=64, log_domain=24, vouch_for_vk=False)


@pytest.mark.parametrize("count,log_domain", [(128, 25), (256, 26)], ids=["2^25", "2^26"])
def test_one_proof_on_the_five_pass_domains(engine, oracle, count, log_domain):
    """The transform schedules between the two above: 2^25 = 6 + 5 + 5 + 5 + 4 stages (128 Falcon-1024 statements), 2^26 = 6 + 5 + 5 + 5 + 5
    (256) -- with the 2^27 of the next test, every domain frw.h says has run."""
    _prove_large_and_check_by_properties(engine, oracle, (10,) * count, seed=count, log_domain=log_domain, vouch_for_vk=count > 128)


def test_configs4_one_proof_for_1024_mixed_statements(engine, oracle):
    """BASELINE configs[4] as written: ONE Groth16 proof for 1,024 mixed statements -- the mix bench.py draws (513 Falcon-512 + 511
    Falcon-1024), C + I = 126.6 M: the 2^27 domain (6 + 6 + 5 + 5 + 5 stages), 121.9 M variables, 83 GB of key."""
    logns = configs4_mix()
    assert (logns.count(9), logns.count(10)) == (513, 511)
    _prove_large_and_check_by_properties(engine, oracle, logns, seed=1024, log_domain=27, vouch_for_vk=True)
