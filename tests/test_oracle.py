"""CPU tests that pin the ORACLE (oracle/) to everything the reference publishes for this path:

* README.md:41-56 count table (variables / constraints per circuit and per ntt conversion);
* every known-answer case of the reference's gadget unit tests, incl. the must-be-unsatisfied ones
  (arithmetics.rs:341-373,413-435,475-507,547-590; range_proofs.rs:360-418,437-504,524-577,596-648);
* poly.rs:252-301 / constraint_counts.rs:107-112: ntt_circuit outputs == the Falcon NTT of the input;
* the NTT table against script/ntt_param.sage (digest fixture, tests/golden/ntt_table.json);
and the closed-form C oracle to the gadget-by-gadget Python oracle, bit for bit, in both encodings.
"""
import glob
import hashlib
import json
import os
import random

import numpy as np
import pytest

import frw_testlib as T
from oracle import falcon_gadgets as G
from oracle.ark_sim import Boolean, ConstraintSystem, FpVar

Q = G.MODULUS
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


# ---------------------------------------------------------------------------------------------
# README count table
# ---------------------------------------------------------------------------------------------
README_COUNTS = {  # logn: (instance, witness, constraints) "verify with ntt"; (0, w, c) "ntt conversion"
    10: ((2049, 156724, 162870), (0, 29696, 30720)),     # README.md:43-44
    9: ((1025, 78386, 81460), (0, 14848, 15360)),        # README.md:54-55
}


@pytest.mark.parametrize("logn", [9, 10])
def test_counts_match_reference_readme(oracle, logn):
    rng = random.Random(logn)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    cs = G.run_reference_flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    assert (cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints()) == README_COUNTS[logn][0]
    assert cs.is_satisfied()                                            # falcon_ntt.rs:159
    # examples/constraint_counts.rs:74-113: deltas of one ntt_circuit on a random polynomial
    cs2 = ConstraintSystem()
    poly = [rng.randrange(Q) for _ in range(1 << logn)]
    pv = G.alloc_vars(cs2, poly, "Witness")
    w0, c0, i0 = cs2.num_witness_variables(), cs2.num_constraints(), cs2.num_instance_variables()
    out = G.ntt_circuit(cs2, pv, G.const_q_power_vars(cs2, logn), G.ntt_param_var(cs2, logn), logn)
    assert (cs2.num_instance_variables() - i0, cs2.num_witness_variables() - w0,
            cs2.num_constraints() - c0) == README_COUNTS[logn][1]
    assert [o.value() for o in out] == G.ntt_clear(poly, logn)          # constraint_counts.rs:107-112
    assert cs2.is_satisfied()
    # the closed-form layouts (oracle C and product header) say the same
    L = oracle.layout(logn)
    assert (L.num_instance, L.num_witness, L.num_constraints) == README_COUNTS[logn][0]


def test_per_gadget_costs():
    """Witness / constraint cost of each gadget: 27/29, 29/30, 30/32 (incl. product), 18/19, 50/52, 52/54."""
    def cost(fn):
        cs = ConstraintSystem()
        a = FpVar.new_witness(cs, 42)
        b = FpVar.new_witness(cs, 4242)
        w0, c0 = cs.num_witness_variables(), cs.num_constraints()
        fn(cs, a, b)
        return cs.num_witness_variables() - w0, cs.num_constraints() - c0
    qv = lambda cs: FpVar.constant(cs, Q)
    assert cost(lambda cs, a, b: G.enforce_less_than_q(cs, a)) == (27, 29)
    assert cost(lambda cs, a, b: G.mod_q(cs, a, qv(cs))) == (29, 30)
    assert cost(lambda cs, a, b: G.add_mod(cs, a, a * b, qv(cs))) == (30, 31)   # + enforce_equal at the call site = 32
    assert cost(lambda cs, a, b: G.l2_norm_var(cs, [a], qv(cs))) == (18, 19)
    assert cost(lambda cs, a, b: G.enforce_less_than_norm_bound(cs, a, 9)) == (50, 52)
    assert cost(lambda cs, a, b: G.enforce_less_than_norm_bound(cs, a, 10)) == (52, 54)


# ---------------------------------------------------------------------------------------------
# known-answer tests of the reference's gadget unit tests
# ---------------------------------------------------------------------------------------------
def _binary_gadget_case(gadget, a, b, c, satisfied):
    """arithmetics.rs test_{mul,add,sub}_mod! macros."""
    cs = ConstraintSystem()
    a_var, b_var = FpVar.new_witness(cs, a), FpVar.new_witness(cs, b)
    c_var = gadget(cs, a_var, b_var, FpVar.constant(cs, Q))
    c_var.enforce_equal(FpVar.new_witness(cs, c))
    assert cs.is_satisfied() == satisfied
    assert (c_var.value() == c) == satisfied


def _mod_q_case(a, b, satisfied):
    """arithmetics.rs:311-338."""
    cs = ConstraintSystem()
    a_var = FpVar.new_witness(cs, a)
    b_var = G.mod_q(cs, a_var, FpVar.constant(cs, Q))
    b_var.enforce_equal(FpVar.new_witness(cs, b))
    assert cs.is_satisfied() == satisfied
    assert (b_var.value() == b) == satisfied


def test_kat_mod_q():
    for a, b, ok in [(6, 6, True), (0, 0, True), (Q, 0, True), (Q + 1, 1, True), (6, 7, False), (5, Q - 1, False)]:
        _mod_q_case(a, b, ok)                                           # arithmetics.rs:346-360
    rng = random.Random(1)
    for _ in range(200):                                                # :365-371 (1000 in the reference)
        t = rng.randrange(1 << 30)
        _mod_q_case(t, t % Q, True)
        _mod_q_case(t, (t + 1) % Q, False)


def test_kat_mul_add_sub_mod():
    for a, b, c, ok in [(6, 7, 42, True), (0, 100, 0, True), (100, 0, 0, True), (5, 12288, 12284, True),
                        (6, 7, 41, False), (5, 12288, 12283, False)]:  # arithmetics.rs:418-432
        _binary_gadget_case(G.mul_mod, a, b, c, ok)
    for a, b, c, ok in [(6, 36, 42, True), (0, 100, 100, True), (100, 0, 100, True), (5, Q - 1, 4, True),
                        (6, 7, 41, False), (5, Q - 1, 3, False)]:      # :480-494
        _binary_gadget_case(G.add_mod, a, b, c, ok)
    for a, b, c, ok in [(78, 36, 42, True), (0, 0, 0, True), (100, 0, 100, True), (0, 100, 12189, True),
                        (78, 36, 41, False), (0, 100, 12188, False)]:  # :552-566
        _binary_gadget_case(G.sub_mod, a, b, c, ok)
    rng = random.Random(2)
    for _ in range(100):                                                # :499-505, :571-588
        a, b = rng.randrange(1 << 30), rng.randrange(1 << 30)
        _binary_gadget_case(G.add_mod, a, b, (a + b) % Q, True)
        _binary_gadget_case(G.add_mod, a, b, (a + b + 1) % Q, False)
        a, b = rng.randrange(Q), rng.randrange(Q)
        _binary_gadget_case(G.sub_mod, a, b, (a - b) % Q, True)
        _binary_gadget_case(G.mul_mod, a, b, a * b % Q, True)


def _unary_case(gadget, value, satisfied):
    cs = ConstraintSystem()
    gadget(cs, FpVar.new_witness(cs, value))
    assert cs.is_satisfied() == satisfied, value


def test_kat_enforce_less_than_q():
    for v, ok in [(42, True), (0, True), (1 << 12, True), (1 << 13, True), (Q - 1, True),
                  (Q, False), (Q + 1, False), (Q * 10000, False)]:      # range_proofs.rs:365-389
        _unary_case(G.enforce_less_than_q, v, ok)
    rng = random.Random(3)
    for _ in range(300):                                                # :394-398
        t = rng.randrange(1 << 15)
        _unary_case(G.enforce_less_than_q, t, t < Q)


@pytest.mark.parametrize("logn", [9, 10])
def test_kat_norm_bound(logn):
    bound = G.SIG_L2_BOUND[logn]
    cases = [(42, True), (0, True), (1 << 25, True), (1 << 24, True), (bound - 1, True),
             (bound, False), (bound + 1, False), (1 << 27, False)]     # range_proofs.rs:442-474
    cases.append((1 << 26, logn == 10))                                 # 2^26: good for 1024, bad for 512
    gadget = lambda cs, a: G.enforce_less_than_norm_bound(cs, a, logn)
    for v, ok in cases:
        _unary_case(gadget, v, ok)
    rng = random.Random(4)
    for _ in range(300):                                                # :479-502
        t = rng.randrange(1 << 27)
        _unary_case(gadget, t, t < bound)
    for t in range(bound - 40, bound + 40):                             # every bit pattern around the bound
        _unary_case(gadget, t, t < bound)


def test_kat_is_less_than_6144_and_1024():
    def half_q(cs, a):                                                  # range_proofs.rs:505-520
        G.is_less_than_6144(cs, a).enforce_equal_const(True)
    for v, ok in [(42, True), (0, True), (6143, True), (6144, False), (6145, False), (Q, False)]:   # :529-547
        _unary_case(half_q, v, ok)
    rng = random.Random(5)
    for _ in range(300):
        t = rng.randrange(Q)
        _unary_case(half_q, t, t < 6144)
    for v, ok in [(42, True), (0, True), (1023, True), (1024, False), (1025, False)]:               # :601-619
        _unary_case(G.enforce_less_than_1024, v, ok)
    for _ in range(100):
        t = rng.randrange(2048)
        _unary_case(G.enforce_less_than_1024, t, t < 1024)


# ---------------------------------------------------------------------------------------------
# NTT: table provenance and independent evaluation
# ---------------------------------------------------------------------------------------------
def test_ntt_table_digest_fixture():
    fx = json.load(open(os.path.join(GOLDEN, "ntt_table.json")))
    assert fx["checked_against_reference_sage"]        # make_golden.py compared it with script/ntt_param.sage:3-132
    blob = np.array(G.NTT_TABLE, dtype=np.uint16).tobytes()
    assert hashlib.sha256(blob).hexdigest() == fx["sha256"]
    assert G.NTT_TABLE[:8] == fx["first8"]


@pytest.mark.parametrize("logn", [9, 10])
def test_ntt_is_evaluation_at_odd_powers(oracle, logn):
    """Output k of the ladder schedule (poly.rs:115-149), reduced mod q, is the input polynomial evaluated at
    psi^(2*bitrev(k)+1), psi = 7^(1024/N): checked by direct Horner evaluation, no butterflies involved."""
    n = 1 << logn
    rng = np.random.default_rng(logn)
    poly = rng.integers(0, Q, size=n, dtype=np.uint16)
    got = oracle.ntt_clear(logn, poly)
    assert got.tolist() == G.ntt_clear(poly.tolist(), logn)
    psi = pow(7, 1024 // n, Q)
    for k in list(range(8)) + [n // 2, n - 1]:
        br = int(format(k, "0%db" % logn)[::-1], 2)
        x = pow(psi, 2 * br + 1, Q)
        acc = 0
        for c in reversed(poly.tolist()):
            acc = (acc * x + c) % Q
        assert acc == int(got[k])
    assert np.array_equal(oracle.ntt_clear(logn, got, inverse=True), poly)


# ---------------------------------------------------------------------------------------------
# closed-form C oracle == gadget-by-gadget Python oracle == golden fixtures
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("logn", [9, 10])
def test_c_oracle_equals_gadget_oracle(oracle, logn):
    rng = random.Random(77 + logn)
    sig, pk, hm, v = T.random_triple(logn, rng)
    cs = G.run_reference_flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    for enc in (0, 1):
        wit, inst, st = oracle.witness_ntt_verify(logn, sig, pk, hm, enc)
        assert st[0] == 0
        assert wit.tobytes() == G.encode_elements(cs.witness_assignment, enc == 1)
        assert inst.tobytes() == G.encode_elements(cs.instance_assignment, enc == 1)


def test_c_oracle_permissive_norm_violation_is_unsatisfied(oracle):
    """cfg(test) behaviour (range_proofs.rs:112-117): truncated bits, unsatisfied system; same bytes in C."""
    logn = 9
    rng = random.Random(9)
    sig, pk, hm, v = T.random_triple(logn, rng, scale=1.6)
    assert T.centred_norm(sig, v) >= G.SIG_L2_BOUND[logn]
    cs = G.run_reference_flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=False)
    assert not cs.is_satisfied()
    with pytest.raises(ValueError):
        G.run_reference_flow(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    wit, inst, st = oracle.witness_ntt_verify(logn, sig, pk, hm, 1)
    assert st[0] == 2
    assert wit.tobytes() == G.encode_elements(cs.witness_assignment, True)


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "witness_*.json"))))
def test_c_oracle_matches_golden(oracle, path):
    fx = json.load(open(path))
    logn = fx["logn"]
    sig, pk, hm = (np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16) for k in ("sig", "pk", "hm"))
    L = oracle.layout(logn)
    assert (L.num_instance, L.num_witness, L.num_constraints) == (fx["num_instance"], fx["num_witness"],
                                                                  fx["num_constraints"])
    for enc, name in ((0, "canonical"), (1, "montgomery")):
        wit, inst, st = oracle.witness_ntt_verify(logn, sig, pk, hm, enc)
        assert st[0] == 0
        assert hashlib.sha256(wit.tobytes()).hexdigest() == fx["witness_sha256"][name]
        assert hashlib.sha256(inst.tobytes()).hexdigest() == fx["instance_sha256"][name]
    # sampled quotients of the first mod_q blocks (canonical limbs -> python int)
    wit, _, _ = oracle.witness_ntt_verify(logn, sig, pk, hm, 0)
    n = 1 << logn
    for k in range(4):
        limbs = wit[0, 29 * n + 29 * k]
        assert sum(int(x) << (64 * i) for i, x in enumerate(limbs)) == int(fx["sample_t"][k])
        assert int(wit[0, 29 * n + 29 * k + 1, 0]) == fx["sample_b"][k]


def test_montgomery_constants():
    """R = 2^256 mod p as ark-ff stores one(); SURVEY 8-a quotes its limbs."""
    p = G.P_BLS12_381_FR if hasattr(G, "P_BLS12_381_FR") else None
    from oracle.ark_sim import P_BLS12_381_FR as p
    assert p == 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
    one = G.encode_elements([1], True)
    limbs = [int.from_bytes(one[8 * i:8 * i + 8], "little") for i in range(4)]
    assert limbs == [0x00000001FFFFFFFE, 0x5884B7FA00034802, 0x998C4FEFECBC4FF5, 0x1824B159ACC5056F]


# ---------------------------------------------------------------------------------------------
# the product's synthetic input generator produces valid statements (checked with the schoolbook product)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("logn", [9, 10])
def test_synth_triples_are_valid_statements(logn):
    import falcon_r1cs_amd as frw
    sig, pk, hm = frw.synth_triples(logn, 3, seed=11, first_index=5)
    again = frw.synth_triples(logn, 1, seed=11, first_index=6)
    assert np.array_equal(again[0][0], sig[1]) and np.array_equal(again[2][0], hm[1])     # counter based
    for i in range(3):
        assert sig[i].max() < Q and pk[i].max() < Q and hm[i].max() < Q
        v = (hm[i].astype(np.int64) - T.negacyclic_mul(sig[i], pk[i])) % Q
        norm = T.centred_norm(sig[i], v)
        assert norm < T.SIG_L2_BOUND[logn]
        sigma2 = norm / (2 << logn)
        assert 0.8 * T.SIGMA[logn] ** 2 < sigma2 < 1.2 * T.SIGMA[logn] ** 2


# ---------------------------------------------------------------------------------------------
# input preparation: codec restatement self-consistency + SHAKE256 anchor
# ---------------------------------------------------------------------------------------------
def test_codec_roundtrip_and_strictness():
    from oracle import falcon_codec as K
    rng = random.Random(12)
    for logn in (9, 10):
        n = 1 << logn
        pk = [rng.randrange(Q) for _ in range(n)]
        assert K.modq_decode(K.modq_encode(pk, logn), logn) == pk
        enc = bytearray(K.modq_encode(pk, logn))
        enc[0] ^= 1
        assert K.modq_decode(bytes(enc), logn) is None                     # wrong header
        bad = list(pk)
        bad[5] = 0x3FFF                                                     # 14-bit value >= q
        acc = 0
        for c in bad:
            acc = (acc << 14) | c
        assert K.modq_decode(bytes([logn]) + acc.to_bytes(14 * n // 8, "big"), logn) is None
        s2 = [max(-2047, min(2047, round(rng.gauss(0, T.SIGMA[logn])))) for _ in range(n)]
        nonce = bytes(rng.randrange(256) for _ in range(40))
        blob = K.comp_encode(s2, logn, nonce)
        assert len(blob) == K.SIG_LEN[logn]
        got = K.comp_decode(blob, logn)
        assert got is not None and got[0] == nonce and got[1] == [x % Q for x in s2]
        t = bytearray(blob)
        t[-1] |= 1
        assert K.comp_decode(bytes(t), logn) is None                        # non-zero padding
        assert K.comp_decode(blob[:200], logn) is None                      # truncated
        t = bytearray(blob)
        t[0] = 0x50 + logn
        assert K.comp_decode(bytes(t), logn) is None                        # wrong header
        # "-0": sign bit set, magnitude 0 -> first body byte 0b1000_0000 followed by the unary terminator
        t = bytearray(blob)
        t[41], t[42] = 0x80, t[42] | 0x80
        assert K.comp_decode(bytes(t), logn) is None


def test_hash_to_point_matches_fips202_stream():
    """hashlib's SHAKE256 is the anchor (FIPS 202 KAT: SHAKE256("") starts 46b9dd2b0ba88d13...)."""
    import hashlib
    from oracle import falcon_codec as K
    assert hashlib.shake_256(b"").hexdigest(8) == "46b9dd2b0ba88d13"
    nonce, msg = bytes(range(40)), b"testing message"
    for logn in (9, 10):
        hm = K.hash_to_point(nonce, msg, logn)
        assert len(hm) == 1 << logn and max(hm) < Q
        stream = hashlib.shake_256(nonce + msg).digest(64)
        first = []
        for i in range(0, 64, 2):
            w = (stream[i] << 8) | stream[i + 1]
            if w < 61445:
                first.append(w % Q)
        assert hm[:len(first)] == first


# ---------------------------------------------------------------------------------------------
# the signed-split variant: falcon_dual_ntt.rs / dual_poly.rs
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("logn", [9, 10])
def test_dual_circuit_oracles_agree_and_are_satisfied(oracle, logn):
    import falcon_r1cs_amd as frw
    rng = random.Random(91 + logn)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    cs = G.run_reference_flow_dual(sig.tolist(), pk.tolist(), hm.tolist(), logn, strict=True)
    assert cs.is_satisfied()                                            # falcon_dual_ntt.rs:168
    L = frw.layout_dual(logn)
    assert (cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints()) == \
        (L.num_instance, L.num_witness, L.num_constraints)
    n = 1 << logn
    assert L.num_witness == 186 * n + 4 + (50 if logn == 9 else 52)
    for enc in (0, 1):
        wit, inst, st = oracle.witness_dual_ntt_verify(logn, sig, pk, hm, enc)
        assert st[0] == 0
        assert wit.tobytes() == G.encode_elements(cs.witness_assignment, enc == 1)
        assert inst.tobytes() == G.encode_elements(cs.instance_assignment, enc == 1)
    # a tampered "pos" part (both pos and neg non-zero at one index) must break the pos*neg = 0 check
    cs2 = ConstraintSystem()
    pos, neg = G.dual_from_poly(sig.tolist())
    pos[3], neg[3] = 5, 7
    G.dual_poly_alloc_vars(cs2, pos, neg, "Witness")
    assert not cs2.is_satisfied()


# ---------------------------------------------------------------------------------------------
# one more published pin of the arkworks front-end simulation: the schoolbook circuit's counts (README.md:45,56)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("logn,want", [(9, (1025, 312882, 315956)), (10, (2049, 1150004, 1156150))])
def test_schoolbook_counts_match_reference_readme(logn, want):
    """falcon_schoolbook.rs is not on the product's path; its counts exercise Var*Var products, FpVar::is_eq
    (AllocatedFp::is_neq: 2 witnesses / 3 constraints -- the rule the dual circuit's is_zero relies on) and
    Boolean::or on Not operands, all reproduced by execution and satisfied."""
    rng = random.Random(50 + logn)
    sig, pk, hm, _ = T.random_triple(logn, rng)
    cs = ConstraintSystem()
    G.FalconSchoolBookVerificationCircuit(sig.tolist(), pk.tolist(), hm.tolist(), logn).generate_constraints(cs, strict=True)
    assert (cs.num_instance_variables(), cs.num_witness_variables(), cs.num_constraints()) == want
    assert cs.is_satisfied()


def test_c_oracle_matches_dual_golden(oracle):
    for path in sorted(glob.glob(os.path.join(GOLDEN, "dual_*.json"))):
        fx = json.load(open(path))
        sig, pk, hm = (np.frombuffer(bytes.fromhex(fx[k]), dtype=np.uint16) for k in ("sig", "pk", "hm"))
        for enc, name in ((0, "canonical"), (1, "montgomery")):
            wit, inst, st = oracle.witness_dual_ntt_verify(fx["logn"], sig, pk, hm, enc)
            assert st[0] == 0 and wit.shape[1] == fx["num_witness"]
            assert hashlib.sha256(wit.tobytes()).hexdigest() == fx["witness_sha256"][name]
            assert hashlib.sha256(inst.tobytes()).hexdigest() == fx["instance_sha256"][name]


# ---------------------------------------------------------------------------------------------------------------------
# genuine Falcon signatures (oracle/falcon_sign.py, tests/golden/falcon_signed.json)
# ---------------------------------------------------------------------------------------------------------------------
def _signed_cases():
    import json
    import os
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "falcon_signed.json")
    return json.load(open(path))["cases"]


def test_signed_fixtures_verify_and_tampering_is_rejected():
    """Falcon spec Alg. 16 on every committed (pk, msg, sig); a flipped message bit, a changed coefficient of s2 and a
    foreign public key must all be rejected."""
    from oracle import falcon_codec as K
    from oracle import falcon_sign as S
    cases = _signed_cases()
    assert sorted(c["logn"] for c in cases) == [9, 9, 10, 10]
    for c in cases:
        logn = c["logn"]
        pkb, msg, sgb = (bytes.fromhex(c[k]) for k in ("pk_bytes", "msg", "sig_bytes"))
        assert len(pkb) == K.PK_LEN[logn] and len(sgb) == K.SIG_LEN[logn]
        assert S.verify(pkb, msg, sgb, logn)
        assert not S.verify(pkb, msg + b"!", sgb, logn)
        nonce, s2 = K.comp_decode(sgb, logn)
        s2 = [x if x <= K.Q // 2 else x - K.Q for x in s2]
        s2[5] += 300
        assert not S.verify(pkb, msg, K.comp_encode(s2, logn, nonce), logn)
        other = [d for d in cases if d["logn"] == logn and d is not c][0]
        assert not S.verify(bytes.fromhex(other["pk_bytes"]), msg, sgb, logn)


def test_signer_is_deterministic_and_keys_solve_the_ntru_equation():
    """keygen + sign from the recorded seeds reproduce the committed Falcon-512 fixtures byte for byte (fixture and
    generator in step), and the key satisfies f G - g F = q with the Gram-Schmidt norm bound of NTRUGen."""
    from oracle import falcon_sign as S
    c = [x for x in _signed_cases() if x["logn"] == 9][0]
    seed, msg = bytes.fromhex(c["key_seed"]), bytes.fromhex(c["msg"])
    sk = S.keygen(9, seed)
    assert sk.public_key_bytes().hex() == c["pk_bytes"]
    assert S.sign(sk, msg, seed).hex() == c["sig_bytes"]
    n = 512
    assert [a - b for a, b in zip(S.pmul(sk.f, sk.G), S.pmul(sk.g, sk.F))] == [S.Q] + [0] * (n - 1)
    assert S.mul_mod_q([x % S.Q for x in sk.f], sk.h) == [x % S.Q for x in sk.g]          # h = g / f mod q
    assert sum(x * x for x in sk.f + sk.g) <= 1.17 ** 2 * S.Q
    # a polynomial product against the schoolbook definition (Kronecker substitution with signed coefficients)
    import random
    rng = random.Random(1)
    a = [rng.randrange(-10 ** 30, 10 ** 30) for _ in range(16)]
    b = [rng.randrange(-10 ** 9, 10 ** 9) for _ in range(16)]
    want = [0] * 16
    for i in range(16):
        for j in range(16):
            k, sgn = (i + j) % 16, (-1 if i + j >= 16 else 1)
            want[k] += sgn * a[i] * b[j]
    assert S.pmul(a, b) == want


def test_signed_fixtures_through_codec_and_closed_form_oracle(oracle):
    """(pk, msg, sig) -> decode + hash_to_point -> (sig, pk, hm) digests -> closed-form C oracle: status OK (the norm bound
    holds for a genuine signature), witness / instance bytes equal the gadget-by-gadget execution recorded in the fixture."""
    import hashlib
    from oracle import falcon_codec as K
    for c in _signed_cases():
        logn = c["logn"]
        pkb, msg, sgb = (bytes.fromhex(c[k]) for k in ("pk_bytes", "msg", "sig_bytes"))
        nonce, sig = K.comp_decode(sgb, logn)
        pk = K.modq_decode(pkb, logn)
        hm = K.hash_to_point(nonce, msg, logn)
        u16 = lambda v: np.array(v, dtype=np.uint16).tobytes()
        assert hashlib.sha256(u16(sig)).hexdigest() == c["sig_sha256"]
        assert hashlib.sha256(u16(pk)).hexdigest() == c["pk_sha256"]
        assert hashlib.sha256(u16(hm)).hexdigest() == c["hm_sha256"]
        wit, inst, st = oracle.witness_ntt_verify(logn, np.array(sig, dtype=np.uint16), np.array(pk, dtype=np.uint16),
                                                  np.array(hm, dtype=np.uint16), 1)
        assert st.tolist() == [0]
        assert hashlib.sha256(wit.tobytes()).hexdigest() == c["witness_sha256_montgomery"]
        assert hashlib.sha256(inst.tobytes()).hexdigest() == c["instance_sha256_montgomery"]
