"""ORACLE (test infrastructure only) -- restatement of the reference's Falcon R1CS gadgets.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this.  PARITY UNPINNED for full-witness values (see ``oracle/ark_sim.py`` / ``oracle/README.md``).

Every function cites the reference file:line (relative to ``/root/reference/``) it follows and
is run against ``oracle/ark_sim.py`` (the arkworks front-end simulation), so variable order,
witness values and constraints come out of the same call sequence the reference executes.

Third-party arithmetic that is absent from ``/root/reference``:

* ``falcon-rust`` (git ``https://github.com/zhenfeizhang/falcon.rs``, no rev pinned,
  ``falcon-r1cs/Cargo.toml:11``): constants ``MODULUS = 12289``, ``N``, ``LOG_N``,
  ``SIG_L2_BOUND``, ``NTT_TABLE``; ``Polynomial`` (mul = negacyclic product mod q, sub mod q),
  ``NTTPolynomial::from`` (Falcon's ``mq_NTT``).  Restated from the Falcon specification
  (round-3 submission, ``vrfy.c``: ``mq_NTT`` with table ``GMb[i] = R*g^bitrev10(i) mod q``,
  ``g = 7``, ``R = 2^16 mod q = 4091``).  ``script/ntt_param.sage:3-132`` of the reference holds
  that table and divides it by 4091, i.e. ``NTT_TABLE[i] = 7^bitrev10(i) mod q``; this is
  checked against the reference's data file by ``tests/golden/make_golden.py`` (a digest of the
  table is committed in ``tests/golden/ntt_table.json``).
"""
from __future__ import annotations

from .ark_sim import (Boolean, ConstraintSystem, FpVar, P_BLS12_381_FR, SynthesisError)

MODULUS = 12289
GENERATOR = 7  # primitive 2048-th root of unity mod q used by Falcon
SIG_L2_BOUND = {9: 34034726, 10: 70265242}  # falcon-rust SIG_L2_BOUND; range_proofs.rs:104,196


def bitrev10(i: int) -> int:
    return int(format(i, "010b")[::-1], 2)


# falcon-rust NTT_TABLE (1024 entries) == script/ntt_param.sage:3-132 `forward` / 4091 mod q
NTT_TABLE = [pow(GENERATOR, bitrev10(i), MODULUS) for i in range(1024)]


# ---------------------------------------------------------------------------------------
# falcon-rust "in the clear" arithmetic used by falcon_ntt.rs:44-51
# ---------------------------------------------------------------------------------------
def ntt_clear(poly, logn):
    """NTTPolynomial::from(&Polynomial): Falcon mq_NTT (forward negacyclic NTT mod q)."""
    n = 1 << logn
    a = list(poly)
    t = n
    m = 1
    while m < n:
        ht = t >> 1
        j1 = 0
        for i in range(m):
            s = NTT_TABLE[m + i]
            for j in range(j1, j1 + ht):
                u = a[j]
                v = a[j + ht] * s % MODULUS
                a[j] = (u + v) % MODULUS
                a[j + ht] = (u - v) % MODULUS
            j1 += t
        t = ht
        m <<= 1
    return a


def poly_mul_clear(a, b):
    """Polynomial * Polynomial: negacyclic product mod (x^N + 1, q) -- schoolbook, independent of the NTT."""
    n = len(a)
    res = [0] * n
    for i, ai in enumerate(a):
        if ai == 0:
            continue
        for j, bj in enumerate(b):
            k = i + j
            if k < n:
                res[k] += ai * bj
            else:
                res[k - n] -= ai * bj
    return [x % MODULUS for x in res]


def poly_sub_clear(a, b):
    return [(x - y) % MODULUS for x, y in zip(a, b)]


def to_bits_le(value: int, nbits: int):
    """a_val.into_repr().to_bits_le() truncated with .take(nbits)."""
    return [(value >> i) & 1 for i in range(nbits)]


# ---------------------------------------------------------------------------------------
# gadgets/misc.rs
# ---------------------------------------------------------------------------------------
def enforce_decompose(a: FpVar, bits):
    """misc.rs:9-24 -- a = sum 2^i bits[i] (Horner from the top bit); 1 constraint."""
    if not bits:
        raise ValueError("Invalid input length: 0")
    res = FpVar.from_boolean(bits[-1])
    for e in reversed(bits[:-1]):
        res = res.double() + FpVar.from_boolean(e)
    res.enforce_equal(a)


def ntt_param_var(cs, logn):
    """misc.rs:67-77 -- NTT_TABLE[0..N] as constants."""
    n = 1 << logn
    return [FpVar.constant(cs, e) for e in NTT_TABLE[:n]]


def l2_norm_var(cs, inputs, modulus_var):
    """misc.rs:30-51 -- sum of squares of the centred lift; 18 witnesses / 19 constraints per element."""
    res = None
    for e in inputs:
        tmp = FpVar.conditionally_select(is_less_than_6144(cs, e), e, modulus_var - e)
        sq = tmp * tmp
        res = sq if res is None else res + sq
    return res


# ---------------------------------------------------------------------------------------
# gadgets/range_proofs.rs
# ---------------------------------------------------------------------------------------
def enforce_less_than_q(cs, a: FpVar, strict=False):
    """range_proofs.rs:42-94.  ``strict`` = the ``#[cfg(not(test))]`` panic at :57-60."""
    a_val = a.value()
    if strict and a_val >= MODULUS:
        raise ValueError("Invalid input: %d" % a_val)
    a_bit_vars = [Boolean.new_witness(cs, x) for x in to_bits_le(a_val, 14)]
    enforce_decompose(a, a_bit_vars)
    # (a[13]==0) or ((a[12]==0) or (kary_or(a[0..12])==0))  == TRUE      :81-89
    # Rust evaluates the receiver before the argument: none of the receivers allocates.
    recv13 = a_bit_vars[13].is_eq_const(False)
    recv12 = a_bit_vars[12].is_eq_const(False)
    low = Boolean.kary_or(a_bit_vars[0:12]).is_eq_const(False)
    recv13.or_(recv12.or_(low)).enforce_equal_const(True)


def is_less_than_6144(cs, a: FpVar) -> Boolean:
    """range_proofs.rs:289-333 -- (a[13]==0) and ((a[12]==0) or (a[11]==0))."""
    a_val = a.value()
    a_bit_vars = [Boolean.new_witness(cs, x) for x in to_bits_le(a_val, 14)]
    enforce_decompose(a, a_bit_vars)
    r13 = a_bit_vars[13].is_eq_const(False)
    r12 = a_bit_vars[12].is_eq_const(False)
    r11 = a_bit_vars[11].is_eq_const(False)
    return r13.and_(r12.or_(r11)).is_eq_const(True)


def enforce_less_than_norm_bound_512(cs, a: FpVar, strict=False):
    """range_proofs.rs:100-186 -- a < 34034726 = 0b10000001110101010000100110 (26 bits)."""
    a_val = a.value()
    if strict and a_val >= SIG_L2_BOUND[9]:
        raise ValueError("Invalid input: %d" % a_val)
    b = [Boolean.new_witness(cs, x) for x in to_bits_le(a_val, 26)]
    enforce_decompose(a, b)
    F = False
    # Receivers are evaluated before arguments, so the kary_* chains allocate in source order
    # (:150, :152, :166, :170, :172) before any of the nested binary gates.
    r25 = b[25].is_eq_const(F)
    k19 = Boolean.kary_or(b[19:25]).is_eq_const(F)
    k16 = Boolean.kary_and(b[16:19]).is_eq_const(F)
    r15 = b[15].is_eq_const(F)
    r14 = b[14].is_eq_const(F)
    r13 = b[13].is_eq_const(F)
    r12 = b[12].is_eq_const(F)
    r11 = b[11].is_eq_const(F)
    r10 = b[10].is_eq_const(F)
    k6 = Boolean.kary_or(b[6:10]).is_eq_const(F)
    r5 = b[5].is_eq_const(F)
    k3 = Boolean.kary_or(b[3:5]).is_eq_const(F)
    k1 = Boolean.kary_and(b[1:3]).is_eq_const(F)
    # nested binary gates, innermost first
    x = k3.and_(k1)
    x = r5.or_(x)
    x = k6.and_(x)
    x = r10.or_(x)
    x = r11.and_(x)
    x = r12.or_(x)
    x = r13.and_(x)
    x = r14.or_(x)
    x = r15.and_(x)
    x = k16.or_(x)
    x = k19.and_(x)
    x = r25.or_(x)
    x.enforce_equal_const(True)


def enforce_less_than_norm_bound_1024(cs, a: FpVar, strict=False):
    """range_proofs.rs:192-272 -- a < 70265242 = 0b100001100000010100110011010 (27 bits)."""
    a_val = a.value()
    if strict and a_val >= SIG_L2_BOUND[10]:
        raise ValueError("Invalid input: %d" % a_val)
    b = [Boolean.new_witness(cs, x) for x in to_bits_le(a_val, 27)]
    enforce_decompose(a, b)
    F = False
    r26 = b[26].is_eq_const(F)
    k22 = Boolean.kary_or(b[22:26]).is_eq_const(F)
    k20 = Boolean.kary_and(b[20:22]).is_eq_const(F)
    k14 = Boolean.kary_or(b[14:20]).is_eq_const(F)
    r13 = b[13].is_eq_const(F)
    r12 = b[12].is_eq_const(F)
    r11 = b[11].is_eq_const(F)
    k9 = Boolean.kary_or(b[9:11]).is_eq_const(F)
    k7 = Boolean.kary_and(b[7:9]).is_eq_const(F)
    k5 = Boolean.kary_or(b[5:7]).is_eq_const(F)
    k3 = Boolean.kary_and(b[3:5]).is_eq_const(F)
    k1 = Boolean.kary_or(b[1:3]).is_eq_const(F)
    x = k3.or_(k1)
    x = k5.and_(x)
    x = k7.or_(x)
    x = k9.and_(x)
    x = r11.or_(x)
    x = r12.and_(x)
    x = r13.or_(x)
    x = k14.and_(x)
    x = k20.or_(x)
    x = k22.and_(x)
    x = r26.or_(x)
    x.enforce_equal_const(True)


def enforce_less_than_norm_bound(cs, a: FpVar, logn, strict=False):
    """range_proofs.rs:274-284 -- cargo feature falcon-512 / falcon-1024 selects the gadget."""
    if logn == 9:
        enforce_less_than_norm_bound_512(cs, a, strict)
    elif logn == 10:
        enforce_less_than_norm_bound_1024(cs, a, strict)
    else:
        raise ValueError("logn must be 9 or 10")


def enforce_less_than_1024(cs, a: FpVar):
    """range_proofs.rs:13-37 (unused by any circuit; kept for its known-answer tests)."""
    a_bit_vars = [Boolean.new_witness(cs, x) for x in to_bits_le(a.value(), 10)]
    enforce_decompose(a, a_bit_vars)


# ---------------------------------------------------------------------------------------
# gadgets/arithmetics.rs
# ---------------------------------------------------------------------------------------
def mod_q(cs, a: FpVar, modulus_var: FpVar, strict=False) -> FpVar:
    """arithmetics.rs:105-149 -- witnesses [t, b] + enforce_less_than_q(b): 29 w / 30 c."""
    a_int = a.value()
    t_int, b_int = divmod(a_int, MODULUS)
    t_var = FpVar.new_witness(cs, t_int)
    b_var = FpVar.new_witness(cs, b_int)
    left = a - t_var * modulus_var
    left.enforce_equal(b_var)
    enforce_less_than_q(cs, b_var, strict)
    return b_var


def add_mod(cs, a: FpVar, b: FpVar, modulus_var: FpVar, strict=False) -> FpVar:
    """arithmetics.rs:214-262 -- c = a + b mod q; witnesses [t, c] + enforce_less_than_q(c)."""
    ab_int = (a.value() + b.value()) % cs.p
    c_int = ab_int % MODULUS
    t_int = (ab_int - c_int) // MODULUS
    t_var = FpVar.new_witness(cs, t_int)
    c_var = FpVar.new_witness(cs, c_int)
    left = (a + b) - t_var * modulus_var
    left.enforce_equal(c_var)
    enforce_less_than_q(cs, c_var, strict)
    return c_var


def mul_mod(cs, a: FpVar, b: FpVar, modulus_var: FpVar, strict=False) -> FpVar:
    """arithmetics.rs:157-209 (dead code in the reference; kept for its known-answer tests)."""
    ab_int = (a.value() * b.value()) % cs.p
    t_int, c_int = divmod(ab_int, MODULUS)
    t_var = FpVar.new_witness(cs, t_int)
    c_var = FpVar.new_witness(cs, c_int)
    left = (a * b) - t_var * modulus_var
    left.enforce_equal(c_var)
    enforce_less_than_q(cs, c_var, strict)
    return c_var


def sub_mod(cs, a: FpVar, b: FpVar, modulus_var: FpVar, strict=False) -> FpVar:
    """arithmetics.rs:269-302 (dead code in the reference; kept for its known-answer tests)."""
    c_int = (a.value() + MODULUS - b.value() % MODULUS) % MODULUS
    c_var = FpVar.new_witness(cs, c_int)
    a.enforce_equal(add_mod(cs, b, c_var, modulus_var, strict))
    return c_var


# ---------------------------------------------------------------------------------------
# gadgets/poly.rs
# ---------------------------------------------------------------------------------------
def alloc_vars(cs, poly, mode):
    """poly.rs:47-63 / :195-211 -- one variable per coefficient, in index order."""
    alloc = FpVar.new_witness if mode == "Witness" else FpVar.new_input
    return [alloc(cs, v) for v in poly]


def const_q_power_vars(cs, logn):
    """falcon_ntt.rs:31-39 -- [q, 2 q^2, 4 q^3, ..., 2^LOG_N q^(LOG_N+1)] as constants."""
    return [FpVar.constant(cs, (1 << (x - 1)) * MODULUS ** x) for x in range(1, logn + 2)]


def ntt_circuit(cs, input_vars, const_vars, param, logn, strict=False):
    """poly.rs:104-159 -- un-reduced integer butterfly ladder (no witnesses) + N x mod_q."""
    n = 1 << logn
    if len(input_vars) != n:
        raise ValueError("input length %d is not N" % len(input_vars))
    output = list(input_vars)
    t = n
    for l in range(logn):
        m = 1 << l
        ht = t // 2
        j1 = 0
        for i in range(m):
            s = param[m + i]
            for j in range(j1, j1 + ht):
                u = output[j]
                v = output[j + ht] * s
                neg_v = const_vars[l + 1] - v
                output[j] = u + v
                output[j + ht] = u + neg_v
            j1 += t
        t = ht
    return [mod_q(cs, e, const_vars[0], strict) for e in output]


# ---------------------------------------------------------------------------------------
# circuits/falcon_ntt.rs
# ---------------------------------------------------------------------------------------
class FalconNTTVerificationCircuit:
    """falcon_ntt.rs:8-17.  The reference holds (pk, msg, sig); sig/pk decoding and hash-to-point live
    in falcon-rust (absent), so the oracle starts from the three coefficient vectors they produce:
    ``sig``, ``pk`` (falcon_ntt.rs:27-28) and ``hm`` (falcon_ntt.rs:44), all in [0, q)."""

    def __init__(self, sig, pk, hm, logn):
        n = 1 << logn
        assert len(sig) == len(pk) == len(hm) == n
        self.sig, self.pk, self.hm, self.logn = list(sig), list(pk), list(hm), logn

    def generate_constraints(self, cs: ConstraintSystem, strict=False):
        """falcon_ntt.rs:26-123."""
        logn = self.logn
        n = 1 << logn
        consts = const_q_power_vars(cs, logn)                      # :31-39
        param_vars = ntt_param_var(cs, logn)                       # :40
        hm_ntt = ntt_clear(self.hm, logn)                          # :45
        uh = poly_mul_clear(self.sig, self.pk)                     # :48
        v = poly_sub_clear(self.hm, uh)                            # :49
        pk_ntt = ntt_clear(self.pk, logn)                          # :51
        sig_poly_vars = alloc_vars(cs, self.sig, "Witness")        # :58-59
        pk_ntt_vars = alloc_vars(cs, pk_ntt, "Input")              # :63
        hm_ntt_vars = alloc_vars(cs, hm_ntt, "Input")              # :67
        v_vars = alloc_vars(cs, v, "Witness")                      # :71
        for e in v_vars:                                           # :73-77
            enforce_less_than_q(cs, e, strict)
        sig_ntt_vars = ntt_circuit(cs, sig_poly_vars, consts, param_vars, logn, strict)   # :88-89
        v_ntt_vars = ntt_circuit(cs, v_vars, consts, param_vars, logn, strict)            # :90-91
        for i in range(n):                                         # :94-111
            # the product is an argument expression: it is evaluated (and its witness allocated)
            # before add_mod's body runs
            prod = sig_ntt_vars[i] * pk_ntt_vars[i]
            hm_ntt_vars[i].enforce_equal(add_mod(cs, v_ntt_vars[i], prod, consts[0], strict))
        l2 = l2_norm_var(cs, v_vars + sig_poly_vars, consts[0])    # :116-120
        enforce_less_than_norm_bound(cs, l2, logn, strict)         # :122


# ---------------------------------------------------------------------------------------
# gadgets/arithmetics.rs:34-100 + circuits/falcon_schoolbook.rs -- NOT on the product's path (SURVEY section 2 rows 3, 8:
# out of scope).  Restated only because README.md:45,56 publishes this circuit's counts: reproducing them by execution
# pins more of the arkworks front-end simulation (Var*Var products, FpVar::is_eq = AllocatedFp::is_neq with its two
# witnesses, Boolean::or on Not operands) than the NTT circuit alone exercises.
# ---------------------------------------------------------------------------------------
def inner_product_mod(cs, a, b, modulus_var, strict=False):
    """arithmetics.rs:34-100 -- witnesses [t, c] first, then the N products, then enforce_less_than_q(c)."""
    if len(a) != len(b) or not a:
        raise ValueError("Invalid input length")
    ab_int = sum(x.value() * y.value() for x, y in zip(a, b)) % cs.p
    t_int, c_int = divmod(ab_int, MODULUS)
    t_var = FpVar.new_witness(cs, t_int)
    c_var = FpVar.new_witness(cs, c_int)
    ab_var = a[0] * b[0]
    for x, y in zip(a[1:], b[1:]):
        ab_var = ab_var + x * y
    left = ab_var - t_var * modulus_var
    left.enforce_equal(c_var)
    enforce_less_than_q(cs, c_var, strict)
    return c_var


class FalconSchoolBookVerificationCircuit:
    """falcon_schoolbook.rs:26-131, from the coefficient vectors (sig, pk, hm)."""

    def __init__(self, sig, pk, hm, logn):
        self.sig, self.pk, self.hm, self.logn = list(sig), list(pk), list(hm), logn

    def generate_constraints(self, cs: ConstraintSystem, strict=False):
        logn = self.logn
        n = 1 << logn
        const_q_var = FpVar.constant(cs, MODULUS)                                   # :30
        uh = poly_mul_clear(self.sig, self.pk)                                      # :38
        v = poly_sub_clear(self.hm, uh)                                             # :39
        sig_poly_vars = [FpVar.new_witness(cs, e) for e in self.sig]                # :45-59
        pk_poly_vars, neg_pk_poly_vars = [], []
        for e in self.pk:                                                           # :66-75
            tmp = FpVar.new_input(cs, e)
            neg_pk_poly_vars.append(const_q_var - tmp)
            pk_poly_vars.append(tmp)
        hm_vars = [FpVar.new_input(cs, e) for e in self.hm]                         # :78-83
        v_pos_vars = []
        for e in v:                                                                 # :86-93
            tmp = FpVar.new_witness(cs, e)
            enforce_less_than_q(cs, tmp, strict)
            v_pos_vars.append(tmp)
        buf = (neg_pk_poly_vars + pk_poly_vars)[::-1]                               # :103-104
        for i in range(n):                                                          # :106-125
            current_col = inner_product_mod(cs, sig_poly_vars, buf[n - 1 - i:2 * n - 1 - i], const_q_var, strict)
            rhs = hm_vars[i] + const_q_var - current_col
            first = rhs.is_eq(v_pos_vars[i])
            second = rhs.is_eq(v_pos_vars[i] + const_q_var)
            first.or_(second).enforce_equal_const(True)
        l2 = l2_norm_var(cs, v_pos_vars + sig_poly_vars, const_q_var)               # :130-134
        enforce_less_than_norm_bound(cs, l2, logn, strict)                          # :135


# ---------------------------------------------------------------------------------------
# gadgets/dual_poly.rs + circuits/falcon_dual_ntt.rs  
# ---------------------------------------------------------------------------------------
MODULUS_OVER_TWO = 6144     # falcon-rust: threshold of the signed lift; unpinned (falcon-rust is not under /root/reference)


def dual_from_poly(poly):
    """falcon-rust DualPolynomial::from(&Polynomial): c < q/2 -> (c, 0) else (0, q - c); pos - neg = poly mod q."""
    pos = [c if c < MODULUS_OVER_TWO else 0 for c in poly]
    neg = [0 if c < MODULUS_OVER_TWO else MODULUS - c for c in poly]
    return pos, neg


def l2_norm_var_without_range_check(inputs):
    """misc.rs:55-65 -- sum of squares, one product witness per element."""
    res = None
    for e in inputs:
        sq = e * e
        res = sq if res is None else res + sq
    return res


def dual_poly_alloc_vars(cs, pos, neg, mode):
    """dual_poly.rs:15-31 -- pos, neg, then sum(pos[i]*neg[i]) == 0 via is_zero (N products + 2 witnesses)."""
    p = alloc_vars(cs, pos, mode)
    n = alloc_vars(cs, neg, mode)
    acc = p[0] * n[0]
    for a, b in zip(p[1:], n[1:]):
        acc = acc + a * b
    acc.is_zero().enforce_equal_const(True)
    return p, n


class FalconDualNTTVerificationCircuit:
    """falcon_dual_ntt.rs:8-17, built from the coefficient vectors falcon-rust derives (:27-28,:43)."""

    def __init__(self, sig, pk, hm, logn):
        n = 1 << logn
        assert len(sig) == len(pk) == len(hm) == n
        self.sig, self.pk, self.hm, self.logn = list(sig), list(pk), list(hm), logn

    def generate_constraints(self, cs: ConstraintSystem, strict=False):
        """falcon_dual_ntt.rs:26-132."""
        logn = self.logn
        n = 1 << logn
        sig_pos, sig_neg = dual_from_poly(self.sig)                # :27  (DualPolynomial from the signed signature)
        consts = const_q_power_vars(cs, logn)                      # :31-39
        param_vars = ntt_param_var(cs, logn)                       # :40
        hm_ntt = ntt_clear(self.hm, logn)                          # :45
        uh_pos = poly_mul_clear(sig_pos, self.pk)                  # :48
        uh_neg = poly_mul_clear(sig_neg, self.pk)                  # :49
        v = [(h - a + b) % MODULUS for h, a, b in zip(self.hm, uh_pos, uh_neg)]   # :50
        v_pos, v_neg = dual_from_poly(v)                           # :51
        pk_ntt = ntt_clear(self.pk, logn)                          # :53
        sp, sn = dual_poly_alloc_vars(cs, sig_pos, sig_neg, "Witness")             # :60-61
        pk_ntt_vars = alloc_vars(cs, pk_ntt, "Input")              # :65
        hm_ntt_vars = alloc_vars(cs, hm_ntt, "Input")              # :69
        vp, vn = dual_poly_alloc_vars(cs, v_pos, v_neg, "Witness") # :73
        sp_ntt = ntt_circuit(cs, sp, consts, param_vars, logn, strict)             # :85-90 (pos then neg, dual_poly.rs:47-48)
        sn_ntt = ntt_circuit(cs, sn, consts, param_vars, logn, strict)
        vp_ntt = ntt_circuit(cs, vp, consts, param_vars, logn, strict)             # :91-92
        vn_ntt = ntt_circuit(cs, vn, consts, param_vars, logn, strict)
        for i in range(n):                                         # :95-116
            left = mod_q(cs, hm_ntt_vars[i] + vn_ntt[i] + sn_ntt[i] * pk_ntt_vars[i], consts[0], strict)
            right = mod_q(cs, vp_ntt[i] + sp_ntt[i] * pk_ntt_vars[i], consts[0], strict)
            left.enforce_equal(right)
        l2 = l2_norm_var_without_range_check(vp + vn + sp + sn)    # :121-129
        enforce_less_than_norm_bound(cs, l2, logn, strict)         # :131


def run_reference_flow_dual(sig, pk, hm, logn, strict=False):
    cs = ConstraintSystem()
    FalconDualNTTVerificationCircuit(sig, pk, hm, logn).generate_constraints(cs, strict)
    return cs


# ---------------------------------------------------------------------------------------
# Encodings of the assignment vectors
# ---------------------------------------------------------------------------------------
R_MONT = (1 << 256) % P_BLS12_381_FR  # ark-ff Fp256 Montgomery radix


def encode_elements(values, montgomery: bool) -> bytes:
    """Field elements as 4 x u64 little-endian limbs; Montgomery form is what ark-ff's Fp256 stores."""
    out = bytearray()
    for v in values:
        x = (v * R_MONT) % P_BLS12_381_FR if montgomery else v
        out += x.to_bytes(32, "little")
    return bytes(out)


class FalconAggregateVerificationCircuit:
    """An aggregate statement (BASELINE configs[4]).  The reference's falcon-aggregate-sig crate is an
    empty stub (falcon-aggregate-sig/src/main.rs:1-3), so there is no reference behaviour to restate beyond what its own
    circuit fixes: FalconNTTVerificationCircuit::generate_constraints (falcon_ntt.rs:26-123) once per statement, in order,
    on ONE constraint system -- public inputs then come out as [pk_ntt_0, hm_ntt_0, pk_ntt_1, hm_ntt_1, ...] (the order
    examples/pok_sig.rs:38-45 builds for one statement), witnesses and constraints as the statements' own, end to end."""

    def __init__(self, statements):
        """statements: [(sig, pk, hm, logn), ...]"""
        self.statements = [FalconNTTVerificationCircuit(sig, pk, hm, logn) for sig, pk, hm, logn in statements]

    def generate_constraints(self, cs: ConstraintSystem, strict=False):
        for c in self.statements:
            c.generate_constraints(cs, strict)


def run_reference_flow_aggregate(statements, strict=False):
    cs = ConstraintSystem()
    FalconAggregateVerificationCircuit(statements).generate_constraints(cs, strict)
    return cs


def run_reference_flow(sig, pk, hm, logn, strict=False):
    """One fresh ConstraintSystem + generate_constraints, as falcon_ntt.rs:143-151 does."""
    cs = ConstraintSystem()
    FalconNTTVerificationCircuit(sig, pk, hm, logn).generate_constraints(cs, strict)
    return cs
