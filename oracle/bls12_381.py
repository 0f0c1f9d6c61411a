"""ORACLE (test infrastructure, never imported by the product): BLS12-381 G1 and the exponents of a Groth16 proof.

What this is for: the reference's consumer of the witness is `Groth16::<Bls12_381>::prove` (examples/pok_sig.rs:30-47).
After `generate_constraints` (the hot path) and `R1CStoQAP::witness_map` (oracle/qap.py) the prover runs multi-scalar
multiplications over the proving key's queries -- ark-groth16 0.3.0 src/prover.rs, create_proof_with_reduction_and_matrices:

    h_acc     = VariableBaseMSM::multi_scalar_mul(&pk.h_query, &h_assignment)        (2^18 - 1 points for Falcon-1024)
    l_aux_acc = VariableBaseMSM::multi_scalar_mul(&pk.l_query, &aux_assignment)
    g_a  = r delta_g1 + a_query[0] + MSM(a_query[1..], assignment) + alpha_g1         (calculate_coeff)
    g1_b = s delta_g1 + b_g1_query[0] + MSM(b_g1_query[1..], assignment) + beta_g1    (g2_b alike, in G2)
    g_c  = s g_a + r g1_b - r s delta_g1 + l_aux_acc + h_acc

and the proving key comes from src/generator.rs, generate_parameters: with t outside the domain, u_i(t), v_i(t), w_i(t)
from R1CStoQAP::instance_map_with_evaluation and zt = t^n - 1,

    h_query[i] = (zt / delta) t^i G1,  i < n - 1;   a_query[i] = u_i(t) G1;   b_g1_query[i] = v_i(t) G1;
    l_query[i] = ((beta u_i + alpha v_i + w_i)(t) / delta) G1 for the witness variables;  gamma_abc likewise with gamma

PARITY UNPINNED against ark-groth16 / ark-ec / ark-bls12-381 0.3.0 (crates.io dependencies absent from /root/reference,
no Rust toolchain): this restates their published algorithm and the curve's published parameters.  It pins itself by
  * q, r from the BLS parametrisation (z = -0xd201000000010000): r = z^4 - z^2 + 1, q = (z - 1)^2 r / 3 + z;
  * the published generator on y^2 = x^3 + 4 and of order r;
  * the Groth16 verification equation in the exponent -- with the toxic waste known, every element's discrete logarithm is
    an Fr value, e(A, B) = e(alpha, beta) e(sum x_i gamma_abc_i, gamma) e(C, delta) becomes
    a b = alpha beta + (sum x_i abc_i) gamma + c delta  (no pairing needed) -- for proofs of small satisfied systems, and its
    failure for a wrong witness;
  * an MSM's expected value without any MSM: sum h_i h_query[i] = (h(t) zt / delta) G1, one scalar multiplication.
Affine points are (x, y) tuples of Python integers, None = the point at infinity."""
Z_BLS = -0xD201000000010000
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
Q = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
G1 = (0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB,
      0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1)
FQ_R = (1 << 384) % Q            # ark-ff's Montgomery radix for Fp384


def check_parameters():
    assert R == Z_BLS ** 4 - Z_BLS ** 2 + 1
    assert Q == (Z_BLS - 1) ** 2 * R // 3 + Z_BLS and (Z_BLS - 1) ** 2 * R % 3 == 0
    assert on_curve(G1) and mul(G1, R) is None and mul(G1, 1) == G1


def on_curve(p):
    return p is None or (p[1] * p[1] - p[0] * p[0] * p[0] - 4) % Q == 0


def neg(p):
    return None if p is None else (p[0], (-p[1]) % Q)


def add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if (p[1] + q[1]) % Q == 0:
            return None
        lam = 3 * p[0] * p[0] * pow(2 * p[1], -1, Q) % Q
    else:
        lam = (q[1] - p[1]) * pow(q[0] - p[0], -1, Q) % Q
    x = (lam * lam - p[0] - q[0]) % Q
    return x, (lam * (p[0] - x) - p[1]) % Q


def mul(p, k):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = add(acc, p)
        p = add(p, p)
        k >>= 1
    return acc


def msm_naive(bases, scalars):
    acc = None
    for b, k in zip(bases, scalars):
        acc = add(acc, mul(b, k))
    return acc


def to_limbs(p):
    """ark-ff bytes of an affine point: x R, y R mod q as 6 + 6 little-endian u64 limbs; infinity = zeros."""
    if p is None:
        return [0] * 12
    return [(v * FQ_R % Q >> (64 * i)) & (2 ** 64 - 1) for v in p for i in range(6)]


def from_limbs(limbs):
    limbs = [int(v) for v in limbs]
    if not any(limbs):
        return None
    inv = pow(FQ_R, -1, Q)
    return tuple(sum(l << (64 * i) for i, l in enumerate(limbs[6 * c:6 * c + 6])) * inv % Q for c in range(2))


# ---- Groth16 in the exponent ------------------------------------------------------------------------------------------------
def _lagrange_at(domain_size, gen, t):
    """EvaluationDomain::evaluate_all_lagrange_coefficients(t) for t outside the domain: L_i(t) = zt w^i / (n (t - w^i)); the
    n inversions by Montgomery's trick (one modular inverse)."""
    zt = (pow(t, domain_size, R) - 1) % R
    ninv = pow(domain_size, -1, R)
    ws, dens, w = [], [], 1
    for _ in range(domain_size):
        ws.append(w)
        dens.append((t - w) % R)
        w = w * gen % R
    pre, acc = [], 1
    for d in dens:
        pre.append(acc)
        acc = acc * d % R
    inv = pow(acc, -1, R)
    out = [0] * domain_size
    c = zt * ninv % R
    for i in range(domain_size - 1, -1, -1):
        out[i] = c * ws[i] % R * (inv * pre[i] % R) % R
        inv = inv * dens[i] % R
    return out, zt


def setup_exponents(matrices, num_instance, num_witness, domain, toxic):
    """generator.rs generate_parameters, discrete logarithms only.  matrices = (A, B, C) as lists of rows [(coeff, column)],
    columns = instance variables (the constant one first) then witness variables; toxic = dict(alpha, beta, gamma, delta, t).
    domain = oracle.qap.Domain(num_constraints + num_instance)."""
    a_m, b_m, c_m = matrices
    nc, nv = len(a_m), num_instance + num_witness
    t, alpha, beta, gamma, delta = (toxic[k] % R for k in ("t", "alpha", "beta", "gamma", "delta"))
    lag, zt = _lagrange_at(domain.size, domain.group_gen, t)
    u, v, w = [0] * nv, [0] * nv, [0] * nv
    for i in range(num_instance):                      # r1cs_to_qap.rs: a[i] = u[num_constraints + i] for the inputs
        u[i] = lag[nc + i]
    for row in range(nc):
        for vec, mat in ((u, a_m), (v, b_m), (w, c_m)):
            for coeff, col in mat[row]:
                vec[col] = (vec[col] + lag[row] * coeff) % R
    dinv, ginv = pow(delta, -1, R), pow(gamma, -1, R)
    abc = [(beta * u[i] + alpha * v[i] + w[i]) % R for i in range(nv)]
    return {"u": u, "v": v, "zt": zt, "toxic": dict(toxic),
            "gamma_abc": [x * ginv % R for x in abc[:num_instance]],
            "l": [x * dinv % R for x in abc[num_instance:]],
            "h": [zt * dinv % R * pow(t, i, R) % R for i in range(domain.size - 1)]}


def prove_exponents(pk, z, h, r, s):
    """prover.rs create_proof_with_reduction_and_matrices, discrete logarithms of (A, B, C).  z = full assignment (the
    constant one first), h = the witness map's coefficients."""
    ni = len(pk["gamma_abc"])
    tx = pk["toxic"]
    a = (tx["alpha"] + sum(zi * ui for zi, ui in zip(z, pk["u"])) + r * tx["delta"]) % R
    b = (tx["beta"] + sum(zi * vi for zi, vi in zip(z, pk["v"])) + s * tx["delta"]) % R
    h_acc = sum(hi * qi for hi, qi in zip(h, pk["h"])) % R          # zip stops at n - 1 like the prover's MSM
    l_acc = sum(zi * li for zi, li in zip(z[ni:], pk["l"])) % R
    c = (s * a + r * b - r * s * tx["delta"] + l_acc + h_acc) % R
    return a, b, c, h_acc


def prove_exponents_from_products(toxic, domain, az, bz, cz, z_instance, instance_terms, h, r, s):
    """prove_exponents for a system too large to hold as Python rows (an aggregate statement: millions of terms), from what
    the proof actually depends on.  With L_row = the Lagrange coefficients at t and nc = len(az):
        sum_j z_j u_j(t) = sum_row L_row (A z)_row + sum_(i < ni) z_i L_(nc + i)       (r1cs_to_qap.rs: the input rows)
        sum_j z_j v_j(t) = sum_row L_row (B z)_row,   sum_j z_j w_j(t) = sum_row L_row (C z)_row
    and the witness-only sum of l_query is those minus the instance variables' share, for which only the matrix entries in
    instance columns are needed: instance_terms = [(matrix 0/1/2, row, column < ni, coefficient), ...].
    Returns (a, b, c, h_acc, gamma_abc) -- the same (a, b, c, h_acc) prove_exponents gives for the same system (pinned on a
    small one in tests/test_bls12_381.py), and the verifying key's gamma_abc exponents for verify_exponents."""
    t, alpha, beta, gamma, delta = (toxic[k] % R for k in ("t", "alpha", "beta", "gamma", "delta"))
    nc, ni, n = len(az), len(z_instance), domain.size
    lag, zt = _lagrange_at(n, domain.group_gen, t)
    dot = lambda vec: sum(l * x for l, x in zip(lag, vec)) % R
    a_t = (dot(az) + sum(lag[nc + i] * z_instance[i] for i in range(ni))) % R
    b_t, c_t = dot(bz), dot(cz)
    uvw = [[0] * ni for _ in range(3)]
    for i in range(ni):
        uvw[0][i] = lag[nc + i]
    for m, row, col, coeff in instance_terms:
        assert col < ni and row < nc
        uvw[m][col] = (uvw[m][col] + lag[row] * coeff) % R
    share = [sum(x * zi for x, zi in zip(uvw[m], z_instance)) % R for m in range(3)]
    dinv, ginv = pow(delta, -1, R), pow(gamma, -1, R)
    l_acc = (beta * (a_t - share[0]) + alpha * (b_t - share[1]) + (c_t - share[2])) * dinv % R
    h_acc, x = 0, zt * dinv % R
    for hi in h[:n - 1]:                                            # the prover's zip with h_query stops at n - 1
        h_acc = (h_acc + hi * x) % R
        x = x * t % R
    a = (alpha + a_t + r * delta) % R
    b = (beta + b_t + s * delta) % R
    c = (s * a + r * b - r * s * delta + l_acc + h_acc) % R
    gamma_abc = [(beta * uvw[0][i] + alpha * uvw[1][i] + uvw[2][i]) * ginv % R for i in range(ni)]
    return a, b, c, h_acc, gamma_abc


def verify_exponents(pk, public_inputs, proof):
    """verifier.rs verify_proof in the exponent: e(A, B) = e(alpha, beta) e(sum x_i gamma_abc_i, gamma) e(C, delta)."""
    a, b, c = proof[:3]
    tx = pk["toxic"]
    acc = sum(x * g for x, g in zip([1] + list(public_inputs), pk["gamma_abc"])) % R
    return (a * b - tx["alpha"] * tx["beta"] - acc * tx["gamma"] - c * tx["delta"]) % R == 0


# ---- G2: y^2 = x^3 + 4 (1 + u) over Fq2 = Fq[u] / (u^2 + 1) ----------------------------------------------------------------------
# Needed for B of a Groth16 proof (prover.rs: g2_b = s delta_g2 + b_g2_query[0] + MSM(b_g2_query[1..], assignment) + beta_g2).
# Fq2 elements are (c0, c1) tuples; the published generator (ark-bls12-381 g2.rs G2_GENERATOR_X / _Y; the IETF draft).
G2 = ((0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8,
       0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E),
      (0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801,
       0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE))


def f2_add(a, b):
    return (a[0] + b[0]) % Q, (a[1] + b[1]) % Q


def f2_sub(a, b):
    return (a[0] - b[0]) % Q, (a[1] - b[1]) % Q


def f2_mul(a, b):
    return (a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q


def f2_inv(a):
    n = pow(a[0] * a[0] + a[1] * a[1], -1, Q)
    return a[0] * n % Q, -a[1] * n % Q


def g2_on_curve(p):
    return p is None or f2_sub(f2_mul(p[1], p[1]), f2_add(f2_mul(f2_mul(p[0], p[0]), p[0]), (4, 4))) == (0, 0)


def g2_neg(p):
    return None if p is None else (p[0], ((-p[1][0]) % Q, (-p[1][1]) % Q))


def g2_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    if p[0] == q[0]:
        if f2_add(p[1], q[1]) == (0, 0):
            return None
        lam = f2_mul(f2_mul((3, 0), f2_mul(p[0], p[0])), f2_inv(f2_add(p[1], p[1])))
    else:
        lam = f2_mul(f2_sub(q[1], p[1]), f2_inv(f2_sub(q[0], p[0])))
    x = f2_sub(f2_sub(f2_mul(lam, lam), p[0]), q[0])
    return x, f2_sub(f2_mul(lam, f2_sub(p[0], x)), p[1])


def g2_mul(p, k):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, p)
        p = g2_add(p, p)
        k >>= 1
    return acc


def g2_to_limbs(p):
    """ark-ff bytes of a G2 affine point: x.c0, x.c1, y.c0, y.c1, each x R mod q as 6 little-endian u64 limbs; infinity = zeros."""
    if p is None:
        return [0] * 24
    return [(v * FQ_R % Q >> (64 * i)) & (2 ** 64 - 1) for c in p for v in c for i in range(6)]


def g2_from_limbs(limbs):
    limbs = [int(v) for v in limbs]
    if not any(limbs):
        return None
    inv = pow(FQ_R, -1, Q)
    v = [sum(l << (64 * i) for i, l in enumerate(limbs[6 * c:6 * c + 6])) * inv % Q for c in range(4)]
    return (v[0], v[1]), (v[2], v[3])


# ---- the pairing: e: G1 x G2 -> Fq12, for verify_proof (ark-groth16 0.3.0 verifier.rs) -------------------------------------------------
# Restated the plain way (no towers, no sparse lines, no cyclotomic tricks): Fq12 = Fq[w] / (w^12 - 2 w^6 + 2) -- with
# u = w^6 - 1 this contains Fq2 = Fq[u] / (u^2 + 1) and w^6 = 1 + u is the twist's non-residue --, the ate Miller loop over
# |z| = 0xd201000000010000 in affine coordinates on E(Fq12), and the final exponentiation as one power (q^12 - 1) / r.
# The sign of z is ignored (the loop runs over |z|): the map is then the inverse of the ate pairing, which is as bilinear and
# non-degenerate as the pairing itself, and every use here compares products of such values with each other.
# Pinned by tests/test_bls12_381.py: bilinearity e(a P, b Q) = e(P, Q)^(a b), e(P, Q)^r = 1, e(P, Q) != 1.
F12_MOD = (2, 0, 0, 0, 0, 0, -2, 0, 0, 0, 0, 0)          # w^12 = 2 w^6 - 2


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):                             # w^k = 2 w^(k-6) - 2 w^(k-12)
        c = t[k]
        if c:
            t[k - 6] += 2 * c
            t[k - 12] -= 2 * c
    return tuple(v % Q for v in t[:12])


def f12_add(a, b):
    return tuple((x + y) % Q for x, y in zip(a, b))


def f12_sub(a, b):
    return tuple((x - y) % Q for x, y in zip(a, b))


F12_ONE = (1,) + (0,) * 11


def f12_from_fq(x):
    return (x % Q,) + (0,) * 11


def f12_from_fq2(a):
    """a0 + a1 u with u = w^6 - 1."""
    return ((a[0] - a[1]) % Q, 0, 0, 0, 0, 0, a[1] % Q, 0, 0, 0, 0, 0)


def _poly_divmod(a, b):
    a = list(a)
    out = [0] * max(1, len(a) - len(b) + 1)
    inv = pow(b[-1], -1, Q)
    for i in range(len(a) - len(b), -1, -1):
        c = a[i + len(b) - 1] * inv % Q
        out[i] = c
        if c:
            for j, y in enumerate(b):
                a[i + j] = (a[i + j] - c * y) % Q
    while len(a) > 1 and a[-1] == 0:
        a.pop()
    return out, a


def f12_inv(a):
    """Extended Euclid in Fq[w] against the modulus."""
    mod = [2, 0, 0, 0, 0, 0, Q - 2, 0, 0, 0, 0, 0, 1]
    r0, r1 = mod, list(a)
    while len(r1) > 1 and r1[-1] == 0:
        r1.pop()
    s0, s1 = [0], [1]
    while any(r1) and len(r1) > 1:
        qq, rem = _poly_divmod(r0, r1)
        prod = [0] * (len(qq) + len(s1) - 1)
        for i, x in enumerate(qq):
            for j, y in enumerate(s1):
                prod[i + j] = (prod[i + j] + x * y) % Q
        s2 = [((s0[i] if i < len(s0) else 0) - (prod[i] if i < len(prod) else 0)) % Q for i in range(max(len(s0), len(prod)))]
        r0, r1, s0, s1 = r1, rem, s1, s2
    inv = pow(r1[0], -1, Q)
    return tuple((s1[i] * inv % Q if i < len(s1) else 0) for i in range(12))


def f12_pow(a, e):
    acc = F12_ONE
    while e:
        if e & 1:
            acc = f12_mul(acc, a)
        a = f12_mul(a, a)
        e >>= 1
    return acc


_W2_INV = f12_inv((0, 0, 1) + (0,) * 9)
_W3_INV = f12_inv((0, 0, 0, 1) + (0,) * 8)


def _untwist(q):
    """E'(Fq2) -> E(Fq12): (x, y) -> (x / w^2, y / w^3): then Y^2 - X^3 = (y^2 - x^3) / w^6 = 4 (1 + u) / (1 + u) = 4."""
    return f12_mul(f12_from_fq2(q[0]), _W2_INV), f12_mul(f12_from_fq2(q[1]), _W3_INV)


def _line(a, b, p):
    """The line through a and b (the tangent if a = b) on E(Fq12), evaluated at p; and a + b."""
    if a[0] != b[0]:
        lam = f12_mul(f12_sub(b[1], a[1]), f12_inv(f12_sub(b[0], a[0])))
    else:
        lam = f12_mul(f12_mul(f12_from_fq(3), f12_mul(a[0], a[0])), f12_inv(f12_add(a[1], a[1])))
    val = f12_sub(f12_sub(p[1], a[1]), f12_mul(lam, f12_sub(p[0], a[0])))
    x3 = f12_sub(f12_sub(f12_mul(lam, lam), a[0]), b[0])
    return val, (x3, f12_sub(f12_mul(lam, f12_sub(a[0], x3)), a[1]))


def miller_loop(p, q):
    """f_{|z|, Q}(P) for P in G1 and Q in G2 (affine tuples; either None -> 1)."""
    if p is None or q is None:
        return F12_ONE
    pp = (f12_from_fq(p[0]), f12_from_fq(p[1]))
    qq = _untwist(q)
    t, f = qq, F12_ONE
    for bit in bin(-Z_BLS)[3:]:
        val, t = _line(t, t, pp)
        f = f12_mul(f12_mul(f, f), val)
        if bit == "1":
            val, t = _line(t, qq, pp)
            f = f12_mul(f, val)
    return f


FINAL_EXPONENT = (Q ** 12 - 1) // R


def pairing(p, q):
    return f12_pow(miller_loop(p, q), FINAL_EXPONENT)


def verify_proof(vk, public_inputs, proof):
    """ark-groth16 0.3.0 verifier.rs verify_proof: e(A, B) = e(alpha_g1, beta_g2) e(sum x_i gamma_abc_g1[i], gamma_g2) e(C, delta_g2)
    (prepared as one product of Miller loops and one final exponentiation: e(A, B) e(-acc, gamma) e(-C, delta) == e(alpha, beta)).
    vk: dict of affine points alpha_g1, beta_g2, gamma_g2, delta_g2 and the list gamma_abc_g1; proof = (A, B, C) affine."""
    acc = vk["gamma_abc_g1"][0]
    for x, g in zip(public_inputs, vk["gamma_abc_g1"][1:]):
        acc = add(acc, mul(g, x))
    f = f12_mul(f12_mul(miller_loop(proof[0], proof[1]), miller_loop(neg(acc), vk["gamma_g2"])), miller_loop(neg(proof[2]), vk["delta_g2"]))
    return f12_pow(f, FINAL_EXPONENT) == pairing(vk["alpha_g1"], vk["beta_g2"])
