"""ORACLE (test infrastructure, never imported by the product): the R1CS -> QAP witness map of Groth16 over BLS12-381 Fr.

What the reference's prover does right after the hot path: `examples/pok_sig.rs:30-47` hands the circuit to
`Groth16::<Bls12_381>::prove`, whose first step after `generate_constraints` + `finalize()` is
`R1CStoQAP::witness_map` -- the coefficients of h(X) = (A(X) B(X) - C(X)) / (X^n - 1).

PARITY UNPINNED: ark-groth16 0.3.0 and ark-poly 0.3.0 are crates.io dependencies (`falcon-r1cs/Cargo.toml:14-19`),
absent from /root/reference, and there is no Rust toolchain here.  This file restates their published algorithm:

  ark-groth16 0.3.0  src/r1cs_to_qap.rs   R1CStoQAP::witness_map
      domain = D::new(num_constraints + num_inputs); a[i] = <A_i, z>, b[i] = <B_i, z> for i < num_constraints;
      a[num_constraints + j] = z[j] for j < num_inputs; ifft(a), ifft(b); coset_fft(a), coset_fft(b); ab = a o b;
      c[i] = <C_i, z>; ifft(c); coset_fft(c); ab -= c; divide_by_vanishing_poly_on_coset(ab); coset_ifft(ab); return ab
      with z = instance_assignment ++ witness_assignment (num_inputs = num_instance_variables, the constant one first)
  ark-poly 0.3.0     src/domain/radix2/mod.rs, src/domain/mod.rs
      size = next_power_of_two(num_coeffs); group_gen = F::get_root_of_unity(size);
      fft: evaluations at group_gen^i in natural order; ifft: its inverse (x size_inv);
      coset_fft = distribute_powers(coeffs, F::multiplicative_generator()) then fft;
      coset_ifft = ifft then distribute_powers(generator_inv);
      divide_by_vanishing_poly_on_coset: multiply by (g^size - 1)^-1
  ark-ff 0.3.0       FftField::get_root_of_unity: two_adic_root_of_unity squared (TWO_ADICITY - log2 size) times
  ark-bls12-381 0.3.0 src/fields/fr.rs    GENERATOR = 7, TWO_ADICITY = 32, TWO_ADIC_ROOT_OF_UNITY = 7^((p-1)/2^32)
      (Montgomery limbs 0xb9b58d8c5f0e466a, 0x5b1b4c801819d7ec, 0x0af53ae352a31e64, 0x5bf3adda19e9b27b --
      `check_constants` below confirms that this is what the formula gives)

All of these are exact maps over a field, so the result does not depend on the FFT schedule: h is the unique
polynomial of degree < n with h(g w^i) = (a b - c)(g w^i) / (g^n - 1).  Besides restating the steps, `check_identity`
verifies the defining property directly, without any FFT: A(tau) B(tau) - C(tau) = h(tau) (tau^n - 1) at a point tau
(true for every tau exactly when the witness satisfies the system), with A(tau) = sum_i a_i L_i(tau).
"""
P = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
GENERATOR = 7
TWO_ADICITY = 32
TWO_ADIC_ROOT_OF_UNITY = pow(GENERATOR, (P - 1) >> TWO_ADICITY, P)
R_MONT = (1 << 256) % P


def check_constants():
    """The Montgomery limbs ark-bls12-381 0.3.0 prints for TWO_ADIC_ROOT_OF_UNITY and GENERATOR."""
    limbs = lambda v: [(v * R_MONT % P >> (64 * i)) & (2**64 - 1) for i in range(4)]
    assert limbs(TWO_ADIC_ROOT_OF_UNITY) == [0xB9B58D8C5F0E466A, 0x5B1B4C801819D7EC, 0x0AF53AE352A31E64, 0x5BF3ADDA19E9B27B]
    assert limbs(GENERATOR) == [0x0000000EFFFFFFF1, 0x17E363D300189C0F, 0xFF9C57876F8457B0, 0x351332208FC5A8C4]
    assert pow(TWO_ADIC_ROOT_OF_UNITY, 1 << 32, P) == 1 and pow(TWO_ADIC_ROOT_OF_UNITY, 1 << 31, P) == P - 1


class Domain:
    """Radix2EvaluationDomain::new(num_coeffs) (ark-poly 0.3.0 domain/radix2/mod.rs)."""

    def __init__(self, num_coeffs):
        size = 1
        while size < num_coeffs:
            size <<= 1
        self.size = size
        self.log_size = size.bit_length() - 1
        if self.log_size > TWO_ADICITY:
            raise ValueError("PolynomialDegreeTooLarge")
        g = TWO_ADIC_ROOT_OF_UNITY
        for _ in range(self.log_size, TWO_ADICITY):           # FftField::get_root_of_unity
            g = g * g % P
        self.group_gen = g
        self.group_gen_inv = pow(g, P - 2, P)
        self.size_inv = pow(size, P - 2, P)
        self.generator_inv = pow(GENERATOR, P - 2, P)

    def _transform(self, vals, root):
        n, a = self.size, list(vals) + [0] * (self.size - len(vals))
        j = 0
        for i in range(1, n):                                 # bit reversal, then decimation in time
            bit = n >> 1
            while j & bit:
                j ^= bit
                bit >>= 1
            j |= bit
            if i < j:
                a[i], a[j] = a[j], a[i]
        m = 2
        while m <= n:
            wm, half = pow(root, n // m, P), m // 2
            tw = [1] * half
            for k in range(1, half):
                tw[k] = tw[k - 1] * wm % P
            for k in range(0, n, m):
                for jj in range(half):
                    t = tw[jj] * a[k + jj + half] % P
                    u = a[k + jj]
                    a[k + jj] = (u + t) % P
                    a[k + jj + half] = (u - t) % P
            m <<= 1
        return a

    def fft(self, coeffs):
        return self._transform(coeffs, self.group_gen)

    def ifft(self, evals):
        return [v * self.size_inv % P for v in self._transform(evals, self.group_gen_inv)]

    @staticmethod
    def distribute_powers(coeffs, g):
        out, p = [], 1
        for c in coeffs:
            out.append(c * p % P)
            p = p * g % P
        return out

    def coset_fft(self, coeffs):
        return self.fft(self.distribute_powers(coeffs, GENERATOR))

    def coset_ifft(self, evals):
        return self.distribute_powers(self.ifft(evals), self.generator_inv)

    def divide_by_vanishing_poly_on_coset(self, evals):
        i = pow((pow(GENERATOR, self.size, P) - 1) % P, P - 2, P)
        return [v * i % P for v in evals]


def evaluate_constraint(row, z):
    """ark-groth16 r1cs_to_qap.rs evaluate_constraint: sum coeff * z[index]."""
    return sum(c * z[i] for c, i in row) % P


def matvec(mats, z):
    """(A z, B z, C z) for matrices given as lists of rows of (coeff, column)."""
    return tuple([evaluate_constraint(r, z) for r in m] for m in mats)


def witness_map_from_products(az, bz, cz, num_inputs, z_inputs):
    """The witness map from the three matrix-vector products on (everything after evaluate_constraint)."""
    nc = len(az)
    d = Domain(nc + num_inputs)
    a = list(az) + [0] * (d.size - nc)
    b = list(bz) + [0] * (d.size - nc)
    a[nc:nc + num_inputs] = [v % P for v in z_inputs[:num_inputs]]
    a = d.coset_fft(d.ifft(a))
    b = d.coset_fft(d.ifft(b))
    ab = [x * y % P for x, y in zip(a, b)]
    c = d.coset_fft(d.ifft(list(cz) + [0] * (d.size - nc)))
    ab = [(x - y) % P for x, y in zip(ab, c)]
    ab = d.divide_by_vanishing_poly_on_coset(ab)
    return d.coset_ifft(ab)


def witness_map(mats, num_inputs, z):
    az, bz, cz = matvec(mats, z)
    return witness_map_from_products(az, bz, cz, num_inputs, z)


def lagrange_at(d, tau):
    """L_i(tau) for the domain's points w^i, i < size (tau outside the domain): (tau^n - 1) w^i / (n (tau - w^i))."""
    n = d.size
    zt = (pow(tau, n, P) - 1) % P
    w, denom = 1, []
    for _ in range(n):
        denom.append((tau - w) % P)
        w = w * d.group_gen % P
    pref, acc = [], 1                                         # batch inversion
    for v in denom:
        pref.append(acc)
        acc = acc * v % P
    inv = pow(acc, P - 2, P)
    out = [0] * n
    for i in range(n - 1, -1, -1):
        out[i] = inv * pref[i] % P
        inv = inv * denom[i] % P
    w, scale = 1, zt * d.size_inv % P
    for i in range(n):
        out[i] = out[i] * w % P * scale % P
        w = w * d.group_gen % P
    return out, zt


def check_identity(az, bz, cz, num_inputs, z_inputs, h, tau):
    """A(tau) B(tau) - C(tau) == h(tau) Z(tau); returns (lhs, rhs).  FFT-free."""
    nc = len(az)
    d = Domain(nc + num_inputs)
    assert len(h) == d.size
    lag, zt = lagrange_at(d, tau)
    a = list(az) + [v % P for v in z_inputs[:num_inputs]]
    at = sum(x * l for x, l in zip(a, lag)) % P
    bt = sum(x * l for x, l in zip(bz, lag)) % P
    ct = sum(x * l for x, l in zip(cz, lag)) % P
    ht, p = 0, 1
    for c in h:
        ht = (ht + c * p) % P
        p = p * tau % P
    return (at * bt - ct) % P, ht * zt % P


def product_high_half(az, bz, num_inputs, z_inputs):
    """hi(X) of a(X) b(X) = lo(X) + X^n hi(X), a and b the interpolants of (A z ++ inputs) and B z on the domain.
    For a witness that satisfies the system this IS witness_map's output: c(X) = (a b) mod (X^n - 1) = lo + hi, so
    a b - c = (X^n - 1) hi.  (For one that does not, witness_map returns the interpolant of (a b - c) / Z on the coset,
    which is not a quotient of anything; hi is still well defined.)  Schoolbook product: small systems only.
    The device computes it with six transforms instead of seven: S = (a b) mod (X^n - 1) = lo + hi from the pointwise
    products on the domain, N = (a b) mod (X^n + 1) = lo - hi from the pointwise products on the coset psi H with
    psi^n = -1, hi = (S - N) / 2."""
    nc = len(az)
    d = Domain(nc + num_inputs)
    n = d.size
    a = list(az) + [0] * (n - nc)
    b = list(bz) + [0] * (n - nc)
    a[nc:nc + num_inputs] = [v % P for v in z_inputs[:num_inputs]]
    ac, bc = d.ifft(a), d.ifft(b)
    prod = [0] * (2 * n)
    for i, x in enumerate(ac):
        if x:
            for j, y in enumerate(bc):
                prod[i + j] = (prod[i + j] + x * y) % P
    return prod[n:2 * n]


def product_high_half_six_transforms(az, bz, num_inputs, z_inputs):
    """The same through the identity the device uses (any size)."""
    nc = len(az)
    d = Domain(nc + num_inputs)
    n = d.size
    a = list(az) + [0] * (n - nc)
    b = list(bz) + [0] * (n - nc)
    a[nc:nc + num_inputs] = [v % P for v in z_inputs[:num_inputs]]
    psi = TWO_ADIC_ROOT_OF_UNITY
    for _ in range(d.log_size + 1, TWO_ADICITY):
        psi = psi * psi % P
    assert psi * psi % P == d.group_gen and pow(psi, n, P) == P - 1
    s = d.ifft([x * y % P for x, y in zip(a, b)])                                   # lo + hi
    ea = d.fft(d.distribute_powers(d.ifft(a), psi))
    eb = d.fft(d.distribute_powers(d.ifft(b), psi))
    neg = d.distribute_powers(d.ifft([x * y % P for x, y in zip(ea, eb)]), pow(psi, P - 2, P))   # lo - hi
    half = pow(2, P - 2, P)
    return [(x - y) * half % P for x, y in zip(s, neg)]
