"""ORACLE (test infrastructure only) -- Falcon key generation and signing, so that a GENUINE signature can go through
the engine end to end, as the reference's own end-to-end test does with falcon-rust:
``KeyPair::keygen -> sign_with_seed("test seed", "testing message") -> verify -> generate_constraints -> is_satisfied``
(/root/reference/falcon-r1cs/src/circuits/falcon_ntt.rs:133-160, examples/constraint_counts.rs:49-72).

falcon-rust (a wrapper of the Falcon round-3 C code) is NOT under /root/reference and cannot be built here, and the
reference holds no (pk, msg, sig) vector.  This file restates the Falcon specification v1.2: NTRUGen with the
Gram-Schmidt norm test (Alg. 5), NTRUSolve by the field-norm tower with Babai reduction (Alg. 6, 7), the LDL* tree
(Alg. 8, 9), fast Fourier sampling (Alg. 11), Sign (Alg. 10) and Verify (Alg. 16); hashing and the encodings are
oracle/falcon_codec.py.  One deliberate simplification: the integer Gaussian sampler is a plain rejection sampler on
floating point, not the specification's constant-time RCDT/BerExp construction -- the distribution is the same
discrete Gaussian, and a signature is genuine iff Verify accepts it, which tests/test_oracle.py checks for every
fixture (any Falcon verifier accepts these keys and signatures; byte-identity with falcon-rust's output for the same
seed is neither claimed nor needed).  PARITY UNPINNED against falcon-rust, like falcon_codec.py.

It is run once, by tests/golden/make_signed.py, to produce the committed fixtures tests/golden/falcon_signed.json;
key generation for Falcon-1024 takes about a minute of Python big-integer arithmetic.
"""
import cmath
import hashlib
import math

import numpy as np

from .falcon_codec import NONCE_LEN, Q, SIG_LEN, comp_decode, comp_encode, hash_to_point, modq_decode, modq_encode

PARAMS = {
    9: {"sigma": 165.7366171829776, "sigmin": 1.2778336969128337, "beta2": 34034726},
    10: {"sigma": 168.38857144654395, "sigmin": 1.298280334344292, "beta2": 70265242},
}


# ---------------------------------------------------------------------------------------------------------------------
# randomness: SHAKE256 in counter mode (deterministic per seed), and the discrete Gaussian over the integers
# ---------------------------------------------------------------------------------------------------------------------
class Rng:
    def __init__(self, seed: bytes):
        self.seed, self.ctr, self.buf = bytes(seed), 0, b""

    def bytes(self, k):
        while len(self.buf) < k:
            self.buf += hashlib.shake_256(self.seed + self.ctr.to_bytes(8, "little")).digest(4096)
            self.ctr += 1
        out, self.buf = self.buf[:k], self.buf[k:]
        return out

    def uniform(self):                       # [0, 1) with 53 random bits
        return (int.from_bytes(self.bytes(7), "little") >> 3) / float(1 << 53)

    def below(self, k):                      # uniform integer in [0, k)
        bits = max(1, (k - 1).bit_length())
        while True:
            x = int.from_bytes(self.bytes((bits + 7) // 8), "little") & ((1 << bits) - 1)
            if x < k:
                return x


def sample_z(mu, sigma, rng):
    """z ~ D_{Z, mu, sigma}: uniform proposal on round(mu) +- ceil(10 sigma), accepted with the Gaussian weight."""
    c = int(round(mu))
    k = int(math.ceil(10.0 * sigma))
    while True:
        z = c - k + rng.below(2 * k + 1)
        if rng.uniform() < math.exp(-((z - mu) ** 2) / (2.0 * sigma * sigma)):
            return z


# ---------------------------------------------------------------------------------------------------------------------
# Z[x] / (x^n + 1) with Python integers (Kronecker substitution: one big multiplication per product)
# ---------------------------------------------------------------------------------------------------------------------
def pmul(a, b):
    n = len(a)
    if n == 1:
        return [a[0] * b[0]]
    w = max(abs(x) for x in a).bit_length() + max(abs(x) for x in b).bit_length() + n.bit_length() + 2
    w = (w + 7) // 8 * 8
    pack = lambda p: sum(x << (w * i) for i, x in enumerate(p))
    prod = pack(a) * pack(b)
    neg = prod < 0
    if neg:
        prod = -prod
    wb = w // 8
    raw = prod.to_bytes(wb * 2 * n + 8, "little")
    half, full = 1 << (w - 1), 1 << w
    coef, carry = [], 0
    for i in range(2 * n):
        r = int.from_bytes(raw[i * wb:(i + 1) * wb], "little") + carry
        carry = 0
        if r >= half:
            r -= full
            carry = 1
        coef.append(-r if neg else r)
    return [coef[i] - coef[i + n] for i in range(n)]


def field_norm(a):
    """N(a)(y) = a_e(y)^2 - y a_o(y)^2 for a(x) = a_e(x^2) + x a_o(x^2)   (spec 3.6)"""
    ne = len(a) // 2
    e2, o2 = pmul(a[0::2], a[0::2]), pmul(a[1::2], a[1::2])
    res = list(e2)
    for i in range(ne - 1):
        res[i + 1] -= o2[i]
    res[0] += o2[ne - 1]
    return res


def lift(a):
    res = [0] * (2 * len(a))
    res[0::2] = a
    return res


def galois_conjugate(a):
    return [x if i % 2 == 0 else -x for i, x in enumerate(a)]


def xgcd(a, b):
    x0, x1, y0, y1 = 1, 0, 0, 1
    while b:
        qq, a, b = a // b, b, a % b
        x0, x1 = x1, x0 - qq * x1
        y0, y1 = y1, y0 - qq * y1
    return a, x0, y0


# FFT over the roots of x^n + 1 for pointwise work (numpy's FFT with a twist): a(zeta_k), zeta_k = exp(i pi (2k+1) / n)
def fft_eval(a):
    n = len(a)
    tw = np.exp(1j * np.pi * np.arange(n) / n)
    return np.fft.ifft(np.asarray(a, dtype=np.float64) * tw) * n


def fft_interp(A):
    n = len(A)
    tw = np.exp(-1j * np.pi * np.arange(n) / n)
    return (np.fft.fft(A) / n * tw).real


def reduce_fg(f, g, F, G):
    """Babai reduction of (F, G) against (f, g) (spec Alg. 7): subtract k (f, g), k = round((F f* + G g*) / (f f* + g g*)),
    working on the top 53 bits of the coefficients until nothing changes."""
    bits = lambda p: max(x.bit_length() for x in p)
    size = max(53, bits(f), bits(g))
    fa = fft_eval([x >> (size - 53) for x in f])
    ga = fft_eval([x >> (size - 53) for x in g])
    den = fa * np.conj(fa) + ga * np.conj(ga)
    while True:
        Size = max(53, bits(F), bits(G))
        if Size < size:
            break
        Fa = fft_eval([x >> (Size - 53) for x in F])
        Ga = fft_eval([x >> (Size - 53) for x in G])
        k = np.rint(fft_interp((Fa * np.conj(fa) + Ga * np.conj(ga)) / den)).astype(np.int64)
        if not k.any():
            break
        kl = [int(x) for x in k]
        fk, gk = pmul(f, kl), pmul(g, kl)
        sh = Size - size
        F = [x - (y << sh) for x, y in zip(F, fk)]
        G = [x - (y << sh) for x, y in zip(G, gk)]
    return F, G


def ntru_solve(f, g):
    """F, G with f G - g F = q mod (x^n + 1)   (spec Alg. 6)"""
    n = len(f)
    if n == 1:
        d, u, v = xgcd(f[0], g[0])
        if d != 1:
            raise ValueError("gcd(N(f), N(g)) != 1")
        return [-Q * v], [Q * u]
    Fp, Gp = ntru_solve(field_norm(f), field_norm(g))
    F = pmul(lift(Fp), galois_conjugate(g))
    G = pmul(lift(Gp), galois_conjugate(f))
    return reduce_fg(f, g, F, G)


# ---------------------------------------------------------------------------------------------------------------------
# arithmetic mod q (key generation: h = g f^-1), by the Vandermonde matrix of the 2n-th roots -- O(n^2), run once
# ---------------------------------------------------------------------------------------------------------------------
def _vandermonde(n):
    psi = pow(7, 2048 // (2 * n), Q)                   # 7 has order 2048 mod q
    e = (np.outer(2 * np.arange(n) + 1, np.arange(n)) % (2 * n)).astype(np.int64)
    pw = np.array([pow(psi, i, Q) for i in range(2 * n)], dtype=np.int64)
    return pw[e], pw[(-e) % (2 * n)]


def public_key(f, g):
    n = len(f)
    V, Vinv = _vandermonde(n)
    fn = V.dot(np.array(f, dtype=np.int64) % Q) % Q
    gn = V.dot(np.array(g, dtype=np.int64) % Q) % Q
    if not fn.all():
        return None
    hn = gn * np.array([pow(int(x), Q - 2, Q) for x in fn], dtype=np.int64) % Q
    return [int(x) for x in (Vinv.T.dot(hn) % Q) * pow(n, Q - 2, Q) % Q]


def mul_mod_q(a, b):
    n = len(a)
    full = np.convolve(np.array(a, dtype=np.int64), np.array(b, dtype=np.int64))
    res = full[:n].copy()
    res[: n - 1] -= full[n:]
    return [int(x) for x in res % Q]


# ---------------------------------------------------------------------------------------------------------------------
# FFT in Falcon's tree order (split / merge, spec 3.7), the LDL* tree and fast Fourier sampling
# ---------------------------------------------------------------------------------------------------------------------
_ROOTS = {1: np.array([-1.0 + 0j])}


def roots(n):
    """roots of x^n + 1, ordered so that roots(2m)[2i], roots(2m)[2i+1] = +-sqrt(roots(m)[i])"""
    if n not in _ROOTS:
        r = roots(n // 2)
        s = np.array([cmath.sqrt(x) for x in r])
        out = np.empty(n, dtype=np.complex128)
        out[0::2], out[1::2] = s, -s
        _ROOTS[n] = out
    return _ROOTS[n]


def merge_fft(f0, f1):
    n = 2 * len(f0)
    s = roots(n)[0::2]
    out = np.empty(n, dtype=np.complex128)
    out[0::2], out[1::2] = f0 + s * f1, f0 - s * f1
    return out


def split_fft(F):
    s = roots(len(F))[0::2]
    return (F[0::2] + F[1::2]) / 2, (F[0::2] - F[1::2]) / (2 * s)


def fft(a):
    a = np.asarray(a, dtype=np.complex128)
    if len(a) == 1:
        return a.copy()
    return merge_fft(fft(a[0::2]), fft(a[1::2]))


def ifft(F):
    if len(F) == 1:
        return F.real.copy()
    f0, f1 = split_fft(F)
    out = np.empty(len(F), dtype=np.float64)
    out[0::2], out[1::2] = ifft(f0), ifft(f1)
    return out


def ffldl(g00, g01, g11):
    """LDL* tree of the self-adjoint 2x2 Gram matrix [[g00, g01], [adj(g01), g11]] in FFT form (spec Alg. 9)"""
    n = len(g00)
    l10 = np.conj(g01) / g00
    d11 = g11 - l10 * np.conj(l10) * g00
    if n == 1:
        return [l10, g00.real.copy(), d11.real.copy()]
    a0, a1 = split_fft(g00)
    b0, b1 = split_fft(d11)
    return [l10, ffldl(a0, a1, a0), ffldl(b0, b1, b0)]


def normalize_tree(tree, sigma):
    if isinstance(tree[1], np.ndarray):              # leaves: d00, d11 of a 1 x 1 block
        tree[1] = sigma / math.sqrt(float(tree[1][0]))
        tree[2] = sigma / math.sqrt(float(tree[2][0]))
    else:
        normalize_tree(tree[1], sigma)
        normalize_tree(tree[2], sigma)


def ffsampling(t0, t1, tree, sigmin, rng):
    """z = (z0, z1) in FFT form, close to t with covariance shaped by the tree (spec Alg. 11)"""
    n = len(t0)
    l10 = tree[0]
    if n == 1:
        z1 = sample_z(float(t1[0].real), tree[2], rng)
        t0b = t0[0] + (t1[0] - z1) * l10[0]
        z0 = sample_z(float(t0b.real), tree[1], rng)
        return np.array([z0 + 0j]), np.array([z1 + 0j])
    z1 = merge_fft(*ffsampling(*split_fft(t1), tree[2], sigmin, rng))
    t0b = t0 + (t1 - z1) * l10
    z0 = merge_fft(*ffsampling(*split_fft(t0b), tree[1], sigmin, rng))
    return z0, z1


# ---------------------------------------------------------------------------------------------------------------------
# keygen / sign / verify
# ---------------------------------------------------------------------------------------------------------------------
class SecretKey:
    def __init__(self, logn, f, g, F, G, h):
        self.logn, self.f, self.g, self.F, self.G, self.h = logn, f, g, F, G, h
        n = 1 << logn
        assert [x - y for x, y in zip(pmul(f, G), pmul(g, F))] == [Q] + [0] * (n - 1)      # the NTRU equation
        self.b00, self.b01 = fft(g), fft([-x for x in f])
        self.b10, self.b11 = fft(G), fft([-x for x in F])
        g00 = self.b00 * np.conj(self.b00) + self.b01 * np.conj(self.b01)
        g01 = self.b00 * np.conj(self.b10) + self.b01 * np.conj(self.b11)
        g11 = self.b10 * np.conj(self.b10) + self.b11 * np.conj(self.b11)
        self.tree = ffldl(g00, g01, g11)
        normalize_tree(self.tree, PARAMS[logn]["sigma"])

    def public_key_bytes(self):
        return modq_encode(self.h, self.logn)


def keygen(logn, seed: bytes):
    """NTRUGen (spec Alg. 5): f, g Gaussian with sigma = 1.17 sqrt(q / 2n), Gram-Schmidt norm <= 1.17^2 q, f invertible
    mod q, then NTRUSolve."""
    n = 1 << logn
    rng = Rng(b"falcon-oracle-keygen" + bytes([logn]) + seed)
    sigma_fg = 1.17 * math.sqrt(Q / (2.0 * n))
    while True:
        f = [sample_z(0.0, sigma_fg, rng) for _ in range(n)]
        g = [sample_z(0.0, sigma_fg, rng) for _ in range(n)]
        fa, ga = fft_eval(f), fft_eval(g)
        den = fa * np.conj(fa) + ga * np.conj(ga)
        n1 = sum(x * x for x in f) + sum(x * x for x in g)
        n2 = float(np.sum(np.abs(Q * np.conj(fa) / den) ** 2 + np.abs(Q * np.conj(ga) / den) ** 2).real) / n
        if max(n1, n2) > 1.17 ** 2 * Q:
            continue
        h = public_key(f, g)
        if h is None:
            continue
        try:
            F, G = ntru_solve(f, g)
        except ValueError:
            continue
        if max(abs(x) for x in F + G) > 127:              # the private-key encoding's range (spec 3.11.5)
            continue
        return SecretKey(logn, f, g, F, G, h)


def sign(sk: SecretKey, msg: bytes, seed: bytes):
    """Sign (spec Alg. 10) -> padded signature bytes: header, 40-byte nonce, compressed s2."""
    logn, n = sk.logn, 1 << sk.logn
    P = PARAMS[logn]
    rng = Rng(b"falcon-oracle-sign" + seed)
    while True:
        nonce = rng.bytes(NONCE_LEN)
        c = fft(hash_to_point(nonce, msg, logn))
        t0, t1 = c * sk.b11 / Q, -c * sk.b01 / Q          # (c, 0) B^-1
        for _ in range(16):
            z0, z1 = ffsampling(t0, t1, sk.tree, P["sigmin"], rng)
            d0, d1 = t0 - z0, t1 - z1
            s1 = np.rint(ifft(d0 * sk.b00 + d1 * sk.b10)).astype(np.int64)
            s2 = np.rint(ifft(d0 * sk.b01 + d1 * sk.b11)).astype(np.int64)
            if int((s1 * s1).sum() + (s2 * s2).sum()) > P["beta2"] or np.abs(s2).max() > 2047:
                continue
            try:
                return comp_encode([int(x) for x in s2], logn, nonce, SIG_LEN[logn])
            except ValueError:
                continue


def verify(pk_bytes: bytes, msg: bytes, sig_bytes: bytes, logn: int) -> bool:
    """Verify (spec Alg. 16): s1 = c - s2 h mod q, accept iff ||(s1, s2)||^2 <= beta^2 (centred representatives)."""
    h = modq_decode(pk_bytes, logn)
    dec = comp_decode(sig_bytes, logn)
    if h is None or dec is None:
        return False
    nonce, s2 = dec
    c = hash_to_point(nonce, msg, logn)
    s1 = [(ci - x) % Q for ci, x in zip(c, mul_mod_q(s2, h))]
    centre = lambda x: x if x <= Q // 2 else x - Q
    return sum(centre(x) ** 2 for x in s1) + sum(centre(x) ** 2 for x in s2) <= PARAMS[logn]["beta2"]
