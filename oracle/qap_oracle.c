/* ORACLE (test infrastructure, never linked into the product): the R1CS -> QAP witness map of Groth16 over
 * BLS12-381 Fr in plain C, following oracle/qap.py step for step (see that file for the ark-groth16 0.3.0 /
 * ark-poly 0.3.0 citations and the "parity unpinned" note).  Fast enough for the full circuits (domain 2^17 / 2^18:
 * about a second); checked against qap.py on small systems and through the FFT-free identity
 * A(tau) B(tau) - C(tau) = h(tau) (tau^n - 1) on the real ones (tests/test_qap.py).
 *
 * Interface: canonical little-endian 4 x u64 everywhere (values < p).
 *   frw_oracle_qap_domain_log(num_constraints, num_inputs)            -> log2 of the domain size
 *   frw_oracle_qap_matvec(...)                                        -> A z, B z, C z
 *   frw_oracle_qap_witness_map(az, bz, cz, nc, num_inputs, z, h)      -> h[domain size]                          */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fe;

static const uint64_t QP[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const uint64_t QR1[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
static const uint64_t QR2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
#define QINV 0xfffffffeffffffffULL

static int ge_p(const uint64_t a[4])
{
    for (int i = 3; i >= 0; i--) {
        if (a[i] > QP[i]) return 1;
        if (a[i] < QP[i]) return 0;
    }
    return 1;
}
static void sub_p(uint64_t a[4])
{
    u128 b = 0;
    for (int i = 0; i < 4; i++) { u128 x = (u128)a[i] - QP[i] - (uint64_t)b; a[i] = (uint64_t)x; b = (x >> 64) & 1; }
}
static fe f_add(fe a, fe b)
{
    fe r; u128 c = 0;
    for (int i = 0; i < 4; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (ge_p(r.l)) sub_p(r.l);
    return r;
}
static fe f_sub(fe a, fe b)
{
    fe r; u128 bw = 0;
    for (int i = 0; i < 4; i++) { u128 x = (u128)a.l[i] - b.l[i] - (uint64_t)bw; r.l[i] = (uint64_t)x; bw = (x >> 64) & 1; }
    if (bw) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + QP[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
    return r;
}
/* Montgomery product a b / 2^256 mod p */
static fe f_mul(fe a, fe b)
{
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a.l[j] * b.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * QINV;
        c = (u128)m * QP[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * QP[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    fe r = {{t[0], t[1], t[2], t[3]}};
    if (t[4] || ge_p(r.l)) sub_p(r.l);
    return r;
}
static fe f_from(const uint64_t c[4]) { fe a, r2; memcpy(a.l, c, 32); memcpy(r2.l, QR2, 32); return f_mul(a, r2); }
static void f_to(fe a, uint64_t out[4]) { fe o = {{1, 0, 0, 0}}; fe r = f_mul(a, o); memcpy(out, r.l, 32); }
static fe f_one(void) { fe r; memcpy(r.l, QR1, 32); return r; }
static fe f_small(uint64_t x) { uint64_t c[4] = {x, 0, 0, 0}; return f_from(c); }
static fe f_pow(fe b, const uint64_t e[4])
{
    fe r = f_one();
    for (int i = 255; i >= 0; i--) {
        r = f_mul(r, r);
        if ((e[i / 64] >> (i % 64)) & 1) r = f_mul(r, b);
    }
    return r;
}
static fe f_pow64(fe b, uint64_t e) { uint64_t x[4] = {e, 0, 0, 0}; return f_pow(b, x); }
static fe f_inv(fe a)
{
    uint64_t e[4]; memcpy(e, QP, 32); e[0] -= 2;           /* p - 2 (no borrow: low limb ends in ...0001) */
    return f_pow(a, e);
}

int frw_oracle_qap_domain_log(uint64_t num_constraints, uint64_t num_inputs)
{
    uint64_t size = 1; int lg = 0;
    while (size < num_constraints + num_inputs) { size <<= 1; lg++; }
    return lg;
}

/* two_adic_root_of_unity = 7^((p-1)/2^32), squared (32 - lg) times (ark-ff get_root_of_unity) */
static fe root_of_unity(int lg)
{
    uint64_t e[4] = {(QP[0] - 1) >> 32 | QP[1] << 32, QP[1] >> 32 | QP[2] << 32, QP[2] >> 32 | QP[3] << 32, QP[3] >> 32};
    fe g = f_pow(f_small(7), e);
    for (int i = lg; i < 32; i++) g = f_mul(g, g);
    return g;
}

/* in-place transform of a[n] with the given primitive n-th root: a'[k] = sum_j a[j] root^(jk), natural order both sides */
static void transform(fe *a, int lg, fe root)
{
    const size_t n = (size_t)1 << lg;
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j |= bit;
        if (i < j) { fe t = a[i]; a[i] = a[j]; a[j] = t; }
    }
    fe *tw = (fe *)malloc(sizeof(fe) * (n / 2 ? n / 2 : 1));
    for (int s = 1; s <= lg; s++) {
        const size_t m = (size_t)1 << s, half = m / 2;
        fe wm = f_pow64(root, n / m);
        tw[0] = f_one();
        for (size_t k = 1; k < half; k++) tw[k] = f_mul(tw[k - 1], wm);
        for (size_t k = 0; k < n; k += m)
            for (size_t j = 0; j < half; j++) {
                fe t = f_mul(tw[j], a[k + j + half]), u = a[k + j];
                a[k + j] = f_add(u, t);
                a[k + j + half] = f_sub(u, t);
            }
    }
    free(tw);
}
static void distribute_powers(fe *a, size_t n, fe g)
{
    fe p = f_one();
    for (size_t i = 0; i < n; i++) { a[i] = f_mul(a[i], p); p = f_mul(p, g); }
}
static void scale(fe *a, size_t n, fe s) { for (size_t i = 0; i < n; i++) a[i] = f_mul(a[i], s); }

typedef struct { int lg; size_t n; fe w, winv, ninv, g, ginv; } dom_t;
static dom_t make_domain(int lg)
{
    dom_t d; d.lg = lg; d.n = (size_t)1 << lg;
    d.w = root_of_unity(lg); d.winv = f_inv(d.w); d.ninv = f_inv(f_small(d.n)); d.g = f_small(7); d.ginv = f_inv(d.g);
    return d;
}
static void ifft(const dom_t *d, fe *a) { transform(a, d->lg, d->winv); scale(a, d->n, d->ninv); }
static void coset_fft(const dom_t *d, fe *a) { distribute_powers(a, d->n, d->g); transform(a, d->lg, d->w); }
static void coset_ifft(const dom_t *d, fe *a) { ifft(d, a); distribute_powers(a, d->n, d->ginv); }

/* rows in CSR: ptr[nc+1], col[nnz], val[nnz][4] canonical; z[(I+W)][4] canonical; out[nc][4] canonical */
void frw_oracle_qap_matvec(uint64_t nc, const uint64_t *ptr, const uint32_t *col, const uint64_t *val, const uint64_t *z,
                           uint64_t num_vars, uint64_t *out)
{
    fe *zm = (fe *)malloc(sizeof(fe) * num_vars);
    for (uint64_t i = 0; i < num_vars; i++) zm[i] = f_from(z + 4 * i);
    for (uint64_t r = 0; r < nc; r++) {
        fe acc = {{0, 0, 0, 0}};
        for (uint64_t k = ptr[r]; k < ptr[r + 1]; k++) acc = f_add(acc, f_mul(f_from(val + 4 * k), zm[col[k]]));
        f_to(acc, out + 4 * r);
    }
    free(zm);
}

/* ark-groth16 0.3.0 R1CStoQAP::witness_map after the evaluate_constraint loops */
int frw_oracle_qap_witness_map(const uint64_t *az, const uint64_t *bz, const uint64_t *cz, uint64_t nc, uint64_t num_inputs,
                               const uint64_t *z /* at least num_inputs x 4 */, uint64_t *h /* domain size x 4 */)
{
    const int lg = frw_oracle_qap_domain_log(nc, num_inputs);
    if (lg > 32) return -1;                                  /* PolynomialDegreeTooLarge */
    const dom_t d = make_domain(lg);
    fe *a = (fe *)calloc(d.n, sizeof(fe)), *b = (fe *)calloc(d.n, sizeof(fe)), *c = (fe *)calloc(d.n, sizeof(fe));
    if (!a || !b || !c) { free(a); free(b); free(c); return -2; }
    for (uint64_t i = 0; i < nc; i++) { a[i] = f_from(az + 4 * i); b[i] = f_from(bz + 4 * i); c[i] = f_from(cz + 4 * i); }
    for (uint64_t j = 0; j < num_inputs; j++) a[nc + j] = f_from(z + 4 * j);
    ifft(&d, a); ifft(&d, b);
    coset_fft(&d, a); coset_fft(&d, b);
    for (size_t i = 0; i < d.n; i++) a[i] = f_mul(a[i], b[i]);
    ifft(&d, c); coset_fft(&d, c);
    for (size_t i = 0; i < d.n; i++) a[i] = f_sub(a[i], c[i]);
    {
        fe one = f_one();
        fe zc = f_inv(f_sub(f_pow64(d.g, d.n), one));         /* divide_by_vanishing_poly_on_coset */
        scale(a, d.n, zc);
    }
    coset_ifft(&d, a);
    for (size_t i = 0; i < d.n; i++) f_to(a[i], h + 4 * i);
    free(a); free(b); free(c);
    return 0;
}

/* hi(X) of a(X) b(X) = lo + X^n hi through the six-transform identity of oracle/qap.py::product_high_half_six_transforms:
 * S = (a b) mod (X^n - 1) from the pointwise products on the domain, N = (a b) mod (X^n + 1) from those on the coset
 * psi H (psi^2 = w), hi = (S - N) / 2.  Equals the witness map's h exactly when the system is satisfied. */
int frw_oracle_qap_product_high_half(const uint64_t *az, const uint64_t *bz, uint64_t nc, uint64_t num_inputs,
                                     const uint64_t *z, uint64_t *h)
{
    const int lg = frw_oracle_qap_domain_log(nc, num_inputs);
    if (lg > 31) return -1;
    const dom_t d = make_domain(lg);
    fe *a = (fe *)calloc(d.n, sizeof(fe)), *b = (fe *)calloc(d.n, sizeof(fe)), *s = (fe *)calloc(d.n, sizeof(fe));
    if (!a || !b || !s) { free(a); free(b); free(s); return -2; }
    for (uint64_t i = 0; i < nc; i++) { a[i] = f_from(az + 4 * i); b[i] = f_from(bz + 4 * i); }
    for (uint64_t j = 0; j < num_inputs; j++) a[nc + j] = f_from(z + 4 * j);
    for (size_t i = 0; i < d.n; i++) s[i] = f_mul(a[i], b[i]);
    ifft(&d, s);                                               /* lo + hi */
    const fe psi = root_of_unity(lg + 1), psi_inv = f_inv(psi);
    ifft(&d, a); ifft(&d, b);
    distribute_powers(a, d.n, psi); distribute_powers(b, d.n, psi);
    transform(a, d.lg, d.w); transform(b, d.lg, d.w);
    for (size_t i = 0; i < d.n; i++) a[i] = f_mul(a[i], b[i]);
    ifft(&d, a);
    distribute_powers(a, d.n, psi_inv);                        /* lo - hi */
    const fe half = f_inv(f_small(2));
    for (size_t i = 0; i < d.n; i++) f_to(f_mul(f_sub(s[i], a[i]), half), h + 4 * i);
    free(a); free(b); free(s);
    return 0;
}
