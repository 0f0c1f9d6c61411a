/* ORACLE (test infrastructure, never linked into the product): BLS12-381 G1 on the CPU -- the group the multi-scalar
 * multiplications of a Groth16 prover run in (examples/pok_sig.rs:30-47 of the reference calls Groth16::<Bls12_381>::prove;
 * ark-groth16 0.3.0 prover.rs: h_acc = VariableBaseMSM::multi_scalar_mul(&pk.h_query, &h_assignment), and the same for
 * a_query / b_g1_query / l_query).
 *
 * PARITY UNPINNED against ark-ec / ark-bls12-381 0.3.0 (crates.io dependencies, absent from /root/reference; no Rust
 * toolchain here).  What this file restates is public mathematics with published parameters:
 *   q  = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
 *   E: y^2 = x^3 + 4 over F_q;  r = |G1| = the Fr modulus of the rest of this repository
 *   G1 generator (x, y) as published (ark-bls12-381 g1.rs G1_GENERATOR_X / _Y, the IETF pairing-friendly-curves draft)
 * pinned by tests/test_bls12_381.py: q and r from the BLS parametrisation z = -0xd201000000010000, the generator on the
 * curve and of order r, this file == oracle/bls12_381.py (Python integers) on random inputs, group laws.
 * Element format = ark-ff's: Fp384 as 6 x u64 little-endian limbs of x * 2^384 mod q; an affine point = x then y
 * (12 x u64), the point at infinity = all zero (0, 0 is not on the curve).
 * The multi-scalar multiplication here (bucket method, one window at a time) is the CPU figure bench.py prints beside the
 * GPU's; a restatement, not arkworks' implementation. */
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;
typedef struct { uint64_t l[6]; } fq;
typedef struct { fq x, y; } g1a;              /* affine; (0, 0) = infinity */
typedef struct { fq x, y, z; } g1j;           /* Jacobian; z = 0 = infinity */

static const uint64_t Q[6] = {0xb9feffffffffaaabull, 0x1eabfffeb153ffffull, 0x6730d2a0f6b0f624ull,
                              0x64774b84f38512bfull, 0x4b1ba7b6434bacd7ull, 0x1a0111ea397fe69aull};
static const uint64_t GX[6] = {0xfb3af00adb22c6bbull, 0x6c55e83ff97a1aefull, 0xa14e3a3f171bac58ull,
                               0xc3688c4f9774b905ull, 0x2695638c4fa9ac0full, 0x17f1d3a73197d794ull};
static const uint64_t GY[6] = {0x0caa232946c5e7e1ull, 0xd03cc744a2888ae4ull, 0x00db18cb2c04b3edull,
                               0xfcf5e095d5d00af6ull, 0xa09e30ed741d8ae4ull, 0x08b3f481e3aaa0f1ull};
static uint64_t Q_INV;                        /* -q^-1 mod 2^64 */
static fq FQ_ONE, FQ_R2;                      /* R mod q, R^2 mod q */
static g1a G1_GEN;
static pthread_once_t once = PTHREAD_ONCE_INIT;

static int fq_geq_q(const uint64_t a[6])
{
    for (int i = 5; i >= 0; i--) {
        if (a[i] > Q[i]) return 1;
        if (a[i] < Q[i]) return 0;
    }
    return 1;
}
static void fq_sub_q(uint64_t a[6])
{
    u128 b = 0;
    for (int i = 0; i < 6; i++) {
        const u128 d = (u128)a[i] - Q[i] - (uint64_t)b;
        a[i] = (uint64_t)d;
        b = (d >> 64) & 1;
    }
}
static void fq_add(fq *r, const fq *a, const fq *b)
{
    u128 c = 0;
    for (int i = 0; i < 6; i++) {
        c += (u128)a->l[i] + b->l[i];
        r->l[i] = (uint64_t)c;
        c >>= 64;
    }
    if (c || fq_geq_q(r->l)) fq_sub_q(r->l);          /* q < 2^381: no carry out of the top limb in fact */
}
static void fq_sub(fq *r, const fq *a, const fq *b)
{
    u128 br = 0;
    uint64_t t[6];
    for (int i = 0; i < 6; i++) {
        const u128 d = (u128)a->l[i] - b->l[i] - (uint64_t)br;
        t[i] = (uint64_t)d;
        br = (d >> 64) & 1;
    }
    if (br) {
        u128 c = 0;
        for (int i = 0; i < 6; i++) {
            c += (u128)t[i] + Q[i];
            t[i] = (uint64_t)c;
            c >>= 64;
        }
    }
    memcpy(r->l, t, sizeof t);
}
/* Montgomery product a b / 2^384 mod q (CIOS) */
static void fq_mul(fq *r, const fq *a, const fq *b)
{
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (u128)a->l[i] * b->l[j] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * Q_INV;
        c = ((u128)m * Q[0] + t[0]) >> 64;
        for (int j = 1; j < 6; j++) {
            c += (u128)m * Q[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
    }
    if (t[6] || fq_geq_q(t)) fq_sub_q(t);
    memcpy(r->l, t, 6 * sizeof(uint64_t));
}
static int fq_is_zero(const fq *a) { return !(a->l[0] | a->l[1] | a->l[2] | a->l[3] | a->l[4] | a->l[5]); }
static int fq_eq(const fq *a, const fq *b) { return !memcmp(a->l, b->l, sizeof a->l); }
static void fq_inv(fq *r, const fq *a)       /* a^(q-2) */
{
    uint64_t e[6];
    memcpy(e, Q, sizeof e);
    e[0] -= 2;
    fq acc = FQ_ONE;
    for (int i = 380; i >= 0; i--) {
        fq_mul(&acc, &acc, &acc);
        if ((e[i / 64] >> (i % 64)) & 1) fq_mul(&acc, &acc, a);
    }
    *r = acc;
}

static void init_once(void)
{
    uint64_t inv = 1;
    for (int i = 0; i < 6; i++) inv *= 2 - Q[0] * inv;               /* Newton: q^-1 mod 2^64 */
    Q_INV = 0 - inv;
    /* R = 2^384 mod q and R^2 by repeated doubling */
    fq one = {{1, 0, 0, 0, 0, 0}}, x = one;
    for (int i = 0; i < 768; i++) {
        fq_add(&x, &x, &x);
        if (i == 383) FQ_ONE = x;
    }
    FQ_R2 = x;
    fq gx, gy;
    memcpy(gx.l, GX, sizeof GX);
    memcpy(gy.l, GY, sizeof GY);
    fq_mul(&G1_GEN.x, &gx, &FQ_R2);
    fq_mul(&G1_GEN.y, &gy, &FQ_R2);
}
static void init(void) { pthread_once(&once, init_once); }

/* ---- G1 --------------------------------------------------------------------------------------------------------- */
static int g1a_is_inf(const g1a *p) { return fq_is_zero(&p->x) && fq_is_zero(&p->y); }
static void g1j_from_affine(g1j *r, const g1a *p)
{
    if (g1a_is_inf(p)) { memset(r, 0, sizeof *r); return; }
    r->x = p->x; r->y = p->y; r->z = FQ_ONE;
}
/* dbl-2009-l (a = 0) */
static void g1j_double(g1j *r, const g1j *p)
{
    if (fq_is_zero(&p->z)) { *r = *p; return; }
    fq a, b, c, d, e, f, t;
    fq_mul(&a, &p->x, &p->x);
    fq_mul(&b, &p->y, &p->y);
    fq_mul(&c, &b, &b);
    fq_add(&t, &p->x, &b); fq_mul(&t, &t, &t); fq_sub(&t, &t, &a); fq_sub(&t, &t, &c); fq_add(&d, &t, &t);
    fq_add(&e, &a, &a); fq_add(&e, &e, &a);
    fq_mul(&f, &e, &e);
    fq z3; fq_mul(&z3, &p->y, &p->z); fq_add(&z3, &z3, &z3);
    fq x3; fq_sub(&x3, &f, &d); fq_sub(&x3, &x3, &d);
    fq c8; fq_add(&c8, &c, &c); fq_add(&c8, &c8, &c8); fq_add(&c8, &c8, &c8);
    fq y3; fq_sub(&y3, &d, &x3); fq_mul(&y3, &e, &y3); fq_sub(&y3, &y3, &c8);
    r->x = x3; r->y = y3; r->z = z3;
}
/* madd-2007-bl, complete (doubling and cancellation handled) */
static void g1j_add_affine(g1j *r, const g1j *p, const g1a *q)
{
    if (g1a_is_inf(q)) { *r = *p; return; }
    if (fq_is_zero(&p->z)) { g1j_from_affine(r, q); return; }
    fq z1z1, u2, s2, h, hh, i, j, rr, v, t;
    fq_mul(&z1z1, &p->z, &p->z);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s2, &q->y, &p->z); fq_mul(&s2, &s2, &z1z1);
    if (fq_eq(&u2, &p->x)) {
        if (fq_eq(&s2, &p->y)) { g1j_double(r, p); return; }
        memset(r, 0, sizeof *r);
        return;
    }
    fq_sub(&h, &u2, &p->x);
    fq_mul(&hh, &h, &h);
    fq_add(&i, &hh, &hh); fq_add(&i, &i, &i);
    fq_mul(&j, &h, &i);
    fq_sub(&rr, &s2, &p->y); fq_add(&rr, &rr, &rr);
    fq_mul(&v, &p->x, &i);
    fq x3; fq_mul(&x3, &rr, &rr); fq_sub(&x3, &x3, &j); fq_sub(&x3, &x3, &v); fq_sub(&x3, &x3, &v);
    fq y3; fq_sub(&y3, &v, &x3); fq_mul(&y3, &rr, &y3); fq_mul(&t, &p->y, &j); fq_add(&t, &t, &t); fq_sub(&y3, &y3, &t);
    fq z3; fq_add(&z3, &p->z, &h); fq_mul(&z3, &z3, &z3); fq_sub(&z3, &z3, &z1z1); fq_sub(&z3, &z3, &hh);
    r->x = x3; r->y = y3; r->z = z3;
}
/* add-2007-bl, complete */
static void g1j_add(g1j *r, const g1j *p, const g1j *q)
{
    if (fq_is_zero(&q->z)) { *r = *p; return; }
    if (fq_is_zero(&p->z)) { *r = *q; return; }
    fq z1z1, z2z2, u1, u2, s1, s2, h, i, j, rr, v, t;
    fq_mul(&z1z1, &p->z, &p->z);
    fq_mul(&z2z2, &q->z, &q->z);
    fq_mul(&u1, &p->x, &z2z2);
    fq_mul(&u2, &q->x, &z1z1);
    fq_mul(&s1, &p->y, &q->z); fq_mul(&s1, &s1, &z2z2);
    fq_mul(&s2, &q->y, &p->z); fq_mul(&s2, &s2, &z1z1);
    if (fq_eq(&u1, &u2)) {
        if (fq_eq(&s1, &s2)) { g1j_double(r, p); return; }
        memset(r, 0, sizeof *r);
        return;
    }
    fq_sub(&h, &u2, &u1);
    fq_add(&i, &h, &h); fq_mul(&i, &i, &i);
    fq_mul(&j, &h, &i);
    fq_sub(&rr, &s2, &s1); fq_add(&rr, &rr, &rr);
    fq_mul(&v, &u1, &i);
    fq x3; fq_mul(&x3, &rr, &rr); fq_sub(&x3, &x3, &j); fq_sub(&x3, &x3, &v); fq_sub(&x3, &x3, &v);
    fq y3; fq_sub(&y3, &v, &x3); fq_mul(&y3, &rr, &y3); fq_mul(&t, &s1, &j); fq_add(&t, &t, &t); fq_sub(&y3, &y3, &t);
    fq z3; fq_add(&z3, &p->z, &q->z); fq_mul(&z3, &z3, &z3); fq_sub(&z3, &z3, &z1z1); fq_sub(&z3, &z3, &z2z2); fq_mul(&z3, &z3, &h);
    r->x = x3; r->y = y3; r->z = z3;
}
static void g1j_to_affine(g1a *r, const g1j *p)
{
    if (fq_is_zero(&p->z)) { memset(r, 0, sizeof *r); return; }
    fq zi, zi2, zi3;
    fq_inv(&zi, &p->z);
    fq_mul(&zi2, &zi, &zi);
    fq_mul(&zi3, &zi2, &zi);
    fq_mul(&r->x, &p->x, &zi2);
    fq_mul(&r->y, &p->y, &zi3);
}
/* many Jacobian points to affine with one inversion (Montgomery's trick) */
static void g1j_batch_to_affine(g1a *out, const g1j *in, size_t n)
{
    fq *pre = (fq *)malloc((n + 1) * sizeof(fq));
    fq acc = FQ_ONE;
    for (size_t i = 0; i < n; i++) {
        pre[i] = acc;
        if (!fq_is_zero(&in[i].z)) fq_mul(&acc, &acc, &in[i].z);
    }
    fq inv;
    fq_inv(&inv, &acc);
    for (size_t i = n; i-- > 0;) {
        if (fq_is_zero(&in[i].z)) { memset(&out[i], 0, sizeof out[i]); continue; }
        fq zi, zi2, zi3;
        fq_mul(&zi, &inv, &pre[i]);
        fq_mul(&inv, &inv, &in[i].z);
        fq_mul(&zi2, &zi, &zi);
        fq_mul(&zi3, &zi2, &zi);
        fq_mul(&out[i].x, &in[i].x, &zi2);
        fq_mul(&out[i].y, &in[i].y, &zi3);
    }
    free(pre);
}
static void g1_scalar_mul(g1j *r, const g1a *base, const uint64_t k[4])
{
    g1j acc;
    memset(&acc, 0, sizeof acc);
    for (int i = 255; i >= 0; i--) {
        g1j_double(&acc, &acc);
        if ((k[i / 64] >> (i % 64)) & 1) g1j_add_affine(&acc, &acc, base);
    }
    *r = acc;
}

/* ---- exported --------------------------------------------------------------------------------------------------- */
void frw_oracle_g1_generator(uint64_t out[12])
{
    init();
    memcpy(out, &G1_GEN, sizeof G1_GEN);
}
/* canonical integer limbs -> ark-ff's Montgomery limbs and back */
void frw_oracle_fq_to_montgomery(const uint64_t in[6], uint64_t out[6])
{
    init();
    fq a; memcpy(a.l, in, sizeof a.l);
    fq_mul((fq *)out, &a, &FQ_R2);
}
void frw_oracle_fq_from_montgomery(const uint64_t in[6], uint64_t out[6])
{
    init();
    fq a, one = {{1, 0, 0, 0, 0, 0}};
    memcpy(a.l, in, sizeof a.l);
    fq_mul((fq *)out, &a, &one);
}
int frw_oracle_g1_on_curve(const uint64_t p[12])
{
    init();
    const g1a *a = (const g1a *)p;
    if (g1a_is_inf(a)) return 1;
    fq y2, x3, four = FQ_ONE;
    fq_mul(&y2, &a->y, &a->y);
    fq_mul(&x3, &a->x, &a->x); fq_mul(&x3, &x3, &a->x);
    fq_add(&four, &four, &four); fq_add(&four, &four, &four);
    fq_add(&x3, &x3, &four);
    return fq_eq(&y2, &x3);
}
void frw_oracle_g1_scalar_mul(const uint64_t base[12], const uint64_t k[4], uint64_t out[12])
{
    init();
    g1j r;
    g1_scalar_mul(&r, (const g1a *)base, k);
    g1j_to_affine((g1a *)out, &r);
}
void frw_oracle_g1_add(const uint64_t a[12], const uint64_t b[12], uint64_t out[12])
{
    init();
    g1j j;
    g1j_from_affine(&j, (const g1a *)a);
    g1j_add_affine(&j, &j, (const g1a *)b);
    g1j_to_affine((g1a *)out, &j);
}

/* k_i G for many scalars: 8-bit fixed-base windows of the generator (what ark-groth16's generator does with
 * FixedBaseMSM to build h_query / a_query / l_query), threads over the scalars */
typedef struct { const g1a *table; const uint64_t *scalars; g1a *out; size_t lo, hi; } fb_job;
static void *fb_worker(void *arg)
{
    fb_job *job = (fb_job *)arg;
    const size_t n = job->hi - job->lo;
    g1j *acc = (g1j *)calloc(n ? n : 1, sizeof(g1j));
    for (size_t i = 0; i < n; i++) {
        const uint64_t *k = job->scalars + 4 * (job->lo + i);
        for (int w = 0; w < 32; w++) {
            const unsigned d = (unsigned)(k[w / 8] >> (8 * (w % 8))) & 0xff;
            if (d) g1j_add_affine(&acc[i], &acc[i], &job->table[w * 256 + d]);
        }
    }
    g1j_batch_to_affine(job->out + job->lo, acc, n);
    free(acc);
    return NULL;
}
void frw_oracle_g1_fixed_base(size_t count, const uint64_t *scalars /* count x 4, canonical */, uint64_t *out /* count x 12 */,
                              int threads)
{
    init();
    g1j *tj = (g1j *)calloc(32 * 256, sizeof(g1j));
    g1a *table = (g1a *)calloc(32 * 256, sizeof(g1a));
    g1j base;
    g1j_from_affine(&base, &G1_GEN);
    for (int w = 0; w < 32; w++) {
        g1a ba;
        g1j_to_affine(&ba, &base);
        for (int d = 1; d < 256; d++) g1j_add_affine(&tj[w * 256 + d], &tj[w * 256 + d - 1], &ba);
        for (int k = 0; k < 8; k++) g1j_double(&base, &base);
    }
    g1j_batch_to_affine(table, tj, 32 * 256);
    free(tj);
    if (threads < 1) threads = 1;
    if (threads > 256) threads = 256;
    pthread_t th[256];
    fb_job jobs[256];
    const size_t step = (count + threads - 1) / threads;
    int used = 0;
    for (int t = 0; t < threads; t++) {
        const size_t lo = (size_t)t * step, hi = lo + step < count ? lo + step : count;
        if (lo >= hi) break;
        jobs[used] = (fb_job){table, scalars, (g1a *)out, lo, hi};
        pthread_create(&th[used], NULL, fb_worker, &jobs[used]);
        used++;
    }
    for (int t = 0; t < used; t++) pthread_join(th[t], NULL);
    free(table);
}

/* sum k_i P_i by the bucket method: windows of c bits, one after the other, signed digits; a thread per window.
 * The CPU figure beside the GPU's (bench.py); also == sum of scalar multiplications in tests. */
typedef struct { const g1a *bases; const uint64_t *scalars; size_t count; int c, window, num_windows; g1j result; } msm_job;
static int digit_signed(const uint64_t k[4], int c, int w, int nw)
{
    /* signed c-bit recoding: digit w of k with the carries of the lower windows folded in */
    int carry = 0, d = 0;
    for (int j = 0; j <= w; j++) {
        const int bit = j * c;
        uint64_t v = bit < 256 ? k[bit / 64] >> (bit % 64) : 0;
        if (bit < 256 && bit % 64 + c > 64 && bit / 64 + 1 < 4) v |= k[bit / 64 + 1] << (64 - bit % 64);
        d = (int)(v & ((1u << c) - 1)) + carry;
        carry = 0;
        if (j + 1 < nw && d >= (1 << (c - 1))) { d -= 1 << c; carry = 1; }
    }
    return d;
}
static void *msm_worker(void *arg)
{
    msm_job *job = (msm_job *)arg;
    const int nb = 1 << (job->c - 1);
    g1j *buckets = (g1j *)calloc((size_t)nb + 1, sizeof(g1j));
    for (size_t i = 0; i < job->count; i++) {
        const int d = digit_signed(job->scalars + 4 * i, job->c, job->window, job->num_windows);
        if (!d) continue;
        g1a p = job->bases[i];
        if (d < 0 && !g1a_is_inf(&p)) { fq zero = {{0}}; fq_sub(&p.y, &zero, &p.y); }
        const int b = d < 0 ? -d : d;
        g1j_add_affine(&buckets[b], &buckets[b], &p);
    }
    g1j run, sum;
    memset(&run, 0, sizeof run);
    memset(&sum, 0, sizeof sum);
    for (int b = nb; b >= 1; b--) {
        g1j_add(&run, &run, &buckets[b]);
        g1j_add(&sum, &sum, &run);
    }
    job->result = sum;
    free(buckets);
    return NULL;
}
void frw_oracle_g1_msm(size_t count, const uint64_t *bases /* count x 12 */, const uint64_t *scalars /* count x 4, canonical */,
                       uint64_t out[12], int window_bits, int threads)
{
    init();
    const int c = window_bits < 2 ? 2 : window_bits > 20 ? 20 : window_bits;
    const int nw = (255 + c - 1) / c + 1;                     /* one more for the carry of the top window */
    msm_job *jobs = (msm_job *)calloc(nw, sizeof(msm_job));
    pthread_t *th = (pthread_t *)calloc(nw, sizeof(pthread_t));
    if (threads < 1) threads = 1;
    for (int w0 = 0; w0 < nw; w0 += threads) {
        const int w1 = w0 + threads < nw ? w0 + threads : nw;
        for (int w = w0; w < w1; w++) {
            jobs[w] = (msm_job){(const g1a *)bases, scalars, count, c, w, nw, {{{0}}, {{0}}, {{0}}}};
            pthread_create(&th[w], NULL, msm_worker, &jobs[w]);
        }
        for (int w = w0; w < w1; w++) pthread_join(th[w], NULL);
    }
    g1j acc;
    memset(&acc, 0, sizeof acc);
    for (int w = nw - 1; w >= 0; w--) {
        for (int k = 0; k < c; k++) g1j_double(&acc, &acc);
        g1j_add(&acc, &acc, &jobs[w].result);
    }
    g1j_to_affine((g1a *)out, &acc);
    free(jobs);
    free(th);
}
