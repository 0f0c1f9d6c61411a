// The optimal ate pairing of BLS12-381, for the Groth16 verifier (frw_verify.cpp): what ark-groth16 0.3.0's verifier.rs
// gets from ark-ec's Bls12 model (miller_loop over prepared G2 points, final_exponentiation) and examples/pok_sig.rs:47
// calls through Groth16::verify.  Written for this file, not transcribed: the tower and the loop are the textbook ones,
// arranged for clarity -- verification is three Miller loops and one final exponentiation per proof, milliseconds on one
// host core, and nothing here is on the witness path.
//
//   Fp    = frw_fq29.h's field, every value CANONICAL (< q) here: comparisons are limb comparisons, no bound bookkeeping
//   Fp2   = Fp[u]  / (u^2 + 1)
//   Fp6   = Fp2[v] / (v^3 - xi),  xi = 1 + u
//   Fp12  = Fp6[w] / (w^2 - v)            (so w^6 = xi: the sextic twist E': y^2 = x^3 + 4 xi  ->  E by (x, y) -> (x / w^2, y / w^3))
//
// Miller loop: affine coordinates on the twist, the slopes' denominators of all pairs of a step inverted together; the line
// through T (tangent) or T and Q, evaluated at P = (xP, yP) and scaled by w^3 (an element of Fp4, which the final
// exponentiation sends to one):   l = (lambda xT - yT) - (lambda xP) v + (yP v) w.
// The loop runs over |z| = 0xd201000000010000 and conjugates at the end (z < 0).
// Final exponentiation: f^((q^6 - 1)(q^2 + 1)) and then the power 3 (q^4 - q^2 + 1) / r
//   = l0 + l1 q + l2 q^2 + l3 q^3,  l3 = (z - 1)^2, l2 = l3 z, l1 = l2 z - l3, l0 = l1 z + 3   (Hayashida, Hayasaka, Teruya;
// the identity is checked with integers in tests/test_pairing_host.py): the result is the CUBE of the reduced ate pairing --
// as bilinear and non-degenerate as the pairing itself (3 does not divide r), and a verifier only compares such values.
#pragma once
#include "frw_fq29.h"

namespace frw {
namespace pairing {

#define FRW_HD __host__ __device__ inline

struct Fp { Fq29 v; };
FRW_HD Fp fp_zero() { Fp r; r.v = fq_zero(); return r; }
FRW_HD Fp fp_one() { Fp r; r.v = fq_const(FQ29_ONE); return r; }          // FQ29_ONE = 2^406 mod q, canonical
FRW_HD Fp fp_add(const Fp &a, const Fp &b) { Fp r; r.v = fq_canonical(fq_add(a.v, b.v)); return r; }
FRW_HD Fp fp_sub(const Fp &a, const Fp &b) { Fp r; r.v = fq_canonical(fq_sub<1>(a.v, b.v)); return r; }
FRW_HD Fp fp_neg(const Fp &a) { return fp_sub(fp_zero(), a); }
FRW_HD Fp fp_mul(const Fp &a, const Fp &b) { Fp r; r.v = fq_canonical(fq_mul(a.v, b.v)); return r; }
FRW_HD Fp fp_inv(const Fp &a) { Fp r; r.v = fq_canonical(fq_inv(a.v)); return r; }
FRW_HD bool fp_eq(const Fp &a, const Fp &b)
{
    uint32_t d = 0;
    for (int i = 0; i < NLQ; i++) d |= a.v.l[i] ^ b.v.l[i];
    return d == 0;
}
FRW_HD bool fp_is_zero(const Fp &a) { return fp_eq(a, fp_zero()); }
// ark-ff's 6 x u64 (x 2^384, canonical) <-> Fp
FRW_HD Fp fp_from_ark(const uint64_t *w) { Fp r; r.v = fq_canonical(fq_from_ark((const uint32_t *)w)); return r; }
FRW_HD void fp_to_ark(const Fp &a, uint64_t *w) { fq_to_ark(a.v, (uint32_t *)w); }

struct Fp2 { Fp c0, c1; };
FRW_HD Fp2 fp2_zero() { Fp2 r; r.c0 = r.c1 = fp_zero(); return r; }
FRW_HD Fp2 fp2_one() { Fp2 r; r.c0 = fp_one(); r.c1 = fp_zero(); return r; }
FRW_HD Fp2 fp2_add(const Fp2 &a, const Fp2 &b) { Fp2 r; r.c0 = fp_add(a.c0, b.c0); r.c1 = fp_add(a.c1, b.c1); return r; }
FRW_HD Fp2 fp2_sub(const Fp2 &a, const Fp2 &b) { Fp2 r; r.c0 = fp_sub(a.c0, b.c0); r.c1 = fp_sub(a.c1, b.c1); return r; }
FRW_HD Fp2 fp2_neg(const Fp2 &a) { Fp2 r; r.c0 = fp_neg(a.c0); r.c1 = fp_neg(a.c1); return r; }
FRW_HD Fp2 fp2_conj(const Fp2 &a) { Fp2 r; r.c0 = a.c0; r.c1 = fp_neg(a.c1); return r; }
FRW_HD Fp2 fp2_mul(const Fp2 &a, const Fp2 &b)
{
    const Fp m0 = fp_mul(a.c0, b.c0), m1 = fp_mul(a.c1, b.c1), m2 = fp_mul(fp_add(a.c0, a.c1), fp_add(b.c0, b.c1));
    Fp2 r; r.c0 = fp_sub(m0, m1); r.c1 = fp_sub(fp_sub(m2, m0), m1); return r;
}
FRW_HD Fp2 fp2_sqr(const Fp2 &a)
{
    const Fp t = fp_mul(a.c0, a.c1);
    Fp2 r; r.c0 = fp_mul(fp_add(a.c0, a.c1), fp_sub(a.c0, a.c1)); r.c1 = fp_add(t, t); return r;
}
FRW_HD Fp2 fp2_mul_fp(const Fp2 &a, const Fp &b) { Fp2 r; r.c0 = fp_mul(a.c0, b); r.c1 = fp_mul(a.c1, b); return r; }
FRW_HD Fp2 fp2_mul_xi(const Fp2 &a) { Fp2 r; r.c0 = fp_sub(a.c0, a.c1); r.c1 = fp_add(a.c0, a.c1); return r; }      // (a0 + a1 u)(1 + u)
FRW_HD Fp2 fp2_inv(const Fp2 &a)
{
    const Fp n = fp_inv(fp_add(fp_mul(a.c0, a.c0), fp_mul(a.c1, a.c1)));
    Fp2 r; r.c0 = fp_mul(a.c0, n); r.c1 = fp_neg(fp_mul(a.c1, n)); return r;
}
FRW_HD bool fp2_eq(const Fp2 &a, const Fp2 &b) { return fp_eq(a.c0, b.c0) && fp_eq(a.c1, b.c1); }
FRW_HD bool fp2_is_zero(const Fp2 &a) { return fp_is_zero(a.c0) && fp_is_zero(a.c1); }
FRW_HD Fp2 fp2_from_ark(const uint64_t *w) { Fp2 r; r.c0 = fp_from_ark(w); r.c1 = fp_from_ark(w + 6); return r; }

struct Fp6 { Fp2 c0, c1, c2; };
FRW_HD Fp6 fp6_zero() { Fp6 r; r.c0 = r.c1 = r.c2 = fp2_zero(); return r; }
FRW_HD Fp6 fp6_one() { Fp6 r; r.c0 = fp2_one(); r.c1 = r.c2 = fp2_zero(); return r; }
FRW_HD Fp6 fp6_add(const Fp6 &a, const Fp6 &b) { Fp6 r; r.c0 = fp2_add(a.c0, b.c0); r.c1 = fp2_add(a.c1, b.c1); r.c2 = fp2_add(a.c2, b.c2); return r; }
FRW_HD Fp6 fp6_sub(const Fp6 &a, const Fp6 &b) { Fp6 r; r.c0 = fp2_sub(a.c0, b.c0); r.c1 = fp2_sub(a.c1, b.c1); r.c2 = fp2_sub(a.c2, b.c2); return r; }
FRW_HD Fp6 fp6_neg(const Fp6 &a) { Fp6 r; r.c0 = fp2_neg(a.c0); r.c1 = fp2_neg(a.c1); r.c2 = fp2_neg(a.c2); return r; }
FRW_HD Fp6 fp6_mul_v(const Fp6 &a) { Fp6 r; r.c0 = fp2_mul_xi(a.c2); r.c1 = a.c0; r.c2 = a.c1; return r; }
FRW_HD Fp6 fp6_mul(const Fp6 &a, const Fp6 &b)
{
    const Fp2 v0 = fp2_mul(a.c0, b.c0), v1 = fp2_mul(a.c1, b.c1), v2 = fp2_mul(a.c2, b.c2);
    Fp6 r;
    r.c0 = fp2_add(v0, fp2_mul_xi(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c1, a.c2), fp2_add(b.c1, b.c2)), v1), v2)));
    r.c1 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c1), fp2_add(b.c0, b.c1)), v0), v1), fp2_mul_xi(v2));
    r.c2 = fp2_add(fp2_sub(fp2_sub(fp2_mul(fp2_add(a.c0, a.c2), fp2_add(b.c0, b.c2)), v0), v2), v1);
    return r;
}
FRW_HD Fp6 fp6_inv(const Fp6 &a)
{
    const Fp2 t0 = fp2_sub(fp2_sqr(a.c0), fp2_mul_xi(fp2_mul(a.c1, a.c2)));
    const Fp2 t1 = fp2_sub(fp2_mul_xi(fp2_sqr(a.c2)), fp2_mul(a.c0, a.c1));
    const Fp2 t2 = fp2_sub(fp2_sqr(a.c1), fp2_mul(a.c0, a.c2));
    const Fp2 d = fp2_add(fp2_mul(a.c0, t0), fp2_mul_xi(fp2_add(fp2_mul(a.c2, t1), fp2_mul(a.c1, t2))));
    const Fp2 di = fp2_inv(d);
    Fp6 r; r.c0 = fp2_mul(t0, di); r.c1 = fp2_mul(t1, di); r.c2 = fp2_mul(t2, di); return r;
}
FRW_HD bool fp6_eq(const Fp6 &a, const Fp6 &b) { return fp2_eq(a.c0, b.c0) && fp2_eq(a.c1, b.c1) && fp2_eq(a.c2, b.c2); }

struct Fp12 { Fp6 c0, c1; };
FRW_HD Fp12 fp12_one() { Fp12 r; r.c0 = fp6_one(); r.c1 = fp6_zero(); return r; }
FRW_HD Fp12 fp12_mul(const Fp12 &a, const Fp12 &b)
{
    const Fp6 aa = fp6_mul(a.c0, b.c0), bb = fp6_mul(a.c1, b.c1);
    Fp12 r;
    r.c1 = fp6_sub(fp6_sub(fp6_mul(fp6_add(a.c0, a.c1), fp6_add(b.c0, b.c1)), aa), bb);
    r.c0 = fp6_add(aa, fp6_mul_v(bb));
    return r;
}
FRW_HD Fp12 fp12_sqr(const Fp12 &a) { return fp12_mul(a, a); }
FRW_HD Fp12 fp12_conj(const Fp12 &a) { Fp12 r; r.c0 = a.c0; r.c1 = fp6_neg(a.c1); return r; }        // the q^6-power
FRW_HD Fp12 fp12_inv(const Fp12 &a)
{
    const Fp6 d = fp6_inv(fp6_sub(fp6_mul(a.c0, a.c0), fp6_mul_v(fp6_mul(a.c1, a.c1))));
    Fp12 r; r.c0 = fp6_mul(a.c0, d); r.c1 = fp6_neg(fp6_mul(a.c1, d)); return r;
}
FRW_HD bool fp12_eq(const Fp12 &a, const Fp12 &b) { return fp6_eq(a.c0, b.c0) && fp6_eq(a.c1, b.c1); }

// gamma[k] = xi^(k (q - 1) / 6), k = 1..5: w^q = gamma[1] w, so the q-power maps the coefficient a of v^i w^j to
// conj(a) gamma[2 i + j].  Computed once (host) by frobenius_constants().
struct FrobeniusConstants { Fp2 gamma[6]; };
FRW_HD Fp2 fp2_pow_words(const Fp2 &a, const uint32_t *e, int bits)
{
    Fp2 acc = fp2_one();
    for (int i = bits - 1; i >= 0; i--) {
        acc = fp2_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = fp2_mul(acc, a);
    }
    return acc;
}
inline FrobeniusConstants frobenius_constants()
{
    uint32_t e[12];
    for (int k = 0; k < 12; k++) e[k] = Q32_[k];
    e[0] -= 1;                                                        // q - 1 (q is odd: no borrow)
    uint64_t rem = 0;
    for (int k = 11; k >= 0; k--) { const uint64_t cur = (rem << 32) | e[k]; e[k] = (uint32_t)(cur / 6); rem = cur % 6; }
    Fp2 xi; xi.c0 = fp_one(); xi.c1 = fp_one();
    FrobeniusConstants fc;
    fc.gamma[0] = fp2_one();
    fc.gamma[1] = fp2_pow_words(xi, e, 384);
    for (int k = 2; k < 6; k++) fc.gamma[k] = fp2_mul(fc.gamma[k - 1], fc.gamma[1]);
    return fc;
}
FRW_HD Fp12 fp12_frobenius(const Fp12 &a, const FrobeniusConstants &fc)
{
    Fp12 r;
    r.c0.c0 = fp2_conj(a.c0.c0);
    r.c0.c1 = fp2_mul(fp2_conj(a.c0.c1), fc.gamma[2]);
    r.c0.c2 = fp2_mul(fp2_conj(a.c0.c2), fc.gamma[4]);
    r.c1.c0 = fp2_mul(fp2_conj(a.c1.c0), fc.gamma[1]);
    r.c1.c1 = fp2_mul(fp2_conj(a.c1.c1), fc.gamma[3]);
    r.c1.c2 = fp2_mul(fp2_conj(a.c1.c2), fc.gamma[5]);
    return r;
}

constexpr uint64_t Z_ABS = 0xd201000000010000ull;                     // |z|; z < 0

struct G1 { Fp x, y; bool inf; };
struct G2 { Fp2 x, y; bool inf; };
FRW_HD bool g1_on_curve(const G1 &p)                                  // y^2 = x^3 + 4
{
    if (p.inf) return true;
    Fp four = fp_one(); four = fp_add(four, four); four = fp_add(four, four);
    return fp_eq(fp_mul(p.y, p.y), fp_add(fp_mul(fp_mul(p.x, p.x), p.x), four));
}
FRW_HD bool g2_on_curve(const G2 &p)                                  // y^2 = x^3 + 4 (1 + u)
{
    if (p.inf) return true;
    Fp four = fp_one(); four = fp_add(four, four); four = fp_add(four, four);
    Fp2 b; b.c0 = four; b.c1 = four;
    return fp2_eq(fp2_sqr(p.y), fp2_add(fp2_mul(fp2_sqr(p.x), p.x), b));
}

// l = c00 + c01 v + (c11 v) w
FRW_HD Fp12 line_value(const Fp2 &lambda, const Fp2 &xt, const Fp2 &yt, const G1 &p)
{
    Fp12 l;
    l.c0.c0 = fp2_sub(fp2_mul(lambda, xt), yt);
    l.c0.c1 = fp2_neg(fp2_mul_fp(lambda, p.x));
    l.c0.c2 = fp2_zero();
    l.c1.c0 = fp2_zero();
    l.c1.c1.c0 = p.y; l.c1.c1.c1 = fp_zero();
    l.c1.c2 = fp2_zero();
    return l;
}

// prod_k f_{|z|, Q_k}(P_k), conjugated; pairs with a point at infinity contribute one.  m <= MAX_PAIRS.
constexpr int MAX_PAIRS = 4;
// `degenerate` (optional) is set when a step met a zero denominator -- a vertical tangent or chord, which points of the
// order-r subgroup never produce (T runs over multiples k Q, 0 < k < |z| < r, and T = +-Q only for k = +-1 mod r): the value
// returned is then meaningless and the caller must reject (a G2 point off the subgroup, vouched for by the caller, can do this).
FRW_HD Fp12 miller_loop(const G1 *ps, const G2 *qs, int m, bool *degenerate = nullptr)
{
    G1 p[MAX_PAIRS];
    G2 q[MAX_PAIRS], t[MAX_PAIRS];
    int n = 0;
    for (int k = 0; k < m && n < MAX_PAIRS; k++)
        if (!ps[k].inf && !qs[k].inf) { p[n] = ps[k]; q[n] = qs[k]; t[n] = qs[k]; n++; }
    Fp12 f = fp12_one();
    if (degenerate) *degenerate = false;
    if (n == 0) return f;
    // the denominators of one step, inverted together: inv[k] = 1 / den[k]
    auto invert_all = [&](Fp2 *den) {
        Fp2 pre[MAX_PAIRS];
        Fp2 acc = fp2_one();
        for (int k = 0; k < n; k++) { pre[k] = acc; acc = fp2_mul(acc, den[k]); }
        if (degenerate && fp2_is_zero(acc)) *degenerate = true;
        acc = fp2_inv(acc);
        for (int k = n - 1; k >= 0; k--) { const Fp2 d = den[k]; den[k] = fp2_mul(acc, pre[k]); acc = fp2_mul(acc, d); }
    };
    for (int bit = 62; bit >= 0; bit--) {                             // |z| has 64 bits; the top one starts T = Q
        f = fp12_sqr(f);
        Fp2 den[MAX_PAIRS];
        for (int k = 0; k < n; k++) den[k] = fp2_add(t[k].y, t[k].y);
        invert_all(den);
        for (int k = 0; k < n; k++) {
            const Fp2 x2 = fp2_sqr(t[k].x);
            const Fp2 lambda = fp2_mul(fp2_add(fp2_add(x2, x2), x2), den[k]);
            f = fp12_mul(f, line_value(lambda, t[k].x, t[k].y, p[k]));
            const Fp2 x3 = fp2_sub(fp2_sub(fp2_sqr(lambda), t[k].x), t[k].x);
            t[k].y = fp2_sub(fp2_mul(lambda, fp2_sub(t[k].x, x3)), t[k].y);
            t[k].x = x3;
        }
        if ((Z_ABS >> bit) & 1ull) {
            for (int k = 0; k < n; k++) den[k] = fp2_sub(q[k].x, t[k].x);
            invert_all(den);
            for (int k = 0; k < n; k++) {
                const Fp2 lambda = fp2_mul(fp2_sub(q[k].y, t[k].y), den[k]);
                f = fp12_mul(f, line_value(lambda, t[k].x, t[k].y, p[k]));
                const Fp2 x3 = fp2_sub(fp2_sub(fp2_sqr(lambda), t[k].x), q[k].x);
                t[k].y = fp2_sub(fp2_mul(lambda, fp2_sub(t[k].x, x3)), t[k].y);
                t[k].x = x3;
            }
        }
    }
    return fp12_conj(f);
}

// a^z for a in the cyclotomic subgroup (where the inverse is the conjugate)
FRW_HD Fp12 cyclotomic_exp_z(const Fp12 &a)
{
    Fp12 acc = a;
    for (int bit = 62; bit >= 0; bit--) {
        acc = fp12_sqr(acc);
        if ((Z_ABS >> bit) & 1ull) acc = fp12_mul(acc, a);
    }
    return fp12_conj(acc);
}

FRW_HD Fp12 final_exponentiation(const Fp12 &f, const FrobeniusConstants &fc)
{
    Fp12 m = fp12_mul(fp12_conj(f), fp12_inv(f));                     // ^(q^6 - 1)
    m = fp12_mul(fp12_frobenius(fp12_frobenius(m, fc), fc), m);       // ^(q^2 + 1)
    const Fp12 t0 = fp12_mul(cyclotomic_exp_z(m), fp12_conj(m));      // m^(z - 1)
    const Fp12 a = fp12_mul(cyclotomic_exp_z(t0), fp12_conj(t0));     // m^l3
    const Fp12 b = cyclotomic_exp_z(a);                               // m^l2
    const Fp12 c = fp12_mul(cyclotomic_exp_z(b), fp12_conj(a));       // m^l1
    const Fp12 d = fp12_mul(cyclotomic_exp_z(c), fp12_mul(fp12_sqr(m), m));   // m^l0
    const Fp12 fb = fp12_frobenius(fp12_frobenius(b, fc), fc);
    const Fp12 fa = fp12_frobenius(fp12_frobenius(fp12_frobenius(a, fc), fc), fc);
    return fp12_mul(fp12_mul(d, fp12_frobenius(c, fc)), fp12_mul(fb, fa));
}

#undef FRW_HD
}  // namespace pairing
}  // namespace frw
