// frw_fq29.h -- BLS12-381 Fq (the 381-bit base field of G1) on the device in fourteen 29-bit limbs, and G1 in XYZZ
// coordinates on top of it: what the multi-scalar multiplications of a Groth16 prover compute with
// (examples/pok_sig.rs:30-47 -> ark-groth16 0.3.0 prover.rs: VariableBaseMSM over pk.h_query, a_query, b_g1_query, l_query).
//
// Same design as frw_fr29.h, for the same reason: v_mad_u64_u32 is what the chip multiplies with, and with 29-bit limbs a
// column of a 14 x 14 product (14 x 2^58) and the column of the Montgomery reduction on top of it (14 x 2^58 more) fit 64
// bits, so a product is 2 x 196 chained multiply-adds with no carry handling in between.  14 x 29 = 406 bits for a 381-bit
// modulus: 2^406 = 2^25 q of headroom, so sums and differences are never reduced -- a difference adds K q, K >= the bound of
// what is subtracted -- and only the products bring values back under 2 q.
// Montgomery radix R'' = 2^406; ark-ff's representation is x 2^384 (six 64-bit limbs), converted on the way in and out.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace frw {

constexpr int NLQ = 14;
constexpr uint32_t MQ29 = (1u << 29) - 1u;

struct Fq29 { uint32_t l[NLQ]; };
struct LimbsQ { uint32_t l[NLQ]; };

// q = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab, 32-bit words
constexpr uint32_t Q32_[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu, 0xf6b0f624u, 0x6730d2a0u,
                               0xf38512bfu, 0x64774b84u, 0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
constexpr LimbsQ make_kq29(uint32_t K)
{
    uint32_t w[14] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t carry = 0;
    for (int k = 0; k < 12; k++) {
        const uint64_t t = (uint64_t)K * Q32_[k] + carry;
        w[k] = (uint32_t)t;
        carry = t >> 32;
    }
    w[12] = (uint32_t)carry;
    LimbsQ r{};
    for (int i = 0; i < NLQ; i++) {
        const int bit = 29 * i, k = bit >> 5, s = bit & 31;
        const uint64_t both = (uint64_t)w[k] | ((uint64_t)w[k + 1] << 32);
        r.l[i] = (uint32_t)(both >> s) & (i == NLQ - 1 ? 0xffffffffu : MQ29);
    }
    return r;
}
constexpr LimbsQ KQ29_1 = make_kq29(1), KQ29_4 = make_kq29(4), KQ29_16 = make_kq29(16), KQ29_64 = make_kq29(64),
                 KQ29_256 = make_kq29(256), KQ29_1024 = make_kq29(1024);
template <uint32_t K> constexpr const LimbsQ &kq29_table()
{
    static_assert(K == 1 || K == 4 || K == 16 || K == 64 || K == 256 || K == 1024, "table missing");
    return K == 1 ? KQ29_1 : K == 4 ? KQ29_4 : K == 16 ? KQ29_16 : K == 64 ? KQ29_64 : K == 256 ? KQ29_256 : KQ29_1024;
}
static_assert(KQ29_1.l[0] == 0x1fffaaabu && KQ29_1.l[13] == 0xdu && KQ29_16.l[13] == 0xd0u, "q in 29-bit limbs");
constexpr uint32_t QINV29 = 0x1ffcfffdu;      // -q^-1 mod 2^29
constexpr uint32_t QPOSINV29 = 0x00030003u;   //  q^-1 mod 2^29
static_assert(((uint64_t)KQ29_1.l[0] * QPOSINV29 & MQ29) == 1u && ((QINV29 + QPOSINV29) & MQ29) == 0u, "inverse of q mod 2^29");
// 2^406 mod q (one), 2^428 mod q (x 2^384 -> x 2^406 by one product), 2^384 mod q (back); tests/test_fq29_host.py re-derives them
constexpr LimbsQ FQ29_ONE = {{0x03a9fb84u, 0x0ba00690u, 0x071288f1u, 0x0f59bcc5u, 0x126cb614u, 0x0585bf36u, 0x1b85ac3du,
                              0x1cf856fau, 0x1891ecbdu, 0x1a7eec05u, 0x155a88f0u, 0x0741ac6du, 0x1317c30fu, 0x00000009u}};
constexpr LimbsQ FQ29_C_IN = {{0x1fddebbdu, 0x1a4f5474u, 0x0291f399u, 0x14d03b3cu, 0x0f6cad2cu, 0x1b4cabcau, 0x1592827cu,
                               0x021c6ac7u, 0x1ec52a84u, 0x16fd5ec4u, 0x0c960da6u, 0x0fd2af6bu, 0x13263591u, 0x0000000bu}};
constexpr LimbsQ FQ29_C_OUT = {{0x0002fffdu, 0x10480000u, 0x0300009du, 0x08001788u, 0x158baebfu, 0x0c2ba9e3u, 0x1d157d22u,
                                0x0a6e0a4au, 0x0d77ce58u, 0x1d12b763u, 0x1701c6a5u, 0x1501c926u, 0x1f65ec3fu, 0x0000000au}};

__host__ __device__ __forceinline__ Fq29 fq_const(const LimbsQ &c)
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = c.l[i];
    return r;
}
__host__ __device__ __forceinline__ Fq29 fq_zero()
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = 0;
    return r;
}

// 12 x 32-bit words (a 384-bit integer, little-endian) <-> limbs
__host__ __device__ __forceinline__ Fq29 fq_unpack(const uint32_t *w)
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) {
        const int bit = 29 * i, k = bit >> 5, s = bit & 31;
        const uint64_t both = (uint64_t)w[k] | (k + 1 < 12 ? (uint64_t)w[k + 1] << 32 : 0ull);
        r.l[i] = (uint32_t)(both >> s) & MQ29;
    }
    return r;
}
// normalised limbs, value < 2^384
__host__ __device__ __forceinline__ void fq_pack(const Fq29 &a, uint32_t *w)
{
#pragma unroll
    for (int k = 0; k < 12; k++) {
        const int bit = 32 * k, i = bit / 29, s = bit - 29 * i;
        uint32_t v = a.l[i] >> s;
        int filled = 29 - s;
        if (filled < 32 && i + 1 < NLQ) { v |= a.l[i + 1] << filled; filled += 29; }
        if (filled < 32 && i + 2 < NLQ) v |= a.l[i + 2] << filled;
        w[k] = v;
    }
}

__host__ __device__ __forceinline__ void fq_normalise(Fq29 &a)
{
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NLQ - 1; i++) {
        const uint32_t x = a.l[i] + c;
        a.l[i] = x & MQ29;
        c = x >> 29;
    }
    a.l[NLQ - 1] += c;
}
// limbs are signed 32-bit differences, the value is >= 0 unless the return value says otherwise (top limb negative)
__host__ __device__ __forceinline__ int fq_normalise_signed(Fq29 &a)
{
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NLQ - 1; i++) {
        const int32_t x = (int32_t)a.l[i] + c;
        a.l[i] = (uint32_t)x & MQ29;
        c = x >> 29;
    }
    const int32_t top = (int32_t)a.l[NLQ - 1] + c;
    a.l[NLQ - 1] = (uint32_t)top;
    return top < 0;
}
__host__ __device__ __forceinline__ Fq29 fq_add(const Fq29 &a, const Fq29 &b)
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = a.l[i] + b.l[i];
    fq_normalise(r);
    return r;
}
// a - b + K q  (b < K q), normalised
template <uint32_t K>
__host__ __device__ __forceinline__ Fq29 fq_sub(const Fq29 &a, const Fq29 &b)
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = a.l[i] - b.l[i] + kq29_table<K>().l[i];
    (void)fq_normalise_signed(r);
    return r;
}
// K q - b  (b < K q): the negative
template <uint32_t K>
__host__ __device__ __forceinline__ Fq29 fq_neg(const Fq29 &b)
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = kq29_table<K>().l[i] - b.l[i];
    (void)fq_normalise_signed(r);
    return r;
}

// Montgomery product a b / 2^406 mod q in two halves: the 27 columns of the integer product, and their reduction.
// Operands: normalised limbs, except that ONE operand of one product may have limbs up to 2^30 (a column is then
// < 14 2^59, and the reduction adds < 14 2^58 to it: < 2^64).  The halves exist on their own because columns add: a b + c d
// costs two multiplications and ONE reduction (both products of normalised operands: 2 x 14 2^58 + 14 2^58 < 2^64).
// Result of a reduction: normalised, < (sum of the products) / 2^406 + q  (< 2 q whenever that sum is < 2^406 q: two
// values below 2^12 q each, or two such pairs).
__host__ __device__ __forceinline__ void fq_mul_cols(const Fq29 &a, const Fq29 &b, uint64_t (&col)[2 * NLQ])
{
#pragma unroll
    for (int k = 0; k < 2 * NLQ - 1; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = (k < NLQ ? 0 : k - (NLQ - 1)); i <= (k < NLQ ? k : NLQ - 1); i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        col[k] = acc;
    }
    col[2 * NLQ - 1] = 0;
}
// the columns of a^2 in 105 multiplications instead of 196: cross terms once, with the doubled operand (a: normalised)
__host__ __device__ __forceinline__ void fq_sqr_cols(const Fq29 &a, uint64_t (&col)[2 * NLQ])
{
    uint32_t twice[NLQ];
#pragma unroll
    for (int i = 0; i < NLQ; i++) twice[i] = a.l[i] << 1;
#pragma unroll
    for (int k = 0; k < 2 * NLQ - 1; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = (k < NLQ ? 0 : k - (NLQ - 1)); 2 * i < k; i++) acc += (uint64_t)twice[i] * a.l[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a.l[k >> 1] * a.l[k >> 1];
        col[k] = acc;
    }
    col[2 * NLQ - 1] = 0;
}
// the columns of a b + c d at once (normalised operands: 2 x 14 2^58 per column, and the reduction's 14 2^58 on top still fit)
__host__ __device__ __forceinline__ void fq_mul2_cols(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d, uint64_t (&col)[2 * NLQ])
{
#pragma unroll
    for (int k = 0; k < 2 * NLQ - 1; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = (k < NLQ ? 0 : k - (NLQ - 1)); i <= (k < NLQ ? k : NLQ - 1); i++) {
            acc += (uint64_t)a.l[i] * b.l[k - i];
            acc += (uint64_t)c.l[i] * d.l[k - i];
        }
        col[k] = acc;
    }
    col[2 * NLQ - 1] = 0;
}
__host__ __device__ __forceinline__ Fq29 fq_reduce_cols(uint64_t (&col)[2 * NLQ])
{
#pragma unroll
    for (int i = 0; i < NLQ; i++) {
        const uint32_t m = ((uint32_t)col[i] * QINV29) & MQ29;
#pragma unroll
        for (int j = 0; j < NLQ; j++) col[i + j] += (uint64_t)m * KQ29_1.l[j];
        col[i + 1] += col[i] >> 29;                                  // the low 29 bits of column i are zero now
    }
    Fq29 r;
#pragma unroll
    for (int k = NLQ; k < 2 * NLQ - 1; k++) {
        r.l[k - NLQ] = (uint32_t)col[k] & MQ29;
        col[k + 1] += col[k] >> 29;
    }
    r.l[NLQ - 1] = (uint32_t)col[2 * NLQ - 1];
    return r;
}
__host__ __device__ __forceinline__ Fq29 fq_mul(const Fq29 &a, const Fq29 &b)
{
    uint64_t col[2 * NLQ];
    fq_mul_cols(a, b, col);
    return fq_reduce_cols(col);
}
__host__ __device__ __forceinline__ Fq29 fq_sqr(const Fq29 &a)
{
    uint64_t col[2 * NLQ];
    fq_sqr_cols(a, col);
    return fq_reduce_cols(col);
}
// a b - c d + K q d, i.e. a b - c d mod q, with one reduction (c < K q; all four normalised): what the Y coordinate of
// every point formula ends with.  < (a b + K q d) / 2^406 + q.
template <uint32_t K>
__host__ __device__ __forceinline__ Fq29 fq_mul_sub(const Fq29 &a, const Fq29 &b, const Fq29 &c, const Fq29 &d)
{
    uint64_t col[2 * NLQ], col2[2 * NLQ];
    fq_mul_cols(a, b, col);
    fq_mul_cols(fq_neg<K>(c), d, col2);
#pragma unroll
    for (int k = 0; k < 2 * NLQ - 1; k++) col[k] += col2[k];
    return fq_reduce_cols(col);
}

// a < 2 q (normalised)  ->  the canonical representative < q
__host__ __device__ __forceinline__ Fq29 fq_canonical(const Fq29 &a)
{
    Fq29 d;
#pragma unroll
    for (int i = 0; i < NLQ; i++) d.l[i] = a.l[i] - KQ29_1.l[i];
    const int neg = fq_normalise_signed(d);
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = neg ? a.l[i] : d.l[i];
    return r;
}
// any lazily reduced value (< 2^12 q) -> canonical: a product with one, then the conditional subtraction
__host__ __device__ __forceinline__ Fq29 fq_reduce(const Fq29 &a) { return fq_canonical(fq_mul(a, fq_const(FQ29_ONE))); }

// a = 0 mod q?  a < 2048 q, normalised.  A multiple k q of q has k = a.l[0] q^-1 mod 2^29: anything else is ruled out by two
// instructions, and only a candidate (one value in 2^18) pays for the exact reduction.
__host__ __device__ __forceinline__ bool fq_is_zero(const Fq29 &a)
{
    if (((a.l[0] * QPOSINV29) & MQ29) >= 2048u) return false;
    const Fq29 c = fq_reduce(a);
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < NLQ; i++) any |= c.l[i];
    return any == 0;
}

// ark-ff's x 2^384 (12 words) -> x 2^406 in limbs (< 2 q), and back to canonical words
__host__ __device__ __forceinline__ Fq29 fq_from_ark(const uint32_t *w) { return fq_mul(fq_unpack(w), fq_const(FQ29_C_IN)); }
__host__ __device__ __forceinline__ void fq_to_ark(const Fq29 &a, uint32_t *w) { fq_pack(fq_canonical(fq_mul(a, fq_const(FQ29_C_OUT))), w); }

// a^(q - 2): the inverse (a != 0 mod q) by Fermat; ~380 squarings + ~230 products.  Kept as the cross-check of fq_inv below
// (tests/test_fq29_host.py); the kernels call fq_inv.
__host__ __device__ inline Fq29 fq_inv_fermat(const Fq29 &a)
{
    // q - 2 in 32-bit words
    uint32_t e[12];
#pragma unroll
    for (int k = 0; k < 12; k++) e[k] = Q32_[k];
    e[0] -= 2;
    Fq29 acc = fq_const(FQ29_ONE);
    for (int i = 380; i >= 0; i--) {
        acc = fq_sqr(acc);
        if ((e[i >> 5] >> (i & 31)) & 1u) acc = fq_mul(acc, a);
    }
    return acc;
}
// The inverse by the binary extended Euclidean algorithm on twelve 32-bit words: the conversion of a result to affine
// coordinates is the tail of every sum and of a proof (one inversion on the critical path of a proof made alone), and
// Fermat's 610 dependent products are 0.6 ms of one lane.  Invariant u = x1 A, v = x2 A (mod q), v odd: if u is odd
// make it the larger of the two and subtract (u -= v, x1 -= x2), then halve u and x1; u reaches 0 in at most 2 x 382 rounds of
// ~200 plain integer instructions, v is then 1 and x2 = 1 / A.  A = a 2^406 comes in, (1 / a) 2^406 = x2 2^(3 x 406) / 2^406 goes out.
// 2^1218 mod q; tests/test_fq29_host.py re-derives it
constexpr LimbsQ FQ29_R3 = {{0x09217d6au, 0x1d6118bau, 0x1114b11cu, 0x0126aee7u, 0x0a55e2c4u, 0x04d63ce0u, 0x154ff87du,
                             0x14555478u, 0x1d1bdc0du, 0x161f98d4u, 0x1d74e921u, 0x09b4345au, 0x1e5ecfb8u, 0x0000000au}};
// 32-bit addition / subtraction with carry: the device has them as single instructions (v_addc_co / v_subb_co); written
// through 64 bits the compiler emitted 64-bit adds and register shuffles, three times the instructions
__host__ __device__ __forceinline__ uint32_t add_c(uint32_t a, uint32_t b, uint32_t &carry)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned out;
    const uint32_t r = __builtin_addc(a, b, carry, &out);
    carry = out;
    return r;
#else
    const uint64_t t = (uint64_t)a + b + carry;
    carry = (uint32_t)(t >> 32);
    return (uint32_t)t;
#endif
}
__host__ __device__ __forceinline__ uint32_t sub_b(uint32_t a, uint32_t b, uint32_t &borrow)
{
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned out;
    const uint32_t r = __builtin_subc(a, b, borrow, &out);
    borrow = out;
    return r;
#else
    const uint64_t t = (uint64_t)a - b - borrow;
    borrow = (uint32_t)(t >> 32) & 1u;
    return (uint32_t)t;
#endif
}
// The same inverse thirty-one rounds at a time (T. Pornin, "Optimized binary GCD for modular inversion", 2020, algorithm 2): the rounds'
// decisions -- is a odd, is a < b -- are taken on 64-bit stand-ins (the low 31 bits of a and b, exact, under their top 33 bits), the
// rounds' effect is four factors of 32 bits, and the 381-bit values see one multiply-add pass per thirty-one rounds: a' = (f0 a + g0 b) /
// 2^31, b' = (f1 a + g1 b) / 2^31 (exact; a sign that a misjudged comparison left is taken back with its factors), and the same on
// u, v modulo q with the division made exact by a multiple of q.  25 x 31 >= 2 x 381 - 1 rounds; a = u y, b = v y (mod q) hold exactly
// throughout, so b == 1 at the end PROVES v = 1 / y -- anything else (y = 0, or a bound that did not hold) falls back to fq_inv_binary.
// 1,200 instructions per thirty-one rounds instead of 6,000: 0.3 -> 0.05 ms of the one inversion a proof made alone waits for.
constexpr uint32_t Q_NEG_INV31 = 0x7ffcfffdu;                 // -1 / q mod 2^31 (tests/test_fq29_host.py re-derives it)
// r (14 words, two's complement) = f x + g y;  x, y: 13 words, unsigned;  |f|, |g| <= 2^31
__host__ __device__ __forceinline__ void inv_lincomb(const uint32_t (&x)[13], int64_t f, const uint32_t (&y)[13], int64_t g, uint32_t (&r)[14])
{
    const uint32_t af = (uint32_t)(f < 0 ? -f : f), ag = (uint32_t)(g < 0 ? -g : g);
    const uint32_t nf = f < 0 ? 0xffffffffu : 0u, ng = g < 0 ? 0xffffffffu : 0u;
    uint64_t cp = 0, cq = 0;
    uint32_t bp = nf & 1u, bq = ng & 1u, carry = 0;           // (-p = ~p + 1: the + 1 rides in as a carry)
#pragma unroll
    for (int k = 0; k < 14; k++) {
        uint32_t pw, qw;
        if (k < 13) {
            cp += (uint64_t)af * x[k]; pw = (uint32_t)cp; cp >>= 32;
            cq += (uint64_t)ag * y[k]; qw = (uint32_t)cq; cq >>= 32;
        } else {
            pw = (uint32_t)cp; qw = (uint32_t)cq;
        }
        pw ^= nf; qw ^= ng;
        const uint64_t tp = (uint64_t)pw + bp, tq = (uint64_t)qw + bq;
        bp = (uint32_t)(tp >> 32); bq = (uint32_t)(tq >> 32);
        const uint64_t t = (uint64_t)(uint32_t)tp + (uint32_t)tq + carry;
        r[k] = (uint32_t)t; carry = (uint32_t)(t >> 32);
    }
}
// r >>= 31 (arithmetic), into 13 words and the sign (all ones when negative)
__host__ __device__ __forceinline__ uint32_t inv_shift31(const uint32_t (&r)[14], uint32_t (&out)[13])
{
#pragma unroll
    for (int k = 0; k < 13; k++) out[k] = (r[k] >> 31) | (r[k + 1] << 1);
    return 0u - (r[13] >> 31);
}
__host__ __device__ inline bool fq_inv_words31(const uint32_t (&y)[12], uint32_t (&inv)[12])
{
    uint32_t a[13], b[13], u[13], v[13];
#pragma unroll
    for (int k = 0; k < 13; k++) { a[k] = k < 12 ? y[k] : 0u; b[k] = k < 12 ? Q32_[k] : 0u; u[k] = k == 0 ? 1u : 0u; v[k] = 0u; }
#pragma nounroll
    for (int round = 0; round < 25; round++) {
        // the stand-ins: the three words under the highest bit of a | b (selects: a register array indexed at run time would live in scratch)
        uint32_t a2 = 0, a1 = 0, a0 = 0, b2 = 0, b1 = 0, b0 = 0, found = 0;
#pragma unroll
        for (int k = 11; k >= 2; k--) {
            const uint32_t take = ((a[k] | b[k]) != 0u && !found) ? 0xffffffffu : 0u;
            a2 |= take & a[k]; a1 |= take & a[k - 1]; a0 |= take & a[k - 2];
            b2 |= take & b[k]; b1 |= take & b[k - 1]; b0 |= take & b[k - 2];
            found |= take;
        }
        uint64_t ta, tb;
        if (found) {
            const int s = __builtin_clz(a2 | b2);                                        // (a2 | b2 != 0)
            const uint64_t xa = ((uint64_t)a2 << 32) | a1, xb = ((uint64_t)b2 << 32) | b1;
            const uint64_t top_a = s ? (xa << s) | (a0 >> (32 - s)) : xa, top_b = s ? (xb << s) | (b0 >> (32 - s)) : xb;
            ta = ((top_a >> 31) << 31) | (a[0] & 0x7fffffffu);
            tb = ((top_b >> 31) << 31) | (b[0] & 0x7fffffffu);
        } else {
            ta = ((uint64_t)a[1] << 32) | a[0];
            tb = ((uint64_t)b[1] << 32) | b[0];
        }
        int64_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll 1
        for (int i = 0; i < 31; i++) {
            const bool odd = (ta & 1u) != 0, swap = odd && ta < tb;
            const uint64_t sa = swap ? tb : ta, sb = swap ? ta : tb;
            const int64_t sf0 = swap ? f1 : f0, sg0 = swap ? g1 : g0, sf1 = swap ? f0 : f1, sg1 = swap ? g0 : g1;
            ta = (odd ? sa - sb : sa) >> 1; tb = sb;
            f0 = odd ? sf0 - sf1 : sf0; g0 = odd ? sg0 - sg1 : sg0;
            f1 = sf1 * 2; g1 = sg1 * 2;
        }
        uint32_t r[14], na[13], nb[13];
        inv_lincomb(a, f0, b, g0, r);
        uint32_t neg = inv_shift31(r, na);
        if (neg) {                                                                       // (a misjudged comparison: -a, with its factors)
            uint32_t c = 1;
#pragma unroll
            for (int k = 0; k < 13; k++) { const uint64_t t = (uint64_t)(~na[k]) + c; na[k] = (uint32_t)t; c = (uint32_t)(t >> 32); }
            f0 = -f0; g0 = -g0;
        }
        inv_lincomb(a, f1, b, g1, r);
        neg = inv_shift31(r, nb);
        if (neg) {
            uint32_t c = 1;
#pragma unroll
            for (int k = 0; k < 13; k++) { const uint64_t t = (uint64_t)(~nb[k]) + c; nb[k] = (uint32_t)t; c = (uint32_t)(t >> 32); }
            f1 = -f1; g1 = -g1;
        }
        // u' = (f0 u + g0 v) / 2^31, v' = (f1 u + g1 v) / 2^31 (mod q): + k q with k = -t / q mod 2^31 makes the division exact;
        // |f u + g v| <= 2^31 q and k q < 2^31 q, so the quotient is in (-q, 2 q): one conditional addition, one subtraction
        uint32_t nu[13], nv[13];
#pragma unroll
        for (int which = 0; which < 2; which++) {
            inv_lincomb(u, which ? f1 : f0, v, which ? g1 : g0, r);
            const uint32_t kq = ((r[0] & 0x7fffffffu) * Q_NEG_INV31) & 0x7fffffffu;
            uint64_t c = 0;
#pragma unroll
            for (int k = 0; k < 14; k++) {
                c += (uint64_t)r[k] + (k < 12 ? (uint64_t)kq * Q32_[k] : 0u);
                r[k] = (uint32_t)c; c >>= 32;
            }
            uint32_t (&out)[13] = which ? nv : nu;
            const uint32_t below = inv_shift31(r, out);
            uint32_t carry = 0;
#pragma unroll
            for (int k = 0; k < 13; k++) { const uint64_t t = (uint64_t)out[k] + (k < 12 ? Q32_[k] & below : 0u) + carry; out[k] = (uint32_t)t; carry = (uint32_t)(t >> 32); }
            uint32_t d[13], borrow = 0;
#pragma unroll
            for (int k = 0; k < 13; k++) d[k] = sub_b(out[k], k < 12 ? Q32_[k] : 0u, borrow);
            const uint32_t keep = 0u - borrow;                                           // out < q: keep it
#pragma unroll
            for (int k = 0; k < 13; k++) out[k] = (keep & out[k]) | (~keep & d[k]);
        }
#pragma unroll
        for (int k = 0; k < 13; k++) { a[k] = na[k]; b[k] = nb[k]; u[k] = nu[k]; v[k] = nv[k]; }
    }
    uint32_t rest = b[0] ^ 1u;
#pragma unroll
    for (int k = 1; k < 13; k++) rest |= b[k];
#pragma unroll
    for (int k = 0; k < 13; k++) rest |= a[k];
#pragma unroll
    for (int k = 0; k < 12; k++) inv[k] = v[k];
    return rest == 0u && v[12] == 0u;
}
__host__ __device__ inline Fq29 fq_inv_binary(const Fq29 &a);
__host__ __device__ inline Fq29 fq_inv(const Fq29 &a)
{
    uint32_t y[12], x[12];
    fq_pack(fq_reduce(a), y);
    if (fq_inv_words31(y, x)) return fq_mul(fq_unpack(x), fq_const(FQ29_R3));
    return fq_inv_binary(a);
}
__host__ __device__ inline Fq29 fq_inv_binary(const Fq29 &a)
{
    uint32_t u[12], v[12], x1[12], x2[12];
    fq_pack(fq_reduce(a), u);
#pragma unroll
    for (int k = 0; k < 12; k++) { v[k] = Q32_[k]; x1[k] = k == 0 ? 1u : 0u; x2[k] = 0u; }
#pragma nounroll
    for (int round = 0; round < 2 * 382; round++) {
        uint32_t any = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) any |= u[k];
        // a lane whose u has reached 0 changes neither v nor x2 any more (u is even, nothing is exchanged): the loop may run on for
        // the other lanes of its wavefront -- and a wave-uniform exit costs no copies of the live values, a per-lane one did
#if defined(__HIP_DEVICE_COMPILE__)
        if (__builtin_amdgcn_ballot_w64(any != 0) == 0) break;
#else
        if (any == 0) break;
#endif
        // Everything below is mask arithmetic ((m & a) | (~m & b) is one v_bfi_b32): written with ?: the compiler made the exchange
        // a chain of predicated branches.  x1, x2 live in [0, q] (q itself stands for 0: the range is closed under the steps).
        const uint32_t odd = 0u - (u[0] & 1u);                           // all ones when u is odd
        uint32_t d[12], e[12], f[12];
        uint32_t borrow = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) d[k] = sub_b(u[k], v[k], borrow);   // u - v
        const uint32_t less = 0u - borrow, take = odd & less;            // u < v;  u odd and u < v: the pairs change roles
        borrow = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) e[k] = sub_b(x1[k], x2[k], borrow); // x1 - x2 (+ q if negative)
        const uint32_t below = 0u - borrow;
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) e[k] = add_c(e[k], Q32_[k] & below, carry);
        borrow = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) f[k] = sub_b(Q32_[k], e[k], borrow); // x2 - x1 = q - (x1 - x2)
        borrow = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) {
            const uint32_t neg = sub_b(d[k] ^ less, less, borrow);        // |u - v|: the two's complement when u < v
            const uint32_t tu = u[k], tx = x1[k];
            u[k] = (odd & neg) | (~odd & tu);
            v[k] = (take & tu) | (~take & v[k]);
            x1[k] = (odd & ((less & f[k]) | (~less & e[k]))) | (~odd & tx);
            x2[k] = (take & tx) | (~take & x2[k]);
        }
        // u /= 2 (even now), x1 /= 2 mod q: (x1 + q) / 2 when x1 is odd; x1 + q < 2^382
        const uint32_t xodd = 0u - (x1[0] & 1u);
        carry = 0;
#pragma unroll
        for (int k = 0; k < 12; k++) x1[k] = add_c(x1[k], Q32_[k] & xodd, carry);
#pragma unroll
        for (int k = 0; k < 12; k++) {
            u[k] = (u[k] >> 1) | (k + 1 < 12 ? u[k + 1] << 31 : 0u);
            x1[k] = (x1[k] >> 1) | (k + 1 < 12 ? x1[k + 1] << 31 : 0u);
        }
    }
    return fq_mul(fq_unpack(x2), fq_const(FQ29_R3));
}

// ---- Fq2 = Fq[u] / (u^2 + 1): the field of G2's coordinates ------------------------------------------------------------------------
// Components are lazily reduced Fq values.  A product leaves c0 < 6 q, c1 < 10 q (Karatsuba: three Fq products), a square
// c0 < 2 q, c1 < 4 q (two); operands may be anything below 2^10 q.
struct Fq2_29 { Fq29 c0, c1; };
__host__ __device__ __forceinline__ Fq2_29 fq2_mul(const Fq2_29 &a, const Fq2_29 &b)
{
    const Fq29 m0 = fq_mul(a.c0, b.c0), m1 = fq_mul(a.c1, b.c1), m2 = fq_mul(fq_add(a.c0, a.c1), fq_add(b.c0, b.c1));
    Fq2_29 r;
    r.c0 = fq_sub<4>(m0, m1);
    r.c1 = fq_sub<4>(fq_sub<4>(m2, m0), m1);
    return r;
}
__host__ __device__ __forceinline__ Fq2_29 fq2_sqr(const Fq2_29 &a)           // a.c1 < 1024 q
{
    Fq2_29 r;
    r.c0 = fq_mul(fq_add(a.c0, a.c1), fq_sub<1024>(a.c0, a.c1));
    const Fq29 t = fq_mul(a.c0, a.c1);
    r.c1 = fq_add(t, t);
    return r;
}

// ---- the two fields as policies of the point formulas below --------------------------------------------------------------------------
// K_MUL >= the bound (in q) of one product, K_2MUL of the sum of two, K_X / K_Y of a stored X / Y coordinate: what a
// difference adds so that it cannot go negative.
struct FqField {
    typedef Fq29 El;
    static constexpr uint32_t K_MUL = 4, K_2MUL = 4, K_X = 16, K_Y = 16;
    static constexpr uint32_t K_AFFINE = K_MUL;      // bound of a window-table row's coordinates (products of this policy)
    static constexpr int WORDS = NLQ, ARK_WORDS = 12, LANES = 1;
    __host__ __device__ static __forceinline__ El mul(const El &a, const El &b) { return fq_mul(a, b); }
    __host__ __device__ static __forceinline__ El sqr(const El &a) { return fq_sqr(a); }
    // a b - c d (c < KC q), one reduction
    template <uint32_t KC> __host__ __device__ static __forceinline__ El mul_sub(const El &a, const El &b, const El &c, const El &d) { return fq_mul_sub<KC>(a, b, c, d); }
    __host__ __device__ static __forceinline__ El add(const El &a, const El &b) { return fq_add(a, b); }
    template <uint32_t K> __host__ __device__ static __forceinline__ El sub(const El &a, const El &b) { return fq_sub<K>(a, b); }
    template <uint32_t K> __host__ __device__ static __forceinline__ El neg(const El &a) { return fq_neg<K>(a); }
    __host__ __device__ static __forceinline__ El zero() { return fq_zero(); }
    __host__ __device__ static __forceinline__ El one() { return fq_const(FQ29_ONE); }
    __host__ __device__ static __forceinline__ bool is_zero(const El &a) { return fq_is_zero(a); }
    __host__ __device__ static inline El inv(const El &a) { return fq_inv(a); }
    __host__ __device__ static __forceinline__ El from_ark(const uint32_t *w) { return fq_from_ark(w); }
    __host__ __device__ static __forceinline__ void to_ark(const El &a, uint32_t *w) { fq_to_ark(a, w); }
    __host__ __device__ static __forceinline__ void store(const El &a, uint32_t *w)
    {
#pragma unroll
        for (int k = 0; k < NLQ; k++) w[k] = a.l[k];
    }
    __host__ __device__ static __forceinline__ El load(const uint32_t *w)
    {
        El a;
#pragma unroll
        for (int k = 0; k < NLQ; k++) a.l[k] = w[k];
        return a;
    }
};
struct Fq2Field {
    typedef Fq2_29 El;
    static constexpr uint32_t K_MUL = 16, K_2MUL = 64, K_X = 256, K_Y = 64;
    static constexpr uint32_t K_AFFINE = K_MUL;
    static constexpr int WORDS = 2 * NLQ, ARK_WORDS = 24, LANES = 1;
    __host__ __device__ static __forceinline__ El mul(const El &a, const El &b) { return fq2_mul(a, b); }
    __host__ __device__ static __forceinline__ El sqr(const El &a) { return fq2_sqr(a); }
    template <uint32_t KC> __host__ __device__ static __forceinline__ El mul_sub(const El &a, const El &b, const El &c, const El &d)
    {
        return sub<K_MUL>(fq2_mul(a, b), fq2_mul(c, d));
    }
    __host__ __device__ static __forceinline__ El add(const El &a, const El &b) { El r; r.c0 = fq_add(a.c0, b.c0); r.c1 = fq_add(a.c1, b.c1); return r; }
    template <uint32_t K> __host__ __device__ static __forceinline__ El sub(const El &a, const El &b)
    {
        El r; r.c0 = fq_sub<K>(a.c0, b.c0); r.c1 = fq_sub<K>(a.c1, b.c1); return r;
    }
    template <uint32_t K> __host__ __device__ static __forceinline__ El neg(const El &a) { El r; r.c0 = fq_neg<K>(a.c0); r.c1 = fq_neg<K>(a.c1); return r; }
    __host__ __device__ static __forceinline__ El zero() { El r; r.c0 = r.c1 = fq_zero(); return r; }
    __host__ __device__ static __forceinline__ El one() { El r; r.c0 = fq_const(FQ29_ONE); r.c1 = fq_zero(); return r; }
    __host__ __device__ static __forceinline__ bool is_zero(const El &a) { return fq_is_zero(a.c0) && fq_is_zero(a.c1); }
    __host__ __device__ static inline El inv(const El &a)                         // conj(a) / (c0^2 + c1^2)
    {
        const Fq29 n = fq_inv(fq_add(fq_mul(a.c0, a.c0), fq_mul(a.c1, a.c1)));
        El r; r.c0 = fq_mul(a.c0, n); r.c1 = fq_mul(fq_neg<1024>(a.c1), n); return r;
    }
    __host__ __device__ static __forceinline__ El from_ark(const uint32_t *w) { El r; r.c0 = fq_from_ark(w); r.c1 = fq_from_ark(w + 12); return r; }
    __host__ __device__ static __forceinline__ void to_ark(const El &a, uint32_t *w) { fq_to_ark(a.c0, w); fq_to_ark(a.c1, w + 12); }
    __host__ __device__ static __forceinline__ void store(const El &a, uint32_t *w) { FqField::store(a.c0, w); FqField::store(a.c1, w + NLQ); }
    __host__ __device__ static __forceinline__ El load(const uint32_t *w) { El r; r.c0 = FqField::load(w); r.c1 = FqField::load(w + NLQ); return r; }
};

#if defined(__HIPCC__)
// Fq2 with the two components in two ADJACENT LANES of a wavefront (even lane: c0, odd lane: c1) -- device only.  A G2 point in
// XYZZ coordinates is 112 registers with both components in one lane, its mixed addition twice what a lane has (the kernels of
// Fq2Field run with 1.4 - 3.8 KB of scratch per lane); here every lane holds one Fq value per coordinate, exactly what a G1
// lane holds, and the cross terms of a product come from the neighbour by DPP (quad_perm [1, 0, 3, 2]: no LDS, no extra wait):
//     (a0 + a1 u)(b0 + b1 u) = (a0 b0 - a1 b1) + (a0 b1 + a1 b0) u      even lane: a b + (-a') b'      odd lane: a' b + a b'
// -- TWO products and ONE reduction per lane (columns add: frw_fq29.h fq_mul_cols), i.e. the same 4 Fq products per Fq2 product
// over the pair that Karatsuba's 3 + its extra additions cost one lane, but no spills; a square is ONE product per lane:
//     (a0 + a1)(a0 - a1) in the even lane, (2 a1) a0 in the odd one.
// Both lanes of a pair must take the same branches (they do wherever a branch depends on F::is_zero or a point's `inf`, which
// are pair-wide by construction).  Results of products are < 2 q per component; operands may be anything below 2^10 q.
struct Fq2Half { Fq29 v; };
__device__ __forceinline__ uint32_t pair_swap_u32(uint32_t x)
{
    return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0xB1 /* quad_perm [1, 0, 3, 2] */, 0xf, 0xf, true);
}
__device__ __forceinline__ Fq29 pair_swap(const Fq29 &a)
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = pair_swap_u32(a.l[i]);
    return r;
}
__device__ __forceinline__ Fq29 fq_select(bool take_a, const Fq29 &a, const Fq29 &b)
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = take_a ? a.l[i] : b.l[i];
    return r;
}
struct Fq2PairField {
    typedef Fq2Half El;
    static constexpr uint32_t K_MUL = 4, K_2MUL = 16, K_X = 256, K_Y = 64;     // a product < 2 q per component; X, Y as stored by either G2 policy
    static constexpr uint32_t K_AFFINE = 16;         // the window tables are written by Fq2Field kernels: rows < 10 q per component
    static constexpr int WORDS = 2 * NLQ, ARK_WORDS = 24, LANES = 2;
    __device__ static __forceinline__ bool odd() { return (threadIdx.x & 1u) != 0; }
    __device__ static __forceinline__ El mul(const El &a, const El &b)
    {
        const Fq29 ap = pair_swap(a.v), bp = pair_swap(b.v);
        const bool o = odd();
        // even: a b + (-a') b';  odd: a' b + a b'
        const Fq29 x = fq_select(o, ap, a.v), y = fq_select(o, a.v, fq_neg<1024>(ap));
        uint64_t col[2 * NLQ];
        fq_mul2_cols(x, b.v, y, bp, col);
        El r;
        r.v = fq_reduce_cols(col);
        return r;
    }
    __device__ static __forceinline__ El sqr(const El &a)
    {
        const Fq29 ap = pair_swap(a.v);
        const bool o = odd();
        // even: (a0 + a1)(a0 - a1);  odd: (a1 + a1) a0
        const Fq29 x = fq_add(a.v, fq_select(o, a.v, ap)), y = fq_select(o, ap, fq_sub<1024>(a.v, ap));
        El r;
        r.v = fq_mul(x, y);
        return r;
    }
    template <uint32_t KC> __device__ static __forceinline__ El mul_sub(const El &a, const El &b, const El &c, const El &d)
    {
        return sub<K_MUL>(mul(a, b), mul(c, d));
    }
    __device__ static __forceinline__ El add(const El &a, const El &b) { El r; r.v = fq_add(a.v, b.v); return r; }
    template <uint32_t K> __device__ static __forceinline__ El sub(const El &a, const El &b) { El r; r.v = fq_sub<K>(a.v, b.v); return r; }
    template <uint32_t K> __device__ static __forceinline__ El neg(const El &a) { El r; r.v = fq_neg<K>(a.v); return r; }
    __device__ static __forceinline__ El zero() { El r; r.v = fq_zero(); return r; }
    __device__ static __forceinline__ El one() { El r; r.v = fq_select(odd(), fq_zero(), fq_const(FQ29_ONE)); return r; }
    __device__ static __forceinline__ bool is_zero(const El &a)                  // of the Fq2 element: the same answer in both lanes
    {
        const uint32_t mine = fq_is_zero(a.v) ? 1u : 0u;
        return (mine & pair_swap_u32(mine)) != 0;
    }
    __device__ static inline El inv(const El &a)                                  // conj(a) / (c0^2 + c1^2)
    {
        const Fq29 sq = fq_sqr(a.v);
        const Fq29 n = fq_inv(fq_add(sq, pair_swap(sq)));
        El r;
        r.v = fq_mul(fq_select(odd(), fq_neg<1024>(a.v), a.v), n);
        return r;
    }
    __device__ static __forceinline__ El from_ark(const uint32_t *w) { El r; r.v = fq_from_ark(w + (odd() ? 12 : 0)); return r; }
    __device__ static __forceinline__ void to_ark(const El &a, uint32_t *w) { fq_to_ark(a.v, w + (odd() ? 12 : 0)); }
    __device__ static __forceinline__ void store(const El &a, uint32_t *w) { FqField::store(a.v, w + (odd() ? NLQ : 0)); }
    __device__ static __forceinline__ El load(const uint32_t *w) { El r; r.v = FqField::load(w + (odd() ? NLQ : 0)); return r; }
};
#endif

// the published generators (ark-bls12-381 g1.rs / g2.rs), x 2^406; tests/test_fq29_host.py re-derives the limbs
constexpr LimbsQ G1_GEN_X29 = {{0x0af58fd1u, 0x1662a68eu, 0x07d2c530u, 0x08993c24u, 0x1e4f4756u, 0x0c5f7ae2u, 0x0f589991u,
                                0x00dc570eu, 0x121b54f5u, 0x05242b28u, 0x17442464u, 0x0ee8a0cdu, 0x1f591099u, 0x00000009u}};
constexpr LimbsQ G1_GEN_Y29 = {{0x1cd65f60u, 0x1919ce91u, 0x0da44145u, 0x0810b2ccu, 0x08f38c44u, 0x00629241u, 0x0f521d8cu,
                                0x0529cbadu, 0x0506077fu, 0x191b2712u, 0x0c587ccbu, 0x1017da2fu, 0x1c72eafau, 0x0000000bu}};
constexpr LimbsQ G2_GEN_X0_29 = {{0x085d7b9eu, 0x0d352838u, 0x124c5276u, 0x1748cfe0u, 0x10081802u, 0x0bfc6540u, 0x01ac1784u,
                                   0x11aab37fu, 0x08f338c1u, 0x19443e06u, 0x1cc29975u, 0x0a0da681u, 0x0a53e408u, 0x00000004u}};
constexpr LimbsQ G2_GEN_X1_29 = {{0x0ba695cbu, 0x1034485fu, 0x124f6eb5u, 0x0778336bu, 0x1e170b2du, 0x12164b09u, 0x1d34eb65u,
                                   0x10a72ef2u, 0x07b96a4eu, 0x194de314u, 0x0fd4fa4au, 0x0178f021u, 0x10455591u, 0x0000000bu}};
constexpr LimbsQ G2_GEN_Y0_29 = {{0x1ec793b8u, 0x1780929bu, 0x0cdd65c4u, 0x126ca64fu, 0x0f9f2f6bu, 0x09485ce0u, 0x0621fd94u,
                                   0x1e0e7efdu, 0x09ba1f86u, 0x06d0a458u, 0x1f790335u, 0x0af83757u, 0x00c11d05u, 0x0000000cu}};
constexpr LimbsQ G2_GEN_Y1_29 = {{0x014e8093u, 0x07f6c9aau, 0x19bd3883u, 0x156ca3e1u, 0x0e34898fu, 0x0840b6feu, 0x0594f57du,
                                   0x149969ecu, 0x12886b3eu, 0x1cbcbe34u, 0x1cae2f62u, 0x08f5e851u, 0x187a6c6cu, 0x00000003u}};

// G1's endomorphism phi(x, y) = (beta x, y) = lambda (x, y) with lambda = z^2 - 1 (128 bits; the group order is
// lambda^2 + lambda + 1): beta is the cube root of unity in Fq that goes with THIS lambda (the other root goes with lambda^2),
// x 2^406; tests/test_fq29_host.py re-derives it and checks phi(G) = lambda G in Python integers.  A scalar k < r splits as
// k = k0 + lambda k1 by plain division, k0 < lambda, k1 <= lambda + 1: two 128-bit halves, half the doublings of k P.
constexpr LimbsQ G1_ENDO_BETA29 = {{0x1195dfebu, 0x1b04e484u, 0x06026044u, 0x086070a2u, 0x1fd68858u, 0x137e9670u, 0x06871e67u,
                                    0x1e736664u, 0x083b24f6u, 0x08a70373u, 0x02a012fdu, 0x0112f94bu, 0x18a2733cu, 0x00000003u}};

// ---- short Weierstrass curves with a = 0 in XYZZ coordinates (x = X / ZZ, y = Y / ZZZ, ZZ^3 = ZZZ^2) over either field:
// G1: y^2 = x^3 + 4 over Fq, G2: y^2 = x^3 + 4 (1 + u) over Fq2 (the formulas never see b).  inf = the point at infinity.
// Bounds kept by every formula (units of q per component): Fq: X < 10, Y < 6, ZZ, ZZZ < 2; Fq2: X < 90, Y < 26, ZZ, ZZZ < 10.
template <class F> struct AffineT { typename F::El x, y; bool inf; };
template <class F> struct XyzzT { typename F::El x, y, zz, zzz; bool inf; };

template <class F> __host__ __device__ __forceinline__ XyzzT<F> pt_identity()
{
    XyzzT<F> r;
    r.x = r.y = r.zz = r.zzz = F::zero();
    r.inf = true;
    return r;
}
template <class F> __host__ __device__ __forceinline__ XyzzT<F> pt_from_affine(const AffineT<F> &p)
{
    XyzzT<F> r;
    r.x = p.x; r.y = p.y; r.zz = r.zzz = F::one();
    r.inf = p.inf;
    return r;
}
// dbl-2008-s-1 (a = 0): 6 M + 3 S
template <class F> __host__ __device__ inline XyzzT<F> pt_double(const XyzzT<F> &p)
{
    if (p.inf) return p;
    const auto u = F::add(p.y, p.y);
    const auto v = F::sqr(u), w = F::mul(u, v);
    const auto s = F::mul(p.x, v);
    const auto xx = F::sqr(p.x);
    const auto m = F::add(F::add(xx, xx), xx);
    XyzzT<F> r;
    r.x = F::template sub<F::K_2MUL>(F::sqr(m), F::add(s, s));
    r.y = F::template mul_sub<F::K_MUL>(m, F::template sub<F::K_X>(s, r.x), w, p.y);
    r.zz = F::mul(v, p.zz);
    r.zzz = F::mul(w, p.zzz);
    r.inf = false;
    return r;
}
// madd-2008-s: 8 M + 2 S; complete (an accumulator that meets its own value doubles, its negative cancels)
template <class F> __host__ __device__ __forceinline__ XyzzT<F> pt_add_affine(const XyzzT<F> &p, const AffineT<F> &q)
{
    if (q.inf) return p;
    if (p.inf) return pt_from_affine(q);
    const auto u2 = F::mul(q.x, p.zz), s2 = F::mul(q.y, p.zzz);
    const auto pp_ = F::template sub<F::K_X>(u2, p.x);
    const auto rr = F::template sub<F::K_Y>(s2, p.y);
    if (F::is_zero(pp_)) {
        if (F::is_zero(rr)) return pt_double(pt_from_affine(q));
        return pt_identity<F>();
    }
    const auto pp = F::sqr(pp_), ppp = F::mul(pp_, pp), qq = F::mul(p.x, pp);
    XyzzT<F> r;
    r.x = F::template sub<F::K_2MUL>(F::template sub<F::K_MUL>(F::sqr(rr), ppp), F::add(qq, qq));
    r.y = F::template mul_sub<F::K_Y>(rr, F::template sub<F::K_X>(qq, r.x), p.y, ppp);
    r.zz = F::mul(p.zz, pp);
    r.zzz = F::mul(p.zzz, ppp);
    r.inf = false;
    return r;
}
// add-2008-s: 12 M + 2 S; complete
template <class F> __host__ __device__ inline XyzzT<F> pt_add(const XyzzT<F> &p, const XyzzT<F> &q)
{
    if (q.inf) return p;
    if (p.inf) return q;
    const auto u1 = F::mul(p.x, q.zz), u2 = F::mul(q.x, p.zz), s1 = F::mul(p.y, q.zzz), s2 = F::mul(q.y, p.zzz);
    const auto pp_ = F::template sub<F::K_MUL>(u2, u1), rr = F::template sub<F::K_MUL>(s2, s1);
    if (F::is_zero(pp_)) {
        if (F::is_zero(rr)) return pt_double(p);
        return pt_identity<F>();
    }
    const auto pp = F::sqr(pp_), ppp = F::mul(pp_, pp), qq = F::mul(u1, pp);
    XyzzT<F> r;
    r.x = F::template sub<F::K_2MUL>(F::template sub<F::K_MUL>(F::sqr(rr), ppp), F::add(qq, qq));
    r.y = F::template mul_sub<F::K_MUL>(rr, F::template sub<F::K_X>(qq, r.x), s1, ppp);
    r.zz = F::mul(F::mul(p.zz, q.zz), pp);
    r.zzz = F::mul(F::mul(p.zzz, q.zzz), ppp);
    r.inf = false;
    return r;
}
// x = X / ZZ, y = Y / ZZZ with one inversion; (0, 0) for the point at infinity
template <class F> __host__ __device__ inline AffineT<F> pt_to_affine(const XyzzT<F> &p)
{
    AffineT<F> r;
    r.inf = p.inf;
    if (p.inf) { r.x = r.y = F::zero(); return r; }
    const auto inv = F::inv(F::mul(p.zz, p.zzz));
    r.x = F::mul(p.x, F::mul(inv, p.zzz));
    r.y = F::mul(p.y, F::mul(inv, p.zz));
    return r;
}

// G1 under its round-3 names
typedef AffineT<FqField> G1Affine29;
typedef XyzzT<FqField> G1Xyzz;
typedef AffineT<Fq2Field> G2Affine29;
typedef XyzzT<Fq2Field> G2Xyzz;
__host__ __device__ __forceinline__ G1Xyzz g1_identity() { return pt_identity<FqField>(); }
__host__ __device__ __forceinline__ G1Xyzz g1_from_affine(const G1Affine29 &p) { return pt_from_affine(p); }
__host__ __device__ inline G1Xyzz g1_double(const G1Xyzz &p) { return pt_double(p); }
__host__ __device__ inline G1Xyzz g1_add_affine(const G1Xyzz &p, const G1Affine29 &q) { return pt_add_affine(p, q); }
__host__ __device__ inline G1Xyzz g1_add(const G1Xyzz &p, const G1Xyzz &q) { return pt_add(p, q); }
__host__ __device__ inline G1Affine29 g1_to_affine(const G1Xyzz &p) { return pt_to_affine(p); }

}  // namespace frw
