// frw_fr29.h -- BLS12-381 Fr on the device in nine 29-bit limbs (redundant: 261 bits for a 255-bit modulus).
//
// Why not the 8 x 32-bit form of frw_fr.h: v_mad_u64_u32 issues at the full VALU rate on gfx950
// (profiles/r02_valu_rates.txt), so a Montgomery product is bound by everything AROUND the multiplies -- with 32-bit
// limbs every partial product needs its 33-bit carry folded in by hand (the compiler spends ~4.5 instructions per
// multiply on it).  With 29-bit limbs a column of the product (<= 9 + 9 partial products < 2^64) accumulates in one
// 64-bit register pair through chained v_mad_u64_u32 alone: ~210 instructions per product instead of ~770.
//
// Representation: x = sum l[i] 2^(29 i), limbs normalised (< 2^29) unless said otherwise, value NOT necessarily < p:
// the transforms carry values up to ~60 p (2^261 = 70.4 p) and reduce only through the products.
// Montgomery radix R' = 2^261.  Data stays in ark-ff's form x R (R = 2^256) throughout: the tables (twiddles, scale
// factors) are stored as w R', and mul29(x R, w R') = x w R.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frw_fr.h"

namespace frw {

constexpr int NL29 = 9;
constexpr uint32_t M29 = (1u << 29) - 1u;

struct F29 { uint32_t l[NL29]; };

// limb i of a 256-bit constant given as 8 x u32
constexpr uint32_t limb29_of(const uint32_t (&w)[8], int i)
{
    const int bit = 29 * i, k = bit >> 5, s = bit & 31;
    const uint64_t lo = w[k], hi = k + 1 < 8 ? w[k + 1] : 0;
    return (uint32_t)(((hi << 32 | lo) >> s) & M29);
}
constexpr uint32_t P32_[8] = FRW_P32;
constexpr uint32_t p29(int i) { return limb29_of(P32_, i); }
// 2 p (< 2^256)
constexpr uint32_t twice_p_word(int k)
{
    return (P32_[k] << 1) | (k ? P32_[k - 1] >> 31 : 0u);
}
constexpr uint32_t P2_32_[8] = {twice_p_word(0), twice_p_word(1), twice_p_word(2), twice_p_word(3),
                                twice_p_word(4), twice_p_word(5), twice_p_word(6), twice_p_word(7)};
constexpr uint32_t p2_29(int i) { return limb29_of(P2_32_, i); }
// the limbs of K p (K <= 32: K p < 2^260) as a constant table (a constexpr FUNCTION of the loop index is not folded in
// device code: it was evaluated at run time through scratch memory)
struct Limbs29 { uint32_t l[NL29]; };
constexpr Limbs29 make_kp29(uint32_t K)
{
    uint32_t w[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t carry = 0;
    for (int k = 0; k < 8; k++) {
        const uint64_t t = (uint64_t)K * P32_[k] + carry;
        w[k] = (uint32_t)t;
        carry = t >> 32;
    }
    w[8] = (uint32_t)carry;
    Limbs29 r{};
    for (int i = 0; i < NL29; i++) {
        const int bit = 29 * i, k = bit >> 5, s = bit & 31;
        const uint64_t both = (uint64_t)w[k] | ((uint64_t)w[k + 1] << 32);
        r.l[i] = (uint32_t)(both >> s) & (i == NL29 - 1 ? 0xffffffffu : M29);
    }
    return r;
}
constexpr Limbs29 KP29_1 = make_kp29(1), KP29_2 = make_kp29(2), KP29_4 = make_kp29(4), KP29_8 = make_kp29(8),
                  KP29_16 = make_kp29(16), KP29_32 = make_kp29(32);
template <uint32_t K> constexpr const Limbs29 &kp29_table()
{
    static_assert(K == 1 || K == 2 || K == 4 || K == 8 || K == 16 || K == 32, "table missing");
    return K == 1 ? KP29_1 : K == 2 ? KP29_2 : K == 4 ? KP29_4 : K == 8 ? KP29_8 : K == 16 ? KP29_16 : KP29_32;
}
static_assert(KP29_2.l[0] == p2_29(0) && KP29_2.l[8] == p2_29(8) && KP29_1.l[3] == p29(3), "K p limbs");
static_assert(p29(0) == 1u, "p = 1 mod 2^29: the Montgomery factor -p^-1 mod 2^29 is 2^29 - 1");

__device__ __forceinline__ F29 f29_unpack(const Fr8 &w)
{
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) {
        const int bit = 29 * i, k = bit >> 5, s = bit & 31;
        const uint64_t both = (uint64_t)w.l[k] | (k + 1 < 8 ? (uint64_t)w.l[k + 1] << 32 : 0ull);
        r.l[i] = (uint32_t)(both >> s) & M29;
    }
    return r;
}

// normalised limbs, value < 2^256
__device__ __forceinline__ Fr8 f29_pack(const F29 &a)
{
    Fr8 r;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        const int bit = 32 * k, i = bit / 29, s = bit - 29 * i;
        uint32_t w = a.l[i] >> s;
        int filled = 29 - s;
        if (filled < 32 && i + 1 < NL29) { w |= a.l[i + 1] << filled; filled += 29; }
        if (filled < 32 && i + 2 < NL29) w |= a.l[i + 2] << filled;
        r.l[k] = w;
    }
    return r;
}

// carry propagation; limbs in: unsigned, < 2^32 - 16
__device__ __forceinline__ void f29_normalise(F29 &a)
{
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL29 - 1; i++) {
        const uint32_t x = a.l[i] + c;
        a.l[i] = x & M29;
        c = x >> 29;
    }
    a.l[NL29 - 1] += c;
}

// carry propagation of limbs that are signed 32-bit differences; the value itself must be >= 0.  Returns the sign of the
// value instead when `sign` is given (top limb negative), leaving the limbs as they came out.
__device__ __forceinline__ int f29_normalise_signed(F29 &a)
{
    int32_t c = 0;
#pragma unroll
    for (int i = 0; i < NL29 - 1; i++) {
        const int32_t x = (int32_t)a.l[i] + c;
        a.l[i] = (uint32_t)x & M29;
        c = x >> 29;
    }
    const int32_t top = (int32_t)a.l[NL29 - 1] + c;
    a.l[NL29 - 1] = (uint32_t)top;
    return top < 0;
}

// a + b, limb-wise, then normalised
__device__ __forceinline__ F29 f29_add(const F29 &a, const F29 &b)
{
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] + b.l[i];
    f29_normalise(r);
    return r;
}

// a - b + 2 p  (b < 2 p), normalised
__device__ __forceinline__ F29 f29_sub_2p(const F29 &a, const F29 &b)
{
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] - b.l[i] + p2_29(i);
    (void)f29_normalise_signed(r);
    return r;
}

// a - b + K p  (b < K p), normalised
template <uint32_t K>
__device__ __forceinline__ F29 f29_sub_kp(const F29 &a, const F29 &b)
{
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] - b.l[i] + kp29_table<K>().l[i];
    (void)f29_normalise_signed(r);
    return r;
}

// a < 4 p  ->  a or a - 2 p, < 2 p
__device__ __forceinline__ F29 f29_reduce_4p(const F29 &a)
{
    F29 d;
#pragma unroll
    for (int i = 0; i < NL29; i++) d.l[i] = a.l[i] - p2_29(i);
    const int neg = f29_normalise_signed(d);
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = neg ? a.l[i] : d.l[i];
    return r;
}

// a < 2 K p  ->  a or a - K p, < K p
template <uint32_t K>
__device__ __forceinline__ F29 f29_cond_sub_kp(const F29 &a)
{
    F29 d;
#pragma unroll
    for (int i = 0; i < NL29; i++) d.l[i] = a.l[i] - kp29_table<K>().l[i];
    const int neg = f29_normalise_signed(d);
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = neg ? a.l[i] : d.l[i];
    return r;
}

// a < 2 p  ->  the canonical representative < p
__device__ __forceinline__ F29 f29_canonical(const F29 &a)
{
    F29 d;
#pragma unroll
    for (int i = 0; i < NL29; i++) d.l[i] = a.l[i] - p29(i);
    const int neg = f29_normalise_signed(d);
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = neg ? a.l[i] : d.l[i];
    return r;
}

// Montgomery product a b / 2^261 mod p, columns in 64 bits.
// a: limbs < 2^31 (need not be normalised); b: normalised limbs.  Result: normalised, < a b / 2^261 + p
// (< 2 p whenever a b < 2^261 p, e.g. b < p and a < 2^261).
// Shaped for instruction-level parallelism: v_mad_u64_u32 issues every ~5 cycles but a dependent one waits several times
// that, and the transforms run two wavefronts per SIMD.  So: first all 81 partial products into their 17 columns (17
// independent chains), then the reduction by operand scanning -- each m[i] feeds eight independent multiply-adds
// (m[i] p[j] into column i + j); the only serial thread is carry -> m[i] -> column i + 1.  (The first form accumulated one
// column at a time, 18 dependent multiply-adds in a row: 13 cycles per multiply-add measured, profiles/r02_qap_v3_ab.txt.)
// the Montgomery reduction of 18 columns (column k has weight 2^(29 k)): sum / 2^261 mod p, normalised, < sum / 2^261 + p
__device__ __forceinline__ F29 f29_reduce_columns(uint64_t (&col)[2 * NL29])
{
#pragma unroll
    for (int i = 0; i < NL29; i++) {
        const uint32_t m = (0u - (uint32_t)col[i]) & M29;          // -p^-1 = -1 mod 2^29
#pragma unroll
        for (int j = 1; j < NL29; j++) col[i + j] += (uint64_t)m * p29(j);
        col[i + 1] += (col[i] + m) >> 29;                           // m p29(0) = m clears the low 29 bits of column i
    }
    F29 r;
#pragma unroll
    for (int k = NL29; k < 2 * NL29 - 1; k++) {
        r.l[k - NL29] = (uint32_t)col[k] & M29;
        col[k + 1] += col[k] >> 29;
    }
    r.l[NL29 - 1] = (uint32_t)col[2 * NL29 - 1];
    return r;
}

__device__ __forceinline__ F29 f29_mul(const F29 &a, const F29 &b)
{
    uint64_t col[2 * NL29];
#pragma unroll
    for (int k = 0; k < 2 * NL29 - 1; k++) {
        uint64_t acc = 0;
#pragma unroll
        for (int i = (k < NL29 ? 0 : k - (NL29 - 1)); i <= (k < NL29 ? k : NL29 - 1); i++) acc += (uint64_t)a.l[i] * b.l[k - i];
        col[k] = acc;
    }
    col[2 * NL29 - 1] = 0;
    return f29_reduce_columns(col);
}

// x / 2^261 mod p for an integer x < 2^261 p given in 29-bit limbs (x[10] may hold the excess), < 2 p
template <int N>
__device__ __forceinline__ F29 f29_redc_wide(const uint32_t (&x)[N])
{
    static_assert(N <= 2 * NL29, "too wide");
    uint64_t col[2 * NL29];
#pragma unroll
    for (int k = 0; k < 2 * NL29; k++) col[k] = k < N ? x[k] : 0;
    return f29_reduce_columns(col);
}

}  // namespace frw
