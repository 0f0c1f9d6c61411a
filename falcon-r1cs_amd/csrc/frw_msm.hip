// frw_msm.hip -- multi-scalar multiplication over BLS12-381 G1 on the device: the step after the QAP witness map in a Groth16
// prover.  examples/pok_sig.rs:30-47 of the reference calls Groth16::<Bls12_381>::prove; ark-groth16 0.3.0 prover.rs
// (create_proof_with_reduction_and_matrices) then computes
//     h_acc = VariableBaseMSM::multi_scalar_mul(&pk.h_query, &h_assignment)          2^18 - 1 points for Falcon-1024
// (and the same over a_query / b_g1_query / l_query with the witness as scalars).  frw_qap_witness_map_dev leaves h in HBM;
// this file consumes it there.
//
// MI355X-first shape.  The bases belong to the proving key: fixed per circuit, shared by every signature ever proved.  So
// (1) frw_msm_g1_load spends 288 GB-class memory on them once: the table holds 2^(16 j) P_i for all sixteen 16-bit windows
//     j (436 MB for 2^18 points, affine, fourteen 29-bit limbs per coordinate), which turns a 255-bit MSM over n points
//     into ONE bucket accumulation over 16 n points with 16-bit signed digits -- no doublings, no per-window passes;
// (2) a call handles a batch of signatures: per signature the 16 n (digit, point) pairs are counting-sorted by bucket
//     (32,768 buckets, ~128 entries each), one thread per bucket adds its entries up in registers (XYZZ coordinates, mixed
//     additions, 8 M + 2 S), gathering the table rows it needs; 512 threads fold the buckets (sum of (b + 1) B_b by running
//     sums over 64-bucket chunks, small multiples, a tree through LDS), and the result leaves as one affine point in
//     ark-ff's bytes.
// Group addition is associative and commutative, so the result does not depend on the order the atomics of the sort
// happened to produce: the output is the unique affine point, bit for bit what any other schedule (or the CPU) gives.
// Field arithmetic: frw_fq29.h.  Bound: vector-ALU issue (a mixed addition is ~5,700 vector instructions), not HBM (a
// signature reads 16 n x 112 B = 0.47 GB of table rows).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <new>
#include <mutex>
#include <vector>
#include "../../include/frw.h"
#include "frw_device.h"
#include "frw_fq29.h"
#include "frw_quad.h"
#include "frw_fr29.h"

namespace frw {

constexpr int MSM_C = 16;                       // window bits
constexpr int MSM_W = 16;                       // windows: 16 x 16 = 256 >= 255 bits
constexpr int MSM_BUCKETS = 1 << (MSM_C - 1);   // signed digits: |d| in 1 .. 2^15
constexpr int MSM_FOLD1_THREADS = 4096;         // partial sums per signature at most (first stage: 8, 32 or 64 buckets per thread)
constexpr int MSM_FOLD_THREADS = 512;           // second stage: one workgroup per signature
// per group (F = FqField: G1, Fq2Field: G2): a table row = x, y limbs (all zero = the point at infinity); a bucket = X, Y, ZZ,
// ZZZ limbs + the infinity flag (padded to 16 bytes); ark-ff's bytes of an affine point
template <class F> struct Grp {
    static constexpr int PT_WORDS = 2 * F::WORDS, BK_WORDS = 4 * F::WORDS + 4, ARK_WORDS = 2 * F::ARK_WORDS;
    static constexpr uint32_t K_AFFINE_Y = F::K_AFFINE;       // bound of a table row's y: what its negation adds
};

struct MsmDev {
    uint32_t n;                 // points
    const uint32_t *table;      // [MSM_W][n][PT_WORDS]
    // narrow handles of up to 2^18 points: sum_(j in m) P_(8 g + j) for every group g of eight consecutive points and every non-empty
    // subset m of it, affine: [ceil(n / 8)][255][PT_WORDS] (as many bytes per point as the window table).  A Falcon witness is 45 % ones
    // in long runs of booleans (bit decompositions, gate outputs): with the pattern of ones of a group as ONE byte, the points whose
    // scalar is one cost one addition per group of eight instead of 3.6.  null: the ones are summed point by point from a list.
    const uint32_t *ones_table;
};

// The narrow sums of ONE sort over up to three window tables in one launch: a proof's a_query, b_g1_query and l_query sums take the
// same scalars, so their digits are sorted once -- and with the tables side by side in the grid the three sums are ONE chain of
// kernels instead of three (a proof made alone waits for the longest chain of small kernels, not for any one of them).  Grid row
// `slot` = table * sigs + signature: the sort's arrays are indexed by the signature, a table's own partial sums and its result by the slot.
struct NmsmTables {
    uint32_t n, sigs;
    const uint32_t *table[3], *ones_table[3];
    __device__ __forceinline__ MsmDev of(uint32_t slot, size_t &sig) const
    {
        const uint32_t t = slot / sigs;
        sig = slot - t * sigs;
        MsmDev m;
        m.n = n;
        m.table = t == 0 ? table[0] : t == 1 ? table[1] : table[2];
        m.ones_table = t == 0 ? ones_table[0] : t == 1 ? ones_table[1] : ones_table[2];
        return m;
    }
};

template <class F> __device__ __forceinline__ AffineT<F> load_row(const uint32_t *row)
{
    constexpr int PW = Grp<F>::PT_WORDS;
    AffineT<F> p;
    uint32_t any = 0;
    const uint4 *v = (const uint4 *)row;
    uint32_t w[PW];
#pragma unroll
    for (int k = 0; k < PW / 4; k++) {
        const uint4 t = v[k];
        w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int k = 0; k < PW; k++) any |= w[k];
    p.x = F::load(w);
    p.y = F::load(w + F::WORDS);
    p.inf = any == 0;
    return p;
}
// the two-lane Fq2: every lane fetches its own component of x and y (56 bytes each, 8-byte aligned) and the pair agrees on `inf`
template <> __device__ __forceinline__ AffineT<Fq2PairField> load_row<Fq2PairField>(const uint32_t *row)
{
    AffineT<Fq2PairField> p;
    const uint32_t *mine = row + (Fq2PairField::odd() ? NLQ : 0);
    uint32_t any = 0;
#pragma unroll
    for (int k = 0; k < NLQ / 2; k++) {
        const uint2 a = *(const uint2 *)(mine + 2 * k), b = *(const uint2 *)(mine + 2 * NLQ + 2 * k);
        p.x.v.l[2 * k] = a.x; p.x.v.l[2 * k + 1] = a.y;
        p.y.v.l[2 * k] = b.x; p.y.v.l[2 * k + 1] = b.y;
        any |= a.x | a.y | b.x | b.y;
    }
    any |= pair_swap_u32(any);
    p.inf = any == 0;
    return p;
}
template <class F> __device__ __forceinline__ void store_row(uint32_t *row, const AffineT<F> &p)
{
    uint32_t w[Grp<F>::PT_WORDS];
    F::store(p.x, w);
    F::store(p.y, w + F::WORDS);
#pragma unroll
    for (int k = 0; k < Grp<F>::PT_WORDS; k++) row[k] = p.inf ? 0u : w[k];
}
// the two-lane Fq2: every lane writes its own component of x and of y
template <> __device__ __forceinline__ void store_row<Fq2PairField>(uint32_t *row, const AffineT<Fq2PairField> &p)
{
    uint32_t *mine = row + (Fq2PairField::odd() ? NLQ : 0);
#pragma unroll
    for (int k = 0; k < NLQ; k++) {
        mine[k] = p.inf ? 0u : p.x.v.l[k];
        mine[2 * NLQ + k] = p.inf ? 0u : p.y.v.l[k];
    }
}
template <class F> __device__ __forceinline__ void store_bucket(uint32_t *b, const XyzzT<F> &p)
{
    F::store(p.x, b); F::store(p.y, b + F::WORDS); F::store(p.zz, b + 2 * F::WORDS); F::store(p.zzz, b + 3 * F::WORDS);
    b[4 * F::WORDS] = p.inf ? 1u : 0u;
}
template <class F> __device__ __forceinline__ XyzzT<F> load_bucket(const uint32_t *b)
{
    XyzzT<F> p;
    p.x = F::load(b); p.y = F::load(b + F::WORDS); p.zz = F::load(b + 2 * F::WORDS); p.zzz = F::load(b + 3 * F::WORDS);
    p.inf = b[4 * F::WORDS] != 0;
    return p;
}
template <class F> __device__ __forceinline__ AffineT<F> load_ark_point(const uint32_t *w)
{
    AffineT<F> p;
    uint32_t any = 0;
    for (int k = 0; k < Grp<F>::ARK_WORDS; k++) any |= w[k];
    p.inf = any == 0;
    p.x = F::from_ark(w);
    p.y = F::from_ark(w + F::ARK_WORDS);
    return p;
}
template <class F> __device__ __forceinline__ void store_ark_point(uint32_t *o, const AffineT<F> &a)
{
    if (a.inf) {
        for (int k = 0; k < Grp<F>::ARK_WORDS; k++) o[k] = 0;
    } else {
        F::to_ark(a.x, o);
        F::to_ark(a.y, o + F::ARK_WORDS);
    }
}

// ---- load time: table[j][i] = 2^(16 j) P_i, affine, Montgomery limbs ----------------------------------------------------------
template <class F>
__global__ __launch_bounds__(64) void msm_precompute_kernel(uint32_t n, const uint32_t *__restrict__ bases /* [n][ARK_WORDS] ark-ff */,
                                                            uint32_t *__restrict__ table, int window_bits /* 16 or 8 */)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int windows = 256 / window_bits;
    AffineT<F> p = load_ark_point<F>(bases + (size_t)i * Grp<F>::ARK_WORDS);
    for (int j = 0; j < windows; j++) {
        store_row<F>(table + ((size_t)j * n + i) * Grp<F>::PT_WORDS, p);
        if (j + 1 == windows) break;
        XyzzT<F> d = pt_from_affine(p);
        for (int k = 0; k < window_bits; k++) d = pt_double(d);
        p = pt_to_affine(d);
    }
}

// the same from table ROWS (a bare handle's: what the device-side setup fills): a key made on the device grows its window tables here
template <class F>
__global__ __launch_bounds__(64) void msm_precompute_rows_kernel(uint32_t n, const uint32_t *__restrict__ rows /* [n][PT_WORDS] */,
                                                                 uint32_t *__restrict__ table, int window_bits /* 16 or 8 */)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    const int windows = 256 / window_bits;
    AffineT<F> p = load_row<F>(rows + (size_t)i * Grp<F>::PT_WORDS);
    for (int j = 0; j < windows; j++) {
        store_row<F>(table + ((size_t)j * n + i) * Grp<F>::PT_WORDS, p);
        if (j + 1 == windows) break;
        XyzzT<F> d = pt_from_affine(p);
        for (int k = 0; k < window_bits; k++) d = pt_double(d);
        p = pt_to_affine(d);
    }
}

// load time: ones_table[g][m - 1] = the sum of the points 8 g + j with bit j of m set (window 0 of the table holds the points)
template <class F>
__global__ __launch_bounds__(64) void msm_ones_table_kernel(uint32_t n, const uint32_t *__restrict__ table, uint32_t *__restrict__ ones_table)
{
    const size_t id = (size_t)blockIdx.x * 64 + threadIdx.x, g = id / 255;
    const uint32_t m = (uint32_t)(id % 255) + 1u;
    if (g >= ((size_t)n + 7) / 8) return;
    XyzzT<F> acc = pt_identity<F>();
    for (int j = 0; j < 8; j++) {
        const size_t i = g * 8 + j;
        if (((m >> j) & 1u) && i < n) acc = pt_add_affine(acc, load_row<F>(table + i * Grp<F>::PT_WORDS));
    }
    store_row<F>(ones_table + id * Grp<F>::PT_WORDS, pt_to_affine(acc));
}

// ---- the scalars: canonical integer, sixteen signed 16-bit digits ---------------------------------------------------------------
// d_j in [-2^15, 2^15] with sum d_j 2^(16 j) = k; the top digit takes the last carry (k < 2^255, so it stays <= 2^15)
// Returns true (and no digits) for the scalar 1: a Falcon witness is 45 % ones (the boolean elements that are set), which
// would all land in bucket 0 of window 0 -- they are summed by msm_ones_kernel instead.
// the scalar as a canonical integer below r
__device__ __forceinline__ Fr8 scalar_canonical(const uint32_t *src, int montgomery)
{
    Fr8 w = fr_load(src);
    if (montgomery) {
        // ark-ff's h R (R = 2^256) -> h: one product with 2^5 in R' = 2^261 arithmetic (frw_fr29.h), then the canonical form
        F29 c;
#pragma unroll
        for (int k = 0; k < NL29; k++) c.l[k] = k ? 0u : 32u;
        w = f29_pack(f29_canonical(f29_mul(f29_unpack(w), c)));
    } else {
        // "canonical" is the caller's promise; any 256-bit integer is brought below r here (2^256 < 4 r: two conditional
        // subtractions), because a top digit above 2^15 would index past the buckets -- and k P = (k mod r) P anyway
        w = f29_pack(f29_canonical(f29_reduce_4p(f29_unpack(w))));
    }
    return w;
}
// detect_one == false (a bare handle, below): the scalar one is digit 1 of window 0 like any other value
__device__ __forceinline__ bool scalar_digits(const uint32_t *src, int montgomery, int (&d)[MSM_W], bool detect_one = true)
{
    const Fr8 w = scalar_canonical(src, montgomery);
    if (detect_one && w.l[0] == 1u && !(w.l[1] | w.l[2] | w.l[3] | w.l[4] | w.l[5] | w.l[6] | w.l[7])) return true;
    int carry = 0;
#pragma unroll
    for (int j = 0; j < MSM_W; j++) {
        int v = (int)((w.l[j >> 1] >> (16 * (j & 1))) & 0xffffu) + carry;
        carry = 0;
        if (j + 1 < MSM_W && v > MSM_BUCKETS) { v -= 1 << MSM_C; carry = 1; }
        d[j] = v;
    }
    return false;
}

// The counting sort of a signature's 16 n (digit, point) pairs by bucket, without a global atomic: the points are cut into
// MSM_SLICES slices, a workgroup per (signature, slice) histograms its slice in LDS (32,768 counters = 128 KB of the CU's
// 160 KB), the histograms are turned into per-slice starting positions inside every bucket, and the same workgroups then hand
// out positions from LDS again.  (With one global atomicAdd per pair the sort took a third of the whole call: 10^10 atomics
// per second is what the L2 gives, and a signature has 4 x 10^6 pairs to place twice.)
// 32 slices per signature for a batch (their histograms live in the buckets' memory until the first bucket is stored: room for 60);
// a call with ONE signature (an aggregate statement's sum over 2^22 points) would keep 32 of the 256 CUs busy with them, so it cuts
// 224, and parks their histograms in the work items' partial sums instead (131,072 x 240 bytes and more: room for 240), which
// nothing writes before the bucket kernel.
constexpr int MSM_SLICES = 32, MSM_SLICES_LONE = 224;
// A BARE handle (round 5) holds the points themselves and nothing else -- 112 bytes a point instead of 1,792: the key of a
// 1,024-statement aggregate is 2^27 points of h_query alone, whose window tables would be 240 GB.  The sum is then sixteen bucket
// accumulations, one per 16-bit window, over the SAME rows: grid row y of every kernel below is window y of ONE scalar vector
// (instead of signature y with all its windows), a row's entries are point indices, and msm_horner_kernel puts the sixteen window
// sums together (15 x 16 doublings).  The number of bucket additions is the table path's: one per (point, non-zero digit).
// the digit of window `win` (a register array indexed at run time would live in scratch: sixteen selects)
__device__ __forceinline__ int msm_digit_of(const int (&d)[MSM_W], uint32_t win)
{
    int v = 0;
#pragma unroll
    for (int j = 0; j < MSM_W; j++) v = (uint32_t)j == win ? d[j] : v;
    return v;
}
__global__ __launch_bounds__(1024) void msm_hist_kernel(uint32_t n, const uint32_t *__restrict__ scalars, size_t sig_stride_words,
                                                        int montgomery, uint32_t *__restrict__ slice_hist /* [row][slice][buckets] */,
                                                        uint32_t *__restrict__ ones_count /* [sig] */, uint32_t *__restrict__ ones_list /* [sig][n] */,
                                                        int bare)
{
    __shared__ uint32_t hist[MSM_BUCKETS];
    const size_t row = blockIdx.y, sig = bare ? 0 : row;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    for (int b = threadIdx.x; b < MSM_BUCKETS; b += 1024) hist[b] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        int d[MSM_W];
        if (scalar_digits(scalars + sig * sig_stride_words + (size_t)i * 8, montgomery, d, !bare)) {
            ones_list[sig * n + atomicAdd(&ones_count[sig], 1u)] = i;
            continue;
        }
        if (bare) {
            const int v = msm_digit_of(d, (uint32_t)row);
            if (v) atomicAdd(&hist[(v < 0 ? -v : v) - 1], 1u);
            continue;
        }
#pragma unroll
        for (int j = 0; j < MSM_W; j++)
            if (d[j]) atomicAdd(&hist[(d[j] < 0 ? -d[j] : d[j]) - 1], 1u);
    }
    __syncthreads();
    uint32_t *out = slice_hist + (row * slices + slice) * MSM_BUCKETS;
    for (int b = threadIdx.x; b < MSM_BUCKETS; b += 1024) out[b] = hist[b];
}

// A bare handle's sixteen window rows would each turn the scalars into digits again -- sixteen times the 32-byte reads and the
// Montgomery products, and of the 2^27-point sum over h_query's 446 ms the counting sort took 95 (profiles/r05_aggregate1024_timeline.txt).
// So: ONE pass writes the digits, window-major (int16[16][n]: two bytes per scalar and window), and the rows' histogram and scatter
// read their own window's.
__global__ __launch_bounds__(256) void msm_digits_kernel(uint32_t n, const uint32_t *__restrict__ scalars, int montgomery, int16_t *__restrict__ digits)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int d[MSM_W];
    (void)scalar_digits(scalars + (size_t)i * 8, montgomery, d, false);
#pragma unroll
    for (int j = 0; j < MSM_W; j++) digits[(size_t)j * n + i] = (int16_t)(d[j] == (1 << 15) ? -32768 : d[j]);     // (+2^15 does not fit: stored as its negative's code, told apart below)
}
// digits are in [-2^15 + 1, 2^15]: the one value outside int16, +2^15, travels as -32768 (which no digit is)
__device__ __forceinline__ int msm_digit_load(const int16_t *digits, size_t at)
{
    const int v = digits[at];
    return v == -32768 ? 1 << 15 : v;
}
__global__ __launch_bounds__(1024) void msm_hist_digits_kernel(uint32_t n, const int16_t *__restrict__ digits, uint32_t *__restrict__ slice_hist)
{
    __shared__ uint32_t hist[MSM_BUCKETS];
    const size_t row = blockIdx.y;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    for (int b = threadIdx.x; b < MSM_BUCKETS; b += 1024) hist[b] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const int v = msm_digit_load(digits, row * n + i);
        if (v) atomicAdd(&hist[(v < 0 ? -v : v) - 1], 1u);
    }
    __syncthreads();
    uint32_t *out = slice_hist + (row * slices + slice) * MSM_BUCKETS;
    for (int b = threadIdx.x; b < MSM_BUCKETS; b += 1024) out[b] = hist[b];
}
__global__ __launch_bounds__(1024) void msm_scatter_digits_kernel(uint32_t n, const int16_t *__restrict__ digits, const uint32_t *__restrict__ offsets,
                                                                  const uint32_t *__restrict__ slice_hist, uint32_t *__restrict__ entries)
{
    __shared__ uint32_t cursor[MSM_BUCKETS];
    const size_t row = blockIdx.y;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    const uint32_t *first = slice_hist + (row * slices + slice) * MSM_BUCKETS, *off = offsets + row * MSM_BUCKETS;
    for (int b = threadIdx.x; b < MSM_BUCKETS; b += 1024) cursor[b] = off[b] + first[b];
    __syncthreads();
    uint32_t *ent = entries + row * (size_t)n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const int v = msm_digit_load(digits, row * n + i);
        if (v) ent[atomicAdd(&cursor[(uint32_t)(v < 0 ? -v : v) - 1u], 1u)] = i | (v < 0 ? 0x80000000u : 0u);
    }
}

// ---- WIDE windows for the largest bare sums (round 5) -------------------------------------------------------------------------------
// A bare handle's windows cost no table, so their width is free: thirteen 20-bit windows instead of sixteen 16-bit ones are 19 % fewer
// bucket additions (one per point and non-zero digit: 13 n instead of 16 n) -- what the 2^27-point sum over h_query of the 1,024-statement
// aggregate spends 350 of its 430 ms on.  2^19 buckets per window do not fit the LDS histograms of the counting sort above; they need not:
// bucket m = |d| - 1 of window j is bucket (m mod 2^15) of ROW j 16 + (m >> 15), and a row is to every kernel from the fine sort on what a
// window (or a signature) was -- 32,768 buckets, work items, combine, the two-stage fold -- 208 rows of them.  The sort is in two levels
// (below: bins of 1,024 buckets, then the buckets of a bin), every pass writing runs.  A row's buckets fold to S1 = sum (lo + 1) B_lo AND
// S0 = sum B_lo; a window is sum_hi S1_hi + 2^15 sum_hi hi S0_hi (msm_wide_window_kernel), the windows join by Horner's rule with 20
// doublings each.  The folds of 13 x 2^19 buckets are a fixed 14 ms, three windows' additions 0.18 ns a point: handles of 2^26 points and
// more take wide windows by themselves (MSM_WIDE_FROM_DEFAULT), smaller ones when loaded so.
constexpr int WIDE_C = 20, WIDE_W = 13, WIDE_H = 1 << (WIDE_C - MSM_C), WIDE_ROWS = WIDE_W * WIDE_H;
constexpr uint32_t WIDE_MAX_ITEMS = 65536;       // >= 32,768 + (32,768 / 1.5) 33 / 32 (finer = 1)
constexpr int WIDE_SLICES = 128;                 // slices of a window in the bins' sort (whole tiles each)
// signed 20-bit digits, d_j in [-2^19 + 1, 2^19], sum d_j 2^(20 j) = k; the top window (bits 240 ..) takes the last carry
__device__ __forceinline__ void scalar_digits_wide(const uint32_t *src, int montgomery, int (&d)[WIDE_W])
{
    const Fr8 w = scalar_canonical(src, montgomery);
    int carry = 0;
#pragma unroll
    for (int j = 0; j < WIDE_W; j++) {
        const int bit = WIDE_C * j, k = bit >> 5, sh = bit & 31;
        const uint64_t two = (uint64_t)w.l[k] | (k + 1 < 8 ? (uint64_t)w.l[k + 1] << 32 : 0ull);
        int v = (int)((two >> sh) & ((1u << WIDE_C) - 1u)) + carry;
        carry = 0;
        if (j + 1 < WIDE_W && v > (1 << (WIDE_C - 1))) { v -= 1 << WIDE_C; carry = 1; }
        d[j] = v;
    }
}
// The sort, every pass writing RUNS (the first version of it scattered 4-byte entries into 208 x 32,768 open cache lines: each such write a
// read-modify-write of a line in HBM, 76 ms for 1.7 x 10^9 of them -- all that three windows' additions saved):
//   digits     int32[13][n], written once
//   bins       per window, its points by bin = m >> 10 (512 bins): a workgroup sorts a TILE of 4,096 entries by bin in LDS and copies the
//              runs out (a bin's run of a tile to consecutive addresses: 8 entries = 64 bytes on average, whole lines)
//   buckets    per bin (262,000 entries of 1,024 buckets: one workgroup, 4 KB of counters): the 4-byte entries of a bin go into a 1 MB region
//              whose 1,024 open lines stay in the L2
// A row's thirty-two bins lie one after the other, so rows and bins share one numbering: bin g = row (g >> 5), buckets [1,024 (g & 31), ..).
constexpr int WIDE_BINS = 512, WIDE_BIN_BUCKETS = MSM_BUCKETS * WIDE_H / WIDE_BINS, WIDE_ALL_BINS = WIDE_W * WIDE_BINS, WIDE_TILE = 4096;
static_assert(WIDE_BIN_BUCKETS == 1024 && WIDE_ALL_BINS == WIDE_ROWS * 32, "a row = 32 bins of 1,024 buckets");
__global__ __launch_bounds__(256) void msm_wide_digits_kernel(uint32_t n, const uint32_t *__restrict__ scalars, int montgomery, int32_t *__restrict__ digits)
{
    const uint32_t i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    int d[WIDE_W];
    scalar_digits_wide(scalars + (size_t)i * 8, montgomery, d);
#pragma unroll
    for (int j = 0; j < WIDE_W; j++) digits[(size_t)j * n + i] = d[j];
}
__global__ __launch_bounds__(1024) void msm_wide_bin_hist_kernel(uint32_t n, const int32_t *__restrict__ digits, uint32_t *__restrict__ slice_hist /* [window][slice][512] */)
{
    __shared__ uint32_t hist[WIDE_BINS];
    const size_t j = blockIdx.y;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = ((n + slices - 1) / slices + WIDE_TILE - 1) / WIDE_TILE * WIDE_TILE;
    const uint32_t lo = (uint64_t)slice * per < n ? slice * per : n, hi = (uint64_t)lo + per < n ? lo + per : n;
    if (threadIdx.x < WIDE_BINS) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        const int v = digits[j * n + i];
        if (v) atomicAdd(&hist[((uint32_t)(v < 0 ? -v : v) - 1u) >> 10], 1u);
    }
    __syncthreads();
    if (threadIdx.x < WIDE_BINS) slice_hist[(j * slices + slice) * WIDE_BINS + threadIdx.x] = hist[threadIdx.x];
}
// every slice's first position inside its bins; the bins' sizes and starts; the rows' (one workgroup)
__global__ __launch_bounds__(1024) void msm_wide_bin_offsets_kernel(uint32_t slices, uint32_t *__restrict__ slice_hist, uint32_t *__restrict__ bin_count,
                                                                    unsigned long long *__restrict__ bin_start, uint32_t *__restrict__ row_count,
                                                                    unsigned long long *__restrict__ row_start)
{
    for (uint32_t g = threadIdx.x; g < (uint32_t)WIDE_ALL_BINS; g += 1024) {
        const uint32_t j = g / WIDE_BINS, bin = g % WIDE_BINS;
        uint32_t run = 0;
        for (uint32_t s_ = 0; s_ < slices; s_++) {
            uint32_t *h = slice_hist + ((size_t)j * slices + s_) * WIDE_BINS + bin;
            const uint32_t c = *h;
            *h = run;
            run += c;
        }
        bin_count[g] = run;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned long long at = 0;
        for (int g = 0; g < WIDE_ALL_BINS; g++) { bin_start[g] = at; at += bin_count[g]; }
    }
    __syncthreads();
    if (threadIdx.x < (unsigned)WIDE_ROWS) {
        const uint32_t r = threadIdx.x;
        uint32_t c = 0;
        for (int k = 0; k < 32; k++) c += bin_count[r * 32 + k];
        row_count[r] = c;
        row_start[r] = bin_start[r * 32];
    }
}
__global__ __launch_bounds__(1024) void msm_wide_bin_scatter_kernel(uint32_t n, const int32_t *__restrict__ digits, const uint32_t *__restrict__ slice_hist,
                                                                    const unsigned long long *__restrict__ bin_start,
                                                                    uint2 *__restrict__ coarse /* point | sign << 31, bucket of the bin */)
{
    __shared__ uint2 stage[WIDE_TILE];
    __shared__ uint16_t sbin[WIDE_TILE];
    __shared__ uint32_t tcount[WIDE_BINS], tstart[WIDE_BINS], tfill[WIDE_BINS], wtotal[WIDE_BINS / 64];
    __shared__ unsigned long long gpos[WIDE_BINS];
    const size_t j = blockIdx.y;
    const uint32_t t = threadIdx.x;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = ((n + slices - 1) / slices + WIDE_TILE - 1) / WIDE_TILE * WIDE_TILE;
    const uint32_t lo = (uint64_t)slice * per < n ? slice * per : n, hi = (uint64_t)lo + per < n ? lo + per : n;
    if (t < WIDE_BINS) gpos[t] = bin_start[j * WIDE_BINS + t] + slice_hist[(j * slices + slice) * WIDE_BINS + t];
    for (uint32_t tile0 = lo; tile0 < hi; tile0 += WIDE_TILE) {
        if (t < WIDE_BINS) { tcount[t] = 0; tfill[t] = 0; }
        __syncthreads();
        int dv[WIDE_TILE / 1024];
#pragma unroll
        for (int k = 0; k < WIDE_TILE / 1024; k++) {
            const uint32_t i = tile0 + t + 1024u * k;
            dv[k] = i < hi ? digits[j * n + i] : 0;
            if (dv[k]) atomicAdd(&tcount[((uint32_t)(dv[k] < 0 ? -dv[k] : dv[k]) - 1u) >> 10], 1u);
        }
        __syncthreads();
        // exclusive prefix sums of the 512 counts: within each of the first eight wavefronts by shuffles, their totals through LDS (two
        // barriers; a scan of nine steps through LDS was eighteen, most of this kernel's waiting)
        {
            uint32_t incl = t < WIDE_BINS ? tcount[t] : 0u;
            const uint32_t own = incl, lane = t & 63;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const uint32_t v = __shfl_up(incl, off, 64);
                if (lane >= (uint32_t)off) incl += v;
            }
            if (t < WIDE_BINS && lane == 63) wtotal[t >> 6] = incl;
            __syncthreads();
            if (t < WIDE_BINS) {
                uint32_t before = 0;
                for (uint32_t w_ = 0; w_ < (t >> 6); w_++) before += wtotal[w_];
                tstart[t] = before + incl - own;
            }
            __syncthreads();
        }
#pragma unroll
        for (int k = 0; k < WIDE_TILE / 1024; k++) {
            if (!dv[k]) continue;
            const uint32_t i = tile0 + t + 1024u * k, m = (uint32_t)(dv[k] < 0 ? -dv[k] : dv[k]) - 1u, bin = m >> 10;
            const uint32_t pos = tstart[bin] + atomicAdd(&tfill[bin], 1u);
            stage[pos] = make_uint2(i | (dv[k] < 0 ? 0x80000000u : 0u), m & (WIDE_BIN_BUCKETS - 1));
            sbin[pos] = (uint16_t)bin;
        }
        __syncthreads();
        const uint32_t total = tstart[WIDE_BINS - 1] + tcount[WIDE_BINS - 1];
        for (uint32_t p = t; p < total; p += 1024) {
            const uint32_t bin = sbin[p];
            coarse[gpos[bin] + (p - tstart[bin])] = stage[p];
        }
        __syncthreads();
        if (t < WIDE_BINS) gpos[t] += tcount[t];
        __syncthreads();
    }
}
// The buckets' sort: a bin's 4-byte entries go into its 1 MB of the entry array, 1,024 lines open at a time -- and with a workgroup per
// bin, 512 bins in flight, those 512 MB of open lines left the L2 one dirty word at a time (33 ms: 77 bytes of HBM traffic per entry; with
// cursors in global memory shared by the workgroups of a bin, 107 ms: device-wide atomics).  So a bin is cut into WIDE_PARTS parts by entry,
// one workgroup each with its cursors in LDS (a part's own first positions come from the parts' histograms), and the parts of a bin run on
// the SAME XCD at the same time (workgroups go to the XCDs round robin, w % 8): two bins in flight per XCD, their lines complete before
// they leave its L2.  The order inside a bucket is what the LDS atomics make it within a part; parts in order.
constexpr int WIDE_PARTS = 32;
static_assert(WIDE_ALL_BINS % 8 == 0, "bins go to the XCDs eight at a time");
struct WidePart { size_t g; uint32_t part, lo, hi; };
__device__ __forceinline__ WidePart wide_part_of(uint32_t w, const uint32_t *__restrict__ bin_count)
{
    WidePart p;
    const uint32_t q = w >> 3;
    p.part = q % WIDE_PARTS;
    p.g = (size_t)(q / WIDE_PARTS) * 8 + (w & 7);
    const uint32_t cnt = bin_count[p.g];
    p.lo = (uint32_t)((uint64_t)cnt * p.part / WIDE_PARTS);
    p.hi = (uint32_t)((uint64_t)cnt * (p.part + 1) / WIDE_PARTS);
    return p;
}
// the sizes of the 1,024 buckets within every part (grid: bins x parts)
__global__ __launch_bounds__(1024) void msm_wide_bucket_hist_kernel(const uint2 *__restrict__ coarse, const uint32_t *__restrict__ bin_count,
                                                                    const unsigned long long *__restrict__ bin_start, uint32_t *__restrict__ part_hist)
{
    __shared__ uint32_t hist[WIDE_BIN_BUCKETS];
    const WidePart p = wide_part_of(blockIdx.x, bin_count);
    const uint2 *src = coarse + bin_start[p.g];
    hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = p.lo + threadIdx.x; i < p.hi; i += 1024) atomicAdd(&hist[src[i].y], 1u);
    __syncthreads();
    part_hist[(p.g * WIDE_PARTS + p.part) * WIDE_BIN_BUCKETS + threadIdx.x] = hist[threadIdx.x];
}
// per bin: a part's first position within each bucket, and the buckets' sizes straight into the rows' count arrays (grid: the 6,656 bins)
__global__ __launch_bounds__(1024) void msm_wide_bucket_parts_kernel(uint32_t *__restrict__ part_hist, uint32_t *__restrict__ counts)
{
    const size_t g = blockIdx.x;
    uint32_t run = 0;
    for (int part = 0; part < WIDE_PARTS; part++) {
        uint32_t *h = part_hist + (g * WIDE_PARTS + part) * WIDE_BIN_BUCKETS + threadIdx.x;
        const uint32_t c = *h;
        *h = run;
        run += c;
    }
    counts[g * WIDE_BIN_BUCKETS + threadIdx.x] = run;
}
__global__ __launch_bounds__(1024) void msm_wide_bucket_scatter_kernel(const uint2 *__restrict__ coarse, const uint32_t *__restrict__ bin_count,
                                                                       const unsigned long long *__restrict__ bin_start, const uint32_t *__restrict__ offsets,
                                                                       const uint32_t *__restrict__ part_hist, const unsigned long long *__restrict__ row_start,
                                                                       uint32_t *__restrict__ entries)
{
    __shared__ uint32_t cursor[WIDE_BIN_BUCKETS];
    const WidePart p = wide_part_of(blockIdx.x, bin_count);
    const uint2 *src = coarse + bin_start[p.g];
    // (relative to the row: offsets are [row][32,768] = [bin][1,024])
    cursor[threadIdx.x] = offsets[p.g * WIDE_BIN_BUCKETS + threadIdx.x] + part_hist[(p.g * WIDE_PARTS + p.part) * WIDE_BIN_BUCKETS + threadIdx.x];
    __syncthreads();
    uint32_t *ent = entries + row_start[p.g >> 5];
    // (a part is some 8,000 entries, eight a thread: all its loads first -- one memory latency for the workgroup instead of eight in a row,
    // 213,000 workgroups long: 15.8 -> ms of this kernel were those)
    constexpr int U = 8;
    for (uint32_t i0 = p.lo + threadIdx.x; i0 < p.hi; i0 += 1024 * U) {
        uint2 e[U];
#pragma unroll
        for (int k = 0; k < U; k++) e[k] = i0 + 1024u * k < p.hi ? src[i0 + 1024u * k] : make_uint2(0u, 0u);
#pragma unroll
        for (int k = 0; k < U; k++)
            if (i0 + 1024u * k < p.hi) ent[atomicAdd(&cursor[e[k].y], 1u)] = e[k].x;
    }
}
// window j = sum_hi S1(j, hi) + 2^15 sum_hi hi S0(j, hi): one lane per window (F::LANES lanes)
template <class F>
__global__ __launch_bounds__(64) void msm_wide_window_kernel(const uint32_t *__restrict__ s1 /* [208][BK_WORDS] */, const uint32_t *__restrict__ s0,
                                                             uint32_t *__restrict__ window_sums /* [13][BK_WORDS] */)
{
    __builtin_amdgcn_s_setprio(2);
    constexpr int BW = Grp<F>::BK_WORDS;
    if (threadIdx.x >= (unsigned)F::LANES) return;
    const size_t j = blockIdx.x;
    XyzzT<F> run = pt_identity<F>(), weighted = pt_identity<F>(), plain = load_bucket<F>(s1 + (j * WIDE_H) * BW);
    for (int hi = WIDE_H - 1; hi >= 1; hi--) {
        run = pt_add(run, load_bucket<F>(s0 + (j * WIDE_H + hi) * BW));
        weighted = pt_add(weighted, run);                            // sum_(hi >= 1) hi S0_hi
        plain = pt_add(plain, load_bucket<F>(s1 + (j * WIDE_H + hi) * BW));
    }
    for (int k = 0; k < MSM_C - 1; k++) weighted = pt_double(weighted);
    store_bucket<F>(window_sums + j * BW, pt_add(plain, weighted));
}

// per bucket: its size (counts) and, in place of every slice's count, the slice's first position inside the bucket
__global__ __launch_bounds__(256) void msm_slice_offsets_kernel(uint32_t *__restrict__ slice_hist, uint32_t *__restrict__ counts, int slices)
{
    const size_t sig = blockIdx.y;
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    uint32_t *h = slice_hist + sig * slices * (size_t)MSM_BUCKETS + b;
    uint32_t run = 0;
#pragma unroll 4
    for (int s_ = 0; s_ < slices; s_++) {
        const uint32_t c = h[(size_t)s_ * MSM_BUCKETS];
        h[(size_t)s_ * MSM_BUCKETS] = run;
        run += c;
    }
    counts[sig * MSM_BUCKETS + b] = run;
}

// exclusive prefix sums of a signature's 32,768 bucket sizes: where each bucket's entries start
__global__ __launch_bounds__(1024) void msm_scan_kernel(const uint32_t *__restrict__ counts, uint32_t *__restrict__ offsets)
{
    __shared__ uint32_t part[1024];
    const size_t sig = blockIdx.x;
    const uint32_t *cnt = counts + sig * MSM_BUCKETS;
    const int t = threadIdx.x;
    constexpr int PER = MSM_BUCKETS / 1024;
    uint32_t local[PER], s = 0;
#pragma unroll
    for (int k = 0; k < PER; k++) { local[k] = s; s += cnt[t * PER + k]; }
    part[t] = s;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? part[t - off] : 0u;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    const uint32_t base = part[t] - s;
#pragma unroll
    for (int k = 0; k < PER; k++) offsets[sig * MSM_BUCKETS + t * PER + k] = base + local[k];
}

__global__ __launch_bounds__(1024) void msm_scatter_kernel(uint32_t n, const uint32_t *__restrict__ scalars, size_t sig_stride_words,
                                                           int montgomery, const uint32_t *__restrict__ offsets,
                                                           const uint32_t *__restrict__ slice_hist, uint32_t *__restrict__ entries, int bare)
{
    __shared__ uint32_t cursor[MSM_BUCKETS];
    const size_t row = blockIdx.y, sig = bare ? 0 : row;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    const uint32_t *first = slice_hist + (row * slices + slice) * MSM_BUCKETS, *off = offsets + row * MSM_BUCKETS;
    for (int b = threadIdx.x; b < MSM_BUCKETS; b += 1024) cursor[b] = off[b] + first[b];
    __syncthreads();
    uint32_t *ent = entries + row * (bare ? (size_t)n : (size_t)MSM_W * n);      // a bare row: one window, entries = point indices
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        int d[MSM_W];
        if (scalar_digits(scalars + sig * sig_stride_words + (size_t)i * 8, montgomery, d, !bare)) continue;
        if (bare) {
            const int v = msm_digit_of(d, (uint32_t)row);
            if (v) ent[atomicAdd(&cursor[(uint32_t)(v < 0 ? -v : v) - 1u], 1u)] = i | (v < 0 ? 0x80000000u : 0u);
            continue;
        }
#pragma unroll
        for (int j = 0; j < MSM_W; j++) {
            if (!d[j]) continue;
            const uint32_t b = (uint32_t)(d[j] < 0 ? -d[j] : d[j]) - 1u;
            ent[atomicAdd(&cursor[b], 1u)] = ((uint32_t)j * n + i) | (d[j] < 0 ? 0x80000000u : 0u);
        }
    }
}

// Work items.  A bucket is added up by one thread, so a bucket far above the mean is a serial chain the whole call waits for:
// a Falcon witness as scalars leaves 35,000 pairs in all, 775 of them in the bucket of digit +-1 (carries of the recoding),
// and a lone wavefront needs ~18 us per addition.  Buckets are therefore cut into ITEMS of at most `split` entries, equal
// parts of their bucket, with split = max(32, 1.5 x the mean bucket size) decided per call on the device: the 2^18-point sum
// of h (mean 128, sigma 11) is not cut at all, the witness-side sums are cut at 32.  Items are laid out by decreasing bucket
// size (a counting sort), so the 64 lanes of a wavefront run equally long; a bucket's items are neighbours, and
// msm_combine_kernel adds them up.  At most 32,768 + 21,845 items whatever the scalars (sum of ceil(c_b / split)).
constexpr int MSM_SIZE_CLASSES = 1024;         // sizes >= this share the first class
constexpr int MSM_MAX_ITEMS = 131072;          // >= 32,768 + 32,768 / (1.5 / 4): the finest split (a lone signature), a multiple of 64
// Sums over more than 2^18 points (an aggregate statement's h_query: ONE signature, 2^22 points, buckets of 2,048 entries) are cut
// finer still: `finer` = 24 makes a bucket of the mean size 16 items of 128 entries -- 524,288 items, four rounds of the chip's
// 131,072 thread slots at two wavefronts per SIMD (with 4 it was 98,304 items: a quarter of the slots idle and half the SIMDs with a
// single wavefront, 1.55 x the time per addition of a 64-signature call: 16.7 ms; with 12, two rounds: 14.4 ms; with 24: 13.4 ms, the
// combine's extra additions included).  Bound: 32,768 + 32,768 finer / 1.5.
// (msm_split_of rounds the split DOWN, by less than one entry in at least 32: 33 / 32 of that bound is what the list must hold; an item is
// two words, bucket and part -- packed into one, `part << 15` overflowed for a bucket cut into more than 2^17 parts, which one digit repeated
// in all the scalars of a 2^22-point sum produces: round 4's ADVICE)
constexpr int MSM_MAX_ITEMS_LARGE = 32768 + (32768 * 24 * 2 / 3) / 32 * 33 + 64;     // 573,504
static_assert(MSM_MAX_ITEMS_LARGE % 64 == 0 && MSM_MAX_ITEMS >= 32768 + (32768 * 4 * 2 / 3) / 32 * 33 + 64, "whole wavefronts; the finest split fits");
// a bare handle's rows are windows of one sum: sixteen of them fill the chip with items of the buckets' own size (finer = 1)
__host__ __device__ constexpr uint32_t msm_max_items(uint32_t n, bool bare = false)
{
    return !bare && n > (1u << 18) ? (uint32_t)MSM_MAX_ITEMS_LARGE : (uint32_t)MSM_MAX_ITEMS;
}
// `finer`: 1 for a batch, 4 for one or two signatures at a time -- a bucket of the mean size is then three items instead of one
// (a lone proof waits for ~128 dependent additions otherwise), and msm_combine_kernel adds them up
__device__ __forceinline__ uint32_t msm_split_of(uint32_t total, uint32_t finer)
{
    const uint32_t s = (uint32_t)(((uint64_t)total * 3 + 2 * MSM_BUCKETS - 1) / (2 * MSM_BUCKETS)) / finer;
    return s < 32u ? 32u : s;
}
__global__ __launch_bounds__(1024) void msm_order_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
                                                         uint32_t *__restrict__ order, uint32_t *__restrict__ item_first /* [sig][buckets] */,
                                                         uint32_t *__restrict__ items /* [sig][max_items][2]: bucket, part */,
                                                         uint32_t *__restrict__ item_count /* [sig] */, uint32_t finer, uint32_t max_items)
{
    __shared__ uint32_t hist[MSM_SIZE_CLASSES];
    const size_t sig = blockIdx.x;
    const uint32_t *cnt = counts + sig * MSM_BUCKETS;
    uint32_t *ord = order + sig * MSM_BUCKETS;
    const int t = threadIdx.x;
    hist[t] = 0;
    __syncthreads();
    auto cls = [](uint32_t c) { return c >= (uint32_t)MSM_SIZE_CLASSES ? 0u : (uint32_t)(MSM_SIZE_CLASSES - 1) - c; };   // big first
    for (int b = t; b < MSM_BUCKETS; b += 1024) atomicAdd(&hist[cls(cnt[b])], 1u);
    __syncthreads();
    // exclusive scan of the 1,024 class counts
    const uint32_t mine = hist[t];
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? hist[t - off] : 0u;
        __syncthreads();
        hist[t] += v;
        __syncthreads();
    }
    const uint32_t excl = hist[t] - mine;
    __syncthreads();
    hist[t] = excl;
    __syncthreads();
    for (int b = t; b < MSM_BUCKETS; b += 1024) ord[atomicAdd(&hist[cls(cnt[b])], 1u)] = (uint32_t)b;
    __threadfence_block();
    __syncthreads();
    // items: rank by rank, ceil(c / split) each (an empty bucket keeps one: its thread stores the identity)
    const uint32_t total = offsets[sig * MSM_BUCKETS + MSM_BUCKETS - 1] + cnt[MSM_BUCKETS - 1], split = msm_split_of(total, finer);
    constexpr int PER = MSM_BUCKETS / 1024;
    uint32_t k[PER], local = 0;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t c = cnt[ord[t * PER + j]];
        k[j] = c <= split ? 1u : (c + split - 1) / split;
        local += k[j];
    }
    hist[t] = local;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const uint32_t v = t >= off ? hist[t - off] : 0u;
        __syncthreads();
        hist[t] += v;
        __syncthreads();
    }
    uint32_t pos = hist[t] - local;
    if (t == 1023) item_count[sig] = hist[t] < max_items ? hist[t] : max_items;     // (the bound above holds; a list is never overrun if it did not)
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const uint32_t b = ord[t * PER + j];
        item_first[sig * MSM_BUCKETS + b] = pos;
        for (uint32_t c = 0; c < k[j] && pos + c < max_items; c++) {
            uint32_t *it = items + (sig * (size_t)max_items + pos + c) * 2;
            it[0] = b;
            it[1] = c;
        }
        pos += k[j];
    }
}

// one thread per (signature, item): the sum of the item's table rows, in registers.  G1: the row of entry k + 1 is fetched
// while entry k is added (an addition is ~5,000 vector instructions: the gather of 112 bytes hides behind it); G2 has no
// registers to spare for that (an accumulator alone is 112).
template <class F, bool PREFETCH>
__global__ __launch_bounds__(64, 2) void msm_bucket_kernel(MsmDev m, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ counts,
                                                           const uint32_t *__restrict__ items, const uint32_t *__restrict__ item_count,
                                                           const uint32_t *__restrict__ entries, uint32_t *__restrict__ partial_items, uint32_t finer,
                                                           uint32_t max_items, size_t ent_stride /* entries per row: 16 n, a bare handle's: n */,
                                                           const unsigned long long *__restrict__ row_base /* wide windows: where each row's entries start; else null */)
{
    constexpr int PW = Grp<F>::PT_WORDS;
    const size_t sig = blockIdx.y;
    const uint32_t it = blockIdx.x * 64 + threadIdx.x;
    if (it >= item_count[sig]) return;
    const uint2 item = *(const uint2 *)(items + (sig * (size_t)max_items + it) * 2);
    const uint32_t b = item.x, chunk = item.y;
    const uint32_t c = counts[sig * MSM_BUCKETS + b];
    const uint32_t total = offsets[sig * MSM_BUCKETS + MSM_BUCKETS - 1] + counts[sig * MSM_BUCKETS + MSM_BUCKETS - 1], split = msm_split_of(total, finer);
    const uint32_t k = c <= split ? 1u : (c + split - 1) / split;
    const uint32_t lo = (uint32_t)((uint64_t)c * chunk / k), cnt = (uint32_t)((uint64_t)c * (chunk + 1) / k) - lo;     // equal parts
    const uint32_t *ent = entries + (row_base ? (size_t)row_base[sig] : sig * ent_stride) + offsets[sig * MSM_BUCKETS + b] + lo;
    XyzzT<F> acc = pt_identity<F>();
    if (PREFETCH) {
        uint32_t e = cnt ? ent[0] : 0u;
        AffineT<F> p = load_row<F>(m.table + (size_t)(e & 0x7fffffffu) * PW);
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t e_next = j + 1 < cnt ? ent[j + 1] : e;
            const AffineT<F> p_next = load_row<F>(m.table + (size_t)(e_next & 0x7fffffffu) * PW);
            if (e >> 31) p.y = F::template neg<Grp<F>::K_AFFINE_Y>(p.y);
            acc = pt_add_affine(acc, p);
            e = e_next;
            p = p_next;
        }
    } else {
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t e = ent[j];
            AffineT<F> p = load_row<F>(m.table + (size_t)(e & 0x7fffffffu) * PW);
            if (e >> 31) p.y = F::template neg<Grp<F>::K_AFFINE_Y>(p.y);
            acc = pt_add_affine(acc, p);
        }
    }
    store_bucket<F>(partial_items + (sig * (size_t)max_items + it) * (size_t)Grp<F>::BK_WORDS, acc);
}

// sum_j 2^(window_bits j) S_j for the window sums S_j of a bare handle (Horner: window_bits doublings and one addition per window:
// 255 point operations of ONE chain -- F::LANES lanes -- whatever the size of the sum: 5 ms beside the 10^9 additions it crowns).
// Grid: one workgroup per sum (the three G1 tables of a proof side by side); sums: [sum][windows][BK_WORDS].
template <class F>
__global__ __launch_bounds__(64) void msm_horner_kernel(const uint32_t *__restrict__ sums, int windows, int window_bits,
                                                        uint32_t *__restrict__ out /* [sum][ARK_WORDS], or [sum][BK_WORDS] */, int xyzz_out,
                                                        uint32_t out_step = 1 /* XYZZ rows: sum i goes to row i out_step */)
{
    __builtin_amdgcn_s_setprio(2);
    constexpr int BW = Grp<F>::BK_WORDS;
    if (threadIdx.x >= (unsigned)F::LANES) return;
    const uint32_t *src = sums + (size_t)blockIdx.x * windows * BW;
    XyzzT<F> acc = load_bucket<F>(src + (size_t)(windows - 1) * BW);
    for (int j = windows - 2; j >= 0; j--) {
        for (int k = 0; k < window_bits; k++) acc = pt_double(acc);
        acc = pt_add(acc, load_bucket<F>(src + (size_t)j * BW));
    }
    if (xyzz_out) store_bucket<F>(out + (size_t)blockIdx.x * out_step * BW, acc);
    else store_ark_point<F>(out + (size_t)blockIdx.x * Grp<F>::ARK_WORDS, pt_to_affine(acc));
}
// the rows of a table that hold a point (not all words zero), in order: per block of 1,024 rows their number, then -- the blocks' running
// totals in between, one workgroup -- their indices
template <int PT_WORDS_>
__global__ __launch_bounds__(1024) void live_rows_kernel(const uint32_t *__restrict__ rows, uint32_t n, uint32_t *__restrict__ block_counts /* in: null or the blocks' first positions */,
                                                         uint32_t *__restrict__ index /* null: count only */)
{
    __shared__ uint32_t wave_total[16];
    const uint32_t i = blockIdx.x * 1024 + threadIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t any = 0;
    if (i < n) {
        const uint4 *v = (const uint4 *)(rows + (size_t)i * PT_WORDS_);
#pragma unroll
        for (int k = 0; k < PT_WORDS_ / 4; k++) { const uint4 w = v[k]; any |= w.x | w.y | w.z | w.w; }
    }
    const unsigned long long ballot = __builtin_amdgcn_ballot_w64(any != 0);
    if (lane == 0) wave_total[wave] = (uint32_t)__popcll(ballot);
    __syncthreads();
    uint32_t before = 0, total = 0;
    for (uint32_t w = 0; w < 16; w++) { before += w < wave ? wave_total[w] : 0u; total += wave_total[w]; }
    if (!index) {
        if (threadIdx.x == 0) block_counts[blockIdx.x] = total;
        return;
    }
    if (any) index[block_counts[blockIdx.x] + before + (uint32_t)__popcll(ballot & ((1ull << lane) - 1))] = i;
}
__global__ __launch_bounds__(1024) void live_rows_scan_kernel(uint32_t *__restrict__ block_counts, uint32_t blocks, uint32_t *__restrict__ total)
{
    // one workgroup: every thread a run of blocks, the runs' totals scanned through LDS
    __shared__ uint32_t run_total[1024];
    const uint32_t per = (blocks + 1023) / 1024, lo = threadIdx.x * per, hi = lo + per < blocks ? lo + per : blocks;
    uint32_t sum = 0;
    for (uint32_t b = lo; b < hi; b++) sum += block_counts[b];
    run_total[threadIdx.x] = sum;
    __syncthreads();
    uint32_t before = 0;
    for (uint32_t t = 0; t < threadIdx.x; t++) before += run_total[t];
    for (uint32_t b = lo; b < hi; b++) { const uint32_t c = block_counts[b]; block_counts[b] = before; before += c; }
    if (threadIdx.x == 1023) *total = before;
}

// bucket b = the sum of its items (one for almost every bucket: a copy)
template <class F>
__global__ __launch_bounds__(64, 2) void msm_combine_kernel(const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ counts,
                                                            const uint32_t *__restrict__ item_first, const uint32_t *__restrict__ partial_items,
                                                            uint32_t *__restrict__ buckets, uint32_t finer, uint32_t max_items)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int BW = Grp<F>::BK_WORDS;
    const size_t sig = blockIdx.y;
    const uint32_t b = blockIdx.x * 64 + threadIdx.x;
    const uint32_t c = counts[sig * MSM_BUCKETS + b];
    const uint32_t total = offsets[sig * MSM_BUCKETS + MSM_BUCKETS - 1] + counts[sig * MSM_BUCKETS + MSM_BUCKETS - 1], split = msm_split_of(total, finer);
    const uint32_t k = c <= split ? 1u : (c + split - 1) / split;
    const uint32_t *src = partial_items + (sig * (size_t)max_items + item_first[sig * MSM_BUCKETS + b]) * (size_t)BW;
    XyzzT<F> acc = load_bucket<F>(src);
    for (uint32_t j = 1; j < k; j++) acc = pt_add(acc, load_bucket<F>(src + (size_t)j * BW));
    store_bucket<F>(buckets + (sig * MSM_BUCKETS + b) * (size_t)BW, acc);
}

// the points whose scalar is one, summed by the threads of the fold's first stage (thread t: every nthreads-th of the list); the
// partial sums enter the fold with weight one
template <class F>
__global__ __launch_bounds__(64, 2) void msm_ones_kernel(MsmDev m, const uint32_t *__restrict__ ones_count, const uint32_t *__restrict__ ones_list,
                                                         uint32_t *__restrict__ partial /* [sig][MSM_FOLD1_THREADS][BK_WORDS] */)
{
    const size_t sig = blockIdx.y;
    const uint32_t nthreads = gridDim.x * 64;
    const uint32_t t = blockIdx.x * 64 + threadIdx.x, cnt = ones_count[sig];
    const uint32_t *list = ones_list + sig * m.n;
    XyzzT<F> acc = pt_identity<F>();
    for (uint32_t k = t; k < cnt; k += nthreads)
        acc = pt_add_affine(acc, load_row<F>(m.table + (size_t)list[k] * Grp<F>::PT_WORDS));      // window 0 of the table = the point itself
    store_bucket<F>(partial + (sig * MSM_FOLD1_THREADS + t) * (size_t)Grp<F>::BK_WORDS, acc);
}

// sum_b (b + 1) B_b per signature, in two stages (one workgroup of 512 threads x 64 buckets each was 200 dependent point
// operations long: 6 ms for G1, 19 for G2, whatever the batch).  Stage 1: 32,768 / CHUNK threads x CHUNK buckets: running sums,
// the multiple CHUNK g x (sum of the chunk) by double-and-add, plus the thread's share of the scalars that are one; overwrites
// `partial`.  CHUNK = 2^LOG_CHUNK is 8 for small batches (latency) and up to 64 for large ones (the multiples are a third of the
// work at 8).
template <class F, int LOG_CHUNK>
__global__ __launch_bounds__(64, 2) void msm_fold1_kernel(const uint32_t *__restrict__ buckets, uint32_t *__restrict__ partial,
                                                          uint32_t *__restrict__ partial_plain /* wide windows: the chunk's UNWEIGHTED sum too; else null */)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int BW = Grp<F>::BK_WORDS, CHUNK = 1 << LOG_CHUNK;
    const size_t sig = blockIdx.y;
    const int g = blockIdx.x * 64 + threadIdx.x;
    const uint32_t *bk = buckets + (sig * MSM_BUCKETS + (size_t)g * CHUNK) * BW;
    XyzzT<F> run = pt_identity<F>(), sum = pt_identity<F>();
    for (int k = CHUNK - 1; k >= 0; k--) {
        run = pt_add(run, load_bucket<F>(bk + (size_t)k * BW));
        sum = pt_add(sum, run);
    }
    // sum = sum_k (k + 1) B_(CHUNK g + k); the buckets' weights are CHUNK g + k + 1: add (CHUNK g) run = 2^LOG_CHUNK (g run)
    XyzzT<F> mult = pt_identity<F>();
    for (int bit = 14 - LOG_CHUNK; bit >= 0; bit--) {
        mult = pt_double(mult);
        if ((g >> bit) & 1) mult = pt_add(mult, run);
    }
    for (int k = 0; k < LOG_CHUNK; k++) mult = pt_double(mult);
    sum = pt_add(sum, mult);
    uint32_t *mine = partial + (sig * MSM_FOLD1_THREADS + g) * (size_t)BW;
    sum = pt_add(sum, load_bucket<F>(mine));                           // the scalars that are one
    store_bucket<F>(mine, sum);
    if (partial_plain) store_bucket<F>(partial_plain + (sig * MSM_FOLD1_THREADS + g) * (size_t)BW, run);
}
// Stage 2: 512 threads add `each` partial sums each, a tree through LDS, one inversion, ark-ff's bytes out
template <class F>
__global__ __launch_bounds__(MSM_FOLD_THREADS) void msm_fold2_kernel(const uint32_t *__restrict__ partial, int each,
                                                                     uint32_t *__restrict__ out /* [batch][ARK_WORDS], or [batch][BK_WORDS] */, int xyzz_out)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int SLOT = 4 * F::WORDS + 1, BW = Grp<F>::BK_WORDS;
    __shared__ uint32_t lds[MSM_FOLD_THREADS / 2 * SLOT];            // 58 KB (G1) / 116 KB (G2): the upper half of a tree level parks here
    const size_t sig = blockIdx.x;
    const int t = threadIdx.x;
    const uint32_t *src = partial + (sig * MSM_FOLD1_THREADS + (size_t)t * each) * BW;
    XyzzT<F> sum = load_bucket<F>(src);
    for (int k = 1; k < each; k++) sum = pt_add(sum, load_bucket<F>(src + (size_t)k * BW));
    for (int stride = MSM_FOLD_THREADS / 2; stride >= 1; stride >>= 1) {
        if (t >= stride && t < 2 * stride) store_bucket<F>(lds + (t - stride) * SLOT, sum);
        __syncthreads();
        if (t < stride) sum = pt_add(sum, load_bucket<F>(lds + t * SLOT));
        __syncthreads();
    }
    // xyzz_out: the sum as it stands, no inversion -- for a caller that goes on adding (frw_groth16_prove_dev)
    if (t == 0) {
        if (xyzz_out) store_bucket<F>(out + sig * BW, sum);
        else store_ark_point<F>(out + sig * Grp<F>::ARK_WORDS, pt_to_affine(sum));
    }
}

// (launch bounds of the narrow kernels: two waves per SIMD for G1; the two-lane G2 policy is written for ONE -- its mixed addition holds an
// accumulator, a row, the neighbour's copies of both operands of a product and 28 64-bit columns: 256 registers and then some, which at a
// bound of two went to AGPR moves and scratch under a warning that the target was missed.  Said so, the compiler may use the AGPRs it has.)
// ---- narrow windows: the witness-side sums ---------------------------------------------------------------------------------------
// The sums over a_query, b_g1_query, b_g2_query and l_query take a WITNESS as scalars: 46 % zeros, 45 % ones, and of the rest
// all but two thousand values are below 2^16 (Falcon-1024: 13 N values that are not boolean, 2 N of them ~146 bits).  Through
// the 16-bit pipeline above such a sum is 35,000 additions into 32,768 buckets -- and then 65,536 full additions to fold mostly
// empty buckets, ten kernels of one wavefront per SIMD each: a third of a whole proof's time for a thirtieth of its
// arithmetic.  With 8-bit windows (32 of them: a table twice as long) the same scalars are 70,000 additions into 128 buckets
// per signature, cut into ~2,048 equal work items, and the fold is one workgroup: a suffix scan and a tree over 128 points.
constexpr int NMSM_C = 8, NMSM_W = 32, NMSM_BUCKETS = 128;
// slices of the counting sort (a workgroup of 1,024 threads each): 16 per signature, 128 beyond 2^18 scalars (an aggregate statement is
// one "signature": 16 workgroups for 2.5 M scalars were 1.2 ms at the head of every other kernel of its proof)
// (1,024 slices for the 2^27-scalar sums were tried: the histogram no faster -- it waited for ONE global counter, see wave_append -- and the
// scatter three times slower: a slice's run inside a bucket is a 1,024th of it, and the entry array's lines left the L2 half written)
__host__ __device__ constexpr uint32_t nmsm_slices(uint32_t n) { return n > (1u << 18) ? 128u : 16u; }
// list[(*count)++] = value for the lanes of a wavefront that ask for it: ONE atomic per wavefront (the scalars equal to one are 55 % of a
// witness: 65 M increments of one address in the 1,024-statement proof, 20 ms of a histogram pass that computes for two)
__device__ __forceinline__ void wave_append(bool want, uint32_t *__restrict__ count, uint32_t *__restrict__ list, uint32_t value)
{
    const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
    if (!mask) return;
    const uint32_t lane = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));   // wanting lanes below this one
    const int leader = __builtin_ctzll(mask);
    uint32_t base = 0;
    if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(count, (uint32_t)__popcll(mask));
    base = __shfl(base, leader, 64);
    if (want) list[base + lane] = value;
}
// Work items per signature: 2,048 for the per-signature circuits (n < 2^18 points), n / 128 rounded up to a multiple of 2,048 beyond
// (an aggregate statement is ONE "signature" with sixteen times the points: 2,048 items would be 32 wavefronts on 1,024 SIMDs, each
// a chain of 550 additions); items <= buckets + total / split <= 128 + target, rounded up to whole wavefronts.
constexpr int NMSM_TARGET_ITEMS_MIN = 2048, NMSM_TARGET_ITEMS_MAX = 65536;
__host__ __device__ constexpr uint32_t nmsm_target_items(uint32_t n)
{
    const uint32_t t = (n / 128 + 2047) / 2048 * 2048;
    return t < (uint32_t)NMSM_TARGET_ITEMS_MIN ? (uint32_t)NMSM_TARGET_ITEMS_MIN : t > (uint32_t)NMSM_TARGET_ITEMS_MAX ? (uint32_t)NMSM_TARGET_ITEMS_MAX : t;
}
__host__ __device__ constexpr uint32_t nmsm_max_items(uint32_t n) { return nmsm_target_items(n) + 256; }
// the scalars equal to one: 256 .. 4,096 threads per signature, by the batch, one partial sum per 64 of them (<= 64: one per thread of
// the fold); with more than 2^18 points up to 65,536 threads, whose <= 1,024 partial sums a small kernel brings down to 64 first
constexpr int NMSM_ONES_MAX = 4096, NMSM_ONES_MAX_LARGE = 65536;
__host__ __device__ constexpr uint32_t nmsm_ones_max(uint32_t n) { return n > (1u << 18) ? (uint32_t)NMSM_ONES_MAX_LARGE : (uint32_t)NMSM_ONES_MAX; }
// threads of the one-workgroup fold: 128 for the buckets + 256 (G1) / 128 (G2: four waves = one per SIMD, the whole register file
// for an addition that needs 300 live registers) for the ones' partial sums
__device__ __forceinline__ bool scalar_digits8(const uint32_t *src, int montgomery, int (&d)[NMSM_W])
{
    const Fr8 w = scalar_canonical(src, montgomery);
    if (w.l[0] == 1u && !(w.l[1] | w.l[2] | w.l[3] | w.l[4] | w.l[5] | w.l[6] | w.l[7])) return true;
    int carry = 0;
#pragma unroll
    for (int j = 0; j < NMSM_W; j++) {
        int v = (int)((w.l[j >> 2] >> (8 * (j & 3))) & 0xffu) + carry;        // the top byte of a value below r is <= 0x73: no carry out
        carry = 0;
        if (j + 1 < NMSM_W && v > NMSM_BUCKETS) { v -= 1 << NMSM_C; carry = 1; }
        d[j] = v;
    }
    return false;
}
__device__ __forceinline__ uint32_t nmsm_split_of(uint32_t total, uint32_t target)
{
    const uint32_t s = (total + target - 1) / target;
    return s < 32u ? 32u : s;
}
__global__ __launch_bounds__(1024) void nmsm_hist_kernel(uint32_t n, const uint32_t *__restrict__ scalars, size_t sig_stride_words, int montgomery,
                                                         uint32_t *__restrict__ slice_hist /* [sig][slice][128] */,
                                                         uint32_t *__restrict__ ones_count /* [sig] */, uint32_t *__restrict__ ones_list /* [sig][n] */,
                                                         int ones_as_mask)
{
    __shared__ uint32_t hist[NMSM_BUCKETS];
    const size_t sig = blockIdx.y;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    if (threadIdx.x < NMSM_BUCKETS) hist[threadIdx.x] = 0;
    __syncthreads();
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        int d[NMSM_W];
        if (scalar_digits8(scalars + sig * sig_stride_words + (size_t)i * 8, montgomery, d)) {
            // the scalar is one: a bit of the signature's mask (zeroed by the caller; the first n / 32 words of its list), or an entry of its list
            if (ones_as_mask) atomicOr(&ones_list[sig * n + (i >> 5)], 1u << (i & 31));
            else ones_list[sig * n + atomicAdd(&ones_count[sig], 1u)] = i;
            continue;
        }
#pragma unroll
        for (int j = 0; j < NMSM_W; j++)
            if (d[j]) atomicAdd(&hist[(d[j] < 0 ? -d[j] : d[j]) - 1], 1u);
    }
    __syncthreads();
    if (threadIdx.x < NMSM_BUCKETS) slice_hist[(sig * slices + slice) * NMSM_BUCKETS + threadIdx.x] = hist[threadIdx.x];
}
// one workgroup of 128 threads per signature: bucket sizes and starts, every slice's first position inside its buckets, the
// work items (bucket | part << 8: equal parts of at most `split` entries)
__global__ __launch_bounds__(NMSM_BUCKETS) void nmsm_plan_kernel(uint32_t *__restrict__ slice_hist, uint32_t *__restrict__ counts, uint32_t *__restrict__ offsets,
                                                                 uint32_t *__restrict__ item_first, uint32_t *__restrict__ items /* [sig][max_items] */,
                                                                 uint32_t *__restrict__ item_count /* [sig] */, uint32_t target, uint32_t max_items, uint32_t slices)
{
    __shared__ uint32_t scan[NMSM_BUCKETS];
    const size_t sig = blockIdx.x;
    const int b = threadIdx.x;
    uint32_t *h = slice_hist + sig * slices * (size_t)NMSM_BUCKETS + b;
    uint32_t c = 0;
    for (uint32_t s_ = 0; s_ < slices; s_++) {
        const uint32_t v = h[(size_t)s_ * NMSM_BUCKETS];
        h[(size_t)s_ * NMSM_BUCKETS] = c;
        c += v;
    }
    auto inclusive_scan = [&](uint32_t mine) {
        scan[b] = mine;
        __syncthreads();
        for (int off = 1; off < NMSM_BUCKETS; off <<= 1) {
            const uint32_t v = b >= off ? scan[b - off] : 0u;
            __syncthreads();
            scan[b] += v;
            __syncthreads();
        }
        const uint32_t r = scan[b];
        __syncthreads();
        return r;
    };
    const uint32_t incl = inclusive_scan(c);
    scan[b] = incl;
    __syncthreads();
    const uint32_t total = scan[NMSM_BUCKETS - 1], split = nmsm_split_of(total, target);
    __syncthreads();
    counts[sig * NMSM_BUCKETS + b] = c;
    offsets[sig * NMSM_BUCKETS + b] = incl - c;
    const uint32_t k = c <= split ? 1u : (c + split - 1) / split;
    const uint32_t kincl = inclusive_scan(k), first = kincl - k;
    item_first[sig * NMSM_BUCKETS + b] = first;
    if (b == NMSM_BUCKETS - 1) item_count[sig] = kincl;
    for (uint32_t part = 0; part < k; part++) items[sig * (size_t)max_items + first + part] = (uint32_t)b | (part << 8);
}
__global__ __launch_bounds__(1024) void nmsm_scatter_kernel(uint32_t n, const uint32_t *__restrict__ scalars, size_t sig_stride_words, int montgomery,
                                                            const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ slice_hist,
                                                            uint32_t *__restrict__ entries /* [sig][32 n] */)
{
    __shared__ uint32_t cursor[NMSM_BUCKETS];
    const size_t sig = blockIdx.y;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    if (threadIdx.x < NMSM_BUCKETS)
        cursor[threadIdx.x] = offsets[sig * NMSM_BUCKETS + threadIdx.x] + slice_hist[(sig * slices + slice) * NMSM_BUCKETS + threadIdx.x];
    __syncthreads();
    uint32_t *ent = entries + sig * (size_t)NMSM_W * n;
    for (uint32_t i = lo + threadIdx.x; i < hi; i += 1024) {
        int d[NMSM_W];
        if (scalar_digits8(scalars + sig * sig_stride_words + (size_t)i * 8, montgomery, d)) continue;
#pragma unroll
        for (int j = 0; j < NMSM_W; j++) {
            if (!d[j]) continue;
            const uint32_t b = (uint32_t)(d[j] < 0 ? -d[j] : d[j]) - 1u;
            ent[atomicAdd(&cursor[b], 1u)] = ((uint32_t)j * n + i) | (d[j] < 0 ? 0x80000000u : 0u);
        }
    }
}
// ---- the narrow sums of a BARE handle (round 5: a key whose window tables would not fit; the points only) ------------------------------
// One scalar vector, thirty-two 8-bit windows: the counting sort is over (window, digit) -- 4,096 counters in 16 KB of LDS, ONE pass over
// the scalars for all the windows -- and from there on window w is what signature w is to the kernels above: its own 128 buckets, its own
// work items, its own fold; an entry is a point index (a table row: the tables have no other), a window's entries start at
// entry_base[w] (the windows' totals differ by orders of magnitude: the low two hold every 14-bit value of a witness, the upper
// thirteen nothing), and msm_horner_kernel puts the thirty-two window sums of every table together.  Scalars equal to one: a list,
// summed with window 0.
// `index` (or null): element k of the sum is scalar -- and table row -- index[k]: the rows of a table that are not the point at infinity
// (frw_groth16_pk.b_index: b_g1_query / b_g2_query hold a point for 59 % of the variables of a Falcon circuit, and of those variables'
// values half of a witness's ones and every full-size value are elsewhere: a third of the additions are left).
__global__ __launch_bounds__(1024) void nmsm_hist_bare_kernel(uint32_t n, const uint32_t *__restrict__ scalars, int montgomery,
                                                              uint32_t *__restrict__ slice_hist /* [window][slice][128] */,
                                                              uint32_t *__restrict__ ones_count /* [0] */, uint32_t *__restrict__ ones_list /* [n] */,
                                                              const uint32_t *__restrict__ index)
{
    // The scalars equal to one (55 % of a witness) go to ONE list with ONE counter: appended a wavefront at a time that was an atomic of
    // one address per wavefront and round of the loop -- 1.5 M of them in a row, 8 ns each: the whole 12 - 21 ms of this kernel, its waves
    // waiting 95 % of their cycles (SQ counters, tools/dev/pmc_sort.sh).  So a workgroup collects its ones in LDS (a counter there costs
    // nothing) and moves them out 3,072 at least at a time: one global atomic per flush, and the list is written in whole lines.
    constexpr uint32_t ONES_CAP = 4096;
    __shared__ uint32_t hist[NMSM_W * NMSM_BUCKETS];
    __shared__ uint32_t ones_buf[ONES_CAP];
    __shared__ uint32_t ones_fill, ones_base;
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    for (int b = threadIdx.x; b < NMSM_W * NMSM_BUCKETS; b += 1024) hist[b] = 0;
    if (threadIdx.x == 0) ones_fill = 0;
    __syncthreads();
    auto flush = [&](uint32_t fill) {                                  // (every thread of the workgroup with the same `fill`; nobody appends meanwhile)
        if (threadIdx.x == 0 && fill) ones_base = atomicAdd(&ones_count[0], fill);
        __syncthreads();
        for (uint32_t t = threadIdx.x; t < fill; t += 1024) ones_list[ones_base + t] = ones_buf[t];
        __syncthreads();
        if (threadIdx.x == 0) ones_fill = 0;
        __syncthreads();
    };
    for (uint32_t k0 = lo; k0 < hi; k0 += 1024) {                      // (the same number of rounds for every thread: barriers inside)
        const uint32_t k = k0 + threadIdx.x;
        bool one = false;
        if (k < hi) {
            const uint32_t i = index ? index[k] : k;
            int d[NMSM_W];
            one = scalar_digits8(scalars + (size_t)i * 8, montgomery, d);
            if (one) {
                ones_buf[atomicAdd(&ones_fill, 1u)] = i;               // (at most 1,024 a round, and a round starts with 3,072 free)
            } else {
#pragma unroll
                for (int j = 0; j < NMSM_W; j++)
                    if (d[j]) atomicAdd(&hist[j * NMSM_BUCKETS + (d[j] < 0 ? -d[j] : d[j]) - 1], 1u);
            }
        }
        __syncthreads();
        const uint32_t fill = ones_fill;                               // (read between two barriers: the same value in every thread)
        __syncthreads();
        if (fill > ONES_CAP - 1024) flush(fill);
    }
    flush(ones_fill);
    __syncthreads();
    for (int b = threadIdx.x; b < NMSM_W * NMSM_BUCKETS; b += 1024)
        slice_hist[((size_t)(b / NMSM_BUCKETS) * slices + slice) * NMSM_BUCKETS + (b % NMSM_BUCKETS)] = hist[b];
}
// where each window's entries start: the running total of the windows before it (64 bits: thirty-two windows of 2^27 scalars can be 2^32 entries)
__global__ __launch_bounds__(64) void nmsm_entry_base_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
                                                             unsigned long long *__restrict__ entry_base /* [32] */)
{
    if (threadIdx.x) return;
    unsigned long long run = 0;
    for (int w = 0; w < NMSM_W; w++) {
        entry_base[w] = run;
        run += (unsigned long long)offsets[w * NMSM_BUCKETS + NMSM_BUCKETS - 1] + counts[w * NMSM_BUCKETS + NMSM_BUCKETS - 1];
    }
}
__global__ __launch_bounds__(1024) void nmsm_scatter_bare_kernel(uint32_t n, const uint32_t *__restrict__ scalars, int montgomery,
                                                                 const uint32_t *__restrict__ offsets /* [window][128] */,
                                                                 const uint32_t *__restrict__ slice_hist,
                                                                 const unsigned long long *__restrict__ entry_base, uint32_t *__restrict__ entries,
                                                                 const uint32_t *__restrict__ index)
{
    __shared__ uint32_t cursor[NMSM_W * NMSM_BUCKETS];
    __shared__ unsigned long long base[NMSM_W];
    const uint32_t slices = gridDim.x, slice = blockIdx.x, per = (n + slices - 1) / slices;
    const uint32_t lo = slice * per, hi = lo + per < n ? lo + per : n;
    for (int b = threadIdx.x; b < NMSM_W * NMSM_BUCKETS; b += 1024)
        cursor[b] = offsets[b] + slice_hist[((size_t)(b / NMSM_BUCKETS) * slices + slice) * NMSM_BUCKETS + (b % NMSM_BUCKETS)];
    if (threadIdx.x < NMSM_W) base[threadIdx.x] = entry_base[threadIdx.x];
    __syncthreads();
    for (uint32_t k = lo + threadIdx.x; k < hi; k += 1024) {
        const uint32_t i = index ? index[k] : k;
        int d[NMSM_W];
        if (scalar_digits8(scalars + (size_t)i * 8, montgomery, d)) continue;
#pragma unroll
        for (int j = 0; j < NMSM_W; j++) {
            if (!d[j]) continue;
            const uint32_t b = (uint32_t)(d[j] < 0 ? -d[j] : d[j]) - 1u;
            entries[base[j] + atomicAdd(&cursor[j * NMSM_BUCKETS + b], 1u)] = i | (d[j] < 0 ? 0x80000000u : 0u);
        }
    }
}
// one thread per (signature, item): the sum of the item's table rows
template <class F, bool PREFETCH>
__global__ __launch_bounds__(64, F::LANES > 1 ? 1 : 2) void nmsm_bucket_kernel(NmsmTables tables, const uint32_t *__restrict__ offsets, const uint32_t *__restrict__ counts,
                                                            const uint32_t *__restrict__ items, const uint32_t *__restrict__ item_count,
                                                            const uint32_t *__restrict__ entries, uint32_t *__restrict__ partial_items,
                                                            uint32_t target, uint32_t max_items,
                                                            const unsigned long long *__restrict__ entry_base /* a bare handle's windows; else null */)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int PW = Grp<F>::PT_WORDS;
    const size_t slot = blockIdx.y;
    size_t sig;
    const MsmDev m = tables.of(blockIdx.y, sig);
    const uint32_t it = (blockIdx.x * 64 + threadIdx.x) / F::LANES;            // F::LANES adjacent lanes share an item (the two-lane Fq2)
    if (it >= item_count[sig]) return;
    const uint32_t item = items[sig * (size_t)max_items + it], b = item & (NMSM_BUCKETS - 1), part = item >> 8;
    const uint32_t c = counts[sig * NMSM_BUCKETS + b];
    const uint32_t total = offsets[sig * NMSM_BUCKETS + NMSM_BUCKETS - 1] + counts[sig * NMSM_BUCKETS + NMSM_BUCKETS - 1], split = nmsm_split_of(total, target);
    const uint32_t k = c <= split ? 1u : (c + split - 1) / split;
    const uint32_t lo = (uint32_t)((uint64_t)c * part / k), cnt = (uint32_t)((uint64_t)c * (part + 1) / k) - lo;     // equal parts
    const uint32_t *ent = entries + (entry_base ? (size_t)entry_base[sig] : sig * (size_t)NMSM_W * m.n) + offsets[sig * NMSM_BUCKETS + b] + lo;
    XyzzT<F> acc = pt_identity<F>();
    if (PREFETCH) {
        uint32_t e = cnt ? ent[0] : 0u;
        AffineT<F> p = load_row<F>(m.table + (size_t)(e & 0x7fffffffu) * PW);
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t e_next = j + 1 < cnt ? ent[j + 1] : e;
            const AffineT<F> p_next = load_row<F>(m.table + (size_t)(e_next & 0x7fffffffu) * PW);
            if (e >> 31) p.y = F::template neg<Grp<F>::K_AFFINE_Y>(p.y);
            acc = pt_add_affine(acc, p);
            e = e_next;
            p = p_next;
        }
    } else {
        for (uint32_t j = 0; j < cnt; j++) {
            const uint32_t e = ent[j];
            AffineT<F> p = load_row<F>(m.table + (size_t)(e & 0x7fffffffu) * PW);
            if (e >> 31) p.y = F::template neg<Grp<F>::K_AFFINE_Y>(p.y);
            acc = pt_add_affine(acc, p);
        }
    }
    store_bucket<F>(partial_items + (slot * (size_t)max_items + it) * (size_t)Grp<F>::BK_WORDS, acc);
}
// the points whose scalar is one: thread t of `gridDim.x * 64` per signature takes every such-th of the list, the 64 sums of a
// workgroup are added up through LDS (six steps instead of the 64 / 128-fold serial addition they would cost the fold), and one
// partial sum per workgroup goes out: gridDim.x <= 64 of them per signature
template <class F>
__global__ __launch_bounds__(64, F::LANES > 1 ? 1 : 2) void nmsm_ones_kernel(NmsmTables tables, const uint32_t *__restrict__ ones_count, const uint32_t *__restrict__ ones_list,
                                                          uint32_t *__restrict__ partial_ones /* [blockIdx.y][groups_stride][BK_WORDS] */, uint32_t groups_stride,
                                                          uint32_t row_step /* 1; a bare handle's tables: 32 -- grid row y is (table y, window 0), the only window with ones */)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int SLOT = 4 * F::WORDS + 1, PER = 64 / F::LANES;                 // PER chains per workgroup (F::LANES lanes each)
    __shared__ uint32_t lds[32 * SLOT];
    const size_t slot = blockIdx.y;
    size_t sig;
    const MsmDev m = tables.of(blockIdx.y * row_step, sig);
    const uint32_t nthreads = gridDim.x * PER, t = blockIdx.x * PER + threadIdx.x / F::LANES;
    const uint32_t *list = ones_list + sig * m.n;
    XyzzT<F> acc = pt_identity<F>();
    if (m.ones_table) {
        // `list` is the mask of ones: one addition per group of eight points whose byte of it is not zero
        const uint32_t groups = (m.n + 7) / 8;
        for (uint32_t g = t; g < groups; g += nthreads) {
            const uint32_t byte = (list[g >> 2] >> (8 * (g & 3))) & 0xffu;
            if (byte) acc = pt_add_affine(acc, load_row<F>(m.ones_table + ((size_t)g * 255 + (byte - 1)) * Grp<F>::PT_WORDS));
        }
    } else {
        const uint32_t cnt = ones_count[sig];
        for (uint32_t k = t; k < cnt; k += nthreads)
            acc = pt_add_affine(acc, load_row<F>(m.table + (size_t)list[k] * Grp<F>::PT_WORDS));  // window 0 of the table = the point itself
    }
    const int lane = threadIdx.x / F::LANES;
    for (int stride = PER / 2; stride >= 1; stride >>= 1) {
        if (lane >= stride && lane < 2 * stride) store_bucket<F>(lds + (lane - stride) * SLOT, acc);
        __syncthreads();
        if (lane < stride) acc = pt_add(acc, load_bucket<F>(lds + lane * SLOT));
        __syncthreads();
    }
    if (lane == 0) store_bucket<F>(partial_ones + (slot * (size_t)groups_stride + blockIdx.x) * (size_t)Grp<F>::BK_WORDS, acc);
}
// the policy a group's narrow sum runs with: G2 in two lanes per point (frw_fq29.h Fq2PairField) -- the layout in memory is that of
// either G2 policy; the window tables and the dense pipeline keep the one-lane form
template <class F> struct BulkPolicy { typedef F type; };
template <> struct BulkPolicy<Fq2Field> { typedef Fq2PairField type; };
// more than 64 partial sums of ones (sums over more than 2^18 points): workgroup g of 64 adds up the partial sums g, g + 64, ... of its
// signature -- one per lane, then a tree through LDS -- and leaves sum g of 64 in the first 64 slots of the second stage's array
template <class F>
__global__ __launch_bounds__(64, F::LANES > 1 ? 1 : 2) void nmsm_ones_fold_kernel(const uint32_t *__restrict__ partial_ones, uint32_t groups, uint32_t groups_stride,
                                                               uint32_t *__restrict__ folded /* [sig][64][BK_WORDS] */)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int SLOT = 4 * F::WORDS + 1, BW = Grp<F>::BK_WORDS, PER = 64 / F::LANES;   // one wavefront: PER chains of F::LANES lanes
    __shared__ uint32_t lds[32 * SLOT];
    const size_t sig = blockIdx.y;
    const uint32_t g = blockIdx.x, lane = threadIdx.x / F::LANES;
    const uint32_t *src = partial_ones + sig * (size_t)groups_stride * BW;
    XyzzT<F> acc = pt_identity<F>();
    for (uint32_t k = g + 64 * lane; k < groups; k += 64 * PER) acc = pt_add(acc, load_bucket<F>(src + (size_t)k * BW));
    for (int stride = PER / 2; stride >= 1; stride >>= 1) {
        if ((int)lane >= stride && (int)lane < 2 * stride) store_bucket<F>(lds + (lane - stride) * SLOT, acc);
        __syncthreads();
        if ((int)lane < stride) acc = pt_add(acc, load_bucket<F>(lds + lane * SLOT));
        __syncthreads();
    }
    if (lane == 0) store_bucket<F>(folded + (sig * 64 + g) * (size_t)BW, acc);
}
// bucket b = the sum of its items, by a workgroup of its own: the buckets of small digits hold many times the mean (the high
// bytes of 14-bit values fall into 48 of them), and fifty items added up by one thread were the longest chain of the whole sum
template <class F>
__global__ __launch_bounds__(64, F::LANES > 1 ? 1 : 2) void nmsm_combine_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
                                                             const uint32_t *__restrict__ item_first, const uint32_t *__restrict__ partial_items,
                                                             uint32_t *__restrict__ buckets /* [slot][128][BK_WORDS] */, uint32_t target, uint32_t max_items,
                                                             uint32_t sigs)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int SLOT = 4 * F::WORDS + 1, BW = Grp<F>::BK_WORDS, PER = 64 / F::LANES;
    __shared__ uint32_t lds[32 * SLOT];
    const size_t slot = blockIdx.y, sig = slot % sigs;
    const uint32_t b = blockIdx.x, lane = threadIdx.x / F::LANES;
    const uint32_t c = counts[sig * NMSM_BUCKETS + b];
    const uint32_t total = offsets[sig * NMSM_BUCKETS + NMSM_BUCKETS - 1] + counts[sig * NMSM_BUCKETS + NMSM_BUCKETS - 1], split = nmsm_split_of(total, target);
    const uint32_t k = c <= split ? 1u : (c + split - 1) / split;
    const uint32_t *src = partial_items + (slot * (size_t)max_items + item_first[sig * NMSM_BUCKETS + b]) * (size_t)BW;
    XyzzT<F> acc = pt_identity<F>();
    for (uint32_t j = lane; j < k; j += PER) acc = pt_add(acc, load_bucket<F>(src + (size_t)j * BW));
    for (int stride = PER / 2; stride >= 1; stride >>= 1) {
        if ((int)lane >= stride && (int)lane < 2 * stride) store_bucket<F>(lds + (lane - stride) * SLOT, acc);
        __syncthreads();
        if ((int)lane < stride) acc = pt_add(acc, load_bucket<F>(lds + lane * SLOT));
        __syncthreads();
    }
    if (lane == 0) store_bucket<F>(buckets + (slot * NMSM_BUCKETS + b) * (size_t)BW, acc);
}

// ONE WAVEFRONT per signature folds the 128 buckets and the ones' partial sums into the result.  (The first version was one
// workgroup of 384 / 256 threads: a workgroup of several wavefronts that is issued while the 2^18-point bucket kernel occupies the
// chip -- two wavefronts of 240 registers on every SIMD -- finds no CU with room for all of it until that kernel ends; 42 ms late in
// profiles/r04_groth16_timeline_64_starved.txt.  A single wavefront takes whatever slot frees up next.)
// PER = 64 / F::LANES chains, NB = 128 / PER adjacent buckets each; bucket b = i NB + k has weight b + 1 = i NB + (k + 1):
//     result = sum_i [ sum_k (k + 1) B_(i,k) ] + NB sum_(i >= 1) Suffix_i + ones,   Suffix_i = sum_(j >= i) sum_k B_(j,k)
//   phase 0   chain i adds up its buckets from the last to the first (their items one after the other, unless nmsm_combine_kernel has
//             done that), parking run_k = B_k + .. + B_(NB-1) after each: run_0 is the chain's total, sum_k run_k its weighted sum
//   phase 1   suffix scan of the totals over the chains (log2 PER steps through LDS)
//   phase 2   x NB (the value added to itself log2 NB times; chain 0 contributes nothing), + the chain's weighted sum, + the ones
//   phase 3   tree over the chains, one inversion, ark-ff's bytes out
// ONE addition site: every step is "sum (+)= the point at p", p in global memory or in LDS.
template <class F>
__global__ __launch_bounds__(64) void nmsm_finish_kernel(const uint32_t *__restrict__ counts, const uint32_t *__restrict__ offsets,
                                                         const uint32_t *__restrict__ item_first, const uint32_t *__restrict__ partial_items,
                                                         const uint32_t *__restrict__ buckets /* [sig][128][BK_WORDS], or null */,
                                                         const uint32_t *__restrict__ partial_ones /* [sig][ones_stride][BK_WORDS] */,
                                                         int ones_groups, uint32_t ones_stride, uint32_t target, uint32_t max_items,
                                                         uint32_t *__restrict__ out /* [slot][ARK_WORDS], or [slot][BK_WORDS] */, int xyzz_out,
                                                         uint32_t sigs, int ones_window0 /* a bare handle: rows are (table, window); the ones' sums are [table] and go with window 0 */)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    constexpr int SLOT = 4 * F::WORDS + 1, BW = Grp<F>::BK_WORDS, PER = 64 / F::LANES, NB = NMSM_BUCKETS / PER;
    constexpr int LOG_PER = PER == 64 ? 6 : 5, LOG_NB = NB == 2 ? 1 : 2;
    static_assert(PER * NB == NMSM_BUCKETS && (1 << LOG_PER) == PER && (1 << LOG_NB) == NB, "64 or 32 chains of 2 or 4 buckets");
    __shared__ uint32_t lds[NMSM_BUCKETS * SLOT];                     // 29 KB (G1) / 58 KB (G2): slot i NB + k
    __shared__ uint32_t longest[NB];
    const size_t slot = blockIdx.x, sig = slot % sigs;
    const int i = threadIdx.x / F::LANES;
    // the items of the chain's buckets (one each when they come combined)
    const uint32_t *src[NB];
    uint32_t mine[NB];
    if (threadIdx.x < NB) longest[threadIdx.x] = 1;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < NB; k++) {
        const int b = i * NB + k;
        if (buckets) {
            src[k] = buckets + (slot * NMSM_BUCKETS + b) * (size_t)BW;
            mine[k] = 1;
        } else {
            const uint32_t c = counts[sig * NMSM_BUCKETS + b];
            const uint32_t total = offsets[sig * NMSM_BUCKETS + NMSM_BUCKETS - 1] + counts[sig * NMSM_BUCKETS + NMSM_BUCKETS - 1], split = nmsm_split_of(total, target);
            mine[k] = c <= split ? 1u : (c + split - 1) / split;
            src[k] = partial_items + (slot * (size_t)max_items + item_first[sig * NMSM_BUCKETS + b]) * (size_t)BW;
            atomicMax(&longest[k], mine[k]);
        }
    }
    __syncthreads();
    uint32_t phase0 = 0;
#pragma unroll
    for (int k = 0; k < NB; k++) phase0 += longest[k] + 1;            // the items of bucket k, then the step that parks run_k
    const int ones_steps = (ones_groups + PER - 1) / PER;
    const uint32_t at_local = phase0, at_reload = at_local + (NB - 1) + 1, at_scan = at_reload + 1, at_weight = at_scan + LOG_PER,
                   at_final = at_weight + LOG_NB, at_ones = at_final + 1, at_tree = at_ones + ones_steps, steps = at_tree + LOG_PER;
    XyzzT<F> sum = pt_identity<F>();
    for (uint32_t step = 0; step < steps; step++) {
        int store_slot = -1;
        bool reset = false, add = false;
        const uint32_t *p = lds;
        if (step < phase0) {
            // bucket k = NB - 1 .. 0
            uint32_t s = step;
            int k = NB - 1;
#pragma unroll
            for (int kk = NB - 1; kk >= 1; kk--)
                if (k == kk && s >= longest[kk] + 1) { s -= longest[kk] + 1; k = kk - 1; }
            const uint32_t *sp = src[0];
            uint32_t mk = mine[0], lk = longest[0];
#pragma unroll
            for (int kk = 1; kk < NB; kk++)
                if (k == kk) { sp = src[kk]; mk = mine[kk]; lk = longest[kk]; }     // (selects: a register array indexed at run time would live in scratch)
            if (s < lk) { add = s < mk; p = sp + (size_t)s * BW; }
            else store_slot = i * NB + k;                              // run_k
        } else if (step < at_reload - 1) {                             // weighted sum: run_0 (held) + run_1 + .. + run_(NB-1)
            add = true; p = lds + (i * NB + 1 + (int)(step - at_local)) * SLOT;
        } else if (step == at_reload - 1) {
            store_slot = i * NB + 1;                                   // the chain's weighted sum parks where run_1 was
        } else if (step == at_reload) {
            reset = true; add = true; p = lds + (i * NB) * SLOT;       // the chain's total again
        } else if (step < at_weight) {                                 // suffix scan, offset 1, 2, .., PER / 2
            const int off = 1 << (step - at_scan);
            store_slot = i * NB;
            add = i + off < PER; p = lds + ((i + off) * NB) * SLOT;
        } else if (step < at_final) {                                  // x 2, log2 NB times
            store_slot = i * NB;
            add = true; p = lds + (i * NB) * SLOT;
        } else if (step == at_final) {
            reset = i == 0;                                            // weight i NB: nothing from the first chain
            add = true; p = lds + (i * NB + 1) * SLOT;
        } else if (step < at_tree) {
            const int g = i + PER * (int)(step - at_ones);
            const size_t ones_row = ones_window0 ? slot / sigs : slot;
            add = g < ones_groups && !(ones_window0 && sig != 0); p = partial_ones + (ones_row * (size_t)ones_stride + (size_t)g) * BW;
        } else {
            const int stride = PER >> (step - at_tree + 1);
            if (i >= stride && i < 2 * stride) store_slot = (i - stride) * NB;
            add = i < stride; p = lds + (i * NB) * SLOT;
        }
        if (store_slot >= 0) store_bucket<F>(lds + store_slot * SLOT, sum);
        __syncthreads();
        if (reset) sum = pt_identity<F>();
        if (add) sum = pt_add(sum, load_bucket<F>(p));
        __syncthreads();
    }
    if (i == 0) {
        if (xyzz_out) store_bucket<F>(out + slot * BW, sum);         // as msm_fold2_kernel
        else store_ark_point<F>(out + slot * Grp<F>::ARK_WORDS, pt_to_affine(sum));
    }
}

// ---- k G for many scalars: the FixedBaseMSM of ark-groth16's generator (generator.rs builds h_query, a_query, l_query and
// b_g2_query this way from the toxic waste).  8-bit windows of the generator: table[w][d] = d 2^(8 w) G, then 32 mixed additions.
constexpr int FB_WINDOWS = 32, FB_DIGITS = 256;
template <class F> __device__ __forceinline__ AffineT<F> group_generator();
template <> __device__ __forceinline__ AffineT<FqField> group_generator<FqField>()
{
    AffineT<FqField> g;
    g.x = fq_const(G1_GEN_X29); g.y = fq_const(G1_GEN_Y29); g.inf = false;
    return g;
}
template <> __device__ __forceinline__ AffineT<Fq2Field> group_generator<Fq2Field>()
{
    AffineT<Fq2Field> g;
    g.x.c0 = fq_const(G2_GEN_X0_29); g.x.c1 = fq_const(G2_GEN_X1_29); g.y.c0 = fq_const(G2_GEN_Y0_29); g.y.c1 = fq_const(G2_GEN_Y1_29);
    g.inf = false;
    return g;
}
template <class F>
__global__ __launch_bounds__(64) void fixed_base_table_kernel(uint32_t *__restrict__ table /* [32][256][PT_WORDS] */)
{
    const int id = blockIdx.x * 64 + threadIdx.x, w = id >> 8, d = id & 255;
    if (w >= FB_WINDOWS) return;
    const AffineT<F> g = group_generator<F>();
    XyzzT<F> acc = pt_identity<F>();
    for (int bit = 7; bit >= 0; bit--) {                      // d G
        acc = pt_double(acc);
        if ((d >> bit) & 1) acc = pt_add_affine(acc, g);
    }
    for (int k = 0; k < 8 * w; k++) acc = pt_double(acc);     // 2^(8 w) d G
    store_row<F>(table + (size_t)id * Grp<F>::PT_WORDS, pt_to_affine(acc));
}
template <class F>
__global__ __launch_bounds__(64) void fixed_base_kernel(size_t count, const uint32_t *__restrict__ scalars /* [count][8], canonical */,
                                                        const uint32_t *__restrict__ table, uint32_t *__restrict__ out /* [count][ARK_WORDS] */)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const Fr8 k = fr_load(scalars + i * 8);
    XyzzT<F> acc = pt_identity<F>();
    for (int w = 0; w < FB_WINDOWS; w++) {
        const uint32_t d = (k.l[w >> 2] >> (8 * (w & 3))) & 255u;
        if (d) acc = pt_add_affine(acc, load_row<F>(table + ((size_t)w * FB_DIGITS + d) * Grp<F>::PT_WORDS));
    }
    store_ark_point<F>(out + i * Grp<F>::ARK_WORDS, pt_to_affine(acc));
}
// ---- the same for a key made on the device (frw_groth16_setup_r1cs_opts: 5 x 10^8 G1 and 1.2 x 10^8 G2 multiples for the 1,024-statement
// aggregate): SIXTEEN-bit windows -- table[w][d] = d 2^(16 w) G, 2^20 rows (117 MB, G2 235 MB), sixteen mixed additions per scalar instead of
// thirty-two -- G2 with two lanes per point (with one lane and 3.8 KB of scratch the 123 M rows of b_g2_query took 85 of the first version's
// 94 seconds), and the multiples written as table rows (29-bit limbs, what load_row reads) straight into a bare handle: a 2^27-point
// h_query never exists as ark-ff bytes anywhere.
constexpr int FBW_WINDOWS = 16, FBW_DIGITS = 65536;
// B_w = 2^(16 w) G, w < 16, one after the other (240 doublings of one chain)
template <class F>
__global__ __launch_bounds__(64) void fixed_base_window_gens_kernel(uint32_t *__restrict__ gens /* [16][PT_WORDS] */)
{
    if (threadIdx.x || blockIdx.x) return;
    XyzzT<F> acc = pt_from_affine(group_generator<F>());
    for (int w = 0; w < FBW_WINDOWS; w++) {
        store_row<F>(gens + (size_t)w * Grp<F>::PT_WORDS, pt_to_affine(acc));
        for (int k = 0; k < 16; k++) acc = pt_double(acc);
    }
}
template <class F>
__global__ __launch_bounds__(64) void fixed_base_wide_table_kernel(const uint32_t *__restrict__ gens, uint32_t *__restrict__ table /* [16][65536][PT_WORDS] */)
{
    const uint32_t id = blockIdx.x * 64 + threadIdx.x, w = id >> 16, d = id & 65535u;
    if (w >= (uint32_t)FBW_WINDOWS) return;
    const AffineT<F> g = load_row<F>(gens + (size_t)w * Grp<F>::PT_WORDS);
    XyzzT<F> acc = pt_identity<F>();
    for (int bit = 15; bit >= 0; bit--) {
        acc = pt_double(acc);
        if ((d >> bit) & 1u) acc = pt_add_affine(acc, g);
    }
    store_row<F>(table + (size_t)id * Grp<F>::PT_WORDS, pt_to_affine(acc));
}
// F: FqField, or Fq2PairField (two adjacent lanes per scalar).  ARK_OUT: ark-ff's affine bytes (the verifying key's points) instead of rows.
template <class F, bool ARK_OUT>
__global__ __launch_bounds__(64) void fixed_base_wide_kernel(size_t count, const uint32_t *__restrict__ scalars /* [count][8], canonical */,
                                                             const uint32_t *__restrict__ table, uint32_t *__restrict__ out)
{
    const size_t i = ((size_t)blockIdx.x * 64 + threadIdx.x) / F::LANES;
    if (i >= count) return;
    const Fr8 k = fr_load(scalars + i * 8);
    XyzzT<F> acc = pt_identity<F>();
    for (int w = 0; w < FBW_WINDOWS; w++) {
        const uint32_t d = (k.l[w >> 1] >> (16 * (w & 1))) & 65535u;
        if (d) acc = pt_add_affine(acc, load_row<F>(table + ((size_t)w * FBW_DIGITS + d) * Grp<F>::PT_WORDS));
    }
    if (ARK_OUT) store_ark_point<F>(out + i * Grp<F>::ARK_WORDS, pt_to_affine(acc));
    else store_row<F>(out + i * Grp<F>::PT_WORDS, pt_to_affine(acc));
}

}  // namespace frw

#if !defined(FRW_MSM_PROBE)     // (tools/kernel_resources.py compiles single instantiations of the kernels above through a probe unit)
// ---- C ABI ------------------------------------------------------------------------------------------------------------------------
struct frw_msm {
    int device;
    int group;                  // 1: G1, 2: G2
    int window_bits;            // 16: the dense pipeline (32,768 buckets), 8: the narrow one (128 buckets)
    bool bare = false;          // the table is the points themselves (one row each): the sums run window by window (round 5)
    bool wide = false;          // ... a dense bare handle on thirteen 20-bit windows (frw::WIDE_*): from 2^23 points, or asked for at load time
    uint64_t row_lo = 0;        // a slice of a sharded key: the index of row 0 in the whole query (the caller offsets the scalars by it)
    frw::MsmDev dev;
    void *table;
    void *ones_table;           // narrow handles of up to 2^18 points (MsmDev::ones_table), else null
};

namespace {
using frw::FqField;
using frw::Fq2Field;
// ---- the dense pipeline's workspace: per grid row (a signature; a bare handle: a window), every array but the last two a multiple of
// four words per row, so that whatever the number of points everything before them is 16-byte aligned (the items are read two words at a time)
template <class F> struct MsmBufs {
    uint32_t *counts, *offsets, *order, *item_first, *item_count, *ones_count, *items, *buckets, *partial, *partial_items, *entries, *ones_list, *end;
    int16_t *digits;            // a bare handle: int16[16][n], the windows' digits of the scalar vector being summed
    uint32_t max_items;
    size_t ent_stride, ones_stride;
};
template <class F> MsmBufs<F> msm_carve(void *ws, size_t rows, uint32_t n, bool bare)
{
    constexpr size_t BW = frw::Grp<F>::BK_WORDS;
    static_assert((size_t)frw::MSM_SLICES * 4 <= BW * 4, "the slice histograms borrow the buckets' memory");
    static_assert((size_t)frw::MSM_SLICES_LONE * frw::MSM_BUCKETS <= (size_t)frw::MSM_MAX_ITEMS * BW, "... or, for a lone signature, the work items'");
    MsmBufs<F> b;
    b.max_items = frw::msm_max_items(n, bare);
    b.ent_stride = bare ? (size_t)n : (size_t)frw::MSM_W * n;
    b.ones_stride = bare ? 0 : (size_t)n;                               // a bare handle's ones are digits like any other
    b.counts = (uint32_t *)ws;
    b.offsets = b.counts + rows * frw::MSM_BUCKETS;
    b.order = b.offsets + rows * frw::MSM_BUCKETS;
    b.item_first = b.order + rows * frw::MSM_BUCKETS;
    b.item_count = b.item_first + rows * frw::MSM_BUCKETS;             // [rows], four words each
    b.ones_count = b.item_count + rows * 4;                            // likewise
    b.items = b.ones_count + rows * 4;                                 // [rows][max_items][2]
    b.buckets = b.items + rows * (size_t)b.max_items * 2;              // the sort's per-slice histograms live here before the buckets are written
    b.partial = b.buckets + rows * (size_t)frw::MSM_BUCKETS * BW;
    b.partial_items = b.partial + rows * (size_t)frw::MSM_FOLD1_THREADS * BW;
    b.entries = b.partial_items + rows * (size_t)b.max_items * BW;
    b.ones_list = b.entries + rows * b.ent_stride;
    b.end = b.ones_list + rows * b.ones_stride;
    b.digits = nullptr;
    if (bare) {
        b.digits = (int16_t *)b.end;
        b.end += ((size_t)frw::MSM_W * n * 2 + 3) / 4;
    }
    return b;
}
// Wide windows (frw::WIDE_*): bare dense handles of 2^26 points and more, and those LOADED so (frw_msm_g1_load_bare(.., narrow = 2, ..): the
// tests do, on the small adversarial vectors and on 2^18 points).  Measured on the 2^27-point sum of the 1,024-statement aggregate
// (profiles/r05_wide_windows_ab.txt): the bucket kernel 350 -> 308 ms, the sort 65 -> 34 ms once every pass of it wrote runs (the first
// version's scatter into 208 x 32,768 open cache lines took 76 ms and gave the gain back), the proof 683 - 698 -> 613 ms.
constexpr uint32_t MSM_WIDE_FROM_DEFAULT = (1u << 26) - 1;           // (h_query of the 2^26 domain has 2^26 - 1 points)
static uint32_t msm_wide_from()
{
    static const uint32_t v = [] {
        const char *e = getenv("FRW_BARE_WIDE_FROM_LOG2");             // (measurement switch: tools/time_aggregate_large.py)
        const int lg = e ? atoi(e) : 0;
        return lg >= 10 && lg <= 31 ? ((uint32_t)1 << lg) - 1 : MSM_WIDE_FROM_DEFAULT;
    }();
    return v;
}
template <class F> struct MsmWideBufs {
    MsmBufs<F> rows;                    // the 208 rows' arrays (entries apart: below)
    uint32_t *partial_plain, *slice_hist, *bin_count, *row_count, *s1, *s0, *window_sums, *entries, *end;
    unsigned long long *row_start, *bin_start;
    uint2 *coarse;
    int32_t *digits;
};
template <class F> MsmWideBufs<F> msm_carve_wide(void *ws, uint32_t n)
{
    constexpr size_t BW = frw::Grp<F>::BK_WORDS, R = frw::WIDE_ROWS;
    MsmWideBufs<F> w;
    MsmBufs<F> &b = w.rows;
    b.max_items = frw::WIDE_MAX_ITEMS;
    b.ent_stride = 0; b.ones_stride = 0; b.digits = nullptr;
    b.counts = (uint32_t *)ws;
    b.offsets = b.counts + R * frw::MSM_BUCKETS;
    b.order = b.offsets + R * frw::MSM_BUCKETS;
    b.item_first = b.order + R * frw::MSM_BUCKETS;
    b.item_count = b.item_first + R * frw::MSM_BUCKETS;
    b.ones_count = b.item_count + R * 4;
    b.items = b.ones_count + R * 4;
    b.buckets = b.items + R * (size_t)b.max_items * 2;
    b.partial = b.buckets + R * (size_t)frw::MSM_BUCKETS * BW;
    w.partial_plain = b.partial + R * (size_t)frw::MSM_FOLD1_THREADS * BW;
    b.partial_items = w.partial_plain + R * (size_t)frw::MSM_FOLD1_THREADS * BW;
    w.s1 = b.partial_items + R * (size_t)b.max_items * BW;
    w.s0 = w.s1 + R * BW;
    w.window_sums = w.s0 + R * BW;
    w.row_start = (unsigned long long *)(w.window_sums + (size_t)frw::WIDE_W * BW);     // (every term so far a multiple of four words)
    w.bin_start = w.row_start + R;
    w.row_count = (uint32_t *)(w.bin_start + frw::WIDE_ALL_BINS);
    w.bin_count = w.row_count + R;
    w.slice_hist = w.bin_count + frw::WIDE_ALL_BINS;
    w.coarse = (uint2 *)(w.slice_hist + (size_t)frw::WIDE_W * frw::WIDE_SLICES * frw::WIDE_BINS);
    w.entries = (uint32_t *)(w.coarse + (size_t)frw::WIDE_W * n);
    w.digits = (int32_t *)w.entries;                  // (the digits are dead once the bins are sorted; the entries are written after that)
    w.end = w.entries + (size_t)frw::WIDE_W * n;
    b.entries = w.entries; b.ones_list = w.end; b.end = w.end;
    return w;
}
// a bare handle: sixteen rows (windows) and their sixteen sums
template <class F> size_t msm_workspace_per_signature(uint32_t n, bool bare = false, bool wide = false)
{
    if (wide) {
        char *const base0 = (char *)(uintptr_t)4096;
        return ((size_t)((char *)msm_carve_wide<F>(base0, n).end - base0) + 15) & ~(size_t)15;
    }
    const size_t rows = bare ? frw::MSM_W : 1;
    char *const base = (char *)(uintptr_t)4096;                       // (a carve of nothing: only the distance to its end is used)
    const size_t bytes = (size_t)((char *)msm_carve<F>(base, rows, n, bare).end - base) + (bare ? (size_t)frw::MSM_W * frw::Grp<F>::BK_WORDS * 4 : 0);
    return (bytes + 15) & ~(size_t)15;
}

// ---- the narrow pipeline's workspace -----------------------------------------------------------------------------------------------
struct NmsmBufs {
    uint32_t *slice_hist, *counts, *offsets, *item_first, *items, *item_count, *ones_count, *ones_list, *entries;      // the sort's
    uint32_t *partial_items, *partial_ones, *folded_ones, *bucket_sums;                                              // a table's own
    uint32_t target, max_items, ones_stride;
    uint32_t n = 0;             // (bare) how many scalars the sort is over: a table's rows, or the rows its index names
    // a bare handle's sort (rows = the thirty-two windows of one scalar vector): where every window's entries start, and its tables'
    // window sums [table][32]
    unsigned long long *entry_base = nullptr;
    uint32_t *window_sums = nullptr;
    uint32_t *end = nullptr;
};
template <class F> NmsmBufs nmsm_carve(void *d_workspace, size_t cnt, uint32_t n)
{
    constexpr int BW = frw::Grp<F>::BK_WORDS;
    NmsmBufs b;
    b.target = frw::nmsm_target_items(n); b.max_items = frw::nmsm_max_items(n); b.ones_stride = frw::nmsm_ones_max(n) / 64;
    b.slice_hist = (uint32_t *)d_workspace;
    b.counts = b.slice_hist + cnt * (size_t)frw::nmsm_slices(n) * frw::NMSM_BUCKETS;
    b.offsets = b.counts + cnt * frw::NMSM_BUCKETS;
    b.item_first = b.offsets + cnt * frw::NMSM_BUCKETS;
    b.items = b.item_first + cnt * frw::NMSM_BUCKETS;
    b.item_count = b.items + cnt * (size_t)b.max_items;                    // [cnt], padded to four words per signature in the budget
    b.ones_count = b.item_count + cnt * 4;                                 // likewise
    b.ones_list = b.ones_count + cnt * 4;
    b.entries = b.ones_list + cnt * (size_t)n;
    b.partial_items = b.entries + cnt * (size_t)frw::NMSM_W * n;           // 16-byte aligned: every term above is a multiple of 4 words per signature but n
    b.partial_items += (4 - ((uintptr_t)b.partial_items >> 2 & 3)) & 3;    // ... at most three words, out of the four the budget adds to the list for it
    b.partial_ones = b.partial_items + cnt * (size_t)b.max_items * BW;
    b.folded_ones = b.partial_ones + cnt * (size_t)b.ones_stride * BW;     // second stage, only when ones_stride > 64
    b.bucket_sums = b.folded_ones + (b.ones_stride > 64 ? cnt * (size_t)64 * BW : 0);
    b.end = b.bucket_sums + cnt * (size_t)frw::NMSM_BUCKETS * BW;
    return b;
}
// ... of a bare handle: ONE sort (thirty-two window rows) and the own arrays of `tables` tables.  The entries are budgeted for the worst
// case, every digit of every scalar non-zero (32 n words); a witness fills a seventieth of that.
// `sorted`: another carve whose sort this one reads (the G2 sum of a proof reads the G1 sums' sort): only the own arrays are carved then.
template <class F> NmsmBufs nmsm_carve_bare(void *d_workspace, size_t tables, uint32_t n, const NmsmBufs *sorted = nullptr)
{
    constexpr int BW = frw::Grp<F>::BK_WORDS;
    constexpr size_t WIN = frw::NMSM_W;
    NmsmBufs b;
    uint32_t *p = (uint32_t *)d_workspace;
    if (sorted) {
        b = *sorted;
    } else {
        b.target = frw::nmsm_target_items(n); b.max_items = frw::nmsm_max_items(n); b.ones_stride = frw::nmsm_ones_max(n) / 64;
        b.n = n;
        b.entry_base = (unsigned long long *)d_workspace;                  // [32]
        b.slice_hist = (uint32_t *)(b.entry_base + WIN);
        b.counts = b.slice_hist + WIN * (size_t)frw::nmsm_slices(n) * frw::NMSM_BUCKETS;
        b.offsets = b.counts + WIN * frw::NMSM_BUCKETS;
        b.item_first = b.offsets + WIN * frw::NMSM_BUCKETS;
        b.items = b.item_first + WIN * frw::NMSM_BUCKETS;
        b.item_count = b.items + WIN * (size_t)b.max_items;
        b.ones_count = b.item_count + WIN * 4;                             // [0] is the one in use
        p = b.ones_count + 4;                                              // (every term so far a multiple of four words: 16-byte aligned)
    }
    b.partial_items = p;
    b.partial_ones = b.partial_items + tables * WIN * (size_t)b.max_items * BW;
    b.folded_ones = b.partial_ones + tables * (size_t)b.ones_stride * BW;
    b.bucket_sums = b.folded_ones + tables * (size_t)64 * BW;
    b.window_sums = b.bucket_sums + tables * WIN * (size_t)frw::NMSM_BUCKETS * BW;
    b.end = b.window_sums + tables * WIN * BW;
    if (!sorted) {
        b.ones_list = b.end;
        b.entries = b.ones_list + (size_t)n;
        b.end = b.entries + WIN * (size_t)n;
    }
    return b;
}
template <class F> size_t nmsm_workspace_per_signature(uint32_t n, bool bare = false)
{
    // slice histograms, counts, offsets, first item of every bucket (all x 128), the item list + counter, the ones' list + counter,
    // the entries (32 n x 4 B), the items' and the ones' partial sums
    // (four words more than the carve: the alignment pad after the list, budgeted here rather than borrowed)
    char *const base = (char *)(uintptr_t)4096;
    const NmsmBufs b = bare ? nmsm_carve_bare<F>(base, 1, n) : nmsm_carve<F>(base, 1, n);
    return (((size_t)((char *)b.end - base) + 16) + 15) & ~(size_t)15;
}

template <class F> int msm_load(int device, int group, size_t num_points, const uint64_t *bases, int window_bits, bool bare, frw_msm **out, bool force_wide = false)
{
    // (a table row index and its window share 31 bits of an entry: 2^26 points x 16 windows, 2^25 x 32; a bare handle's entries are point indices)
    if (!out || !bases || num_points == 0 || num_points > (bare ? ((size_t)1 << 31) - 1 : (size_t)1 << (window_bits == 8 ? 25 : 26))) return FRW_E_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FRW_E_NO_DEVICE;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return frw::record_hip_error(e, "hipSetDevice");
    frw_msm *m = new (std::nothrow) frw_msm;
    if (!m) return FRW_E_OUT_OF_MEMORY;
    m->device = device;
    m->group = group;
    m->window_bits = window_bits;
    m->bare = bare;
    m->wide = bare && window_bits == 16 && (force_wide || num_points >= msm_wide_from());
    m->table = nullptr;
    m->ones_table = nullptr;
    m->dev.ones_table = nullptr;
    m->dev.n = (uint32_t)num_points;
    void *d_bases = nullptr;
    const size_t table_bytes = (size_t)(bare ? 1 : 256 / window_bits) * num_points * frw::Grp<F>::PT_WORDS * 4, ark_bytes = (size_t)frw::Grp<F>::ARK_WORDS * 4;
    e = hipMalloc(&m->table, table_bytes);
    if (e == hipSuccess) e = hipMalloc(&d_bases, num_points * ark_bytes);
    if (e == hipSuccess) e = hipMemcpy(d_bases, bases, num_points * ark_bytes, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(frw::msm_precompute_kernel<F>, dim3((unsigned)((num_points + 63) / 64)), dim3(64), 0, nullptr,
                           (uint32_t)num_points, (const uint32_t *)d_bases, (uint32_t *)m->table, bare ? 256 : window_bits);     // (256-bit "windows": one row per point)
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (d_bases) (void)hipFree(d_bases);
    d_bases = nullptr;
    // the subset sums of every group of eight points, for the scalars equal to one (narrow handles of per-signature size: an aggregate
    // statement's tables are tens of gigabytes as it is, and its ones go through the list)
    if (e == hipSuccess && !bare && window_bits == 8 && num_points <= ((size_t)1 << 18)) {
        const size_t entries = (num_points + 7) / 8 * 255;
        e = hipMalloc(&m->ones_table, entries * frw::Grp<F>::PT_WORDS * 4);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(frw::msm_ones_table_kernel<F>, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, nullptr, (uint32_t)num_points,
                               (const uint32_t *)m->table, (uint32_t *)m->ones_table);
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (e != hipSuccess) {
        frw_msm_free(m);
        return frw::record_hip_error(e, "frw_msm_load");
    }
    m->dev.table = (const uint32_t *)m->table;
    m->dev.ones_table = (const uint32_t *)m->ones_table;
    *out = m;
    return FRW_OK;
}

template <class F> int fixed_base(int device, size_t count, const uint64_t *scalars, uint64_t *out)
{
    if (count && (!scalars || !out)) return FRW_E_INVALID_ARG;
    if (count == 0) return FRW_OK;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FRW_E_NO_DEVICE;
    hipError_t e = hipSetDevice(device);
    void *d_table = nullptr, *d_sc = nullptr, *d_out = nullptr;
    const size_t ark_bytes = (size_t)frw::Grp<F>::ARK_WORDS * 4;
    if (e == hipSuccess) e = hipMalloc(&d_table, (size_t)frw::FB_WINDOWS * frw::FB_DIGITS * frw::Grp<F>::PT_WORDS * 4);
    if (e == hipSuccess) e = hipMalloc(&d_sc, count * 32);
    if (e == hipSuccess) e = hipMalloc(&d_out, count * ark_bytes);
    if (e == hipSuccess) e = hipMemcpy(d_sc, scalars, count * 32, hipMemcpyHostToDevice);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(frw::fixed_base_table_kernel<F>, dim3(frw::FB_WINDOWS * frw::FB_DIGITS / 64), dim3(64), 0, nullptr, (uint32_t *)d_table);
        hipLaunchKernelGGL(frw::fixed_base_kernel<F>, dim3((unsigned)((count + 63) / 64)), dim3(64), 0, nullptr, count,
                           (const uint32_t *)d_sc, (const uint32_t *)d_table, (uint32_t *)d_out);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpy(out, d_out, count * ark_bytes, hipMemcpyDeviceToHost);
    for (void *p : {d_table, d_sc, d_out})
        if (p) (void)hipFree(p);
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_fixed_base");
}

// One chunk of the dense pipeline: `rows` grid rows -- signatures with all their windows, or (a bare handle) the sixteen windows of ONE
// scalar vector -- to `rows` sums in `out` (XYZZ buckets or ark-ff's affine bytes).
template <class F, bool PREFETCH>
hipError_t msm_rows(const frw_msm *m, size_t rows, const uint32_t *sc, size_t stride_words, int montgomery, uint32_t *out, bool xyzz_out,
                    void *d_workspace, hipStream_t st)
{
    const uint32_t n = m->dev.n;
    const bool bare = m->bare;
    const MsmBufs<F> b = msm_carve<F>(d_workspace, rows, n, bare);
    uint32_t *slice_hist = b.buckets;                                // [rows][32][32,768], dead before the first bucket is stored (rows == 1: see below)
    const int slices = rows == 1 ? frw::MSM_SLICES_LONE : frw::MSM_SLICES;
    if (rows == 1) slice_hist = b.partial_items;                     // [224][32,768], dead before the first work item's sum is stored
    const dim3 sgrid((unsigned)slices, (unsigned)rows);
    hipError_t e = hipMemsetAsync(b.ones_count, 0, rows * 16, st);
    if (e != hipSuccess) return e;
    if (bare) {
        hipLaunchKernelGGL(frw::msm_digits_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, sc, montgomery, b.digits);
        hipLaunchKernelGGL(frw::msm_hist_digits_kernel, sgrid, dim3(1024), 0, st, n, (const int16_t *)b.digits, slice_hist);
    } else {
        hipLaunchKernelGGL(frw::msm_hist_kernel, sgrid, dim3(1024), 0, st, n, sc, stride_words, montgomery, slice_hist, b.ones_count, b.ones_list, 0);
    }
    hipLaunchKernelGGL(frw::msm_slice_offsets_kernel, dim3(frw::MSM_BUCKETS / 256, (unsigned)rows), dim3(256), 0, st, slice_hist, b.counts, slices);
    hipLaunchKernelGGL(frw::msm_scan_kernel, dim3((unsigned)rows), dim3(1024), 0, st, b.counts, b.offsets);
    // (four at a time: 12.6 ms with the finer split, 12.1 without); beyond 2^18 points: see MSM_MAX_ITEMS_LARGE
    // (a lone Falcon-1024 proof with 2 / 3 / 4: 4.72 / 4.27 / 4.05 ms -- the finer cut shares the chip better with the proof's other
    // sums; 5, with a longer item list, changes nothing any more: 3.86 - 3.91 against 3.90 ms, the sums overlap either way)
    // (a bare handle's sixteen windows are sixteen rows of equal buckets: 524,288 items as they are)
    const uint32_t finer = bare ? 1u : rows <= 2 ? (n > (1u << 18) ? 24u : 4u) : 1u;
    hipLaunchKernelGGL(frw::msm_order_kernel, dim3((unsigned)rows), dim3(1024), 0, st, b.counts, b.offsets, b.order, b.item_first, b.items, b.item_count, finer,
                       b.max_items);
    if (bare) hipLaunchKernelGGL(frw::msm_scatter_digits_kernel, sgrid, dim3(1024), 0, st, n, (const int16_t *)b.digits, b.offsets, slice_hist, b.entries);
    else hipLaunchKernelGGL(frw::msm_scatter_kernel, sgrid, dim3(1024), 0, st, n, sc, stride_words, montgomery, b.offsets, slice_hist, b.entries, 0);
    hipLaunchKernelGGL((frw::msm_bucket_kernel<F, PREFETCH>), dim3(b.max_items / 64, (unsigned)rows), dim3(64), 0, st, m->dev, b.offsets,
                       b.counts, b.items, b.item_count, b.entries, b.partial_items, finer, b.max_items, b.ent_stride, (const unsigned long long *)nullptr);
    hipLaunchKernelGGL(frw::msm_combine_kernel<F>, dim3(frw::MSM_BUCKETS / 64, (unsigned)rows), dim3(64), 0, st, b.offsets, b.counts, b.item_first,
                       b.partial_items, b.buckets, finer, b.max_items);
    // first stage of the fold: enough threads to occupy the chip (~2^16), as few as that allows
    // (64 per call with 16 / 32 / 64 buckets per thread: 927 - 929 / 927 - 930 / 919 proofs/s -- the fold is 3 of a call's 69 ms)
    const int log_chunk = rows >= 128 ? 6 : rows >= 32 ? 5 : 3;
    const unsigned t1 = (unsigned)frw::MSM_BUCKETS >> log_chunk;
    frw::MsmDev ones_dev = m->dev;
    if (bare) ones_dev.n = 0;                                        // (no list: every row's count is zero and its slice of the list the same empty one)
    hipLaunchKernelGGL(frw::msm_ones_kernel<F>, dim3(t1 / 64, (unsigned)rows), dim3(64), 0, st, ones_dev, b.ones_count, b.ones_list, b.partial);
    if (log_chunk == 6) hipLaunchKernelGGL((frw::msm_fold1_kernel<F, 6>), dim3(t1 / 64, (unsigned)rows), dim3(64), 0, st, b.buckets, b.partial, (uint32_t *)nullptr);
    else if (log_chunk == 5) hipLaunchKernelGGL((frw::msm_fold1_kernel<F, 5>), dim3(t1 / 64, (unsigned)rows), dim3(64), 0, st, b.buckets, b.partial, (uint32_t *)nullptr);
    else hipLaunchKernelGGL((frw::msm_fold1_kernel<F, 3>), dim3(t1 / 64, (unsigned)rows), dim3(64), 0, st, b.buckets, b.partial, (uint32_t *)nullptr);
    hipLaunchKernelGGL(frw::msm_fold2_kernel<F>, dim3((unsigned)rows), dim3(frw::MSM_FOLD_THREADS), 0, st, b.partial,
                       (int)(t1 / frw::MSM_FOLD_THREADS), out, xyzz_out ? 1 : 0);
    return hipGetLastError();
}
// ONE scalar vector over a bare handle on wide windows (frw::WIDE_*): the coarse sort, the rows' fine sorts, work items, buckets, the
// two-stage folds (weighted and plain), the windows, Horner's rule
template <class F, bool PREFETCH>
hipError_t msm_rows_wide(const frw_msm *m, const uint32_t *sc, int montgomery, uint32_t *out, bool xyzz_out, void *d_workspace, hipStream_t st)
{
    constexpr unsigned R = frw::WIDE_ROWS;
    const uint32_t n = m->dev.n;
    const MsmWideBufs<F> w = msm_carve_wide<F>(d_workspace, n);
    const MsmBufs<F> &b = w.rows;
    hipError_t e = hipMemsetAsync(b.ones_count, 0, R * 16, st);
    if (e != hipSuccess) return e;
    const dim3 sgrid(frw::WIDE_SLICES, frw::WIDE_W);
    hipLaunchKernelGGL(frw::msm_wide_digits_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, sc, montgomery, w.digits);
    hipLaunchKernelGGL(frw::msm_wide_bin_hist_kernel, sgrid, dim3(1024), 0, st, n, (const int32_t *)w.digits, w.slice_hist);
    hipLaunchKernelGGL(frw::msm_wide_bin_offsets_kernel, dim3(1), dim3(1024), 0, st, (uint32_t)frw::WIDE_SLICES, w.slice_hist, w.bin_count, w.bin_start, w.row_count,
                       w.row_start);
    hipLaunchKernelGGL(frw::msm_wide_bin_scatter_kernel, sgrid, dim3(1024), 0, st, n, (const int32_t *)w.digits, (const uint32_t *)w.slice_hist,
                       (const unsigned long long *)w.bin_start, w.coarse);
    uint32_t *part_hist = b.buckets;                                  // [6,656][32][1,024]: 0.9 GB of the buckets' 1.5, dead before the first bucket is stored
    static_assert((size_t)frw::WIDE_ALL_BINS * frw::WIDE_PARTS * frw::WIDE_BIN_BUCKETS <= (size_t)R * frw::MSM_BUCKETS * frw::Grp<F>::BK_WORDS, "the parts' histograms borrow the buckets");
    const dim3 pgrid(frw::WIDE_ALL_BINS * frw::WIDE_PARTS);
    hipLaunchKernelGGL(frw::msm_wide_bucket_hist_kernel, pgrid, dim3(1024), 0, st, (const uint2 *)w.coarse, (const uint32_t *)w.bin_count,
                       (const unsigned long long *)w.bin_start, part_hist);
    hipLaunchKernelGGL(frw::msm_wide_bucket_parts_kernel, dim3(frw::WIDE_ALL_BINS), dim3(1024), 0, st, part_hist, b.counts);
    hipLaunchKernelGGL(frw::msm_scan_kernel, dim3(R), dim3(1024), 0, st, b.counts, b.offsets);
    hipLaunchKernelGGL(frw::msm_order_kernel, dim3(R), dim3(1024), 0, st, b.counts, b.offsets, b.order, b.item_first, b.items, b.item_count, 1u, b.max_items);
    hipLaunchKernelGGL(frw::msm_wide_bucket_scatter_kernel, pgrid, dim3(1024), 0, st, (const uint2 *)w.coarse, (const uint32_t *)w.bin_count,
                       (const unsigned long long *)w.bin_start, (const uint32_t *)b.offsets, (const uint32_t *)part_hist, (const unsigned long long *)w.row_start,
                       w.entries);
    hipLaunchKernelGGL((frw::msm_bucket_kernel<F, PREFETCH>), dim3(b.max_items / 64, R), dim3(64), 0, st, m->dev, b.offsets, b.counts, b.items, b.item_count,
                       w.entries, b.partial_items, 1u, b.max_items, (size_t)0, (const unsigned long long *)w.row_start);
    hipLaunchKernelGGL(frw::msm_combine_kernel<F>, dim3(frw::MSM_BUCKETS / 64, R), dim3(64), 0, st, b.offsets, b.counts, b.item_first, b.partial_items,
                       b.buckets, 1u, b.max_items);
    // the folds: 64 buckets per thread (208 rows x 512 threads fill the chip), weighted and plain; then 512 threads per row and sum
    constexpr int LOG_CHUNK = 6;
    const unsigned t1 = (unsigned)frw::MSM_BUCKETS >> LOG_CHUNK;
    frw::MsmDev ones_dev = m->dev;
    ones_dev.n = 0;
    hipLaunchKernelGGL(frw::msm_ones_kernel<F>, dim3(t1 / 64, R), dim3(64), 0, st, ones_dev, b.ones_count, b.ones_list, b.partial);     // (no ones: the partial sums start as the identity)
    hipLaunchKernelGGL((frw::msm_fold1_kernel<F, LOG_CHUNK>), dim3(t1 / 64, R), dim3(64), 0, st, b.buckets, b.partial, w.partial_plain);
    hipLaunchKernelGGL(frw::msm_fold2_kernel<F>, dim3(R), dim3(frw::MSM_FOLD_THREADS), 0, st, b.partial, (int)(t1 / frw::MSM_FOLD_THREADS), w.s1, 1);
    hipLaunchKernelGGL(frw::msm_fold2_kernel<F>, dim3(R), dim3(frw::MSM_FOLD_THREADS), 0, st, w.partial_plain, (int)(t1 / frw::MSM_FOLD_THREADS), w.s0, 1);
    hipLaunchKernelGGL(frw::msm_wide_window_kernel<F>, dim3(frw::WIDE_W), dim3(64), 0, st, (const uint32_t *)w.s1, (const uint32_t *)w.s0, w.window_sums);
    hipLaunchKernelGGL(frw::msm_horner_kernel<F>, dim3(1), dim3(64), 0, st, (const uint32_t *)w.window_sums, (int)frw::WIDE_W, (int)frw::WIDE_C, out, xyzz_out ? 1 : 0);
    return hipGetLastError();
}
// the whole call for one group; `d_out` rows are ARK_WORDS / 2 uint64_t
template <class F, bool PREFETCH>
int msm_run(const frw_msm *m, size_t batch, const uint64_t *d_scalars, size_t scalar_stride, int montgomery, uint64_t *d_out,
            void *d_workspace, size_t workspace_bytes, hipStream_t st, bool xyzz_out = false)
{
    const uint32_t n = m->dev.n;
    const size_t per = msm_workspace_per_signature<F>(n, m->bare, m->wide);
    size_t chunk = workspace_bytes / per;
    if (chunk == 0 || ((uintptr_t)d_workspace & 15)) return FRW_E_INVALID_ARG;
    if (chunk > 32768) chunk = 32768;                                  // grid.y
    constexpr int BW = frw::Grp<F>::BK_WORDS;
    hipError_t e = hipSetDevice(m->device);
    if (m->wide) {
        for (size_t sig = 0; e == hipSuccess && sig < batch; sig++)
            e = msm_rows_wide<F, PREFETCH>(m, (const uint32_t *)(d_scalars + sig * scalar_stride * 4), montgomery,
                                           (uint32_t *)d_out + sig * (xyzz_out ? BW : frw::Grp<F>::ARK_WORDS), xyzz_out, d_workspace, st);
        return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_msm_dev");
    }
    if (m->bare) {
        // one scalar vector at a time: its sixteen windows fill the grid; the window sums wait at the end of the workspace for Horner's rule
        uint32_t *window_sums = (uint32_t *)((char *)d_workspace + per) - (size_t)frw::MSM_W * BW;
        for (size_t sig = 0; e == hipSuccess && sig < batch; sig++) {
            e = msm_rows<F, PREFETCH>(m, frw::MSM_W, (const uint32_t *)(d_scalars + sig * scalar_stride * 4), 0, montgomery, window_sums, true, d_workspace, st);
            if (e != hipSuccess) break;
            hipLaunchKernelGGL(frw::msm_horner_kernel<F>, dim3(1), dim3(64), 0, st, window_sums, frw::MSM_W, frw::MSM_C,
                               (uint32_t *)d_out + sig * (xyzz_out ? BW : frw::Grp<F>::ARK_WORDS), xyzz_out ? 1 : 0);
            e = hipGetLastError();
        }
        return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_msm_dev");
    }
    for (size_t lo = 0; e == hipSuccess && lo < batch; lo += chunk) {
        const size_t cnt = batch - lo < chunk ? batch - lo : chunk;
        e = msm_rows<F, PREFETCH>(m, cnt, (const uint32_t *)(d_scalars + lo * scalar_stride * 4), scalar_stride * 8, montgomery,
                                  (uint32_t *)d_out + lo * (xyzz_out ? BW : frw::Grp<F>::ARK_WORDS), xyzz_out, d_workspace, st);
    }
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_msm_dev");
}
// ---- the narrow pipeline in two halves: the counting sort of the scalars' digits (which knows nothing of the points), and the sums
// over one window table with a sort's result.  The four witness-side sums of a proof take the SAME scalars -- z ++ [1, r, s] against
// a_query, b_g1_query, b_g2_query and l_query, each table padded to the same length with points at infinity -- so a proof sorts once
// (frw_groth16_prove_dev); frw_msm_g1_dev / _g2_dev on a narrow handle are one sort and one sum.
// ones_as_mask: every table that will be summed with this sort has the subset sums of its groups of eight (MsmDev::ones_table): the
// scalars equal to one are then recorded as a bit mask (the first n / 32 words of each signature's list) instead of a list
hipError_t nmsm_sort(const NmsmBufs &b, uint32_t n, size_t cnt, const uint32_t *sc, size_t stride_words, int montgomery, bool ones_as_mask, hipStream_t st)
{
    const uint32_t slices = frw::nmsm_slices(n);
    const dim3 sgrid(slices, (unsigned)cnt);
    hipError_t e = hipMemsetAsync(b.ones_count, 0, cnt * 4, st);
    if (e == hipSuccess && ones_as_mask) e = hipMemsetAsync(b.ones_list, 0, cnt * (size_t)n * 4, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frw::nmsm_hist_kernel, sgrid, dim3(1024), 0, st, n, sc, stride_words, montgomery, b.slice_hist, b.ones_count, b.ones_list,
                       ones_as_mask ? 1 : 0);
    hipLaunchKernelGGL(frw::nmsm_plan_kernel, dim3((unsigned)cnt), dim3(frw::NMSM_BUCKETS), 0, st, b.slice_hist, b.counts, b.offsets, b.item_first, b.items,
                       b.item_count, b.target, b.max_items, slices);
    hipLaunchKernelGGL(frw::nmsm_scatter_kernel, sgrid, dim3(1024), 0, st, n, sc, stride_words, montgomery, b.offsets, b.slice_hist, b.entries);
    return hipGetLastError();
}
// ... of ONE scalar vector for bare handles: all thirty-two windows in one pass (`b` from nmsm_carve_bare)
// `index`: the sum is over the scalars (and rows) index[0 .. n) of `sc` (null: over 0 .. n)
hipError_t nmsm_sort_bare(const NmsmBufs &b, uint32_t n, const uint32_t *sc, int montgomery, hipStream_t st, const uint32_t *index = nullptr)
{
    const uint32_t slices = frw::nmsm_slices(n);
    hipError_t e = hipMemsetAsync(b.ones_count, 0, 16, st);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(frw::nmsm_hist_bare_kernel, dim3(slices), dim3(1024), 0, st, n, sc, montgomery, b.slice_hist, b.ones_count, b.ones_list, index);
    hipLaunchKernelGGL(frw::nmsm_plan_kernel, dim3(frw::NMSM_W), dim3(frw::NMSM_BUCKETS), 0, st, b.slice_hist, b.counts, b.offsets, b.item_first, b.items,
                       b.item_count, b.target, b.max_items, slices);
    hipLaunchKernelGGL(frw::nmsm_entry_base_kernel, dim3(1), dim3(64), 0, st, b.counts, b.offsets, b.entry_base);
    hipLaunchKernelGGL(frw::nmsm_scatter_bare_kernel, dim3(slices), dim3(1024), 0, st, n, sc, montgomery, b.offsets, b.slice_hist, b.entry_base, b.entries,
                       index);
    return hipGetLastError();
}
// `sorted`: whose sort arrays to read; `own`: where the tables' partial sums go -- carved for `tables * cnt` signatures when there is
// more than one table (the same carve as `sorted` for a sum on its own); d_out: [tables][cnt] results.  Every kernel here is a grid of
// single wavefronts: none of them can be kept waiting by a kernel that fills the chip.
// ones_stream / ones_done: the sums of the scalars that are one need nothing of the work items' sums and the other way round -- as two
// launches on one stream the second waited for the first.  A caller with a second stream that is ordered after the sort (and an event
// to spare) gives it here: the ones are summed there, `st` waits for them before the fold (a quarter of a millisecond off the chain of
// a proof made alone).
template <class F, bool PREFETCH>
hipError_t nmsm_accumulate(const frw_msm *const *ms, int tables, const NmsmBufs &sorted, const NmsmBufs &own, size_t cnt, uint32_t *d_out, bool ones_as_mask,
                           hipStream_t st, bool xyzz_out = false, bool ones_elsewhere = false, hipStream_t ones_stream = nullptr, hipEvent_t ones_done = nullptr)
{
    if (!ones_elsewhere) ones_stream = st;                          // (a caller's stream may be the null stream: the flag says whether it was given)
    const uint32_t n = ms[0]->dev.n;
    frw::NmsmTables dev;
    dev.n = n; dev.sigs = (uint32_t)cnt;
    for (int t = 0; t < 3; t++) {
        const frw_msm *m = ms[t < tables ? t : 0];
        dev.table[t] = m->dev.table;
        dev.ones_table[t] = ones_as_mask ? m->dev.ones_table : nullptr;   // not as a mask: the sort left a list of the ones
    }
    const unsigned rows = (unsigned)(cnt * (size_t)tables);
    typedef typename frw::BulkPolicy<F>::type FB;
    // two lanes per point (G2): WITHOUT the prefetch of the next row the bucket kernel is 249 registers = two waves per SIMD; with it 256 + 26
    // AGPRs = one (round 4 ran it that way).  FRW_G2_PREFETCH in the environment brings the prefetch back (the A/B of profiles/r05_g2_*).
    static const bool g2_prefetch = std::getenv("FRW_G2_PREFETCH") != nullptr;
    // the ones' partial sums: enough threads to occupy the chip (~2^16 over the batch), between 256 and 4,096 per signature
    int ones_threads = 256;
    while (ones_threads < (int)frw::nmsm_ones_max(n) && (size_t)ones_threads * rows < 65536) ones_threads <<= 1;
    hipLaunchKernelGGL(frw::nmsm_ones_kernel<FB>, dim3((unsigned)ones_threads / 64, rows), dim3(64), 0, ones_stream, dev, sorted.ones_count, sorted.ones_list,
                       own.partial_ones, own.ones_stride, 1u);
    const uint32_t *ones_for_finish = own.partial_ones;
    uint32_t ones_groups = (uint32_t)ones_threads / 64, ones_finish_stride = own.ones_stride;
    if (ones_groups > 64) {
        hipLaunchKernelGGL(frw::nmsm_ones_fold_kernel<FB>, dim3(64, rows), dim3(64), 0, ones_stream, own.partial_ones, ones_groups, own.ones_stride, own.folded_ones);
        ones_for_finish = own.folded_ones;
        ones_groups = 64;
        ones_finish_stride = 64;
    }
    if (ones_elsewhere) {
        const hipError_t e = hipEventRecord(ones_done, ones_stream);
        if (e != hipSuccess) return e;
    }
    if (PREFETCH || (FB::LANES > 1 && g2_prefetch))
        hipLaunchKernelGGL((frw::nmsm_bucket_kernel<FB, true>), dim3((sorted.max_items * FB::LANES + 63) / 64, rows), dim3(64), 0, st, dev,
                           sorted.offsets, sorted.counts, sorted.items, sorted.item_count, sorted.entries, own.partial_items, sorted.target, sorted.max_items,
                           (const unsigned long long *)nullptr);
    else
        hipLaunchKernelGGL((frw::nmsm_bucket_kernel<FB, false>), dim3((sorted.max_items * FB::LANES + 63) / 64, rows), dim3(64), 0, st, dev,
                           sorted.offsets, sorted.counts, sorted.items, sorted.item_count, sorted.entries, own.partial_items, sorted.target, sorted.max_items,
                           (const unsigned long long *)nullptr);
    // the items of a bucket: added up by a workgroup per bucket where latency counts (a lone proof: 12 ms -> 8.6), by the fold's own
    // thread where throughput does (the combine is nine times the wave-level additions: 3 % of a 64-proof call)
    const bool combine = cnt <= 16;
    if (combine)
        hipLaunchKernelGGL(frw::nmsm_combine_kernel<FB>, dim3(frw::NMSM_BUCKETS, rows), dim3(64), 0, st, sorted.counts, sorted.offsets, sorted.item_first,
                           own.partial_items, own.bucket_sums, sorted.target, sorted.max_items, (uint32_t)cnt);
    if (ones_elsewhere) {
        const hipError_t e = hipStreamWaitEvent(st, ones_done, 0);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(frw::nmsm_finish_kernel<FB>, dim3(rows), dim3(64), 0, st, sorted.counts, sorted.offsets, sorted.item_first, own.partial_items,
                       combine ? own.bucket_sums : (uint32_t *)nullptr, ones_for_finish, (int)ones_groups, ones_finish_stride, sorted.target, sorted.max_items,
                       d_out, xyzz_out ? 1 : 0, (uint32_t)cnt, 0);
    return hipGetLastError();
}
// The sums of ONE scalar vector over `tables` bare handles that share its sort (`b` from nmsm_carve_bare<F>(.., tables, n): the sort's
// arrays and the tables' own): grid row = (table, window); a window's buckets are cut into work items, combined by a workgroup per bucket
// (the low windows hold every small value of a witness: a bucket of 250,000 entries is a thousand items), folded by one wavefront per
// (table, window); the scalars equal to one -- a list -- are summed by up to 65,536 threads per table and join window 0; Horner's rule
// over the thirty-two window sums.  d_out: [tables] results.
template <class F, bool PREFETCH>
hipError_t nmsm_accumulate_bare(const frw_msm *const *ms, int tables, const NmsmBufs &b, uint32_t *d_out, hipStream_t st, bool xyzz_out, uint32_t out_step = 1)
{
    const uint32_t n = b.n;                                          // (the sort's: the tables' rows, or those an index names)
    frw::NmsmTables dev;
    dev.n = n; dev.sigs = frw::NMSM_W;
    for (int t = 0; t < 3; t++) {
        dev.table[t] = ms[t < tables ? t : 0]->dev.table;
        dev.ones_table[t] = nullptr;
    }
    const unsigned rows = (unsigned)tables * frw::NMSM_W;
    typedef typename frw::BulkPolicy<F>::type FB;
    static const bool g2_prefetch = std::getenv("FRW_G2_PREFETCH") != nullptr;       // (see nmsm_accumulate)
    const uint32_t ones_threads = frw::nmsm_ones_max(n);             // 4,096, or 65,536 beyond 2^18 points
    hipLaunchKernelGGL(frw::nmsm_ones_kernel<FB>, dim3(ones_threads / 64, (unsigned)tables), dim3(64), 0, st, dev, b.ones_count, b.ones_list, b.partial_ones,
                       b.ones_stride, (uint32_t)frw::NMSM_W);
    const uint32_t *ones_for_finish = b.partial_ones;
    uint32_t ones_groups = ones_threads / 64, ones_finish_stride = b.ones_stride;
    if (ones_groups > 64) {
        hipLaunchKernelGGL(frw::nmsm_ones_fold_kernel<FB>, dim3(64, (unsigned)tables), dim3(64), 0, st, b.partial_ones, ones_groups, b.ones_stride, b.folded_ones);
        ones_for_finish = b.folded_ones;
        ones_groups = 64;
        ones_finish_stride = 64;
    }
    if (PREFETCH || (FB::LANES > 1 && g2_prefetch))
        hipLaunchKernelGGL((frw::nmsm_bucket_kernel<FB, true>), dim3((b.max_items * FB::LANES + 63) / 64, rows), dim3(64), 0, st, dev,
                           b.offsets, b.counts, b.items, b.item_count, b.entries, b.partial_items, b.target, b.max_items, (const unsigned long long *)b.entry_base);
    else
        hipLaunchKernelGGL((frw::nmsm_bucket_kernel<FB, false>), dim3((b.max_items * FB::LANES + 63) / 64, rows), dim3(64), 0, st, dev,
                           b.offsets, b.counts, b.items, b.item_count, b.entries, b.partial_items, b.target, b.max_items, (const unsigned long long *)b.entry_base);
    hipLaunchKernelGGL(frw::nmsm_combine_kernel<FB>, dim3(frw::NMSM_BUCKETS, rows), dim3(64), 0, st, b.counts, b.offsets, b.item_first,
                       b.partial_items, b.bucket_sums, b.target, b.max_items, (uint32_t)frw::NMSM_W);
    hipLaunchKernelGGL(frw::nmsm_finish_kernel<FB>, dim3(rows), dim3(64), 0, st, b.counts, b.offsets, b.item_first, b.partial_items,
                       b.bucket_sums, ones_for_finish, (int)ones_groups, ones_finish_stride, b.target, b.max_items,
                       b.window_sums, 1, (uint32_t)frw::NMSM_W, 1);
    hipLaunchKernelGGL(frw::msm_horner_kernel<FB>, dim3((unsigned)tables), dim3(64), 0, st, b.window_sums, frw::NMSM_W, frw::NMSM_C, d_out, xyzz_out ? 1 : 0,
                       out_step);
    return hipGetLastError();
}
// the narrow pipeline for one group
template <class F, bool PREFETCH>
int nmsm_run(const frw_msm *m, size_t batch, const uint64_t *d_scalars, size_t scalar_stride, int montgomery, uint64_t *d_out,
             void *d_workspace, size_t workspace_bytes, hipStream_t st)
{
    const uint32_t n = m->dev.n;
    const size_t per = nmsm_workspace_per_signature<F>(n, m->bare);
    size_t chunk = workspace_bytes / per;
    if (chunk == 0 || ((uintptr_t)d_workspace & 15)) return FRW_E_INVALID_ARG;
    if (chunk > 32768) chunk = 32768;                                  // grid.y
    hipError_t e = hipSetDevice(m->device);
    if (m->bare) {
        const NmsmBufs b = nmsm_carve_bare<F>(d_workspace, 1, n);
        for (size_t sig = 0; e == hipSuccess && sig < batch; sig++) {
            e = nmsm_sort_bare(b, n, (const uint32_t *)(d_scalars + sig * scalar_stride * 4), montgomery, st);
            if (e == hipSuccess) e = nmsm_accumulate_bare<F, PREFETCH>(&m, 1, b, (uint32_t *)(d_out + sig * (frw::Grp<F>::ARK_WORDS / 2)), st, false);
        }
        return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_msm_dev");
    }
    for (size_t lo = 0; e == hipSuccess && lo < batch; lo += chunk) {
        const size_t cnt = batch - lo < chunk ? batch - lo : chunk;
        const NmsmBufs b = nmsm_carve<F>(d_workspace, cnt, n);
        e = nmsm_sort(b, n, cnt, (const uint32_t *)(d_scalars + lo * scalar_stride * 4), scalar_stride * 8, montgomery, m->ones_table != nullptr, st);
        if (e == hipSuccess)
            e = nmsm_accumulate<F, PREFETCH>(&m, 1, b, b, cnt, (uint32_t *)(d_out + lo * (frw::Grp<F>::ARK_WORDS / 2)), m->ones_table != nullptr, st);
    }
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_msm_dev");
}
}  // namespace

extern "C" void frw_msm_free(frw_msm *m)
{
    if (!m) return;
    (void)hipSetDevice(m->device);
    if (m->table) (void)hipFree(m->table);
    if (m->ones_table) (void)hipFree(m->ones_table);
    delete m;
}

extern "C" int frw_msm_g1_load(int device, size_t num_points, const uint64_t *bases, frw_msm **out)
{
    return msm_load<FqField>(device, 1, num_points, bases, 16, false, out);
}
extern "C" int frw_msm_g1_load_narrow(int device, size_t num_points, const uint64_t *bases, frw_msm **out)
{
    return msm_load<FqField>(device, 1, num_points, bases, 8, false, out);
}
extern "C" int frw_msm_g2_load_narrow(int device, size_t num_points, const uint64_t *bases, frw_msm **out)
{
    return msm_load<Fq2Field>(device, 2, num_points, bases, 8, false, out);
}
extern "C" int frw_msm_g2_load(int device, size_t num_points, const uint64_t *bases, frw_msm **out)
{
    return msm_load<Fq2Field>(device, 2, num_points, bases, 16, false, out);
}
extern "C" int frw_msm_g1_load_bare(int device, size_t num_points, const uint64_t *bases, int narrow, frw_msm **out)
{
    if (narrow < 0 || narrow > 2) return FRW_E_INVALID_ARG;
    return msm_load<FqField>(device, 1, num_points, bases, narrow == 1 ? 8 : 16, true, out, narrow == 2);
}
extern "C" int frw_msm_g2_load_bare(int device, size_t num_points, const uint64_t *bases, int narrow, frw_msm **out)
{
    if (narrow < 0 || narrow > 2) return FRW_E_INVALID_ARG;
    return msm_load<Fq2Field>(device, 2, num_points, bases, narrow == 1 ? 8 : 16, true, out, narrow == 2);
}
extern "C" int frw_g1_fixed_base(int device, size_t count, const uint64_t *scalars, uint64_t *out)
{
    return fixed_base<FqField>(device, count, scalars, out);
}
extern "C" int frw_g2_fixed_base(int device, size_t count, const uint64_t *scalars, uint64_t *out)
{
    return fixed_base<Fq2Field>(device, count, scalars, out);
}

extern "C" int frw_msm_info(const frw_msm *m, frw_msm_info_t *out)
{
    if (!m || !out) return FRW_E_INVALID_ARG;
    const bool g2 = m->group == 2;
    out->num_points = m->dev.n;
    const bool narrow = m->window_bits == 8;
    out->window_bits = m->wide ? frw::WIDE_C : m->window_bits;
    out->num_windows = m->wide ? frw::WIDE_W : 256 / m->window_bits;
    out->table_bytes = (uint64_t)(m->bare ? 1 : out->num_windows) * m->dev.n * (g2 ? frw::Grp<Fq2Field>::PT_WORDS : frw::Grp<FqField>::PT_WORDS) * 4;
    if (m->ones_table) out->table_bytes += (uint64_t)((m->dev.n + 7) / 8) * 255 * (g2 ? frw::Grp<Fq2Field>::PT_WORDS : frw::Grp<FqField>::PT_WORDS) * 4;
    const uint32_t n = m->dev.n;
    const bool bare = m->bare;
    out->workspace_bytes_per_signature = narrow ? (g2 ? nmsm_workspace_per_signature<Fq2Field>(n, bare) : nmsm_workspace_per_signature<FqField>(n, bare))
                                                : (g2 ? msm_workspace_per_signature<Fq2Field>(n, bare, m->wide) : msm_workspace_per_signature<FqField>(n, bare, m->wide));
    return FRW_OK;
}

extern "C" int frw_msm_g1_dev(const frw_msm *m, size_t batch, const uint64_t *d_scalars, size_t scalar_stride, int montgomery,
                              uint64_t *d_out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!m || m->group != 1 || (batch && (!d_scalars || !d_out || !d_workspace)) || scalar_stride < m->dev.n) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (m->window_bits == 8) return nmsm_run<FqField, true>(m, batch, d_scalars, scalar_stride, montgomery, d_out, d_workspace, workspace_bytes, (hipStream_t)stream);
    return msm_run<FqField, true>(m, batch, d_scalars, scalar_stride, montgomery, d_out, d_workspace, workspace_bytes, (hipStream_t)stream);
}
extern "C" int frw_msm_g2_dev(const frw_msm *m, size_t batch, const uint64_t *d_scalars, size_t scalar_stride, int montgomery,
                              uint64_t *d_out, void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!m || m->group != 2 || (batch && (!d_scalars || !d_out || !d_workspace)) || scalar_stride < m->dev.n) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (m->window_bits == 8) return nmsm_run<Fq2Field, false>(m, batch, d_scalars, scalar_stride, montgomery, d_out, d_workspace, workspace_bytes, (hipStream_t)stream);
    return msm_run<Fq2Field, false>(m, batch, d_scalars, scalar_stride, montgomery, d_out, d_workspace, workspace_bytes, (hipStream_t)stream);
}

extern "C" int frw_groth16_msm_h_dev(const frw_msm *m, size_t batch, const uint64_t *d_h, size_t domain_size, uint64_t *d_out,
                                     void *d_workspace, size_t workspace_bytes, void *stream)
{
    // prover.rs: the scalars are h's coefficients 0 .. n - 2 (h_query has n - 1 points; the zip drops the last coefficient)
    if (!m || (size_t)m->dev.n + 1 != domain_size) return FRW_E_INVALID_ARG;
    return frw_msm_g1_dev(m, batch, d_h, domain_size, 1, d_out, d_workspace, workspace_bytes, stream);
}

// ---- a whole Groth16 proof per signature, on the device ------------------------------------------------------------------------------
// ark-groth16 0.3.0 prover.rs, create_proof_with_reduction_and_matrices (what examples/pok_sig.rs:30-47 of the reference runs),
// for every signature of a resident batch of witnesses:
//     h    = witness_map(...)                                                   frw_qap_witness_map_dev
//     g_a  = r delta_g1 + a_query[0] + MSM(a_query[1..], assignment) + alpha_g1      one sum over  a_query ++ [alpha, delta1]   with z ++ [1, r]
//     g1_b = s delta_g1 + b_g1_query[0] + MSM(b_g1_query[1..], ...) + beta_g1        the sum over  b_g1_query ++ [beta1]        with z ++ [1]   (g1_b - s delta1)
//     g2_b = s delta_g2 + b_g2_query[0] + MSM(b_g2_query[1..], ...) + beta_g2        one sum over  b_g2_query ++ [beta2, O, delta2]  with z ++ [1, r, s]
//     g_c  = s g_a + r g1_b - r s delta_g1 + MSM(l_query, aux) + MSM(h_query, h)  =  L + H + s g_a + r (g1_b - s delta1)
// z = instance_assignment ++ witness_assignment (z[0] = 1 is the scalar of query[0]); the blinding terms ride in the sums as
// extra points, the two scalar multiplications of g_c are done by four lanes each at the end (frw_quad.h).
namespace frw {
// 2^517 mod p in nine 29-bit limbs: f29_mul(x, 2^517) = x 2^256, the Montgomery form of a canonical scalar
__device__ const uint32_t FR_TO_MONT[NL29] = {0x1e538d9eu, 0x19e99103u, 0x13b31eccu, 0x04e2d5e4u, 0x181dac62u, 0x115f1ba1u,
                                              0x1e414fbbu, 0x11b3009cu, 0x00013fecu};
// the tail of every signature's scalar vector: [1, r, s] in Montgomery form after the nv assignment values
__global__ __launch_bounds__(64) void groth16_tails_kernel(size_t batch, const uint32_t *__restrict__ rs /* [batch][2][8] canonical */,
                                                           uint32_t *__restrict__ zext, size_t stride_words, size_t nv)
{
    const size_t sig = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (sig >= batch) return;
    uint32_t *dst = zext + sig * stride_words + nv * 8;
    constexpr uint32_t one[8] = FRW_R32;
    Fr8 o;
#pragma unroll
    for (int k = 0; k < 8; k++) o.l[k] = one[k];
    fr_store(dst, o);
    F29 c;
#pragma unroll
    for (int k = 0; k < NL29; k++) c.l[k] = FR_TO_MONT[k];
    for (int j = 0; j < 2; j++)
        fr_store(dst + 8 * (j + 1), f29_pack(f29_canonical(f29_mul(f29_unpack(fr_load(rs + (sig * 2 + j) * 8)), c))));
}
// k (any 256-bit value; taken mod the group order r) = k0 + lambda k1, lambda = z^2 - 1 = 0xac45a4010001a40200000000ffffffff,
// r = lambda^2 + lambda + 1: k0 = k mod lambda, k1 = k div lambda <= lambda + 1 < 2^128 -- for both blinding factors of every proof,
// one thread each (plain schoolbook division, a bit at a time: two scalars per proof).  out: [batch][2][k0 (2 x u64) | k1 (2 x u64)].
// On the device so that a call with its blinding factors in device memory never waits on the host (frw_groth16_prove_rs_dev).
__global__ __launch_bounds__(64) void groth16_split_kernel(size_t count, const uint64_t *__restrict__ rs /* [count][4] */, uint64_t *__restrict__ out /* [count][4] */)
{
    const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const uint64_t R[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    const uint64_t L0 = 0x00000000ffffffffULL, L1 = 0xac45a4010001a402ULL;
    uint64_t k[4] = {rs[i * 4], rs[i * 4 + 1], rs[i * 4 + 2], rs[i * 4 + 3]};
    for (int round = 0; round < 2; round++) {                       // 2^256 < 3 r: at most two subtractions
        bool ge = true;
        for (int j = 3; j >= 0; j--)
            if (k[j] != R[j]) { ge = k[j] > R[j]; break; }
        if (!ge) break;
        uint64_t borrow = 0;
        for (int j = 0; j < 4; j++) {
            const uint64_t d = k[j] - R[j], d2 = d - borrow;
            borrow = (k[j] < R[j]) | (d < borrow);
            k[j] = d2;
        }
    }
    uint64_t r0 = 0, r1 = 0, q0 = 0, q1 = 0;
    for (int bit = 255; bit >= 0; bit--) {
        const uint64_t top = r1 >> 63;
        r1 = (r1 << 1) | (r0 >> 63);
        r0 = (r0 << 1) | ((k[bit >> 6] >> (bit & 63)) & 1ULL);
        q1 = (q1 << 1) | (q0 >> 63);                                 // k < r: the quotient's bits above 127 are zero
        q0 <<= 1;
        if (top || r1 > L1 || (r1 == L1 && r0 >= L0)) {
            const uint64_t b = r0 < L0;
            r0 -= L0;
            r1 = r1 - L1 - b;
            q0 |= 1ULL;
        }
    }
    out[i * 4] = r0; out[i * 4 + 1] = r1; out[i * 4 + 2] = q0; out[i * 4 + 3] = q1;
}
// k P for one point per signature (s g_a and r (g1_b - s delta1) of g_c); runs on the side stream that made P.  A point that exists
// only now: the chain of doublings is the latency of a proof made alone, so the scalar comes split by the endomorphism
// (k = k0 + lambda k1: groth16_split_kernel) and the two 128-bit halves share their doublings: 128
// doublings and ~96 additions of P, phi(P) or P + phi(P) instead of 256 and ~128.  P arrives in XYZZ coordinates as its sum left it
// and k P leaves the same way (the proof needs one inversion, at the very end: groth16_finish_kernel).  phi(P) and P + phi(P) come
// for free in the same denominators -- phi(x, y) = (beta x, y), and since the two have the same y their chord is horizontal:
// P + phi(P) = (-(1 + beta) x, -y) -- so a step is ONE doubling and ONE addition whose operand is selected.  (Until late in round 4
// one thread per scalar multiplication ran this loop with the one-lane formulas of frw_fq29.h: 3.1 ms of a lone proof's 5.6.)
// On FOUR lanes (frw_quad.h): the formulas' independent products side by side -- three dependent levels per
// doubling instead of nine products, four per addition instead of fourteen -- with the results in a register file in LDS.  One
// quad per workgroup: its lanes take the same branches (they depend on the scalar's bits and on zero tests every lane sees
// alike), so nothing diverges, and a batch of scalar multiplications spreads over as many SIMDs.
struct QuadLaneExec {
    uint32_t *lds;
    uint32_t lane;
    // LDS serves a wavefront's requests in order; what must not happen is the COMPILER moving a level's reads above the previous
    // level's writes (per lane they touch different slots)
    static __device__ __forceinline__ void fence()
    {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
    static __device__ __forceinline__ uint32_t from_lane0(uint32_t x)
    {
        return (uint32_t)__builtin_amdgcn_mov_dpp((int)x, 0x00 /* quad_perm [0, 0, 0, 0] */, 0xf, 0xf, true);
    }
    template <const quad::Step &S> __device__ __forceinline__ void step(uint32_t qx, uint32_t qy)
    {
        Fq29 a;
        const Fq29 r = quad::compute<S>(lds, lane, qx, qy, a);
        quad::store<S>(lds, lane, r);
        fence();
    }
    template <const quad::Step &S> __device__ __forceinline__ void step_test(uint32_t qx, uint32_t qy, bool &pz)
    {
        Fq29 a;
        const Fq29 r = quad::compute<S>(lds, lane, qx, qy, a);
        quad::store<S>(lds, lane, r);
        fence();
        pz = from_lane0(fq_is_zero(a) ? 1u : 0u) != 0;
    }
    __device__ __forceinline__ bool degenerate(uint32_t qx, uint32_t qy)
    {
        uint32_t inf = 0;
        if (lane == 0) inf = quad::add_degenerate(lds, qx, qy) ? 1u : 0u;
        fence();
        return from_lane0(inf) != 0;
    }
};
// Both scalar multiplications of a batch in one launch: workgroup b < batch is s g_a of signature b (the second scalar of its pair),
// workgroup batch + b is r (g1_b - s delta1) (the first); pts = [2][batch] points as the sums left them, out likewise.
__global__ __launch_bounds__(4) void groth16_scale_quad_kernel(size_t batch, const uint32_t *__restrict__ split /* [batch][2][8]: k0 | k1, 128 bits each */,
                                                               const uint32_t *__restrict__ pts /* [2][batch][BK_WORDS] */,
                                                               uint32_t *__restrict__ out /* [2][batch][BK_WORDS] */)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    typedef FqField F;
    constexpr int BW = Grp<F>::BK_WORDS;
    __shared__ __attribute__((aligned(16))) uint32_t file[quad::NSLOTS * quad::SLOT_WORDS];
    const size_t slot = blockIdx.x;
    const uint32_t lane = threadIdx.x;
    if (slot >= 2 * batch) return;
    const size_t sig = slot < batch ? slot : slot - batch;
    const int which = slot < batch ? 1 : 0;
    if (pts[slot * BW + 4 * F::WORDS] != 0) {                                              // k O = O
        if (lane == 0) store_bucket<F>(out + slot * BW, pt_identity<F>());
        return;
    }
    if (lane == 0) quad::setup(file, load_bucket<F>(pts + slot * BW));
    QuadLaneExec::fence();
    const uint32_t *k = split + (sig * 2 + which) * 8;
    uint32_t k0[4], k1[4];
#pragma unroll
    for (int i = 0; i < 4; i++) { k0[i] = k[i]; k1[i] = k[4 + i]; }
    QuadLaneExec ex{file, lane};
    const bool inf = quad::scalar_mul(ex, k0, k1);
    if (lane == 0) store_bucket<F>(out + slot * BW, inf ? pt_identity<F>() : quad::running_point(file, false));
}
// (A/B, negative, round 4: a workgroup of three wavefronts per scalar multiplication -- one running the 128 doublings from the LOW end
// of the scalar and parking 2^i P in LDS, two adding the parked points where k0 / k1 have bits set, a tree, phi on the k1 sum -- is
// bit-exact and SLOWER for a proof made alone, in two builds (the second with nothing but the doubled point live in the doubler's loop
// and the doubling inlined): 5.5 and 4.9 ms against the 3.3 ms of the chain above; profiles/r04_groth16_scale_wave_ab.txt.  Removed.)
// C = L + H + s A + r B1', then the proof row A (12 u64) | B (24) | C (12).  TWO lanes per proof: the even one adds up C, the odd one
// holds A, and both turn their XYZZ point into ark-ff's affine bytes at once -- the one inversion on a proof's critical path
// (the sums and the scalar multiplications hand their results over in XYZZ coordinates).
__global__ __launch_bounds__(64) void groth16_finish_kernel(size_t batch, const uint32_t *__restrict__ a_pts, const uint32_t *__restrict__ sa_pts,
                                                            const uint32_t *__restrict__ rb1_pts, const uint32_t *__restrict__ l_pts,
                                                            const uint32_t *__restrict__ h_pts /* all five [batch][BK_WORDS] */,
                                                            const uint32_t *__restrict__ b2_pts /* [batch][48], affine */,
                                                            uint32_t *__restrict__ proofs /* [batch][96] */)
{
    __builtin_amdgcn_s_setprio(2);                               // latency, not throughput: ahead of a chip-filling bucket kernel's waves on the same SIMD
    typedef FqField F;
    constexpr int BW = Grp<F>::BK_WORDS;
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x, sig = t >> 1;
    const bool holds_a = (t & 1) != 0;
    if (sig >= batch) return;
    XyzzT<F> acc = load_bucket<F>((holds_a ? a_pts : l_pts) + sig * BW);
#pragma nounroll
    for (int j = 0; j < 3; j++) {
        const uint32_t *p = j == 0 ? h_pts : j == 1 ? sa_pts : rb1_pts;
        if (!holds_a) acc = pt_add(acc, load_bucket<F>(p + sig * BW));
    }
    uint32_t *o = proofs + sig * 96;
    if (!holds_a)
        for (int k = 0; k < 48; k++) o[24 + k] = b2_pts[sig * 48 + k];
    store_ark_point<F>(o + (holds_a ? 0 : 72), pt_to_affine(acc));
}
// ---- a key in slices (round 5): every rank sums its slice of every query; the five partial sums of a rank leave as ark-ff's affine
// bytes -- A | B1' | L | H (G1, 12 u64 each) | B (G2, 24) -- and any rank puts `world` of them together
__global__ __launch_bounds__(64) void groth16_partial_kernel(const uint32_t *__restrict__ g1_pts /* [4][BK_WORDS]: A, B1', L, H */,
                                                             const uint32_t *__restrict__ b2_ark /* [48] */, uint32_t *__restrict__ out /* [144] */)
{
    const uint32_t t = threadIdx.x;
    if (t < 4) store_ark_point<FqField>(out + t * 24, pt_to_affine(load_bucket<FqField>(g1_pts + t * Grp<FqField>::BK_WORDS)));
    else if (t < 4 + 48) out[96 + (t - 4)] = b2_ark[t - 4];
}
__global__ __launch_bounds__(64) void groth16_combine_kernel(uint32_t world, const uint32_t *__restrict__ partials /* [world][144] */,
                                                             uint32_t *__restrict__ g1_pts /* [4][BK_WORDS] */, uint32_t *__restrict__ b2_ark /* [48] */)
{
    const uint32_t t = threadIdx.x;
    if (t < 4) {
        XyzzT<FqField> acc = pt_identity<FqField>();
        for (uint32_t r = 0; r < world; r++) acc = pt_add_affine(acc, load_ark_point<FqField>(partials + (size_t)r * 144 + t * 24));
        store_bucket<FqField>(g1_pts + t * Grp<FqField>::BK_WORDS, acc);
    } else if (t == 4) {
        XyzzT<Fq2Field> acc = pt_identity<Fq2Field>();
        for (uint32_t r = 0; r < world; r++) acc = pt_add_affine(acc, load_ark_point<Fq2Field>(partials + (size_t)r * 144 + 96));
        store_ark_point<Fq2Field>(b2_ark, pt_to_affine(acc));
    }
}
}  // namespace frw

struct frw_groth16_pk {
    int device;
    uint64_t num_instance, num_witness, domain_size;
    frw_msm *h, *a, *b1, *l, *b2;
    // round 5: a key of bare handles (the points only: 83 GB for the 1,024-statement aggregate instead of 3.6 KB per variable), and a
    // key in slices: this handle holds rows [z_lo, z_hi) of the nv + 3 rows of the four witness-side tables and [h_lo, h_hi) of h_query
    bool bare = false;
    uint32_t rank = 0, world = 1;
    uint64_t z_lo = 0, z_hi = 0, h_lo = 0, h_hi = 0;
    // (bare) the rows of b_g2_query -- and with them of b_g1_query -- that hold a point, in order: the sums over these two tables are over
    // those rows only.  A variable that no constraint has on its B side has the point at infinity there: 41 % of a Falcon circuit's, and
    // with them half of a witness's ones and all its full-size values -- 114 M additions per table against 37.5 M (the 1,024-statement mix).
    uint32_t *b_index = nullptr;
    uint32_t b_rows = 0;
    // FOUR streams of the key's own, created one after the other -- as many as the device has hardware queues (HIP's default), so
    // that no two of them wait in line for the same queue.  `main`: the witness map + the sum over h_query; side[0]: the three G1
    // witness-side sums as one chain of kernels, then both scalar multiplications; side[2]: the sum over b_g2_query; side[1]: the G1
    // sums' ones when latency counts.  The caller's stream (a fifth stream on four queues: it shares one, which one is not ours to know)
    // sorts the scalars' digits, forks into the others by events and joins them again (frw_groth16_prove_dev has the order things are
    // enqueued in, and why).  Two things learnt from timelines (profiles/r03_groth16_timeline_64.txt, r04_groth16_timeline_64.txt,
    // r04_aggregate16_timeline.txt): when the caller's stream carried the witness map it sat behind a side stream's whole chain on a
    // shared queue (a quarter of the call); and a kernel whose workgroups have several wavefronts (the sorts' 1,024 threads) does not
    // start once the 2^18-point bucket kernel runs -- two wavefronts of 240 registers per SIMD leave no CU with room for such a
    // workgroup -- so the sort goes first and everything after it is grids of single wavefronts.
    hipStream_t main, side[3];
    hipEvent_t fork, sorted, ones_done, join[4];   // fork: the call's inputs are in place; sorted: the digits of its scalars too; join[3]: main
    std::mutex enqueue;         // the side streams and events are the key's: one call at a time puts its work on them
};

extern "C" void frw_groth16_pk_free(frw_groth16_pk *pk)
{
    if (!pk) return;
    for (frw_msm *m : {pk->h, pk->a, pk->b1, pk->l, pk->b2}) frw_msm_free(m);
    (void)hipSetDevice(pk->device);
    if (pk->b_index) (void)hipFree(pk->b_index);
    for (int i = 0; i < 4; i++) {
        if (i < 3 && pk->side[i]) (void)hipStreamDestroy(pk->side[i]);
        if (pk->join[i]) (void)hipEventDestroy(pk->join[i]);
    }
    if (pk->main) (void)hipStreamDestroy(pk->main);
    if (pk->fork) (void)hipEventDestroy(pk->fork);
    if (pk->sorted) (void)hipEventDestroy(pk->sorted);
    if (pk->ones_done) (void)hipEventDestroy(pk->ones_done);
    delete pk;
}

namespace {
// the key's own streams and events (see the struct)
int pk_create(int device, uint64_t ni, uint64_t nw, uint64_t n, frw_groth16_pk **out)
{
    frw_groth16_pk *pk = new (std::nothrow) frw_groth16_pk;
    if (!pk) return FRW_E_OUT_OF_MEMORY;
    pk->device = device;
    pk->num_instance = ni; pk->num_witness = nw; pk->domain_size = n;
    pk->h = pk->a = pk->b1 = pk->l = pk->b2 = nullptr;
    pk->fork = pk->sorted = pk->ones_done = nullptr;
    pk->main = nullptr;
    for (int i = 0; i < 4; i++) { if (i < 3) pk->side[i] = nullptr; pk->join[i] = nullptr; }
    pk->z_lo = 0; pk->z_hi = ni + nw + 3; pk->h_lo = 0; pk->h_hi = n - 1;
    hipError_t e = hipSetDevice(device);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pk->fork, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pk->sorted, hipEventDisableTiming);
    if (e == hipSuccess) e = hipEventCreateWithFlags(&pk->ones_done, hipEventDisableTiming);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&pk->main, hipStreamNonBlocking);
    for (int i = 0; i < 4 && e == hipSuccess; i++) {
        if (i < 3) e = hipStreamCreateWithFlags(&pk->side[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&pk->join[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) {
        frw_groth16_pk_free(pk);
        return frw::record_hip_error(e, "frw_groth16_pk_load");
    }
    *out = pk;
    return FRW_OK;
}
// a bare handle with room for `rows` rows and nothing in them yet
template <class F> int msm_alloc_bare_t(int device, int group, int window_bits, size_t rows, uint64_t row_lo, frw_msm **out)
{
    if (!out || rows == 0 || rows >= ((size_t)1 << 31)) return FRW_E_INVALID_ARG;
    *out = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return frw::record_hip_error(e, "hipSetDevice");
    frw_msm *m = new (std::nothrow) frw_msm;
    if (!m) return FRW_E_OUT_OF_MEMORY;
    m->device = device; m->group = group; m->window_bits = window_bits; m->bare = true; m->row_lo = row_lo;
    m->wide = window_bits == 16 && rows >= msm_wide_from();
    m->table = nullptr; m->ones_table = nullptr;
    m->dev.n = (uint32_t)rows; m->dev.ones_table = nullptr;
    e = hipMalloc(&m->table, rows * frw::Grp<F>::PT_WORDS * 4);
    if (e != hipSuccess) { delete m; return frw::record_hip_error(e, "frw_msm: the key's rows"); }
    m->dev.table = (const uint32_t *)m->table;
    *out = m;
    return FRW_OK;
}
// rows [first, first + count) of a bare handle from `count` ark-ff points in HOST memory, a few megabytes at a time
template <class F> int msm_upload_rows(frw_msm *m, size_t first, size_t count, const std::function<const uint64_t *(size_t)> &point_of /* null: infinity */)
{
    constexpr size_t AW = frw::Grp<F>::ARK_WORDS / 2, CHUNK = (size_t)1 << 18;
    std::vector<uint64_t> stage(std::min(count, CHUNK) * AW);
    void *d_stage = nullptr;
    hipError_t e = hipSetDevice(m->device);
    if (e == hipSuccess) e = hipMalloc(&d_stage, stage.size() * 8);
    for (size_t lo = 0; e == hipSuccess && lo < count; lo += CHUNK) {
        const size_t cnt = std::min(CHUNK, count - lo);
        for (size_t i = 0; i < cnt; i++) {
            const uint64_t *src = point_of(lo + i);
            if (src) std::memcpy(&stage[i * AW], src, AW * 8);
            else std::memset(&stage[i * AW], 0, AW * 8);
        }
        e = hipMemcpy(d_stage, stage.data(), cnt * AW * 8, hipMemcpyHostToDevice);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(frw::msm_precompute_kernel<F>, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, nullptr, (uint32_t)cnt, (const uint32_t *)d_stage,
                           (uint32_t *)m->table + (first + lo) * frw::Grp<F>::PT_WORDS, 256);
        e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
    if (d_stage) (void)hipFree(d_stage);
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_groth16_pk_load: rows");
}
}  // namespace

namespace {
// a bare handle -> the handle with window tables over the same points (the bare one is freed)
template <class F> int msm_expand_tables_t(frw_msm **pm)
{
    frw_msm *b = *pm;
    const size_t n = b->dev.n;
    if (!b->bare || n > ((size_t)1 << (b->window_bits == 8 ? 25 : 26))) return FRW_E_INVALID_ARG;
    hipError_t e = hipSetDevice(b->device);
    frw_msm *m = new (std::nothrow) frw_msm;
    if (!m) return FRW_E_OUT_OF_MEMORY;
    m->device = b->device; m->group = b->group; m->window_bits = b->window_bits; m->bare = false; m->row_lo = 0;
    m->table = nullptr; m->ones_table = nullptr; m->dev.n = (uint32_t)n; m->dev.ones_table = nullptr;
    if (e == hipSuccess) e = hipMalloc(&m->table, (size_t)(256 / b->window_bits) * n * frw::Grp<F>::PT_WORDS * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(frw::msm_precompute_rows_kernel<F>, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, nullptr, (uint32_t)n, (const uint32_t *)b->table,
                           (uint32_t *)m->table, b->window_bits);
        e = hipGetLastError();
    }
    if (e == hipSuccess && b->window_bits == 8 && n <= ((size_t)1 << 18)) {
        const size_t entries = (n + 7) / 8 * 255;
        e = hipMalloc(&m->ones_table, entries * frw::Grp<F>::PT_WORDS * 4);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(frw::msm_ones_table_kernel<F>, dim3((unsigned)((entries + 63) / 64)), dim3(64), 0, nullptr, (uint32_t)n,
                               (const uint32_t *)m->table, (uint32_t *)m->ones_table);
            e = hipGetLastError();
        }
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e != hipSuccess) {
        frw_msm_free(m);
        return frw::record_hip_error(e, "frw_groth16_setup: window tables");
    }
    m->dev.table = (const uint32_t *)m->table;
    m->dev.ones_table = (const uint32_t *)m->ones_table;
    frw_msm_free(b);
    *pm = m;
    return FRW_OK;
}
}  // namespace

namespace frw {
int msm_expand_tables(frw_msm **m) { return (*m)->group == 1 ? msm_expand_tables_t<FqField>(m) : msm_expand_tables_t<Fq2Field>(m); }
// slice `rank` of `world` of `total` rows
void groth16_shard_range(uint64_t total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi)
{
    if (world <= 1) { *lo = 0; *hi = total; return; }
    *lo = total / world * rank + std::min<uint64_t>(rank, total % world);
    *hi = *lo + total / world + (rank < total % world ? 1 : 0);
}
int msm_alloc_bare(int device, int group, int window_bits, size_t rows, uint64_t row_lo, frw_msm **out)
{
    return group == 1 ? msm_alloc_bare_t<FqField>(device, 1, window_bits, rows, row_lo, out)
                      : msm_alloc_bare_t<Fq2Field>(device, 2, window_bits, rows, row_lo, out);
}
int fixed_base_gen_create(int device, FixedBaseGen *g)
{
    g->g1 = g->g2 = nullptr;
    hipError_t e = hipSetDevice(device);
    void *gens1 = nullptr, *gens2 = nullptr;
    if (e == hipSuccess) e = hipMalloc((void **)&g->g1, (size_t)FBW_WINDOWS * FBW_DIGITS * Grp<FqField>::PT_WORDS * 4);
    if (e == hipSuccess) e = hipMalloc((void **)&g->g2, (size_t)FBW_WINDOWS * FBW_DIGITS * Grp<Fq2Field>::PT_WORDS * 4);
    if (e == hipSuccess) e = hipMalloc(&gens1, (size_t)FBW_WINDOWS * Grp<FqField>::PT_WORDS * 4);
    if (e == hipSuccess) e = hipMalloc(&gens2, (size_t)FBW_WINDOWS * Grp<Fq2Field>::PT_WORDS * 4);
    if (e == hipSuccess) {
        hipLaunchKernelGGL(fixed_base_window_gens_kernel<FqField>, dim3(1), dim3(64), 0, nullptr, (uint32_t *)gens1);
        hipLaunchKernelGGL(fixed_base_window_gens_kernel<Fq2Field>, dim3(1), dim3(64), 0, nullptr, (uint32_t *)gens2);
        hipLaunchKernelGGL(fixed_base_wide_table_kernel<FqField>, dim3(FBW_WINDOWS * FBW_DIGITS / 64), dim3(64), 0, nullptr, (const uint32_t *)gens1, g->g1);
        hipLaunchKernelGGL(fixed_base_wide_table_kernel<Fq2Field>, dim3(FBW_WINDOWS * FBW_DIGITS / 64), dim3(64), 0, nullptr, (const uint32_t *)gens2, g->g2);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (gens1) (void)hipFree(gens1);
    if (gens2) (void)hipFree(gens2);
    if (e != hipSuccess) { fixed_base_gen_free(g); return record_hip_error(e, "fixed-base tables"); }
    return FRW_OK;
}
void fixed_base_gen_free(FixedBaseGen *g)
{
    if (g->g1) (void)hipFree(g->g1);
    if (g->g2) (void)hipFree(g->g2);
    g->g1 = g->g2 = nullptr;
}
// rows [first_row, first_row + count) of a bare handle = d_scalars[i] x the group's generator (canonical scalars in device memory)
hipError_t msm_fill_fixed_base(frw_msm *m, const FixedBaseGen &g, size_t first_row, size_t count, const uint32_t *d_scalars, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    if (!m->bare || first_row + count > m->dev.n) return hipErrorInvalidValue;
    if (m->group == 1)
        hipLaunchKernelGGL((fixed_base_wide_kernel<FqField, false>), dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, count, d_scalars, (const uint32_t *)g.g1,
                           (uint32_t *)m->table + first_row * Grp<FqField>::PT_WORDS);
    else
        hipLaunchKernelGGL((fixed_base_wide_kernel<Fq2PairField, false>), dim3((unsigned)((count + 31) / 32)), dim3(64), 0, st, count, d_scalars,
                           (const uint32_t *)g.g2, (uint32_t *)m->table + first_row * Grp<Fq2Field>::PT_WORDS);
    return hipGetLastError();
}
// ... as ark-ff's affine bytes in device memory (the verifying key's points)
hipError_t fixed_base_ark_dev(const FixedBaseGen &g, int group, size_t count, const uint32_t *d_scalars, uint32_t *d_out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    if (group == 1)
        hipLaunchKernelGGL((fixed_base_wide_kernel<FqField, true>), dim3((unsigned)((count + 63) / 64)), dim3(64), 0, st, count, d_scalars, (const uint32_t *)g.g1, d_out);
    else
        hipLaunchKernelGGL((fixed_base_wide_kernel<Fq2PairField, true>), dim3((unsigned)((count + 31) / 32)), dim3(64), 0, st, count, d_scalars,
                           (const uint32_t *)g.g2, d_out);
    return hipGetLastError();
}
// a key of bare handles from its five tables (ownership passes to the key, also on failure)
int groth16_pk_assemble(int device, uint64_t ni, uint64_t nw, uint64_t n, uint32_t rank, uint32_t world, frw_msm *h, frw_msm *a, frw_msm *b1, frw_msm *l,
                        frw_msm *b2, frw_groth16_pk **out)
{
    const bool bare = h->bare;
    frw_groth16_pk *pk = nullptr;
    const int rc = pk_create(device, ni, nw, n, &pk);
    if (rc != FRW_OK) {
        for (frw_msm *m : {h, a, b1, l, b2}) frw_msm_free(m);
        return rc;
    }
    pk->h = h; pk->a = a; pk->b1 = b1; pk->l = l; pk->b2 = b2;
    pk->bare = bare;
    pk->rank = rank; pk->world = world < 1 ? 1 : world;
    groth16_shard_range(ni + nw + 3, rank, pk->world, &pk->z_lo, &pk->z_hi);
    groth16_shard_range(n - 1, rank, pk->world, &pk->h_lo, &pk->h_hi);
    if (bare) {
        // the rows of b_g2_query that hold a point (b_g1_query's are the same variables': row nv + 2 -- delta_2 -- is the one more G2 has)
        const uint32_t rows = b2->dev.n, blocks = (rows + 1023) / 1024;
        uint32_t *counts = nullptr, live = 0;
        hipError_t e = hipSetDevice(device);
        if (e == hipSuccess) e = hipMalloc((void **)&counts, ((size_t)blocks + 1) * 4);
        if (e == hipSuccess) {
            constexpr int PW2 = Grp<Fq2Field>::PT_WORDS;
            hipLaunchKernelGGL(live_rows_kernel<PW2>, dim3(blocks), dim3(1024), 0, nullptr, (const uint32_t *)b2->dev.table, rows, counts, (uint32_t *)nullptr);
            hipLaunchKernelGGL(live_rows_scan_kernel, dim3(1), dim3(1024), 0, nullptr, counts, blocks, counts + blocks);
            e = hipMemcpy(&live, counts + blocks, 4, hipMemcpyDeviceToHost);
            // (a slice without a single point in these tables still sums something: row 0, the point at infinity)
            if (e == hipSuccess) e = hipMalloc((void **)&pk->b_index, (size_t)(live ? live : 1) * 4);
            if (e == hipSuccess) e = hipMemset(pk->b_index, 0, 4);
            if (e == hipSuccess && live) {
                hipLaunchKernelGGL(live_rows_kernel<PW2>, dim3(blocks), dim3(1024), 0, nullptr, (const uint32_t *)b2->dev.table, rows, counts, pk->b_index);
                e = hipDeviceSynchronize();
            }
            pk->b_rows = live ? live : 1;
        }
        if (counts) (void)hipFree(counts);
        if (e != hipSuccess) {
            frw_groth16_pk_free(pk);
            return record_hip_error(e, "frw_groth16_pk_load (the rows of b_g2_query that hold a point)");
        }
    }
    *out = pk;
    return FRW_OK;
}
}  // namespace frw

extern "C" int frw_groth16_pk_load_opts(int device, const frw_groth16_pk_desc_t *d, const frw_groth16_key_opts_t *opts, frw_groth16_pk **out)
{
    if (!out || !d || !d->alpha_g1 || !d->beta_g1 || !d->delta_g1 || !d->beta_g2 || !d->delta_g2 || !d->a_query || !d->b_g1_query ||
        !d->b_g2_query || !d->h_query || !d->l_query || d->num_instance == 0 || d->domain_size < 2)
        return FRW_E_INVALID_ARG;
    *out = nullptr;
    const size_t ni = (size_t)d->num_instance, nv = (size_t)(d->num_instance + d->num_witness);
    int mode = opts ? opts->mode : FRW_KEY_AUTO;
    const uint32_t world = opts && opts->world > 1 ? opts->world : 1, rank = opts ? opts->rank : 0;
    if ((mode != FRW_KEY_AUTO && mode != FRW_KEY_TABLES && mode != FRW_KEY_BARE) || rank >= world) return FRW_E_INVALID_ARG;
    if (mode == FRW_KEY_AUTO) mode = world > 1 || nv > FRW_KEY_AUTO_TABLE_VARIABLES ? FRW_KEY_BARE : FRW_KEY_TABLES;
    if (mode == FRW_KEY_TABLES && world > 1) return FRW_E_INVALID_ARG;          // a key in slices is a key of bare handles
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FRW_E_NO_DEVICE;
    if (mode == FRW_KEY_BARE) {
        // rows [z_lo, z_hi) of the four witness-side tables (a_query ++ [alpha, delta, O], b_g1_query ++ [beta, O, O],
        // O x I ++ l_query ++ [O, O, O], b_g2_query ++ [beta2, O, delta2]) and [h_lo, h_hi) of h_query, as they are
        uint64_t z_lo, z_hi, h_lo, h_hi;
        frw::groth16_shard_range(nv + 3, rank, world, &z_lo, &z_hi);
        frw::groth16_shard_range((uint64_t)d->domain_size - 1, rank, world, &h_lo, &h_hi);
        if (z_hi == z_lo || h_hi == h_lo) return FRW_E_INVALID_ARG;             // more ranks than rows
        frw_msm *t[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};          // h, a, b1, l, b2
        int rc = frw::msm_alloc_bare(device, 1, 16, h_hi - h_lo, h_lo, &t[0]);
        for (int k = 1; k < 5 && rc == FRW_OK; k++) rc = frw::msm_alloc_bare(device, k == 4 ? 2 : 1, 8, z_hi - z_lo, z_lo, &t[k]);
        try {
            if (rc == FRW_OK) rc = msm_upload_rows<FqField>(t[0], 0, h_hi - h_lo, [&](size_t i) { return d->h_query + 12 * (h_lo + i); });
            if (rc == FRW_OK)
                rc = msm_upload_rows<FqField>(t[1], 0, z_hi - z_lo, [&](size_t i) -> const uint64_t * {
                    const size_t g = z_lo + i;
                    return g < nv ? d->a_query + 12 * g : g == nv ? d->alpha_g1 : g == nv + 1 ? d->delta_g1 : nullptr;
                });
            if (rc == FRW_OK)
                rc = msm_upload_rows<FqField>(t[2], 0, z_hi - z_lo, [&](size_t i) -> const uint64_t * {
                    const size_t g = z_lo + i;
                    return g < nv ? d->b_g1_query + 12 * g : g == nv ? d->beta_g1 : nullptr;
                });
            if (rc == FRW_OK)
                rc = msm_upload_rows<FqField>(t[3], 0, z_hi - z_lo, [&](size_t i) -> const uint64_t * {
                    const size_t g = z_lo + i;
                    return g >= ni && g < nv ? d->l_query + 12 * (g - ni) : nullptr;
                });
            if (rc == FRW_OK)
                rc = msm_upload_rows<Fq2Field>(t[4], 0, z_hi - z_lo, [&](size_t i) -> const uint64_t * {
                    const size_t g = z_lo + i;
                    return g < nv ? d->b_g2_query + 24 * g : g == nv ? d->beta_g2 : g == nv + 2 ? d->delta_g2 : nullptr;
                });
        } catch (const std::exception &) {
            rc = FRW_E_OUT_OF_MEMORY;
        }
        if (rc != FRW_OK) {
            for (frw_msm *m : t) frw_msm_free(m);
            return rc;
        }
        return frw::groth16_pk_assemble(device, d->num_instance, d->num_witness, d->domain_size, rank, world, t[0], t[1], t[2], t[3], t[4], out);
    }
    frw_groth16_pk *pk = nullptr;
    int rc = pk_create(device, d->num_instance, d->num_witness, d->domain_size, &pk);
    if (rc != FRW_OK) return rc;
    try {
        // The four witness-side sums take the same scalars, z ++ [1, r, s] (nv + 3 of them), and so ONE sort of their digits: every
        // table has nv + 3 rows, the points that a sum does not have being the point at infinity (all-zero rows, which cost an
        // addition nothing): a_query ++ [alpha, delta, O], b_g1_query ++ [beta, O, O], O x I ++ l_query ++ [O, O, O], and
        // b_g2_query ++ [beta2, O, delta2].
        std::vector<uint64_t> g1(12 * (nv + 3)), g2(24 * (nv + 3), 0);
        std::fill(g1.begin(), g1.end(), 0);
        std::memcpy(g1.data(), d->a_query, nv * 96);
        std::memcpy(g1.data() + 12 * nv, d->alpha_g1, 96);
        std::memcpy(g1.data() + 12 * (nv + 1), d->delta_g1, 96);
        rc = frw_msm_g1_load_narrow(device, nv + 3, g1.data(), &pk->a);       // the four witness-side sums: narrow windows
        if (rc == FRW_OK) {
            std::fill(g1.begin(), g1.end(), 0);
            std::memcpy(g1.data(), d->b_g1_query, nv * 96);
            std::memcpy(g1.data() + 12 * nv, d->beta_g1, 96);
            rc = frw_msm_g1_load_narrow(device, nv + 3, g1.data(), &pk->b1);
        }
        if (rc == FRW_OK) {
            std::fill(g1.begin(), g1.end(), 0);
            std::memcpy(g1.data() + 12 * (size_t)d->num_instance, d->l_query, (size_t)d->num_witness * 96);
            rc = frw_msm_g1_load_narrow(device, nv + 3, g1.data(), &pk->l);
        }
        if (rc == FRW_OK) rc = frw_msm_g1_load(device, (size_t)d->domain_size - 1, d->h_query, &pk->h);
        if (rc == FRW_OK) {
            std::memcpy(g2.data(), d->b_g2_query, nv * 192);
            std::memcpy(g2.data() + 24 * nv, d->beta_g2, 192);                 // row nv + 1 stays the point at infinity (the scalar there is r)
            std::memcpy(g2.data() + 24 * (nv + 2), d->delta_g2, 192);
            rc = frw_msm_g2_load_narrow(device, nv + 3, g2.data(), &pk->b2);
        }
    } catch (const std::exception &) {
        rc = FRW_E_OUT_OF_MEMORY;
    }
    if (rc != FRW_OK) {
        frw_groth16_pk_free(pk);
        return rc;
    }
    *out = pk;
    return FRW_OK;
}
extern "C" int frw_groth16_pk_load(int device, const frw_groth16_pk_desc_t *d, frw_groth16_pk **out)
{
    const frw_groth16_key_opts_t opts = {FRW_KEY_TABLES, 0, 1};                // (as before round 5: window tables, whatever the size)
    return frw_groth16_pk_load_opts(device, d, &opts, out);
}
extern "C" const frw_msm *frw_groth16_pk_query(const frw_groth16_pk *pk, int which)
{
    if (!pk) return nullptr;
    return which == FRW_QUERY_H ? pk->h : which == FRW_QUERY_A ? pk->a : which == FRW_QUERY_B1 ? pk->b1 : which == FRW_QUERY_L ? pk->l
         : which == FRW_QUERY_B2 ? pk->b2 : nullptr;
}
extern "C" int frw_groth16_pk_info(const frw_groth16_pk *pk, frw_groth16_pk_info_t *out)
{
    if (!pk || !out) return FRW_E_INVALID_ARG;
    out->mode = pk->bare ? FRW_KEY_BARE : FRW_KEY_TABLES;
    out->rank = pk->rank; out->world = pk->world;
    out->z_lo = pk->z_lo; out->z_hi = pk->z_hi; out->h_lo = pk->h_lo; out->h_hi = pk->h_hi;
    out->key_bytes = 0;
    frw_msm_info_t mi;
    for (const frw_msm *m : {pk->h, pk->a, pk->b1, pk->l, pk->b2})
        if (frw_msm_info(m, &mi) == FRW_OK) out->key_bytes += mi.table_bytes;
    return FRW_OK;
}

namespace {
struct Groth16Sizes { size_t qap, h, zext, msm[5], msm_all, pts, per; };
Groth16Sizes groth16_sizes(const frw_groth16_pk *pk, const frw_r1cs *r)
{
    Groth16Sizes s{};
    frw_qap_info_t q;
    frw_msm_info_t mi;
    if (frw_qap_info(r, &q) != FRW_OK) return s;
    const size_t nv = (size_t)(pk->num_instance + pk->num_witness);
    s.qap = q.workspace_bytes_per_signature;
    s.h = (size_t)pk->domain_size * 32;
    s.zext = (nv + 3) * 32;
    if (pk->bare) {
        // msm[0]: the sum over h_query (its window rows); msm[1]: the sort of the slice's scalars and the own arrays of a_query and
        // l_query; msm[2]: the sort of the scalars of the rows b_g1_query / b_g2_query hold a point in (pk->b_index) and b_g1_query's own
        // arrays; msm[4]: b_g2_query's own arrays (it reads msm[2]'s sort)
        const uint32_t nz = pk->a->dev.n;
        char *const base = (char *)(uintptr_t)4096;
        frw_msm_info(pk->h, &mi);
        s.msm[0] = (mi.workspace_bytes_per_signature + 255) & ~(size_t)255;
        const NmsmBufs g1 = nmsm_carve_bare<FqField>(base, 2, nz);
        s.msm[1] = ((size_t)((char *)g1.end - base) + 255) & ~(size_t)255;
        const NmsmBufs gb = nmsm_carve_bare<FqField>(base, 1, pk->b_rows);
        s.msm[2] = ((size_t)((char *)gb.end - base) + 255) & ~(size_t)255;
        const NmsmBufs g2 = nmsm_carve_bare<Fq2Field>(base, 1, pk->b_rows, &gb);
        s.msm[4] = ((size_t)((char *)g2.end - base) + 255) & ~(size_t)255;
        // (the witness map's workspace and the sum over h_query's are ONE region: the sum starts, on the same stream, when the map is
        // through and has left h -- 30 GB of the 2^27 domain's workspace)
        s.qap = (s.qap + 255) & ~(size_t)255;
        s.msm_all = std::max(s.msm[0], s.qap) - s.qap + s.msm[1] + s.msm[2] + s.msm[4];
        s.pts = 6 * (size_t)frw::Grp<FqField>::BK_WORDS * 4 + 192 + 64 + 64;
        s.per = ((s.qap + s.h + s.zext + s.msm_all + s.pts) + 255) & ~(size_t)255;
        return s;
    }
    int i = 0;
    for (const frw_msm *m : {pk->h, pk->a, pk->b1, pk->l, pk->b2}) {                // each sum has a workspace of its own: they overlap in time
        frw_msm_info(m, &mi);
        s.msm[i] = (mi.workspace_bytes_per_signature + 255) & ~(size_t)255;
        s.msm_all += s.msm[i++];
    }
    // A, B1', L, H, s A, r B1' (G1, XYZZ: 240 bytes each), B (G2, affine), r and s, and their split halves
    s.pts = 6 * (size_t)frw::Grp<FqField>::BK_WORDS * 4 + 192 + 64 + 64;
    s.per = ((s.qap + s.h + s.zext + s.msm_all + s.pts) + 255) & ~(size_t)255;
    return s;
}
}  // namespace

extern "C" size_t frw_groth16_workspace_bytes(const frw_groth16_pk *pk, const frw_r1cs *r, size_t batch_in_flight)
{
    if (!pk || !r) return 0;
    return groth16_sizes(pk, r).per * (pk->bare ? (batch_in_flight ? 1 : 0) : batch_in_flight);     // a bare key proves one statement at a time
}

// (diagnostic: what the witness-side sums of a key of bare handles add up for the scalars z -- the benchmark prices its roofline with these)
extern "C" int frw_diag_groth16_side_counts(const frw_groth16_pk *pk, const uint64_t *d_z, void *d_workspace, size_t workspace_bytes, void *stream,
                                            uint64_t *out)
{
    if (!pk || !pk->bare || !d_z || !d_workspace || !out || ((uintptr_t)d_workspace & 255)) return FRW_E_INVALID_ARG;
    const uint32_t nz = pk->a->dev.n;
    const NmsmBufs g1 = nmsm_carve_bare<FqField>(d_workspace, 2, nz);
    const size_t first = ((size_t)((char *)g1.end - (char *)d_workspace) + 255) & ~(size_t)255;
    const NmsmBufs gb = nmsm_carve_bare<FqField>((char *)d_workspace + first, 1, pk->b_rows);
    if ((size_t)((char *)gb.end - (char *)d_workspace) > workspace_bytes) return FRW_E_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSetDevice(pk->device);
    if (e == hipSuccess) e = nmsm_sort_bare(g1, nz, (const uint32_t *)d_z, 1, st);
    if (e == hipSuccess) e = nmsm_sort_bare(gb, pk->b_rows, (const uint32_t *)d_z, 1, st, pk->b_index);
    out[0] = pk->b_rows;
    for (int k = 0; k < 2 && e == hipSuccess; k++) {
        const NmsmBufs &b = k ? gb : g1;
        std::vector<uint32_t> counts((size_t)frw::NMSM_W * frw::NMSM_BUCKETS);
        uint32_t ones = 0;
        e = hipMemcpyAsync(counts.data(), b.counts, counts.size() * 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipMemcpyAsync(&ones, b.ones_count, 4, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        uint64_t total = 0;
        for (uint32_t c : counts) total += c;
        out[1 + 2 * k] = total;
        out[2 + 2 * k] = ones;
    }
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_diag_groth16_side_counts");
}

namespace {
// A key of bare handles: one statement at a time (its sums fill the chip by themselves: 2 x 10^9 bucket additions for the 1,024-statement
// aggregate).  The same five sums, the same streams; the sort of the scalars' digits is the bare handles' (thirty-two windows, one pass),
// shared by the three G1 tables -- one chain of kernels -- and the G2 table; the sum over h_query runs window by window.
// d_proofs: [batch][48] (a whole key), or d_partial: [batch][72] (a slice: the five partial sums, frw.h FRW_GROTH16_PARTIAL_WORDS).
int groth16_prove_bare(const frw_groth16_pk *pk, const frw_r1cs *r1cs, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                       const uint64_t *rs_host, const uint64_t *rs_dev, uint64_t *d_proofs, uint64_t *d_partial, uint32_t *d_num_unsatisfied,
                       void *d_workspace, size_t workspace_bytes, void *stream)
{
    const Groth16Sizes sz = groth16_sizes(pk, r1cs);
    if (workspace_bytes < sz.per || ((uintptr_t)d_workspace & 255)) return FRW_E_INVALID_ARG;
    const size_t I = (size_t)pk->num_instance, W = (size_t)pk->num_witness, nv = I + W, n = (size_t)pk->domain_size, stride = nv + 3;
    const uint32_t nz = pk->a->dev.n;                                    // rows of this slice of the witness-side tables
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSetDevice(pk->device);
    int rc = FRW_OK;
    bool forked = false;
    std::unique_lock<std::mutex> lock(const_cast<frw_groth16_pk *>(pk)->enqueue, std::defer_lock);
    for (size_t b = 0; e == hipSuccess && rc == FRW_OK && b < batch; b++) {
        if (lock.owns_lock()) lock.unlock();
        char *base = (char *)d_workspace;
        void *qap_ws = base;
        char *h_ws = base;                            base += std::max(sz.qap, sz.msm[0]);        // one region, one stream: groth16_sizes
        uint64_t *h = (uint64_t *)base;               base += sz.h;
        uint64_t *zext = (uint64_t *)base;            base += sz.zext;
        char *g1_ws = base;                           base += sz.msm[1];
        char *gb_ws = base;                           base += sz.msm[2];
        char *g2_ws = base;                           base += sz.msm[4];
        constexpr size_t XW = frw::Grp<FqField>::BK_WORDS;
        uint32_t *pA = (uint32_t *)base, *pH = pA + 3 * XW, *pSA = pH + XW, *pRB1 = pSA + XW;      // [A | B1' | L] [H] [s A | r B1']
        uint64_t *pB2 = (uint64_t *)(pRB1 + XW), *d_rs = pB2 + 24, *d_split = d_rs + 8;
        const uint64_t *wit = d_witness + b * W * 4, *inst = d_instance + b * I * 4;
        const uint64_t *rs_now = rs_dev ? rs_dev + b * 8 : d_rs;
        if (rs_host) {
            e = hipMemcpyAsync(d_rs, rs_host + b * 8, 64, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) break;
        }
        lock.lock();
        hipLaunchKernelGGL(frw::groth16_split_kernel, dim3(1), dim3(64), 0, st, (size_t)2, rs_now, d_split);
        e = hipMemcpyAsync(zext, inst, I * 32, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpyAsync(zext + I * 4, wit, W * 32, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(frw::groth16_tails_kernel, dim3(1), dim3(64), 0, st, (size_t)1, (const uint32_t *)rs_now, (uint32_t *)zext, stride * 8, nv);
        e = hipEventRecord(pk->fork, st);
        forked = true;
        if (e == hipSuccess) e = hipStreamWaitEvent(pk->main, pk->fork, 0);
        if (e != hipSuccess) break;
        // the sorts first, on the caller's stream (workgroups of 1,024 threads: they would not start under the bucket kernels): one over
        // all the slice's scalars for a_query and l_query, one over the rows that b_g1_query / b_g2_query hold a point in
        const NmsmBufs g1 = nmsm_carve_bare<FqField>(g1_ws, 2, nz);
        const NmsmBufs gb = nmsm_carve_bare<FqField>(gb_ws, 1, pk->b_rows);
        const NmsmBufs g2 = nmsm_carve_bare<Fq2Field>(g2_ws, 1, pk->b_rows, &gb);
        // (both before any sum starts: a sort's workgroups are 1,024 threads of 70 - 128 registers -- a whole CU's register file -- and
        // with the G2 sum's kernels already on the chip the long sort's scatter took 39 ms instead of 7: profiles/r05_witness_sorts.txt)
        e = nmsm_sort_bare(g1, nz, (const uint32_t *)(zext + pk->z_lo * 4), 1, st);
        if (e == hipSuccess) e = nmsm_sort_bare(gb, pk->b_rows, (const uint32_t *)(zext + pk->z_lo * 4), 1, st, pk->b_index);
        if (e != hipSuccess) break;
        e = hipEventRecord(pk->sorted, st);
        for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipStreamWaitEvent(pk->side[i], pk->sorted, 0);
        if (e != hipSuccess) break;
        // (the witness-side sums first: the witness map of a mixed aggregate is hundreds of launches, and what the one host thread enqueues
        // behind them starts that much later -- 50 ms of an idle chip at the head of the 1,024-statement proof, profiles/r05_aggregate1024_timeline.txt)
        // pA: [A | B1' | L]: a_query and l_query share a chain of kernels (rows 0 and 2), b_g1_query has the key's spare stream
        const frw_msm *g1s[2] = {pk->a, pk->l};
        e = nmsm_accumulate_bare<FqField, true>(g1s, 2, g1, pA, pk->side[0], true, 2);
        if (e != hipSuccess) break;
        e = nmsm_accumulate_bare<Fq2Field, false>(&pk->b2, 1, g2, (uint32_t *)pB2, pk->side[2], false);
        if (e != hipSuccess) break;
        e = nmsm_accumulate_bare<FqField, true>(&pk->b1, 1, gb, pA + XW, pk->side[1], true);
        if (e != hipSuccess) break;
        rc = frw_qap_witness_map_dev(r1cs, 1, wit, inst, h, d_num_unsatisfied ? d_num_unsatisfied + b : nullptr, qap_ws, sz.qap, pk->main);
        if (rc != FRW_OK) break;
        // h's coefficients [h_lo, h_hi) against this slice of h_query
        rc = msm_run<FqField, true>(pk->h, 1, h + pk->h_lo * 4, n, 1, (uint64_t *)pH, h_ws, sz.msm[0], pk->main, true);
        if (rc != FRW_OK) break;
        for (int i = 0; i < 3 && e == hipSuccess; i++) e = hipEventRecord(pk->join[i], pk->side[i]);
        if (e == hipSuccess) e = hipEventRecord(pk->join[3], pk->main);
        for (int i = 0; i < 4 && e == hipSuccess; i++) e = hipStreamWaitEvent(st, pk->join[i], 0);
        if (e != hipSuccess) break;
        if (d_partial) {
            hipLaunchKernelGGL(frw::groth16_partial_kernel, dim3(1), dim3(64), 0, st, (const uint32_t *)pA, (const uint32_t *)pB2,
                               (uint32_t *)(d_partial + b * FRW_GROTH16_PARTIAL_WORDS));
        } else {
            hipLaunchKernelGGL(frw::groth16_scale_quad_kernel, dim3(2), dim3(4), 0, st, (size_t)1, (const uint32_t *)d_split, (const uint32_t *)pA, pSA);
            hipLaunchKernelGGL(frw::groth16_finish_kernel, dim3(1), dim3(64), 0, st, (size_t)1, (const uint32_t *)pA, (const uint32_t *)pSA,
                               (const uint32_t *)pRB1, (const uint32_t *)(pA + 2 * XW), (const uint32_t *)pH, (const uint32_t *)pB2,
                               (uint32_t *)(d_proofs + b * 48));
        }
        e = hipGetLastError();
        // the next statement reuses the workspace: everything of this one must be through (the streams are ordered by `st` from here on)
    }
    if (e != hipSuccess || rc != FRW_OK) {
        if (forked) (void)hipDeviceSynchronize();
        return e != hipSuccess ? frw::record_hip_error(e, "frw_groth16_prove_dev") : rc;
    }
    return FRW_OK;
}
// rs_host: the blinding factors in host memory (uploaded, and waited for: the array may be short-lived); rs_dev: in device memory
int groth16_prove(const frw_groth16_pk *pk, const frw_r1cs *r1cs, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                  const uint64_t *rs_host, const uint64_t *rs_dev, uint64_t *d_proofs, uint32_t *d_num_unsatisfied, void *d_workspace,
                  size_t workspace_bytes, void *stream)
{
    const uint64_t *rs = rs_host ? rs_host : rs_dev;
    if (!pk || !r1cs || (batch && (!d_witness || !d_instance || !rs || !d_proofs || !d_workspace))) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    frw_qap_info_t q;
    if (frw_qap_info(r1cs, &q) != FRW_OK || q.domain_size != pk->domain_size || q.num_instance != pk->num_instance) return FRW_E_INVALID_ARG;
    if (pk->bare) {
        if (pk->world > 1) return FRW_E_INVALID_ARG;                        // a slice makes partial sums: frw_groth16_prove_partial_dev
        return groth16_prove_bare(pk, r1cs, batch, d_witness, d_instance, rs_host, rs_dev, d_proofs, nullptr, d_num_unsatisfied, d_workspace, workspace_bytes, stream);
    }
    const Groth16Sizes sz = groth16_sizes(pk, r1cs);
    const size_t chunk = std::min<size_t>(workspace_bytes / sz.per, 4096);
    if (chunk == 0 || ((uintptr_t)d_workspace & 255)) return FRW_E_INVALID_ARG;
    const size_t I = (size_t)pk->num_instance, W = (size_t)pk->num_witness, nv = I + W, n = (size_t)pk->domain_size, stride = nv + 3;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSetDevice(pk->device);
    int rc = FRW_OK;
    bool forked = false;
    // The key's side streams and events take one chunk's work at a time (pk->enqueue); the host-side wait for the upload of r, s
    // happens BEFORE the lock is taken, so provers that share a key wait for their own stream only, not for each other's.
    std::unique_lock<std::mutex> lock(const_cast<frw_groth16_pk *>(pk)->enqueue, std::defer_lock);
    for (size_t lo = 0; e == hipSuccess && rc == FRW_OK && lo < batch; lo += chunk) {
        if (lock.owns_lock()) lock.unlock();
        const size_t cnt = std::min(chunk, batch - lo);
        char *base = (char *)d_workspace;
        void *qap_ws = base;                          base += cnt * sz.qap;
        uint64_t *h = (uint64_t *)base;               base += cnt * sz.h;
        uint64_t *zext = (uint64_t *)base;            base += cnt * sz.zext;
        char *msm_ws[5];
        for (int i = 0; i < 5; i++) { msm_ws[i] = base; base += cnt * sz.msm[i]; }
        constexpr size_t XW = frw::Grp<FqField>::BK_WORDS;               // the G1 points stay in XYZZ coordinates until the proof is put together
        uint32_t *pA = (uint32_t *)base, *pB1 = pA + cnt * XW, *pL = pB1 + cnt * XW, *pH = pL + cnt * XW, *pSA = pH + cnt * XW, *pRB1 = pSA + cnt * XW;
        (void)pB1;                                                       // [A | B1' | L] is what the three G1 sums write, [s A | r B1'] the scalar multiplications
        uint64_t *pB2 = (uint64_t *)(pRB1 + cnt * XW), *d_rs = pB2 + cnt * 24, *d_split = d_rs + cnt * 8;
        const uint64_t *wit = d_witness + lo * W * 4, *inst = d_instance + lo * I * 4;
        // the blinding factors: a host array (the prover draws them) is uploaded before anything reads them -- and waited for, the
        // array may be pageable and short-lived; factors that are in device memory already cost no wait (and the call is capturable)
        const uint64_t *rs_now = rs_dev ? rs_dev + lo * 8 : d_rs;
        if (rs_host) {
            e = hipMemcpyAsync(d_rs, rs_host + lo * 8, cnt * 64, hipMemcpyHostToDevice, st);
            if (e == hipSuccess) e = hipStreamSynchronize(st);
            if (e != hipSuccess) break;
        }
        lock.lock();
        hipLaunchKernelGGL(frw::groth16_split_kernel, dim3((unsigned)((2 * cnt + 63) / 64)), dim3(64), 0, st, 2 * cnt, rs_now, d_split);
        // z ++ [1, r, s] per signature, then the witness-side sums on their own streams ...
        e = hipMemcpy2DAsync(zext, stride * 32, inst, I * 32, I * 32, cnt, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess) e = hipMemcpy2DAsync(zext + I * 4, stride * 32, wit, W * 32, W * 32, cnt, hipMemcpyDeviceToDevice, st);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(frw::groth16_tails_kernel, dim3((unsigned)((cnt + 63) / 64)), dim3(64), 0, st, cnt, (const uint32_t *)rs_now,
                           (uint32_t *)zext, stride * 8, nv);
        // Four chains of small kernels, and ONE host thread that enqueues them at 15 - 35 us a launch (some 45 launches: a chain
        // enqueued last starts a millisecond late; profiles/r04_groth16_batch1_latency.txt) -- so after the short sort the head of
        // the longest chain goes first, and the sum over h_query -- which cannot start before the witness map is through, and whose
        // bucket kernel then fills the chip -- last (for a lone proof, enqueued right behind the map instead: the same 3.92 - 3.97 ms).
        //   main      the witness map (a dozen launches, 0.7 ms; needs nothing of the sort) and the sum over h_query
        //   caller's  the sort of the digits of z ++ [1, r, s]: ONE counting sort for all four witness-side sums (its arrays live in
        //             b_g2_query's workspace; the workspaces of a_query, b_g1_query and l_query, one after the other, are the three G1
        //             sums' as one)
        //   side[0]   the three G1 sums (A, B1' and L land one after the other) as ONE chain of kernels, then both scalar
        //             multiplications as soon as their points exist
        //   side[1]   for a few proofs at a time the G1 sums' ones, beside their work items; nothing otherwise
        //   side[2]   G2
        // The key's four streams were created one after the other and sit on four different hardware queues (HIP's default); the
        // caller's stream shares a queue with ONE of them, which one depends on how many streams its process made before.  So nothing
        // but the short sort (first) and the assembly (last) runs on the caller's stream: the ones' sums, put there at first, waited
        // behind the whole witness-map chain in a process whose stream shared `main`'s queue (the sixteen-statement aggregate under the
        // profiler: 26.6 ms with them there, 22.8 ms with them on side[1]).
        const bool lone = cnt <= 4;                                        // latency counts, not throughput
        e = hipEventRecord(pk->fork, st);
        forked = true;
        if (e == hipSuccess) e = hipStreamWaitEvent(pk->main, pk->fork, 0);
        if (e != hipSuccess) break;
        // the sort first: its kernels are workgroups of 1,024 threads, which do not start under the sum over h_query's bucket kernel
        // (0.8 ms late in a trace where the host was slow to get to them)
        const NmsmBufs sorted = nmsm_carve<Fq2Field>(msm_ws[4], cnt, (uint32_t)stride);
        const bool ones_as_mask = pk->a->ones_table && pk->b1->ones_table && pk->l->ones_table && pk->b2->ones_table;
        e = nmsm_sort(sorted, (uint32_t)stride, cnt, (const uint32_t *)zext, stride * 8, 1, ones_as_mask, st);
        if (e != hipSuccess) break;
        e = hipEventRecord(pk->sorted, st);
        for (int i = 0; i < 3 && e == hipSuccess; i++)
            if (i != 1 || lone) e = hipStreamWaitEvent(pk->side[i], pk->sorted, 0);
        if (e != hipSuccess) break;
        rc = frw_qap_witness_map_dev(r1cs, cnt, wit, inst, h, d_num_unsatisfied ? d_num_unsatisfied + lo : nullptr, qap_ws, cnt * sz.qap, pk->main);
        if (rc != FRW_OK) break;
        const frw_msm *g1s[3] = {pk->a, pk->b1, pk->l};
        e = nmsm_accumulate<FqField, true>(g1s, 3, sorted, nmsm_carve<FqField>(msm_ws[1], 3 * cnt, (uint32_t)stride), cnt, pA, ones_as_mask, pk->side[0], true,
                                           lone, pk->side[1], pk->ones_done);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(frw::groth16_scale_quad_kernel, dim3((unsigned)(2 * cnt)), dim3(4), 0, pk->side[0], cnt, (const uint32_t *)d_split,
                           (const uint32_t *)pA, pSA);
        e = nmsm_accumulate<Fq2Field, false>(&pk->b2, 1, sorted, sorted, cnt, (uint32_t *)pB2, ones_as_mask, pk->side[2]);
        if (e != hipSuccess) break;
        // (h_query has n - 1 points and the dense pipeline: frw_groth16_pk_load; the scalars are h's coefficients 0 .. n - 2)
        rc = msm_run<FqField, true>(pk->h, cnt, h, n, 1, (uint64_t *)pH, msm_ws[0], cnt * sz.msm[0], pk->main, true);
        if (rc != FRW_OK) break;
        // (side[1] is neither recorded nor waited for: what it carries, if anything, side[0] has waited for -- and inside a stream
        // capture an event of a stream that is not part of the capture could not be waited for)
        for (int i = 0; i < 3 && e == hipSuccess; i += 2) e = hipEventRecord(pk->join[i], pk->side[i]);
        if (e == hipSuccess) e = hipEventRecord(pk->join[3], pk->main);
        if (e != hipSuccess) break;
        for (int i = 0; i < 4 && e == hipSuccess; i++)
            if (i != 1) e = hipStreamWaitEvent(st, pk->join[i], 0);
        if (e != hipSuccess) break;
        hipLaunchKernelGGL(frw::groth16_finish_kernel, dim3((unsigned)((2 * cnt + 63) / 64)), dim3(64), 0, st, cnt, (const uint32_t *)pA,
                           (const uint32_t *)pSA, (const uint32_t *)pRB1, (const uint32_t *)pL, (const uint32_t *)pH, (const uint32_t *)pB2,
                           (uint32_t *)(d_proofs + lo * 48));
        e = hipGetLastError();
    }
    if (e != hipSuccess || rc != FRW_OK) {
        // a call that gave up half way may have work on the side streams that nothing joins: wait for it here, so that the
        // caller can release the workspace when the error comes back
        if (forked) (void)hipDeviceSynchronize();
        return e != hipSuccess ? frw::record_hip_error(e, "frw_groth16_prove_dev") : rc;
    }
    return FRW_OK;
}
}  // namespace

extern "C" int frw_groth16_prove_dev(const frw_groth16_pk *pk, const frw_r1cs *r1cs, size_t batch, const uint64_t *d_witness,
                                     const uint64_t *d_instance, const uint64_t *rs, uint64_t *d_proofs, uint32_t *d_num_unsatisfied,
                                     void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!rs && batch) return FRW_E_INVALID_ARG;
    return groth16_prove(pk, r1cs, batch, d_witness, d_instance, rs, nullptr, d_proofs, d_num_unsatisfied, d_workspace, workspace_bytes, stream);
}
extern "C" int frw_groth16_prove_rs_dev(const frw_groth16_pk *pk, const frw_r1cs *r1cs, size_t batch, const uint64_t *d_witness,
                                        const uint64_t *d_instance, const uint64_t *d_rs, uint64_t *d_proofs, uint32_t *d_num_unsatisfied,
                                        void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!d_rs && batch) return FRW_E_INVALID_ARG;
    return groth16_prove(pk, r1cs, batch, d_witness, d_instance, nullptr, d_rs, d_proofs, d_num_unsatisfied, d_workspace, workspace_bytes, stream);
}

// ---- a key in slices: the partial sums of one rank, and their combination ---------------------------------------------------------------
extern "C" int frw_groth16_prove_partial_dev(const frw_groth16_pk *pk, const frw_r1cs *r1cs, size_t batch, const uint64_t *d_witness,
                                             const uint64_t *d_instance, const uint64_t *rs, uint64_t *d_partial, uint32_t *d_num_unsatisfied,
                                             void *d_workspace, size_t workspace_bytes, void *stream)
{
    if (!pk || !pk->bare || !r1cs || (batch && (!d_witness || !d_instance || !rs || !d_partial || !d_workspace))) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    frw_qap_info_t q;
    if (frw_qap_info(r1cs, &q) != FRW_OK || q.domain_size != pk->domain_size || q.num_instance != pk->num_instance) return FRW_E_INVALID_ARG;
    return groth16_prove_bare(pk, r1cs, batch, d_witness, d_instance, rs, nullptr, nullptr, d_partial, d_num_unsatisfied, d_workspace, workspace_bytes, stream);
}

extern "C" int frw_groth16_prove_combine_dev(const frw_groth16_pk *pk, size_t world, const uint64_t *d_partials, const uint64_t *rs, uint64_t *d_proof,
                                             void *d_workspace, size_t workspace_bytes, void *stream)
{
    constexpr size_t XW = frw::Grp<FqField>::BK_WORDS;
    if (!pk || world == 0 || world > 65536 || !d_partials || !rs || !d_proof || !d_workspace || workspace_bytes < FRW_GROTH16_COMBINE_WORKSPACE ||
        ((uintptr_t)d_workspace & 255))
        return FRW_E_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSetDevice(pk->device);
    uint32_t *pA = (uint32_t *)d_workspace, *pH = pA + 3 * XW, *pSA = pH + XW, *pRB1 = pSA + XW;
    uint64_t *pB2 = (uint64_t *)(pRB1 + XW), *d_rs = pB2 + 24, *d_split = d_rs + 8;
    static_assert(6 * XW * 4 + 192 + 64 + 64 <= FRW_GROTH16_COMBINE_WORKSPACE, "the combine step's points fit its workspace");
    if (e == hipSuccess) e = hipMemcpyAsync(d_rs, rs, 64, hipMemcpyHostToDevice, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);                     // (the array may be short-lived)
    if (e != hipSuccess) return frw::record_hip_error(e, "frw_groth16_prove_combine_dev");
    hipLaunchKernelGGL(frw::groth16_split_kernel, dim3(1), dim3(64), 0, st, (size_t)2, (const uint64_t *)d_rs, d_split);
    hipLaunchKernelGGL(frw::groth16_combine_kernel, dim3(1), dim3(64), 0, st, (uint32_t)world, (const uint32_t *)d_partials, pA, (uint32_t *)pB2);
    hipLaunchKernelGGL(frw::groth16_scale_quad_kernel, dim3(2), dim3(4), 0, st, (size_t)1, (const uint32_t *)d_split, (const uint32_t *)pA, pSA);
    hipLaunchKernelGGL(frw::groth16_finish_kernel, dim3(1), dim3(64), 0, st, (size_t)1, (const uint32_t *)pA, (const uint32_t *)pSA, (const uint32_t *)pRB1,
                       (const uint32_t *)(pA + 2 * XW), (const uint32_t *)pH, (const uint32_t *)pB2, (uint32_t *)d_proof);
    e = hipGetLastError();
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_groth16_prove_combine_dev");
}
#endif   // FRW_MSM_PROBE
