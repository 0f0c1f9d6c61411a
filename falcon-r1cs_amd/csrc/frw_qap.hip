// frw_qap.hip -- the R1CS -> QAP witness map on the device: h(X) = (A(X) B(X) - C(X)) / (X^n - 1) over BLS12-381 Fr.
//
// What a Groth16 prover does with the witness right after the hot path (examples/pok_sig.rs:30-47 calls
// Groth16::prove; ark-groth16 0.3.0 r1cs_to_qap.rs R1CStoQAP::witness_map):
//     a = A z (+ the instance values in rows C .. C+I), b = B z, c = C z       over the domain of n-th roots of unity
//     a, b, c <- coset_fft(ifft(.));   ab = a o b - c;   ab /= (g^n - 1);   h = coset_ifft(ab)
// Here, with the transforms arranged so that no permutation pass and no scattered access is needed:
//     r1cs_eval_kernel writes A z, B z, C z in constraint order (ark-ff's 8 x 32-bit Montgomery form)
//     inverse transform, decimation in time; its first pass reads that order through the bit-reversed tile
//         (64 rows 2^(L-6) apart x 16 adjacent rows), appends the instance rows and the zero padding on the fly;
//         its last pass multiplies by g^k / n on the way out
//     forward transform, decimation in frequency (natural in, bit-reversed out)      -> values on the coset
//     inverse transform, decimation in time, first pass fused with a b - c, last pass with g^-k / (n (g^n - 1)) and the
//         canonical reduction                                                        -> h, ark-ff form, natural order
//
// Every transform is three passes over HBM (6 + 6 + (L - 12) butterfly stages) in four-step form: a pass is a set of
// plain 2^T-point transforms that need the 64-th roots of unity only, followed by ONE multiplication per element by a
// per-index factor read alongside the data (the twist towards the next pass, or the scale factor of the witness map).
// A workgroup of ONE wavefront takes a tile of 2^T rows x 2^(9-T) adjacent elements; a thread holds 8 elements in
// registers and does three stages on them (radix 8), the tile is exchanged once through LDS -- in two halves, so that
// a wavefront needs 9 KB of it and registers, not LDS, decide how many wavefronts a SIMD holds (three; with the
// 128-thread workgroups and 37 KB tiles of round 2 it was two, and they met at four barriers per pass) --, three (or
// two) more stages, the multiplication, and out.  A/B timing of the first version (one LDS round trip and one gathered twiddle per stage)
// showed the passes waiting for LDS and for the gathers, not for the multiplier (profiles/r02_qap_v2_ab.txt).
// tools/dev/qap_fourstep_model.py and qap_radix8_model.py are exact-integer models of the index and twiddle arithmetic.
//
// Arithmetic: nine 29-bit limbs, lazily reduced (frw_fr29.h); between passes elements are < 2 p and travel packed in
// 8 x 32 bits, the same form the sparse products arrive in and h leaves in.  Field arithmetic is exact, so the result is
// the same element for element as any other schedule's.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include "frw_device.h"
#include "frw_fr29.h"

namespace frw {

constexpr int QAP_TILE = 512;           // elements per tile
constexpr int QAP_THREADS = 64;         // one wavefront, 8 elements per thread
constexpr int QAP_HALF = QAP_TILE / 2;  // elements in LDS at a time (one 32-bit word per element and limb plane)

enum { PASS_FIRST = 0, PASS_DIT_SH0 = 1, PASS_DIT = 2, PASS_DIF = 3, PASS_DIF_SH0 = 4, PASS_DIT_DIF = 5 };
enum { LOAD_PLAIN = 0, LOAD_AB_MINUS_C = 1, LOAD_AB = 2, LOAD_PRODUCTS_AB = 3 };
enum { STORE_PLAIN = 0, STORE_FACTOR = 1, STORE_FACTOR_A = 2, STORE_FACTOR_CANONICAL = 3, STORE_CONST_ADD_CANONICAL = 4 };

struct NttPass {
    const uint32_t *src;        // [arrays][n][8] u32, working order (PASS_FIRST: unused)
    uint32_t *dst;              // [arrays][n][8] u32 (may be src: every tile reads all it needs before it writes)
    size_t src_stride, dst_stride;    // words between consecutive arrays
    const uint32_t *factor;     // STORE_FACTOR*: [n][8], the multiplier of each working index as x R' packed
    const uint32_t *factor_a;   // STORE_FACTOR_A: the table for arrays 0, 3, 6, ... instead
    const uint32_t *roots;      // 64-th roots of unity ^k, k < 32, x R' in nine limbs, 12 words apart
    const uint32_t *roots2, *factor2;   // PASS_DIT_DIF: roots and per-index factor of the decimation-in-frequency half
    // PASS_FIRST sources
    const uint32_t *abc;        // [signatures][3][C][8] u32
    const uint32_t *instance;   // [signatures][I][8] u32
    uint32_t num_constraints, num_instance;
    int L, sh;                  // log n; lowest index bit this pass transforms
    int per_sig;                // PASS_FIRST: arrays per signature in this launch (array y = signature y / per_sig, matrix y % per_sig)
    uint32_t cfac[9];           // STORE_CONST_ADD_CANONICAL: the constant factor, x R' in nine limbs
    // list mode (the seven-transform map for the few signatures whose witness violates the system): grid.y = arrays per
    // signature, and a workgroup does its tile of every listed signature in turn; array index = list[k] x gridDim.y + blockIdx.y
    const uint32_t *list, *list_count;
};

// A/B builds (tools/ab_qap.py): -DFRW_QAP_NO_MUL replaces the products by additions, -DFRW_QAP_NO_STAGES skips the
// butterflies (passes become copies + the element-wise factor).  Results are then wrong; timings tell what a pass waits for.
#if defined(FRW_QAP_NO_MUL)
#define QAP_MUL(a, b) f29_add(a, b)
#else
#define QAP_MUL(a, b) f29_mul(a, b)
#endif

// -DFRW_QAP_NO_LAUNDER (tools/ab_qap.py): the indices of stores and LDS slots are NOT derived afresh where they are used (round 4's
// code: the compiler keeps them from where the elements were loaded, in scratch memory) -- the A/B of round 5's removal of that scratch
#if defined(FRW_QAP_NO_LAUNDER)
#define QAP_LAUNDER(v) ((void)0)
#else
#define QAP_LAUNDER(v) asm volatile("" : "+v"(v))
#endif

// compile-time loop: `#pragma unroll` is a request, and a loop over a thread's eight elements that stays a loop puts the
// register array into scratch memory (it did, for the loop that multiplies by the factor and stores)
template <int N, class F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (N > 0) {
        static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ F29 root_get(const uint32_t *roots, int idx)
{
    const uint4 a = *(const uint4 *)(roots + idx * 12), b = *(const uint4 *)(roots + idx * 12 + 4);
    F29 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    r.l[8] = roots[idx * 12 + 8];
    return r;
}

// the butterflies' additions and subtractions; -DFRW_QAP_AB_NO_NORMALISE (tools/ab_qap.py) drops their carry propagation --
// results are then wrong; the timing is the ceiling of what any lazier limb representation could gain
#if defined(FRW_QAP_AB_NO_NORMALISE)
__device__ __forceinline__ F29 bf_add(const F29 &a, const F29 &b)
{
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] + b.l[i];
    return r;
}
template <uint32_t K> __device__ __forceinline__ F29 bf_sub(const F29 &a, const F29 &b)
{
    F29 r;
#pragma unroll
    for (int i = 0; i < NL29; i++) r.l[i] = a.l[i] - b.l[i] + kp29_table<K>().l[i];
    return r;
}
#else
__device__ __forceinline__ F29 bf_add(const F29 &a, const F29 &b) { return f29_add(a, b); }
template <uint32_t K> __device__ __forceinline__ F29 bf_sub(const F29 &a, const F29 &b) { return f29_sub_kp<K>(a, b); }
#endif

// decimation in time: (u, v) -> (u + w v, u - w v).  KV = bound of v in units of p when there is no product (w == 1).
template <uint32_t KV>
__device__ __forceinline__ void dit_one(F29 &u, F29 &v)
{
    const F29 s = bf_add(u, v);
    v = bf_sub<KV>(u, v);
    u = s;
}
__device__ __forceinline__ void dit_mul(F29 &u, F29 &v, const F29 &w)
{
    v = QAP_MUL(v, w);                    // < 2 p
    const F29 s = bf_add(u, v);
    v = bf_sub<2>(u, v);
    u = s;
}
// decimation in frequency: (u, v) -> (u + v, (u - v) w), everything < 2 p
__device__ __forceinline__ void dif_one(F29 &u, F29 &v)
{
    const F29 s = f29_reduce_4p(bf_add(u, v));
    v = f29_reduce_4p(bf_sub<2>(u, v));
    u = s;
}
__device__ __forceinline__ void dif_mul(F29 &u, F29 &v, const F29 &w)
{
    const F29 s = f29_reduce_4p(bf_add(u, v));
    v = QAP_MUL(bf_sub<2>(u, v), w);
    u = s;
    __builtin_amdgcn_sched_barrier(0);       // one butterfly's temporaries at a time (see round_high)
}

// The same butterflies without the conditional subtractions (PASS_DIF with a factor at the end): sums and differences
// are left to grow -- inputs < B p give u + v < 2 B p and u - v + B p < 2 B p -- and only the products bring the values
// back under 2 p: three stages take < 2 p to < 16 p, one conditional subtraction of 8 p between the rounds, three more
// stages to < 64 p (2^261 = 70.4 p), and the per-index factor at the end of the pass (a product) returns < 2 p.  A
// decimation-in-frequency pass spent a third of its vector instructions on those subtractions (two per butterfly).
template <uint32_t B>
__device__ __forceinline__ void dif_one_lazy(F29 &u, F29 &v)
{
    const F29 s = bf_add(u, v);
    v = bf_sub<B>(u, v);
    u = s;
}
template <uint32_t B>
__device__ __forceinline__ void dif_mul_lazy(F29 &u, F29 &v, const F29 &w)
{
    const F29 s = bf_add(u, v);
    v = QAP_MUL(bf_sub<B>(u, v), w);
    u = s;
    __builtin_amdgcn_sched_barrier(0);
}

// Stages 1..3 of a 2^T-point transform on the 8 elements x[e] = row 8 g + e: the twiddle of stage t for the pair whose
// lower row is r is (2^t-th root)^(r mod 2^(t-1)), i.e. a 64-th root with exponent (e mod 2^(t-1)) 2^(6-t) -- the same
// three constants (w8, w16, w24) for every thread.
template <bool DIF, bool LAZY = false>
__device__ __forceinline__ void round_low(F29 (&x)[8], const F29 &w8, const F29 &w16, const F29 &w24)
{
    if (DIF && LAZY) {                                     // inputs < 8 p
        dif_one_lazy<8>(x[0], x[4]); dif_mul_lazy<8>(x[1], x[5], w8); dif_mul_lazy<8>(x[2], x[6], w16); dif_mul_lazy<8>(x[3], x[7], w24);
        dif_one_lazy<16>(x[0], x[2]); dif_mul_lazy<16>(x[1], x[3], w16); dif_one_lazy<16>(x[4], x[6]); dif_mul_lazy<16>(x[5], x[7], w16);
        dif_one_lazy<32>(x[0], x[1]); dif_one_lazy<32>(x[2], x[3]); dif_one_lazy<32>(x[4], x[5]); dif_one_lazy<32>(x[6], x[7]);   // -> < 64 p
    } else if (!DIF) {
        dit_one<2>(x[0], x[1]); dit_one<2>(x[2], x[3]); dit_one<2>(x[4], x[5]); dit_one<2>(x[6], x[7]);      // -> < 4 p
        dit_one<4>(x[0], x[2]); dit_mul(x[1], x[3], w16); dit_one<4>(x[4], x[6]); dit_mul(x[5], x[7], w16);  // -> < 8 p
        dit_one<8>(x[0], x[4]); dit_mul(x[1], x[5], w8); dit_mul(x[2], x[6], w16); dit_mul(x[3], x[7], w24); // -> < 16 p
    } else {
        dif_one(x[0], x[4]); dif_mul(x[1], x[5], w8); dif_mul(x[2], x[6], w16); dif_mul(x[3], x[7], w24);
        dif_one(x[0], x[2]); dif_mul(x[1], x[3], w16); dif_one(x[4], x[6]); dif_mul(x[5], x[7], w16);
        dif_one(x[0], x[1]); dif_one(x[2], x[3]); dif_one(x[4], x[5]); dif_one(x[6], x[7]);
    }
}

// Stages 4..T on the 8 elements x[e]: T == 6: row 8 e + g (three stages over e); T == 5: row 8 (e & 3) + g + 4 (e >> 2)
// (two stages over the low two bits of e, two independent groups).  Twiddles depend on the thread (g): table look-ups.
template <bool DIF, int T, bool LAZY = false>
__device__ __forceinline__ void round_high(F29 (&x)[8], const uint32_t *roots, int g)
{
    if (T == 4) {
        // one stage: rows (e & 1) 8 + g + 2 (e >> 1), g < 2 -- four pairs (x[2 m], x[2 m + 1]) whose lower row is g + 2 m, twiddle
        // (16-th root)^(g + 2 m); values < 2 p in, < 4 p out (decimation in frequency, lazy) / < 18 p (decimation in time)
        static_for<4>([&](auto mc) {
            constexpr int m = decltype(mc)::value;
            const F29 w = root_get(roots, (g + 2 * m) << 2);
            if (!DIF) dit_mul(x[2 * m], x[2 * m + 1], w);
            else if (LAZY) dif_mul_lazy<2>(x[2 * m], x[2 * m + 1], w);
            else dif_mul(x[2 * m], x[2 * m + 1], w);
            __builtin_amdgcn_sched_barrier(0);
        });
    } else if (DIF && LAZY) {                              // inputs < 2 p; T == 6: -> < 16 p, T == 5: -> < 8 p
        if (T == 6) {
            dif_mul_lazy<2>(x[0], x[4], root_get(roots, g)); dif_mul_lazy<2>(x[1], x[5], root_get(roots, g + 8));
            dif_mul_lazy<2>(x[2], x[6], root_get(roots, g + 16)); dif_mul_lazy<2>(x[3], x[7], root_get(roots, g + 24));
            { const F29 w0 = root_get(roots, g << 1), w1 = root_get(roots, (g + 8) << 1);
              dif_mul_lazy<4>(x[0], x[2], w0); dif_mul_lazy<4>(x[1], x[3], w1); dif_mul_lazy<4>(x[4], x[6], w0); dif_mul_lazy<4>(x[5], x[7], w1); }
            { const F29 w = root_get(roots, g << 2);
              dif_mul_lazy<8>(x[0], x[1], w); dif_mul_lazy<8>(x[2], x[3], w); dif_mul_lazy<8>(x[4], x[5], w); dif_mul_lazy<8>(x[6], x[7], w); }
        } else {
            // (each stage's twiddles fetched in its stage: six of them held at once were 54 registers)
            { const F29 wa0 = root_get(roots, g << 1), wa1 = root_get(roots, (g + 8) << 1);
              dif_mul_lazy<2>(x[0], x[2], wa0); dif_mul_lazy<2>(x[1], x[3], wa1); }
            { const F29 wb0 = root_get(roots, (g + 4) << 1), wb1 = root_get(roots, (g + 12) << 1);
              dif_mul_lazy<2>(x[4], x[6], wb0); dif_mul_lazy<2>(x[5], x[7], wb1); }
            __builtin_amdgcn_sched_barrier(0);
            { const F29 wa = root_get(roots, g << 2);
              dif_mul_lazy<4>(x[0], x[1], wa); dif_mul_lazy<4>(x[2], x[3], wa); }
            { const F29 wb = root_get(roots, (g + 4) << 2);
              dif_mul_lazy<4>(x[4], x[5], wb); dif_mul_lazy<4>(x[6], x[7], wb); }
        }
    } else if (T == 6) {
        if (!DIF) {
            // (the fences: as in the other direction below -- every stage's twiddles fetched in its stage, not all seven up front)
            { const F29 w = root_get(roots, g << 2);                                         // stage 4
              dit_mul(x[0], x[1], w); dit_mul(x[2], x[3], w); dit_mul(x[4], x[5], w); dit_mul(x[6], x[7], w); }
            __builtin_amdgcn_sched_barrier(0);
            { const F29 w0 = root_get(roots, g << 1), w1 = root_get(roots, (g + 8) << 1);    // stage 5
              dit_mul(x[0], x[2], w0); dit_mul(x[1], x[3], w1); dit_mul(x[4], x[6], w0); dit_mul(x[5], x[7], w1); }
            __builtin_amdgcn_sched_barrier(0);
            dit_mul(x[0], x[4], root_get(roots, g)); dit_mul(x[1], x[5], root_get(roots, g + 8));                  // stage 6
            __builtin_amdgcn_sched_barrier(0);
            dit_mul(x[2], x[6], root_get(roots, g + 16)); dit_mul(x[3], x[7], root_get(roots, g + 24));
        } else {
            // the scheduler would fetch all seven twiddles up front (63 registers on top of the 72 of x and the 45 of a
            // product in flight: spills at three wavefronts per SIMD); the fences keep every stage's fetches in its stage
            dif_mul(x[0], x[4], root_get(roots, g)); dif_mul(x[1], x[5], root_get(roots, g + 8));
            __builtin_amdgcn_sched_barrier(0);
            dif_mul(x[2], x[6], root_get(roots, g + 16)); dif_mul(x[3], x[7], root_get(roots, g + 24));
            __builtin_amdgcn_sched_barrier(0);
            { const F29 w0 = root_get(roots, g << 1), w1 = root_get(roots, (g + 8) << 1);
              dif_mul(x[0], x[2], w0); dif_mul(x[1], x[3], w1); dif_mul(x[4], x[6], w0); dif_mul(x[5], x[7], w1); }
            __builtin_amdgcn_sched_barrier(0);
            { const F29 w = root_get(roots, g << 2);
              dif_mul(x[0], x[1], w); dif_mul(x[2], x[3], w); dif_mul(x[4], x[5], w); dif_mul(x[6], x[7], w); }
        }
    } else {
        // stage 4: the two groups' twiddles wa, wb; stage 5: wa0, wa1, wb0, wb1 -- each fetched in its stage
        if (!DIF) {
            { const F29 wa = root_get(roots, g << 2);
              dit_mul(x[0], x[1], wa); dit_mul(x[2], x[3], wa); }
            { const F29 wb = root_get(roots, (g + 4) << 2);
              dit_mul(x[4], x[5], wb); dit_mul(x[6], x[7], wb); }
            __builtin_amdgcn_sched_barrier(0);
            { const F29 wa0 = root_get(roots, g << 1), wa1 = root_get(roots, (g + 8) << 1);
              dit_mul(x[0], x[2], wa0); dit_mul(x[1], x[3], wa1); }
            { const F29 wb0 = root_get(roots, (g + 4) << 1), wb1 = root_get(roots, (g + 12) << 1);
              dit_mul(x[4], x[6], wb0); dit_mul(x[5], x[7], wb1); }
        } else {
            { const F29 wa0 = root_get(roots, g << 1), wa1 = root_get(roots, (g + 8) << 1);
              dif_mul(x[0], x[2], wa0); dif_mul(x[1], x[3], wa1); }
            { const F29 wb0 = root_get(roots, (g + 4) << 1), wb1 = root_get(roots, (g + 12) << 1);
              dif_mul(x[4], x[6], wb0); dif_mul(x[5], x[7], wb1); }
            __builtin_amdgcn_sched_barrier(0);
            { const F29 wa = root_get(roots, g << 2);
              dif_mul(x[0], x[1], wa); dif_mul(x[2], x[3], wa); }
            { const F29 wb = root_get(roots, (g + 4) << 2);
              dif_mul(x[4], x[5], wb); dif_mul(x[6], x[7], wb); }
        }
    }
}

// One pass = T consecutive radix-2 stages on the index bits [sh, sh + T) of the working order, then the per-index factor.
// Tile geometry: PASS_DIT / PASS_DIF (sh > 0): rows = index bits [sh, sh + T), columns = index bits [0, 9 - T): runs of
// 256 / 512 bytes in memory, threads read and write their elements directly.  PASS_DIT_SH0 / PASS_DIF_SH0 (sh == 0,
// T == 6): 512 consecutive elements, row = low six bits: they go through LDS on both sides (memory order <-> the rows a
// thread holds), since a thread's eight rows would be adjacent in memory on one of them.  PASS_FIRST: the sh == 0 pass of
// a decimation-in-time transform whose input is still in constraint order in the products' buffers: working index =
// bitrev(natural index), so the 64 rows are the natural-index bits [L - 6, L) reversed and the 8 columns the
// natural-index bits [0, 3) (256-byte runs on the way in); the output is a memory-order tile of the working array.
template <int MODE, int T, int LOAD, int STORE, bool LIST = false>
__global__ __launch_bounds__(QAP_THREADS, 3) void ntt_pass_kernel(const NttPass p)
{
    constexpr bool DIF = MODE == PASS_DIF || MODE == PASS_DIF_SH0;
    constexpr bool FUSED = MODE == PASS_DIT_DIF;                                 // in as PASS_DIT, out as PASS_DIF
    constexpr bool MEMORDER = MODE == PASS_DIT_SH0 || MODE == PASS_DIF_SH0;     // tile = 512 consecutive elements
    constexpr int CB = 9 - T, COLS = 1 << CB;
    static_assert(T == 6 || ((T == 5 || T == 4) && (MODE == PASS_DIT || MODE == PASS_DIF || MODE == PASS_DIT_DIF)), "four- and five-stage passes only above bit 0");
    static_assert(!FUSED || (LOAD == LOAD_PLAIN && (STORE == STORE_FACTOR || STORE == STORE_FACTOR_A)), "the fused pass: plain in, factor between and after");
    __shared__ uint32_t lds[NL29 * QAP_HALF];
    const int tid = threadIdx.x, c = tid & (COLS - 1), g = tid >> CB;           // T == 6: 8 columns x 8 g; T == 5: 16 x 4; T == 4: 32 x 2
    const uint32_t tileid = blockIdx.x;
    const size_t n = (size_t)1 << p.L;
    const uint32_t turns = LIST ? *p.list_count : 1u;                // LIST is a template parameter: the loop costs the usual kernels nothing
    for (uint32_t turn = 0; turn < turns; turn++) {                  // (one turn unless in list mode; the body is not indented for it)
    const uint32_t array_y = LIST ? p.list[turn] * gridDim.y + blockIdx.y : blockIdx.y;
    const uint32_t *src = p.src + (size_t)array_y * p.src_stride;
    uint32_t *dst = p.dst + (size_t)array_y * p.dst_stride;
    const uint32_t *factor = STORE == STORE_FACTOR_A && array_y % 3 == 0 ? p.factor_a : p.factor;

    // rows of the eight elements a thread holds in the low round (stages 1..3) and in the high round (stages 4..T)
    auto row_low = [&](int e) { return g * 8 + e; };
    auto row_high = [&](int e) { return T == 6 ? e * 8 + g : T == 5 ? (e & 3) * 8 + g + 4 * (e >> 2) : (e & 1) * 8 + g + 2 * (e >> 1); };
    // working index of (row, column)
    uint32_t lowmid = 0, high = 0;
    if (MODE == PASS_FIRST) high = (__brev(tileid) >> (32 - (p.L - 9))) << 6;
    else if (MEMORDER) high = tileid * (uint32_t)QAP_TILE;
    else {
        lowmid = (tileid & ((1u << (p.sh - CB)) - 1u)) << CB;
        high = (tileid >> (p.sh - CB)) << (p.sh + T);
    }
    auto widx = [&](int row, int col) -> uint32_t {
        if (MODE == PASS_FIRST) return (uint32_t)row | high | ((__brev((uint32_t)col) >> 29) << (p.L - 3));
        if (MEMORDER) return high + (uint32_t)(col * 64 + row);
        return high | ((uint32_t)row << p.sh) | lowmid | (uint32_t)col;
    };
    auto load_elem = [&](uint32_t gidx) -> F29 {
        F29 v = f29_unpack(fr_load(src + (size_t)gidx * 8));
        if (LOAD == LOAD_AB_MINUS_C) {
            // the three arrays of a signature lie n elements apart: (a 2^5 R)(b R) / R' - c R = (a b - c) R, < 2 p
            const F29 b = f29_unpack(fr_load(src + (n + gidx) * 8)), cc = f29_unpack(fr_load(src + (2 * n + gidx) * 8));
            v = f29_reduce_4p(f29_sub_kp<2>(QAP_MUL(v, b), cc));
        }
        if (LOAD == LOAD_AB) v = QAP_MUL(v, f29_unpack(fr_load(src + (n + gidx) * 8)));     // the two arrays of a signature: a b R / 32, < 2 p
        return v;
    };
    auto store_elem = [&](uint32_t gidx, F29 v) {
        if (STORE == STORE_CONST_ADD_CANONICAL) {
            // times a constant, plus what the destination already holds (< 2 p), canonical
            F29 cf;
#pragma unroll
            for (int k = 0; k < NL29; k++) cf.l[k] = p.cfac[k];
            v = f29_reduce_4p(f29_add(QAP_MUL(v, cf), f29_unpack(fr_load(dst + (size_t)gidx * 8))));
            v = f29_canonical(v);
        } else if (STORE != STORE_PLAIN) {
            v = QAP_MUL(v, f29_unpack(fr_load((FUSED ? p.factor2 : factor) + (size_t)gidx * 8)));      // < 2 p
        }
        if (STORE == STORE_FACTOR_CANONICAL) v = f29_canonical(v);
        fr_store(dst + (size_t)gidx * 8, f29_pack(v));
    };
    auto lds_put = [&](int s, const F29 &v) {
#pragma unroll
        for (int k = 0; k < NL29; k++) lds[k * QAP_HALF + s] = v.l[k];
    };
    auto lds_get = [&](int s) -> F29 {
        F29 v;
#pragma unroll
        for (int k = 0; k < NL29; k++) v.l[k] = lds[k * QAP_HALF + s];
        return v;
    };
    // The workgroup is one wavefront: LDS traffic is ordered by waiting for the wave's own LDS operations (the compiler
    // drops the s_barrier of a one-wave workgroup and keeps the wait).
    auto wave_sync = [&]() { __syncthreads(); };

    // ---- the exchange between the two rounds, half a tile at a time ------------------------------------------------------
    // An 8 x 8 transpose per column between the g of a thread and the e of its registers (T == 6), or 4 x 4 transposes
    // (T == 5); each half moves four registers of every thread.  The transpose is an involution, so low -> high
    // (decimation in time) and high -> low (decimation in frequency) are the same code.
    // Slot of (register j of the half, thread g, column c) and its bank (slot mod 32; a 32-lane group conflicts):
    //   T == 6: ((8 j + g) 8 + c) with bits 3, 4 XOR-ed by j: writers (g = 0..3 or 4..7, all c) differ in g & 3, readers
    //           (they read j' = g & 3, g' = e) differ in j'
    //   T == 5: ((4 g + j) 16 + c) with bit 4 XOR-ed by g: writers (two g, all c) differ in g & 1, readers (they read
    //           g' = e & 3, j' = g) in j' & 1
    //   T == 4: ((2 j + g) 32 + c): a 32-lane group is one g and all c
    auto exchange = [&](F29 (&x)[8]) {
        // (the slots from a laundered thread index: the fused pass exchanges twice, and slot addresses kept from the first exchange for
        // the second were four registers in scratch memory for the nine thousand instructions in between)
        uint32_t tid_x = (uint32_t)tid;
        QAP_LAUNDER(tid_x);
        const int c = (int)(tid_x & (COLS - 1)), g = (int)(tid_x >> CB);
        auto xslot = [&](int j, int gg) {
            return T == 6 ? (((8 * j + gg) * 8 + c) ^ (j << 3)) : T == 5 ? (((4 * gg + j) * 16 + c) ^ ((gg & 1) << 4)) : ((2 * j + gg) * 32 + c);
        };
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (T == 6) {
                // half h: the threads g < 4 send (and then receive into) their registers 4 h .. 4 h + 3, the threads g >= 4
                // their registers 4 (1 - h) .. : what thread g' needs from thread e' travels in half (g' >> 2) ^ (e' >> 2),
                // i.e. lands in the registers the receiver has just sent -- nothing is held aside
                if ((g >> 2) == h) {
                    static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; lds_put(xslot(j, g), x[j]); });
                } else {
                    static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; lds_put(xslot(j, g), x[4 + j]); });
                }
                wave_sync();
                if ((g >> 2) == h) {
                    static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; x[j] = lds_get(xslot(g & 3, j)); });
                } else {
                    static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; x[4 + j] = lds_get(xslot(g & 3, 4 + j)); });
                }
            } else if (T == 5) {
                static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; lds_put(xslot(j, g), x[4 * h + j]); });
                wave_sync();
                static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; x[4 * h + j] = lds_get(xslot(g, j)); });
            } else {
                // T == 4: row 8 g + e (e = 4 h + j) <-> row (e' & 1) 8 + g' + 2 (e' >> 1): register e' = 4 h + j' of thread g' comes from
                // thread j' & 1, its register 4 h + g' + 2 (j' >> 1) -- a 2 x 2 transpose between the two g of a column, both ways
                static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; lds_put(xslot(j, g), x[4 * h + j]); });
                wave_sync();
                static_for<4>([&](auto jc) { constexpr int j = decltype(jc)::value; x[4 * h + j] = lds_get(xslot(g + 2 * (j >> 1), j & 1)); });
            }
            wave_sync();
        }
    };
    // ---- memory order <-> register rows (sh == 0 passes, T == 6), four columns at a time --------------------------------
    // Slot of (row, column cc of the four): cc 64 + (row ^ swz(cc)).  A 32-lane group touches it as 32 consecutive rows
    // of one column (global order), as rows 8 g + e of four g x four columns (low round) or as rows 8 e + g (high round):
    // swz maps the two column bits to e2 and e0 + e3, independent of {e3, e4} as of {e0, e1}: no bank is hit twice.
    auto mslot = [&](int row, int cc) { return cc * 64 + (row ^ (((cc & 1) << 2) | ((cc >> 1) * 9))); };

    F29 x[8];

    // ---- in: the eight elements of the first round -----------------------------------------------------------------------
    if (MODE == PASS_FIRST) {
        const size_t sig = array_y / p.per_sig, which = array_y % p.per_sig;
        const uint32_t *rows = p.abc + (sig * 3 + which) * (size_t)p.num_constraints * 8;
        const uint32_t *inst = p.instance + sig * (size_t)p.num_instance * 8;
        static_for<8>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const uint32_t ihi = __brev((uint32_t)row_low(e)) >> 26;                 // row = rev6(natural bits [L - 6, L))
            const uint32_t i = (ihi << (p.L - 6)) | (tileid << 3) | (uint32_t)c;     // natural (constraint) index
            Fr8 w;
#pragma unroll
            for (int k = 0; k < 8; k++) w.l[k] = 0;
            if (LOAD == LOAD_PRODUCTS_AB) {
                // (A z)_i (B z)_i for the constraint rows, zero beyond (B z has no instance rows): a b R / 32, < 2 p
                if (i < p.num_constraints)
                    x[e] = QAP_MUL(f29_unpack(fr_load(rows + (size_t)i * 8)), f29_unpack(fr_load(rows + ((size_t)p.num_constraints + i) * 8)));
                else
                    x[e] = f29_unpack(w);
            } else {
                if (i < p.num_constraints) w = fr_load(rows + (size_t)i * 8);
                else if (which == 0 && i - p.num_constraints < p.num_instance) w = fr_load(inst + (size_t)(i - p.num_constraints) * 8);
                x[e] = f29_unpack(w);
            }
        });
    } else if (MEMORDER) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
#pragma unroll
            for (int k = 0; k < 4; k++) lds_put(mslot(tid, k), load_elem(high + (uint32_t)(tid + 64 * (4 * h + k))));
            wave_sync();
            if ((c >> 2) == h)
                static_for<8>([&](auto ec) { constexpr int e = decltype(ec)::value; x[e] = lds_get(mslot(DIF ? row_high(e) : row_low(e), c & 3)); });
            wave_sync();
        }
    } else {
        static_for<8>([&](auto ec) { constexpr int e = decltype(ec)::value; x[e] = load_elem(widx(DIF ? row_high(e) : row_low(e), c)); });
    }

#if !defined(FRW_QAP_NO_STAGES)
    // ---- first round, exchange, second round ---------------------------------------------------------------------------------
    // (the three constants of the low round are fetched where that round is: held across the other one they cost 27
    // registers the decimation-in-frequency passes do not have)
    constexpr bool LAZY = MODE == PASS_DIF && STORE == STORE_FACTOR;     // see dif_mul_lazy
    if (FUSED) {
        // The last pass of an inverse transform and the first pass of the forward transform that follows it work on the same
        // index bits, i.e. on the same tile, and the first leaves a thread the rows (8 e + g) the second starts from: the
        // element-wise factor between them is applied in registers and the array makes one trip through memory instead of two.
        round_low<false>(x, root_get(p.roots, 8), root_get(p.roots, 16), root_get(p.roots, 24));
        exchange(x);
        round_high<false, T>(x, p.roots, g);                                                                          // < 22 p
        static_for<8>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            x[e] = QAP_MUL(x[e], f29_unpack(fr_load(factor + (size_t)widx(row_high(e), c) * 8)));                     // < 2 p
        });
        round_high<true, T, true>(x, p.roots2, g);
        exchange(x);
        if (T == 6) static_for<8>([&](auto ec) { constexpr int e = decltype(ec)::value; x[e] = f29_cond_sub_kp<8>(x[e]); });
        round_low<true, true>(x, root_get(p.roots2, 8), root_get(p.roots2, 16), root_get(p.roots2, 24));             // < 64 p; the factor at the store: < 2 p
    } else {
    if (DIF) round_high<true, T, LAZY>(x, p.roots, g);
    else round_low<false>(x, root_get(p.roots, 8), root_get(p.roots, 16), root_get(p.roots, 24));
    exchange(x);
    if (LAZY && T == 6) static_for<8>([&](auto ec) { constexpr int e = decltype(ec)::value; x[e] = f29_cond_sub_kp<8>(x[e]); });   // < 16 p -> < 8 p
    if (DIF) round_low<true, LAZY>(x, root_get(p.roots, 8), root_get(p.roots, 16), root_get(p.roots, 24));
    else round_high<false, T>(x, p.roots, g);
    }
#endif

    // ---- out ---------------------------------------------------------------------------------------------------------------------
    if (MODE == PASS_FIRST || MEMORDER) {
        // rows of the last round -> LDS -> memory order, four columns at a time
        // (the stores' indices from a laundered thread index: see the other branch)
        uint32_t tid_st = (uint32_t)tid;
        QAP_LAUNDER(tid_st);
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if ((c >> 2) == h)
                static_for<8>([&](auto ec) { constexpr int e = decltype(ec)::value; lds_put(mslot(DIF ? row_low(e) : row_high(e), c & 3), x[e]); });
            wave_sync();
#pragma unroll
            for (int k = 0; k < 4; k++) store_elem(widx((int)tid_st, 4 * h + k), lds_get(mslot((int)tid_st, k)));
            wave_sync();
        }
    } else {
        // The indices of the eight elements are a handful of integer operations on (tile, thread) -- but computed once, where the elements
        // are loaded, the compiler keeps the eight 64-bit addresses alive across the whole pass for these stores (an in-place pass
        // stores where it loaded): sixteen registers the butterflies do not have at three wavefronts per SIMD -- they went to scratch
        // memory (round 4: 28 - 112 bytes per lane in the upper passes).  The thread index is laundered through an empty asm, so the
        // indices are derived again here, from scratch only in the other sense.
        uint32_t tid_out = (uint32_t)tid;
        QAP_LAUNDER(tid_out);
        const int c_out = (int)(tid_out & (COLS - 1)), g_out = (int)(tid_out >> CB);
        auto row_low_out = [&](int e) { return g_out * 8 + e; };
        auto row_high_out = [&](int e) { return T == 6 ? e * 8 + g_out : T == 5 ? (e & 3) * 8 + g_out + 4 * (e >> 2) : (e & 1) * 8 + g_out + 2 * (e >> 1); };
        static_for<8>([&](auto ec) {
            constexpr int e = decltype(ec)::value;
            const int row = DIF || FUSED ? row_low_out(e) : row_high_out(e);
            store_elem(high | ((uint32_t)row << p.sh) | lowmid | (uint32_t)c_out, x[e]);
        });
    }
    if (LIST) wave_sync();                                           // the next turn reuses the LDS
    }
}

#if !defined(FRW_QAP_PROBE)     // (tools/kernel_resources.sh compiles single instantiations of the pass kernel through a probe unit)
namespace {
template <int MODE, int T, int LOAD, int STORE>
hipError_t launch_pass(const NttPass &p, unsigned arrays, hipStream_t st)
{
    const unsigned tiles = (unsigned)(((size_t)1 << p.L) / QAP_TILE);
    if (p.list) hipLaunchKernelGGL((ntt_pass_kernel<MODE, T, LOAD, STORE, true>), dim3(tiles, arrays), dim3(QAP_THREADS), 0, st, p);
    else hipLaunchKernelGGL((ntt_pass_kernel<MODE, T, LOAD, STORE>), dim3(tiles, arrays), dim3(QAP_THREADS), 0, st, p);
    return hipGetLastError();
}
// an upper pass (sh > 0) of four, five or six stages
template <int MODE, int STORE>
hipError_t launch_upper(int t, const NttPass &p, unsigned arrays, hipStream_t st)
{
    return t == 6 ? launch_pass<MODE, 6, LOAD_PLAIN, STORE>(p, arrays, st)
         : t == 5 ? launch_pass<MODE, 5, LOAD_PLAIN, STORE>(p, arrays, st)
                  : launch_pass<MODE, 4, LOAD_PLAIN, STORE>(p, arrays, st);
}

enum { XF_IFFT_FROM_PRODUCTS, XF_FFT, XF_IFFT_POINTWISE_TO_H,
       XF_IFFT_FROM_PRODUCTS_PSI, XF_IFFT_AB_TO_H, XF_IFFT_PRODUCTS_AB_ADD_H };

// One transform over `arrays` arrays with the element-wise steps of the witness map fused into its first and last pass:
//   XF_IFFT_FROM_PRODUCTS     bit-reversed -> natural; reads A z, B z, C z (constraint order); x g^k / n at the end
//   XF_FFT                    natural -> bit-reversed, in place
//   XF_IFFT_POINTWISE_TO_H    bit-reversed -> natural; a b - c at the start; x g^-k / (n (g^n - 1)), canonical, into h
// An inverse transform runs its passes from the lowest index bits up (q.pass_sh / q.pass_t: pass 0 = bits [0, 6), K passes
// in all), every pass but the last leaving the twist towards the next one; a forward transform runs them from the top down.
// Unless built with -DFRW_QAP_NO_FUSE, the last (top) pass of an inverse transform that is followed by a forward one also
// does that one's first (top) pass: PASS_DIT_DIF.
hipError_t transform(int kind, const QapDev &q, const NttPass &base, uint32_t *work, size_t work_stride, uint32_t *h,
                     unsigned arrays, hipStream_t st)
{
    const int L = q.log_n, K = q.num_passes, top = K - 1;
    if (K < 2) return hipErrorInvalidValue;
    NttPass p = base;
    p.L = L;
    hipError_t e;
    if (kind == XF_FFT) {
        p.roots = q.roots_fwd;
        p.src = work; p.dst = work; p.src_stride = p.dst_stride = work_stride;
#if defined(FRW_QAP_NO_FUSE)
        const int first = top;
#else
        const int first = top - 1;                          // the inverse transform before it has done the top pass: PASS_DIT_DIF
#endif
        for (int k = first; k >= 1; k--) {
            p.sh = q.pass_sh[k]; p.factor = q.twist_fwd[k - 1];
            if ((e = launch_upper<PASS_DIF, STORE_FACTOR>(q.pass_t[k], p, arrays, st)) != hipSuccess) return e;
        }
        p.sh = 0; p.factor = nullptr;
        return launch_pass<PASS_DIF_SH0, 6, LOAD_PLAIN, STORE_PLAIN>(p, arrays, st);
    }
    p.roots = q.roots_inv;
    // the passes between the first and the last of an inverse transform
    auto middle = [&]() -> hipError_t {
        for (int k = 1; k < top; k++) {
            p.sh = q.pass_sh[k]; p.factor = q.twist_inv[k];
            const hipError_t me = launch_upper<PASS_DIT, STORE_FACTOR>(q.pass_t[k], p, arrays, st);
            if (me != hipSuccess) return me;
        }
        return hipSuccess;
    };
    const int tt = q.pass_t[top];
    if (kind == XF_IFFT_FROM_PRODUCTS_PSI) {
        // the two-array variant of XF_IFFT_FROM_PRODUCTS for the six-transform quotient: x psi^k / n at the end
        p.src = work; p.dst = work; p.src_stride = p.dst_stride = work_stride;
        p.per_sig = 2;
        p.sh = 0; p.factor = q.twist_inv[0];
        if ((e = launch_pass<PASS_FIRST, 6, LOAD_PLAIN, STORE_FACTOR>(p, arrays, st)) != hipSuccess) return e;
        if ((e = middle()) != hipSuccess) return e;
        p.sh = q.pass_sh[top]; p.factor = q.scale_psi_in;
#if defined(FRW_QAP_NO_FUSE)
        return launch_upper<PASS_DIT, STORE_FACTOR>(tt, p, arrays, st);
#else
        p.roots2 = q.roots_fwd; p.factor2 = q.twist_fwd[top - 1];     // ... and the first pass of the XF_FFT that follows
        return launch_upper<PASS_DIT_DIF, STORE_FACTOR>(tt, p, arrays, st);
#endif
    }
    if (kind == XF_IFFT_AB_TO_H) {
        // a b on the coset psi H from the two arrays of each signature, in place on the first; -16 psi^-k / n into h
        p.src = work; p.src_stride = 2 * work_stride; p.dst = work; p.dst_stride = 2 * work_stride;
        p.sh = 0; p.factor = q.twist_inv[0];
        if ((e = launch_pass<PASS_DIT_SH0, 6, LOAD_AB, STORE_FACTOR>(p, arrays, st)) != hipSuccess) return e;
        if ((e = middle()) != hipSuccess) return e;
        p.sh = q.pass_sh[top]; p.factor = q.scale_psi_out; p.dst = h; p.dst_stride = work_stride;
        return launch_upper<PASS_DIT, STORE_FACTOR>(tt, p, arrays, st);
    }
    if (kind == XF_IFFT_PRODUCTS_AB_ADD_H) {
        // (A z)_i (B z)_i on the domain itself -> coefficients of (a b) mod (X^n - 1), x 16 / n, added to what h holds
        p.src = work; p.src_stride = 2 * work_stride; p.dst = work; p.dst_stride = 2 * work_stride;
        p.per_sig = 1;
        p.sh = 0; p.factor = q.twist_inv[0];
        if ((e = launch_pass<PASS_FIRST, 6, LOAD_PRODUCTS_AB, STORE_FACTOR>(p, arrays, st)) != hipSuccess) return e;
        if ((e = middle()) != hipSuccess) return e;
        p.sh = q.pass_sh[top]; p.factor = nullptr; p.dst = h; p.dst_stride = work_stride;
        for (int k = 0; k < 9; k++) p.cfac[k] = q.sixteen_over_n[k];
        return launch_upper<PASS_DIT, STORE_CONST_ADD_CANONICAL>(tt, p, arrays, st);
    }
    if (kind == XF_IFFT_FROM_PRODUCTS) {
        p.per_sig = 3;
        p.src = work; p.dst = work; p.src_stride = p.dst_stride = work_stride;
        p.sh = 0; p.factor = q.twist_inv[0];
        if ((e = launch_pass<PASS_FIRST, 6, LOAD_PLAIN, STORE_FACTOR>(p, arrays, st)) != hipSuccess) return e;
        if ((e = middle()) != hipSuccess) return e;
        p.sh = q.pass_sh[top]; p.factor = q.scale_in; p.factor_a = q.scale_in_a;
#if defined(FRW_QAP_NO_FUSE)
        return launch_upper<PASS_DIT, STORE_FACTOR_A>(tt, p, arrays, st);
#else
        p.roots2 = q.roots_fwd; p.factor2 = q.twist_fwd[top - 1];     // ... and the first pass of the XF_FFT that follows
        return launch_upper<PASS_DIT_DIF, STORE_FACTOR_A>(tt, p, arrays, st);
#endif
    }
    // a b - c from the three arrays of each signature, in place on the first of them; the last pass writes h
    p.src = work; p.src_stride = 3 * work_stride; p.dst = work; p.dst_stride = 3 * work_stride;
    p.sh = 0; p.factor = q.twist_inv[0];
    if ((e = launch_pass<PASS_DIT_SH0, 6, LOAD_AB_MINUS_C, STORE_FACTOR>(p, arrays, st)) != hipSuccess) return e;
    if ((e = middle()) != hipSuccess) return e;
    p.sh = q.pass_sh[top]; p.factor = q.scale_out; p.dst = h; p.dst_stride = work_stride;
    return launch_upper<PASS_DIT, STORE_FACTOR_CANONICAL>(tt, p, arrays, st);
}
}  // namespace

// ---- diagnostics: the VALU issue rates the transforms' roofline is priced with, measured on the device at hand ----------
// KIND 0: v_add_u32 (full-rate class: the carry / mask instructions of a product), 1: v_mad_u64_u32 (the multiply-adds),
// 2: f29_mul itself.  Every wave runs ITER rounds of 16 (KIND 2: 4) independent chains; four waves per SIMD.
constexpr int DIAG_ITER = 2048;
template <int KIND>
__global__ __launch_bounds__(256) void valu_rate_kernel(uint64_t *out, uint32_t seed)
{
    uint32_t x = seed + threadIdx.x, y = seed * 3 + 1;
    uint64_t acc = 0;
    if (KIND == 2) {
        F29 v[4], w;
#pragma unroll
        for (int k = 0; k < NL29; k++) w.l[k] = (seed * 2654435761u + k * 40503u) & M29;
        w.l[NL29 - 1] &= 0x3fffff;                                  // < p
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < NL29; k++) v[j].l[k] = (x * (j + 3) + k) & M29;
        for (int it = 0; it < DIAG_ITER / 8; it++)
#pragma unroll
            for (int j = 0; j < 4; j++) v[j] = f29_mul(v[j], w);
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
            for (int k = 0; k < NL29; k++) acc += v[j].l[k];
    } else {
        uint64_t a[16];
#pragma unroll
        for (int i = 0; i < 16; i++) a[i] = x + i;
        for (int it = 0; it < DIAG_ITER; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++) {
                if (KIND == 0) { uint32_t t = (uint32_t)a[i]; asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(y)); a[i] = t; }
                else asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc");
            }
        }
#pragma unroll
        for (int i = 0; i < 16; i++) acc += a[i];
    }
    out[(size_t)blockIdx.x * 256 + threadIdx.x] = acc;
}

// out[0] = v_add_u32, out[1] = v_mad_u64_u32 in wave-instructions per SIMD per microsecond; out[2] = f29_mul products per
// second over the chip (the multiplier loop above: no loads, no butterflies); out[3] = SIMDs.  `scratch`: num_cu x 8 KiB.
hipError_t diag_valu_rates(int num_cu, void *scratch, double out[4], hipStream_t st)
{
    const int grid = num_cu * 4;                                   // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0 = nullptr, e1 = nullptr;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    float ms[3] = {0, 0, 0};
    for (int kind = 0; kind < 3 && e == hipSuccess; kind++) {
        for (int rep = 0; rep < 2 && e == hipSuccess; rep++) {     // the second run is the timed one
            if (rep) e = hipEventRecord(e0, st);
            if (kind == 0) hipLaunchKernelGGL(valu_rate_kernel<0>, dim3(grid), dim3(256), 0, st, (uint64_t *)scratch, 1u + rep);
            else if (kind == 1) hipLaunchKernelGGL(valu_rate_kernel<1>, dim3(grid), dim3(256), 0, st, (uint64_t *)scratch, 1u + rep);
            else hipLaunchKernelGGL(valu_rate_kernel<2>, dim3(grid), dim3(256), 0, st, (uint64_t *)scratch, 1u + rep);
            if (e == hipSuccess) e = hipGetLastError();
            if (rep && e == hipSuccess) e = hipEventRecord(e1, st);
        }
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms[kind], e0, e1);
    }
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    if (e != hipSuccess) return e;
    out[0] = 4.0 * DIAG_ITER * 16 / (ms[0] * 1e3);
    out[1] = 4.0 * DIAG_ITER * 16 / (ms[1] * 1e3);
    out[2] = (double)grid * 256 * (DIAG_ITER / 8) * 4 / (ms[2] * 1e-3);
    out[3] = 4.0 * num_cu;
    return hipSuccess;
}

// the signatures of a chunk whose witness violates the system (num_unsatisfied != 0), in any order, and how many
__global__ __launch_bounds__(256) void qap_list_kernel(const uint32_t *__restrict__ num_unsatisfied, uint32_t cnt, uint32_t *__restrict__ list,
                                                       uint32_t *__restrict__ list_count)
{
    for (uint32_t i = threadIdx.x; i < cnt; i += 256)
        if (num_unsatisfied[i]) list[atomicAdd(list_count, 1u)] = i;
}

constexpr size_t QAP_FLAGS_BYTES = 64;                              // per signature in flight: its count of violated rows, its place in the list, the list's length
size_t qap_workspace_bytes_per_signature(const R1csDev &r, const QapDev &q)
{
    return 3 * (size_t)r.num_constraints * 32 + 3 * 32 * ((size_t)1 << q.log_n) + QAP_FLAGS_BYTES;
}

// workspace per signature in flight: A z, B z, C z (3 C x 32 B) + three working arrays (3 n x 32 B) + 64 bytes of flags; the
// batch is cut into chunks that fit.
//   QAP_SIX     h = hi of a b (six transforms): ark-groth16's h for every witness that satisfies the system
//   QAP_SEVEN   ark-groth16's witness_map as it is written (seven transforms), whatever the witness
//   QAP_EXACT   QAP_SIX for the whole chunk, then QAP_SEVEN in list mode for the signatures whose witness violates the system
//               (none, normally: 18 launches of a few hundred workgroups that read a zero and leave): ark-groth16's h for
//               EVERY input at the price of six transforms
namespace {
enum { QAP_SIX, QAP_SEVEN, QAP_EXACT };
hipError_t qap_run(int mode, const R1csDev &r, const QapDev &q, size_t batch, const uint64_t *witness, const uint64_t *instance, uint64_t *h,
                   uint32_t *num_unsatisfied, void *workspace, size_t workspace_bytes, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    const int L = q.log_n;
    if (q.num_passes < 2) return hipErrorInvalidValue;           // no pass schedule for this domain (qap_pass_schedule)
    const size_t n = (size_t)1 << L, per_sig = qap_workspace_bytes_per_signature(r, q);
    size_t chunk = workspace_bytes / per_sig;
    if (chunk == 0) return hipErrorInvalidValue;
    if (chunk > 16384) chunk = 16384;                            // grid.y = 3 x chunk <= 65535
    for (size_t lo = 0; lo < batch; lo += chunk) {
        const size_t cnt = batch - lo < chunk ? batch - lo : chunk;
        uint32_t *abc = (uint32_t *)workspace;
        uint32_t *work = abc + cnt * 3 * (size_t)r.num_constraints * 8;
        uint32_t *own_flags = work + cnt * 3 * n * 8, *list = own_flags + cnt, *list_count = list + cnt;
        uint32_t *flags = num_unsatisfied ? num_unsatisfied + lo : (mode == QAP_EXACT ? own_flags : nullptr);
        const uint64_t *wit = witness + lo * (size_t)r.num_witness * 4, *inst = instance + lo * (size_t)r.num_instance * 4;
        uint32_t *hh = (uint32_t *)(h + lo * n * 4);
        hipError_t e = launch_r1cs_check(r, cnt, wit, inst, flags, (uint64_t *)abc, st, work);   // the working arrays are idle during the products: they lend the scratch
        if (e != hipSuccess) return e;
        NttPass p{};
        p.abc = abc;
        p.instance = (const uint32_t *)inst;
        p.num_constraints = r.num_constraints;
        p.num_instance = r.num_instance;
        if (mode != QAP_SEVEN) {
            // a b on the domain and on the coset psi H: two arrays per signature
            if ((e = transform(XF_IFFT_FROM_PRODUCTS_PSI, q, p, work, n * 8, nullptr, (unsigned)(2 * cnt), st)) != hipSuccess) return e;
            if ((e = transform(XF_FFT, q, p, work, n * 8, nullptr, (unsigned)(2 * cnt), st)) != hipSuccess) return e;
            if ((e = transform(XF_IFFT_AB_TO_H, q, p, work, n * 8, hh, (unsigned)cnt, st)) != hipSuccess) return e;
            if ((e = transform(XF_IFFT_PRODUCTS_AB_ADD_H, q, p, work, n * 8, hh, (unsigned)cnt, st)) != hipSuccess) return e;
            if (mode == QAP_SIX) continue;
            if ((e = hipMemsetAsync(list_count, 0, 4, st)) != hipSuccess) return e;
            hipLaunchKernelGGL(qap_list_kernel, dim3(1), dim3(256), 0, st, flags, (uint32_t)cnt, list, list_count);
            if ((e = hipGetLastError()) != hipSuccess) return e;
            p.list = list;
            p.list_count = list_count;
        }
        const unsigned per3 = p.list ? 3u : (unsigned)(3 * cnt), per1 = p.list ? 1u : (unsigned)cnt;
        // ifft + distribute_powers(g) (A z with an extra 2^5, which the a b product in R' = 2^261 arithmetic takes out again)
        if ((e = transform(XF_IFFT_FROM_PRODUCTS, q, p, work, n * 8, nullptr, per3, st)) != hipSuccess) return e;
        // fft: a, b, c on the coset (bit-reversed order)
        if ((e = transform(XF_FFT, q, p, work, n * 8, nullptr, per3, st)) != hipSuccess) return e;
        // (a b - c) / Z on the coset, coset_ifft
        if ((e = transform(XF_IFFT_POINTWISE_TO_H, q, p, work, n * 8, hh, per1, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace

hipError_t launch_qap_witness_map(const R1csDev &r, const QapDev &q, size_t batch, const uint64_t *witness,
                                  const uint64_t *instance, uint64_t *h, uint32_t *num_unsatisfied, void *workspace,
                                  size_t workspace_bytes, hipStream_t st)
{
#if defined(FRW_QAP_SEVEN_ALWAYS)
    return qap_run(QAP_SEVEN, r, q, batch, witness, instance, h, num_unsatisfied, workspace, workspace_bytes, st);
#else
    return qap_run(QAP_EXACT, r, q, batch, witness, instance, h, num_unsatisfied, workspace, workspace_bytes, st);
#endif
}

// The same quotient with six transforms instead of seven, for witnesses that satisfy the system (num_unsatisfied says):
// a(X) b(X) = lo + X^n hi;  S = lo + hi from the pointwise products on the domain, N = lo - hi from the pointwise products
// on the coset psi H (psi^n = -1);  h = hi = (S - N) / 2.  C z is still computed -- for the satisfaction count -- but not
// transformed.  For a witness that violates the system the result is hi all the same, which is then NOT what ark-groth16
// returns (nor a quotient of anything).
hipError_t launch_qap_quotient(const R1csDev &r, const QapDev &q, size_t batch, const uint64_t *witness,
                               const uint64_t *instance, uint64_t *h, uint32_t *num_unsatisfied, void *workspace,
                               size_t workspace_bytes, hipStream_t st)
{
    return qap_run(QAP_SIX, r, q, batch, witness, instance, h, num_unsatisfied, workspace, workspace_bytes, st);
}

#endif   // FRW_QAP_PROBE

}  // namespace frw
