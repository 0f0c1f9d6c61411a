// frw_qap.hip -- the R1CS -> QAP witness map on the device: h(X) = (A(X) B(X) - C(X)) / (X^n - 1) over BLS12-381 Fr.
//
// What a Groth16 prover does with the witness right after the hot path (examples/pok_sig.rs:30-47 calls
// Groth16::prove; ark-groth16 0.3.0 r1cs_to_qap.rs R1CStoQAP::witness_map):
//     a = A z (+ the instance values in rows C .. C+I), b = B z, c = C z       over the domain of n-th roots of unity
//     a, b, c <- coset_fft(ifft(.));   ab = a o b - c;   ab /= (g^n - 1);   h = coset_ifft(ab)
// Here, with the transforms arranged so that no permutation pass and no scattered access is needed:
//     r1cs_check_kernel writes A z, B z, C z in constraint order (ark-ff's 8 x 32-bit Montgomery form)
//     inverse transform, decimation in time; its first pass reads that order through the bit-reversed tile
//         (64 rows 2^(L-6) apart x 16 adjacent rows), appends the instance rows and the zero padding on the fly, and
//         writes the working form; its last pass multiplies by g^k / n on the way out
//     forward transform, decimation in frequency (natural in, bit-reversed out)      -> values on the coset
//     inverse transform, decimation in time, first pass fused with a b - c, last pass with g^-k / (n (g^n - 1)),
//         canonical reduction and re-packing                                          -> h, ark-ff form, natural order
// Working form (frw_fr29.h): nine 29-bit limbs per element, one plane per limb ([9][n] u32 per array), values lazily
// reduced (< 2^261).  Every transform is three passes over HBM (6 + 6 + (L - 12) butterfly stages), a workgroup taking a
// tile of 2^T rows x 16 adjacent elements through LDS.  Field arithmetic is exact, so the result is the same element
// for element as any other schedule's.
//
// Cost per signature (n = 2^18): 7 transforms x 9 n + 5 n Montgomery products (~17.8 M): bound by VALU issue
// (~290 instructions per butterfly), not by HBM (~0.5 GB of traffic per signature).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frw_device.h"
#include "frw_fr29.h"

namespace frw {

constexpr int QAP_COLS = 16;            // adjacent elements per tile row
constexpr int QAP_MAX_T = 6;            // butterfly stages per pass
constexpr int QAP_TILE = (1 << QAP_MAX_T) * QAP_COLS;

enum { PASS_FIRST = 0, PASS_DIT_SH0 = 1, PASS_DIT = 2, PASS_DIF = 3, PASS_DIF_SH0 = 4 };
enum { LOAD_PLAIN = 0, LOAD_AB_MINUS_C = 1 };
enum { STORE_PLAIN = 0, STORE_SCALE = 1, STORE_SCALE_PACK = 2 };

struct NttPass {
    uint32_t *x;                // working arrays: [arrays][9][n] u32
    size_t x_stride;            // words between consecutive arrays of this launch
    const uint32_t *tw;         // root^k in R' form, planes [9][n/2]
    const uint32_t *scale;      // STORE_SCALE*: factor per natural index, planes [9][n]
    const uint32_t *scale0;     // STORE_SCALE: the table for arrays 0, 3, 6, ... (A z carries an extra 2^5, see below)
    uint32_t *out_packed;       // STORE_SCALE_PACK: [arrays][n][8] u32 (ark-ff form)
    // PASS_FIRST sources
    const uint32_t *abc;        // [signatures][3][C][8] u32
    const uint32_t *instance;   // [signatures][I][8] u32
    uint32_t num_constraints, num_instance;
    int L, sh, T;               // log n; lowest index bit this pass transforms; stages in this pass
};

// A/B builds (tools/ab_qap.py): -DFRW_QAP_NO_TW reads one fixed twiddle, -DFRW_QAP_NO_MUL replaces the products by additions,
// -DFRW_QAP_NO_STAGES skips the butterflies (passes become copies).  Results are then wrong, timings tell what a pass waits for.
#if defined(FRW_QAP_NO_MUL)
#define QAP_MUL(a, b) f29_add(a, b)
#else
#define QAP_MUL(a, b) f29_mul(a, b)
#endif
#if defined(FRW_QAP_NO_TW)
#define QAP_TW(p, n, idx) planes_get((p).tw, (n) / 2, 1)
#else
#define QAP_TW(p, n, idx) planes_get((p).tw, (n) / 2, idx)
#endif

__device__ __forceinline__ F29 planes_get(const uint32_t *base, size_t plane_words, size_t idx)
{
    F29 r;
#pragma unroll
    for (int k = 0; k < NL29; k++) r.l[k] = base[k * plane_words + idx];
    return r;
}
__device__ __forceinline__ void planes_put(uint32_t *base, size_t plane_words, size_t idx, const F29 &v)
{
#pragma unroll
    for (int k = 0; k < NL29; k++) base[k * plane_words + idx] = v.l[k];
}

// One pass = T consecutive radix-2 stages on the index bits [sh, sh + T) of the working order.
//   decimation in time (PASS_FIRST, PASS_DIT_SH0, PASS_DIT): stages bottom-up, (u, v) -> (u + w v, u - w v); values grow
//       by < 2 p per stage and are reduced only by the products (and by the scale factor of the last pass);
//   decimation in frequency (PASS_DIF, PASS_DIF_SH0): stages top-down, (u, v) -> (u + v, (u - v) w), everything kept < 2 p.
// Tile geometry: *_SH0 (sh == 0): 16 2^T consecutive elements, tile-linear order = memory order (row = low bits);
// otherwise the 16 columns are the index bits [0, 4) and the rows the bits [sh, sh + T).
// PASS_FIRST is the sh == 0 pass of a decimation-in-time transform whose input is still in natural order in the packed
// buffers: working index pos = bitrev(i), so the 64 rows are the natural-index bits [L - 6, L) reversed and the 16
// columns the natural-index bits [0, 4): 512-byte runs on the way in, 256-byte runs per plane on the way out.
// LOAD / STORE (the element-wise steps fused into the first / last pass of a transform) are compile-time: a run-time
// `store_op` made the compiler thread the scale-table select through the store loop and leave the pointer undefined on
// one path (memory fault on the first GPU run of this file).
template <int MODE, int LOAD, int STORE>
__global__ __launch_bounds__(BLOCK) void ntt_pass_kernel(const NttPass p)
{
    constexpr bool DIF = MODE == PASS_DIF || MODE == PASS_DIF_SH0;
    constexpr bool SH0 = MODE == PASS_DIT_SH0 || MODE == PASS_DIF_SH0;      // memory-order tile
    constexpr bool LOWJ = MODE == PASS_DIT || MODE == PASS_DIF;             // twiddle exponent includes the low index bits
    __shared__ uint32_t tile[NL29 * QAP_TILE];
    const int tid = threadIdx.x;
    const int R = 1 << p.T, elems = R * QAP_COLS;
    const uint32_t tileid = blockIdx.x;
    const size_t n = (size_t)1 << p.L;
    uint32_t *xa = p.x + (size_t)blockIdx.y * p.x_stride;
    const uint32_t *sc = STORE == STORE_SCALE && blockIdx.y % 3 == 0 ? p.scale0 : p.scale;

    // ---- tile-linear index -> LDS slot and working index ---------------------------------------------------------------
    // slots: SH0: lin (= c R + r);  otherwise r 16 + c
    uint32_t lowmid = 0, high = 0;
    if (MODE == PASS_FIRST) {
        // tileid = natural-index bits [4, L - 6); working index = r | rev(tileid) << 6 | rev4(c) << (L - 4)
        high = (__brev(tileid) >> (32 - (p.L - 10))) << 6;
    } else if (SH0) {
        high = tileid * (uint32_t)elems;
    } else {
        lowmid = (tileid & ((1u << (p.sh - 4)) - 1u)) << 4;
        high = (tileid >> (p.sh - 4)) << (p.sh + p.T);
    }
    auto widx = [&](int lin) -> uint32_t {          // working index of the element in LDS slot `lin`
        if (MODE == PASS_FIRST) return (uint32_t)(lin >> 4) | high | ((__brev((uint32_t)(lin & 15)) >> 28) << (p.L - 4));
        if (SH0) return high + (uint32_t)lin;
        return high | ((uint32_t)(lin >> 4) << p.sh) | lowmid | (uint32_t)(lin & 15);
    };

    // ---- load ------------------------------------------------------------------------------------------------------------
    if (MODE == PASS_FIRST) {
        const size_t sig = blockIdx.y / 3, which = blockIdx.y % 3;
        const uint32_t *src = p.abc + (sig * 3 + which) * (size_t)p.num_constraints * 8;
        const uint32_t *inst = p.instance + sig * (size_t)p.num_instance * 8;
        for (int lin = tid; lin < elems; lin += BLOCK) {
            const uint32_t ihi = (uint32_t)(lin >> 4), c = (uint32_t)(lin & 15);
            const uint32_t i = (ihi << (p.L - 6)) | (tileid << 4) | c;          // natural (constraint) index
            Fr8 w;
            if (i < p.num_constraints) w = fr_load(src + (size_t)i * 8);
            else if (which == 0 && i - p.num_constraints < p.num_instance) w = fr_load(inst + (size_t)(i - p.num_constraints) * 8);
            else {
#pragma unroll
                for (int k = 0; k < 8; k++) w.l[k] = 0;
            }
            const F29 v = f29_unpack(w);
            const int slot = (int)(__brev(ihi) >> 26) * QAP_COLS + (int)c;     // row r = rev6(ihi)
#pragma unroll
            for (int k = 0; k < NL29; k++) tile[k * QAP_TILE + slot] = v.l[k];
        }
    } else {
        for (int lin = tid; lin < elems; lin += BLOCK) {
            const uint32_t g = widx(lin);
            F29 v = planes_get(xa, n, g);
            if (LOAD == LOAD_AB_MINUS_C) {
                // arrays of one signature lie 9 n words apart: (a 2^5 R)(b R) / R' - c R = (a b - c) R
                const F29 b = planes_get(xa + NL29 * n, n, g), c = planes_get(xa + 2 * NL29 * n, n, g);
                v = f29_reduce_4p(f29_sub_2p(f29_mul(v, b), c));                 // < 2 p
            }
#pragma unroll
            for (int k = 0; k < NL29; k++) tile[k * QAP_TILE + lin] = v.l[k];
        }
    }
    __syncthreads();

    // ---- stages ----------------------------------------------------------------------------------------------------------
#if !defined(FRW_QAP_NO_STAGES)
    for (int ti = 0; ti < p.T; ti++) {
        const int t = DIF ? p.T - ti : ti + 1, s = p.sh + t, hr = 1 << (t - 1);
        for (int k = tid; k < elems / 2; k += BLOCK) {
            const int b = SH0 ? k & (R / 2 - 1) : k >> 4, c = SH0 ? k >> (p.T - 1) : k & 15;
            const int r_lo = b & (hr - 1), r = ((b >> (t - 1)) << t) | r_lo;
            const int s0 = SH0 ? c * R + r : r * QAP_COLS + c, s1 = s0 + (SH0 ? hr : hr * QAP_COLS);
            const uint32_t j = LOWJ ? ((uint32_t)r_lo << p.sh) | lowmid | (uint32_t)c : (uint32_t)r_lo;
            F29 u, v;
#pragma unroll
            for (int q = 0; q < NL29; q++) { u.l[q] = tile[q * QAP_TILE + s0]; v.l[q] = tile[q * QAP_TILE + s1]; }
            F29 nu, nv;
            if (DIF) {
                nu = f29_reduce_4p(f29_add(u, v));
                nv = f29_sub_2p(u, v);                                       // < 4 p
                if (s == 1) nv = f29_reduce_4p(nv);                          // the twiddle is one
                else nv = QAP_MUL(nv, QAP_TW(p, n, (size_t)j << (p.L - s)));
            } else {
                if (s != 1) v = QAP_MUL(v, QAP_TW(p, n, (size_t)j << (p.L - s)));      // < 2 p
                nu = f29_add(u, v);
                nv = f29_sub_2p(u, v);
            }
#pragma unroll
            for (int q = 0; q < NL29; q++) { tile[q * QAP_TILE + s0] = nu.l[q]; tile[q * QAP_TILE + s1] = nv.l[q]; }
        }
        __syncthreads();
    }
#endif

    // ---- store -----------------------------------------------------------------------------------------------------------
    for (int lin = tid; lin < elems; lin += BLOCK) {
        const uint32_t g = widx(lin);
        F29 v;
#pragma unroll
        for (int k = 0; k < NL29; k++) v.l[k] = tile[k * QAP_TILE + lin];
        if (STORE != STORE_PLAIN) v = f29_mul(v, planes_get(sc, n, g));     // < 2 p
        if (STORE == STORE_SCALE_PACK)
            fr_store(p.out_packed + ((size_t)blockIdx.y * n + g) * 8, f29_pack(f29_canonical(v)));
        else
            planes_put(xa, n, g, v);
    }
}

namespace {
template <int MODE, int LOAD, int STORE>
hipError_t launch_pass(const NttPass &p, unsigned arrays, hipStream_t st)
{
    const unsigned tiles = (unsigned)(((size_t)1 << p.L) >> (p.T + 4));
    hipLaunchKernelGGL((ntt_pass_kernel<MODE, LOAD, STORE>), dim3(tiles, arrays), dim3(BLOCK), 0, st, p);
    return hipGetLastError();
}

enum { XF_IFFT_FROM_PACKED, XF_FFT, XF_IFFT_POINTWISE_TO_PACKED };

// One transform over `arrays` working arrays, in place, with the element-wise steps of the witness map fused into its
// first and last pass:
//   XF_IFFT_FROM_PACKED          bit-reversed -> natural; reads A z, B z, C z (packed, natural order); x g^k / n at the end
//   XF_FFT                       natural -> bit-reversed
//   XF_IFFT_POINTWISE_TO_PACKED  bit-reversed -> natural; a b - c at the start; x g^-k / (n (g^n - 1)), packed, at the end
hipError_t transform(int kind, const QapDev &q, const NttPass &base, unsigned arrays, hipStream_t st)
{
    const int L = q.log_n;
    const bool dif = kind == XF_FFT;
    int shs[8], ts[8], np = 0;
    for (int sh = 0; sh < L; sh += QAP_MAX_T) { shs[np] = sh; ts[np] = L - sh < QAP_MAX_T ? L - sh : QAP_MAX_T; np++; }
    for (int i = 0; i < np; i++) {
        const int k = dif ? np - 1 - i : i;
        const bool last = i == np - 1;
        NttPass p = base;
        p.tw = dif ? q.tw_fwd : q.tw_inv;
        p.L = L; p.sh = shs[k]; p.T = ts[k];
        hipError_t e;
        if (kind == XF_FFT)
            e = p.sh == 0 ? launch_pass<PASS_DIF_SH0, LOAD_PLAIN, STORE_PLAIN>(p, arrays, st)
                          : launch_pass<PASS_DIF, LOAD_PLAIN, STORE_PLAIN>(p, arrays, st);
        else if (kind == XF_IFFT_FROM_PACKED)
            e = p.sh == 0 ? launch_pass<PASS_FIRST, LOAD_PLAIN, STORE_PLAIN>(p, arrays, st)
                : last    ? launch_pass<PASS_DIT, LOAD_PLAIN, STORE_SCALE>(p, arrays, st)
                          : launch_pass<PASS_DIT, LOAD_PLAIN, STORE_PLAIN>(p, arrays, st);
        else
            e = p.sh == 0 ? launch_pass<PASS_DIT_SH0, LOAD_AB_MINUS_C, STORE_PLAIN>(p, arrays, st)
                : last    ? launch_pass<PASS_DIT, LOAD_PLAIN, STORE_SCALE_PACK>(p, arrays, st)
                          : launch_pass<PASS_DIT, LOAD_PLAIN, STORE_PLAIN>(p, arrays, st);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace

size_t qap_workspace_bytes_per_signature(const R1csDev &r, const QapDev &q)
{
    return 3 * (size_t)r.num_constraints * 32 + 3 * (size_t)NL29 * 4 * ((size_t)1 << q.log_n);
}

// workspace per signature in flight: A z, B z, C z packed (3 C x 32 B) + three working arrays (3 x 36 n B); the batch is
// cut into chunks that fit
hipError_t launch_qap_witness_map(const R1csDev &r, const QapDev &q, size_t batch, const uint64_t *witness,
                                  const uint64_t *instance, uint64_t *h, uint32_t *num_unsatisfied, void *workspace,
                                  size_t workspace_bytes, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    const int L = q.log_n;
    if (L < 12 || L > 30) return hipErrorInvalidValue;          // tiles of 2^10 elements, at least two passes; 32-bit indices
    const size_t n = (size_t)1 << L, per_sig = qap_workspace_bytes_per_signature(r, q);
    size_t chunk = workspace_bytes / per_sig;
    if (chunk == 0) return hipErrorInvalidValue;
    if (chunk > 16384) chunk = 16384;                            // grid.y = 3 x chunk <= 65535
    for (size_t lo = 0; lo < batch; lo += chunk) {
        const size_t cnt = batch - lo < chunk ? batch - lo : chunk;
        uint32_t *abc = (uint32_t *)workspace;
        uint32_t *work = abc + cnt * 3 * (size_t)r.num_constraints * 8;
        const uint64_t *wit = witness + lo * (size_t)r.num_witness * 4, *inst = instance + lo * (size_t)r.num_instance * 4;
        hipError_t e = launch_r1cs_check(r, cnt, wit, inst, num_unsatisfied ? num_unsatisfied + lo : nullptr, (uint64_t *)abc, st);
        if (e != hipSuccess) return e;
        NttPass p{};
        p.x = work;
        p.x_stride = NL29 * n;
        p.abc = abc;
        p.instance = (const uint32_t *)inst;
        p.num_constraints = r.num_constraints;
        p.num_instance = r.num_instance;
        p.scale = q.scale_in;
        p.scale0 = q.scale_in_a;
        // ifft + distribute_powers(g) (A z with an extra 2^5, which the a b product in R' = 2^261 arithmetic takes out again)
        if ((e = transform(XF_IFFT_FROM_PACKED, q, p, (unsigned)(3 * cnt), st)) != hipSuccess) return e;
        // fft: a, b, c on the coset (bit-reversed order)
        if ((e = transform(XF_FFT, q, p, (unsigned)(3 * cnt), st)) != hipSuccess) return e;
        // (a b - c) / Z on the coset, coset_ifft: one array per signature (in place on a's), packed into h at the end
        p.x_stride = 3 * NL29 * n;
        p.scale = q.scale_out;
        p.out_packed = (uint32_t *)(h + lo * n * 4);
        if ((e = transform(XF_IFFT_POINTWISE_TO_PACKED, q, p, (unsigned)cnt, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace frw
