// frw_qap.hip -- the R1CS -> QAP witness map on the device: h(X) = (A(X) B(X) - C(X)) / (X^n - 1) over BLS12-381 Fr.
//
// What a Groth16 prover does with the witness right after the hot path (examples/pok_sig.rs:30-47 calls
// Groth16::prove; ark-groth16 0.3.0 r1cs_to_qap.rs R1CStoQAP::witness_map):
//     a = A z (+ the instance values in rows C .. C+I), b = B z, c = C z       over the domain of n-th roots of unity
//     a, b, c <- coset_fft(ifft(.));   ab = a o b - c;   ab /= (g^n - 1);   h = coset_ifft(ab)
// Here, with the transforms arranged so that no permutation pass is needed:
//     r1cs_check_kernel (qap mode) writes row i of A z, B z, C z at position bitrev(i)
//     inverse transform, decimation in time (bit-reversed in, natural out), store fused with x g^k / n
//     forward transform, decimation in frequency (natural in, bit-reversed out)         -> values on the coset, bitrev order
//     inverse transform, decimation in time, load fused with a b - c, store fused with x g^-k / (n (g^n - 1))    -> h
// Every transform is three passes over HBM (6 + 6 + (log n - 12) butterfly stages), a workgroup taking a tile of
// 2^T rows x 16 contiguous elements (512-byte runs) through LDS.  The field arithmetic is exact, so the result is the
// same element for element as any other schedule's -- what the parity tests rely on.
//
// Cost per signature (n = 2^18): 7 transforms x 9 n + 5 n Montgomery products (~17 M) -- the kernel is bound by the
// integer multiplier (v_mad_u64_u32), not by HBM (~0.4 GB of traffic).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frw_device.h"
#include "frw_fr.h"

namespace frw {

constexpr int QAP_COLS = 16;            // contiguous elements per tile row
constexpr int QAP_MAX_T = 6;            // butterfly stages per pass
constexpr int QAP_TILE = (1 << QAP_MAX_T) * QAP_COLS;

struct NttPass {
    const uint32_t *x;          // [arrays][...]: source
    uint32_t *out;              // destination (== x: in place)
    size_t x_stride, out_stride;      // 32-bit words between consecutive arrays
    const uint32_t *tw;         // root^k, k < n/2
    const uint32_t *scale;      // store_op 1: factor per (natural) index
    int L, sh, T;               // log n; lowest index bit this pass transforms; stages in this pass
    int load_op;                // 0: x[i];  1: x[i] * x[n + i] - x[2 n + i]
    int store_op;               // 0: plain; 1: times scale[i]
};

__device__ __forceinline__ Fr8 lds_get(const uint4 *tile, int slot)
{
    const uint4 a = tile[2 * slot], b = tile[2 * slot + 1];
    Fr8 r;
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}
__device__ __forceinline__ void lds_put(uint4 *tile, int slot, const Fr8 &v)
{
    tile[2 * slot] = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    tile[2 * slot + 1] = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// One pass = T consecutive radix-2 stages on the index bits [sh, sh + T).  DIF: stages from the top bit down, butterfly
// (u + v, (u - v) w); otherwise decimation in time: stages from the bottom bit up, butterfly (u + w v, u - w v).
// SH0 (sh == 0): the tile is 16 2^T contiguous elements and the tile-linear order is column-major (row = low bits);
// otherwise the 16 columns are the index bits [0, 4) and a tile row is a 512-byte run.
template <bool DIF, bool SH0>
__global__ __launch_bounds__(BLOCK) void ntt_pass_kernel(const NttPass p)
{
    __shared__ uint4 tile[QAP_TILE * 2];
    const int tid = threadIdx.x;
    const int R = 1 << p.T, elems = R * QAP_COLS;
    const uint32_t tileid = blockIdx.x;
    const size_t n = (size_t)1 << p.L;
    const uint32_t lowmid = SH0 ? 0u : (tileid & ((1u << (p.sh - 4)) - 1u)) << 4;
    const uint32_t high = SH0 ? tileid * (uint32_t)elems : (tileid >> (p.sh - 4)) << (p.sh + p.T);
    auto gidx = [&](int lin) -> uint32_t {
        return SH0 ? high + (uint32_t)lin : high | ((uint32_t)(lin >> 4) << p.sh) | lowmid | (uint32_t)(lin & 15);
    };
    const uint32_t *src = p.x + (size_t)blockIdx.y * p.x_stride;
    for (int lin = tid; lin < elems; lin += BLOCK) {
        const uint32_t g = gidx(lin);
        Fr8 v = fr_load(src + (size_t)g * 8);
        if (p.load_op == 1) v = fr_sub(fr_mul(v, fr_load(src + (n + g) * 8)), fr_load(src + (2 * n + g) * 8));
        lds_put(tile, lin, v);
    }
    __syncthreads();
    for (int ti = 0; ti < p.T; ti++) {
        const int t = DIF ? p.T - ti : ti + 1, s = p.sh + t, hr = 1 << (t - 1);
        for (int k = tid; k < elems / 2; k += BLOCK) {
            const int b = SH0 ? k & (R / 2 - 1) : k >> 4, c = SH0 ? k >> (p.T - 1) : k & 15;
            const int r_lo = b & (hr - 1), r = ((b >> (t - 1)) << t) | r_lo;
            const int s0 = SH0 ? c * R + r : r * QAP_COLS + c, s1 = s0 + (SH0 ? hr : hr * QAP_COLS);
            const uint32_t j = SH0 ? (uint32_t)r_lo : ((uint32_t)r_lo << p.sh) | lowmid | (uint32_t)c;
            Fr8 u = lds_get(tile, s0), v = lds_get(tile, s1);
            if (s == 1) {                                   // the twiddle is one
                const Fr8 d = fr_sub(u, v);
                u = fr_add(u, v);
                v = d;
            } else {
                const Fr8 w = fr_load(p.tw + ((size_t)j << (p.L - s)) * 8);
                if (DIF) {
                    const Fr8 d = fr_sub(u, v);
                    u = fr_add(u, v);
                    v = fr_mul(d, w);
                } else {
                    v = fr_mul(v, w);
                    const Fr8 d = fr_sub(u, v);
                    u = fr_add(u, v);
                    v = d;
                }
            }
            lds_put(tile, s0, u);
            lds_put(tile, s1, v);
        }
        __syncthreads();
    }
    uint32_t *dst = p.out + (size_t)blockIdx.y * p.out_stride;
    for (int lin = tid; lin < elems; lin += BLOCK) {
        const uint32_t g = gidx(lin);
        Fr8 v = lds_get(tile, lin);
        if (p.store_op == 1) v = fr_mul(v, fr_load(p.scale + (size_t)g * 8));
        fr_store(dst + (size_t)g * 8, v);
    }
}

// rows C .. n of the three arrays: a[C + j] = z_j for the I instance variables (the constant one first), zero elsewhere
__global__ __launch_bounds__(BLOCK) void qap_pad_kernel(uint32_t *__restrict__ abc, int L, uint32_t num_constraints,
                                                        uint32_t num_instance, const uint32_t *__restrict__ instance)
{
    const size_t n = (size_t)1 << L, sig = blockIdx.y;
    uint32_t *o = abc + sig * 3 * n * 8;
    const uint32_t *inst = instance + sig * (size_t)num_instance * 8;
    const uint4 z = make_uint4(0, 0, 0, 0);
    for (size_t row = num_constraints + (size_t)blockIdx.x * BLOCK + threadIdx.x; row < n; row += (size_t)gridDim.x * BLOCK) {
        const size_t pos = __brev((uint32_t)row) >> (32 - L);
        uint4 *a = (uint4 *)(o + pos * 8), *b = (uint4 *)(o + (n + pos) * 8), *c = (uint4 *)(o + (2 * n + pos) * 8);
        const size_t j = row - num_constraints;
        if (j < num_instance) {
            a[0] = *(const uint4 *)(inst + j * 8);
            a[1] = *(const uint4 *)(inst + j * 8 + 4);
        } else {
            a[0] = z; a[1] = z;
        }
        b[0] = z; b[1] = z; c[0] = z; c[1] = z;
    }
}

namespace {
// passes of one transform over `arrays` arrays; first_load / last_store fuse the element-wise steps around it
hipError_t transform(bool dif, const QapDev &q, const uint32_t *x, size_t x_stride, uint32_t *out, size_t out_stride,
                     unsigned arrays, bool fused_load, const uint32_t *last_scale, hipStream_t st)
{
    const int L = q.log_n;
    int shs[8], ts[8], np = 0;
    for (int sh = 0; sh < L; sh += QAP_MAX_T) { shs[np] = sh; ts[np] = L - sh < QAP_MAX_T ? L - sh : QAP_MAX_T; np++; }
    for (int i = 0; i < np; i++) {
        const int k = dif ? np - 1 - i : i;
        NttPass p;
        p.x = i == 0 ? x : out;
        p.x_stride = i == 0 ? x_stride : out_stride;
        p.out = out;
        p.out_stride = out_stride;
        p.tw = dif ? q.tw_fwd : q.tw_inv;
        p.scale = last_scale;
        p.L = L; p.sh = shs[k]; p.T = ts[k];
        p.load_op = i == 0 && fused_load ? 1 : 0;
        p.store_op = i == np - 1 && last_scale ? 1 : 0;
        const unsigned tiles = (unsigned)(((size_t)1 << L) >> (p.T + 4));
        const dim3 grid(tiles, arrays);
        if (dif) {
            if (p.sh == 0) hipLaunchKernelGGL((ntt_pass_kernel<true, true>), grid, dim3(BLOCK), 0, st, p);
            else hipLaunchKernelGGL((ntt_pass_kernel<true, false>), grid, dim3(BLOCK), 0, st, p);
        } else {
            if (p.sh == 0) hipLaunchKernelGGL((ntt_pass_kernel<false, true>), grid, dim3(BLOCK), 0, st, p);
            else hipLaunchKernelGGL((ntt_pass_kernel<false, false>), grid, dim3(BLOCK), 0, st, p);
        }
        const hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}
}  // namespace

// workspace: 3 n field elements per signature in flight; the batch is cut into chunks that fit
hipError_t launch_qap_witness_map(const R1csDev &r, const QapDev &q, size_t batch, const uint64_t *witness,
                                  const uint64_t *instance, uint64_t *h, uint32_t *num_unsatisfied, void *workspace,
                                  size_t workspace_bytes, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    const int L = q.log_n;
    if (L < 10 || L > 30) return hipErrorInvalidValue;          // a tile is 2^10 elements; 32-bit element indices
    const size_t n = (size_t)1 << L, per_sig = 3 * n * 32;
    size_t chunk = workspace_bytes / per_sig;
    if (chunk == 0) return hipErrorInvalidValue;
    if (chunk > 16384) chunk = 16384;                            // grid.y = 3 x chunk <= 65535
    uint32_t *ws = (uint32_t *)workspace;
    for (size_t lo = 0; lo < batch; lo += chunk) {
        const size_t cnt = batch - lo < chunk ? batch - lo : chunk;
        const uint64_t *wit = witness + lo * (size_t)r.num_witness * 4, *inst = instance + lo * (size_t)r.num_instance * 4;
        hipError_t e = launch_r1cs_check(r, cnt, wit, inst, num_unsatisfied ? num_unsatisfied + lo : nullptr, (uint64_t *)ws, st, L);
        if (e != hipSuccess) return e;
        const size_t pad_rows = n - r.num_constraints;
        if (pad_rows) {
            const unsigned gx = (unsigned)((pad_rows + BLOCK - 1) / BLOCK);
            hipLaunchKernelGGL(qap_pad_kernel, dim3(gx > 256 ? 256 : gx, (unsigned)cnt), dim3(BLOCK), 0, st, ws, L,
                               r.num_constraints, r.num_instance, (const uint32_t *)inst);
            if ((e = hipGetLastError()) != hipSuccess) return e;
        }
        // ifft + distribute_powers(g), then fft: a, b, c on the coset (bit-reversed order)
        if ((e = transform(false, q, ws, n * 8, ws, n * 8, (unsigned)(3 * cnt), false, q.scale_in, st)) != hipSuccess) return e;
        if ((e = transform(true, q, ws, n * 8, ws, n * 8, (unsigned)(3 * cnt), false, nullptr, st)) != hipSuccess) return e;
        // (a b - c) / Z on the coset, coset_ifft
        uint32_t *hh = (uint32_t *)(h + lo * n * 4);
        if ((e = transform(false, q, ws, 3 * n * 8, hh, n * 8, (unsigned)cnt, true, q.scale_out, st)) != hipSuccess) return e;
    }
    return hipSuccess;
}

}  // namespace frw
