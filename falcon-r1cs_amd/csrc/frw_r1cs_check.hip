// frw_r1cs_check.hip -- batch R1CS satisfaction check on the device.
//
// The matrices are the ones the host mirror emits from the gadget definitions (host/frw_host.hpp, to_matrices():
// every symbolic LC inlined) -- NOT the closed form the witness kernels implement -- so this is the reference's
// `assert!(cs.is_satisfied())` (circuits/falcon_ntt.rs:159) run against an independently derived constraint system,
// for every signature of a full-size launch, where the witnesses lie (HBM).
//
// One thread evaluates one constraint row of one signature: three sparse dot products over BLS12-381 Fr in Montgomery
// form and one field multiplication.  Rows are visited in order of decreasing length (host-computed permutation), so
// the 64 rows of a wavefront have similar length (the 2N ladder rows have N+1 terms, everything else a handful).
// ALU-bound (a 256-bit Montgomery product per term); a verification utility, not a throughput path.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frw_device.h"
#include "frw_fr29.h"

namespace frw {

__device__ __forceinline__ Fr8 row_dot(const R1csMatrixDev &m, uint32_t row, const uint32_t *__restrict__ wit,
                                       const uint32_t *__restrict__ inst, uint32_t num_instance)
{
    Fr8 acc;
#pragma unroll
    for (int i = 0; i < 8; i++) acc.l[i] = 0;
    const uint64_t lo = m.row_ptr[row], hi = m.row_ptr[row + 1];
    for (uint64_t t = lo; t < hi; t++) {
        const uint32_t col = m.col[t];
        const Fr8 z = fr_load(col < num_instance ? inst + (size_t)col * 8 : wit + (size_t)(col - num_instance) * 8);
        acc = fr_add(acc, fr_mul(fr_load(m.val + t * 8), z));
    }
    return acc;
}

// abc != nullptr: also write Az, Bz, Cz (Montgomery form) as [batch][3][num_constraints][8 x u32] -- the inputs of a
// prover's QAP witness map (what ark-groth16 computes on the CPU right after generate_constraints).
// LONG_DONE: the long rows' products are already in abc (r1cs_long_rows_kernel ran first) and are read back instead of
// recomputed; rows are then visited in constraint order (neighbouring rows touch neighbouring witnesses).
template <bool LONG_DONE>
__global__ __launch_bounds__(BLOCK) void r1cs_check_kernel(R1csDev r, size_t batch, const uint32_t *__restrict__ witness,
                                                           const uint32_t *__restrict__ instance,
                                                           unsigned int *__restrict__ num_unsatisfied,
                                                           uint32_t *__restrict__ abc)
{
    const size_t sig = blockIdx.y;
    if (sig >= batch) return;
    const uint32_t *wit = witness + sig * (size_t)r.num_witness * 8;
    const uint32_t *inst = instance + sig * (size_t)r.num_instance * 8;
    uint32_t *o = abc ? abc + sig * (size_t)3 * r.num_constraints * 8 : nullptr;
    unsigned bad = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < r.num_constraints; i += gridDim.x * BLOCK) {
        const uint32_t row = LONG_DONE ? i : r.order[i];
        const uint32_t mask = LONG_DONE ? r.long_mask[row] : 0u;
        const Fr8 az = mask & 1u ? fr_load(o + (size_t)row * 8) : row_dot(r.a, row, wit, inst, r.num_instance);
        const Fr8 bz = mask & 2u ? fr_load(o + ((size_t)r.num_constraints + row) * 8) : row_dot(r.b, row, wit, inst, r.num_instance);
        const Fr8 cz = mask & 4u ? fr_load(o + ((size_t)2 * r.num_constraints + row) * 8) : row_dot(r.c, row, wit, inst, r.num_instance);
        if (o) {
            if (!(mask & 1u)) fr_store(o + (size_t)row * 8, az);
            if (!(mask & 2u)) fr_store(o + ((size_t)r.num_constraints + row) * 8, bz);
            if (!(mask & 4u)) fr_store(o + ((size_t)2 * r.num_constraints + row) * 8, cz);
        }
        const Fr8 ab = fr_mul(az, bz);              // (Az R)(Bz R)/R = Az Bz R
        bool eq = true;
#pragma unroll
        for (int k = 0; k < 8; k++) eq &= ab.l[k] == cz.l[k];
        bad += eq ? 0u : 1u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && bad && num_unsatisfied) atomicAdd(&num_unsatisfied[sig], bad);
}

// One wavefront per (long row, group of LONG_SIGS signatures): lane l multiplies terms l, l + 64, ... (coefficient loaded
// once, used for every signature of the group), the 64 partial sums are added across the wavefront, lane 0 stores.
// Products are f29_mul(z R, c R') = z c R, sums kept < 2 p.
constexpr int LONG_SIGS = 4;
__global__ __launch_bounds__(WAVE) void r1cs_long_rows_kernel(R1csDev r, size_t batch, const uint32_t *__restrict__ witness,
                                                              const uint32_t *__restrict__ instance, uint32_t *__restrict__ abc)
{
    const R1csLongRow d = r.long_rows[blockIdx.x];
    const int lane = threadIdx.x;
    const size_t sig0 = (size_t)blockIdx.y * LONG_SIGS;
    F29 acc[LONG_SIGS];
#pragma unroll
    for (int s = 0; s < LONG_SIGS; s++)
#pragma unroll
        for (int k = 0; k < NL29; k++) acc[s].l[k] = 0;
    for (uint32_t ch = 0; ch < d.num_chunks; ch++) {
        const size_t chunk = (size_t)d.first_chunk + ch;
        const uint32_t col = r.long_col[chunk * WAVE + lane];
        F29 c;
#pragma unroll
        for (int k = 0; k < NL29; k++) c.l[k] = r.long_coef[(chunk * NL29 + k) * WAVE + lane];
#pragma unroll
        for (int s = 0; s < LONG_SIGS; s++) {
            const size_t sig = sig0 + s < batch ? sig0 + s : batch - 1;       // a ragged last group repeats the last signature
            const uint32_t *zp = col < r.num_instance ? instance + (sig * r.num_instance + col) * 8
                                                       : witness + (sig * r.num_witness + (col - r.num_instance)) * 8;
            const F29 z = f29_unpack(fr_load(zp));
            acc[s] = f29_reduce_4p(f29_add(acc[s], f29_mul(z, c)));
        }
    }
#pragma unroll
    for (int s = 0; s < LONG_SIGS; s++) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            F29 other;
#pragma unroll
            for (int k = 0; k < NL29; k++) other.l[k] = (uint32_t)__shfl_xor((int)acc[s].l[k], off, WAVE);
            acc[s] = f29_reduce_4p(f29_add(acc[s], other));
        }
        if (lane == 0 && sig0 + s < batch)
            fr_store(abc + (((sig0 + s) * 3 + d.matrix) * (size_t)r.num_constraints + d.row) * 8, f29_pack(f29_canonical(acc[s])));
    }
}

hipError_t launch_r1cs_check(const R1csDev &r, size_t batch, const uint64_t *witness, const uint64_t *instance,
                             uint32_t *num_unsatisfied, uint64_t *abc, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;
    if (num_unsatisfied) {
        hipError_t e = hipMemsetAsync(num_unsatisfied, 0, batch * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
    }
    const unsigned gx = (r.num_constraints + BLOCK - 1) / BLOCK;
    if (abc && r.num_long) {
        hipLaunchKernelGGL(r1cs_long_rows_kernel, dim3(r.num_long, (unsigned)((batch + LONG_SIGS - 1) / LONG_SIGS)), dim3(WAVE), 0, st,
                           r, batch, (const uint32_t *)witness, (const uint32_t *)instance, (uint32_t *)abc);
        hipLaunchKernelGGL(r1cs_check_kernel<true>, dim3(gx > 64 ? 64 : gx, (unsigned)batch), dim3(BLOCK), 0, st, r, batch,
                           (const uint32_t *)witness, (const uint32_t *)instance, num_unsatisfied, (uint32_t *)abc);
        return hipGetLastError();
    }
    // check only (no buffer to park the long rows' products in): one thread per row, longest rows first, so that the dense
    // ladder rows spread over many waves
    hipLaunchKernelGGL(r1cs_check_kernel<false>, dim3(gx > 64 ? 64 : gx, (unsigned)batch), dim3(BLOCK), 0, st, r, batch,
                       (const uint32_t *)witness, (const uint32_t *)instance, num_unsatisfied, (uint32_t *)abc);
    return hipGetLastError();
}

}  // namespace frw
