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
#include "frw_fr.h"

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
__global__ __launch_bounds__(BLOCK) void r1cs_check_kernel(R1csDev r, size_t batch, const uint32_t *__restrict__ witness,
                                                           const uint32_t *__restrict__ instance,
                                                           unsigned int *__restrict__ num_unsatisfied,
                                                           uint32_t *__restrict__ abc)
{
    const size_t sig = blockIdx.y;
    if (sig >= batch) return;
    const uint32_t *wit = witness + sig * (size_t)r.num_witness * 8;
    const uint32_t *inst = instance + sig * (size_t)r.num_instance * 8;
    unsigned bad = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < r.num_constraints; i += gridDim.x * BLOCK) {
        const uint32_t row = r.order[i];
        const Fr8 az = row_dot(r.a, row, wit, inst, r.num_instance);
        const Fr8 bz = row_dot(r.b, row, wit, inst, r.num_instance);
        const Fr8 cz = row_dot(r.c, row, wit, inst, r.num_instance);
        if (abc) {
            uint32_t *o = abc + sig * (size_t)3 * r.num_constraints * 8;
            fr_store(o + (size_t)row * 8, az);
            fr_store(o + ((size_t)r.num_constraints + row) * 8, bz);
            fr_store(o + ((size_t)2 * r.num_constraints + row) * 8, cz);
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

hipError_t launch_r1cs_check(const R1csDev &r, size_t batch, const uint64_t *witness, const uint64_t *instance,
                             uint32_t *num_unsatisfied, uint64_t *abc, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;
    if (num_unsatisfied) {
        hipError_t e = hipMemsetAsync(num_unsatisfied, 0, batch * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
    }
    // enough workgroups per signature that the dense ladder rows (first in `order`) spread over many waves
    const unsigned gx = (r.num_constraints + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(r1cs_check_kernel, dim3(gx > 64 ? 64 : gx, (unsigned)batch), dim3(BLOCK), 0, st, r, batch,
                       (const uint32_t *)witness, (const uint32_t *)instance, num_unsatisfied, (uint32_t *)abc);
    return hipGetLastError();
}

}  // namespace frw
