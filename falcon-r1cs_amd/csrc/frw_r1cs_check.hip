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

namespace frw {

#define FRW_P32 {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u}
#define FRW_R32 {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u}

struct Fr8 { uint32_t l[8]; };

__device__ __forceinline__ Fr8 fr_load(const uint32_t *p)
{
    Fr8 r;
    const uint4 a = *(const uint4 *)p, b = *(const uint4 *)(p + 4);
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

// r = (a + b) mod p, inputs < p
__device__ __forceinline__ Fr8 fr_add(const Fr8 &a, const Fr8 &b)
{
    constexpr uint32_t P[8] = FRW_P32;
    Fr8 s, d;
    uint32_t c = 0, bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)a.l[i] + b.l[i] + c;
        s.l[i] = (uint32_t)x;
        c = (uint32_t)(x >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)s.l[i] - P[i] - bw;
        d.l[i] = (uint32_t)x;
        bw = (uint32_t)(x >> 63);
    }
    const bool ge = c || !bw;              // p < 2^255, so c is never set; kept for clarity
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = ge ? d.l[i] : s.l[i];
    return s;
}

// Montgomery product a * b / 2^256 mod p (CIOS, 32-bit limbs; -p^-1 mod 2^32 = 0xffffffff)
__device__ __forceinline__ Fr8 fr_mul(const Fr8 &a, const Fr8 &b)
{
    constexpr uint32_t P[8] = FRW_P32;
    uint32_t T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t acc;
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc = (uint64_t)a.l[i] * b.l[j] + T[j] + c;
            T[j] = (uint32_t)acc;
            c = (uint32_t)(acc >> 32);
        }
        acc = (uint64_t)T[8] + c;
        T[8] = (uint32_t)acc;
        const uint32_t t9 = (uint32_t)(acc >> 32);
        const uint32_t m = 0u - T[0];
        acc = (uint64_t)m * P[0] + T[0];
        c = (uint32_t)(acc >> 32);
#pragma unroll
        for (int j = 1; j < 8; j++) {
            acc = (uint64_t)m * P[j] + T[j] + c;
            T[j - 1] = (uint32_t)acc;
            c = (uint32_t)(acc >> 32);
        }
        acc = (uint64_t)T[8] + c;
        T[7] = (uint32_t)acc;
        T[8] = t9 + (uint32_t)(acc >> 32);
    }
    Fr8 r, d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)T[i] - P[i] - bw;
        d.l[i] = (uint32_t)x;
        bw = (uint32_t)(x >> 63);
    }
    const bool ge = T[8] || !bw;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = ge ? d.l[i] : T[i];
    return r;
}

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

__device__ __forceinline__ void fr_store(uint32_t *p, const Fr8 &v)
{
    *(uint4 *)p = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    *(uint4 *)(p + 4) = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
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
    if ((threadIdx.x & (WAVE - 1)) == 0 && bad) atomicAdd(&num_unsatisfied[sig], bad);
}

hipError_t launch_r1cs_check(const R1csDev &r, size_t batch, const uint64_t *witness, const uint64_t *instance,
                             uint32_t *num_unsatisfied, uint64_t *abc, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;
    hipError_t e = hipMemsetAsync(num_unsatisfied, 0, batch * sizeof(uint32_t), st);
    if (e != hipSuccess) return e;
    // enough workgroups per signature that the dense ladder rows (first in `order`) spread over many waves
    const unsigned gx = (r.num_constraints + BLOCK - 1) / BLOCK;
    hipLaunchKernelGGL(r1cs_check_kernel, dim3(gx > 64 ? 64 : gx, (unsigned)batch), dim3(BLOCK), 0, st, r, batch,
                       (const uint32_t *)witness, (const uint32_t *)instance, num_unsatisfied, (uint32_t *)abc);
    return hipGetLastError();
}

}  // namespace frw
