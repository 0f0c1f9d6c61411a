// mfma_twiddle_bound.hip -- bounded experiment: can the data x CONSTANT products of the QAP transforms
// (every butterfly multiplies by a twiddle that all signatures of a batch share) run on the matrix cores?
//
// The formulation: 7-bit limbs (int8 MFMA operands are signed; 37 limbs cover 259 bits), X = [limbs x signatures], and each of the
// three constant products of a Montgomery multiplication is a Toeplitz matrix times X:
//     T = Toep(w) X (74 columns),   m = Toep(p') T_lo mod 2^259 (37 columns),   U = T + Toep(p) m   (74 columns, upper half / 2^259)
// = 32 instructions v_mfma_i32_32x32x32_i8 per 64 products (K padded to 64, M to 96 / 64).  Between them the 20-bit column sums
// have to become 7-bit limbs again -- on the vector ALU, which is the unit the existing path (frw_fr29.h, 230 vector
// instructions per product) is bound by.  This program measures the two halves separately, at four waves per SIMD each, which
// bounds the combined kernel from above (the two pipes can overlap at best):
//   (a) the issue rate of v_mfma_i32_32x32x32_i8;
//   (b) the vector-ALU work per product that remains: three carry propagations over 37 columns, the 74 additions of U, packing
//       the limbs four to a register for the next MFMA, one lane-half exchange per four columns (the accumulator layout puts
//       a signature's columns 4 k .. 4 k + 3 in one lane and the next four in lane + 32).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/mfma_twiddle_bound.hip -o /tmp/mfma_bound && /tmp/mfma_bound
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

constexpr int ITER = 2048;

__global__ __launch_bounds__(256) void mfma_rate(int *out, int seed)
{
    v4i a = {seed, seed + 1, seed + 2, seed + 3}, b = {seed * 3, seed * 5, seed * 7, seed * 9};
    v16i c0 = {}, c1 = {}, c2 = {}, c3 = {};
    for (int it = 0; it < ITER; it++) {
        c0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_i32_32x32x32_i8(a, b, c3, 0, 0, 0);
    }
    int s = 0;
    for (int k = 0; k < 16; k++) s += c0[k] + c1[k] + c2[k] + c3[k];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the vector-ALU remainder of ONE product, per signature; a lane owns half of a signature's columns (groups of four), so the
// per-lane work below is for 2 x 37 / 2 columns per step and a lane pair completes one product: counted as half a product per lane
__device__ __forceinline__ void carry_half(int (&col)[20], int &carry_io)
{
    // 5 groups of 4 columns (the lane's share of 37-40 columns): propagate inside a group, hand the carry to the other lane half
#pragma unroll
    for (int g = 0; g < 5; g++) {
        int c = carry_io;
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int t = col[4 * g + k] + c;
            col[4 * g + k] = t & 127;
            c = t >> 7;
        }
        carry_io = __builtin_amdgcn_ds_swizzle(c, 0x401f);      // stand-in for the exchange with lane ^ 32 (one cross-lane op per group)
    }
}
__global__ __launch_bounds__(256) void valu_remainder(int *out, int seed)
{
    int t[40], m[20], acc = 0;
    for (int k = 0; k < 40; k++) t[k] = (seed * (k + 3) + threadIdx.x * 977) & 0xfffff;          // 20-bit column sums of Toep(w) X
    for (int it = 0; it < ITER / 8; it++) {
        int lo[20], hi[20], carry = it;
#pragma unroll
        for (int k = 0; k < 20; k++) { lo[k] = t[k]; hi[k] = t[20 + k]; }
        carry_half(lo, carry);                                     // T_lo -> limbs
        int packed[5];
#pragma unroll
        for (int g = 0; g < 5; g++) packed[g] = lo[4 * g] | lo[4 * g + 1] << 8 | lo[4 * g + 2] << 16 | lo[4 * g + 3] << 24;   // MFMA operand
#pragma unroll
        for (int k = 0; k < 20; k++) m[k] = (packed[k >> 2] >> (8 * (k & 3))) * 5 + t[k];          // stand-in for the MFMA result Toep(p') T_lo
        carry = 0;
        carry_half(m, carry);                                      // m -> limbs (mod 2^259)
#pragma unroll
        for (int g = 0; g < 5; g++) packed[g] ^= m[4 * g] | m[4 * g + 1] << 8 | m[4 * g + 2] << 16 | m[4 * g + 3] << 24;
#pragma unroll
        for (int k = 0; k < 20; k++) { lo[k] += m[k] * 3; hi[k] += packed[k >> 2] & 0xffff; }      // U = T + Toep(p) m: 2 x 20 additions
        carry_half(lo, carry);                                     // the low half only to get its carry out
        carry_half(hi, carry);                                     // U_hi -> limbs: the product
#pragma unroll
        for (int g = 0; g < 5; g++) packed[g] = hi[4 * g] | hi[4 * g + 1] << 8 | hi[4 * g + 2] << 16 | hi[4 * g + 3] << 24;
#pragma unroll
        for (int g = 0; g < 5; g++) acc += packed[g];
#pragma unroll
        for (int k = 0; k < 40; k++) t[k] = (t[k] + acc + k) & 0xfffff;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <class K>
static float run(K kernel, int *buf, int grid)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, buf, 1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(256), 0, 0, buf, 2);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, grid = cus * 4;            // 4 waves per SIMD
    int *buf;
    hipMalloc(&buf, (size_t)grid * 256 * 4);
    const float ms_a = run(mfma_rate, buf, grid);
    const double mfma_per_simd_us = 4.0 * ITER * 4 / (ms_a * 1e3);    // instructions per SIMD per microsecond
    const double simds = 4.0 * cus;
    // 32 MFMA per 64 products
    const double g_products_mfma = mfma_per_simd_us * 1e6 * simds / 32.0 * 64.0 / 1e9;
    const float ms_b = run(valu_remainder, buf, grid);
    // per lane and iteration: half a product (the other lane half does the rest)
    const double products = (double)grid * 256 * (ITER / 8) * 0.5;
    const double g_products_valu = products / (ms_b * 1e-3) / 1e9;
    printf("%s, %d CUs\n", p.name, cus);
    printf("v_mfma_i32_32x32x32_i8: %.3f ms -> %.1f instructions/SIMD/us = %.2f dense int8 PetaOP/s over the chip\n", ms_a, mfma_per_simd_us,
           mfma_per_simd_us * 1e6 * simds * 65536.0 / 1e15);
    printf("matrix-core side alone (32 MFMA per 64 products):            %8.1f G products/s\n", g_products_mfma);
    printf("vector-ALU remainder alone (carries, additions, packing):     %8.1f G products/s\n", g_products_valu);
    printf("upper bound of the combined kernel (perfect overlap) = min:   %8.1f G products/s\n",
           g_products_mfma < g_products_valu ? g_products_mfma : g_products_valu);
    return 0;
}
