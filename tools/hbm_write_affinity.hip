// hbm_write_affinity.hip -- does HBM write bandwidth depend on WHICH XCD writes WHICH addresses?
// For a granule size G and a rotation r, a workgroup running on XCC x only writes granules g with (g % 8 + r) % 8 == x
// (granules drawn from a per-XCC queue).  If some (G, r) beats the unconstrained stream, the witness kernel could hand
// tiles to workgroups by address affinity.  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wa tools/hbm_write_affinity.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill_affinity(v4u *__restrict__ out, size_t gran16, size_t ngran, int rot,
                                                     unsigned long long *counters, int constrained)
{
    __shared__ unsigned long long next;
    const unsigned xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20) & 0xf;        // HW_REG_XCC_ID
    v4u v = {xcc, threadIdx.x, 3, 4};
    for (;;) {
        if (threadIdx.x == 0) next = atomicAdd(&counters[constrained ? xcc * 16 : 0], 1ull);
        __syncthreads();
        unsigned long long j = next;
        __syncthreads();
        size_t g = constrained ? j * 8 + ((xcc + 8 - rot) & 7) : j;
        if (g >= ngran) break;
        v4u *o = out + g * gran16;
        for (size_t i = threadIdx.x; i < gran16; i += 256) o[i] = v;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)16 << 30;
    v4u *buf; unsigned long long *cnt;
    CK(hipMalloc((void **)&buf, bytes));
    CK(hipMalloc((void **)&cnt, 8 * 16 * 8));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t grans[] = {4096, 65536, 1 << 20, 2 << 20, 16 << 20, 256 << 20};
    for (size_t G : grans) {
        const size_t ngran = bytes / G;
        for (int mode = -1; mode < 8; mode++) {
            float best = 1e9;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemsetAsync(cnt, 0, 8 * 16 * 8, 0));
                CK(hipEventRecord(e0));
                hipLaunchKernelGGL(fill_affinity, dim3(1024), dim3(256), 0, 0, buf, G / 16, ngran, mode < 0 ? 0 : mode, cnt, mode >= 0);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("G=%9zu  %s  %.3f ms  %.1f GB/s\n", G, mode < 0 ? "unconstrained" : (char[]){'r', 'o', 't', (char)('0' + mode), 0}, best,
                   (double)ngran * G / best / 1e6);
        }
    }
    return 0;
}
