// hbm_write_ceiling.hip -- what a pure 16-B-per-lane write stream reaches on this chip (the ceiling the witness
// kernel is priced against, next to the 8 TB/s spec peak).  Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wc tools/hbm_write_ceiling.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int NT>
__global__ __launch_bounds__(256) void fill(v4u *__restrict__ out, size_t n16, uint32_t seed)
{
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    v4u v = {seed, seed ^ threadIdx.x, 3, 4};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += stride) {
        if (NT) __builtin_nontemporal_store(v, &out[i]); else out[i] = v;
    }
}

// block-contiguous variant: each workgroup owns a contiguous slab (like one signature's witness)
template <int NT>
__global__ __launch_bounds__(256) void fill_slab(v4u *__restrict__ out, size_t slab16, size_t nslabs, uint32_t seed)
{
    v4u v = {seed, seed ^ threadIdx.x, 3, 4};
    for (size_t s = blockIdx.x; s < nslabs; s += gridDim.x) {
        v4u *o = out + s * slab16;
        for (size_t i = threadIdx.x; i < slab16; i += 256) {
            if (NT) __builtin_nontemporal_store(v, &o[i]); else o[i] = v;
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)20 << 30;
    const size_t n16 = bytes / 16;
    v4u *buf;
    CK(hipMalloc((void **)&buf, bytes));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {512, 1024, 2048, 4096, 16384};
    for (int nt = 0; nt < 2; nt++)
        for (int g : grids) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                if (nt) hipLaunchKernelGGL(fill<1>, dim3(g), dim3(256), 0, 0, buf, n16, rep);
                else hipLaunchKernelGGL(fill<0>, dim3(g), dim3(256), 0, 0, buf, n16, rep);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("grid-stride  nt=%d grid=%6d  %.3f ms  %.1f GB/s\n", nt, g, best, bytes / best / 1e6);
        }
    const size_t slab16 = 5015168 / 16;            // one Falcon-1024 witness
    const size_t nslabs = n16 / slab16;
    for (int nt = 0; nt < 2; nt++)
        for (int g : {512, 768, 1024, 2048}) {
            float best = 1e9;
            for (int rep = 0; rep < 4; rep++) {
                CK(hipEventRecord(e0));
                if (nt) hipLaunchKernelGGL(fill_slab<1>, dim3(g), dim3(256), 0, 0, buf, slab16, nslabs, rep);
                else hipLaunchKernelGGL(fill_slab<0>, dim3(g), dim3(256), 0, 0, buf, slab16, nslabs, rep);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms < best) best = ms;
            }
            printf("per-wg slabs nt=%d grid=%6d  %.3f ms  %.1f GB/s\n", nt, g, best, nslabs * slab16 * 16 / best / 1e6);
        }
    return 0;
}
