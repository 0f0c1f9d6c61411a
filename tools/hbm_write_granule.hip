// hbm_write_granule.hip -- sustained pure-write bandwidth vs the size of the contiguous unit a workgroup writes and
// how units are handed out (static stride vs one atomic queue).  Interleaved rounds, medians.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wg tools/hbm_write_granule.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void fill(v4u *__restrict__ out, size_t gran16, size_t ngran, unsigned long long *counter, int dynamic)
{
    __shared__ unsigned long long next;
    v4u v = {blockIdx.x, threadIdx.x, 3, 4};
    size_t g = blockIdx.x;
    while (g < ngran) {
        v4u *o = out + g * gran16;
        for (size_t i = threadIdx.x; i < gran16; i += 256) o[i] = v;
        if (dynamic) {
            __syncthreads();
            if (threadIdx.x == 0) next = gridDim.x + atomicAdd(counter, 1ull);
            __syncthreads();
            g = next;
        } else g += gridDim.x;
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)20 << 30;
    v4u *buf; unsigned long long *cnt;
    CK(hipMalloc((void **)&buf, bytes));
    CK(hipMalloc((void **)&cnt, 64));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { size_t G; int grid; int dyn; std::vector<float> ms; };
    std::vector<Cfg> cfgs;
    for (size_t G : {(size_t)256 << 10, (size_t)512 << 10, (size_t)1 << 20, (size_t)2 << 20, (size_t)4 << 20, (size_t)5015168, (size_t)8 << 20, (size_t)32 << 20})
        for (int grid : {512, 768, 1024, 2048})
            for (int dyn : {0, 1}) cfgs.push_back({G, grid, dyn, {}});
    for (int round = 0; round < 7; round++)
        for (auto &c : cfgs) {
            const size_t ngran = bytes / c.G;
            CK(hipMemsetAsync(cnt, 0, 8, 0));
            CK(hipEventRecord(e0));
            for (int k = 0; k < 3; k++) {
                hipLaunchKernelGGL(fill, dim3(c.grid), dim3(256), 0, 0, buf, c.G / 16, ngran, cnt, c.dyn);
                CK(hipMemsetAsync(cnt, 0, 8, 0));
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round) c.ms.push_back(ms / 3);
        }
    for (auto &c : cfgs) {
        std::sort(c.ms.begin(), c.ms.end());
        const size_t ngran = bytes / c.G;
        printf("G=%9zu grid=%5d %s  median %.3f ms  %.1f GB/s\n", c.G, c.grid, c.dyn ? "queue " : "static", c.ms[c.ms.size() / 2],
               (double)ngran * c.G / c.ms[c.ms.size() / 2] / 1e6);
    }
    return 0;
}
