// hbm_write_alloc.hip -- does the write stream depend on how the buffer was allocated?  hipMalloc (default, cached in L2),
// hipExtMallocWithFlags(hipDeviceMallocUncached), hipExtMallocWithFlags(hipDeviceMallocFinegrained); and on the data
// (constant vs per-lane varying vs all zero).  768 workgroups, units of one Falcon-1024 witness, 32,256 units per launch.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wa tools/hbm_write_alloc.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr size_t UNIT16 = 5015168 / 16;

template <int DATA>
__global__ __launch_bounds__(256) void fill(v4u *__restrict__ out, size_t nunits)
{
    v4u v = {0, 0, 0, 0};
    if (DATA == 1) v = v4u{0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau};
    for (size_t u = blockIdx.x; u < nunits; u += gridDim.x) {
        v4u *o = out + u * UNIT16;
        for (size_t i = threadIdx.x; i < UNIT16; i += 256) {
            if (DATA == 2) { v.x = (uint32_t)(i * 2654435761u) ^ (uint32_t)u; v.y = v.x * 40503u; v.z = ~v.x; v.w = v.y ^ 0x9e3779b9u; }
            o[i] = v;
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const size_t nunits = 32256, bytes = nunits * UNIT16 * 16;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const char *anames[3] = {"hipMalloc", "uncached", "finegrained"};
    const char *dnames[3] = {"all zero", "constant (Montgomery one)", "pseudo-random per chunk"};
    for (int a = 0; a < 3; a++) {
        v4u *buf = nullptr;
        hipError_t e = a == 0 ? hipMalloc((void **)&buf, bytes)
                     : hipExtMallocWithFlags((void **)&buf, bytes, a == 1 ? hipDeviceMallocUncached : hipDeviceMallocFinegrained);
        if (e != hipSuccess) { printf("%s: allocation failed (%s)\n", anames[a], hipGetErrorString(e)); continue; }
        for (int d = 0; d < 3; d++) {
            std::vector<float> ms;
            for (int round = 0; round < 4; round++) {
                CK(hipEventRecord(e0));
                for (int k = 0; k < 2; k++) {
                    if (d == 0) hipLaunchKernelGGL(fill<0>, dim3(768), dim3(256), 0, 0, buf, nunits);
                    else if (d == 1) hipLaunchKernelGGL(fill<1>, dim3(768), dim3(256), 0, 0, buf, nunits);
                    else hipLaunchKernelGGL(fill<2>, dim3(768), dim3(256), 0, 0, buf, nunits);
                }
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float t; CK(hipEventElapsedTime(&t, e0, e1));
                if (round) ms.push_back(t / 2);
            }
            std::sort(ms.begin(), ms.end());
            printf("%-12s %-28s median %.3f ms  %.1f GB/s\n", anames[a], dnames[d], ms[ms.size() / 2], (double)bytes / ms[ms.size() / 2] / 1e6);
        }
        CK(hipFree(buf));
    }
    return 0;
}
