// hbm_write_pairs.hip -- does a lane that owns a whole 32-byte element (two 16-byte stores at +0 and +16, i.e. every
// store instruction covers 2 KiB at 50 % density) stream to HBM as fast as the fully contiguous 16-B-per-lane form?
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wp tools/hbm_write_pairs.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <algorithm>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void fill(v4u *__restrict__ out, size_t slab16, size_t nslabs)
{
    v4u a = {blockIdx.x, threadIdx.x, 3, 4}, b = {5, 6, 7, 8};
    for (size_t s = blockIdx.x; s < nslabs; s += gridDim.x) {
        v4u *o = out + s * slab16;
        if (MODE == 0) {                 // 16 B per lane, contiguous: 1 KiB per instruction
            for (size_t i = threadIdx.x; i < slab16; i += 256) o[i] = a;
        } else {                         // 32 B per lane: two instructions, each 16 B at stride 32 B
            for (size_t i = threadIdx.x; 2 * i + 1 < slab16; i += 256) { o[2 * i] = a; o[2 * i + 1] = b; }
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const size_t bytes = (size_t)20 << 30, slab16 = 5015168 / 16, nslabs = bytes / 16 / slab16;
    v4u *buf;
    CK(hipMalloc((void **)&buf, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    std::vector<float> t[2];
    for (int round = 0; round < 9; round++)
        for (int mode = 0; mode < 2; mode++) {
            CK(hipEventRecord(e0));
            for (int k = 0; k < 3; k++) {
                if (mode == 0) hipLaunchKernelGGL(fill<0>, dim3(768), dim3(256), 0, 0, buf, slab16, nslabs);
                else hipLaunchKernelGGL(fill<1>, dim3(768), dim3(256), 0, 0, buf, slab16, nslabs);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round) t[mode].push_back(ms / 3);
        }
    for (int mode = 0; mode < 2; mode++) {
        std::sort(t[mode].begin(), t[mode].end());
        printf("%s  median %.3f ms  %.1f GB/s\n", mode ? "32 B per lane (2 x 16 B, stride 32)" : "16 B per lane (contiguous)       ",
               t[mode][t[mode].size() / 2], (double)nslabs * slab16 * 16 / t[mode][t[mode].size() / 2] / 1e6);
    }
    return 0;
}
