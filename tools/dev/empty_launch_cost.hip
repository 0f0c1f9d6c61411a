// What does a launch cost whose every workgroup leaves at once?  (The question behind "run the seven-transform witness map only
// for the signatures whose witness violates the system, as a predicated second set of launches": 18 such launches per call.)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/empty_launch_cost tools/dev/empty_launch_cost.hip && /tmp/empty_launch_cost
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ __launch_bounds__(64) void predicated(const uint32_t *flags, uint32_t *out)
{
    if (flags[blockIdx.y / 3] == 0) return;
    out[blockIdx.y * gridDim.x + blockIdx.x] = threadIdx.x;
}
int main()
{
    const int sigs = 64;
    uint32_t *flags, *out;
    hipMalloc(&flags, sigs * 4);
    hipMemset(flags, 0, sigs * 4);
    hipMalloc(&out, 512 * 3 * sigs * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int tiles : {512, 256}) {
        for (int rep = 0; rep < 2; rep++) {
            hipEventRecord(e0, nullptr);
            for (int k = 0; k < 18 * 20; k++) hipLaunchKernelGGL(predicated, dim3(tiles, 3 * sigs), dim3(64), 0, nullptr, flags, out);
            hipEventRecord(e1, nullptr);
            hipEventSynchronize(e1);
            float ms = 0;
            hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("grid (%d, %d) x 64 threads, every workgroup leaves at once: %.1f us per launch, %.3f ms per 18 launches\n", tiles, 3 * sigs, ms * 1e3 / (18 * 20), ms / 20);
        }
    }
    return 0;
}
