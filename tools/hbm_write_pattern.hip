// hbm_write_pattern.hip -- does the write stream care HOW a workgroup's four wavefronts walk their 5 MB unit?
//   A  workgroup-contiguous: the 256 lanes write 4 KiB per step, steps consecutive (the calibration stream of bench.py)
//   B  wave-private tiles:   every wavefront walks its own 58 KiB tile 1 KiB per step, the four tiles adjacent
//                            (what the witness kernel's tile writer does)
//   D  interleaved:          the four wavefronts share one tile, wavefront w writes the 1 KiB steps w, w+4, w+8, ...
//   E  blocked hand-out:     as A, but workgroup b owns the contiguous run of units [b R, (b+1) R) instead of the stride
//                            b, b + grid, ...: the chip's concurrent writes are spread over the whole buffer at all times
// Same bytes, same grid, static hand-out of units, interleaved rounds, medians.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/wp tools/hbm_write_pattern.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr size_t UNIT16 = 5015168 / 16;      // one Falcon-1024 witness in 16-byte chunks
constexpr size_t TILE16 = 58 * 64;           // a mod_q tile: 58 stores of 64 lanes

template <int PATTERN>
__global__ __launch_bounds__(256) void fill(v4u *__restrict__ out, size_t nunits)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    v4u v = {blockIdx.x, threadIdx.x, 3, 4};
    const size_t per = (nunits + gridDim.x - 1) / gridDim.x;
    const size_t first = PATTERN == 3 ? blockIdx.x * per : blockIdx.x;
    const size_t last = PATTERN == 3 ? (first + per < nunits ? first + per : nunits) : nunits;
    const size_t step = PATTERN == 3 ? 1 : gridDim.x;
    for (size_t u = first; u < last; u += step) {
        v4u *o = out + u * UNIT16;
        if (PATTERN == 0 || PATTERN == 3) {
            for (size_t i = threadIdx.x; i < UNIT16; i += 256) o[i] = v;
        } else {
            const size_t ntiles = UNIT16 / TILE16;            // 84 full tiles; the remainder is written A-style
            if (PATTERN == 1) {
                for (size_t t = wave; t < ntiles; t += 4)
                    for (size_t it = 0; it < 58; it++) o[t * TILE16 + it * 64 + lane] = v;
            } else {
                for (size_t t = 0; t < ntiles; t++)
                    for (size_t it = wave; it < 58; it += 4) o[t * TILE16 + it * 64 + lane] = v;
            }
            for (size_t i = ntiles * TILE16 + threadIdx.x; i < UNIT16; i += 256) o[i] = v;
        }
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv)
{
    const size_t nunits = argc > 1 ? (size_t)atol(argv[1]) : 8192;      // 8,192 units = 41 GB; 16,384 = bench.py's buffer
    v4u *buf;
    CK(hipMalloc((void **)&buf, nunits * UNIT16 * 16));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    struct Cfg { int pat; int grid; std::vector<float> ms; };
    std::vector<Cfg> cfgs;
    for (int grid : {768, 1536, 3072, 6144})
        for (int pat : {0, 1, 2, 3}) cfgs.push_back({pat, grid, {}});
    for (int round = 0; round < 7; round++)
        for (auto &c : cfgs) {
            CK(hipEventRecord(e0));
            for (int k = 0; k < 2; k++) {
                if (c.pat == 0) hipLaunchKernelGGL(fill<0>, dim3(c.grid), dim3(256), 0, 0, buf, nunits);
                else if (c.pat == 1) hipLaunchKernelGGL(fill<1>, dim3(c.grid), dim3(256), 0, 0, buf, nunits);
                else if (c.pat == 2) hipLaunchKernelGGL(fill<2>, dim3(c.grid), dim3(256), 0, 0, buf, nunits);
                else hipLaunchKernelGGL(fill<3>, dim3(c.grid), dim3(256), 0, 0, buf, nunits);
            }
            CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            if (round) c.ms.push_back(ms / 2);
        }
    const char *names[4] = {"A workgroup-contiguous", "B wave-private tiles  ", "D interleaved         ", "E blocked hand-out    "};
    for (auto &c : cfgs) {
        std::sort(c.ms.begin(), c.ms.end());
        printf("units=%zu %s grid=%5d  median %.3f ms  %.1f GB/s\n", nunits, names[c.pat], c.grid, c.ms[c.ms.size() / 2],
               (double)nunits * UNIT16 * 16 / c.ms[c.ms.size() / 2] / 1e6);
    }
    return 0;
}
