// repro_store_hazard.hip -- how many wait states does gfx950 need between a 128-bit buffer store whose soffset is an
// SGPR and a VALU write to the store's data registers?  (LLVM's hazard recogniser pads the immediate-soffset form only;
// DESIGN.md section 5.1.)  Each wavefront stores a known pattern, then overwrites data register 0 after PAD wait states;
// the host counts 16-byte chunks whose first dword came out as the overwrite value.  Many workgroups per CU, so that
// waves share SIMDs (the corruption was only ever seen under co-residency).
// Build + run: hipcc --offload-arch=gfx950 -O3 -o /tmp/rsh tools/dev/repro_store_hazard.hip && /tmp/rsh
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>

typedef uint32_t v4u __attribute__((ext_vector_type(4)));

template <int PAD>      // PAD = -1: no padding; PAD = n >= 0: s_nop n  (n + 1 wait states)
__global__ __launch_bounds__(256) void k(v4u *out, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const uint64_t base = (uint64_t)(out + wave * (size_t)iters * 64);
    // raw buffer resource in SGPRs: base[47:0], stride 0, num_records = bytes, word 3 = 32-bit data format
    const v4u rsrc = {(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)base),
                      (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(base >> 32) & 0xffffu),
                      (uint32_t)__builtin_amdgcn_readfirstlane(iters * 1024), 0x00020000u};
    const int voff = lane * 16;
    for (int it = 0; it < iters; it++) {
        const int soff = __builtin_amdgcn_readfirstlane(it * 1024);
        const uint32_t a = 0x11110000u + lane;
        if (PAD < 0)
            asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, 0x2222\n\tv_mov_b32 v22, 0x3333\n\tv_mov_b32 v23, 0x4444\n\t"
                         "s_nop 4\n\t"
                         "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\t"
                         "v_mov_b32 v20, 0xdead\n\t"
                         "s_nop 4" ::"v"(a), "v"(voff), "s"(rsrc), "s"(soff) : "v20", "v21", "v22", "v23", "memory");
        else
            asm volatile("v_mov_b32 v20, %0\n\tv_mov_b32 v21, 0x2222\n\tv_mov_b32 v22, 0x3333\n\tv_mov_b32 v23, 0x4444\n\t"
                         "s_nop 4\n\t"
                         "buffer_store_dwordx4 v[20:23], %1, %2, %3 offen\n\t"
                         "s_nop %4\n\t"
                         "v_mov_b32 v20, 0xdead\n\t"
                         "s_nop 4" ::"v"(a), "v"(voff), "s"(rsrc), "s"(soff), "n"(PAD) : "v20", "v21", "v22", "v23", "memory");
    }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main()
{
    const int grid = 2048, iters = 512;                 // 8 workgroups per CU resident, 4 GB of output
    const size_t chunks = (size_t)grid * 4 * iters * 64;
    v4u *d;
    CK(hipMalloc((void **)&d, chunks * 16));
    std::vector<v4u> h(chunks);
    for (int pad = -1; pad <= 2; pad++) {
        CK(hipMemset(d, 0xff, chunks * 16));
        for (int rep = 0; rep < 4; rep++) {
            if (pad == -1) hipLaunchKernelGGL(k<-1>, dim3(grid), dim3(256), 0, 0, d, iters);
            else if (pad == 0) hipLaunchKernelGGL(k<0>, dim3(grid), dim3(256), 0, 0, d, iters);
            else if (pad == 1) hipLaunchKernelGGL(k<1>, dim3(grid), dim3(256), 0, 0, d, iters);
            else hipLaunchKernelGGL(k<2>, dim3(grid), dim3(256), 0, 0, d, iters);
        }
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(h.data(), d, chunks * 16, hipMemcpyDeviceToHost));
        size_t bad = 0, other = 0;
        for (size_t i = 0; i < chunks; i++) {
            const uint32_t want = 0x11110000u + (uint32_t)(i & 63);
            if (h[i].x == 0xdead) bad++;
            else if (h[i].x != want || h[i].y != 0x2222 || h[i].z != 0x3333 || h[i].w != 0x4444) other++;
        }
        printf("wait states between the store and the overwrite: %d  ->  %zu of %zu chunks carry the overwrite value (%.4f %%), %zu otherwise wrong\n",
               pad + 1, bad, chunks, 100.0 * bad / chunks, other);
    }
    return 0;
}
