// valu_rates.hip -- issue rate of the VALU instructions a 256-bit Montgomery product is built from, on the device at hand.
// Every wave runs ITER iterations of 16 independent dependency chains of one instruction; 4 waves per SIMD.
// Prints wave-instructions per SIMD per microsecond and the ratio to v_add_u32 (full rate).
//   hipcc --offload-arch=gfx950 -O3 tools/dev/valu_rates.hip -o /tmp/valu_rates && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

constexpr int ITER = 4096;

#define CHAINS16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(uint64_t *out, uint32_t seed)
{
    uint64_t a[16];
    double d[16];
    uint32_t x = seed + threadIdx.x, y = seed * 3 + 1;
    const double fx = 1.0 + seed * 1e-9, fy = 0.5;
#pragma unroll
    for (int i = 0; i < 16; i++) { a[i] = x + i; d[i] = (double)(x + i); }
    for (int it = 0; it < ITER; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++) {
            if (KIND == 0) { uint32_t t = (uint32_t)a[i]; asm volatile("v_add_u32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(y)); a[i] = t; }
            if (KIND == 1) { asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(x), "v"(y) : "vcc"); }
            if (KIND == 2) { uint32_t t = (uint32_t)a[i]; asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(y)); a[i] = t; }
            if (KIND == 3) { uint32_t t = (uint32_t)a[i]; asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(t) : "v"(t), "v"(y)); a[i] = t; }
            if (KIND == 4) { asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(d[i]) : "v"(fx), "v"(fy)); }
            if (KIND == 5) { asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(a[i]) : "v"((uint64_t)y)); }
            if (KIND == 6) { uint32_t t = (uint32_t)a[i]; asm volatile("v_mad_u32_u24 %0, %1, %2, %0" : "+v"(t) : "v"(x), "v"(y)); a[i] = t; }
            if (KIND == 7) { uint32_t t = (uint32_t)a[i]; asm volatile("v_mul_hi_u32_u24 %0, %1, %2" : "=v"(t) : "v"(t), "v"(y)); a[i] = t; }
            if (KIND == 8) { uint32_t t = (uint32_t)a[i]; asm volatile("v_mov_b32 %0, %1" : "=v"(t) : "v"(t)); a[i] = t; }
            if (KIND == 9) { uint32_t t = (uint32_t)a[i]; asm volatile("v_addc_co_u32 %0, vcc, %1, %2, vcc" : "=v"(t) : "v"(t), "v"(y) : "vcc"); a[i] = t; }
            if (KIND == 10) { uint32_t t = (uint32_t)a[i]; asm volatile("v_add3_u32 %0, %1, %2, %3" : "=v"(t) : "v"(t), "v"(y), "v"(x)); a[i] = t; }
            if (KIND == 11) { asm volatile("v_mul_f64 %0, %0, %1" : "+v"(d[i]) : "v"(fx)); }
            if (KIND == 12) { asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i]) : "v"(fy)); }
        }
    }
    uint64_t s = 0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += a[i] + (uint64_t)d[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
double run(const char *name, uint64_t *buf, int cus, double base)
{
    const int grid = cus * 4;                       // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(256), 0, 0, buf, 1u);
    hipEventRecord(e0);
    hipLaunchKernelGGL(rate_kernel<KIND>, dim3(grid), dim3(256), 0, 0, buf, 2u);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr_per_simd = 4.0 * ITER * 16;           // waves per SIMD x instructions per wave
    const double rate = wave_instr_per_simd / (ms * 1e3);         // per microsecond
    printf("%-18s %8.3f ms  %8.1f wave-instr/SIMD/us  %5.2f x v_add_u32 time\n", name, ms, rate, base > 0 ? base / rate : 1.0);
    return rate;
}

int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    uint64_t *buf;
    hipMalloc(&buf, (size_t)cus * 4 * 256 * 8);
    printf("%s, %d CUs, %d MHz\n", p.name, cus, p.clockRate / 1000);
    const double base = run<0>("v_add_u32", buf, cus, 0);
    run<1>("v_mad_u64_u32", buf, cus, base);
    run<2>("v_mul_lo_u32", buf, cus, base);
    run<3>("v_mul_hi_u32", buf, cus, base);
    run<4>("v_fma_f64", buf, cus, base);
    run<11>("v_mul_f64", buf, cus, base);
    run<12>("v_add_f64", buf, cus, base);
    run<5>("v_lshl_add_u64", buf, cus, base);
    run<6>("v_mad_u32_u24", buf, cus, base);
    run<7>("v_mul_hi_u32_u24", buf, cus, base);
    run<8>("v_mov_b32", buf, cus, base);
    run<9>("v_addc_co_u32", buf, cus, base);
    run<10>("v_add3_u32", buf, cus, base);
    return 0;
}
