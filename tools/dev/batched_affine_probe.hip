// batched_affine_probe.hip -- the bounded experiment round 4's review asked for before batched-affine bucket accumulation is built or
// dismissed: how many point additions per second does the CORE of such a scheme sustain on this chip, beside the XYZZ mixed additions
// of msm_bucket_kernel under the same conditions?
//
//   A  (what the product does)   every thread: acc (+)= table[idx[k]], k < K, XYZZ mixed addition (8 M + 2 S, frw_fq29.h pt_add_affine),
//                                the next row fetched while the current one is added -- msm_bucket_kernel's loop
//   B  (one level of a batched-affine tree)   every thread: K independent sums table[a[k]] + table[b[k]] in AFFINE coordinates with ONE
//                                inversion for all K of them (Montgomery's trick inside the thread: a forward pass of running products
//                                parked in memory, fq_inv, a backward pass): per addition 1 M forward, 2 M for its inverse, 2 M + 1 S for
//                                the chord; 56 B of prefix written and read, two rows of 112 B gathered twice (their x for the forward
//                                pass, x and y for the backward), one row written.  Degenerate pairs (equal x) are not handled: a real
//                                kernel flags them for the complete formula, the probe's points are distinct.
// B is an UPPER bound on what a batched-affine bucket kernel could do: it has no tree bookkeeping, no ragged bucket sizes, no special
// cases, and K additions per inversion is as large as one likes.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I falcon-r1cs_amd/csrc tools/dev/batched_affine_probe.hip -o /tmp/ba_probe && /tmp/ba_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "frw_fq29.h"

using namespace frw;
constexpr int PW = 2 * NLQ;                    // words of a table row: x, y in fourteen 29-bit limbs each

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } } while (0)

__device__ __forceinline__ G1Affine29 row_load(const uint32_t *row)
{
    G1Affine29 p;
    const uint4 *v = (const uint4 *)row;
    uint32_t w[PW];
#pragma unroll
    for (int k = 0; k < PW / 4; k++) { const uint4 t = v[k]; w[4 * k] = t.x; w[4 * k + 1] = t.y; w[4 * k + 2] = t.z; w[4 * k + 3] = t.w; }
    p.x = FqField::load(w);
    p.y = FqField::load(w + NLQ);
    p.inf = false;
    return p;
}
__device__ __forceinline__ void row_store(uint32_t *row, const Fq29 &x, const Fq29 &y)
{
#pragma unroll
    for (int k = 0; k < NLQ; k++) { row[k] = x.l[k]; row[NLQ + k] = y.l[k]; }
}
__device__ __forceinline__ Fq29 g1_gen_x() { return fq_const(G1_GEN_X29); }
__device__ __forceinline__ Fq29 g1_gen_y() { return fq_const(G1_GEN_Y29); }

// table[i] = (i + 1) G, affine
__global__ __launch_bounds__(64) void make_points_kernel(uint32_t n, uint32_t *table)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    G1Affine29 g;
    g.x = g1_gen_x(); g.y = g1_gen_y(); g.inf = false;
    G1Xyzz acc = g1_identity();
    for (int bit = 23; bit >= 0; bit--) {
        acc = g1_double(acc);
        if (((i + 1) >> bit) & 1u) acc = g1_add_affine(acc, g);
    }
    const G1Affine29 a = g1_to_affine(acc);
    row_store(table + (size_t)i * PW, a.x, a.y);
}

// A: msm_bucket_kernel's loop
__global__ __launch_bounds__(64, 2) void xyzz_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ idx, uint32_t per_thread,
                                                     uint32_t *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x, threads = (size_t)gridDim.x * 64;
    G1Xyzz acc = g1_identity();
    uint32_t e = idx[t];
    G1Affine29 p = row_load(table + (size_t)e * PW);
    for (uint32_t k = 0; k < per_thread; k++) {
        const uint32_t e_next = k + 1 < per_thread ? idx[(size_t)(k + 1) * threads + t] : e;
        const G1Affine29 p_next = row_load(table + (size_t)e_next * PW);
        acc = pt_add_affine(acc, p);                                   // (the template itself: it is force-inlined, the g1_ wrapper is a call)
        e = e_next;
        p = p_next;
    }
    uint32_t *o = out + t * 4 * NLQ;
#pragma unroll
    for (int k = 0; k < NLQ; k++) { o[k] = acc.x.l[k]; o[NLQ + k] = acc.y.l[k]; o[2 * NLQ + k] = acc.zz.l[k]; o[3 * NLQ + k] = acc.zzz.l[k]; }
}

// B: one level of a batched-affine tree, Montgomery's trick inside the thread
__global__ __launch_bounds__(64, 2) void affine_level_kernel(const uint32_t *__restrict__ table, const uint32_t *__restrict__ ia,
                                                             const uint32_t *__restrict__ ib, uint32_t per_thread, uint32_t *__restrict__ prefix,
                                                             uint32_t *__restrict__ out)
{
    const size_t t = (size_t)blockIdx.x * 64 + threadIdx.x, threads = (size_t)gridDim.x * 64;
    Fq29 acc = fq_const(FQ29_ONE);
    for (uint32_t k = 0; k < per_thread; k++) {
        const size_t at = (size_t)k * threads + t;
        const Fq29 x1 = FqField::load(table + (size_t)ia[at] * PW), x2 = FqField::load(table + (size_t)ib[at] * PW);
        uint32_t *pr = prefix + at * NLQ;
#pragma unroll
        for (int j = 0; j < NLQ; j++) pr[j] = acc.l[j];
        acc = fq_mul(acc, fq_sub<256>(x2, x1));
    }
    Fq29 inv = fq_inv(acc);
    for (uint32_t k = per_thread; k-- > 0;) {
        const size_t at = (size_t)k * threads + t;
        const G1Affine29 p = row_load(table + (size_t)ia[at] * PW), q = row_load(table + (size_t)ib[at] * PW);
        const Fq29 pre = FqField::load(prefix + at * NLQ);
        const Fq29 dx = fq_sub<256>(q.x, p.x), dy = fq_sub<256>(q.y, p.y);
        const Fq29 dinv = fq_mul(inv, pre);                            // 1 / dx
        inv = fq_mul(inv, dx);
        const Fq29 lam = fq_mul(dy, dinv);
        const Fq29 x3 = fq_sub<64>(fq_sqr(lam), fq_add(p.x, q.x));       // lambda^2 - x1 - x2: rows < 4 q each, x3 < 66 q
        const Fq29 y3 = fq_sub<16>(fq_mul(lam, fq_sub<256>(p.x, x3)), p.y);
        row_store(out + at * PW, x3, y3);
    }
}

// the same sums through the product's complete formula, for the comparison: canonical ark-ff bytes of both
__global__ __launch_bounds__(64) void check_kernel(const uint32_t *table, const uint32_t *ia, const uint32_t *ib, const uint32_t *got, uint32_t count,
                                                   uint32_t *mismatches)
{
    const uint32_t i = blockIdx.x * 64 + threadIdx.x;
    if (i >= count) return;
    const G1Affine29 p = row_load(table + (size_t)ia[i] * PW), q = row_load(table + (size_t)ib[i] * PW);
    const G1Affine29 want = g1_to_affine(g1_add_affine(g1_from_affine(p), q));
    uint32_t a[12], b[12], bad = 0;
    fq_to_ark(want.x, a); fq_to_ark(FqField::load(got + (size_t)i * PW), b);
    for (int k = 0; k < 12; k++) bad |= a[k] ^ b[k];
    fq_to_ark(want.y, a); fq_to_ark(FqField::load(got + (size_t)i * PW + NLQ), b);
    for (int k = 0; k < 12; k++) bad |= a[k] ^ b[k];
    if (bad) atomicAdd(mismatches, 1u);
}

int main(int argc, char **argv)
{
    const uint32_t n = argc > 1 ? (uint32_t)std::atoi(argv[1]) : (1u << 22);           // table points (470 MB: like a 2^22-point h_query)
    const uint32_t threads = 131072;                                                   // two wavefronts on every SIMD of 256 CUs
    CHECK(hipSetDevice(0));
    uint32_t *table, *idx_a, *idx_b, *prefix, *out, *mism;
    const uint32_t kmax = 512;
    CHECK(hipMalloc(&table, (size_t)n * PW * 4));
    CHECK(hipMalloc(&idx_a, (size_t)threads * kmax * 4));
    CHECK(hipMalloc(&idx_b, (size_t)threads * kmax * 4));
    CHECK(hipMalloc(&prefix, (size_t)threads * kmax * NLQ * 4));
    CHECK(hipMalloc(&out, (size_t)threads * kmax * PW * 4));
    CHECK(hipMalloc(&mism, 4));
    hipLaunchKernelGGL(make_points_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, n, table);
    std::vector<uint32_t> a((size_t)threads * kmax), b(a.size());
    uint64_t s = 0x9E3779B97F4A7C15ull;
    auto next = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (uint32_t)(s >> 20); };
    for (size_t i = 0; i < a.size(); i++) {
        a[i] = next() % n;
        do b[i] = next() % n; while (b[i] == a[i]);
    }
    CHECK(hipMemcpy(idx_a, a.data(), a.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(idx_b, b.data(), b.size() * 4, hipMemcpyHostToDevice));
    CHECK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    std::printf("# %u table points (%.0f MB of rows), %u threads = 2 wavefronts per SIMD; additions per second, second of two runs\n", n, n * 112.0 / 1e6, threads);
    for (uint32_t k : {64u, 128u, 256u, 512u}) {
        float ms_a = 0, ms_b = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(xyzz_kernel, dim3(threads / 64), dim3(64), 0, nullptr, table, idx_a, k, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_a, e0, e1));
        }
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(affine_level_kernel, dim3(threads / 64), dim3(64), 0, nullptr, table, idx_a, idx_b, k, prefix, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_b, e0, e1));
        }
        // the same additions on twice the threads (four wavefronts per SIMD: B needs 117 registers), half as many each
        float ms_b4 = 0;
        for (int rep = 0; rep < 2; rep++) {
            CHECK(hipEventRecord(e0));
            hipLaunchKernelGGL(affine_level_kernel, dim3(2 * threads / 64), dim3(64), 0, nullptr, table, idx_a, idx_b, k / 2, prefix, out);
            CHECK(hipEventRecord(e1));
            CHECK(hipEventSynchronize(e1));
            CHECK(hipEventElapsedTime(&ms_b4, e0, e1));
        }
        CHECK(hipMemset(mism, 0, 4));
        hipLaunchKernelGGL(check_kernel, dim3(4096 / 64), dim3(64), 0, nullptr, table, idx_a, idx_b, out, 4096u, mism);
        uint32_t bad = 0;
        CHECK(hipMemcpy(&bad, mism, 4, hipMemcpyDeviceToHost));
        const double adds = (double)threads * k;
        std::printf("K = %3u per thread   A  XYZZ mixed additions %7.3f ms = %6.2f G/s    B  batched affine (one inversion per %u) %7.3f ms = %6.2f G/s    B / A = %.3f    "
                    "B on four wavefronts per SIMD (one inversion per %u) %7.3f ms = %6.2f G/s  B / A = %.3f    B's first 4,096 sums against the complete formula: %u differ\n",
                    k, ms_a, adds / ms_a / 1e6, k, ms_b, adds / ms_b / 1e6, ms_a / ms_b, k / 2, ms_b4, adds / ms_b4 / 1e6, ms_a / ms_b4, bad);
    }
    return 0;
}
