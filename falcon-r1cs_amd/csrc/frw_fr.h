// frw_fr.h -- BLS12-381 Fr arithmetic on the device: 8 x 32-bit limbs, Montgomery form (ark-ff's Fp256 bytes).
// Shared by frw_r1cs_check.hip (sparse products) and frw_qap.hip (number-theoretic transforms over Fr).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace frw {

#define FRW_P32 {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u}
#define FRW_R32 {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u}

struct Fr8 { uint32_t l[8]; };

__device__ __forceinline__ Fr8 fr_load(const uint32_t *p)
{
    Fr8 r;
    const uint4 a = *(const uint4 *)p, b = *(const uint4 *)(p + 4);
    r.l[0] = a.x; r.l[1] = a.y; r.l[2] = a.z; r.l[3] = a.w; r.l[4] = b.x; r.l[5] = b.y; r.l[6] = b.z; r.l[7] = b.w;
    return r;
}

// r = (a + b) mod p, inputs < p
__device__ __forceinline__ Fr8 fr_add(const Fr8 &a, const Fr8 &b)
{
    constexpr uint32_t P[8] = FRW_P32;
    Fr8 s, d;
    uint32_t c = 0, bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)a.l[i] + b.l[i] + c;
        s.l[i] = (uint32_t)x;
        c = (uint32_t)(x >> 32);
    }
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)s.l[i] - P[i] - bw;
        d.l[i] = (uint32_t)x;
        bw = (uint32_t)(x >> 63);
    }
    const bool ge = c || !bw;              // p < 2^255, so c is never set; kept for clarity
#pragma unroll
    for (int i = 0; i < 8; i++) s.l[i] = ge ? d.l[i] : s.l[i];
    return s;
}

// Montgomery product a * b / 2^256 mod p (CIOS, 32-bit limbs; -p^-1 mod 2^32 = 0xffffffff)
__device__ __forceinline__ Fr8 fr_mul(const Fr8 &a, const Fr8 &b)
{
    constexpr uint32_t P[8] = FRW_P32;
    uint32_t T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int i = 0; i < 8; i++) {
        uint64_t acc;
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            acc = (uint64_t)a.l[i] * b.l[j] + T[j] + c;
            T[j] = (uint32_t)acc;
            c = (uint32_t)(acc >> 32);
        }
        acc = (uint64_t)T[8] + c;
        T[8] = (uint32_t)acc;
        const uint32_t t9 = (uint32_t)(acc >> 32);
        const uint32_t m = 0u - T[0];
        acc = (uint64_t)m * P[0] + T[0];
        c = (uint32_t)(acc >> 32);
#pragma unroll
        for (int j = 1; j < 8; j++) {
            acc = (uint64_t)m * P[j] + T[j] + c;
            T[j - 1] = (uint32_t)acc;
            c = (uint32_t)(acc >> 32);
        }
        acc = (uint64_t)T[8] + c;
        T[7] = (uint32_t)acc;
        T[8] = t9 + (uint32_t)(acc >> 32);
    }
    Fr8 r, d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)T[i] - P[i] - bw;
        d.l[i] = (uint32_t)x;
        bw = (uint32_t)(x >> 63);
    }
    const bool ge = T[8] || !bw;
#pragma unroll
    for (int i = 0; i < 8; i++) r.l[i] = ge ? d.l[i] : T[i];
    return r;
}

__device__ __forceinline__ void fr_store(uint32_t *p, const Fr8 &v)
{
    *(uint4 *)p = make_uint4(v.l[0], v.l[1], v.l[2], v.l[3]);
    *(uint4 *)(p + 4) = make_uint4(v.l[4], v.l[5], v.l[6], v.l[7]);
}

// r = (a - b) mod p, inputs < p
__device__ __forceinline__ Fr8 fr_sub(const Fr8 &a, const Fr8 &b)
{
    constexpr uint32_t P[8] = FRW_P32;
    Fr8 d;
    uint32_t bw = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)a.l[i] - b.l[i] - bw;
        d.l[i] = (uint32_t)x;
        bw = (uint32_t)(x >> 63);
    }
    const uint32_t mask = 0u - bw;               // borrow: add p back
    uint32_t c = 0;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const uint64_t x = (uint64_t)d.l[i] + (P[i] & mask) + c;
        d.l[i] = (uint32_t)x;
        c = (uint32_t)(x >> 32);
    }
    return d;
}

}  // namespace frw
