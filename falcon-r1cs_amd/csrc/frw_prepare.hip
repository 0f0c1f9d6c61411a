// frw_prepare.hip -- input preparation in front of the witness kernels: what the reference does
// with falcon-rust before any gadget runs (falcon-r1cs/src/circuits/falcon_ntt.rs:27-28,44):
//     sig_poly = Polynomial::from(&sig)            -> decode_signatures_kernel   (Falcon spec Alg. 18, Decompress)
//     pk_poly  = Polynomial::from(&pk)             -> decode_public_keys_kernel  (14-bit modq_decode)
//     hm       = Polynomial::from_hash_of_message(msg, sig.nonce())  -> hash_to_point_kernel (SHAKE256, Alg. 3)
// falcon-rust is not under /root/reference; the formats are the Falcon specification's (v1.2 sections 3.7, 3.11).
//
// These kernels are latency/ALU work on a few KB per signature (Keccak-f[1600]: ~17 permutations per Falcon-1024
// hash), two orders of magnitude below the witness kernel's 5 MB write stream; one lane per signature keeps every
// lane of a wavefront in the same permutation.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frw_device.h"

namespace frw {

__constant__ uint64_t KECCAK_RC[24] = {
    0x0000000000000001ull, 0x0000000000008082ull, 0x800000000000808aull, 0x8000000080008000ull, 0x000000000000808bull,
    0x0000000080000001ull, 0x8000000080008081ull, 0x8000000000008009ull, 0x000000000000008aull, 0x0000000000000088ull,
    0x0000000080008009ull, 0x000000008000000aull, 0x000000008000808bull, 0x800000000000008bull, 0x8000000000008089ull,
    0x8000000000008003ull, 0x8000000000008002ull, 0x8000000000000080ull, 0x000000000000800aull, 0x800000008000000aull,
    0x8000000080008081ull, 0x8000000000008080ull, 0x0000000080000001ull, 0x8000000080008008ull};

__device__ __forceinline__ uint64_t rol64(uint64_t x, int n) { return (x << n) | (x >> (64 - n)); }

// Keccak-f[1600], state as 25 lanes A[x + 5y]; every index below is a compile-time constant (registers, no scratch)
__device__ __forceinline__ void keccak_f1600(uint64_t (&a)[25])
{
#pragma unroll 1
    for (int r = 0; r < 24; r++) {
        uint64_t c0 = a[0] ^ a[5] ^ a[10] ^ a[15] ^ a[20];
        uint64_t c1 = a[1] ^ a[6] ^ a[11] ^ a[16] ^ a[21];
        uint64_t c2 = a[2] ^ a[7] ^ a[12] ^ a[17] ^ a[22];
        uint64_t c3 = a[3] ^ a[8] ^ a[13] ^ a[18] ^ a[23];
        uint64_t c4 = a[4] ^ a[9] ^ a[14] ^ a[19] ^ a[24];
        const uint64_t d0 = c4 ^ rol64(c1, 1), d1 = c0 ^ rol64(c2, 1), d2 = c1 ^ rol64(c3, 1), d3 = c2 ^ rol64(c4, 1),
                       d4 = c3 ^ rol64(c0, 1);
        // theta + rho + pi: b[y + 5((2x+3y) mod 5)] = rol(a[x+5y] ^ d[x], rot[x][y])
        const uint64_t b0 = a[0] ^ d0;
        const uint64_t b10 = rol64(a[1] ^ d1, 1), b20 = rol64(a[2] ^ d2, 62), b5 = rol64(a[3] ^ d3, 28), b15 = rol64(a[4] ^ d4, 27);
        const uint64_t b16 = rol64(a[5] ^ d0, 36), b1 = rol64(a[6] ^ d1, 44), b11 = rol64(a[7] ^ d2, 6), b21 = rol64(a[8] ^ d3, 55),
                       b6 = rol64(a[9] ^ d4, 20);
        const uint64_t b7 = rol64(a[10] ^ d0, 3), b17 = rol64(a[11] ^ d1, 10), b2 = rol64(a[12] ^ d2, 43), b12 = rol64(a[13] ^ d3, 25),
                       b22 = rol64(a[14] ^ d4, 39);
        const uint64_t b23 = rol64(a[15] ^ d0, 41), b8 = rol64(a[16] ^ d1, 45), b18 = rol64(a[17] ^ d2, 15), b3 = rol64(a[18] ^ d3, 21),
                       b13 = rol64(a[19] ^ d4, 8);
        const uint64_t b14 = rol64(a[20] ^ d0, 18), b24 = rol64(a[21] ^ d1, 2), b9 = rol64(a[22] ^ d2, 61), b19 = rol64(a[23] ^ d3, 56),
                       b4 = rol64(a[24] ^ d4, 14);
        // chi (+ iota on lane 0)
        a[0] = b0 ^ (~b1 & b2) ^ KECCAK_RC[r]; a[1] = b1 ^ (~b2 & b3); a[2] = b2 ^ (~b3 & b4); a[3] = b3 ^ (~b4 & b0); a[4] = b4 ^ (~b0 & b1);
        a[5] = b5 ^ (~b6 & b7); a[6] = b6 ^ (~b7 & b8); a[7] = b7 ^ (~b8 & b9); a[8] = b8 ^ (~b9 & b5); a[9] = b9 ^ (~b5 & b6);
        a[10] = b10 ^ (~b11 & b12); a[11] = b11 ^ (~b12 & b13); a[12] = b12 ^ (~b13 & b14); a[13] = b13 ^ (~b14 & b10); a[14] = b14 ^ (~b10 & b11);
        a[15] = b15 ^ (~b16 & b17); a[16] = b16 ^ (~b17 & b18); a[17] = b17 ^ (~b18 & b19); a[18] = b18 ^ (~b19 & b15); a[19] = b19 ^ (~b15 & b16);
        a[20] = b20 ^ (~b21 & b22); a[21] = b21 ^ (~b22 & b23); a[22] = b22 ^ (~b23 & b24); a[23] = b23 ^ (~b24 & b20); a[24] = b24 ^ (~b20 & b21);
    }
}

constexpr int SHAKE256_RATE = 136;
constexpr int NONCE_LEN = 40;

// hm = HashToPoint(nonce || msg): SHAKE256, big-endian 16-bit words, accept w < 5q as w mod q (Falcon spec Alg. 3)
__global__ __launch_bounds__(BLOCK) void hash_to_point_kernel(int logn, size_t batch, const uint8_t *__restrict__ nonces,
                                                              const uint8_t *__restrict__ msgs,
                                                              const uint64_t *__restrict__ msg_off, uint16_t *__restrict__ hm)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    const int n = 1 << logn;
    const uint8_t *nonce = nonces + s * NONCE_LEN;
    const uint8_t *msg = msgs + msg_off[s];
    // the host entry point rejects decreasing offsets; a device-side caller that breaks the precondition of
    // include/frw.h gets an empty message here rather than a 2^64-byte absorb loop
    const uint64_t o0 = msg_off[s], o1 = msg_off[s + 1];
    const size_t total = NONCE_LEN + (size_t)(o1 >= o0 ? o1 - o0 : 0);
    uint64_t st[25];
#pragma unroll
    for (int i = 0; i < 25; i++) st[i] = 0;
    // absorb nonce || msg || pad10*1 with the SHAKE domain bits (0x1F ... 0x80)
    for (size_t pos = 0;; pos += SHAKE256_RATE) {
#pragma unroll
        for (int l = 0; l < SHAKE256_RATE / 8; l++) {
            uint64_t w = 0;
#pragma unroll
            for (int b = 0; b < 8; b++) {
                const size_t p = pos + (size_t)(l * 8 + b);
                uint32_t byte = 0;
                if (p < NONCE_LEN) byte = nonce[p];
                else if (p < total) byte = msg[p - NONCE_LEN];
                else if (p == total) byte = 0x1F;
                w |= (uint64_t)byte << (8 * b);
            }
            st[l] ^= w;
        }
        const bool last = total < pos + SHAKE256_RATE;
        if (last) st[16] ^= 0x8000000000000000ull;
        keccak_f1600(st);
        if (last) break;
    }
    // squeeze
    uint16_t *out = hm + s * (size_t)n;
    int cnt = 0;
    while (cnt < n) {
#pragma unroll
        for (int w = 0; w < SHAKE256_RATE / 2; w++) {
            const uint64_t lane = st[w / 4];
            const int sh = (w % 4) * 16;
            const uint32_t v = (uint32_t)((lane >> sh) & 0xff) << 8 | (uint32_t)((lane >> (sh + 8)) & 0xff);
            if (v < 5 * Q && cnt < n) out[cnt++] = (uint16_t)(v % Q);
        }
        if (cnt < n) keccak_f1600(st);
    }
}

// pk: header 0x00 + logn, then N x 14 bits, big-endian bit order
__global__ __launch_bounds__(BLOCK) void decode_public_keys_kernel(int logn, size_t batch, const uint8_t *__restrict__ pk_bytes,
                                                                   uint16_t *__restrict__ pk, int32_t *__restrict__ status)
{
    const int n = 1 << logn;
    const size_t pk_len = 1 + (size_t)14 * n / 8;
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= batch * (size_t)n) return;
    const size_t s = idx >> logn;
    const int k = (int)(idx & (size_t)(n - 1));
    const uint8_t *p = pk_bytes + s * pk_len;
    const int bit = 14 * k, off = bit & 7;
    const uint8_t *q = p + 1 + (bit >> 3);
    uint32_t acc = (uint32_t)q[0] << 16 | (uint32_t)q[1] << 8;
    if (off + 14 > 16) acc |= q[2];
    const uint32_t c = (acc >> (24 - 14 - off)) & 0x3fffu;
    pk[idx] = (uint16_t)c;
    if (c >= Q || (k == 0 && p[0] != (uint8_t)logn)) atomicMax(&status[s], ST_DECODE);
}

// sig: header 0x30 + logn, 40-byte nonce, compressed coefficients, zero padding up to sig_len
__global__ __launch_bounds__(BLOCK) void decode_signatures_kernel(int logn, size_t batch, const uint8_t *__restrict__ sig_bytes,
                                                                  size_t sig_len, uint16_t *__restrict__ sig,
                                                                  uint8_t *__restrict__ nonce_out, int32_t *__restrict__ status)
{
    const size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= batch) return;
    const int n = 1 << logn;
    const uint8_t *p = sig_bytes + s * sig_len;
    uint16_t *out = sig + s * (size_t)n;
    bool ok = sig_len > 1 + NONCE_LEN && p[0] == (uint8_t)(0x30 + logn);
    if (ok && nonce_out)
        for (int i = 0; i < NONCE_LEN; i++) nonce_out[s * NONCE_LEN + i] = p[1 + i];
    const uint8_t *body = p + 1 + NONCE_LEN;
    const size_t body_len = ok ? sig_len - 1 - NONCE_LEN : 0;
    size_t v = 0;
    uint32_t acc = 0;
    int acc_len = 0;
    for (int u = 0; ok && u < n; u++) {
        if (v >= body_len) { ok = false; break; }
        acc = (acc << 8) | body[v++];
        const uint32_t b = acc >> acc_len;
        const uint32_t sign = b & 128u;
        uint32_t m = b & 127u;
        for (;;) {
            if (acc_len == 0) {
                if (v >= body_len) { ok = false; break; }
                acc = (acc << 8) | body[v++];
                acc_len = 8;
            }
            acc_len--;
            if ((acc >> acc_len) & 1u) break;
            m += 128;
            if (m > 2047) { ok = false; break; }
        }
        if (sign && m == 0) ok = false;             // "-0" is not a valid encoding
        if (ok) out[u] = (uint16_t)(sign ? Q - m : m);
    }
    if (ok && (acc & ((1u << acc_len) - 1u))) ok = false;          // unused bits of the last byte
    for (; ok && v < body_len; v++)
        if (body[v]) ok = false;                                    // padding
    status[s] = ok ? ST_OK : ST_DECODE;
}

hipError_t launch_hash_to_point(int logn, size_t batch, const uint8_t *nonces, const uint8_t *msgs, const uint64_t *msg_off,
                                uint16_t *hm, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(hash_to_point_kernel, dim3((unsigned)((batch + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, logn, batch, nonces,
                       msgs, msg_off, hm);
    return hipGetLastError();
}

hipError_t launch_decode_public_keys(int logn, size_t batch, const uint8_t *pk_bytes, uint16_t *pk, int32_t *status, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(status, 0, batch * sizeof(int32_t), st);
    if (e != hipSuccess) return e;
    const size_t total = batch << logn;
    hipLaunchKernelGGL(decode_public_keys_kernel, dim3((unsigned)((total + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, logn, batch,
                       pk_bytes, pk, status);
    return hipGetLastError();
}

hipError_t launch_decode_signatures(int logn, size_t batch, const uint8_t *sig_bytes, size_t sig_len, uint16_t *sig,
                                    uint8_t *nonce_out, int32_t *status, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    hipLaunchKernelGGL(decode_signatures_kernel, dim3((unsigned)((batch + BLOCK - 1) / BLOCK)), dim3(BLOCK), 0, st, logn, batch,
                       sig_bytes, sig_len, sig, nonce_out, status);
    return hipGetLastError();
}

}  // namespace frw
