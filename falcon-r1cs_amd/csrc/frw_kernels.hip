// frw_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Falcon-verification R1CS witness engine.
//
// What the path is (reference citations relative to /root/reference/):
//   FalconNTTVerificationCircuit::generate_constraints, falcon-r1cs/src/circuits/falcon_ntt.rs:26-123,
//   with the gadgets it drives (poly.rs:104-159, arithmetics.rs:105-149,214-262,
//   range_proofs.rs:42-94,100-186,192-272,289-333, misc.rs:9-51).
// The circuit is the same for every signature, so the witness is a closed-form function of
// (sig, pk, hm); this file evaluates it for a batch, one workgroup (4 wavefronts) per signature.
//
// Shape of the work: 6 KB of input becomes 5.0 MB of output per Falcon-1024 signature, 91 % of it
// field elements that are 0 or 1.  The kernel is an HBM write stream (measured in round 2, driver's box: 1.00 of a
// compute-free write stream on the same device, 6.85 TB/s = 0.86 of the 8 TB/s spec -- 0.79-0.86 over the boxes of the
// pool --, HBM traffic = 1.0005 x the algorithmic bytes); the integer work (mod-q NTTs, the un-reduced 160-bit butterfly ladder, short
// divisions, Montgomery conversions, the tile writer's 3 vector instructions per store) lives in LDS/registers
// and overlaps with the stores of the other resident wavefronts.  No MFMA: nothing here is GEMM-shaped.
//
// Kernels in this file: witness_ntt_verify_kernel (the hot path; ENC = 2 writes the compact encoding),
// witness_dual_ntt_verify_kernel (the signed-split variant, falcon_dual_ntt.rs), ntt_modq_kernel (ntt_circuit alone),
// expand_kernel (compact -> arkworks layout), gadget_kernel (the gadgets called on their own), digest_kernel and
// write_stream_kernel (verification / calibration utilities).
//
// Data flow of one workgroup
//   1. inputs -> LDS (u16), range check
//   2. mod-q NTT of sig, pk, hm; v_ntt = hm_ntt - sig_ntt*pk_ntt; v = INTT(v_ntt)        (falcon_ntt.rs:44-51)
//   3. "small" segments S0,S1,S2,S5,S6,S7 + instance vector, written tile by tile
//   4. ladder(sig) in LDS (limb-major u32[5][N])  -> S3 tiles  (poly.rs:113-156)
//   5. ladder(v)   in the same LDS                -> S4 tiles
// A tile = 64 gadget blocks = one wavefront: lane k computes block k's few non-boolean elements
// (converted to the field encoding, parked in a per-wave LDS slab) and a <=30-bit mask of its
// boolean elements; the wave then walks the tile's bytes in order, every lane producing 16 B per
// store instruction, so each buffer_store_dwordx4 writes 1 KiB of contiguous HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <mutex>
#include "frw_device.h"

namespace frw {

typedef uint32_t v4u __attribute__((ext_vector_type(4)));     // 16 B = one lane's share of a store instruction
__device__ __forceinline__ v4u mk4(uint32_t a, uint32_t b, uint32_t c, uint32_t d) { v4u r = {a, b, c, d}; return r; }

// ------------------------------------------------------------------------------------------------
// BLS12-381 scalar field, 32-bit little-endian limbs (ark-ff Fp256; gadgets/poly.rs:244)
// ------------------------------------------------------------------------------------------------
#define FRW_P32  {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u}
// R  = 2^256 mod p  (Montgomery form of 1)
#define FRW_R32  {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau, 0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u}
// K1 = 2^288 mod p: REDC_1(x * K1) = x * 2^256 mod p for a one-limb x
#define FRW_K1   {0xcaaf6b13u, 0x355094eau, 0x69a568efu, 0xf6b10cb3u, 0x40cc3869u, 0xe2c926a6u, 0xed269aadu, 0x736a6d3bu}
// K5 = 2^416 mod p: REDC_5(x * K5) = x * 2^256 mod p for a five-limb x
#define FRW_K5   {0x9afbc14cu, 0x5d23afe0u, 0x2b0e40d0u, 0x1deef9adu, 0x203ba106u, 0xe07e784du, 0xa251b319u, 0x562ca75au}
// -p^-1 mod 2^32 = 0xffffffff, i.e. the Montgomery quotient digit is simply -T[0].

// One CIOS round: T (9 limbs) <- (T + x*K + m*p) / 2^32.
template <typename KArr>
__device__ __forceinline__ void cios_round(uint32_t (&T)[9], uint32_t x, const KArr &K)
{
    constexpr uint32_t P[8] = FRW_P32;
    uint64_t acc;
    uint32_t c = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        acc = (uint64_t)x * K[j] + T[j] + c;
        T[j] = (uint32_t)acc;
        c = (uint32_t)(acc >> 32);
    }
    acc = (uint64_t)T[8] + c;
    T[8] = (uint32_t)acc;
    uint32_t t9 = (uint32_t)(acc >> 32);
    uint32_t m = 0u - T[0];
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

// T < 2p on entry (T[8] == 0); subtract p once if T >= p.
__device__ __forceinline__ void cond_sub_p(uint32_t (&T)[9], uint32_t (&out)[8])
{
    constexpr uint32_t P[8] = FRW_P32;
    uint32_t d[8];
    uint32_t bw = 0;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        uint64_t x = (uint64_t)T[j] - P[j] - bw;
        d[j] = (uint32_t)x;
        bw = (uint32_t)(x >> 63);
    }
#pragma unroll
    for (int j = 0; j < 8; j++) out[j] = bw ? T[j] : d[j];
}

// integer < 2^32  ->  field element in the requested encoding
template <int ENC>
__device__ __forceinline__ void encode_u32(uint32_t x, uint32_t (&out)[8])
{
#if defined(FRW_FAKE_ENCODE)      // timing experiments only (tools/ab_variants.py): what would cheaper encodes buy?
    if (true) {
#else
    if (ENC == 0) {
#endif
        out[0] = x;
#pragma unroll
        for (int j = 1; j < 8; j++) out[j] = 0;
    } else {
        constexpr uint32_t K[8] = FRW_K1;
        uint32_t T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        cios_round(T, x, K);
        cond_sub_p(T, out);
    }
}

// integer < 2^160 (five limbs)  ->  field element
template <int ENC>
__device__ __forceinline__ void encode_u160(const uint32_t (&x)[5], uint32_t (&out)[8])
{
#if defined(FRW_FAKE_ENCODE)
    if (true) {
#else
    if (ENC == 0) {
#endif
#pragma unroll
        for (int j = 0; j < 5; j++) out[j] = x[j];
        out[5] = out[6] = out[7] = 0;
    } else {
        constexpr uint32_t K[8] = FRW_K5;
        uint32_t T[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < 5; i++) cios_round(T, x[i], K);
        cond_sub_p(T, out);
    }
}

// ------------------------------------------------------------------------------------------------
// small helpers
// ------------------------------------------------------------------------------------------------
// Workgroup barrier.  __syncthreads() also drains the wave's outstanding global stores (s_waitcnt vmcnt(0)) before every
// s_barrier, although nothing in these kernels hands global data from one wave to another; FRW_FULL_BARRIER=0 swaps in
// an LDS-only barrier (s_waitcnt lgkmcnt(0); s_barrier) that lets a tile's stores drain behind the next ladder.
// Measured (round 2, tools/ab_variants.py lds= full=-DFRW_FULL_BARRIER=1: 32,768 Falcon-1024, 8,192 Falcon-512 signatures,
// 4,096 ntt_modq polynomials): no difference beyond 0.1 % -- the kernels wait for HBM either way -- so the plain
// __syncthreads() stays.
#ifndef FRW_FULL_BARRIER
#define FRW_FULL_BARRIER 1
#endif
__device__ __forceinline__ void lds_barrier()
{
#if FRW_FULL_BARRIER
    __syncthreads();
#else
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
}

__device__ __forceinline__ uint32_t mod_q_u32(uint32_t x) { return x % Q; }   // constant divisor: mul_hi + fixups

// a (five limbs) = t*q + b: schoolbook short division in 16-bit steps (arithmetics.rs:127-134)
__device__ __forceinline__ uint32_t divmod_q_u160(const uint32_t (&a)[5], uint32_t (&t)[5])
{
    uint32_t r = 0;
#pragma unroll
    for (int i = 4; i >= 0; i--) {
        uint32_t cur = (r << 16) | (a[i] >> 16);          // r < q < 2^14  ->  cur < 2^30
        uint32_t qh = cur / Q;
        r = cur - qh * Q;
        cur = (r << 16) | (a[i] & 0xffffu);
        uint32_t ql = cur / Q;
        r = cur - ql * Q;
        t[i] = (qh << 16) | ql;
    }
    return r;
}

// enforce_less_than_q's 27 booleans as a bit mask (range_proofs.rs:62-89):
//   bits 0..13 a0..a13 | bits 14..24 w0..w10 = fold-left OR of a0..a11 | bit 25 w10&a12 | bit 26 bit25&a13
__device__ __forceinline__ uint32_t ltq_mask(uint32_t a)
{
    uint32_t p = a & 0xfffu;
    p |= p << 1; p |= p << 2; p |= p << 4; p |= p << 8;       // bit i = OR(a0..ai)
    uint32_t w = (p >> 1) & 0x7ffu;                            // w0..w10
    uint32_t w11 = (w >> 10) & (a >> 12) & 1u;
    uint32_t w12 = w11 & (a >> 13) & 1u;
    return (a & 0x3fffu) | (w << 14) | (w11 << 25) | (w12 << 26);
}

// is_less_than_6144's 16 booleans (range_proofs.rs:304-328): a0..a13, w0 = a11&a12, w1 = nor(a13, w0)
__device__ __forceinline__ uint32_t lt6144_mask(uint32_t a)
{
    uint32_t w0 = (a >> 11) & (a >> 12) & 1u;
    uint32_t w1 = (~(a >> 13)) & (~w0) & 1u;
    return (a & 0x3fffu) | (w0 << 14) | (w1 << 15);
}

// enforce_less_than_norm_bound_{512,1024}: bits then gates in allocation order (range_proofs.rs:100-186, 192-272)
__device__ uint64_t norm_mask_512(uint64_t a)
{
    auto b = [a](int i) -> uint32_t { return (uint32_t)(a >> i) & 1u; };
    uint32_t g[24];
    g[0] = b(19) | b(20); g[1] = g[0] | b(21); g[2] = g[1] | b(22); g[3] = g[2] | b(23); g[4] = g[3] | b(24);
    g[5] = b(16) & b(17); g[6] = g[5] & b(18);
    g[7] = b(6) | b(7); g[8] = g[7] | b(8); g[9] = g[8] | b(9);
    g[10] = b(3) | b(4);
    g[11] = b(1) & b(2);
    g[12] = (g[10] | g[11]) ^ 1u;
    g[13] = b(5) & (g[12] ^ 1u);
    g[14] = (g[9] | g[13]) ^ 1u;
    g[15] = b(10) & (g[14] ^ 1u);
    g[16] = (b(11) | g[15]) ^ 1u;
    g[17] = b(12) & (g[16] ^ 1u);
    g[18] = (b(13) | g[17]) ^ 1u;
    g[19] = b(14) & (g[18] ^ 1u);
    g[20] = (b(15) | g[19]) ^ 1u;
    g[21] = g[6] & (g[20] ^ 1u);
    g[22] = (g[4] | g[21]) ^ 1u;
    g[23] = b(25) & (g[22] ^ 1u);
    uint64_t m = a & ((1ull << 26) - 1);
#pragma unroll
    for (int i = 0; i < 24; i++) m |= (uint64_t)g[i] << (26 + i);
    return m;
}

__device__ uint64_t norm_mask_1024(uint64_t a)
{
    auto b = [a](int i) -> uint32_t { return (uint32_t)(a >> i) & 1u; };
    uint32_t g[25];
    g[0] = b(22) | b(23); g[1] = g[0] | b(24); g[2] = g[1] | b(25);
    g[3] = b(20) & b(21);
    g[4] = b(14) | b(15); g[5] = g[4] | b(16); g[6] = g[5] | b(17); g[7] = g[6] | b(18); g[8] = g[7] | b(19);
    g[9] = b(9) | b(10);
    g[10] = b(7) & b(8);
    g[11] = b(5) | b(6);
    g[12] = b(3) & b(4);
    g[13] = b(1) | b(2);
    g[14] = g[13] & g[12];
    g[15] = (g[11] | g[14]) ^ 1u;
    g[16] = g[10] & (g[15] ^ 1u);
    g[17] = (g[9] | g[16]) ^ 1u;
    g[18] = b(11) & (g[17] ^ 1u);
    g[19] = (b(12) | g[18]) ^ 1u;
    g[20] = b(13) & (g[19] ^ 1u);
    g[21] = (g[8] | g[20]) ^ 1u;
    g[22] = g[3] & (g[21] ^ 1u);
    g[23] = (g[2] | g[22]) ^ 1u;
    g[24] = b(26) & (g[23] ^ 1u);
    uint64_t m = a & ((1ull << 27) - 1);
#pragma unroll
    for (int i = 0; i < 25; i++) m |= (uint64_t)g[i] << (27 + i);
    return m;
}

// ------------------------------------------------------------------------------------------------
// mod-q NTT / inverse NTT on u16 arrays in LDS, whole workgroup (Falcon mq_NTT schedule == poly.rs:115-149)
// ------------------------------------------------------------------------------------------------
template <int LOGN, int NPOLY>
__device__ __forceinline__ void ntt_modq_lds(uint16_t *const (&a)[NPOLY], const uint16_t *tw, int tid)
{
    constexpr int N = 1 << LOGN;
#pragma unroll 1
    for (int l = 0; l < LOGN; l++) {
        const int sh = LOGN - 1 - l;            // log2(ht)
        const int ht = 1 << sh;
        for (int bf = tid; bf < N / 2; bf += BLOCK) {
            const int i = bf >> sh;
            const int j = ((i << sh) << 1) + (bf & (ht - 1));
            const uint32_t s = tw[(1 << l) + i];
#pragma unroll
            for (int p = 0; p < NPOLY; p++) {
                uint32_t u = a[p][j];
                uint32_t v = mod_q_u32(a[p][j + ht] * s);
                uint32_t x = u + v;
                uint32_t y = u + Q - v;
                a[p][j] = (uint16_t)(x >= Q ? x - Q : x);
                a[p][j + ht] = (uint16_t)(y >= Q ? y - Q : y);
            }
        }
        lds_barrier();
    }
}

template <int LOGN>
__device__ __forceinline__ void intt_modq_lds(uint16_t *a, const uint16_t *itw, int tid)
{
    constexpr int N = 1 << LOGN;
    constexpr uint32_t NINV = LOGN == 9 ? 12265u : 12277u;     // N^-1 mod q
#pragma unroll 1
    for (int l = LOGN - 1; l >= 0; l--) {
        const int sh = LOGN - 1 - l;
        const int ht = 1 << sh;
        for (int bf = tid; bf < N / 2; bf += BLOCK) {
            const int i = bf >> sh;
            const int j = ((i << sh) << 1) + (bf & (ht - 1));
            const uint32_t s = itw[(1 << l) + i];
            uint32_t x = a[j], y = a[j + ht];
            uint32_t u = x + y;
            uint32_t w = mod_q_u32((x + Q - y) * s);
            a[j] = (uint16_t)(u >= Q ? u - Q : u);
            a[j + ht] = (uint16_t)w;
        }
        lds_barrier();
    }
    for (int j = tid; j < N; j += BLOCK) a[j] = (uint16_t)mod_q_u32(a[j] * NINV);
    lds_barrier();
}

// ------------------------------------------------------------------------------------------------
// the un-reduced butterfly ladder over the integers (poly.rs:113-149), limb-major u32[5][N] in LDS.
// Round l reads LIN limbs and writes LOUT: values entering round l are < B_l, B_0 = q,
// B_{l+1} = B_l + C_{l+1}, C_k = 2^k q^(k+1) (falcon_ntt.rs:31-39): 29,43,58,72,87,102,116,131,145,160 bits.
// ------------------------------------------------------------------------------------------------
template <int LOGN, int L, int LIN, int LOUT>
__device__ __forceinline__ void ladder_round(uint32_t *lad, const uint16_t *tw, const uint32_t *ck, int tid)
{
    constexpr int N = 1 << LOGN;
    constexpr int sh = LOGN - 1 - L;
    constexpr int ht = 1 << sh;
    uint32_t c[LOUT];
#pragma unroll
    for (int k = 0; k < LOUT; k++) c[k] = ck[(L + 1) * 5 + k];
    for (int bf = tid; bf < N / 2; bf += BLOCK) {
        const int i = bf >> sh;
        const int j = ((i << sh) << 1) + (bf & (ht - 1));
        const uint32_t s = tw[(1 << L) + i];
        uint32_t u[LOUT], v[LOUT];
#pragma unroll
        for (int k = 0; k < LOUT; k++) u[k] = k < LIN ? lad[k * N + j] : 0u;
        // v = out[j+ht] * s                                                   poly.rs:136
        uint32_t carry = 0;
#pragma unroll
        for (int k = 0; k < LIN; k++) {
            uint64_t acc = (uint64_t)lad[k * N + j + ht] * s + carry;
            v[k] = (uint32_t)acc;
            carry = (uint32_t)(acc >> 32);
        }
        if constexpr (LOUT > LIN) v[LIN] = carry;
        // out[j] = u + v ; out[j+ht] = u + (C_{l+1} - v)                      poly.rs:137-142
        uint32_t cy = 0, bw = 0, cy2 = 0;
#pragma unroll
        for (int k = 0; k < LOUT; k++) {
            uint64_t x = (uint64_t)u[k] + v[k] + cy;
            cy = (uint32_t)(x >> 32);
            uint64_t d = (uint64_t)c[k] - v[k] - bw;
            bw = (uint32_t)(d >> 63);
            uint64_t y = (uint64_t)u[k] + (uint32_t)d + cy2;
            cy2 = (uint32_t)(y >> 32);
            lad[k * N + j] = (uint32_t)x;
            lad[k * N + j + ht] = (uint32_t)y;
        }
    }
    lds_barrier();
}

template <int LOGN>
__device__ __forceinline__ void ladder_lds(uint32_t *lad, const uint16_t *in, const uint16_t *tw, const uint32_t *ck, int tid)
{
    constexpr int N = 1 << LOGN;
    for (int j = tid; j < N; j += BLOCK) lad[j] = in[j];
    lds_barrier();
    ladder_round<LOGN, 0, 1, 1>(lad, tw, ck, tid);
    ladder_round<LOGN, 1, 1, 2>(lad, tw, ck, tid);
    ladder_round<LOGN, 2, 2, 2>(lad, tw, ck, tid);
    ladder_round<LOGN, 3, 2, 3>(lad, tw, ck, tid);
    ladder_round<LOGN, 4, 3, 3>(lad, tw, ck, tid);
    ladder_round<LOGN, 5, 3, 4>(lad, tw, ck, tid);
    ladder_round<LOGN, 6, 4, 4>(lad, tw, ck, tid);
    ladder_round<LOGN, 7, 4, 5>(lad, tw, ck, tid);
    ladder_round<LOGN, 8, 5, 5>(lad, tw, ck, tid);
    if constexpr (LOGN == 10) ladder_round<LOGN, 9, 5, 5>(lad, tw, ck, tid);
}

// ------------------------------------------------------------------------------------------------
// tile writer.  A tile is 64 consecutive gadget blocks of BLK field elements; block k belongs to lane k.
// Element `pos` of a block is either one of the block's NVAL non-boolean values (encoded by lane k and
// parked in the wave's LDS slab as 16-byte halves) or a boolean taken from bit `pos` of lane k's mask.
// The wave emits the tile's 64*BLK*32 bytes in address order, 16 B per lane = 1 KiB per store instruction.
//
// Round-2 form (3 vector instructions per store; round 1 needed ~26):
//  * the tile's booleans are first packed into a bit stream in block order, 32 bits per lane (lane i holds
//    stream bits [32 i, 32 i + 32): three ds_bpermute + shifts, once per tile).  Store `it` covers elements
//    32 it .. 32 it + 31, i.e. exactly dword `it` of the stream: v_readlane -> SGPR, s_bitreplicate_b64_b32
//    doubles every bit (two lanes per element) and the result IS the lane mask of a v_cndmask.
//  * every lane then reads its 16 bytes from LDS: value lanes from the slab, boolean lanes from constants woven into
//    the slab (zero in slot 0, this half of one in slot 1: `bit ? + slot stride : + 0`).  Which address a lane reads depends only
//    on (store index mod period, lane) -- the pattern of value positions repeats every lcm(64, 2 BLK) chunks -- so it
//    comes from a small per-workgroup table (ds_read_u16, immediate offset) and one v_add3 adds the wave's slab base.
//  * the store is a buffer_store_dwordx4 with the signature's buffer resource, the lane's 16-byte offset in a VGPR and
//    the running tile offset in an SGPR: no vector address arithmetic at all.
// ------------------------------------------------------------------------------------------------
#define FRW_LDS __attribute__((address_space(3)))

// LDS geometry of the value slabs.  A (slot, wave, half) row holds the 64 lanes' 16-byte halves in four groups of 16,
// each group followed by a 16-byte constant: slab_addr(k) = (k / 16) * SLAB_GRP + (k % 16) * 16.  The constants of
// slot 0 are zero, those of slot 1 are this half of the field element one, so a boolean lane reads
// row(slot 0) + 256 + (bit ? SLAB_SLOT : 0) -- and because the constants repeat with the group stride, adding the
// per-period block advance (a multiple of 16 blocks = SLAB_GRP bytes) to EVERY lane's address moves value lanes to the
// next blocks and boolean lanes to another copy of the same constant.  All rows of one slot are contiguous, so what a
// wave reads is at a wave-independent offset from B_w = slab + wave * SLAB_WBLK.  Slots 0 and 1 are the slab proper;
// slot 2 (S5 only) aliases the first SLAB_SLOT bytes of what follows the slab in LDS (the ladder array, idle while
// the small segments are written).
constexpr int SLAB_GRP = 16 * 16 + 16;             // 272
constexpr int SLAB_HALF = 4 * SLAB_GRP;            // 1,088
constexpr int SLAB_WBLK = 2 * SLAB_HALF;           // 2,176
constexpr int SLAB_SLOT = WAVES * SLAB_WBLK;       // 8,704
constexpr int SLAB_BYTES = 2 * SLAB_SLOT;          // 17,408
__device__ __forceinline__ constexpr int slab_addr(int k) { return (k >> 4) * SLAB_GRP + (k & 15) * 16; }

// Period structure of a tile shape: the (block, position) pattern of chunk 64 j + lane repeats after P stores, during
// which the block index advances by KSTEP.
template <int BLK> struct TileShape;
template <> struct TileShape<29> { static constexpr int P = 29, NPER = 2, KSTEP = 32, NVAL = 2, VFIRST = 0, ROW0 = 0; };   // mod_q block
template <> struct TileShape<30> { static constexpr int P = 15, NPER = 4, KSTEP = 16, NVAL = 3, VFIRST = 0, ROW0 = 29; };  // pointwise block
template <> struct TileShape<18> { static constexpr int P = 9, NPER = 4, KSTEP = 16, NVAL = 2, VFIRST = 16, ROW0 = 44; };  // l2 block
constexpr int VTAB_ROWS = 53;
static_assert(TileShape<29>::P * TileShape<29>::NPER == 58 && TileShape<30>::P * TileShape<30>::NPER == 60 &&
              TileShape<18>::P * TileShape<18>::NPER == 36, "a tile is 2 BLK stores");

// vtab[row][lane] = LDS byte offset (relative to B_w) of the 16 bytes lane `lane` stores in store `row - ROW0` of a
// period when the element's boolean is 0.  Filled once per workgroup (the kernels are persistent).
__device__ __forceinline__ void init_vtab(uint16_t *vtab, int rows, int tid)
{
    for (int idx = tid; idx < rows * WAVE; idx += BLOCK) {
        const int row = idx >> 6, lane = idx & 63;
        int blk, nval, vfirst, j;
        if (row < TileShape<30>::ROW0) { blk = 29; nval = 2; vfirst = 0; j = row; }
        else if (row < TileShape<18>::ROW0) { blk = 30; nval = 3; vfirst = 0; j = row - TileShape<30>::ROW0; }
        else { blk = 18; nval = 2; vfirst = 16; j = row - TileShape<18>::ROW0; }
        const int c = j * WAVE + lane, e = c >> 1, h = c & 1;
        const int k = e / blk, slot = e - k * blk - vfirst;
        vtab[idx] = (uint16_t)((unsigned)slot < (unsigned)nval ? slot * SLAB_SLOT + h * SLAB_HALF + slab_addr(k)
                                                               : h * SLAB_HALF + 256);
    }
}

// the wave's constants: 16 lanes write the four group constants of (slot 0 | 1, half 0 | 1): zero | this half of one
template <int ENC>
__device__ __forceinline__ void init_slab_const(uint32_t slab_w, int lane)
{
    constexpr uint32_t R[8] = FRW_R32;
    if (lane < 16) {
        const int slot = lane >> 3, h = (lane >> 2) & 1, g = lane & 3;
        v4u c = mk4(0, 0, 0, 0);
        if (slot == 1) {
            if (ENC == 0) c = h ? mk4(0, 0, 0, 0) : mk4(1, 0, 0, 0);
            else c = h ? mk4(R[4], R[5], R[6], R[7]) : mk4(R[0], R[1], R[2], R[3]);
        }
        *(FRW_LDS v4u *)(uintptr_t)(slab_w + slot * SLAB_SLOT + h * SLAB_HALF + g * SLAB_GRP + 256) = c;
    }
}

// what a wave needs to emit tiles
struct WaveCtx {
    uint32_t slab;       // LDS byte address of B_w
    uint32_t vtab;       // LDS byte address of vtab[0][lane]
    int lane;
};

__device__ __forceinline__ void slab_put(uint32_t slab_w, int slot, int lane, const uint32_t (&e)[8])
{
    const uint32_t a = slab_w + slot * SLAB_SLOT + slab_addr(lane);
    *(FRW_LDS v4u *)(uintptr_t)a = mk4(e[0], e[1], e[2], e[3]);
    *(FRW_LDS v4u *)(uintptr_t)(a + SLAB_HALF) = mk4(e[4], e[5], e[6], e[7]);
}

// Output store.  Round 1 measured plain stores ahead of non-temporal ones (+2.8 % N=1024, +5 % N=512); FRW_STORE_AUX
// sets the cache-policy bits of the buffer store for A/B builds (tools/ab_variants.py), FRW_NO_STORE compiles the
// stores out (instruction time only).
#ifndef FRW_STORE_AUX
#define FRW_STORE_AUX 0
#endif
__device__ __forceinline__ void tile_store(v4u val, __amdgpu_buffer_rsrc_t rsrc, int voff, uint32_t soff)
{
#if defined(FRW_NO_STORE)
    asm volatile("" ::"v"(val), "v"(voff), "s"(soff));
#else
    __builtin_amdgcn_raw_buffer_store_b128(val, rsrc, voff, (int)soff, FRW_STORE_AUX);
    // Measured on MI355X (round 2, tools/dev/diag_verify.py): a VALU write to the data registers in the instruction
    // right after a 128-bit buffer store corrupts some lanes of the stored data, also when `soffset` is an SGPR -- the
    // case the compiler's hazard recogniser exempts (it pads only the immediate-soffset form).  The asm reads `val`, so
    // the registers stay allocated up to the two wait states it provides; tests/test_isa_hazards.py checks the
    // generated code.
    asm volatile("s_nop 1" ::"v"(val));
#endif
}

// buffer resource over [base, base + bytes): raw buffer, no swizzle, 32-bit data format (word 3 = 0x00020000 on gfx9-family)
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void *base, uint32_t bytes)
{
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ unsigned long long bit_double(uint32_t s)
{
    unsigned long long m;
    asm("s_bitreplicate_b64_b32 %0, %1" : "=s"(m) : "s"(s));
    return m;
}

__device__ __forceinline__ void lds_fence()
{
    __builtin_amdgcn_s_waitcnt(0xc07f);   // lgkmcnt(0): same wave, a wait on its LDS operations is enough
    __builtin_amdgcn_wave_barrier();
}

// lane i <- bits [32 i, 32 i + 32) of the concatenation of the 64 lanes' BLK-bit masks (upper mask bits must be 0)
template <int BLK>
__device__ __forceinline__ uint32_t pack_bits(uint32_t mask, int lane)
{
    static_assert(BLK >= 16 && BLK <= 30, "a 32-bit window covers at most three blocks");
    const int k0 = (lane * 32) / BLK, o = lane * 32 - k0 * BLK;
    const uint32_t m0 = (uint32_t)__shfl((int)mask, k0 & 63, WAVE);
    const uint32_t m1 = (uint32_t)__shfl((int)mask, (k0 + 1) & 63, WAVE);
    const uint32_t m2 = (uint32_t)__shfl((int)mask, (k0 + 2) & 63, WAVE);
    uint32_t d = (m0 >> o) | (m1 << (BLK - o));
    if (2 * BLK - o < 32) d |= m2 << (2 * BLK - o);
    return d;
}

// a tile of 64 one-element blocks (S0, S1, the instance vector): lane k's element is in slot 0
__device__ __forceinline__ void emit_values(__amdgpu_buffer_rsrc_t rsrc, uint32_t soff, const WaveCtx &w)
{
    lds_fence();
#pragma unroll
    for (int it = 0; it < 2; it++) {
        const v4u val = *(const FRW_LDS v4u *)(uintptr_t)(w.slab + (w.lane & 1) * SLAB_HALF + slab_addr((w.lane >> 1) + 32 * it));
        tile_store(val, rsrc, w.lane * 16, soff + it * 1024);
    }
    lds_fence();                           // the slab is rewritten by the next tile
}

// a tile of 64 all-zero elements (the dual circuit's pos*neg products)
__device__ __forceinline__ void emit_zeros(__amdgpu_buffer_rsrc_t rsrc, uint32_t soff, const WaveCtx &w)
{
    tile_store(mk4(0, 0, 0, 0), rsrc, w.lane * 16, soff);
    tile_store(mk4(0, 0, 0, 0), rsrc, w.lane * 16, soff + 1024);
}

// a tile of 64 blocks of BLK elements; booleans in `mask` (bit pos = element pos, zeros at value positions)
template <int BLK>
__device__ __forceinline__ void emit_tile(__amdgpu_buffer_rsrc_t rsrc, uint32_t soff, uint32_t mask, const WaveCtx &w)
{
    lds_fence();                           // the slab writes of all lanes have landed
    const uint32_t bits = pack_bits<BLK>(mask, w.lane);
    const int voff = w.lane * 16;
    if constexpr (BLK == 27) {             // enforce_less_than_q on its own: booleans only
        const uint32_t a0 = w.slab + (w.lane & 1) * SLAB_HALF + 256, a1 = a0 + SLAB_SLOT;
#pragma unroll 1
        for (int g = 0; g < 6; g++) {
#pragma unroll
            for (int j = 0; j < 9; j++) {
                const int it = g * 9 + j;
                const bool b = __builtin_amdgcn_inverse_ballot_w64(bit_double(__builtin_amdgcn_readlane(bits, it)));
                const v4u val = *(const FRW_LDS v4u *)(uintptr_t)(b ? a1 : a0);
                tile_store(val, rsrc, voff, soff + it * 1024);
            }
        }
    } else {
        using S = TileShape<BLK>;
        const uint32_t trow = w.vtab + S::ROW0 * WAVE * 2;
        uint32_t base = w.slab;
#pragma unroll 1
        for (int per = 0; per < S::NPER; per++) {
#pragma unroll
            for (int j = 0; j < S::P; j++) {
                const int it = per * S::P + j;
                const bool b = __builtin_amdgcn_inverse_ballot_w64(bit_double(__builtin_amdgcn_readlane(bits, it)));
                const uint32_t a = *(const FRW_LDS uint16_t *)(uintptr_t)(trow + j * WAVE * 2);
                const v4u val = *(const FRW_LDS v4u *)(uintptr_t)(a + (b ? (uint32_t)SLAB_SLOT : 0u) + base);
                tile_store(val, rsrc, voff, soff + it * 1024);
                if (j % 8 == 7) __builtin_amdgcn_sched_barrier(0);     // bound the scheduler's hoisting (registers)
            }
            base += (S::KSTEP / 16) * SLAB_GRP;
        }
    }
    lds_fence();                           // the slab is rewritten by the next tile: all lanes have finished reading it
}

// ------------------------------------------------------------------------------------------------
// generic tile writer of the stand-alone gadget kernel (ragged counts, 64-bit masks; not a throughput path): lane l of
// iteration `it` produces chunk 64 it + l, fetching block k's mask with one ds_bpermute and value halves from a
// [slot][half][lane] slab.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void gslab_put(v4u *slab, int slot, int lane, const uint32_t (&e)[8])
{
    slab[(slot * 2 + 0) * WAVE + lane] = mk4(e[0], e[1], e[2], e[3]);
    slab[(slot * 2 + 1) * WAVE + lane] = mk4(e[4], e[5], e[6], e[7]);
}

template <typename M>
__device__ __forceinline__ M lane_read(M v, int k)
{
    if constexpr (sizeof(M) == 4) return (M)__shfl((int)v, k, WAVE);
    else return (M)__shfl((unsigned long long)v, k, WAVE);
}

template <int ENC, int BLK, int NVAL, int VFIRST, typename MASK>
__device__ __forceinline__ void emit_tile_generic(v4u *__restrict__ out, const v4u *slab, MASK mask, int lane, int nchunks)
{
    constexpr uint32_t R[8] = FRW_R32;
    const int half = lane & 1;
    v4u one;
    if (ENC == 0) one = half ? mk4(0, 0, 0, 0) : mk4(1, 0, 0, 0);
    else one = half ? mk4(R[4], R[5], R[6], R[7]) : mk4(R[0], R[1], R[2], R[3]);
    lds_fence();
    for (int it = 0; it < 2 * BLK; it++) {
        const int e = it * 32 + (lane >> 1);          // element index inside the tile
        const int k = (e / BLK) & (WAVE - 1);         // owning block == owning lane
        const int pos = e - (e / BLK) * BLK;
        const MASK mk = lane_read(mask, k);
        v4u val = (uint32_t)(mk >> pos) & 1u ? one : mk4(0, 0, 0, 0);
        if (NVAL > 0) {
            const int slot = pos - VFIRST;
            const bool isval = (unsigned)slot < (unsigned)NVAL;
            const v4u lv = slab[((isval ? slot : 0) * 2 + half) * WAVE + k];
            if (isval) val = lv;
        }
        if (it * WAVE + lane < nchunks) out[it * WAVE + lane] = val;
    }
    lds_fence();
}

// ------------------------------------------------------------------------------------------------
// work distribution.  A launch is a persistent grid (the resident workgroups) with STATIC striding: workgroup b takes
// signatures b, b + grid, b + 2 grid, ...  All workgroups then advance through their 2.5 / 5 MB units in step, which is
// the access pattern the HBM write path likes best: measured in round 2 (profiles/r02_scheduling_ab.txt, one process,
// interleaved) a 15,360-signature Falcon-1024 launch (20 exact rounds of 768) runs at 6,412 GB/s = 0.99 of the
// compute-free write stream on the same device, the per-launch atomic work queue of round 1 at 5,987 GB/s.
// What static striding cannot absorb is a ragged last round, so the signatures beyond the last full round (and every
// signature of a batch too small to fill the grid five times over) are cut into five work items of ~30 N elements
// each -- {instance, S0, S1, S2}, {S5}, {S6, S7, status}, {S3}, {S4} -- strided over the grid the same way, each
// recomputing the part of the clear arithmetic it needs (a single Falcon-1024 witness: 167 us -> ~60 us).
// ------------------------------------------------------------------------------------------------
constexpr int PARTS = 5;

// ------------------------------------------------------------------------------------------------
// LDS carve-up of one workgroup.  The NTT-domain arrays are dead once the small segments are written and the ladder
// array is dead until then, so they share memory: behind the third value slot (first SLAB_SLOT bytes of `lad`) as many
// of them as fit live inside `lad`.  Falcon-1024: 50.3 KB -> 3 workgroups per CU; Falcon-512: 40.1 KB -> 4.
// ------------------------------------------------------------------------------------------------
template <int LOGN, bool COMPACT = false>
struct alignas(16) Smem {
    static constexpr int N = 1 << LOGN;
    static constexpr int IN_LAD = (5 * N * 4 - SLAB_SLOT) / (2 * N) < 4 ? (5 * N * 4 - SLAB_SLOT) / (2 * N) : 4;   // 4 | 1
    unsigned char slab[COMPACT ? 16 : SLAB_BYTES];   // value slots 0, 1 of the four waves + their constants (no tile writer in compact mode)
    uint32_t lad[5 * N];                 // ladder integers, limb-major; before the ladders: slot 2, then NTT-domain arrays
    uint16_t tw[N];
    uint16_t sig[N], v[N];               // coefficient domain
    uint16_t ntt_rest[(4 - IN_LAD) * N + 8];     // the NTT-domain arrays that do not fit inside `lad`
    uint16_t vtab[COMPACT ? 8 : VTAB_ROWS * WAVE];
    unsigned long long norm;
    int bad;
#if defined(FRW_LDS_PAD) && FRW_LDS_PAD > 0
    uint32_t pad[FRW_LDS_PAD / 4];        // occupancy experiments only (tools/ab_variants.py)
#endif
    // NTT-domain array i of {nsig, npk, nhm, nv}, all reduced mod q; nv first so that it is the one inside `lad` at N = 512
    __device__ __forceinline__ uint16_t *ntt_arr(int i)
    {
        uint16_t *in_lad = (uint16_t *)((unsigned char *)lad + SLAB_SLOT);
        return i < IN_LAD ? in_lad + i * N : ntt_rest + (i - IN_LAD) * N;
    }
};

// a rejected signature (coefficient >= q: the reference panics, range_proofs.rs:57-60) leaves zeros, not stale memory
__device__ __forceinline__ void zero_fill(v4u *p, size_t chunks, int tid)
{
    for (size_t i = tid; i < chunks; i += BLOCK) p[i] = mk4(0, 0, 0, 0);
}

// ------------------------------------------------------------------------------------------------
// kernel: full verify-with-ntt witness (falcon_ntt.rs:26-123)
// ------------------------------------------------------------------------------------------------
// Work items [0, full) are whole signatures; items full + j are part j % 5 of signature full + j / 5 (see above).
template <int LOGN, int ENC>
__global__ __launch_bounds__(BLOCK) void witness_ntt_verify_kernel(
    const Tables *__restrict__ tab, size_t batch, size_t full,
    const uint16_t *__restrict__ g_sig, const uint16_t *__restrict__ g_pk, const uint16_t *__restrict__ g_hm,
    v4u *__restrict__ g_wit, v4u *__restrict__ g_inst, int32_t *__restrict__ g_status)
{
    constexpr int N = 1 << LOGN;
    constexpr int NB = LOGN == 9 ? 50 : 52;
    constexpr size_t W = 153 * (size_t)N + NB;
    constexpr size_t I = 2 * (size_t)N + 1;
    constexpr int TILES = N / WAVE;
    constexpr uint32_t TILE1 = WAVE * 32;            // bytes of a tile of one-element blocks
    constexpr bool COMPACT = ENC == 2;               // FRW_ENC_COMPACT: g_wit is the compact buffer, g_inst unused
    constexpr int VENC = COMPACT ? 1 : ENC;          // (compact mode stores plain integers and encodes nothing)
    constexpr CompactLayout CL = compact_layout(LOGN);
    __shared__ Smem<LOGN, COMPACT> sm;

    const int tid = threadIdx.x;
    const int lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    uint16_t *const s_nv = sm.ntt_arr(0), *const s_nsig = sm.ntt_arr(1), *const s_npk = sm.ntt_arr(2), *const s_nhm = sm.ntt_arr(3);
    WaveCtx wc;
    wc.slab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.slab + wave * SLAB_WBLK;
    wc.vtab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.vtab + lane * 2;
    wc.lane = lane;

    for (int j = tid; j < N; j += BLOCK) sm.tw[j] = tab->tw[j];
    if constexpr (!COMPACT) {
        init_vtab(sm.vtab, VTAB_ROWS, tid);
        init_slab_const<ENC>(wc.slab, lane);
    }

    const size_t items = full + (batch - full) * PARTS;
    for (size_t item = blockIdx.x; item < items; item += gridDim.x) {
        const bool whole = item < full;
        const size_t s = whole ? item : full + (item - full) / PARTS;
        const int part = whole ? -1 : (int)((item - full) % PARTS);
        auto does = [&](int p) { return part < 0 || part == p; };             // workgroup-uniform
        v4u *wit = COMPACT ? nullptr : g_wit + s * W * 2;
        v4u *inst = COMPACT ? nullptr : g_inst + s * I * 2;
        unsigned char *const cbase = COMPACT ? (unsigned char *)g_wit + s * CL.bytes : nullptr;
        uint32_t *const cs32 = (uint32_t *)cbase;                           // compact: the 11 N small values
        uint32_t *const ct = (uint32_t *)(cbase + CL.t_off);                //          the 2 N quotients, 5 limbs each
        uint32_t *const cb = (uint32_t *)(cbase + CL.bits_off);             //          boolean bit words
        uint32_t *const ci = (uint32_t *)(cbase + CL.instance_off);         //          instance values
        // ---- 1. load + range check ----------------------------------------------------------
        if (tid == 0) { sm.norm = 0; sm.bad = 0; }
        lds_barrier();
        int bad = 0;
        for (int j = tid; j < N; j += BLOCK) {
            uint32_t a = g_sig[s * N + j], b = g_pk[s * N + j], c = g_hm[s * N + j];
            bad |= (a >= Q) | (b >= Q) | (c >= Q);
            sm.sig[j] = (uint16_t)a; s_nsig[j] = (uint16_t)a; s_npk[j] = (uint16_t)b; s_nhm[j] = (uint16_t)c;
        }
        if (bad) sm.bad = 1;
        lds_barrier();
        if (sm.bad) {                                    // uniform across the workgroup
            if (tid == 0 && does(2)) g_status[s] = ST_COEFF_RANGE;
            if constexpr (COMPACT) {
                // all zeros, except that the record says why: the status word travels with it (item 0 clears the data, item
                // 2 owns the tail, as for an accepted signature -- two workgroups may be at work on a split signature)
                static_assert(CL.status_off % 16 == 0 && (CL.bytes - CL.status_off) % 4 == 0, "compact tail");
                if (does(0)) zero_fill((v4u *)cbase, CL.status_off / 16, tid);
                if (does(2) && tid < (int)((CL.bytes - CL.status_off) / 4))
                    ((uint32_t *)(cbase + CL.status_off))[tid] = tid == 0 ? (uint32_t)ST_COEFF_RANGE : 0u;
            } else if (does(0)) {
                zero_fill(wit, W * 2, tid); zero_fill(inst, I * 2, tid);
            }
            lds_barrier();
            continue;
        }
        // ---- 2. clear arithmetic (only what this item's segments need) ---------------------------
        if (does(0) || does(1) || does(2) || does(4)) {
            {
                uint16_t *const polys[3] = {s_nsig, s_npk, s_nhm};
                ntt_modq_lds<LOGN, 3>(polys, sm.tw, tid);                      // falcon_ntt.rs:45,51
            }
            for (int j = tid; j < N; j += BLOCK) {
                uint32_t x = s_nhm[j] + Q - mod_q_u32((uint32_t)s_nsig[j] * s_npk[j]);
                x = x >= Q ? x - Q : x;
                s_nv[j] = (uint16_t)x; sm.v[j] = (uint16_t)x;
            }
            lds_barrier();
            if (does(0) || does(2) || does(4)) intt_modq_lds<LOGN>(sm.v, tab->itw, tid);   // v = hm - sig*pk   :48-49
        }

        const __amdgpu_buffer_rsrc_t rw = make_rsrc(wit, COMPACT ? 0u : (uint32_t)(W * 32));
        const __amdgpu_buffer_rsrc_t ri = make_rsrc(inst, COMPACT ? 0u : (uint32_t)(I * 32));

        // ---- 3. small segments ---------------------------------------------------------------
        uint32_t e8[8];
        // instance_assignment[0] = 1; then pk_ntt, hm_ntt                                   :63,:67
        if (!COMPACT && tid < 2 && does(0)) {
            constexpr uint32_t R[8] = FRW_R32;
            v4u one = ENC == 0 ? (tid ? mk4(0, 0, 0, 0) : mk4(1, 0, 0, 0))
                                 : (tid ? mk4(R[4], R[5], R[6], R[7]) : mk4(R[0], R[1], R[2], R[3]));
            inst[tid] = one;
        }
        if (does(0) || does(1))
        for (int t = wave; t < TILES; t += WAVES) {
            const int k = t * WAVE + lane;
            if (does(0)) {
            const uint32_t vk = sm.v[k];
            if constexpr (COMPACT) {
                ci[k] = s_npk[k];
                ci[N + k] = s_nhm[k];
                cs32[k] = sm.sig[k];
                cs32[N + k] = vk;
                const uint32_t wd = pack_bits<27>(ltq_mask(vk), lane);
                if (lane < 54) cb[t * 54 + lane] = wd;
            } else {
            encode_u32<VENC>(s_npk[k], e8); slab_put(wc.slab, 0, lane, e8);
            emit_values(ri, 32 + t * TILE1, wc);
            encode_u32<VENC>(s_nhm[k], e8); slab_put(wc.slab, 0, lane, e8);
            emit_values(ri, 32 + (N / WAVE + t) * TILE1, wc);
            // S0 sig, S1 v                                                                  :58-59,:71
            encode_u32<VENC>(sm.sig[k], e8); slab_put(wc.slab, 0, lane, e8);
            emit_values(rw, t * TILE1, wc);
            encode_u32<VENC>(vk, e8); slab_put(wc.slab, 0, lane, e8);
            emit_values(rw, (N / WAVE + t) * TILE1, wc);
            // S2 enforce_less_than_q(v[k])                                                  :73-77
            emit_tile<27>(rw, (2 * N + t * WAVE * 27) * 32, ltq_mask(vk), wc);
            }
            }
            // S5 pointwise: [prod, t, c, ltq(c)]                                            :94-111
            if (does(1)) {
                const uint32_t prod = (uint32_t)s_nsig[k] * s_npk[k];
                const uint32_t ab = s_nv[k] + prod;                       // arithmetics.rs:238
                const uint32_t tq = ab / Q, c = ab - tq * Q;              // :242-243
                if constexpr (COMPACT) {
                    cs32[4 * N + 3 * k] = prod;
                    cs32[4 * N + 3 * k + 1] = tq;
                    cs32[4 * N + 3 * k + 2] = c;
                    const uint32_t wd = pack_bits<27>(ltq_mask(c), lane);
                    if (lane < 54) cb[3 * CL.seg_words + t * 54 + lane] = wd;
                } else {
                encode_u32<VENC>(prod, e8); slab_put(wc.slab, 0, lane, e8);
                encode_u32<VENC>(tq, e8);   slab_put(wc.slab, 1, lane, e8);
                encode_u32<VENC>(c, e8);    slab_put(wc.slab, 2, lane, e8);
                emit_tile<30>(rw, (87 * N + t * WAVE * 30) * 32, ltq_mask(c) << 3, wc);
                }
            }
        }
        // S6 l2_norm_var over v || sig: [a0..a13, w0, w1, r, sq]                             :116-120
        unsigned long long nrm = 0;
        if (does(2))
        for (int t = wave; t < 2 * TILES; t += WAVES) {
            const int k = t * WAVE + lane;
            const uint32_t a = k < N ? sm.v[k] : sm.sig[k - N];
            const uint32_t m = lt6144_mask(a);
            const uint32_t r = (m >> 15) & 1u ? a : Q - a;                // misc.rs:35-46
            const uint32_t sq = r * r;
            nrm += sq;
            if constexpr (COMPACT) {
                cs32[7 * N + 2 * k] = r;
                cs32[7 * N + 2 * k + 1] = sq;
                const uint32_t wd = pack_bits<16>(m, lane);
                if (lane < 32) cb[4 * CL.seg_words + t * 32 + lane] = wd;
            } else {
            encode_u32<VENC>(r, e8);  slab_put(wc.slab, 0, lane, e8);
            encode_u32<VENC>(sq, e8); slab_put(wc.slab, 1, lane, e8);
            emit_tile<18>(rw, (117 * N + t * WAVE * 18) * 32, m, wc);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor((unsigned long long)nrm, off, WAVE);
        if (lane == 0) atomicAdd(&sm.norm, nrm);
        lds_barrier();
        // S7 enforce_less_than_norm_bound                                                   :122
        if (wave == 0 && does(2)) {
            const unsigned long long norm = sm.norm;
            const unsigned long long nm = LOGN == 9 ? norm_mask_512(norm) : norm_mask_1024(norm);
            const int32_t verdict = norm >= (LOGN == 9 ? 34034726ull : 70265242ull) ? ST_NORM_BOUND : ST_OK;
            if constexpr (COMPACT) {
                // the two words of S7, then zeros up to the instance values; the status word and zeros up to the stride:
                // no byte of a record is left as it was found
                constexpr int GAP = (int)((CL.instance_off - CL.bits_off) / 4 - CL.bit_words);
                constexpr int TAIL = (int)((CL.bytes - CL.status_off) / 4);
                static_assert(2 + GAP <= WAVE && TAIL <= WAVE, "compact padding is written by one wave");
                if (lane < 2 + GAP) cb[4 * CL.seg_words + N + lane] = lane < 2 ? (uint32_t)(nm >> (32 * lane)) : 0u;
                if (lane < TAIL) ((uint32_t *)(cbase + CL.status_off))[lane] = lane == 0 ? (uint32_t)verdict : 0u;
            } else {
            constexpr uint32_t R[8] = FRW_R32;
            const int half = lane & 1;
            v4u one = ENC == 0 ? (half ? mk4(0, 0, 0, 0) : mk4(1, 0, 0, 0))
                                 : (half ? mk4(R[4], R[5], R[6], R[7]) : mk4(R[0], R[1], R[2], R[3]));
            v4u *o = wit + (size_t)153 * N * 2;
            for (int c = lane; c < NB * 2; c += WAVE) {
                const int pos = c >> 1;
                o[c] = (nm >> pos) & 1ull ? one : mk4(0, 0, 0, 0);
            }
            }
            if (lane == 0) g_status[s] = verdict;
        }

        // ---- 4./5. ladders: S3 = mod_q blocks of NTT(sig), S4 = of NTT(v)                   :88-91
#pragma unroll 1
        for (int which = 0; which < 2; which++) {
            if (!does(3 + which)) continue;
            ladder_lds<LOGN>(sm.lad, which ? sm.v : sm.sig, sm.tw, &tab->ck[0][0], tid);
            const uint32_t seg = (uint32_t)(which ? 58 : 29) * N * 32;
            for (int t = wave; t < TILES; t += WAVES) {
                const int k = t * WAVE + lane;
                uint32_t a[5], q5[5];
#pragma unroll
                for (int i = 0; i < 5; i++) a[i] = sm.lad[i * N + k];
                const uint32_t b = divmod_q_u160(a, q5);                  // arithmetics.rs:127-134
                if constexpr (COMPACT) {
                    cs32[(which ? 3 : 2) * N + k] = b;
#pragma unroll
                    for (int i = 0; i < 5; i++) ct[((size_t)(which ? N : 0) + k) * 5 + i] = q5[i];
                    const uint32_t wd = pack_bits<27>(ltq_mask(b), lane);
                    if (lane < 54) cb[(which ? 2 : 1) * CL.seg_words + t * 54 + lane] = wd;
                } else {
                encode_u160<VENC>(q5, e8); slab_put(wc.slab, 0, lane, e8);    // t_var :137
                encode_u32<VENC>(b, e8);   slab_put(wc.slab, 1, lane, e8);    // b_var :138
                emit_tile<29>(rw, seg + t * WAVE * 29 * 32, ltq_mask(b) << 2, wc);
                }
            }
            lds_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kernel: the signed-split variant, FalconDualNTTVerificationCircuit::generate_constraints
// (circuits/falcon_dual_ntt.rs:26-132, gadgets/dual_poly.rs:15-51, gadgets/misc.rs:55-65).  Same building blocks,
// different layout (units: field elements; W = 186 N + 4 + nb):
//   0      sig.pos N | N sig.neg N | 2N pos*neg products N | 3N is_zero [is_not_equal, multiplier]       dual_poly.rs:20-29
//   3N+2   v.pos N | 4N+2 v.neg N | 5N+2 products N | 6N+2 is_zero 2                                     falcon_dual_ntt.rs:73
//   6N+4   mod_q blocks of ntt_circuit(sig.pos), +29N (sig.neg), +58N (v.pos), +87N (v.neg)             :85-92
//   122N+4 per i: [sig_ntt.neg*pk_ntt, t, b, ltq(b)] [sig_ntt.pos*pk_ntt, t, b, ltq(b)]  (2 x 30)        :95-116
//   182N+4 squares of v.pos, v.neg, sig.pos, sig.neg (4N)                                                :121-129
//   186N+4 norm bound (50 | 52)                                                                          :131
// A DualPolynomial has pos[i]*neg[i] = 0 for every i, so the products are zeros, is_not_equal = 0, multiplier = 1.
// ------------------------------------------------------------------------------------------------
template <int LOGN>
struct alignas(16) SmemDual {
    static constexpr int N = 1 << LOGN;
    unsigned char slab[SLAB_BYTES];
    uint32_t lad[5 * N];                  // also: third value slot of the pointwise tiles while no ladder is alive
    uint16_t tw[N];
    uint16_t sp[N], sn[N], vp[N], vn[N];          // coefficient domain (signed split, threshold 6144)
    uint16_t nsp[N], nsn[N], nvp[N], nvn[N], npk[N], nhm[N];   // NTT domain
    uint16_t vtab[TileShape<18>::ROW0 * WAVE];   // shapes 29 and 30 only
    unsigned long long norm;
    int bad;
};

template <int LOGN, int ENC>
__global__ __launch_bounds__(BLOCK) void witness_dual_ntt_verify_kernel(
    const Tables *__restrict__ tab, size_t batch,
    const uint16_t *__restrict__ g_sig, const uint16_t *__restrict__ g_pk, const uint16_t *__restrict__ g_hm,
    v4u *__restrict__ g_wit, v4u *__restrict__ g_inst, int32_t *__restrict__ g_status)
{
    constexpr int N = 1 << LOGN;
    constexpr int NB = LOGN == 9 ? 50 : 52;
    constexpr size_t W = 186 * (size_t)N + 4 + NB;
    constexpr size_t I = 2 * (size_t)N + 1;
    constexpr int TILES = N / WAVE;
    constexpr uint32_t TILE1 = WAVE * 32;
    constexpr uint32_t HALF_Q = 6144;
    __shared__ SmemDual<LOGN> sm;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WaveCtx wc;
    wc.slab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.slab + wave * SLAB_WBLK;
    wc.vtab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.vtab + lane * 2;
    wc.lane = lane;
    for (int j = tid; j < N; j += BLOCK) sm.tw[j] = tab->tw[j];
    init_vtab(sm.vtab, TileShape<18>::ROW0, tid);
    init_slab_const<ENC>(wc.slab, lane);
    constexpr uint32_t R[8] = FRW_R32;
    const int half = lane & 1;
    const v4u one = ENC == 0 ? (half ? mk4(0, 0, 0, 0) : mk4(1, 0, 0, 0))
                             : (half ? mk4(R[4], R[5], R[6], R[7]) : mk4(R[0], R[1], R[2], R[3]));

    for (size_t s = blockIdx.x; s < batch; s += gridDim.x) {
        v4u *wit = g_wit + s * W * 2;
        v4u *inst = g_inst + s * I * 2;
        if (tid == 0) { sm.norm = 0; sm.bad = 0; }
        lds_barrier();
        int bad = 0;
        for (int j = tid; j < N; j += BLOCK) {
            const uint32_t a = g_sig[s * N + j], b = g_pk[s * N + j], c = g_hm[s * N + j];
            bad |= (a >= Q) | (b >= Q) | (c >= Q);
            const uint16_t p = (uint16_t)(a < HALF_Q ? a : 0), m = (uint16_t)(a < HALF_Q ? 0 : Q - a);   // falcon_dual_ntt.rs:27
            sm.sp[j] = p; sm.sn[j] = m; sm.nsp[j] = p; sm.nsn[j] = m;
            sm.npk[j] = (uint16_t)b; sm.nhm[j] = (uint16_t)c;
        }
        if (bad) sm.bad = 1;
        lds_barrier();
        if (sm.bad) {
            if (tid == 0) g_status[s] = ST_COEFF_RANGE;
            zero_fill(wit, W * 2, tid);
            zero_fill(inst, I * 2, tid);
            lds_barrier();                  // every wave has read sm.bad before thread 0 clears it for the next item
            continue;
        }
        {
            uint16_t *const polys[4] = {sm.nsp, sm.nsn, sm.npk, sm.nhm};
            ntt_modq_lds<LOGN, 4>(polys, sm.tw, tid);                              // :45,:53
        }
        for (int j = tid; j < N; j += BLOCK) {                                     // v = hm - uh_pos + uh_neg   :48-50
            const uint32_t sg = sm.nsp[j] + Q - sm.nsn[j];
            uint32_t x = sm.nhm[j] + Q - mod_q_u32(mod_q_u32(sg) * sm.npk[j]);
            sm.vp[j] = (uint16_t)(x >= Q ? x - Q : x);
        }
        lds_barrier();
        intt_modq_lds<LOGN>(sm.vp, tab->itw, tid);
        for (int j = tid; j < N; j += BLOCK) {                                     // DualPolynomial::from(&v)   :51
            const uint32_t a = sm.vp[j];
            const uint16_t p = (uint16_t)(a < HALF_Q ? a : 0), m = (uint16_t)(a < HALF_Q ? 0 : Q - a);
            sm.vp[j] = p; sm.vn[j] = m; sm.nvp[j] = p; sm.nvn[j] = m;
        }
        lds_barrier();
        {
            uint16_t *const polys[2] = {sm.nvp, sm.nvn};
            ntt_modq_lds<LOGN, 2>(polys, sm.tw, tid);
        }

        const __amdgpu_buffer_rsrc_t rw = make_rsrc(wit, (uint32_t)(W * 32));
        const __amdgpu_buffer_rsrc_t ri = make_rsrc(inst, (uint32_t)(I * 32));
        uint32_t e8[8];
        if (tid < 2) inst[tid] = one;                                              // instance_assignment[0] = 1
        if (tid < 4) {                                                             // is_zero: [0, 1] twice
            const v4u z = mk4(0, 0, 0, 0);
            wit[(size_t)3 * N * 2 + tid] = tid < 2 ? z : one;
            wit[((size_t)6 * N + 2) * 2 + tid] = tid < 2 ? z : one;
        }
        unsigned long long nrm = 0;
        for (int t = wave; t < TILES; t += WAVES) {
            const int k = t * WAVE + lane;
            const uint32_t to = t * TILE1;
            encode_u32<ENC>(sm.npk[k], e8); slab_put(wc.slab, 0, lane, e8);
            emit_values(ri, 32 + to, wc);
            encode_u32<ENC>(sm.nhm[k], e8); slab_put(wc.slab, 0, lane, e8);
            emit_values(ri, 32 + N * 32 + to, wc);
            // dual_poly.rs:20-21 pos, neg; :24-27 products (all zero)
#pragma unroll 1
            for (int q4 = 0; q4 < 4; q4++) {                                       // v.pos, v.neg, sig.pos, sig.neg
                const uint16_t *src = q4 == 0 ? sm.vp : q4 == 1 ? sm.vn : q4 == 2 ? sm.sp : sm.sn;
                const uint32_t lin = q4 == 0 ? 3 * N + 2 : q4 == 1 ? 4 * N + 2 : q4 == 2 ? 0 : N;
                const uint32_t val = src[k];
                encode_u32<ENC>(val, e8); slab_put(wc.slab, 0, lane, e8);
                emit_values(rw, lin * 32 + to, wc);
                // misc.rs:58-62 squares, in the order v.pos, v.neg, sig.pos, sig.neg
                const uint32_t sq = val * val;
                nrm += sq;
                encode_u32<ENC>(sq, e8); slab_put(wc.slab, 0, lane, e8);
                emit_values(rw, ((182 + q4) * N + 4) * 32 + to, wc);
            }
            emit_zeros(rw, 2 * N * 32 + to, wc);
            emit_zeros(rw, (5 * N + 2) * 32 + to, wc);
        }
        // pointwise (:95-116): 2N blocks of 30, block 2i = left, 2i+1 = right
        for (int t = wave; t < 2 * TILES; t += WAVES) {
            const int j = t * WAVE + lane, i = j >> 1;
            const uint32_t pkv = sm.npk[i];
            uint32_t prod, a;
            if (j & 1) { prod = (uint32_t)sm.nsp[i] * pkv; a = sm.nvp[i] + prod; }              // right :109-114
            else { prod = (uint32_t)sm.nsn[i] * pkv; a = sm.nhm[i] + sm.nvn[i] + prod; }        // left  :100-107
            const uint32_t tq = a / Q, b = a - tq * Q;
            encode_u32<ENC>(prod, e8); slab_put(wc.slab, 0, lane, e8);
            encode_u32<ENC>(tq, e8);   slab_put(wc.slab, 1, lane, e8);
            encode_u32<ENC>(b, e8);    slab_put(wc.slab, 2, lane, e8);
            emit_tile<30>(rw, (122 * N + 4 + t * WAVE * 30) * 32, ltq_mask(b) << 3, wc);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor((unsigned long long)nrm, off, WAVE);
        if (lane == 0) atomicAdd(&sm.norm, nrm);
        lds_barrier();
        if (wave == 0) {                                                           // :131
            const unsigned long long norm = sm.norm;
            const unsigned long long nm = LOGN == 9 ? norm_mask_512(norm) : norm_mask_1024(norm);
            v4u *o = wit + ((size_t)186 * N + 4) * 2;
            for (int c = lane; c < NB * 2; c += WAVE) o[c] = (nm >> (c >> 1)) & 1ull ? one : mk4(0, 0, 0, 0);
            if (lane == 0) g_status[s] = norm >= (LOGN == 9 ? 34034726ull : 70265242ull) ? ST_NORM_BOUND : ST_OK;
        }
        // four ladders (:85-92): sig.pos, sig.neg, v.pos, v.neg
#pragma unroll 1
        for (int which = 0; which < 4; which++) {
            const uint16_t *in = which == 0 ? sm.sp : which == 1 ? sm.sn : which == 2 ? sm.vp : sm.vn;
            ladder_lds<LOGN>(sm.lad, in, sm.tw, &tab->ck[0][0], tid);
            const uint32_t seg = ((6 + 29 * which) * N + 4) * 32;
            for (int t = wave; t < TILES; t += WAVES) {
                const int k = t * WAVE + lane;
                uint32_t a[5], q5[5];
#pragma unroll
                for (int i = 0; i < 5; i++) a[i] = sm.lad[i * N + k];
                const uint32_t b = divmod_q_u160(a, q5);
                encode_u160<ENC>(q5, e8); slab_put(wc.slab, 0, lane, e8);
                encode_u32<ENC>(b, e8);   slab_put(wc.slab, 1, lane, e8);
                emit_tile<29>(rw, seg + t * WAVE * 29 * 32, ltq_mask(b) << 2, wc);
            }
            lds_barrier();
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kernel: NTTPolyVar::ntt_circuit alone (poly.rs:104-159): N mod_q blocks + the reduced NTT
// ------------------------------------------------------------------------------------------------
template <int LOGN>
struct alignas(16) SmemNtt {
    static constexpr int N = 1 << LOGN;
    unsigned char slab[SLAB_BYTES];
    uint32_t lad[5 * N];
    uint16_t tw[N];
    uint16_t in[N];
    uint16_t vtab[TileShape<30>::ROW0 * WAVE];    // shape 29 only
    int bad;
};

template <int LOGN, int ENC>
__global__ __launch_bounds__(BLOCK) void ntt_modq_kernel(
    const Tables *__restrict__ tab, size_t batch,
    const uint16_t *__restrict__ g_poly,
    v4u *__restrict__ g_wit, uint16_t *__restrict__ g_ntt, int32_t *__restrict__ g_status)
{
    constexpr int N = 1 << LOGN;
    constexpr int TILES = N / WAVE;
    constexpr size_t WN = (size_t)29 * N;
    __shared__ SmemNtt<LOGN> sm;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WaveCtx wc;
    wc.slab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.slab + wave * SLAB_WBLK;
    wc.vtab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.vtab + lane * 2;
    wc.lane = lane;
    for (int j = tid; j < N; j += BLOCK) sm.tw[j] = tab->tw[j];
    init_vtab(sm.vtab, TileShape<30>::ROW0, tid);
    init_slab_const<ENC>(wc.slab, lane);

    for (size_t s = blockIdx.x; s < batch; s += gridDim.x) {
        if (tid == 0) sm.bad = 0;
        lds_barrier();
        int bad = 0;
        for (int j = tid; j < N; j += BLOCK) {
            uint32_t a = g_poly[s * N + j];
            bad |= a >= Q;
            sm.in[j] = (uint16_t)a;
        }
        if (bad) sm.bad = 1;
        lds_barrier();
        if (tid == 0) g_status[s] = sm.bad ? ST_COEFF_RANGE : ST_OK;
        if (sm.bad) {
            zero_fill(g_wit + s * WN * 2, WN * 2, tid);
            for (int j = tid; j < N; j += BLOCK) g_ntt[s * N + j] = 0;
            lds_barrier();
            continue;
        }
        ladder_lds<LOGN>(sm.lad, sm.in, sm.tw, &tab->ck[0][0], tid);
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(g_wit + s * WN * 2, (uint32_t)(WN * 32));
        for (int t = wave; t < TILES; t += WAVES) {
            const int k = t * WAVE + lane;
            uint32_t a[5], q5[5], e8[8];
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = sm.lad[i * N + k];
            const uint32_t b = divmod_q_u160(a, q5);
            g_ntt[s * N + k] = (uint16_t)b;
            encode_u160<ENC>(q5, e8); slab_put(wc.slab, 0, lane, e8);
            encode_u32<ENC>(b, e8);   slab_put(wc.slab, 1, lane, e8);
            emit_tile<29>(rw, t * WAVE * 29 * 32, ltq_mask(b) << 2, wc);
        }
        lds_barrier();
    }
}

// ------------------------------------------------------------------------------------------------
// kernel: FRW_ENC_COMPACT -> the arkworks buffers (witness_assignment / instance_assignment, Montgomery), i.e. exactly
// what witness_ntt_verify_kernel<LOGN, 1> writes.  The integers of the compact buffer are converted to Montgomery form
// (the same CIOS rounds the generator uses) into the wave's slab, the booleans of a tile are cut out of the bit array
// (two ds_bpermute per lane) and handed to emit_tile as the per-block mask.  A receiver of an all-gathered compact chunk
// runs this locally: 0.11 MB cross the fabric per Falcon-1024 signature instead of 5.08 MB.
// ------------------------------------------------------------------------------------------------
struct alignas(16) SmemExpand {
    unsigned char slab[3 * SLAB_SLOT];
    uint16_t vtab[VTAB_ROWS * WAVE];
};

// bits [k NB, k NB + NB) of a tile's bit words (lane i holds word i, lanes >= 2 NB hold zero)
template <int NB>
__device__ __forceinline__ uint32_t unpack_bits(uint32_t word, int lane)
{
    const int off = lane * NB, i0 = off >> 5, o = off & 31;
    const uint32_t lo = (uint32_t)__shfl((int)word, i0, WAVE), hi = (uint32_t)__shfl((int)word, (i0 + 1) & 63, WAVE);
    const uint64_t both = (uint64_t)lo | ((uint64_t)hi << 32);
    return (uint32_t)(both >> o) & ((1u << NB) - 1u);
}

template <int LOGN>
__global__ __launch_bounds__(BLOCK) void expand_kernel(size_t batch, const unsigned char *__restrict__ g_compact,
                                                       v4u *__restrict__ g_wit, v4u *__restrict__ g_inst)
{
    constexpr int N = 1 << LOGN;
    constexpr int NB = LOGN == 9 ? 50 : 52;
    constexpr size_t W = 153 * (size_t)N + NB;
    constexpr size_t I = 2 * (size_t)N + 1;
    constexpr int TILES = N / WAVE;
    constexpr uint32_t TILE1 = WAVE * 32;
    constexpr CompactLayout CL = compact_layout(LOGN);
    __shared__ SmemExpand sm;
    const int tid = threadIdx.x, lane = tid & (WAVE - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    WaveCtx wc;
    wc.slab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.slab + wave * SLAB_WBLK;
    wc.vtab = (uint32_t)(uintptr_t)(FRW_LDS void *)sm.vtab + lane * 2;
    wc.lane = lane;
    init_vtab(sm.vtab, VTAB_ROWS, tid);
    init_slab_const<1>(wc.slab, lane);
    lds_barrier();
    constexpr uint32_t R[8] = FRW_R32;
    const v4u one = lane & 1 ? mk4(R[4], R[5], R[6], R[7]) : mk4(R[0], R[1], R[2], R[3]);

    for (size_t s = blockIdx.x; s < batch; s += gridDim.x) {
        const unsigned char *cbase = g_compact + s * CL.bytes;
        const uint32_t *cs32 = (const uint32_t *)cbase;
        const uint32_t *ct = (const uint32_t *)(cbase + CL.t_off);
        const uint32_t *cb = (const uint32_t *)(cbase + CL.bits_off);
        const uint32_t *ci = (const uint32_t *)(cbase + CL.instance_off);
        v4u *wit = g_wit + s * W * 2;
        v4u *inst = g_inst + s * I * 2;
        const __amdgpu_buffer_rsrc_t rw = make_rsrc(wit, (uint32_t)(W * 32));
        const __amdgpu_buffer_rsrc_t ri = make_rsrc(inst, (uint32_t)(I * 32));
        uint32_t e8[8];
        // a rejected signature (FRW_ST_COEFF_RANGE in the record's status word) expands to what the direct kernel leaves for
        // it: zeros in the witness AND the instance vector (no leading one) -- workgroup-uniform
        if (*(const uint32_t *)(cbase + CL.status_off) == (uint32_t)ST_COEFF_RANGE) {
            zero_fill(wit, W * 2, tid);
            zero_fill(inst, I * 2, tid);
            continue;
        }
        // Every loop below keeps the loads of its NEXT tile in flight while the current tile is emitted (58 stores): under
        // a saturated write stream an HBM read takes long enough that four waves per SIMD do not hide it otherwise
        // (measured: +5 % from over-subscribing the grid before this was done, tools/ab_variants.py --workload expand).
        // instance: [one, pk_ntt, hm_ntt]; S0, S1; S2 (booleans only)
        if (tid < 2) inst[tid] = one;
        {
            uint32_t n_pk, n_hm, n_sig, n_v, n_wd;
            auto ld = [&](int t) {
                const int k = t * WAVE + lane;
                n_pk = ci[k]; n_hm = ci[N + k]; n_sig = cs32[k]; n_v = cs32[N + k];
                n_wd = lane < 54 ? cb[t * 54 + lane] : 0u;
            };
            ld(wave);
            for (int t = wave; t < TILES; t += WAVES) {
                const uint32_t c_pk = n_pk, c_hm = n_hm, c_sig = n_sig, c_v = n_v, c_wd = n_wd;
                if (t + WAVES < TILES) ld(t + WAVES);
                encode_u32<1>(c_pk, e8); slab_put(wc.slab, 0, lane, e8);
                emit_values(ri, 32 + t * TILE1, wc);
                encode_u32<1>(c_hm, e8); slab_put(wc.slab, 0, lane, e8);
                emit_values(ri, 32 + (N / WAVE + t) * TILE1, wc);
                encode_u32<1>(c_sig, e8); slab_put(wc.slab, 0, lane, e8);
                emit_values(rw, t * TILE1, wc);
                encode_u32<1>(c_v, e8); slab_put(wc.slab, 0, lane, e8);
                emit_values(rw, (N / WAVE + t) * TILE1, wc);
                emit_tile<27>(rw, (2 * N + t * WAVE * 27) * 32, unpack_bits<27>(c_wd, lane), wc);
            }
        }
        // S3, S4: [t, b, ltq(b)]
#pragma unroll 1
        for (int which = 0; which < 2; which++) {
            uint32_t n_q[5], n_b, n_wd;
            auto ld = [&](int t) {
                const int k = t * WAVE + lane;
#pragma unroll
                for (int i = 0; i < 5; i++) n_q[i] = ct[((size_t)(which ? N : 0) + k) * 5 + i];
                n_b = cs32[(which ? 3 : 2) * N + k];
                n_wd = lane < 54 ? cb[(which ? 2 : 1) * CL.seg_words + t * 54 + lane] : 0u;
            };
            ld(wave);
            for (int t = wave; t < TILES; t += WAVES) {
                uint32_t q5[5];
#pragma unroll
                for (int i = 0; i < 5; i++) q5[i] = n_q[i];
                const uint32_t c_b = n_b, c_wd = n_wd;
                if (t + WAVES < TILES) ld(t + WAVES);
                encode_u160<1>(q5, e8); slab_put(wc.slab, 0, lane, e8);
                encode_u32<1>(c_b, e8); slab_put(wc.slab, 1, lane, e8);
                emit_tile<29>(rw, ((which ? 58 : 29) * N + t * WAVE * 29) * 32, unpack_bits<27>(c_wd, lane) << 2, wc);
            }
        }
        // S5: [prod, t, c, ltq(c)]
        {
            uint32_t n_a[3], n_wd;
            auto ld = [&](int t) {
                const int k = t * WAVE + lane;
#pragma unroll
                for (int i = 0; i < 3; i++) n_a[i] = cs32[4 * N + 3 * k + i];
                n_wd = lane < 54 ? cb[3 * CL.seg_words + t * 54 + lane] : 0u;
            };
            ld(wave);
            for (int t = wave; t < TILES; t += WAVES) {
                const uint32_t c0 = n_a[0], c1 = n_a[1], c2 = n_a[2], c_wd = n_wd;
                if (t + WAVES < TILES) ld(t + WAVES);
                encode_u32<1>(c0, e8); slab_put(wc.slab, 0, lane, e8);
                encode_u32<1>(c1, e8); slab_put(wc.slab, 1, lane, e8);
                encode_u32<1>(c2, e8); slab_put(wc.slab, 2, lane, e8);
                emit_tile<30>(rw, (87 * N + t * WAVE * 30) * 32, unpack_bits<27>(c_wd, lane) << 3, wc);
            }
        }
        // S6: [a0..a13, w0, w1, r, sq]
        {
            uint32_t n_r, n_sq, n_wd;
            auto ld = [&](int t) {
                const int k = t * WAVE + lane;
                n_r = cs32[7 * N + 2 * k]; n_sq = cs32[7 * N + 2 * k + 1];
                n_wd = lane < 32 ? cb[4 * CL.seg_words + t * 32 + lane] : 0u;
            };
            ld(wave);
            for (int t = wave; t < 2 * TILES; t += WAVES) {
                const uint32_t c_r = n_r, c_sq = n_sq, c_wd = n_wd;
                if (t + WAVES < 2 * TILES) ld(t + WAVES);
                encode_u32<1>(c_r, e8);  slab_put(wc.slab, 0, lane, e8);
                encode_u32<1>(c_sq, e8); slab_put(wc.slab, 1, lane, e8);
                emit_tile<18>(rw, (117 * N + t * WAVE * 18) * 32, unpack_bits<16>(c_wd, lane), wc);
            }
        }
        // S7
        if (wave == 0) {
            const unsigned long long nm = (unsigned long long)cb[4 * CL.seg_words + N] |
                                          ((unsigned long long)cb[4 * CL.seg_words + N + 1] << 32);
            v4u *o = wit + (size_t)153 * N * 2;
            for (int c = lane; c < NB * 2; c += WAVE) o[c] = (nm >> (c >> 1)) & 1ull ? one : mk4(0, 0, 0, 0);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// kernel: stand-alone gadget blocks (the reference's gadget API called outside the full circuit, as its unit tests
// do): one block of BLK field elements per item, 64 items per wavefront tile.
//   G_LESS_THAN_Q   a:u64            -> ltq block, 27            range_proofs.rs:42-94
//   G_MOD_Q         a:5 x u32 limbs  -> [t, b, ltq(b)], 29       arithmetics.rs:105-149
//   G_ADD_MOD       a:u64, b:u64     -> [t, c, ltq(c)], 29       arithmetics.rs:214-262
//   G_L2_ELEM       a:u64            -> [a0..a13, w0, w1, r, sq], 18   misc.rs:35-47 + range_proofs.rs:289-333
//   G_NORM_512/1024 a:u64            -> bits + gates, 50 / 52    range_proofs.rs:100-186 / 192-272
// status: 0, or ST_COEFF_RANGE when an input is outside what the path can produce (see include/frw.h).
// ------------------------------------------------------------------------------------------------
template <int KIND> struct GadgetShape;
template <> struct GadgetShape<G_LESS_THAN_Q> { static constexpr int BLK = 27, NVAL = 0, VFIRST = 0; using mask_t = uint32_t; };
template <> struct GadgetShape<G_MOD_Q>       { static constexpr int BLK = 29, NVAL = 2, VFIRST = 0; using mask_t = uint32_t; };
template <> struct GadgetShape<G_ADD_MOD>     { static constexpr int BLK = 29, NVAL = 2, VFIRST = 0; using mask_t = uint32_t; };
template <> struct GadgetShape<G_L2_ELEM>     { static constexpr int BLK = 18, NVAL = 2, VFIRST = 16; using mask_t = uint32_t; };
template <> struct GadgetShape<G_NORM_512>    { static constexpr int BLK = 50, NVAL = 0, VFIRST = 0; using mask_t = uint64_t; };
template <> struct GadgetShape<G_NORM_1024>   { static constexpr int BLK = 52, NVAL = 0, VFIRST = 0; using mask_t = uint64_t; };

template <int KIND, int ENC>
__global__ __launch_bounds__(BLOCK) void gadget_kernel(size_t count, const void *__restrict__ in_a,
                                                       const uint64_t *__restrict__ in_b, v4u *__restrict__ out,
                                                       int32_t *__restrict__ status)
{
    using S = GadgetShape<KIND>;
    using mask_t = typename S::mask_t;
    __shared__ v4u slab_all[WAVES][2 * 2 * WAVE];
    const int lane = threadIdx.x & (WAVE - 1), wave = threadIdx.x >> 6;
    v4u *slab = slab_all[wave];
    const size_t tiles = (count + WAVE - 1) / WAVE;
    for (size_t t = (size_t)blockIdx.x * WAVES + wave; t < tiles; t += (size_t)gridDim.x * WAVES) {
        const size_t item = t * WAVE + lane;
        const bool live = item < count;
        mask_t mask = 0;
        int st = ST_OK;
        uint32_t e8[8];
        if (KIND == G_LESS_THAN_Q) {
            const uint64_t a = live ? ((const uint64_t *)in_a)[item] : 0;
            mask = ltq_mask((uint32_t)a & 0x3fffu);                  // to_bits_le().take(14): range_proofs.rs:62-69
        } else if (KIND == G_MOD_Q) {
            uint32_t a[5], q5[5];
#pragma unroll
            for (int i = 0; i < 5; i++) a[i] = live ? ((const uint32_t *)in_a)[item * 5 + i] : 0;
            const uint32_t b = divmod_q_u160(a, q5);
            encode_u160<ENC>(q5, e8); gslab_put(slab, 0, lane, e8);
            encode_u32<ENC>(b, e8);   gslab_put(slab, 1, lane, e8);
            mask = ltq_mask(b) << 2;
        } else if (KIND == G_ADD_MOD) {
            const uint64_t a = live ? ((const uint64_t *)in_a)[item] : 0, b = live ? in_b[item] : 0;
            const uint64_t ab = a + b;                                // arithmetics.rs:238
            if (ab < a) st = ST_COEFF_RANGE;                          // beyond 64 bits: not representable here
            const uint64_t tq = ab / Q;
            const uint32_t c = (uint32_t)(ab - tq * Q);               // :242-243
            const uint32_t t5[5] = {(uint32_t)tq, (uint32_t)(tq >> 32), 0, 0, 0};
            encode_u160<ENC>(t5, e8); gslab_put(slab, 0, lane, e8);
            encode_u32<ENC>(c, e8);   gslab_put(slab, 1, lane, e8);
            mask = ltq_mask(c) << 2;
        } else if (KIND == G_L2_ELEM) {
            const uint64_t a = live ? ((const uint64_t *)in_a)[item] : 0;
            if (a > Q) st = ST_COEFF_RANGE;                           // q - e would wrap in the field
            const uint32_t e = (uint32_t)a;
            mask = lt6144_mask(e & 0x3fffu);
            const uint32_t r = (mask >> 15) & 1u ? e : Q - e;
            encode_u32<ENC>(r, e8); gslab_put(slab, 0, lane, e8);
            const uint64_t sq = (uint64_t)r * r;
            const uint32_t s5[5] = {(uint32_t)sq, (uint32_t)(sq >> 32), 0, 0, 0};
            encode_u160<ENC>(s5, e8); gslab_put(slab, 1, lane, e8);
        } else {
            const uint64_t a = live ? ((const uint64_t *)in_a)[item] : 0;
            mask = KIND == G_NORM_512 ? norm_mask_512(a) : norm_mask_1024(a);
        }
        if (live && status) status[item] = st;
        const size_t remaining = count - t * WAVE;
        const int nchunks = (int)(remaining >= WAVE ? WAVE : remaining) * S::BLK * 2;
        emit_tile_generic<ENC, S::BLK, S::NVAL, S::VFIRST, mask_t>(out + t * WAVE * S::BLK * 2, slab, mask, lane, nchunks);
    }
}

// ------------------------------------------------------------------------------------------------
// per-item digest: out[i] = sum_j splitmix64(buf[i][j] + j*golden)  (order independent)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

__global__ __launch_bounds__(BLOCK) void digest_kernel(const uint64_t *__restrict__ buf, size_t words, size_t items,
                                                       unsigned long long *__restrict__ out, int blocks_per_item)
{
    const size_t item = blockIdx.x / blocks_per_item;
    const int part = blockIdx.x % blocks_per_item;
    if (item >= items) return;
    const uint64_t *p = buf + item * words;
    uint64_t h = 0;
    for (size_t j = (size_t)part * BLOCK + threadIdx.x; j < words; j += (size_t)blocks_per_item * BLOCK)
        h += splitmix64(p[j] + j * 0x9E3779B97F4A7C15ull);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) h += __shfl_xor((unsigned long long)h, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0) atomicAdd(&out[item], (unsigned long long)h);
}

// ------------------------------------------------------------------------------------------------
// diagnostic: a compute-free write stream with the witness kernel's store shape (each workgroup fills contiguous
// slabs, 16 B per lane, 1 KiB per store instruction).  bench.py runs it on the same device, in the same process,
// next to the timed region, so that `roofline.achieved` can be read against what the device's HBM actually takes.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(BLOCK) void write_stream_kernel(v4u *__restrict__ out, size_t slab16, size_t nslabs, uint32_t seed)
{
    v4u v = mk4(seed, seed ^ threadIdx.x, 3u, 4u);
    for (size_t s = blockIdx.x; s < nslabs; s += gridDim.x) {
        v4u *o = out + s * slab16;
        for (size_t i = threadIdx.x; i < slab16; i += BLOCK) {
#if defined(FRW_NO_STORE)
            asm volatile("" ::"v"(v), "v"(&o[i]));
#else
            o[i] = v;
#endif
        }
    }
}

hipError_t launch_write_stream(void *buf, size_t bytes, size_t slab_bytes, int num_cu, hipStream_t st)
{
    const size_t slab16 = slab_bytes / 16;
    if (slab16 == 0) return hipErrorInvalidValue;
    const size_t nslabs = bytes / (slab16 * 16);
    if (nslabs == 0) return hipSuccess;
    // 24 workgroups per CU: three times what a CU holds of this register-light kernel.  The calibration is meant to be the
    // best compute-free stream known for the device (tools/hbm_write_pattern.hip, 32,256 units: 768 workgroups 6,615 GB/s,
    // 3,072 6,736, 6,144 6,947), not a stream handicapped to the witness kernel's occupancy.
    size_t grid = (size_t)num_cu * 24;
    if (grid > nslabs) grid = nslabs;
    hipLaunchKernelGGL(write_stream_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, st, (v4u *)buf, slab16, nslabs, 7u);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// launchers (called from frw_capi.cpp)
// ------------------------------------------------------------------------------------------------
// Persistent grid: as many workgroups as the device keeps resident (LDS-limited: 2-3 per CU), each
// striding over the batch.  Residency is asked from the runtime once per kernel instantiation.
// Filled exactly once per process (std::call_once in init_launch_config, which every context creation calls before
// any launch), read-only afterwards: contexts on several devices / threads share it safely.
static int g_occ_verify[4], g_occ_dual[4], g_occ_ntt[4];       // resident workgroups per CU, [(LOGN-9)*2 + ENC]
static int g_occ_compact[2], g_occ_expand[2];                  // [LOGN - 9]
static std::once_flag g_occ_once;

template <typename K>
static void query_residency(K kernel, int &cache)
{
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, BLOCK, 0) != hipSuccess || per_cu < 1) per_cu = 2;
    cache = per_cu;
}

// Workgroups to launch for `batch` items.  `cap` = what the device keeps resident (per_cu x CUs).  Launches with at least
// two items per workgroup to spare use up to OVERSUB x cap: with more workgroups than the device holds, the ones that
// finish are replaced at different moments, which keeps the workgroups' compute and store phases from lining up across
// the chip (measured, tools/ab_variants.py: 32,768 Falcon-1024 signatures +1.3 ... 3.5 % at 2 x ... 4 x the resident 768,
// no more at 6 x and 8 x; 8,192 Falcon-512 signatures +2.7 % at 2,048 and +4.4 % at 4,096 workgroups over 1,024, -9 % at
// one signature per workgroup; the expansion kernel, whose items are all stores behind a few loads, +4.5 % at 8 x:
// profiles/r02_scheduling_ab.txt).
#ifndef FRW_OVERSUB
#define FRW_OVERSUB 4
#endif
static int resident_grid(size_t batch, int num_cu, int per_cu, int oversub = FRW_OVERSUB)
{
    const size_t cap = (size_t)(per_cu > 0 ? per_cu : 2) * (size_t)num_cu;
    if (batch <= cap) return (int)batch;
#if defined(FRW_FORCE_GRID)          // A/B builds only
    return FRW_FORCE_GRID;
#endif
    // beyond the resident capacity only while every workgroup keeps at least two items
    size_t top = cap;
    if (batch >= 2 * cap) top = std::min((size_t)oversub * cap, batch / 2);
    // Every workgroup streams one 2.5-5 MB signature at a time, so a batch that is not a multiple of the grid ends in a
    // tail where most CUs idle.  For short launches (< 8 rounds) a grid between 5/8 and 8/8 of the target that divides the
    // batch wins (4,096 signatures: 512 x 8 rounds beats 768 x 5.33 by 1.1 %); long launches amortise the tail, and the
    // witness kernel splits it besides.
    if (batch < 8 * top)
        for (size_t g = top; g * 8 >= top * 5; g--)
            if (batch % g == 0) return (int)g;
    return (int)top;
}

// Called once per context: asks the runtime for the residency of every persistent kernel, so that no launch ever
// has to (a launch may be inside a stream capture, where such queries are not allowed).
void init_launch_config()
{
    std::call_once(g_occ_once, [] {
#define FRW_Q(K, CACHE, LOGN, ENC) query_residency(K<LOGN, ENC>, CACHE[(LOGN - 9) * 2 + ENC])
#define FRW_QV(LOGN, ENC) query_residency(witness_ntt_verify_kernel<LOGN, ENC>, g_occ_verify[(LOGN - 9) * 2 + ENC])
        FRW_QV(9, 0); FRW_QV(9, 1); FRW_QV(10, 0); FRW_QV(10, 1);
#undef FRW_QV
        FRW_Q(witness_dual_ntt_verify_kernel, g_occ_dual, 9, 0); FRW_Q(witness_dual_ntt_verify_kernel, g_occ_dual, 9, 1);
        FRW_Q(witness_dual_ntt_verify_kernel, g_occ_dual, 10, 0); FRW_Q(witness_dual_ntt_verify_kernel, g_occ_dual, 10, 1);
        FRW_Q(ntt_modq_kernel, g_occ_ntt, 9, 0); FRW_Q(ntt_modq_kernel, g_occ_ntt, 9, 1);
        FRW_Q(ntt_modq_kernel, g_occ_ntt, 10, 0); FRW_Q(ntt_modq_kernel, g_occ_ntt, 10, 1);
#undef FRW_Q
        query_residency(witness_ntt_verify_kernel<9, 2>, g_occ_compact[0]);
        query_residency(witness_ntt_verify_kernel<10, 2>, g_occ_compact[1]);
        query_residency(expand_kernel<9>, g_occ_expand[0]);
        query_residency(expand_kernel<10>, g_occ_expand[1]);
    });
}

// (grid, full) of a witness launch: all-split for small batches, one workgroup per signature up to the resident capacity,
// else the resident grid (or, for short launches, a divisor of the batch close to it) with the ragged tail split.
static void verify_shape(size_t batch, int num_cu, int occ, bool may_split, int &grid, size_t &full)
{
    const size_t cap = (size_t)(occ > 0 ? occ : 2) * (size_t)num_cu;
    if (may_split && batch * PARTS <= cap) {     // measured: 1 signature 167 -> 56 us, 64 signatures 172 -> 95 us; at 256 it loses
        grid = (int)(batch * PARTS);
        full = 0;
        return;
    }
    grid = resident_grid(batch, num_cu, occ);
    full = may_split ? batch / (size_t)grid * (size_t)grid : batch;
}

void launch_shape_witness_ntt_verify(int num_cu, int logn, int enc, size_t batch, int out[4])
{
    const int occ = enc == 2 ? g_occ_compact[logn - 9] : g_occ_verify[(logn - 9) * 2 + (enc & 1)];
    int grid;
    size_t full;
    verify_shape(batch, num_cu, occ, true, grid, full);
    out[0] = grid;
    out[1] = occ;
    out[2] = num_cu;
    out[3] = (int)(batch - full);          // signatures cut into five work items
}

hipError_t launch_witness_ntt_verify(const Tables *tab, int num_cu, int logn, int enc, size_t batch,
                                     const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                                     uint64_t *wit, uint64_t *inst, int32_t *status, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    int grid;
    size_t full;
    verify_shape(batch, num_cu, g_occ_verify[(logn - 9) * 2 + enc], true, grid, full);
#define FRW_LAUNCH(LOGN, ENC)                                                                                              \
    hipLaunchKernelGGL((witness_ntt_verify_kernel<LOGN, ENC>), dim3(grid), dim3(BLOCK), 0, st, tab, batch, full, sig, pk, hm, \
                       (v4u *)wit, (v4u *)inst, status)
    if (logn == 9 && enc == 0) FRW_LAUNCH(9, 0);
    else if (logn == 9) FRW_LAUNCH(9, 1);
    else if (enc == 0) FRW_LAUNCH(10, 0);
    else FRW_LAUNCH(10, 1);
#undef FRW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_witness_ntt_verify_compact(const Tables *tab, int num_cu, int logn, size_t batch,
                                             const uint16_t *sig, const uint16_t *pk, const uint16_t *hm, void *compact,
                                             int32_t *status, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    int grid;
    size_t full;
    verify_shape(batch, num_cu, g_occ_compact[logn - 9], true, grid, full);
    if (logn == 9)
        hipLaunchKernelGGL((witness_ntt_verify_kernel<9, 2>), dim3(grid), dim3(BLOCK), 0, st, tab, batch, full, sig, pk, hm,
                           (v4u *)compact, (v4u *)nullptr, status);
    else
        hipLaunchKernelGGL((witness_ntt_verify_kernel<10, 2>), dim3(grid), dim3(BLOCK), 0, st, tab, batch, full, sig, pk, hm,
                           (v4u *)compact, (v4u *)nullptr, status);
    return hipGetLastError();
}

hipError_t launch_expand(int num_cu, int logn, size_t batch, const void *compact, uint64_t *wit, uint64_t *inst, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
    const int grid = resident_grid(batch, num_cu, g_occ_expand[logn - 9], 8);
    if (logn == 9)
        hipLaunchKernelGGL((expand_kernel<9>), dim3(grid), dim3(BLOCK), 0, st, batch, (const unsigned char *)compact, (v4u *)wit,
                           (v4u *)inst);
    else
        hipLaunchKernelGGL((expand_kernel<10>), dim3(grid), dim3(BLOCK), 0, st, batch, (const unsigned char *)compact, (v4u *)wit,
                           (v4u *)inst);
    return hipGetLastError();
}

hipError_t launch_witness_dual_ntt_verify(const Tables *tab, int num_cu, int logn, int enc,
                                          size_t batch, const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                                          uint64_t *wit, uint64_t *inst, int32_t *status, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
#define FRW_LAUNCH(LOGN, ENC)                                                                                       \
    do {                                                                                                            \
        const int grid = resident_grid(batch, num_cu, g_occ_dual[(LOGN - 9) * 2 + ENC]);                            \
        hipLaunchKernelGGL((witness_dual_ntt_verify_kernel<LOGN, ENC>), dim3(grid), dim3(BLOCK), 0, st, tab,        \
                           batch, sig, pk, hm, (v4u *)wit, (v4u *)inst, status);                                    \
    } while (0)
    if (logn == 9 && enc == 0) FRW_LAUNCH(9, 0);
    else if (logn == 9) FRW_LAUNCH(9, 1);
    else if (enc == 0) FRW_LAUNCH(10, 0);
    else FRW_LAUNCH(10, 1);
#undef FRW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_ntt_modq(const Tables *tab, int num_cu, int logn, int enc, size_t batch, const uint16_t *poly,
                           uint64_t *wit, uint16_t *ntt_out, int32_t *status, hipStream_t st)
{
    if (batch == 0) return hipSuccess;
#define FRW_LAUNCH(LOGN, ENC)                                                                                  \
    do {                                                                                                       \
        const int grid = resident_grid(batch, num_cu, g_occ_ntt[(LOGN - 9) * 2 + ENC]);                        \
        hipLaunchKernelGGL((ntt_modq_kernel<LOGN, ENC>), dim3(grid), dim3(BLOCK), 0, st, tab, batch,           \
                           poly, (v4u *)wit, ntt_out, status);                                                 \
    } while (0)
    if (logn == 9 && enc == 0) FRW_LAUNCH(9, 0);
    else if (logn == 9) FRW_LAUNCH(9, 1);
    else if (enc == 0) FRW_LAUNCH(10, 0);
    else FRW_LAUNCH(10, 1);
#undef FRW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_gadget(int kind, int enc, size_t count, const void *a, const uint64_t *b, uint64_t *out,
                         int32_t *status, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    size_t tiles = (count + WAVE - 1) / WAVE;
    size_t grid = (tiles + WAVES - 1) / WAVES;
    if (grid > 8192) grid = 8192;
#define FRW_LAUNCH(KIND)                                                                                         \
    case KIND:                                                                                                   \
        if (enc == 0) hipLaunchKernelGGL((gadget_kernel<KIND, 0>), dim3((unsigned)grid), dim3(BLOCK), 0, st, count, a, b, (v4u *)out, status); \
        else hipLaunchKernelGGL((gadget_kernel<KIND, 1>), dim3((unsigned)grid), dim3(BLOCK), 0, st, count, a, b, (v4u *)out, status);          \
        break
    switch (kind) {
        FRW_LAUNCH(G_LESS_THAN_Q);
        FRW_LAUNCH(G_MOD_Q);
        FRW_LAUNCH(G_ADD_MOD);
        FRW_LAUNCH(G_L2_ELEM);
        FRW_LAUNCH(G_NORM_512);
        FRW_LAUNCH(G_NORM_1024);
    default: return hipErrorInvalidValue;
    }
#undef FRW_LAUNCH
    return hipGetLastError();
}

hipError_t launch_digest(const uint64_t *buf, size_t words, size_t items, uint64_t *out, hipStream_t st)
{
    if (items == 0) return hipSuccess;
    hipError_t e = hipMemsetAsync(out, 0, items * sizeof(uint64_t), st);
    if (e != hipSuccess) return e;
    int bpi = (int)((words + (size_t)BLOCK * 64 - 1) / ((size_t)BLOCK * 64));
    if (bpi < 1) bpi = 1;
    if (bpi > 64) bpi = 64;
    const size_t grid = items * (size_t)bpi;
    if (grid > 0x7fffffffull) return hipErrorInvalidValue;
    hipLaunchKernelGGL(digest_kernel, dim3((unsigned)grid), dim3(BLOCK), 0, st, buf, words, items,
                       (unsigned long long *)out, bpi);
    return hipGetLastError();
}

}  // namespace frw
