// frw_capi.cpp -- the C ABI of include/frw.h over the gfx950 kernels of frw_kernels.hip.
// Host-side only: contexts, table construction, argument checking, chunked host-buffer variants.
// There is deliberately no CPU compute path here: without a HIP device every entry point that
// would produce a witness returns FRW_E_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <new>
#include <vector>

#include "../../include/frw.h"
#include "frw_arena.h"
#include "frw_device.h"

struct frw_ctx {
    int device;
    int num_cu;
    frw::Tables *d_tables;
    frw::HostArena arena;            // streams, events, device and page-locked memory of the host-buffer entry points
};

namespace {

thread_local char g_last_error[256] = "";

int hip_fail(hipError_t e, const char *what)
{
    snprintf(g_last_error, sizeof g_last_error, "%s: %s", what, hipGetErrorString(e));
    return e == hipErrorOutOfMemory ? FRW_E_OUT_OF_MEMORY : FRW_E_HIP;
}

#define FRW_HIP(call)                                     \
    do {                                                  \
        hipError_t e_ = (call);                           \
        if (e_ != hipSuccess) return hip_fail(e_, #call); \
    } while (0)

uint32_t powmod(uint32_t b, uint32_t e)
{
    uint64_t r = 1, x = b;
    while (e) {
        if (e & 1) r = r * x % frw::Q;
        x = x * x % frw::Q;
        e >>= 1;
    }
    return (uint32_t)r;
}

// falcon-rust NTT_TABLE[i] = 7^bitrev10(i) mod q (script/ntt_param.sage:3-132: Falcon's GMb / R);
// C_k = 2^k q^(k+1) (falcon_ntt.rs:31-39).
void build_tables(frw::Tables &t)
{
    for (uint32_t i = 0; i < 1024; i++) {
        uint32_t r = 0;
        for (int b = 0; b < 10; b++)
            if (i & (1u << b)) r |= 1u << (9 - b);
        t.tw[i] = (uint16_t)powmod(7, r);
        t.itw[i] = (uint16_t)powmod(7, (2048 - r) % 2048);
    }
    uint32_t c[5] = {frw::Q, 0, 0, 0, 0};
    memcpy(t.ck[0], c, sizeof c);
    for (int k = 1; k <= 10; k++) {
        uint64_t carry = 0;
        for (int i = 0; i < 5; i++) {
            carry += (uint64_t)c[i] * (2 * frw::Q);
            c[i] = (uint32_t)carry;
            carry >>= 32;
        }
        memcpy(t.ck[k], c, sizeof c);
    }
}

bool bad_common(const frw_ctx *ctx, int logn, int encoding)
{
    return !ctx || (logn != 9 && logn != 10) || (encoding != FRW_ENC_CANONICAL && encoding != FRW_ENC_MONTGOMERY);
}

}  // namespace

namespace frw {
int record_hip_error(hipError_t e, const char *what) { return hip_fail(e, what); }
}  // namespace frw

extern "C" {

int frw_layout(int logn, frw_layout_t *out)
{
    if (!out || (logn != 9 && logn != 10)) return FRW_E_INVALID_ARG;
    const int n = 1 << logn, nb = logn == 9 ? 50 : 52;
    const int len[FRW_NUM_SEGMENTS] = {n, n, 27 * n, 29 * n, 29 * n, 30 * n, 36 * n, nb};
    int off = 0;
    out->logn = logn;
    out->n = n;
    for (int i = 0; i < FRW_NUM_SEGMENTS; i++) {
        out->seg_off[i] = off;
        out->seg_len[i] = len[i];
        off += len[i];
    }
    out->num_witness = off;
    out->num_instance = 2 * n + 1;
    out->num_constraints = 159 * n + nb + 2;
    return FRW_OK;
}

const char *frw_strerror(int code)
{
    switch (code) {
    case FRW_OK: return "ok";
    case FRW_E_INVALID_ARG: return "invalid argument";
    case FRW_E_NO_DEVICE: return "no usable HIP device (this library has no CPU path)";
    case FRW_E_HIP: return "HIP runtime error (see frw_last_error)";
    case FRW_E_OUT_OF_MEMORY: return "out of device memory";
    case FRW_E_RANGE: return "strict mode: a signature failed its range checks";
    default: return "unknown error";
    }
}

const char *frw_last_error(void) { return g_last_error; }

int frw_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int frw_ctx_create(int device, frw_ctx **out)
{
    if (!out) return FRW_E_INVALID_ARG;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0 || device < 0 || device >= n) {
        snprintf(g_last_error, sizeof g_last_error, "no HIP device %d (%d visible)", device, n);
        return FRW_E_NO_DEVICE;
    }
    FRW_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    FRW_HIP(hipGetDeviceProperties(&prop, device));
    frw_ctx *ctx = new (std::nothrow) frw_ctx;
    if (!ctx) return FRW_E_OUT_OF_MEMORY;
    ctx->device = device;
    ctx->num_cu = prop.multiProcessorCount;
    ctx->d_tables = nullptr;
    frw::Tables host;
    build_tables(host);
    hipError_t e = hipMalloc((void **)&ctx->d_tables, sizeof(frw::Tables));
    if (e == hipSuccess) e = hipMemcpy(ctx->d_tables, &host, sizeof host, hipMemcpyHostToDevice);
    if (e == hipSuccess) e = ctx->arena.init();
    if (e != hipSuccess) {
        frw_ctx_destroy(ctx);
        return hip_fail(e, "context setup");
    }
    // everything a launch needs is known now (residency of the persistent kernels): the _dev entry points are one
    // kernel launch each -- no allocation, no query, no per-launch device state -- and therefore stream-capture safe
    frw::init_launch_config();
    *out = ctx;
    return FRW_OK;
}

void frw_ctx_destroy(frw_ctx *ctx)
{
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    if (ctx->d_tables) (void)hipFree(ctx->d_tables);
    ctx->arena.destroy();
    delete ctx;
}

int frw_diag_host_allocations(frw_ctx *ctx, uint64_t *count)
{
    if (!ctx || !count) return FRW_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->arena.mu);
    *count = ctx->arena.allocations;
    return FRW_OK;
}

int frw_ctx_trim(frw_ctx *ctx)
{
    if (!ctx) return FRW_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(ctx->arena.mu);
    FRW_HIP(hipSetDevice(ctx->device));
    ctx->arena.trim();
    return FRW_OK;
}

int frw_witness_ntt_verify_dev(frw_ctx *ctx, int logn, size_t batch, const uint16_t *d_sig, const uint16_t *d_pk,
                               const uint16_t *d_hm, int encoding, uint64_t *d_witness, uint64_t *d_instance,
                               int32_t *d_status, void *stream)
{
    if (encoding == FRW_ENC_COMPACT)       // d_witness = compact buffer, d_instance unused (may be NULL)
        return frw_witness_ntt_verify_compact_dev(ctx, logn, batch, d_sig, d_pk, d_hm, d_witness, d_status, stream);
    if (bad_common(ctx, logn, encoding)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_sig || !d_pk || !d_hm || !d_witness || !d_instance || !d_status) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_witness_ntt_verify(ctx->d_tables, ctx->num_cu, logn, encoding, batch, d_sig, d_pk, d_hm,
                                           d_witness, d_instance, d_status, (hipStream_t)stream));
    return FRW_OK;
}

int frw_ntt_modq_dev(frw_ctx *ctx, int logn, size_t batch, const uint16_t *d_poly, int encoding, uint64_t *d_witness,
                     uint16_t *d_ntt_out, int32_t *d_status, void *stream)
{
    if (bad_common(ctx, logn, encoding)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_poly || !d_witness || !d_ntt_out || !d_status) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_ntt_modq(ctx->d_tables, ctx->num_cu, logn, encoding, batch, d_poly, d_witness, d_ntt_out,
                                 d_status, (hipStream_t)stream));
    return FRW_OK;
}

namespace {
// Host-buffer driver shared by the two circuits.  Working memory comes from the context's arena (grow-only: a call after
// the first allocates nothing).  Inputs travel through the arena's page-locked buffer, so the three polynomials of a
// chunk are ONE host-to-device copy whatever memory the caller holds them in.  A batch that fits one chunk -- the
// reference's own call pattern is one signature per generate_constraints -- runs on one stream: copy in, kernel, copies
// out, one synchronisation.  Longer batches use two device slots and two streams: while the copy stream drains chunk k
// (5 MB per signature over PCIe), the compute stream already fills chunk k+1.  With output buffers from frw_host_alloc
// (pinned) every copy out is a true asynchronous DMA; with pageable memory the runtime stages the copies itself and the
// overlap degrades gracefully, the results are the same.
int witness_host(frw_ctx *ctx, bool dual, int logn, size_t batch, const uint16_t *sig, const uint16_t *pk,
                 const uint16_t *hm, int encoding, uint64_t *witness, uint64_t *instance, int32_t *status, int strict)
{
    const bool compact = !dual && encoding == FRW_ENC_COMPACT;    // `witness` = compact buffer, `instance` unused
    if (compact ? bad_common(ctx, logn, FRW_ENC_MONTGOMERY) : bad_common(ctx, logn, encoding)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!sig || !pk || !hm || !witness || (!instance && !compact) || !status) return FRW_E_INVALID_ARG;
    frw::HostArena &A = ctx->arena;
    std::lock_guard<std::mutex> lock(A.mu);
    frw::DrainOnExit drain(A);
    FRW_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << logn;
    const size_t nb = logn == 9 ? 50 : 52;
    const size_t wbytes = compact ? frw::compact_layout(logn).bytes : (dual ? 186 * n + 4 + nb : 153 * n + nb) * 32;
    const size_t ibytes = compact ? 0 : (2 * n + 1) * 32;
    const size_t chunk = std::min<size_t>(batch, compact ? 2048 : 256);      // 2 x (<= 1.6 GB) of device witness
    const int nbuf = batch > chunk ? 2 : 1;
    const size_t in_bytes = 3 * chunk * n * 2;                              // sig | pk | hm of one chunk, contiguous
    struct Slot { uint16_t *in; char *wit, *inst; int32_t *st; } slot[2];
    for (int b = 0; b < nbuf; b++) {
        frw::Carve size(nullptr);
        size.take(in_bytes); size.take(chunk * wbytes); size.take(chunk * ibytes + 16); size.take(chunk * sizeof(int32_t));
        FRW_HIP(A.reserve_device(b, size.off));
        frw::Carve c(A.d_slot[b]);
        slot[b].in = c.take<uint16_t>(in_bytes);
        slot[b].wit = c.take<char>(chunk * wbytes);
        slot[b].inst = c.take<char>(chunk * ibytes + 16);
        slot[b].st = c.take<int32_t>(chunk * sizeof(int32_t));
    }
    FRW_HIP(A.reserve_pinned((size_t)nbuf * in_bytes));
    const uint16_t *src[3] = {sig, pk, hm};
    size_t k = 0;
    for (size_t lo = 0; lo < batch; lo += chunk, k++) {
        const size_t cnt = std::min(chunk, batch - lo);
        const int b = (int)(k & 1);
        // slot b was last used by chunk k - 2: its copies out have finished before its memory is written again (the
        // host waits too: it is about to overwrite the page-locked inputs that chunk's H2D copy read)
        if (k >= 2) FRW_HIP(hipEventSynchronize(A.drained[b]));
        uint16_t *stage = (uint16_t *)((char *)A.h_pin + (size_t)b * in_bytes);
        for (int j = 0; j < 3; j++) memcpy(stage + (size_t)j * cnt * n, src[j] + lo * n, cnt * n * 2);
        FRW_HIP(hipMemcpyAsync(slot[b].in, stage, 3 * cnt * n * 2, hipMemcpyHostToDevice, A.compute));
        const uint16_t *d_sig = slot[b].in, *d_pk = d_sig + cnt * n, *d_hm = d_pk + cnt * n;
        if (dual)
            FRW_HIP(frw::launch_witness_dual_ntt_verify(ctx->d_tables, ctx->num_cu, logn, encoding, cnt, d_sig, d_pk, d_hm,
                                                        (uint64_t *)slot[b].wit, (uint64_t *)slot[b].inst, slot[b].st, A.compute));
        else if (compact)
            FRW_HIP(frw::launch_witness_ntt_verify_compact(ctx->d_tables, ctx->num_cu, logn, cnt, d_sig, d_pk, d_hm, slot[b].wit,
                                                           slot[b].st, A.compute));
        else
            FRW_HIP(frw::launch_witness_ntt_verify(ctx->d_tables, ctx->num_cu, logn, encoding, cnt, d_sig, d_pk, d_hm,
                                                   (uint64_t *)slot[b].wit, (uint64_t *)slot[b].inst, slot[b].st, A.compute));
        hipStream_t out = A.compute;
        if (nbuf == 2) {
            FRW_HIP(hipEventRecord(A.done[b], A.compute));
            FRW_HIP(hipStreamWaitEvent(A.copy, A.done[b], 0));
            out = A.copy;
        }
        FRW_HIP(hipMemcpyAsync((char *)witness + lo * wbytes, slot[b].wit, cnt * wbytes, hipMemcpyDeviceToHost, out));
        if (!compact)
            FRW_HIP(hipMemcpyAsync((char *)instance + lo * ibytes, slot[b].inst, cnt * ibytes, hipMemcpyDeviceToHost, out));
        FRW_HIP(hipMemcpyAsync(status + lo, slot[b].st, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, out));
        if (nbuf == 2) FRW_HIP(hipEventRecord(A.drained[b], A.copy));
    }
    FRW_HIP(hipStreamSynchronize(A.compute));
    if (nbuf == 2) FRW_HIP(hipStreamSynchronize(A.copy));
    drain.settled = true;
    bool any_bad = false;
    for (size_t i = 0; i < batch; i++) any_bad |= status[i] != FRW_ST_OK;
    return strict && any_bad ? FRW_E_RANGE : FRW_OK;
}
}  // namespace

int frw_witness_ntt_verify(frw_ctx *ctx, int logn, size_t batch, const uint16_t *sig, const uint16_t *pk,
                           const uint16_t *hm, int encoding, uint64_t *witness, uint64_t *instance, int32_t *status,
                           int strict)
{
    return witness_host(ctx, false, logn, batch, sig, pk, hm, encoding, witness, instance, status, strict);
}

int frw_witness_dual_ntt_verify(frw_ctx *ctx, int logn, size_t batch, const uint16_t *sig, const uint16_t *pk,
                                const uint16_t *hm, int encoding, uint64_t *witness, uint64_t *instance,
                                int32_t *status, int strict)
{
    return witness_host(ctx, true, logn, batch, sig, pk, hm, encoding, witness, instance, status, strict);
}

int frw_witness_dual_ntt_verify_dev(frw_ctx *ctx, int logn, size_t batch, const uint16_t *d_sig, const uint16_t *d_pk,
                                    const uint16_t *d_hm, int encoding, uint64_t *d_witness, uint64_t *d_instance,
                                    int32_t *d_status, void *stream)
{
    if (bad_common(ctx, logn, encoding)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_sig || !d_pk || !d_hm || !d_witness || !d_instance || !d_status) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_witness_dual_ntt_verify(ctx->d_tables, ctx->num_cu, logn, encoding, batch, d_sig,
                                                d_pk, d_hm, d_witness, d_instance, d_status, (hipStream_t)stream));
    return FRW_OK;
}

int frw_compact_layout(int logn, frw_compact_layout_t *out)
{
    if (!out || (logn != 9 && logn != 10)) return FRW_E_INVALID_ARG;
    const frw::CompactLayout c = frw::compact_layout(logn);
    const uint64_t n = (uint64_t)1 << logn, seg = 27 * n;
    out->logn = logn;
    out->n = (int32_t)n;
    out->bytes_per_signature = c.bytes;
    out->small_off = 0;
    out->num_small = c.num_small;
    out->t_off = c.t_off;
    out->num_t = c.num_t;
    out->bits_off = c.bits_off;
    out->num_bit_words = c.bit_words;
    for (int i = 0; i < 5; i++) out->bit_seg_off[i] = (uint64_t)i * seg;
    out->bit_seg_off[5] = 4 * seg + 32 * n;
    out->instance_off = c.instance_off;
    out->num_instance_values = c.num_instance;
    out->status_off = c.status_off;
    return FRW_OK;
}

int frw_witness_ntt_verify_compact_dev(frw_ctx *ctx, int logn, size_t batch, const uint16_t *d_sig, const uint16_t *d_pk,
                                       const uint16_t *d_hm, void *d_compact, int32_t *d_status, void *stream)
{
    if (bad_common(ctx, logn, FRW_ENC_MONTGOMERY)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_sig || !d_pk || !d_hm || !d_compact || !d_status || ((uintptr_t)d_compact & 15)) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_witness_ntt_verify_compact(ctx->d_tables, ctx->num_cu, logn, batch, d_sig, d_pk, d_hm, d_compact,
                                                   d_status, (hipStream_t)stream));
    return FRW_OK;
}

int frw_expand_dev(frw_ctx *ctx, int logn, size_t batch, const void *d_compact, uint64_t *d_witness, uint64_t *d_instance,
                   void *stream)
{
    if (bad_common(ctx, logn, FRW_ENC_MONTGOMERY)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_compact || !d_witness || !d_instance || ((uintptr_t)d_compact & 15)) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_expand(ctx->num_cu, logn, batch, d_compact, d_witness, d_instance, (hipStream_t)stream));
    return FRW_OK;
}

// Host-side expansion of compact signatures received over PCIe: a format conversion (integers -> ark-ff's Montgomery
// representation, bits -> 0 / 1 elements), the counterpart of frw_expand_dev for a consumer that wants arkworks' vectors
// in host memory.  BLS12-381 scalar field, 64-bit limbs.
namespace {
constexpr uint64_t FR_P[4] = {0xffffffff00000001ull, 0x53bda402fffe5bfeull, 0x3339d80809a1d805ull, 0x73eda753299d7d48ull};
constexpr uint64_t FR_R2[4] = {0xc999e990f3f29c6dull, 0x2b6cedcb87925c23ull, 0x05d314967254398full, 0x0748d9d99f59ff11ull};
constexpr uint64_t FR_ONE[4] = {0x00000001fffffffeull, 0x5884b7fa00034802ull, 0x998c4fefecbc4ff5ull, 0x1824b159acc5056full};
constexpr uint64_t FR_INV = 0xfffffffeffffffffull;          // -p^-1 mod 2^64

// out = x * 2^256 mod p for x < 2^192 given as three 64-bit limbs: Montgomery product of x and R^2 (CIOS)
inline void fr_to_montgomery(const uint64_t x[3], uint64_t out[4])
{
    typedef unsigned __int128 u128;
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        const uint64_t xi = i < 3 ? x[i] : 0;
        u128 c = 0;
        for (int j = 0; j < 4; j++) {
            c += (u128)xi * FR_R2[j] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[4] = (uint64_t)c;
        t[5] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * FR_INV;
        c = ((u128)m * FR_P[0] + t[0]) >> 64;
        for (int j = 1; j < 4; j++) {
            c += (u128)m * FR_P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[4];
        t[3] = (uint64_t)c;
        t[4] = t[5] + (uint64_t)(c >> 64);
    }
    uint64_t d[4];
    unsigned borrow = 0;
    for (int j = 0; j < 4; j++) {
        const u128 diff = (u128)t[j] - FR_P[j] - borrow;
        d[j] = (uint64_t)diff;
        borrow = (unsigned)((diff >> 64) & 1);
    }
    const bool ge = t[4] != 0 || !borrow;
    for (int j = 0; j < 4; j++) out[j] = ge ? d[j] : t[j];
}
}  // namespace

int frw_expand_host(int logn, size_t batch, const void *compact, uint64_t *witness, uint64_t *instance)
{
    if ((logn != 9 && logn != 10) || (batch && (!compact || !witness || !instance))) return FRW_E_INVALID_ARG;
    const frw::CompactLayout c = frw::compact_layout(logn);
    const size_t n = (size_t)1 << logn, nb = logn == 9 ? 50 : 52, W = 153 * n + nb, I = 2 * n + 1;
    for (size_t s = 0; s < batch; s++) {
        const unsigned char *base = (const unsigned char *)compact + s * c.bytes;
        const uint32_t *small = (const uint32_t *)base;                   // next small value
        const uint32_t *tq = (const uint32_t *)(base + c.t_off);           // next 5-limb quotient
        const uint32_t *bits = (const uint32_t *)(base + c.bits_off);
        const uint32_t *ins = (const uint32_t *)(base + c.instance_off);
        size_t bit = 0;
        uint64_t *w = witness + s * W * 4, *in = instance + s * I * 4;
        if (*(const uint32_t *)(base + c.status_off) == (uint32_t)FRW_ST_COEFF_RANGE) {   // rejected: zeros, as the direct path
            memset(w, 0, W * 32);
            memset(in, 0, I * 32);
            continue;
        }
        auto value = [&](int count) {
            for (int i = 0; i < count; i++, w += 4) {
                const uint64_t x[3] = {*small++, 0, 0};
                fr_to_montgomery(x, w);
            }
        };
        auto quotient = [&]() {
            const uint64_t x[3] = {(uint64_t)tq[0] | ((uint64_t)tq[1] << 32), (uint64_t)tq[2] | ((uint64_t)tq[3] << 32), tq[4]};
            fr_to_montgomery(x, w);
            tq += 5;
            w += 4;
        };
        auto booleans = [&](int count) {
            for (int i = 0; i < count; i++, bit++, w += 4) {
                if ((bits[bit >> 5] >> (bit & 31)) & 1u) memcpy(w, FR_ONE, 32);
                else memset(w, 0, 32);
            }
        };
        memcpy(in, FR_ONE, 32);
        for (size_t k = 0; k < 2 * n; k++) {
            const uint64_t x[3] = {ins[k], 0, 0};
            fr_to_montgomery(x, in + 4 + 4 * k);
        }
        value((int)(2 * n));                                              // S0, S1
        booleans((int)(27 * n));                                          // S2
        // S3, S4: [t, b, 27 booleans]; the b values of S3 (N) and of S4 (N) follow sig and v in `small`, the quotients of
        // S3 then S4 are in `t`
        for (int seg = 0; seg < 2; seg++)
            for (size_t k = 0; k < n; k++) { quotient(); value(1); booleans(27); }
        for (size_t k = 0; k < n; k++) { value(3); booleans(27); }         // S5: [prod, t, c, 27 booleans]
        for (size_t k = 0; k < 2 * n; k++) { booleans(16); value(2); }     // S6: [16 booleans, r, sq]
        bit = (4 * c.seg_words + n) * 32;                                 // S7 has two words of its own
        booleans((int)nb);
    }
    return FRW_OK;
}

int frw_diag_launch_shape(frw_ctx *ctx, int logn, int encoding, size_t batch, int32_t out[4])
{
    if (!ctx || !out || (logn != 9 && logn != 10) || encoding < 0 || encoding > 2) return FRW_E_INVALID_ARG;
    int o[4];
    frw::launch_shape_witness_ntt_verify(ctx->num_cu, logn, encoding, batch, o);
    for (int i = 0; i < 4; i++) out[i] = o[i];
    return FRW_OK;
}

int frw_layout_dual(int logn, frw_layout_dual_t *out)
{
    if (!out || (logn != 9 && logn != 10)) return FRW_E_INVALID_ARG;
    const int n = 1 << logn, nb = logn == 9 ? 50 : 52;
    const int len[FRW_NUM_SEGMENTS_DUAL] = {n, n, n, 2, n, n, n, 2, 29 * n, 29 * n, 29 * n, 29 * n, 60 * n, 4 * n, nb};
    int off = 0;
    out->logn = logn;
    out->n = n;
    for (int i = 0; i < FRW_NUM_SEGMENTS_DUAL; i++) {
        out->seg_off[i] = off;
        out->seg_len[i] = len[i];
        off += len[i];
    }
    out->num_witness = off;
    out->num_instance = 2 * n + 1;
    out->num_constraints = 189 * n + 10 + nb;
    return FRW_OK;
}

int frw_ntt_modq(frw_ctx *ctx, int logn, size_t batch, const uint16_t *poly, int encoding, uint64_t *witness,
                 uint16_t *ntt_out, int32_t *status)
{
    if (bad_common(ctx, logn, encoding)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!poly || !witness || !ntt_out || !status) return FRW_E_INVALID_ARG;
    frw::HostArena &A = ctx->arena;
    std::lock_guard<std::mutex> lock(A.mu);
    frw::DrainOnExit drain(A);
    FRW_HIP(hipSetDevice(ctx->device));
    // same pipeline as witness_host: arena memory, one stream for a batch that fits a chunk, else the D2H of chunk k
    // overlaps the kernel of chunk k+1
    const size_t n = (size_t)1 << logn, wbytes = 29 * n * 32;
    const size_t chunk = std::min<size_t>(batch, 2048);
    const int nbuf = batch > chunk ? 2 : 1;
    struct Slot { uint16_t *in, *out; char *wit; int32_t *st; } slot[2];
    for (int b = 0; b < nbuf; b++) {
        frw::Carve size(nullptr);
        size.take(chunk * n * 2); size.take(chunk * n * 2); size.take(chunk * wbytes); size.take(chunk * sizeof(int32_t));
        FRW_HIP(A.reserve_device(b, size.off));
        frw::Carve c(A.d_slot[b]);
        slot[b].in = c.take<uint16_t>(chunk * n * 2);
        slot[b].out = c.take<uint16_t>(chunk * n * 2);
        slot[b].wit = c.take<char>(chunk * wbytes);
        slot[b].st = c.take<int32_t>(chunk * sizeof(int32_t));
    }
    FRW_HIP(A.reserve_pinned((size_t)nbuf * chunk * n * 2));
    size_t k = 0;
    for (size_t lo = 0; lo < batch; lo += chunk, k++) {
        const size_t cnt = std::min(chunk, batch - lo);
        const int b = (int)(k & 1);
        if (k >= 2) FRW_HIP(hipEventSynchronize(A.drained[b]));
        uint16_t *stage = (uint16_t *)((char *)A.h_pin + (size_t)b * chunk * n * 2);
        memcpy(stage, poly + lo * n, cnt * n * 2);
        FRW_HIP(hipMemcpyAsync(slot[b].in, stage, cnt * n * 2, hipMemcpyHostToDevice, A.compute));
        FRW_HIP(frw::launch_ntt_modq(ctx->d_tables, ctx->num_cu, logn, encoding, cnt, slot[b].in, (uint64_t *)slot[b].wit,
                                     slot[b].out, slot[b].st, A.compute));
        hipStream_t out = A.compute;
        if (nbuf == 2) {
            FRW_HIP(hipEventRecord(A.done[b], A.compute));
            FRW_HIP(hipStreamWaitEvent(A.copy, A.done[b], 0));
            out = A.copy;
        }
        FRW_HIP(hipMemcpyAsync((char *)witness + lo * wbytes, slot[b].wit, cnt * wbytes, hipMemcpyDeviceToHost, out));
        FRW_HIP(hipMemcpyAsync(ntt_out + lo * n, slot[b].out, cnt * n * 2, hipMemcpyDeviceToHost, out));
        FRW_HIP(hipMemcpyAsync(status + lo, slot[b].st, cnt * sizeof(int32_t), hipMemcpyDeviceToHost, out));
        if (nbuf == 2) FRW_HIP(hipEventRecord(A.drained[b], A.copy));
    }
    FRW_HIP(hipStreamSynchronize(A.compute));
    if (nbuf == 2) FRW_HIP(hipStreamSynchronize(A.copy));
    drain.settled = true;
    return FRW_OK;
}

int frw_hash_to_point_dev(frw_ctx *ctx, int logn, size_t batch, const uint8_t *d_nonces, const uint8_t *d_msgs,
                          const uint64_t *d_msg_off, uint16_t *d_hm, void *stream)
{
    if (bad_common(ctx, logn, FRW_ENC_CANONICAL)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_nonces || !d_msgs || !d_msg_off || !d_hm) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_hash_to_point(logn, batch, d_nonces, d_msgs, d_msg_off, d_hm, (hipStream_t)stream));
    return FRW_OK;
}

int frw_decode_public_keys_dev(frw_ctx *ctx, int logn, size_t batch, const uint8_t *d_pk_bytes, uint16_t *d_pk,
                               int32_t *d_status, void *stream)
{
    if (bad_common(ctx, logn, FRW_ENC_CANONICAL)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_pk_bytes || !d_pk || !d_status) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_decode_public_keys(logn, batch, d_pk_bytes, d_pk, d_status, (hipStream_t)stream));
    return FRW_OK;
}

int frw_decode_signatures_dev(frw_ctx *ctx, int logn, size_t batch, const uint8_t *d_sig_bytes, size_t sig_len,
                              uint16_t *d_sig, uint8_t *d_nonce_out, int32_t *d_status, void *stream)
{
    if (bad_common(ctx, logn, FRW_ENC_CANONICAL)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!d_sig_bytes || !d_sig || !d_status || sig_len <= 1 + FRW_NONCE_LEN) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_decode_signatures(logn, batch, d_sig_bytes, sig_len, d_sig, d_nonce_out, d_status, (hipStream_t)stream));
    return FRW_OK;
}

int frw_prepare_inputs(frw_ctx *ctx, int logn, size_t batch, const uint8_t *pk_bytes, const uint8_t *sig_bytes,
                       size_t sig_len, const uint8_t *msgs, const uint64_t *msg_off, uint16_t *sig, uint16_t *pk,
                       uint16_t *hm, int32_t *status)
{
    if (bad_common(ctx, logn, FRW_ENC_CANONICAL)) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    if (!pk_bytes || !sig_bytes || !msg_off || !sig || !pk || !hm || !status || sig_len <= 1 + FRW_NONCE_LEN) return FRW_E_INVALID_ARG;
    if (!msgs && msg_off[batch] != msg_off[0]) return FRW_E_INVALID_ARG;
    for (size_t i = 0; i < batch; i++)
        if (msg_off[i + 1] < msg_off[i]) return FRW_E_INVALID_ARG;        // offsets must be non-decreasing
    frw::HostArena &A = ctx->arena;
    std::lock_guard<std::mutex> lock(A.mu);
    frw::DrainOnExit drain(A);
    FRW_HIP(hipSetDevice(ctx->device));
    const size_t n = (size_t)1 << logn, pk_len = FRW_PK_LEN(logn);
    const size_t msg_bytes = (size_t)(msg_off[batch] - msg_off[0]);
    // one device slot, carved; the encoded inputs travel in ONE copy through the page-locked buffer, the results and both
    // status vectors come back in ONE copy
    frw::Carve hs(nullptr);                                             // layout of the staged inputs == their device layout
    const size_t o_pkb = hs.off; hs.take(batch * pk_len);
    const size_t o_sigb = hs.off; hs.take(batch * sig_len);
    const size_t o_msgs = hs.off; hs.take(msg_bytes ? msg_bytes : 1);
    const size_t o_off = hs.off; hs.take((batch + 1) * sizeof(uint64_t));
    const size_t in_total = hs.off;
    frw::Carve os(nullptr);                                             // layout of the results
    const size_t o_sig = os.off; os.take(batch * n * 2);
    const size_t o_pk = os.off; os.take(batch * n * 2);
    const size_t o_hm = os.off; os.take(batch * n * 2);
    const size_t o_st1 = os.off; os.take(batch * sizeof(int32_t));
    const size_t o_st2 = os.off; os.take(batch * sizeof(int32_t));
    const size_t out_total = os.off;
    const size_t nonce_bytes = (batch * FRW_NONCE_LEN + 255) & ~(size_t)255;
    FRW_HIP(A.reserve_device(0, in_total + out_total + nonce_bytes));
    FRW_HIP(A.reserve_pinned(std::max(in_total, out_total)));
    char *d_in = (char *)A.d_slot[0], *d_out = d_in + in_total, *d_nonce = d_out + out_total, *h = (char *)A.h_pin;
    memcpy(h + o_pkb, pk_bytes, batch * pk_len);
    memcpy(h + o_sigb, sig_bytes, batch * sig_len);
    if (msg_bytes) memcpy(h + o_msgs, msgs + msg_off[0], msg_bytes);
    uint64_t *off = (uint64_t *)(h + o_off);
    for (size_t i = 0; i <= batch; i++) off[i] = msg_off[i] - msg_off[0];
    hipStream_t st = A.compute;
    FRW_HIP(hipMemcpyAsync(d_in, h, in_total, hipMemcpyHostToDevice, st));
    FRW_HIP(hipMemsetAsync(d_nonce, 0, batch * FRW_NONCE_LEN, st));
    FRW_HIP(frw::launch_decode_public_keys(logn, batch, (const uint8_t *)(d_in + o_pkb), (uint16_t *)(d_out + o_pk),
                                           (int32_t *)(d_out + o_st1), st));
    FRW_HIP(frw::launch_decode_signatures(logn, batch, (const uint8_t *)(d_in + o_sigb), sig_len, (uint16_t *)(d_out + o_sig),
                                          (uint8_t *)d_nonce, (int32_t *)(d_out + o_st2), st));
    FRW_HIP(frw::launch_hash_to_point(logn, batch, (const uint8_t *)d_nonce, (const uint8_t *)(d_in + o_msgs),
                                      (const uint64_t *)(d_in + o_off), (uint16_t *)(d_out + o_hm), st));
    FRW_HIP(hipStreamSynchronize(st));                                   // the staged inputs have been read
    FRW_HIP(hipMemcpyAsync(h, d_out, out_total, hipMemcpyDeviceToHost, st));
    FRW_HIP(hipStreamSynchronize(st));
    memcpy(sig, h + o_sig, batch * n * 2);
    memcpy(pk, h + o_pk, batch * n * 2);
    memcpy(hm, h + o_hm, batch * n * 2);
    const int32_t *st1 = (const int32_t *)(h + o_st1), *st2 = (const int32_t *)(h + o_st2);
    for (size_t i = 0; i < batch; i++) status[i] = st2[i] != FRW_ST_OK ? st2[i] : st1[i];
    return FRW_OK;
}

int frw_gadget_block_len(int kind)
{
    static const int len[6] = {27, 29, 29, 18, 50, 52};
    return kind >= 0 && kind < 6 ? len[kind] : FRW_E_INVALID_ARG;
}

int frw_gadget_dev(frw_ctx *ctx, int kind, size_t count, const void *d_a, const uint64_t *d_b, int encoding,
                   uint64_t *d_out, int32_t *d_status, void *stream)
{
    if (bad_common(ctx, 10, encoding) || frw_gadget_block_len(kind) < 0) return FRW_E_INVALID_ARG;
    if (count == 0) return FRW_OK;
    if (!d_a || !d_out || (kind == FRW_G_ADD_MOD && !d_b)) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_gadget(kind, encoding, count, d_a, d_b, d_out, d_status, (hipStream_t)stream));
    return FRW_OK;
}

int frw_gadget(frw_ctx *ctx, int kind, size_t count, const void *a, const uint64_t *b, int encoding, uint64_t *out,
               int32_t *status)
{
    if (bad_common(ctx, 10, encoding) || frw_gadget_block_len(kind) < 0) return FRW_E_INVALID_ARG;
    if (count == 0) return FRW_OK;
    if (!a || !out || (kind == FRW_G_ADD_MOD && !b)) return FRW_E_INVALID_ARG;
    frw::HostArena &A = ctx->arena;
    std::lock_guard<std::mutex> lock(A.mu);
    frw::DrainOnExit drain(A);
    FRW_HIP(hipSetDevice(ctx->device));
    const size_t in_bytes = count * (kind == FRW_G_MOD_Q ? 20 : 8), b_bytes = kind == FRW_G_ADD_MOD ? count * 8 : 0;
    const size_t out_bytes = count * (size_t)frw_gadget_block_len(kind) * 32;
    frw::Carve size(nullptr);
    size.take(in_bytes); size.take(b_bytes); size.take(out_bytes); size.take(count * sizeof(int32_t));
    FRW_HIP(A.reserve_device(0, size.off));
    frw::Carve c(A.d_slot[0]);
    void *d_a = c.take(in_bytes);
    uint64_t *d_b = c.take<uint64_t>(b_bytes);
    uint64_t *d_out = c.take<uint64_t>(out_bytes);
    int32_t *d_st = c.take<int32_t>(count * sizeof(int32_t));
    hipStream_t st = A.compute;
    FRW_HIP(hipMemcpyAsync(d_a, a, in_bytes, hipMemcpyHostToDevice, st));
    if (b_bytes) FRW_HIP(hipMemcpyAsync(d_b, b, b_bytes, hipMemcpyHostToDevice, st));
    FRW_HIP(frw::launch_gadget(kind, encoding, count, d_a, b_bytes ? d_b : nullptr, d_out, d_st, st));
    FRW_HIP(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, st));
    if (status) FRW_HIP(hipMemcpyAsync(status, d_st, count * sizeof(int32_t), hipMemcpyDeviceToHost, st));
    FRW_HIP(hipStreamSynchronize(st));
    return FRW_OK;
}

int frw_digest_dev(frw_ctx *ctx, const uint64_t *d_buf, size_t words_per_item, size_t items, uint64_t *d_out,
                   void *stream)
{
    if (!ctx || !d_buf || !d_out) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_digest(d_buf, words_per_item, items, d_out, (hipStream_t)stream));
    return FRW_OK;
}

int frw_diag_write_stream_dev(frw_ctx *ctx, void *d_buf, size_t bytes, size_t slab_bytes, void *stream)
{
    if (!ctx || !d_buf || slab_bytes < 16) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(frw::launch_write_stream(d_buf, bytes, slab_bytes, ctx->num_cu, (hipStream_t)stream));
    return FRW_OK;
}

int frw_diag_valu_rates(frw_ctx *ctx, double out[4])
{
    if (!ctx || !out) return FRW_E_INVALID_ARG;
    frw::HostArena &A = ctx->arena;
    std::lock_guard<std::mutex> lock(A.mu);
    frw::DrainOnExit drain(A);
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(A.reserve_device(0, (size_t)ctx->num_cu * 8192));
    FRW_HIP(frw::diag_valu_rates(ctx->num_cu, A.d_slot[0], out, A.compute));
    return FRW_OK;
}

int frw_host_alloc(frw_ctx *ctx, size_t bytes, void **ptr)
{
    if (!ctx || !ptr) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(hipHostMalloc(ptr, bytes ? bytes : 1, hipHostMallocDefault));
    return FRW_OK;
}

int frw_host_free(frw_ctx *ctx, void *ptr)
{
    // ctx may be NULL (page-locked memory does not belong to a device; a buffer may outlive the context it came from)
    if (ctx) FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(hipHostFree(ptr));
    return FRW_OK;
}

int frw_malloc(frw_ctx *ctx, size_t bytes, void **d_ptr)
{
    if (!ctx || !d_ptr) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(hipMalloc(d_ptr, bytes));
    return FRW_OK;
}

int frw_free(frw_ctx *ctx, void *d_ptr)
{
    if (!ctx) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(hipFree(d_ptr));
    return FRW_OK;
}

int frw_memcpy_h2d(frw_ctx *ctx, void *d_dst, const void *src, size_t bytes)
{
    if (!ctx) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(hipMemcpy(d_dst, src, bytes, hipMemcpyHostToDevice));
    return FRW_OK;
}

int frw_memcpy_d2h(frw_ctx *ctx, void *dst, const void *d_src, size_t bytes)
{
    if (!ctx) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(hipMemcpy(dst, d_src, bytes, hipMemcpyDeviceToHost));
    return FRW_OK;
}

int frw_synchronize(frw_ctx *ctx, void *stream)
{
    if (!ctx) return FRW_E_INVALID_ARG;
    FRW_HIP(hipSetDevice(ctx->device));
    FRW_HIP(hipStreamSynchronize((hipStream_t)stream));
    return FRW_OK;
}

}  // extern "C"
