// frw_arena.h -- working memory of the host-buffer entry points (frw_witness_ntt_verify, frw_ntt_modq, frw_gadget,
// frw_prepare_inputs, frw_qap_witness_map): device buffers, a page-locked bounce buffer, two streams and four events that
// belong to the context (or the R1CS handle) and only ever grow.
//
// Why: the reference's consumer calls generate_constraints once per signature (examples/constraint_counts.rs:61-63,
// examples/pok_sig.rs:24-32), so the drop-in entry point is called with batch = 1 over and over.  Allocating and freeing
// five device buffers per call (hipFree synchronises the device) cost more than the 43 us the kernel takes for one
// signature; with the arena a call after the first allocates nothing (frw_diag_host_allocations counts).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>
#include <mutex>

namespace frw {

struct HostArena {
    static constexpr int SLOTS = 2;       // chunk k is copied out while chunk k+1 is computed
    std::mutex mu;                        // the host-buffer entry points of one owner run one at a time
    void *d_slot[SLOTS] = {nullptr, nullptr};
    size_t d_cap[SLOTS] = {0, 0};
    void *h_pin = nullptr;                // page-locked: inputs on their way in, status words on their way out
    size_t h_cap = 0;
    hipStream_t compute = nullptr, copy = nullptr;
    hipEvent_t done[SLOTS] = {nullptr, nullptr}, drained[SLOTS] = {nullptr, nullptr};
    uint64_t allocations = 0;             // hipMalloc + hipHostMalloc calls made so far

    hipError_t init()
    {
        hipError_t e = hipStreamCreateWithFlags(&compute, hipStreamNonBlocking);
        if (e == hipSuccess) e = hipStreamCreateWithFlags(&copy, hipStreamNonBlocking);
        for (int b = 0; b < SLOTS && e == hipSuccess; b++) {
            e = hipEventCreateWithFlags(&done[b], hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventCreateWithFlags(&drained[b], hipEventDisableTiming);
        }
        return e;
    }
    // everything the arena holds goes back to the device; the arena stays usable (the next call allocates again)
    void trim()
    {
        for (int b = 0; b < SLOTS; b++) {
            if (d_slot[b]) (void)hipFree(d_slot[b]);
            d_slot[b] = nullptr;
            d_cap[b] = 0;
        }
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        h_cap = 0;
    }
    void destroy()
    {
        trim();
        for (int b = 0; b < SLOTS; b++) {
            if (done[b]) (void)hipEventDestroy(done[b]);
            if (drained[b]) (void)hipEventDestroy(drained[b]);
            done[b] = drained[b] = nullptr;
        }
        if (compute) (void)hipStreamDestroy(compute);
        if (copy) (void)hipStreamDestroy(copy);
        compute = copy = nullptr;
    }
    hipError_t reserve_device(int slot, size_t bytes)
    {
        if (d_cap[slot] >= bytes) return hipSuccess;
        if (d_slot[slot]) (void)hipFree(d_slot[slot]);      // no call is in flight: the entry points synchronise before they return, DrainOnExit on every other way out
        d_slot[slot] = nullptr;
        d_cap[slot] = 0;
        const hipError_t e = hipMalloc(&d_slot[slot], bytes);
        if (e == hipSuccess) {
            d_cap[slot] = bytes;
            allocations++;
        }
        return e;
    }
    hipError_t reserve_pinned(size_t bytes)
    {
        if (h_cap >= bytes) return hipSuccess;
        if (h_pin) (void)hipHostFree(h_pin);
        h_pin = nullptr;
        h_cap = 0;
        const hipError_t e = hipHostMalloc(&h_pin, bytes, hipHostMallocDefault);
        if (e == hipSuccess) {
            h_cap = bytes;
            allocations++;
        }
        return e;
    }
};

// An entry point that gives up half way through its pipeline (a HIP error after the first enqueue) must not hand the caller's
// buffers back while copies into them, or kernels on the arena's slots, are still running -- the caller may free them, and the
// next call reuses the slot.  Declared right after the lock; `settled` is set once the call's own final synchronisations have
// been made, so the successful path pays nothing.
struct DrainOnExit {
    HostArena &a;
    bool settled = false;
    explicit DrainOnExit(HostArena &arena) : a(arena) {}
    ~DrainOnExit()
    {
        if (settled) return;
        if (a.compute) (void)hipStreamSynchronize(a.compute);
        if (a.copy) (void)hipStreamSynchronize(a.copy);
    }
};

// carves 256-byte aligned pieces out of one slot (run once with base = nullptr to size the slot)
struct Carve {
    char *base;
    size_t off = 0;
    explicit Carve(void *b) : base((char *)b) {}
    template <class T = void>
    T *take(size_t bytes)
    {
        T *p = (T *)(base + off);
        off += (bytes + 255) & ~(size_t)255;
        return p;
    }
};

}  // namespace frw
