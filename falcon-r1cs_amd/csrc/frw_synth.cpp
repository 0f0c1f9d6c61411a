// frw_synth.cpp -- synthetic, always-valid (sig, pk, hm) triples (host side).
//
// Stands where the reference's tests and examples call KeyPair::keygen() + sign_with_seed()
// (falcon-r1cs/src/circuits/falcon_ntt.rs:134-138, examples/constraint_counts.rs:49-58): the
// circuit never checks the hash, hm is a public input (falcon_ntt.rs:65-67), so any triple with
// hm = v + sig*pk mod (x^N+1, q) and a short (sig, v) is a valid statement.
//
//   sig[i], v[i] = round(sigma * g), g ~ Irwin-Hall(12) standard normal approximation, mapped to [0,q)
//   sigma        = 165.7366 (Falcon-512) / 168.3886 (Falcon-1024)   (Falcon parameter sets)
//   pk[i]        = uniform in [0, q)
//   hm           = v + sig (*) pk   via the negacyclic NTT mod q
// Integer arithmetic only, counter based (splitmix64 keyed by seed, triple index, attempt), so
// every host, rank and run draws bit-identical inputs.
#include <stdint.h>
#include <stddef.h>
#include <vector>

#include "../../include/frw.h"

namespace {

constexpr uint32_t Q = 12289;

struct Rng {
    uint64_t s;
    uint64_t next()
    {
        uint64_t x = (s += 0x9E3779B97F4A7C15ull);
        x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
        x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
        return x ^ (x >> 31);
    }
};

uint32_t powmod(uint32_t b, uint32_t e)
{
    uint64_t r = 1, x = b;
    while (e) {
        if (e & 1) r = r * x % Q;
        x = x * x % Q;
        e >>= 1;
    }
    return (uint32_t)r;
}

struct Tw {
    uint32_t f[1024], i[1024];
    Tw()
    {
        for (uint32_t k = 0; k < 1024; k++) {
            uint32_t r = 0;
            for (int b = 0; b < 10; b++)
                if (k & (1u << b)) r |= 1u << (9 - b);
            f[k] = powmod(7, r);
            i[k] = powmod(7, (2048 - r) % 2048);
        }
    }
};

void ntt(uint32_t *a, int logn, const Tw &tw)
{
    const int n = 1 << logn;
    int t = n;
    for (int m = 1; m < n; m <<= 1) {
        const int ht = t >> 1;
        for (int i = 0, j1 = 0; i < m; i++, j1 += t) {
            const uint32_t s = tw.f[m + i];
            for (int j = j1; j < j1 + ht; j++) {
                const uint32_t u = a[j], v = a[j + ht] * s % Q;
                a[j] = (u + v) % Q;
                a[j + ht] = (u + Q - v) % Q;
            }
        }
        t = ht;
    }
}

void intt(uint32_t *a, int logn, const Tw &tw)
{
    const int n = 1 << logn;
    int t = 1;
    for (int m = n; m > 1; m >>= 1) {
        const int hm = m >> 1;
        for (int i = 0, j1 = 0; i < hm; i++, j1 += 2 * t) {
            const uint32_t s = tw.i[hm + i];
            for (int j = j1; j < j1 + t; j++) {
                const uint32_t u = a[j], v = a[j + t];
                a[j] = (u + v) % Q;
                a[j + t] = (u + Q - v) % Q * s % Q;
            }
        }
        t <<= 1;
    }
    const uint32_t ninv = powmod((uint32_t)n, Q - 2);
    for (int i = 0; i < n; i++) a[i] = a[i] * ninv % Q;
}

// round(sigma * Irwin-Hall(12)) with sigma in 16.16 fixed point
int32_t gauss(Rng &r, int64_t sigma_fx)
{
    int64_t s = 0;
    for (int k = 0; k < 3; k++) {
        const uint64_t x = r.next();
        s += (int64_t)(x & 0xffff) + (int64_t)((x >> 16) & 0xffff) + (int64_t)((x >> 32) & 0xffff) + (int64_t)(x >> 48);
    }
    s -= 6 * 65535;                                  // mean of twelve U[0, 65535]
    return (int32_t)((s * sigma_fx + ((int64_t)1 << 31)) >> 32);   // / 2^16 (unit variance) / 2^16 (fixed point), round half up
}

}  // namespace

extern "C" int frw_synth_triples(int logn, size_t batch, uint64_t seed, uint64_t first_index, uint16_t *sig,
                                 uint16_t *pk, uint16_t *hm)
{
    if ((logn != 9 && logn != 10) || (batch && (!sig || !pk || !hm))) return FRW_E_INVALID_ARG;
    static const Tw tw;
    const int n = 1 << logn;
    const int64_t sigma_fx = logn == 9 ? 10861714 : 11035515;     // round(165.7366 * 2^16), round(168.3886 * 2^16)
    const int64_t bound = logn == 9 ? 34034726 : 70265242;        // SIG_L2_BOUND (range_proofs.rs:104,196)
    std::vector<uint32_t> s(n), v(n), h(n);
    for (size_t b = 0; b < batch; b++) {
        const uint64_t idx = first_index + b;
        for (uint64_t attempt = 0;; attempt++) {
            Rng r{seed ^ (idx * 0xD1342543DE82EF95ull) ^ (attempt * 0xA0761D6478BD642Full) ^ ((uint64_t)logn << 56)};
            (void)r.next();
            int64_t norm = 0;
            for (int i = 0; i < n; i++) {
                const int32_t a = gauss(r, sigma_fx), c = gauss(r, sigma_fx);
                norm += (int64_t)a * a + (int64_t)c * c;
                s[i] = (uint32_t)(a < 0 ? a + (int32_t)Q : a);
                v[i] = (uint32_t)(c < 0 ? c + (int32_t)Q : c);
            }
            if (norm >= bound) continue;                           // rare: redraw
            for (int i = 0; i < n; i++) h[i] = (uint32_t)((r.next() >> 32) * Q >> 32);
            uint16_t *os = sig + b * n, *op = pk + b * n, *oh = hm + b * n;
            for (int i = 0; i < n; i++) { os[i] = (uint16_t)s[i]; op[i] = (uint16_t)h[i]; }
            ntt(s.data(), logn, tw);
            ntt(h.data(), logn, tw);
            ntt(v.data(), logn, tw);
            for (int i = 0; i < n; i++) h[i] = (v[i] + s[i] * h[i]) % Q;
            intt(h.data(), logn, tw);
            for (int i = 0; i < n; i++) oh[i] = (uint16_t)h[i];
            break;
        }
    }
    return FRW_OK;
}
