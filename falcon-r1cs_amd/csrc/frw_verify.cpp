// Groth16 verification over BLS12-381: ark-groth16 0.3.0 verifier.rs (prepare_verifying_key, prepare_inputs,
// verify_proof_with_prepared_inputs) -- what examples/pok_sig.rs:34-47 ends with.  HOST code, like the reference's: one
// proof is a 2N-term sum of small multiples of the key's gamma_abc points, three Miller loops and one final exponentiation,
// ~15 ms on one core; a batch runs one proof per host thread.  The device is not involved and need not be present.
#include <stdint.h>
#include <algorithm>
#include <atomic>
#include <cstring>
#include <new>
#include <system_error>
#include <thread>
#include <vector>

#include "../../include/frw.h"
#include "frw_pairing.h"
#include "host/frw_host.hpp"

namespace {
using namespace frw;
using namespace frw::pairing;
using frw::host::Fr;

// ark-ff affine point (x | y, 6 x u64 each, x 2^384; all zero = the point at infinity) -> both forms used here
G1Affine29 g1_lazy_from_ark(const uint64_t *w)
{
    G1Affine29 p;
    uint64_t any = 0;
    for (int k = 0; k < 12; k++) any |= w[k];
    p.inf = any == 0;
    p.x = fq_canonical(fq_from_ark((const uint32_t *)w));
    p.y = fq_canonical(fq_from_ark((const uint32_t *)(w + 6)));
    return p;
}
G1 g1_strict(const G1Affine29 &p) { G1 r; r.x.v = fq_canonical(p.x); r.y.v = fq_canonical(p.y); r.inf = p.inf; return r; }
G2 g2_from_ark(const uint64_t *w)
{
    G2 p;
    uint64_t any = 0;
    for (int k = 0; k < 24; k++) any |= w[k];
    p.inf = any == 0;
    p.x = fp2_from_ark(w);
    p.y = fp2_from_ark(w + 12);
    return p;
}
G2 g2_neg(const G2 &p) { G2 r = p; r.y = fp2_neg(p.y); return r; }

// r P = O?  (r = the group order: the scalar field's modulus)
template <class F> bool in_subgroup(const AffineT<F> &p)
{
    if (p.inf) return true;
    XyzzT<F> acc = pt_identity<F>();
    for (int bit = 254; bit >= 0; bit--) {
        acc = pt_double(acc);
        if ((Fr::P[bit >> 6] >> (bit & 63)) & 1ull) acc = pt_add_affine(acc, p);
    }
    return acc.inf || F::is_zero(acc.zz);
}
AffineT<Fq2Field> g2_lazy(const G2 &p)
{
    AffineT<Fq2Field> r;
    r.x.c0 = p.x.c0.v; r.x.c1 = p.x.c1.v; r.y.c0 = p.y.c0.v; r.y.c1 = p.y.c1.v; r.inf = p.inf;
    return r;
}

// ark's deserialiser rejects a field element whose representation is not below the modulus; raw limbs are taken here, so the
// same is asked of them BEFORE any arithmetic reduces them silently (x and x + q would otherwise be one point with two encodings)
constexpr uint64_t FQ_MODULUS[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                    0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
bool fq_limbs_below_modulus(const uint64_t *w)
{
    for (int k = 5; k >= 0; k--) {
        if (w[k] < FQ_MODULUS[k]) return true;
        if (w[k] > FQ_MODULUS[k]) return false;
    }
    return false;
}
bool coordinates_canonical(const uint64_t *w, int coordinates)
{
    for (int k = 0; k < coordinates; k++)
        if (!fq_limbs_below_modulus(w + 6 * k)) return false;
    return true;
}

int bit_length(const uint64_t c[4])
{
    for (int k = 3; k >= 0; k--)
        if (c[k]) return 64 * k + 64 - __builtin_clzll(c[k]);
    return 0;
}
bool below_modulus(const uint64_t c[4])
{
    for (int k = 3; k >= 0; k--) {
        if (c[k] < Fr::P[k]) return true;
        if (c[k] > Fr::P[k]) return false;
    }
    return false;
}
}  // namespace

struct frw_groth16_vk {
    size_t num_instance;
    std::vector<G1Affine29> gamma_abc;
    G2 gamma_neg, delta_neg;
    Fp12 alpha_beta;                        // final_exponentiation(miller_loop(alpha_g1, beta_g2))
    FrobeniusConstants fc;
};

extern "C" int frw_groth16_vk_load(const uint64_t *vk, size_t num_instance, frw_groth16_vk **out)
{
    return frw_groth16_vk_load_opts(vk, num_instance, 0, out);
}
extern "C" int frw_groth16_vk_load_opts(const uint64_t *vk, size_t num_instance, int flags, frw_groth16_vk **out)
{
    if (!vk || !out || num_instance == 0) return FRW_E_INVALID_ARG;
    *out = nullptr;
    frw_groth16_vk *k = new (std::nothrow) frw_groth16_vk();
    if (!k) return FRW_E_OUT_OF_MEMORY;
    try {
        k->num_instance = num_instance;
        k->fc = frobenius_constants();
        if (!coordinates_canonical(vk, 2 + 4 + 4 + 4)) { delete k; return FRW_E_INVALID_ARG; }
        const G1Affine29 alpha = g1_lazy_from_ark(vk);
        const G2 beta = g2_from_ark(vk + 12), gamma = g2_from_ark(vk + 36), delta = g2_from_ark(vk + 60);
        bool ok = g1_on_curve(g1_strict(alpha)) && g2_on_curve(beta) && g2_on_curve(gamma) && g2_on_curve(delta) &&
                  in_subgroup(alpha) && in_subgroup(g2_lazy(beta)) && in_subgroup(g2_lazy(gamma)) && in_subgroup(g2_lazy(delta));
        k->gamma_abc.resize(num_instance);
        // gamma_abc_g1: canonical limbs, on the curve, in the subgroup of order r -- what ark's deserialiser checks of a key it
        // reads (a ladder per point: a few host threads for the 32,769 points of a sixteen-statement aggregate's key)
        if (ok) {
            const size_t hw = std::max(1u, std::thread::hardware_concurrency());
            const bool vouched = (flags & FRW_VK_POINTS_ARE_CHECKED) != 0;
            const size_t threads = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(num_instance / 256, hw), 64));
            std::atomic<bool> good{true};
            auto work = [&](size_t tid) {
                for (size_t i = tid; i < num_instance && good; i += threads) {
                    const uint64_t *w = vk + 84 + 12 * i;
                    if (!coordinates_canonical(w, 2)) { good = false; break; }
                    k->gamma_abc[i] = g1_lazy_from_ark(w);
                    if (!vouched && (!g1_on_curve(g1_strict(k->gamma_abc[i])) || !in_subgroup(k->gamma_abc[i]))) good = false;
                }
            };
            std::vector<std::thread> pool;
            for (size_t t = 1; t < threads; t++) pool.emplace_back(work, t);
            work(0);
            for (auto &t : pool) t.join();
            ok = good;
        }
        if (!ok) { delete k; return FRW_E_INVALID_ARG; }
        k->gamma_neg = g2_neg(gamma);
        k->delta_neg = g2_neg(delta);
        const G1 a = g1_strict(alpha);
        k->alpha_beta = final_exponentiation(miller_loop(&a, &beta, 1), k->fc);
    } catch (...) {
        delete k;
        return FRW_E_OUT_OF_MEMORY;
    }
    *out = k;
    return FRW_OK;
}

extern "C" void frw_groth16_vk_free(frw_groth16_vk *vk) { delete vk; }

namespace {
// sum_i x_i gamma_abc[i]: buckets of 8-bit windows, as many windows as the longest input needs (Falcon's are below 2^14)
// (an aggregate of 1,024 statements has 1.57 M inputs: the points are dealt to a few host threads, each with buckets of its own)
G1Xyzz prepare_inputs(const frw_groth16_vk &vk, const std::vector<uint64_t> &canon)
{
    const size_t n = vk.num_instance;
    int bits = 1;
    for (size_t i = 0; i < n; i++) bits = std::max(bits, bit_length(&canon[4 * i]));
    const int windows = (bits + 7) / 8;
    const size_t hw = std::max(1u, std::thread::hardware_concurrency());
    const size_t parts = std::max<size_t>(1, std::min<size_t>(std::min<size_t>(n / 16384, hw), 64));
    std::vector<G1Xyzz> partial(parts, g1_identity());
    auto work = [&](size_t part) {
        const size_t lo = n * part / parts, hi = n * (part + 1) / parts;
        G1Xyzz total = g1_identity();
        std::vector<G1Xyzz> bucket(255);
        for (int w = windows - 1; w >= 0; w--) {
            for (int s = 0; s < 8; s++) total = g1_double(total);
            for (auto &b : bucket) b = g1_identity();
            for (size_t i = lo; i < hi; i++) {
                const unsigned d = (unsigned)(canon[4 * i + (w >> 3)] >> (8 * (w & 7))) & 0xffu;
                if (d && !vk.gamma_abc[i].inf) bucket[d - 1] = g1_add_affine(bucket[d - 1], vk.gamma_abc[i]);
            }
            G1Xyzz run = g1_identity(), sum = g1_identity();
            for (int d = 254; d >= 0; d--) { run = g1_add(run, bucket[d]); sum = g1_add(sum, run); }
            total = g1_add(total, sum);
        }
        partial[part] = total;
    };
    {
        struct Joiner {
            std::vector<std::thread> th;
            ~Joiner() { for (auto &x : th) if (x.joinable()) x.join(); }
        } pool;
        size_t started = 1;
        try {
            for (size_t part = 1; part < parts; part++, started++) pool.th.emplace_back(work, part);
        } catch (const std::system_error &) {
            // no more threads to be had: this one does the rest
        }
        work(0);
        for (size_t part = started; part < parts; part++) work(part);
    }
    G1Xyzz total = partial[0];
    for (size_t part = 1; part < parts; part++) total = g1_add(total, partial[part]);
    return total;
}

int verify_one(const frw_groth16_vk &vk, const uint64_t *inputs, int encoding, const uint64_t *proof, int flags)
{
    const size_t n = vk.num_instance;
    std::vector<uint64_t> canon(4 * n);
    for (size_t i = 0; i < n; i++) {
        // the raw limbs, in either encoding, must be a representation below r: checked before the conversion, which reduces
        if (!below_modulus(inputs + 4 * i)) return -1;
        if (encoding == FRW_ENC_MONTGOMERY) Fr::from_montgomery(inputs + 4 * i).to_canonical(&canon[4 * i]);
        else std::memcpy(&canon[4 * i], inputs + 4 * i, 32);
    }
    if (canon[0] != 1 || canon[1] || canon[2] || canon[3]) return -1;         // the instance vector starts with the constant one
    if (!coordinates_canonical(proof, 8)) return -1;                          // A (x, y), B (x.c0, x.c1, y.c0, y.c1), C (x, y)
    const G1Affine29 a = g1_lazy_from_ark(proof), c = g1_lazy_from_ark(proof + 36);
    const G2 b = g2_from_ark(proof + 12);
    const G1 as = g1_strict(a), cs = g1_strict(c);
    if (!g1_on_curve(as) || !g2_on_curve(b) || !g1_on_curve(cs)) return -1;
    if (!(flags & FRW_VERIFY_POINTS_ARE_CHECKED) && (!in_subgroup(a) || !in_subgroup(g2_lazy(b)) || !in_subgroup(c))) return -1;
    const G1Affine29 acc = g1_to_affine(prepare_inputs(vk, canon));
    // e(A, B) e(acc, -gamma) e(C, -delta) == e(alpha, beta)
    const G1 ps[3] = {as, g1_strict(acc), cs};
    const G2 qs[3] = {b, vk.gamma_neg, vk.delta_neg};
    bool degenerate = false;
    const Fp12 f = miller_loop(ps, qs, 3, &degenerate);
    if (degenerate) return -1;              // only a point outside the subgroup gets here (FRW_VERIFY_POINTS_ARE_CHECKED and not true)
    return fp12_eq(final_exponentiation(f, vk.fc), vk.alpha_beta) ? 1 : 0;
}
}  // namespace

extern "C" int frw_groth16_verify(const frw_groth16_vk *vk, size_t batch, const uint64_t *instance, int encoding,
                                  const uint64_t *proofs, int flags, int32_t *accepted)
{
    if (!vk || (batch && (!instance || !proofs || !accepted))) return FRW_E_INVALID_ARG;
    if (encoding != FRW_ENC_MONTGOMERY && encoding != FRW_ENC_CANONICAL) return FRW_E_INVALID_ARG;
    try {
        const size_t hw = std::max(1u, std::thread::hardware_concurrency());
        const size_t threads = std::min<size_t>(std::min<size_t>(batch, hw), 32);
        std::atomic<size_t> next{0};
        std::atomic<bool> failed{false};
        auto work = [&]() {
            try {
                for (size_t i; (i = next.fetch_add(1)) < batch;)
                    accepted[i] = verify_one(*vk, instance + i * vk->num_instance * 4, encoding, proofs + i * 48, flags);
            } catch (...) {
                failed = true;
            }
        };
        if (threads <= 1) work();
        else {
            std::vector<std::thread> pool;
            try {
                for (size_t t = 0; t < threads; t++) pool.emplace_back(work);
            } catch (...) {                                              // no more threads to be had: the ones there are finish the batch
                if (pool.empty()) work();
            }
            for (auto &t : pool) t.join();
        }
        return failed ? FRW_E_OUT_OF_MEMORY : FRW_OK;
    } catch (...) {
        return FRW_E_OUT_OF_MEMORY;
    }
}

// diagnostics for the parity tests: the value the verifier's pairing gives for one pair, as the twelve coefficients of
// 1, w, ..., w^11 in Fq[w] / (w^12 - 2 w^6 + 2) (one polynomial basis, what a checker without towers uses), ark-ff's 6 x u64 each
extern "C" int frw_diag_pairing(const uint64_t *g1, const uint64_t *g2, uint64_t *out)
{
    if (!g1 || !g2 || !out) return FRW_E_INVALID_ARG;
    const G1 p = g1_strict(g1_lazy_from_ark(g1));
    const G2 q = g2_from_ark(g2);
    if (!g1_on_curve(p) || !g2_on_curve(q)) return FRW_E_INVALID_ARG;
    const FrobeniusConstants fc = frobenius_constants();
    const Fp12 e = final_exponentiation(miller_loop(&p, &q, 1), fc);
    // a (in Fp2) v^i w^j = (a0 - a1) w^k + a1 w^(k + 6), k = 2 i + j: u = w^6 - 1
    const Fp2 *coef[6] = {&e.c0.c0, &e.c1.c0, &e.c0.c1, &e.c1.c1, &e.c0.c2, &e.c1.c2};
    for (int k = 0; k < 6; k++) {
        fp_to_ark(fp_sub(coef[k]->c0, coef[k]->c1), out + 6 * k);
        fp_to_ark(coef[k]->c1, out + 6 * (k + 6));
    }
    return FRW_OK;
}
