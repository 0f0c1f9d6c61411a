// frw_r1cs.cpp -- R1CS matrix export: the step right after the hot path for any real prover
// (examples/pok_sig.rs:30-32 hands the circuit to Groth16, which calls cs.finalize() / to_matrices()).
// Structure only: runs the host mirror (host/frw_host.hpp) in setup mode -- no values, no GPU -- inlines every
// symbolic linear combination and writes A, B, C in a small CSR file a prover can ingest without arkworks.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <algorithm>
#include <array>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <stdexcept>
#include <system_error>
#include <exception>
#include <numeric>
#include <vector>

#include "../../include/frw.h"
#include "frw_arena.h"
#include "frw_device.h"
#include "host/frw_host.hpp"

namespace {
frw::host::ConstraintMatrices build_matrices(int circuit, int logn)
{
    using namespace frw::host;
    const size_t N = (size_t)1 << logn;
    auto cs = ConstraintSystem::new_ref();
    cs->set_setup_mode(true);
    Polynomial z{std::vector<uint16_t>(N, 0)};
    if (circuit == FRW_CIRCUIT_NTT) FalconNTTVerificationCircuit::build_circuit(z, z, z, logn).generate_constraints(cs);
    else FalconDualNTTVerificationCircuit::build_circuit(z, z, z, logn).generate_constraints(cs);
    return cs->to_matrices();
}
}  // namespace

// Device-resident matrices for frw_r1cs_check_dev
struct frw_r1cs {
    int device;
    int circuit = FRW_CIRCUIT_NTT, logn = 0;            // logn == 0: an aggregate statement
    frw::R1csDev dev;
    frw::QapDev qap;
    std::vector<void *> allocs;
    // a per-signature handle keeps its matrices on the host too (100 MB): frw_groth16_setup_r1cs(_opts) needs them again -- by rows for
    // the host-made keys, by columns for the device-made ones -- and arkworks' finalize() inlining takes seconds
    std::shared_ptr<const frw::host::ConstraintMatrices> host_matrices;
    mutable frw::HostArena arena;     // working memory of frw_qap_witness_map (host buffers in, host buffers out)
    // an aggregate (frw_r1cs_load_aggregate): the per-signature systems it is made of (owned), its statements and its runs
    frw_r1cs *base[2] = {nullptr, nullptr};             // Falcon-512, Falcon-1024
    std::vector<int32_t> statement_logn;
    std::vector<frw::R1csAggRun> runs;
    frw::R1csAgg agg{};
};

namespace {
using frw::host::Fr;
Fr pow_limbs(const Fr &b, const uint64_t e[4])
{
    Fr r = Fr::one();
    for (int i = 255; i >= 0; i--) {
        r = r * r;
        if ((e[i / 64] >> (i % 64)) & 1) r = r * b;
    }
    return r;
}
Fr inverse(const Fr &a)
{
    const uint64_t e[4] = {Fr::P[0] - 2, Fr::P[1], Fr::P[2], Fr::P[3]};
    return pow_limbs(a, e);
}
int domain_log(uint64_t num_coeffs)
{
    int lg = 0;
    while (((uint64_t)1 << lg) < num_coeffs) lg++;
    return lg;
}
// Tables of Radix2EvaluationDomain::new(num_constraints + num_instance) (ark-poly 0.3.0), Montgomery limbs as the
// device reads them.  group_gen = two_adic_root_of_unity^(2^(32 - log n)) (ark-ff 0.3.0 get_root_of_unity); the root is
// ark-bls12-381 0.3.0's TWO_ADIC_ROOT_OF_UNITY = 7^((p-1)/2^32) (tests/test_qap.py derives these limbs from the formula).
// 4 x 64-bit limbs -> nine 29-bit limbs
void split29(const uint64_t l[4], uint32_t out[9])
{
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, q = bit >> 6, sft = bit & 63;
        uint64_t limb = l[q] >> sft;
        if (sft > 35 && q + 1 < 4) limb |= l[q + 1] << (64 - sft);
        out[i] = (uint32_t)(limb & 0x1fffffffu);
    }
}
// x R' (R' = 2^261 = 2^5 R: the Montgomery limbs of 32 x) in nine 29-bit limbs
void limbs29(const Fr &v, uint32_t out[9])
{
    const Fr x = v * Fr::from(32);
    for (int i = 0; i < 9; i++) {
        const int bit = 29 * i, q = bit >> 6, sft = bit & 63;
        uint64_t limb = x.l[q] >> sft;
        if (sft > 35 && q + 1 < 4) limb |= x.l[q + 1] << (64 - sft);
        out[i] = (uint32_t)(limb & 0x1fffffffu);
    }
}
// x R' packed in 8 x 32 bits (x R' mod p < p)
void packed29(const Fr &v, uint32_t out[8])
{
    const Fr x = v * Fr::from(32);
    for (int k = 0; k < 4; k++) { out[2 * k] = (uint32_t)x.l[k]; out[2 * k + 1] = (uint32_t)(x.l[k] >> 32); }
}
// ---- the tables of the witness map's transforms, made on the device (round 5) --------------------------------------------------------
// Every table is first x base^(e(i)) over the whole domain -- thirteen tables of 32 n bytes for a five-pass domain: 56 GB at 2^27, which the
// host built one sequential product at a time until round 4 (and held twice).  The host now supplies, per base, two small power tables
// (base^k for k < 2^14 and base^(k 2^14)) and the constants; qap_table_kernel (frw_setup.hip) does the rest: 2 products per entry.
struct PowTabHost { std::vector<uint32_t> lo, hi; };
PowTabHost pow_tab_host(const Fr &base, int L)
{
    PowTabHost t;
    const size_t nlo = (size_t)frw::SETUP_POW_LO, nhi = L > frw::SETUP_POW_LO_BITS ? (size_t)1 << (L - frw::SETUP_POW_LO_BITS) : 1;
    t.lo.resize(nlo * 8);
    t.hi.resize(nhi * 8);
    Fr x = Fr::one();
    for (size_t k = 0; k < nlo; k++) { packed29(x, &t.lo[8 * k]); x = x * base; }
    const Fr step = x;                                           // base^(2^14)
    x = Fr::one();
    for (size_t k = 0; k < nhi; k++) { packed29(x, &t.hi[8 * k]); x = x * step; }
    return t;
}
frw::SetupConst setup_const(const Fr &v)
{
    frw::SetupConst c;
    limbs29(v, c.l);
    return c;
}
}  // namespace

extern "C" void frw_r1cs_free(frw_r1cs *r)
{
    if (!r) return;
    for (frw_r1cs *b : r->base) frw_r1cs_free(b);
    (void)hipSetDevice(r->device);
    for (void *p : r->allocs) (void)hipFree(p);
    r->arena.destroy();
    delete r;
}

namespace {
// the transform tables of the domain of (num_constraints + num_instance) coefficients, uploaded; r->qap.num_passes == 0 if the
// device has no pass schedule for that domain
// Tables of Radix2EvaluationDomain::new(num_constraints + num_instance) (ark-poly 0.3.0), Montgomery limbs as the
// device reads them.  group_gen = two_adic_root_of_unity^(2^(32 - log n)) (ark-ff 0.3.0 get_root_of_unity); the root is
// ark-bls12-381 0.3.0's TWO_ADIC_ROOT_OF_UNITY = 7^((p-1)/2^32) (tests/test_qap.py derives these limbs from the formula).
const uint64_t TWO_ADIC_ROOT_LIMBS[4] = {0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL, 0x5bf3adda19e9b27bULL};
void upload_qap_tables(frw_r1cs *r, uint64_t num_constraints, uint64_t num_instance)
{
    auto upload = [&](const void *src, size_t bytes) -> void * {
        void *d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) throw std::runtime_error("hipMalloc");
        r->allocs.push_back(d);
        if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy");
        return d;
    };
    r->qap = frw::QapDev{};
    const int L = domain_log(num_constraints + num_instance);
    if (L > 32) throw std::runtime_error("PolynomialDegreeTooLarge");
    r->qap.log_n = L;
    r->qap.num_passes = frw::qap_pass_schedule(L, r->qap.pass_t, r->qap.pass_sh);
    if (!r->qap.num_passes) return;                                 // no device witness map for such a domain (frw_qap.hip)
    const size_t n = (size_t)1 << L;
    // what the tables take: 2 (K - 1) twists + 5 scale tables of 32 n bytes -- asked of the device before anything is allocated
    {
        size_t free_b = 0, total_b = 0;
        const size_t need = (size_t)(2 * (r->qap.num_passes - 1) + 5) * n * 32;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess && need + (need >> 4) > free_b) throw std::bad_alloc();
    }
    Fr w = Fr::from_montgomery(TWO_ADIC_ROOT_LIMBS);
    for (int i = L; i < 32; i++) w = w * w;
    // psi: the primitive 2n-th root of unity whose square is w (get_root_of_unity(2 n))
    Fr psi = Fr::from_montgomery(TWO_ADIC_ROOT_LIMBS);
    for (int i = L + 1; i < 32; i++) psi = psi * psi;
    const Fr winv = inverse(w), g = Fr::from(7), ginv = inverse(g), ninv = inverse(Fr::from(n)), two5 = Fr::from(32);
    const Fr zinv = inverse(g.pow(n) - Fr::one());             // divide_by_vanishing_poly_on_coset
    const Fr sixteen_n = Fr::from(16) * ninv;
    // one table: its base's power tables go up, the kernel runs, the power tables go again
    auto make = [&](const Fr &base, int mode, int sh, int ts, const Fr *first) -> const uint32_t * {
        const PowTabHost pt = pow_tab_host(base, L);
        void *d_lo = nullptr, *d_hi = nullptr, *d_out = nullptr;
        hipError_t e = hipMalloc(&d_out, n * 32);
        if (e == hipSuccess) r->allocs.push_back(d_out);
        if (e == hipSuccess) e = hipMalloc(&d_lo, pt.lo.size() * 4);
        if (e == hipSuccess) e = hipMalloc(&d_hi, pt.hi.size() * 4);
        if (e == hipSuccess) e = hipMemcpy(d_lo, pt.lo.data(), pt.lo.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(d_hi, pt.hi.data(), pt.hi.size() * 4, hipMemcpyHostToDevice);
        if (e == hipSuccess) {
            const frw::SetupPowTab t{(const uint32_t *)d_lo, (const uint32_t *)d_hi};
            frw::SetupConst c{};
            if (first) c = setup_const(*first);
            e = frw::launch_qap_table(n, t, mode, sh, ts, L, first ? &c : nullptr, (uint32_t *)d_out, nullptr);
        }
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (d_lo) (void)hipFree(d_lo);
        if (d_hi) (void)hipFree(d_hi);
        if (e != hipSuccess) throw std::runtime_error("transform table");
        return (const uint32_t *)d_out;
    };
    const Fr in_a = ninv * two5, out_f = ninv * zinv, psi_out = -sixteen_n;
    r->qap.scale_in = make(g, 0, 0, 0, &ninv);                      // g^k / n
    r->qap.scale_in_a = make(g, 0, 0, 0, &in_a);                    // 2^5 g^k / n
    r->qap.scale_out = make(ginv, 0, 0, 0, &out_f);                 // g^-k / (n (g^n - 1))
    r->qap.scale_psi_in = make(psi, 0, 0, 0, &ninv);                // psi^k / n
    r->qap.scale_psi_out = make(inverse(psi), 0, 0, 0, &psi_out);   // -16 psi^-k / n
    limbs29(sixteen_n, r->qap.sixteen_over_n);
    // the twist tables: a pass on the index bits [sh, sh + T) goes with the factor
    // root^((i mod 2^sh) * bitrev_T((i >> sh) mod 2^T) * 2^(L - sh - T)) on index i (tools/dev/qap_fourstep_model.py).
    // table k goes with pass k + 1: the inverse transform's pass k leaves it for the next pass, the forward transform's pass
    // k + 1 applies it on its way out
    for (int k = 0; k + 1 < r->qap.num_passes; k++) {
        r->qap.twist_fwd[k] = make(w, 1, r->qap.pass_sh[k + 1], r->qap.pass_t[k + 1], nullptr);
        r->qap.twist_inv[k] = make(winv, 1, r->qap.pass_sh[k + 1], r->qap.pass_t[k + 1], nullptr);
    }
    // 64-th roots, nine 29-bit limbs, 12 words apart
    auto roots = [&](const Fr &root) {
        std::vector<uint32_t> v(32 * 12, 0u);
        const Fr step = root.pow(n >> 6);
        Fr x = Fr::one();
        for (int k = 0; k < 32; k++) { limbs29(x, &v[12 * k]); x = x * step; }
        return (const uint32_t *)upload(v.data(), v.size() * 4);
    };
    r->qap.roots_fwd = roots(w);
    r->qap.roots_inv = roots(winv);
}
}  // namespace

extern "C" int frw_r1cs_load(int device, int circuit, int logn, frw_r1cs **out)
{
    if (!out || (logn != 9 && logn != 10) || (circuit != FRW_CIRCUIT_NTT && circuit != FRW_CIRCUIT_DUAL_NTT)) return FRW_E_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FRW_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FRW_E_HIP;
    frw_r1cs *r = nullptr;
    try {
        auto matrices = std::make_shared<const frw::host::ConstraintMatrices>(build_matrices(circuit, logn));
        const frw::host::ConstraintMatrices &m = *matrices;
        r = new frw_r1cs;
        r->host_matrices = matrices;
        r->device = device;
        r->circuit = circuit;
        r->logn = logn;
        r->dev = frw::R1csDev{};
        if (r->arena.init() != hipSuccess) throw std::runtime_error("stream / event creation");
        r->dev.num_instance = (uint32_t)m.num_instance_variables;
        r->dev.num_witness = (uint32_t)m.num_witness_variables;
        r->dev.num_constraints = (uint32_t)m.num_constraints;
        auto upload = [&](const void *src, size_t bytes) -> void * {
            void *d = nullptr;
            if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) throw std::runtime_error("hipMalloc");
            r->allocs.push_back(d);
            if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy");
            return d;
        };
        auto put = [&](const std::vector<frw::host::ConstraintMatrices::Row> &rows, frw::R1csMatrixDev &dst) {
            std::vector<uint64_t> ptr{0};
            std::vector<uint32_t> col, val, cls, val29;
            const Fr one = Fr::one(), minus_one = -Fr::one(), two5 = Fr::from(32);
            for (const auto &row : rows) {
                for (const auto &t : row) {
                    if (t.second >> 30) throw std::runtime_error("too many variables");
                    col.push_back(t.second);
                    cls.push_back(t.second | (t.first == one ? 1u << 30 : t.first == minus_one ? 2u << 30 : 0u));
                    const Fr c29 = t.first * two5;                       // Montgomery limbs of 32 c = c R' mod p
                    for (int k = 0; k < 4; k++) { val.push_back((uint32_t)t.first.l[k]); val.push_back((uint32_t)(t.first.l[k] >> 32)); }   // Montgomery limbs
                    for (int k = 0; k < 4; k++) { val29.push_back((uint32_t)c29.l[k]); val29.push_back((uint32_t)(c29.l[k] >> 32)); }
                }
                ptr.push_back(col.size());
            }
            dst.row_ptr = (const uint64_t *)upload(ptr.data(), ptr.size() * 8);
            dst.col = (const uint32_t *)upload(col.data(), col.size() * 4);
            dst.val = (const uint32_t *)upload(val.data(), val.size() * 4);
            dst.col_class = (const uint32_t *)upload(cls.data(), cls.size() * 4);
            dst.val29 = (const uint32_t *)upload(val29.data(), val29.size() * 4);
        };
        put(m.a, r->dev.a);
        put(m.b, r->dev.b);
        put(m.c, r->dev.c);
        std::vector<uint32_t> order(m.num_constraints);
        std::iota(order.begin(), order.end(), 0u);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) {
            return m.a[x].size() + m.b[x].size() + m.c[x].size() > m.a[y].size() + m.b[y].size() + m.c[y].size();
        });
        r->dev.order = (const uint32_t *)upload(order.data(), order.size() * 4);
        // the short rows flattened (frw_device.h: flat_*): one header and one run of term words per row, coefficients by index
        {
            const std::vector<frw::host::ConstraintMatrices::Row> *ms[3] = {&m.a, &m.b, &m.c};
            const Fr one = Fr::one(), minus_one = -Fr::one(), two5 = Fr::from(32);
            std::vector<Fr> coefs{one, minus_one};
            std::vector<uint32_t> fhead, fterm;
            bool fits = std::getenv("FRW_R1CS_NO_FLAT") == nullptr;     // (diagnostics: tests run the CSR walk, the route of a circuit that does not fit)
            for (size_t i = 0; i < m.num_constraints && fits; i++) {
                const uint32_t row = order[i];               // (rows in their own order instead: stores coalesce, lengths diverge -- 1,727 us against 618)
                const size_t first_term = fterm.size();
                uint32_t counts = 0;
                for (uint32_t mi = 0; mi < 3 && fits; mi++) {
                    const auto &terms = (*ms[mi])[row];
                    if (terms.size() >= frw::R1CS_LONG_ROW) { counts |= 1u << (24 + mi); continue; }
                    counts |= (uint32_t)terms.size() << (8 * mi);
                    for (const auto &t : terms) {
                        size_t idx = 0;
                        while (idx < coefs.size() && !(coefs[idx] == t.first)) idx++;
                        if (idx == coefs.size()) coefs.push_back(t.first);
                        if (idx >= frw::R1CS_FLAT_COEFS || (t.second >> 24)) { fits = false; break; }
                        fterm.push_back(t.second | (uint32_t)idx << 24);
                    }
                }
                fhead.push_back(row);
                fhead.push_back(counts);
                fhead.push_back((uint32_t)(first_term / 4));
                fhead.push_back(0u);
                while (fterm.size() % 4) fterm.push_back(0u);            // (0: the constant one with coefficient +1 -- beyond the row's count, never added)
            }
            r->dev.flat_term = nullptr;
            r->dev.flat_num_coefs = 0;
            if (fits) {
                std::vector<uint32_t> fcoef;
                for (const Fr &c : coefs) {
                    const Fr c29 = c * two5;                             // Montgomery limbs of 32 c = c R' mod p, as val29
                    for (int k = 0; k < 4; k++) { fcoef.push_back((uint32_t)c29.l[k]); fcoef.push_back((uint32_t)(c29.l[k] >> 32)); }
                }
                r->dev.flat_head = (const uint32_t *)upload(fhead.data(), fhead.size() * 4);
                r->dev.flat_coef = (const uint32_t *)upload(fcoef.data(), fcoef.size() * 4);
                r->dev.flat_num_coefs = (uint32_t)coefs.size();
                r->dev.flat_term = (const uint32_t *)upload(fterm.data(), fterm.size() * 4);
            }
        }
        // long rows: chunked, coefficient planes (see frw_device.h)
        std::vector<frw::R1csLongRow> lrows;
        std::vector<uint32_t> lcol, lcoef;
        std::vector<uint8_t> lmask(m.num_constraints, 0);
        std::vector<uint32_t> lslot(3 * m.num_constraints, 0u);
        const std::vector<frw::host::ConstraintMatrices::Row> *mats[3] = {&m.a, &m.b, &m.c};
        for (uint32_t mi = 0; mi < 3; mi++)
            for (size_t row = 0; row < m.num_constraints; row++) {
                const auto &terms = (*mats[mi])[row];
                if (terms.size() < frw::R1CS_LONG_ROW) continue;
                const uint32_t chunks = (uint32_t)((terms.size() + 63) / 64), first = (uint32_t)(lcol.size() / 64);
                lslot[(size_t)mi * m.num_constraints + row] = (uint32_t)lrows.size();
                lrows.push_back({mi, (uint32_t)row, first, chunks});
                lmask[row] |= (uint8_t)(1u << mi);
                lcol.resize((size_t)(first + chunks) * 64, 0u);
                lcoef.resize((size_t)(first + chunks) * 9 * 64, 0u);
                for (size_t t = 0; t < terms.size(); t++) {
                    uint32_t l[9];
                    limbs29(terms[t].first, l);
                    const size_t ch = first + t / 64, lane = t % 64;
                    lcol[ch * 64 + lane] = terms[t].second;
                    for (int k = 0; k < 9; k++) lcoef[(ch * 9 + k) * 64 + lane] = l[k];
                }
            }
        // the variables the long rows read, and each term's index into that list
        std::vector<uint32_t> lvars(lcol);
        std::sort(lvars.begin(), lvars.end());
        lvars.erase(std::unique(lvars.begin(), lvars.end()), lvars.end());
        std::vector<uint32_t> lcidx(lcol.size());
        for (size_t t = 0; t < lcol.size(); t++)
            lcidx[t] = (uint32_t)(std::lower_bound(lvars.begin(), lvars.end(), lcol[t]) - lvars.begin());
        r->dev.num_long_vars = (uint32_t)lvars.size();
        r->dev.long_vars = (const uint32_t *)upload(lvars.data(), lvars.size() * 4);
        r->dev.long_cidx = (const uint32_t *)upload(lcidx.data(), lcidx.size() * 4);
        {
            const Fr rp = Fr::from(32);                        // limbs: R' mod p as an integer
            const Fr k = Fr::from_canonical(rp.l);             // value R' mod p; limbs: R' R mod p
            split29(k.l, r->dev.k_rrp);
        }
        r->dev.num_long = (uint32_t)lrows.size();
        r->dev.long_rows = (const frw::R1csLongRow *)upload(lrows.data(), lrows.size() * sizeof(frw::R1csLongRow));
        r->dev.long_col = (const uint32_t *)upload(lcol.data(), lcol.size() * 4);
        r->dev.long_coef = (const uint32_t *)upload(lcoef.data(), lcoef.size() * 4);
        r->dev.long_mask = (const uint8_t *)upload(lmask.data(), lmask.size());
        r->dev.long_slot = (const uint32_t *)upload(lslot.data(), lslot.size() * 4);
        upload_qap_tables(r, m.num_constraints, m.num_instance_variables);
        *out = r;
        return FRW_OK;
    } catch (const std::exception &) {
        frw_r1cs_free(r);
        return FRW_E_OUT_OF_MEMORY;
    }
}

// ---- an aggregate statement: `count` Falcon verifications on ONE constraint system ------------------------------------------------
// (host/frw_host.hpp FalconAggregateVerificationCircuit: generate_constraints of FalconNTTVerificationCircuit once per statement,
// in order.)  Nothing of the aggregate's size is built on the host or stored on the device but the transform tables of its
// domain: the matrices are the per-signature systems' blocks (frw_device.h R1csAgg).
extern "C" int frw_r1cs_load_aggregate(int device, size_t count, const int32_t *logn, frw_r1cs **out)
{
    if (!out || !logn || count == 0 || count > 65535) return FRW_E_INVALID_ARG;
    for (size_t i = 0; i < count; i++)
        if (logn[i] != 9 && logn[i] != 10) return FRW_E_INVALID_ARG;
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FRW_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FRW_E_HIP;
    frw_r1cs *r = new (std::nothrow) frw_r1cs;
    if (!r) return FRW_E_OUT_OF_MEMORY;
    r->device = device;
    r->dev = frw::R1csDev{};
    r->qap = frw::QapDev{};
    int rc = FRW_OK;
    try {
        if (r->arena.init() != hipSuccess) throw std::runtime_error("stream / event creation");
        r->statement_logn.assign(logn, logn + count);
        for (int g = 0; g < 2 && rc == FRW_OK; g++)
            if (std::find(r->statement_logn.begin(), r->statement_logn.end(), 9 + g) != r->statement_logn.end())
                rc = frw_r1cs_load(device, FRW_CIRCUIT_NTT, 9 + g, &r->base[g]);
        if (rc == FRW_OK) {
            uint64_t wit = 0, pub = 0, rows = 0;
            for (size_t i = 0; i < count;) {
                size_t j = i;
                while (j < count && logn[j] == logn[i]) j++;
                const frw::R1csDev &b = r->base[logn[i] - 9]->dev;
                r->runs.push_back(frw::R1csAggRun{&b, (uint32_t)i, (uint32_t)(j - i), wit, pub, rows});
                wit += (uint64_t)(j - i) * b.num_witness;
                pub += (uint64_t)(j - i) * (b.num_instance - 1);
                rows += (uint64_t)(j - i) * b.num_constraints;
                i = j;
            }
            if (wit + pub + 1 >= ((uint64_t)1 << 30) || rows >= ((uint64_t)1 << 31)) throw std::invalid_argument("aggregate too large");
            r->agg.num_statements = (uint32_t)count;
            r->agg.num_runs = (uint32_t)r->runs.size();
            r->agg.runs = r->runs.data();
            // per parameter set: where each of its statements sits (R1csView::offs), for the evaluation kernels' one launch per set
            for (int g = 0; g < 2; g++) {
                std::vector<uint64_t> offs;
                for (const frw::R1csAggRun &run : r->runs) {
                    if (run.base != (r->base[g] ? &r->base[g]->dev : nullptr)) continue;
                    for (uint32_t k = 0; k < run.count; k++) {
                        offs.push_back(run.wit_off + (uint64_t)k * run.base->num_witness);
                        offs.push_back(run.pub_off + (uint64_t)k * (run.base->num_instance - 1));
                        offs.push_back(run.row_off + (uint64_t)k * run.base->num_constraints);
                    }
                }
                r->agg.set[g] = frw::R1csAggSet{r->base[g] ? &r->base[g]->dev : nullptr, (uint32_t)(offs.size() / 3), nullptr};
                if (offs.empty()) continue;
                void *d = nullptr;
                if (hipMalloc(&d, offs.size() * 8) != hipSuccess) throw std::bad_alloc();
                r->allocs.push_back(d);
                if (hipMemcpy(d, offs.data(), offs.size() * 8, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy");
                r->agg.set[g].offs = (const uint64_t *)d;
            }
            r->dev.agg = &r->agg;
            r->dev.num_instance = (uint32_t)(pub + 1);
            r->dev.num_witness = (uint32_t)wit;
            r->dev.num_constraints = (uint32_t)rows;
            upload_qap_tables(r, rows, pub + 1);
        }
    } catch (const std::invalid_argument &) {
        rc = FRW_E_INVALID_ARG;                                      // a statement beyond the 2^30 variables / 2^31 rows the kernels index
    } catch (const std::exception &) {
        rc = FRW_E_OUT_OF_MEMORY;
    }
    if (rc != FRW_OK) {
        frw_r1cs_free(r);
        return rc;
    }
    *out = r;
    return FRW_OK;
}

extern "C" int frw_r1cs_info(const frw_r1cs *r, frw_r1cs_info_t *out)
{
    if (!r || !out) return FRW_E_INVALID_ARG;
    out->num_statements = r->dev.agg ? r->agg.num_statements : 1;
    out->num_instance = r->dev.num_instance;
    out->num_witness = r->dev.num_witness;
    out->num_constraints = r->dev.num_constraints;
    out->log_domain_size = r->qap.log_n;
    out->witness_map_on_device = r->qap.num_passes >= 2;
    out->count_logn9 = out->count_logn10 = 0;
    if (r->dev.agg) {
        for (int32_t l : r->statement_logn) (l == 9 ? out->count_logn9 : out->count_logn10)++;
    } else {
        (r->logn == 9 ? out->count_logn9 : out->count_logn10) = 1;
    }
    return FRW_OK;
}

// instance_assignment / witness_assignment of the aggregate's constraint system from the batches the witness entry points wrote:
// statement i takes the next unused signature of its parameter set's batch (the order FalconAggregateVerificationCircuit feeds
// them in).  Device-to-device copies on `stream`, one pair per run.
extern "C" int frw_aggregate_assign_dev(const frw_r1cs *r, const uint64_t *d_witness_512, const uint64_t *d_instance_512,
                                        const uint64_t *d_witness_1024, const uint64_t *d_instance_1024, uint64_t *d_witness,
                                        uint64_t *d_instance, void *stream)
{
    if (!r || !r->dev.agg || !d_witness || !d_instance) return FRW_E_INVALID_ARG;
    const uint64_t *wsrc[2] = {d_witness_512, d_witness_1024}, *isrc[2] = {d_instance_512, d_instance_1024};
    for (int g = 0; g < 2; g++)
        if (r->base[g] && (!wsrc[g] || !isrc[g])) return FRW_E_INVALID_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipSetDevice(r->device);
    // the aggregate's ONE constant: the leading element of any statement's own instance vector (the witness entry points write it in the
    // batches' encoding) -- a device-to-device copy, so the call stays on the stream and may be captured
    const uint64_t *first_one = r->base[0] ? isrc[0] : isrc[1];
    if (e == hipSuccess) e = hipMemcpyAsync(d_instance, first_one, 32, hipMemcpyDeviceToDevice, st);
    size_t used[2] = {0, 0};
    for (const frw::R1csAggRun &run : r->runs) {
        if (e != hipSuccess) break;
        const int g = r->statement_logn[run.first] - 9;
        const size_t W = run.base->num_witness, I = run.base->num_instance, cnt = run.count;
        e = hipMemcpyAsync(d_witness + run.wit_off * 4, wsrc[g] + used[g] * W * 4, cnt * W * 32, hipMemcpyDeviceToDevice, st);
        if (e == hipSuccess)       // the public inputs of every statement of the run, without its own leading one
            e = hipMemcpy2DAsync(d_instance + (1 + run.pub_off) * 4, (I - 1) * 32, isrc[g] + (used[g] * I + 1) * 4, I * 32, (I - 1) * 32, cnt,
                                 hipMemcpyDeviceToDevice, st);
        used[g] += cnt;
    }
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_aggregate_assign_dev");
}

namespace {
int r1cs_eval(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
              uint32_t *d_num_unsatisfied, uint64_t *d_abc, void *d_scratch, void *stream)
{
    hipError_t e = hipSetDevice(r->device);
    if (e != hipSuccess) return frw::record_hip_error(e, "hipSetDevice");
    for (size_t lo = 0; lo < batch; lo += 32768) {
        const size_t cnt = std::min<size_t>(32768, batch - lo);
        e = frw::launch_r1cs_check(r->dev, cnt, d_witness + lo * (size_t)r->dev.num_witness * 4,
                                   d_instance + lo * (size_t)r->dev.num_instance * 4, d_num_unsatisfied + lo,
                                   d_abc ? d_abc + lo * (size_t)3 * r->dev.num_constraints * 4 : nullptr,
                                   (hipStream_t)stream, d_scratch);     // the chunks run one after the other: one scratch
        if (e != hipSuccess) return frw::record_hip_error(e, "frw_r1cs_eval_dev");
    }
    return FRW_OK;
}
}  // namespace

extern "C" int frw_r1cs_eval_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                                 uint32_t *d_num_unsatisfied, uint64_t *d_abc, void *stream)
{
    if (!r || (batch && (!d_witness || !d_instance || !d_num_unsatisfied))) return FRW_E_INVALID_ARG;
    return r1cs_eval(r, batch, d_witness, d_instance, d_num_unsatisfied, d_abc, nullptr, stream);
}

extern "C" size_t frw_r1cs_eval_scratch_bytes(const frw_r1cs *r, size_t batch, int with_products)
{
    if (!r) return 0;
    return frw::r1cs_check_scratch_bytes(r->dev, std::min<size_t>(batch, 32768), with_products != 0);
}

extern "C" int frw_r1cs_eval_scratch_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                                         uint32_t *d_num_unsatisfied, uint64_t *d_abc, void *d_scratch, size_t scratch_bytes,
                                         void *stream)
{
    if (!r || (batch && (!d_witness || !d_instance || !d_num_unsatisfied))) return FRW_E_INVALID_ARG;
    const size_t need = frw_r1cs_eval_scratch_bytes(r, batch, d_abc != nullptr);
    if (batch && need && (!d_scratch || scratch_bytes < need || ((uintptr_t)d_scratch & 15))) return FRW_E_INVALID_ARG;
    return r1cs_eval(r, batch, d_witness, d_instance, d_num_unsatisfied, d_abc, need ? d_scratch : nullptr, stream);
}

extern "C" int frw_qap_info(const frw_r1cs *r, frw_qap_info_t *out)
{
    if (!r || !out) return FRW_E_INVALID_ARG;
    out->log_domain_size = r->qap.log_n;
    out->domain_size = (uint64_t)1 << r->qap.log_n;
    out->num_constraints = r->dev.num_constraints;
    out->num_instance = r->dev.num_instance;
    out->workspace_bytes_per_signature = frw::qap_workspace_bytes_per_signature(r->dev, r->qap);
    return FRW_OK;
}

extern "C" int frw_qap_witness_map_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                                       uint64_t *d_h, uint32_t *d_num_unsatisfied, void *d_workspace, size_t workspace_bytes,
                                       void *stream)
{
    if (!r || (batch && (!d_witness || !d_instance || !d_h || !d_workspace))) return FRW_E_INVALID_ARG;
    if (r->qap.num_passes < 2) return FRW_E_INVALID_ARG;                       // no pass schedule for this domain (frw_device.h qap_pass_schedule)
    if (batch && workspace_bytes < frw::qap_workspace_bytes_per_signature(r->dev, r->qap)) return FRW_E_INVALID_ARG;
    hipError_t e = hipSetDevice(r->device);
    if (e == hipSuccess)
        e = frw::launch_qap_witness_map(r->dev, r->qap, batch, d_witness, d_instance, d_h, d_num_unsatisfied, d_workspace,
                                        workspace_bytes, (hipStream_t)stream);
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_qap_witness_map_dev");
}

extern "C" int frw_qap_quotient_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                                    uint64_t *d_h, uint32_t *d_num_unsatisfied, void *d_workspace, size_t workspace_bytes,
                                    void *stream)
{
    if (!r || (batch && (!d_witness || !d_instance || !d_h || !d_workspace))) return FRW_E_INVALID_ARG;
    if (r->qap.num_passes < 2) return FRW_E_INVALID_ARG;
    if (batch && workspace_bytes < frw::qap_workspace_bytes_per_signature(r->dev, r->qap)) return FRW_E_INVALID_ARG;
    hipError_t e = hipSetDevice(r->device);
    if (e == hipSuccess)
        e = frw::launch_qap_quotient(r->dev, r->qap, batch, d_witness, d_instance, d_h, d_num_unsatisfied, d_workspace,
                                     workspace_bytes, (hipStream_t)stream);
    return e == hipSuccess ? FRW_OK : frw::record_hip_error(e, "frw_qap_quotient_dev");
}

// Host buffers in, host buffers out: the same map for a caller that holds `Vec<Fr>`s (arkworks' witness_assignment /
// instance_assignment bytes) -- copies in, frw_qap_witness_map_dev in chunks of 64 signatures, copies out.
extern "C" int frw_qap_witness_map(const frw_r1cs *r, size_t batch, const uint64_t *witness, const uint64_t *instance,
                                   uint64_t *h, uint32_t *num_unsatisfied)
{
    if (!r || (batch && (!witness || !instance || !h))) return FRW_E_INVALID_ARG;
    if (r->qap.num_passes < 2) return FRW_E_INVALID_ARG;
    if (batch == 0) return FRW_OK;
    frw::HostArena &A = r->arena;
    std::lock_guard<std::mutex> lock(A.mu);
    frw::DrainOnExit drain(A);
    hipError_t e = hipSetDevice(r->device);
    if (e != hipSuccess) return frw::record_hip_error(e, "hipSetDevice");
    const size_t n = (size_t)1 << r->qap.log_n, W = r->dev.num_witness, I = r->dev.num_instance;
    const size_t chunk = std::min<size_t>(batch, 64), per = frw::qap_workspace_bytes_per_signature(r->dev, r->qap);
    // one arena slot, carved (grow-only: a second call of the same size allocates nothing)
    frw::Carve size(nullptr);
    size.take(chunk * W * 32); size.take(chunk * I * 32); size.take(chunk * n * 32); size.take(chunk * per);
    size.take(chunk * sizeof(uint32_t));
    if ((e = A.reserve_device(0, size.off)) != hipSuccess) return frw::record_hip_error(e, "frw_qap_witness_map: device memory");
    frw::Carve c(A.d_slot[0]);
    uint64_t *d_wit = c.take<uint64_t>(chunk * W * 32), *d_inst = c.take<uint64_t>(chunk * I * 32);
    uint64_t *d_h = c.take<uint64_t>(chunk * n * 32);
    void *d_ws = c.take(chunk * per);
    uint32_t *d_bad = c.take<uint32_t>(chunk * sizeof(uint32_t));
    hipStream_t st = A.compute;
    for (size_t lo = 0; lo < batch; lo += chunk) {
        const size_t cnt = std::min(chunk, batch - lo);
        if ((e = hipMemcpyAsync(d_wit, witness + lo * W * 4, cnt * W * 32, hipMemcpyHostToDevice, st)) == hipSuccess)
            e = hipMemcpyAsync(d_inst, instance + lo * I * 4, cnt * I * 32, hipMemcpyHostToDevice, st);
        if (e == hipSuccess)
            e = frw::launch_qap_witness_map(r->dev, r->qap, cnt, d_wit, d_inst, d_h, d_bad, d_ws, chunk * per, st);
        if (e == hipSuccess) e = hipMemcpyAsync(h + lo * n * 4, d_h, cnt * n * 32, hipMemcpyDeviceToHost, st);
        if (e == hipSuccess && num_unsatisfied)
            e = hipMemcpyAsync(num_unsatisfied + lo, d_bad, cnt * sizeof(uint32_t), hipMemcpyDeviceToHost, st);
        if (e == hipSuccess) e = hipStreamSynchronize(st);
        if (e != hipSuccess) return frw::record_hip_error(e, "frw_qap_witness_map");
    }
    return FRW_OK;
}

extern "C" int frw_r1cs_diag_host_allocations(const frw_r1cs *r, uint64_t *count)
{
    if (!r || !count) return FRW_E_INVALID_ARG;
    std::lock_guard<std::mutex> lock(r->arena.mu);
    *count = r->arena.allocations;
    return FRW_OK;
}

extern "C" int frw_r1cs_check_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                                  uint32_t *d_num_unsatisfied, void *stream)
{
    return frw_r1cs_eval_dev(r, batch, d_witness, d_instance, d_num_unsatisfied, nullptr, stream);
}

extern "C" int frw_r1cs_export(int circuit, int logn, const char *path, uint64_t *counts /* 6 x u64, may be NULL */)
{
    using namespace frw::host;
    if ((logn != 9 && logn != 10) || !path || (circuit != FRW_CIRCUIT_NTT && circuit != FRW_CIRCUIT_DUAL_NTT)) return FRW_E_INVALID_ARG;
    try {
        ConstraintMatrices m = build_matrices(circuit, logn);
        if (counts) {
            counts[0] = m.num_instance_variables; counts[1] = m.num_witness_variables; counts[2] = m.num_constraints;
            counts[3] = m.non_zero(m.a); counts[4] = m.non_zero(m.b); counts[5] = m.non_zero(m.c);
        }
        return m.write(path) ? FRW_OK : FRW_E_INVALID_ARG;
    } catch (const std::exception &) {
        return FRW_E_INVALID_ARG;
    }
}

// ---- Groth16 setup: ark-groth16 0.3.0 generator.rs generate_parameters, with the toxic waste GIVEN -------------------------------
// (circuit_specific_setup draws alpha, beta, gamma, delta and the evaluation point t from an rng -- examples/pok_sig.rs:30-31 --
// and random generators of G1 and G2; here the caller supplies the five field elements and the published generators are used.)
// Host: the circuit's matrices, the Lagrange coefficients at t (instance_map_with_evaluation: u_i(t), v_i(t), w_i(t), zt), the
// scalars of every query.  Device: the queries themselves, fixed-base multiples (frw_g1_fixed_base / frw_g2_fixed_base), and the
// loaded proving key.  vk_out: alpha_g1 (12 u64) | beta_g2 (24) | gamma_g2 (24) | delta_g2 (24) | gamma_abc_g1 [I][12].
namespace {
// one block of a block-diagonal system: the matrices of a per-signature circuit and where its rows, public inputs and witness
// variables sit in the whole (a single circuit is one block at zero)
struct SetupBlock { const frw::host::ConstraintMatrices *m; size_t row_off, pub_off, wit_off; };

int groth16_setup_blocks(int device, const std::vector<SetupBlock> &blocks, size_t ni, size_t nw, size_t nc, const uint64_t *toxic,
                         frw_groth16_pk **pk_out, uint64_t *vk_out)
{
    using namespace frw::host;
    const size_t nv = ni + nw;
    const int L = domain_log(nc + ni);
    if (L > 30) return FRW_E_INVALID_ARG;
    const size_t n = (size_t)1 << L;
    const Fr alpha = Fr::from_canonical(toxic), beta = Fr::from_canonical(toxic + 4), gamma = Fr::from_canonical(toxic + 8),
             delta = Fr::from_canonical(toxic + 12), t = Fr::from_canonical(toxic + 16);
    const uint64_t root_limbs[4] = {0xb9b58d8c5f0e466aULL, 0x5b1b4c801819d7ecULL, 0x0af53ae352a31e64ULL, 0x5bf3adda19e9b27bULL};
    Fr w = Fr::from_montgomery(root_limbs);
    for (int i = L; i < 32; i++) w = w * w;
    const Fr zt = t.pow(n) - Fr::one();
    if (zt.is_zero() || gamma.is_zero() || delta.is_zero()) return FRW_E_INVALID_ARG;       // t must lie outside the domain
    // L_i(t) = zt w^i / (n (t - w^i)); the n inversions by Montgomery's trick
    std::vector<Fr> lag(n), den(n), pre(n);
    {
        Fr wi = Fr::one(), acc = Fr::one();
        for (size_t i = 0; i < n; i++) { lag[i] = wi; den[i] = t - wi; pre[i] = acc; acc = acc * den[i]; wi = wi * w; }
        Fr inv = inverse(acc);
        const Fr c = zt * inverse(Fr::from(n));
        for (size_t i = n; i-- > 0;) { lag[i] = c * lag[i] * (inv * pre[i]); inv = inv * den[i]; }
    }
    std::vector<Fr>().swap(den);
    std::vector<Fr>().swap(pre);
    std::vector<Fr> u(nv, Fr::zero()), v(nv, Fr::zero()), ww(nv, Fr::zero());
    for (size_t i = 0; i < ni; i++) u[i] = lag[nc + i];                                    // r1cs_to_qap.rs: the input rows
    // the blocks touch disjoint variables but for column 0 (the constant one): a few host threads, each with its own sums there
    const unsigned hw = std::max(1u, std::min<unsigned>(std::thread::hardware_concurrency(), 32u));
    const size_t nthreads = std::min<size_t>(hw, blocks.size());
    std::vector<std::array<Fr, 3>> col0(nthreads, {Fr::zero(), Fr::zero(), Fr::zero()});
    auto work = [&](size_t tid) {
        for (size_t b = tid; b < blocks.size(); b += nthreads) {
            const SetupBlock &blk = blocks[b];
            const size_t mi = blk.m->num_instance_variables;
            std::vector<Fr> *dst[3] = {&u, &v, &ww};
            const std::vector<ConstraintMatrices::Row> *mats[3] = {&blk.m->a, &blk.m->b, &blk.m->c};
            for (int k = 0; k < 3; k++)
                for (size_t r = 0; r < blk.m->num_constraints; r++) {
                    const Fr &lr = lag[blk.row_off + r];
                    for (const auto &e : (*mats[k])[r]) {
                        const Fr x = lr * e.first;
                        if (e.second == 0) col0[tid][k] = col0[tid][k] + x;
                        else {
                            const size_t col = e.second < mi ? blk.pub_off + e.second : ni + blk.wit_off + (e.second - mi);
                            (*dst[k])[col] = (*dst[k])[col] + x;
                        }
                    }
                }
        }
    };
    {
        // (a thread that cannot be started must not take the process down with the ones that were: they are joined whatever happens)
        struct Joiner {
            std::vector<std::thread> th;
            ~Joiner() { for (auto &x : th) if (x.joinable()) x.join(); }
        } pool;
        size_t started = 1;
        try {
            for (size_t tid = 1; tid < nthreads; tid++, started++) pool.th.emplace_back(work, tid);
        } catch (const std::system_error &) {
            // no more threads to be had: this one does the rest
        }
        work(0);
        for (size_t tid = started; tid < nthreads; tid++) work(tid);
    }
    for (size_t tid = 0; tid < nthreads; tid++) { u[0] = u[0] + col0[tid][0]; v[0] = v[0] + col0[tid][1]; ww[0] = ww[0] + col0[tid][2]; }
    std::vector<Fr>().swap(lag);
    const Fr dinv = inverse(delta), ginv = inverse(gamma);
    auto canon = [](const std::vector<Fr> &x) { std::vector<uint64_t> o(4 * x.size()); for (size_t i = 0; i < x.size(); i++) x[i].to_canonical(&o[4 * i]); return o; };
    std::vector<Fr> lq(nw), hq(n - 1), abc(ni), fixed = {alpha, beta, delta, gamma};
    for (size_t i = 0; i < nv; i++) {
        const Fr x = beta * u[i] + alpha * v[i] + ww[i];
        if (i < ni) abc[i] = x * ginv; else lq[i - ni] = x * dinv;
    }
    { Fr x = zt * dinv; for (size_t i = 0; i + 1 < n; i++) { hq[i] = x; x = x * t; } }
    std::vector<uint64_t> a_q(12 * nv), b1_q(12 * nv), b2_q(24 * nv), h_q(12 * (n - 1)), l_q(12 * nw), f1(12 * 4), f2(24 * 4), abc_q(12 * ni);
    int rc = frw_g1_fixed_base(device, nv, canon(u).data(), a_q.data());
    std::vector<Fr>().swap(u);
    const std::vector<uint64_t> vc = canon(v);
    std::vector<Fr>().swap(v);
    if (rc == FRW_OK) rc = frw_g1_fixed_base(device, nv, vc.data(), b1_q.data());
    if (rc == FRW_OK) rc = frw_g2_fixed_base(device, nv, vc.data(), b2_q.data());
    if (rc == FRW_OK) rc = frw_g1_fixed_base(device, n - 1, canon(hq).data(), h_q.data());
    std::vector<Fr>().swap(hq);
    if (rc == FRW_OK) rc = frw_g1_fixed_base(device, nw, canon(lq).data(), l_q.data());
    if (rc == FRW_OK) rc = frw_g1_fixed_base(device, 4, canon(fixed).data(), f1.data());
    if (rc == FRW_OK) rc = frw_g2_fixed_base(device, 4, canon(fixed).data(), f2.data());
    if (rc == FRW_OK && vk_out) rc = frw_g1_fixed_base(device, ni, canon(abc).data(), abc_q.data());
    if (rc != FRW_OK) return rc;
    frw_groth16_pk_desc_t d{};
    d.num_instance = ni; d.num_witness = nw; d.domain_size = n;
    d.alpha_g1 = &f1[0]; d.beta_g1 = &f1[12]; d.delta_g1 = &f1[24];
    d.beta_g2 = &f2[24]; d.delta_g2 = &f2[48];
    d.a_query = a_q.data(); d.b_g1_query = b1_q.data(); d.b_g2_query = b2_q.data(); d.h_query = h_q.data(); d.l_query = l_q.data();
    rc = frw_groth16_pk_load(device, &d, pk_out);
    if (rc == FRW_OK && vk_out) {
        std::memcpy(vk_out, &f1[0], 96);                       // alpha_g1
        std::memcpy(vk_out + 12, &f2[24], 192);                // beta_g2
        std::memcpy(vk_out + 36, &f2[72], 192);                // gamma_g2
        std::memcpy(vk_out + 60, &f2[48], 192);                // delta_g2
        std::memcpy(vk_out + 84, abc_q.data(), ni * 96);       // gamma_abc_g1
    }
    return rc;
}

// ---- the same on the device, into a key of bare handles (round 5) -----------------------------------------------------------------------
// Nothing of the statement's size is made on the host: L_i(t) for the whole domain, the transposed sparse products, the queries' scalars
// and the query points themselves (written straight into their table rows) are kernels of frw_setup.hip / frw_msm.hip.  The host supplies
// the per-signature matrices by columns (once per parameter set), two power tables each for w and t, and a dozen constants.
struct DeviceRun { const frw::host::ConstraintMatrices *m; uint32_t first, count; uint64_t wit_off, pub_off, row_off; };
struct DeviceBuffers {
    std::vector<void *> all;
    ~DeviceBuffers() { for (void *p : all) if (p) (void)hipFree(p); }
    void *get(size_t bytes)
    {
        void *d = nullptr;
        if (hipMalloc(&d, bytes ? bytes : 16) != hipSuccess) throw std::bad_alloc();
        all.push_back(d);
        return d;
    }
    void *put(const void *src, size_t bytes)
    {
        void *d = get(bytes);
        if (bytes && hipMemcpy(d, src, bytes, hipMemcpyHostToDevice) != hipSuccess) throw std::runtime_error("hipMemcpy");
        return d;
    }
    void drop(void *p)
    {
        for (void *&q : all)
            if (q == p) { (void)hipFree(q); q = nullptr; }
    }
};
void hip_ok(hipError_t e, const char *what)
{
    if (e != hipSuccess) { frw::record_hip_error(e, what); throw std::runtime_error(what); }
}

int groth16_setup_device(int device, const std::vector<DeviceRun> &runs, uint32_t statements, size_t ni, size_t nw, size_t nc, const uint64_t *toxic,
                         bool tables, uint32_t rank, uint32_t world, frw_groth16_pk **pk_out, uint64_t *vk_out)
{
    using namespace frw::host;
    const size_t nv = ni + nw;
    const int L = domain_log(nc + ni);
    if (L > 30 || L < frw::SETUP_POW_LO_BITS || rank >= world) return FRW_E_INVALID_ARG;
    const size_t n = (size_t)1 << L;
    const Fr alpha = Fr::from_canonical(toxic), beta = Fr::from_canonical(toxic + 4), gamma = Fr::from_canonical(toxic + 8),
             delta = Fr::from_canonical(toxic + 12), t = Fr::from_canonical(toxic + 16);
    Fr w = Fr::from_montgomery(TWO_ADIC_ROOT_LIMBS);
    for (int i = L; i < 32; i++) w = w * w;
    const Fr zt = t.pow(n) - Fr::one();
    if (zt.is_zero() || gamma.is_zero() || delta.is_zero()) return FRW_E_INVALID_ARG;       // t must lie outside the domain
    const Fr dinv = inverse(delta), ginv = inverse(gamma);
    uint64_t z_lo, z_hi, h_lo, h_hi;
    frw::groth16_shard_range(nv + 3, rank, world, &z_lo, &z_hi);
    frw::groth16_shard_range(n - 1, rank, world, &h_lo, &h_hi);
    if (z_hi == z_lo || h_hi == h_lo) return FRW_E_INVALID_ARG;                              // more ranks than rows
    if (hipSetDevice(device) != hipSuccess) return FRW_E_NO_DEVICE;
    frw_msm *tab[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};                         // h, a, b1, l, b2
    frw::FixedBaseGen gen{nullptr, nullptr};
    int rc = FRW_OK;
    // FRW_SETUP_TIMING in the environment: the phases' wall times on stderr (each after a device synchronisation)
    const bool timing = std::getenv("FRW_SETUP_TIMING") != nullptr;
    auto t_last = std::chrono::steady_clock::now();
    auto mark = [&](const char *what) {
        if (!timing) return;
        (void)hipDeviceSynchronize();
        const auto now = std::chrono::steady_clock::now();
        std::fprintf(stderr, "frw setup: %-28s %8.3f s\n", what, std::chrono::duration<double>(now - t_last).count());
        t_last = now;
    };
    try {
        DeviceBuffers dev;
        auto pow_tab = [&](const Fr &base) {
            const PowTabHost pt = pow_tab_host(base, L);
            return frw::SetupPowTab{(const uint32_t *)dev.put(pt.lo.data(), pt.lo.size() * 4), (const uint32_t *)dev.put(pt.hi.data(), pt.hi.size() * 4)};
        };
        // (1) L_i(t) = zt w^i / (n (t - w^i)) for the whole domain
        uint32_t *lag = (uint32_t *)dev.get(n * 32);
        const frw::SetupPowTab wt = pow_tab(w);
        mark("host constants");
        hip_ok(frw::launch_setup_lagrange(n, wt, setup_const(t), setup_const(zt * inverse(Fr::from(n))), setup_const(Fr::one()), lag, nullptr), "setup: Lagrange coefficients");
        mark("Lagrange coefficients");
        // (2) u, v, w: the per-signature matrices by columns, a run of statements per launch
        std::vector<const ConstraintMatrices *> distinct;
        std::vector<std::array<frw::SetupCsc, 3>> csc;
        std::vector<void *> csc_bufs;
        for (const DeviceRun &run : runs) {
            if (std::find(distinct.begin(), distinct.end(), run.m) != distinct.end()) continue;
            distinct.push_back(run.m);
            std::array<frw::SetupCsc, 3> three;
            const std::vector<ConstraintMatrices::Row> *mats[3] = {&run.m->a, &run.m->b, &run.m->c};
            const size_t V = run.m->num_instance_variables + run.m->num_witness_variables;
            for (int k = 0; k < 3; k++) {
                std::vector<uint32_t> ptr(V + 1, 0u);
                for (const auto &row : *mats[k])
                    for (const auto &e : row) ptr[e.second + 1]++;
                for (size_t c = 0; c < V; c++) ptr[c + 1] += ptr[c];
                std::vector<uint32_t> fill(ptr.begin(), ptr.end() - 1), rows(ptr[V]), val((size_t)ptr[V] * 8);
                for (size_t r = 0; r < mats[k]->size(); r++)
                    for (const auto &e : (*mats[k])[r]) {
                        const uint32_t at = fill[e.second]++;
                        rows[at] = (uint32_t)r;
                        packed29(e.first, &val[(size_t)at * 8]);
                    }
                void *b0 = dev.put(ptr.data(), ptr.size() * 4), *b1 = dev.put(rows.data(), rows.size() * 4), *b2 = dev.put(val.data(), val.size() * 4);
                csc_bufs.insert(csc_bufs.end(), {b0, b1, b2});
                three[k] = frw::SetupCsc{(const uint32_t *)b0, (const uint32_t *)b1, (const uint32_t *)b2};
            }
            csc.push_back(three);
        }
        // one launch per parameter set: the offsets of all its statements (a run of the aggregate is `count` consecutive ones)
        std::vector<frw::SetupRun> druns;
        for (size_t which = 0; which < distinct.size(); which++) {
            const ConstraintMatrices &m = *distinct[which];
            std::vector<uint64_t> offs;
            for (const DeviceRun &run : runs) {
                if (run.m != distinct[which]) continue;
                for (uint32_t k = 0; k < run.count; k++) {
                    offs.push_back(run.wit_off + (uint64_t)k * m.num_witness_variables);
                    offs.push_back(run.pub_off + (uint64_t)k * (m.num_instance_variables - 1));
                    offs.push_back(run.row_off + (uint64_t)k * m.num_constraints);
                    offs.push_back(run.first + k);
                }
            }
            void *d_offs = dev.put(offs.data(), offs.size() * 8);
            csc_bufs.push_back(d_offs);
            for (size_t lo = 0; lo < offs.size() / 4; lo += 65535) {           // grid.y
                frw::SetupRun d{};
                for (int k = 0; k < 3; k++) d.m[k] = csc[which][k];
                d.num_inst = (uint32_t)m.num_instance_variables;
                d.num_vars = (uint32_t)(m.num_instance_variables + m.num_witness_variables);
                d.num_constraints = (uint32_t)m.num_constraints;
                d.count = (uint32_t)std::min<size_t>(65535, offs.size() / 4 - lo);
                d.offs = (const uint64_t *)d_offs + 4 * lo;
                druns.push_back(d);
            }
        }
        mark("matrices by columns (host)");
        uint32_t *uvw = (uint32_t *)dev.get(3 * nv * 32), *col0 = (uint32_t *)dev.get((size_t)statements * 3 * 32);
        hip_ok(frw::launch_setup_columns(druns.data(), druns.size(), statements, ni, nc, nv, lag, uvw, col0, nullptr), "setup: the QAP at t");
        hip_ok(hipDeviceSynchronize(), "setup: the QAP at t");
        mark("u, v, w at t");
        dev.drop(lag);
        for (void *p : csc_bufs) dev.drop(p);
        // (3) the five tables, row by row in place
        rc = frw::msm_alloc_bare(device, 1, 16, h_hi - h_lo, h_lo, &tab[0]);
        for (int k = 1; k < 5 && rc == FRW_OK; k++) rc = frw::msm_alloc_bare(device, k == 4 ? 2 : 1, 8, z_hi - z_lo, z_lo, &tab[k]);
        if (rc == FRW_OK) rc = frw::fixed_base_gen_create(device, &gen);
        if (rc != FRW_OK) throw std::runtime_error("tables");
        const size_t zrows = z_hi - z_lo, hrows = h_hi - h_lo;
        uint32_t *scal = (uint32_t *)dev.get(std::max(std::max(zrows, hrows), ni) * 32);
        const frw::SetupConst ca = setup_const(alpha), cb = setup_const(beta), cg = setup_const(ginv), cd = setup_const(dinv);
        // rows [z_lo, z_hi) of a witness-side table: variables [from, nv) by the kernel (kind, which), then the three tail rows
        auto tails = [&](const Fr *tail) {
            for (size_t j = 0; j < 3; j++) {
                const size_t g = nv + j;
                if (g < z_lo || g >= z_hi || !tail) continue;
                uint64_t canon[4];
                tail[j].to_canonical(canon);
                hip_ok(hipMemcpy(scal + (g - z_lo) * 8, canon, 32, hipMemcpyHostToDevice), "setup: scalars");
            }
        };
        auto var_scalars = [&](int kind, int which, size_t from) {
            hip_ok(hipMemset(scal, 0, zrows * 32), "setup: scalars");
            const size_t lo = std::max<size_t>(z_lo, from), hi = std::min<size_t>(z_hi, nv);
            if (lo < hi) hip_ok(frw::launch_setup_var_scalars(kind, lo, hi - lo, ni, nv, uvw, which, ca, cb, cg, cd, scal + (lo - z_lo) * 8, nullptr), "setup: scalars");
        };
        const Fr zero = Fr::zero();
        const Fr tail_a[3] = {alpha, delta, zero}, tail_b1[3] = {beta, zero, zero}, tail_b2[3] = {beta, zero, delta};
        var_scalars(0, 0, 0);
        tails(tail_a);
        mark("tables allocated, generators");
        hip_ok(frw::msm_fill_fixed_base(tab[1], gen, 0, zrows, scal, nullptr), "setup: a_query");
        mark("a_query");
        var_scalars(0, 1, 0);
        tails(tail_b1);
        hip_ok(frw::msm_fill_fixed_base(tab[2], gen, 0, zrows, scal, nullptr), "setup: b_g1_query");
        mark("b_g1_query");
        tails(tail_b2);                                             // the same scalars but for the tail: b_g2_query ++ [beta2, O, delta2]
        hip_ok(frw::msm_fill_fixed_base(tab[4], gen, 0, zrows, scal, nullptr), "setup: b_g2_query");
        mark("b_g2_query");
        var_scalars(1, 0, ni);                                      // l_query: the witness variables; the instance rows stay the point at infinity
        hip_ok(frw::msm_fill_fixed_base(tab[3], gen, 0, zrows, scal, nullptr), "setup: l_query");
        mark("l_query");
        {
            const frw::SetupPowTab tt = pow_tab(t);
            hip_ok(frw::launch_setup_h_scalars(h_lo, hrows, tt, setup_const(zt * dinv), scal, nullptr), "setup: scalars");
            hip_ok(frw::msm_fill_fixed_base(tab[0], gen, 0, hrows, scal, nullptr), "setup: h_query");
        }
        mark("h_query");
        // (4) the verifying key: alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | gamma_abc_g1 [ni]
        if (vk_out) {
            uint32_t *pts = (uint32_t *)dev.get(std::max<size_t>(ni * 96, 4 * 192));
            hip_ok(frw::launch_setup_var_scalars(1, 0, ni, ni, nv, uvw, 0, ca, cb, cg, cd, scal, nullptr), "setup: scalars");
            hip_ok(frw::fixed_base_ark_dev(gen, 1, ni, scal, pts, nullptr), "setup: gamma_abc_g1");
            hip_ok(hipMemcpy(vk_out + 84, pts, ni * 96, hipMemcpyDeviceToHost), "setup: gamma_abc_g1");
            uint64_t fixed[16];
            alpha.to_canonical(fixed); beta.to_canonical(fixed + 4); gamma.to_canonical(fixed + 8); delta.to_canonical(fixed + 12);
            hip_ok(hipMemcpy(scal, fixed, sizeof(fixed), hipMemcpyHostToDevice), "setup: scalars");
            hip_ok(frw::fixed_base_ark_dev(gen, 1, 1, scal, pts, nullptr), "setup: alpha_g1");
            hip_ok(hipMemcpy(vk_out, pts, 96, hipMemcpyDeviceToHost), "setup: alpha_g1");
            hip_ok(frw::fixed_base_ark_dev(gen, 2, 3, scal + 8, pts, nullptr), "setup: the verifying key's G2 points");
            hip_ok(hipMemcpy(vk_out + 12, pts, 3 * 192, hipMemcpyDeviceToHost), "setup: the verifying key's G2 points");
        }
        hip_ok(hipDeviceSynchronize(), "setup");
        mark("verifying key");
        // a key of window tables: grown from the rows, table by table (the rows go as their table comes)
        if (tables) {
            dev.drop(uvw);
            dev.drop(scal);
            for (int k = 0; k < 5 && rc == FRW_OK; k++) rc = frw::msm_expand_tables(&tab[k]);
            if (rc != FRW_OK) throw std::runtime_error("window tables");
        }
    } catch (const std::bad_alloc &) {
        rc = FRW_E_OUT_OF_MEMORY;
    } catch (const std::exception &) {
        if (rc == FRW_OK) rc = FRW_E_HIP;
    }
    frw::fixed_base_gen_free(&gen);
    if (rc != FRW_OK) {
        for (frw_msm *m : tab) frw_msm_free(m);
        return rc;
    }
    return frw::groth16_pk_assemble(device, ni, nw, n, rank, world, tab[0], tab[1], tab[2], tab[3], tab[4], pk_out);
}
}  // namespace

extern "C" int frw_groth16_setup(int device, int circuit, int logn, const uint64_t *toxic /* [5][4]: alpha, beta, gamma, delta, t; canonical */,
                                 frw_groth16_pk **pk_out, uint64_t *vk_out)
{
    using namespace frw::host;
    if (!toxic || !pk_out || (logn != 9 && logn != 10) || (circuit != FRW_CIRCUIT_NTT && circuit != FRW_CIRCUIT_DUAL_NTT)) return FRW_E_INVALID_ARG;
    *pk_out = nullptr;
    try {
        const ConstraintMatrices m = build_matrices(circuit, logn);
        return groth16_setup_blocks(device, {SetupBlock{&m, 0, 0, 0}}, m.num_instance_variables, m.num_witness_variables, m.num_constraints,
                                    toxic, pk_out, vk_out);
    } catch (const std::exception &) {
        return FRW_E_OUT_OF_MEMORY;
    }
}

// The same for the system behind any handle -- in particular an aggregate statement (frw_r1cs_load_aggregate): the QAP at t is
// evaluated block by block from the per-signature matrices, the key has one query point per variable of the WHOLE statement
// (a_query, b_g1_query, b_g2_query: 1 + sum (2 N_i + W_i) points; h_query: domain - 1), and vk_out takes 84 + 12 (1 + sum 2 N_i)
// uint64_t.  opts (null: FRW_KEY_AUTO, the whole key): window tables made through the host (as until round 4), or bare handles made on
// the device end to end (groth16_setup_device), whole or one rank's slices.
extern "C" int frw_groth16_setup_r1cs_opts(const frw_r1cs *r, const uint64_t *toxic, const frw_groth16_key_opts_t *opts, frw_groth16_pk **pk_out,
                                           uint64_t *vk_out)
{
    using namespace frw::host;
    if (!r || !toxic || !pk_out) return FRW_E_INVALID_ARG;
    *pk_out = nullptr;
    int mode = opts ? opts->mode : FRW_KEY_AUTO;
    const uint32_t world = opts && opts->world > 1 ? opts->world : 1, rank = opts ? opts->rank : 0;
    if ((mode != FRW_KEY_AUTO && mode != FRW_KEY_TABLES && mode != FRW_KEY_BARE) || rank >= world) return FRW_E_INVALID_ARG;
    const size_t nv = (size_t)r->dev.num_instance + r->dev.num_witness;
    if (mode == FRW_KEY_AUTO) mode = world > 1 || nv > FRW_KEY_AUTO_TABLE_VARIABLES ? FRW_KEY_BARE : FRW_KEY_TABLES;
    if (mode == FRW_KEY_TABLES && world > 1) return FRW_E_INVALID_ARG;
    // either kind of key is made on the device (groth16_setup_device; window tables are grown from the rows); FRW_SETUP_ON_HOST in the
    // environment keeps round 4's host-side evaluation of the QAP for keys of window tables -- the tests hold the two against each other
    const bool on_host = mode == FRW_KEY_TABLES && std::getenv("FRW_SETUP_ON_HOST") != nullptr;
    try {
        if (!r->dev.agg) {
            if (!r->host_matrices) return FRW_E_INVALID_ARG;
            const ConstraintMatrices &m = *r->host_matrices;
            if (on_host)
                return groth16_setup_blocks(r->device, {SetupBlock{&m, 0, 0, 0}}, m.num_instance_variables, m.num_witness_variables, m.num_constraints,
                                            toxic, pk_out, vk_out);
            return groth16_setup_device(r->device, {DeviceRun{&m, 0, 1, 0, 0, 0}}, 1, m.num_instance_variables, m.num_witness_variables,
                                        m.num_constraints, toxic, mode == FRW_KEY_TABLES, rank, world, pk_out, vk_out);
        }
        const ConstraintMatrices *mats[2] = {nullptr, nullptr};
        for (int g = 0; g < 2; g++)
            if (r->base[g]) mats[g] = r->base[g]->host_matrices.get();
        if (!on_host) {
            std::vector<DeviceRun> runs;
            for (const frw::R1csAggRun &run : r->runs)
                runs.push_back(DeviceRun{mats[r->statement_logn[run.first] - 9], run.first, run.count, run.wit_off, run.pub_off, run.row_off});
            return groth16_setup_device(r->device, runs, r->agg.num_statements, r->dev.num_instance, r->dev.num_witness, r->dev.num_constraints, toxic,
                                        mode == FRW_KEY_TABLES, rank, world, pk_out, vk_out);
        }
        std::vector<SetupBlock> blocks;
        for (const frw::R1csAggRun &run : r->runs) {
            const ConstraintMatrices &m = *mats[r->statement_logn[run.first] - 9];
            for (uint32_t k = 0; k < run.count; k++)
                blocks.push_back(SetupBlock{&m, (size_t)(run.row_off + (uint64_t)k * m.num_constraints),
                                            (size_t)(run.pub_off + (uint64_t)k * (m.num_instance_variables - 1)),
                                            (size_t)(run.wit_off + (uint64_t)k * m.num_witness_variables)});
        }
        return groth16_setup_blocks(r->device, blocks, r->dev.num_instance, r->dev.num_witness, r->dev.num_constraints, toxic, pk_out, vk_out);
    } catch (const std::exception &) {
        return FRW_E_OUT_OF_MEMORY;
    }
}
extern "C" int frw_groth16_setup_r1cs(const frw_r1cs *r, const uint64_t *toxic, frw_groth16_pk **pk_out, uint64_t *vk_out)
{
    return frw_groth16_setup_r1cs_opts(r, toxic, nullptr, pk_out, vk_out);
}

// ---- diagnostics: p(t) on the device for a polynomial whose coefficients no host arithmetic could visit (2^27 of them) --------------------
// d_coeffs: uint64_t[n][4], ark-ff's Montgomery form, coefficient k at index k (what frw_qap_witness_map_dev writes); t: host, canonical;
// out: host, p(t) as a canonical integer.  Synchronous; allocates its own scratch (32 bytes per 4,096 coefficients + two small tables).
extern "C" int frw_diag_poly_eval_dev(int device, uint64_t n, const uint64_t *d_coeffs, const uint64_t *t, uint64_t *out)
{
    if (!d_coeffs || !t || !out || n == 0 || n > ((uint64_t)1 << 30)) return FRW_E_INVALID_ARG;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) return FRW_E_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return FRW_E_HIP;
    try {
        const Fr tv = Fr::from_canonical(t);
        const int L = std::max(domain_log(n), (int)frw::SETUP_POW_LO_BITS);
        const PowTabHost pt = pow_tab_host(tv, L);
        DeviceBuffers dev;
        const frw::SetupPowTab tt{(const uint32_t *)dev.put(pt.lo.data(), pt.lo.size() * 4), (const uint32_t *)dev.put(pt.hi.data(), pt.hi.size() * 4)};
        uint32_t *part = (uint32_t *)dev.get(frw::poly_eval_scratch_bytes(n)), *res = (uint32_t *)dev.get(32);
        hip_ok(frw::launch_poly_eval(n, (const uint32_t *)d_coeffs, setup_const(tv), tt, part, res, nullptr), "frw_diag_poly_eval_dev");
        uint64_t limbs[4];
        hip_ok(hipMemcpy(limbs, res, 32, hipMemcpyDeviceToHost), "frw_diag_poly_eval_dev");
        Fr::from_montgomery(limbs).to_canonical(out);
        return FRW_OK;
    } catch (const std::bad_alloc &) {
        return FRW_E_OUT_OF_MEMORY;
    } catch (const std::exception &) {
        return FRW_E_HIP;
    }
}
