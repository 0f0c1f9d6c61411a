// frw_r1cs_check.hip -- batch R1CS satisfaction check on the device.
//
// The matrices are the ones the host mirror emits from the gadget definitions (host/frw_host.hpp, to_matrices():
// every symbolic LC inlined) -- NOT the closed form the witness kernels implement -- so this is the reference's
// `assert!(cs.is_satisfied())` (circuits/falcon_ntt.rs:159) run against an independently derived constraint system,
// for every signature of a full-size launch, where the witnesses lie (HBM).
//
// Two uses.  (1) frw_r1cs_check_dev, check only: r1cs_check_kernel, one thread per constraint row of one signature --
// three sparse dot products over BLS12-381 Fr (8 x 32-bit Montgomery form) and one field multiplication; rows visited in
// order of decreasing length so that the 64 rows of a wavefront have similar length.  A verification utility.
// (2) frw_r1cs_eval_dev / the QAP witness map (frw_qap.hip), which need A z, B z, C z in HBM: the 2 N + 1 rows that
// arkworks' finalize() turns into N-term linear combinations go to a wavefront each (r1cs_long_rows_small_kernel: plain
// values of their variables extracted once per signature by r1cs_zsmall_kernel, one multiply-add per limb per term;
// r1cs_long_rows_kernel: field products, when no scratch is to be had), everything else to r1cs_eval_kernel (thread per
// row, nine-limb arithmetic of frw_fr29.h, +1 / -1 coefficients without a product).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frw_device.h"
#include "frw_fr29.h"

namespace frw {

struct StatementVars;
__device__ __forceinline__ Fr8 row_dot(const R1csMatrixDev &m, uint32_t row, const uint32_t *__restrict__ wit,
                                       const uint32_t *__restrict__ inst, const uint32_t *__restrict__ one, uint32_t num_instance)
{
    Fr8 acc;
#pragma unroll
    for (int i = 0; i < 8; i++) acc.l[i] = 0;
    const uint64_t lo = m.row_ptr[row], hi = m.row_ptr[row + 1];
    for (uint64_t t = lo; t < hi; t++) {
        const uint32_t col = m.col[t];
        const Fr8 z = fr_load(col >= num_instance ? wit + (size_t)(col - num_instance) * 8 : col ? inst + (size_t)col * 8 : one);
        acc = fr_add(acc, fr_mul(fr_load(m.val + t * 8), z));
    }
    return acc;
}

// One statement's variables through a view (frw_device.h R1csView): the plain batch of the witness entry points, or a run
// of statements inside an aggregate assignment.
struct StatementVars {
    const uint32_t *wit, *inst, *one;
    uint32_t num_instance;
    __device__ __forceinline__ const uint32_t *at(uint32_t col) const
    {
        return col >= num_instance ? wit + (size_t)(col - num_instance) * 8 : col ? inst + (size_t)col * 8 : one;
    }
};
__device__ __forceinline__ StatementVars statement_vars(const R1csDev &r, const R1csView &v, size_t sig)
{
    if (v.offs) return StatementVars{v.wit + (size_t)v.offs[3 * sig] * 8, v.inst + (size_t)v.offs[3 * sig + 1] * 8, v.one, r.num_instance};
    return StatementVars{v.wit + sig * v.wit_stride, v.inst + sig * v.inst_stride, v.one + sig * v.one_stride, r.num_instance};
}
__device__ __forceinline__ uint32_t *product_ptr(const R1csView &v, size_t sig, uint32_t matrix, uint32_t row)
{
    return v.abc + ((v.offs ? (size_t)v.offs[3 * sig + 2] : sig * v.abc_sig_stride) + (size_t)matrix * v.abc_mat_stride + row) * 8;
}

// The fast path of frw_r1cs_eval_dev (abc != nullptr): thread per row in nine-limb arithmetic (frw_fr29.h).
// Term classes (host-computed, top two bits of the column index): coefficient +1 / -1 (55 % of the non-long terms of the
// Falcon circuits: no product), anything else as f29_mul(z R, c R') with c R' stored packed.  Rows are visited by
// decreasing length (r.order), so the 64 rows of a wavefront have similar length; matrices in which the row is long
// (r.long_mask) are skipped -- r1cs_long_rows_kernel has already put their product into abc.
constexpr uint32_t TERM_PLUS_ONE = 1u, TERM_MINUS_ONE = 2u;      // 0: general coefficient
__device__ __forceinline__ F29 row_dot29(const R1csMatrixDev &m, uint32_t row, const StatementVars &z_of)
{
    F29 acc;
#pragma unroll
    for (int i = 0; i < NL29; i++) acc.l[i] = 0;
    const uint64_t lo = m.row_ptr[row], hi = m.row_ptr[row + 1];
    for (uint64_t t = lo; t < hi; t++) {
        const uint32_t cc = m.col_class[t], col = cc & 0x3fffffffu, cls = cc >> 30;
        const F29 z = f29_unpack(fr_load(z_of.at(col)));
        if (cls == TERM_PLUS_ONE) acc = f29_reduce_4p(f29_add(acc, z));
        else if (cls == TERM_MINUS_ONE) acc = f29_reduce_4p(f29_sub_2p(acc, z));
        else acc = f29_reduce_4p(f29_add(acc, f29_mul(z, f29_unpack(fr_load(m.val29 + t * 8)))));
    }
    return acc;                                    // < 2 p
}

// (A z)(B z) = C z for one row, all three canonical in the form x R: (32 Az R)(Bz R) / R' = Az Bz R against Cz R -- ONE product
// (with both sides brought to the form x R / 32 it was two, a fifth of the evaluation kernels' instructions).  32 x a canonical
// value is below 2^260: a left operand f29_mul takes as it is once its limbs are carried.
__device__ __forceinline__ bool row_holds(const F29 (&v)[3])
{
    F29 a32;
    uint32_t carry = 0;
#pragma unroll
    for (int k = 0; k < NL29; k++) {
        const uint32_t x = v[0].l[k];
        a32.l[k] = k + 1 < NL29 ? ((x << 5) & M29) | carry : (x << 5) | carry;
        carry = x >> 24;
    }
    const F29 ab = f29_canonical(f29_mul(a32, v[1]));
    bool eq = true;
#pragma unroll
    for (int k = 0; k < NL29; k++) eq &= ab.l[k] == v[2].l[k];
    return eq;
}

// STORE: write A z, B z, C z to abc (and read the long rows' products from there).  !STORE: check only -- nothing is
// written; the long rows' products are read from `long_out` ([signature][long row], r.long_slot maps (matrix, row) to it).
#ifndef FRW_EVAL_WAVES
#define FRW_EVAL_WAVES 4
#endif
template <bool STORE>
__global__ __launch_bounds__(BLOCK, FRW_EVAL_WAVES) void r1cs_eval_kernel(R1csDev r, size_t batch, const R1csView view,
                                                          const uint32_t *__restrict__ long_out)
{
    const size_t sig = blockIdx.y;
    if (sig >= batch) return;
    const StatementVars z_of = statement_vars(r, view, sig);
    const uint32_t *lo = STORE ? nullptr : long_out + sig * (size_t)r.num_long * 8;
    unsigned bad = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < r.num_constraints; i += gridDim.x * BLOCK) {
        const uint32_t row = r.order[i], mask = r.long_mask[row];
        F29 v[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            const R1csMatrixDev &mat = m == 0 ? r.a : m == 1 ? r.b : r.c;
            uint32_t *om = STORE ? product_ptr(view, sig, m, row) : nullptr;
            if (mask & (1u << m)) {
                v[m] = f29_unpack(fr_load(STORE ? om : lo + (size_t)r.long_slot[(size_t)m * r.num_constraints + row] * 8));
            } else {
                v[m] = f29_canonical(row_dot29(mat, row, z_of));
                if (STORE) fr_store(om, f29_pack(v[m]));
            }
        }
        bad += row_holds(v) ? 0u : 1u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && bad && view.flags) atomicAdd(&view.flags[sig * view.flag_stride], bad);
}

// The same work from the flattened rows (frw_device.h flat_*; r.flat_term != null).  r1cs_eval_kernel walks three CSR
// matrices per row -- order -> long_mask -> row_ptr x 2 -> col_class -> variable (-> coefficient), three times over: a dozen
// dependent loads for a row of three terms, and its waves spend two thirds of their cycles parked on them
// (profiles/r03_qap_counters.txt).  Here a row is ONE header (coalesced), its term words four at a time (independent loads),
// then their variables (independent again): three dependent loads for the 100,000 three-term rows of a Falcon-1024 system, five
// for the six-term ones; the 16 distinct coefficients of a Falcon circuit wait in LDS, and a wavefront whose rows have none but
// +1 / -1 (the boolean gates) skips the field product altogether.  Every (M z)_row is stored canonical, so the order of
// summation cannot show: bit for bit what r1cs_eval_kernel writes.
template <bool STORE>
__global__ __launch_bounds__(BLOCK, FRW_EVAL_WAVES) void r1cs_eval_flat_kernel(R1csDev r, size_t batch, const R1csView view,
                                                                               const uint32_t *__restrict__ long_out)
{
    __shared__ uint32_t coef[R1CS_FLAT_COEFS * 8];
    for (uint32_t k = threadIdx.x; k < r.flat_num_coefs * 8; k += BLOCK) coef[k] = r.flat_coef[k];
    __syncthreads();
    const size_t sig = blockIdx.y;
    if (sig >= batch) return;
    const StatementVars z_of = statement_vars(r, view, sig);
    const uint32_t *lo = STORE ? nullptr : long_out + sig * (size_t)r.num_long * 8;
    unsigned bad = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < r.num_constraints; i += gridDim.x * BLOCK) {
        const uint4 head = ((const uint4 *)r.flat_head)[i];
        const uint4 *terms = (const uint4 *)r.flat_term + head.z;
        const uint32_t row = head.x, n_a = head.y & 0xffu, n_ab = n_a + ((head.y >> 8) & 0xffu), n = n_ab + ((head.y >> 16) & 0xffu),
                       mask = head.y >> 24;
        // ONE running sum: the terms come A first, then B, then C, so the sum is parked (and restarted) where t reaches nA and nA + nB
        F29 acc[3], run;
#pragma unroll
        for (int k = 0; k < NL29; k++) run.l[k] = acc[0].l[k] = acc[1].l[k] = acc[2].l[k] = 0;
        for (uint32_t t0 = 0; t0 < n; t0 += 4) {
            const uint4 w4 = terms[t0 >> 2];                                                      // (padded with 0: the constant one -- not added)
            const uint32_t w[4] = {w4.x, w4.y, w4.z, w4.w};
            Fr8 zw[4];
#pragma unroll
            for (int k = 0; k < 4; k++)
                if (__any(t0 + k < n)) zw[k] = fr_load(z_of.at(w[k] & 0x00ffffffu));
                else zw[k] = zw[0];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t t = t0 + k, idx = w[k] >> 24;
                const bool live = t < n;
                if (!__any(live)) break;
                const bool at_b = live && t == n_a, at_c = live && t == n_ab;                      // (both when B is empty: A's sum parks, B's stays zero)
#pragma unroll
                for (int j = 0; j < NL29; j++) {
                    acc[0].l[j] = at_b ? run.l[j] : acc[0].l[j];
                    acc[1].l[j] = at_c && !at_b ? run.l[j] : acc[1].l[j];
                    run.l[j] = at_b || at_c ? 0u : run.l[j];
                }
                F29 c = f29_unpack(zw[k]);
                if (__any(live && idx >= 2u)) {                                                    // some row of the wavefront has a general coefficient here
                    const F29 prod = f29_mul(c, f29_unpack(fr_load(coef + (idx < 2u ? 0u : idx) * 8)));
#pragma unroll
                    for (int j = 0; j < NL29; j++) c.l[j] = idx >= 2u ? prod.l[j] : c.l[j];
                }
                const F29 next = idx == 1u ? f29_reduce_4p(f29_sub_2p(run, c)) : f29_reduce_4p(f29_add(run, c));
#pragma unroll
                for (int j = 0; j < NL29; j++) run.l[j] = live ? next.l[j] : run.l[j];
            }
        }
        // what is still running belongs to the last matrix that has terms
        {
            const bool to_a = n_a == n, to_b = !to_a && n_ab == n;
#pragma unroll
            for (int j = 0; j < NL29; j++) {
                acc[0].l[j] = to_a ? run.l[j] : acc[0].l[j];
                acc[1].l[j] = to_b ? run.l[j] : acc[1].l[j];
                acc[2].l[j] = !to_a && !to_b ? run.l[j] : acc[2].l[j];
            }
        }
        F29 v[3];
#pragma unroll
        for (int m = 0; m < 3; m++) {
            uint32_t *om = STORE ? product_ptr(view, sig, m, row) : nullptr;
            if (mask & (1u << m)) {
                v[m] = f29_unpack(fr_load(STORE ? om : lo + (size_t)r.long_slot[(size_t)m * r.num_constraints + row] * 8));
            } else {
                v[m] = f29_canonical(acc[m]);
                if (STORE) fr_store(om, f29_pack(v[m]));
            }
        }
        bad += row_holds(v) ? 0u : 1u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && bad && view.flags) atomicAdd(&view.flags[sig * view.flag_stride], bad);
}

// Check only, when no scratch is to be had for the long rows' products: one thread per row in the 8 x 32-bit form, longest
// rows first, so that the dense ladder rows spread over many waves.
__global__ __launch_bounds__(BLOCK) void r1cs_check_kernel(R1csDev r, size_t batch, const R1csView view)
{
    const size_t sig = blockIdx.y;
    if (sig >= batch) return;
    const StatementVars sv = statement_vars(r, view, sig);
    const uint32_t *wit = sv.wit, *inst = sv.inst, *one = sv.one;
    unsigned bad = 0;
    for (uint32_t i = blockIdx.x * BLOCK + threadIdx.x; i < r.num_constraints; i += gridDim.x * BLOCK) {
        const uint32_t row = r.order[i];
        const Fr8 az = row_dot(r.a, row, wit, inst, one, r.num_instance);
        const Fr8 bz = row_dot(r.b, row, wit, inst, one, r.num_instance);
        const Fr8 cz = row_dot(r.c, row, wit, inst, one, r.num_instance);
        const Fr8 ab = fr_mul(az, bz);              // (Az R)(Bz R)/R = Az Bz R
        bool eq = true;
#pragma unroll
        for (int k = 0; k < 8; k++) eq &= ab.l[k] == cz.l[k];
        bad += eq ? 0u : 1u;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) bad += __shfl_xor(bad, off, WAVE);
    if ((threadIdx.x & (WAVE - 1)) == 0 && bad && view.flags) atomicAdd(&view.flags[sig * view.flag_stride], bad);
}

// One wavefront per (long row, group of LONG_SIGS signatures): lane l multiplies terms l, l + 64, ... (coefficient loaded
// once, used for every signature of the group), the 64 partial sums are added across the wavefront, lane 0 stores.
// Products are f29_mul(z R, c R') = z c R (< 2 p); they are summed limb-wise with a carry pass every four and a reduction
// (a product with R' mod p) every sixteen, so the running value stays far below 2^261 = 70 p.
constexpr int LONG_SIGS = 4;
// where a long row's product goes: its slot in abc, or (check only) [signature][long row] of the scratch
__device__ __forceinline__ uint32_t *long_row_out(const R1csDev &r, const R1csLongRow &d, const R1csView &view, uint32_t *long_out, size_t sig)
{
    return view.abc ? product_ptr(view, sig, d.matrix, d.row) : long_out + (sig * r.num_long + blockIdx.x) * 8;
}

__global__ __launch_bounds__(WAVE) void r1cs_long_rows_kernel(R1csDev r, size_t batch, const R1csView view, uint32_t *__restrict__ long_out)
{
    const R1csLongRow d = r.long_rows[blockIdx.x];
    const int lane = threadIdx.x;
    const size_t sig0 = (size_t)blockIdx.y * LONG_SIGS;
    F29 acc[LONG_SIGS];
#pragma unroll
    for (int s = 0; s < LONG_SIGS; s++)
#pragma unroll
        for (int k = 0; k < NL29; k++) acc[s].l[k] = 0;
    // R' mod p = 2^5 R mod p as an integer: f29_mul(x, R' mod p) = x mod p, brought under 2 p
    constexpr uint32_t R32[8] = FRW_R32;
    Fr8 r_words;
#pragma unroll
    for (int k = 0; k < 8; k++) r_words.l[k] = R32[k];
    F29 one_rp = f29_unpack(r_words);
#pragma unroll
    for (int k = 0; k < 5; k++) one_rp = f29_reduce_4p(f29_add(one_rp, one_rp));       // x 32, kept < 2 p
    one_rp = f29_canonical(one_rp);
    for (uint32_t ch = 0; ch < d.num_chunks; ch++) {
        const size_t chunk = (size_t)d.first_chunk + ch;
        const uint32_t col = r.long_col[chunk * WAVE + lane];
        F29 c;
#pragma unroll
        for (int k = 0; k < NL29; k++) c.l[k] = r.long_coef[(chunk * NL29 + k) * WAVE + lane];
#pragma unroll
        for (int s = 0; s < LONG_SIGS; s++) {
            const size_t sig = sig0 + s < batch ? sig0 + s : batch - 1;       // a ragged last group repeats the last signature
            const F29 z = f29_unpack(fr_load(statement_vars(r, view, sig).at(col)));
            const F29 prod = f29_mul(z, c);                       // < 2 p, normalised
#pragma unroll
            for (int k = 0; k < NL29; k++) acc[s].l[k] += prod.l[k];          // lazily: limbs < 4 x 2^29 between carries
        }
        if ((ch & 3u) == 3u) {
#pragma unroll
            for (int s = 0; s < LONG_SIGS; s++) f29_normalise(acc[s]);
        }
        if ((ch & 15u) == 15u) {                                  // every 16 terms bring the sum (< 34 p) back under 2 p
#pragma unroll
            for (int s = 0; s < LONG_SIGS; s++) acc[s] = f29_mul(acc[s], one_rp);
        }
    }
#pragma unroll
    for (int s = 0; s < LONG_SIGS; s++) {
        f29_normalise(acc[s]);
        acc[s] = f29_mul(acc[s], one_rp);                         // x R' / R' = x, < 2 p
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            F29 other;
#pragma unroll
            for (int k = 0; k < NL29; k++) other.l[k] = (uint32_t)__shfl_xor((int)acc[s].l[k], off, WAVE);
            acc[s] = f29_reduce_4p(f29_add(acc[s], other));
        }
        if (lane == 0 && sig0 + s < batch) fr_store(long_row_out(r, d, view, long_out, sig0 + s), f29_pack(f29_canonical(acc[s])));
    }
}

// The plain value of every variable the long rows read, when it is below 2^28 (else ~0): f29_mul(z R, 32) = z.
__global__ __launch_bounds__(BLOCK) void r1cs_zsmall_kernel(R1csDev r, size_t batch, const R1csView view, uint32_t *__restrict__ zs)
{
    const size_t sig = blockIdx.y;
    const uint32_t u = blockIdx.x * BLOCK + threadIdx.x;
    if (sig >= batch || u >= r.num_long_vars) return;
    const uint32_t col = r.long_vars[u];
    const uint32_t *zp = statement_vars(r, view, sig).at(col);
    F29 c32;
#pragma unroll
    for (int k = 0; k < NL29; k++) c32.l[k] = k ? 0u : 32u;
    const F29 v = f29_canonical(f29_mul(f29_unpack(fr_load(zp)), c32));
    uint32_t hi = 0;
#pragma unroll
    for (int k = 1; k < NL29; k++) hi |= v.l[k];
    zs[sig * r.num_long_vars + u] = (hi == 0 && v.l[0] < (1u << 28)) ? v.l[0] : 0xffffffffu;
}

// The long rows again, for witnesses whose long-row variables are small integers (zs from r1cs_zsmall_kernel): a term is
// c R' (nine limbs) times a 28-bit integer, accumulated per limb in 64 bits without any reduction -- one multiply-add per
// limb instead of a field product.  A term whose variable is not small takes the field product and enters the same
// columns as 32 z c R.  At the end the 64 lanes' columns are carried into 29-bit limbs, added across the wavefront, and
// X = 32 R sum(c z) becomes sum(c z) R through one Montgomery reduction (X / R') and one product with R R' mod p.
// (occupancy A/B, tools/ab_qap.py, profiles/r04_qap_long_rows_occupancy_ab.txt: asking for three waves per SIMD -- 168 registers, 128 bytes
// of scratch -- changes nothing, 1.136 against 1.139 ms of sparse products per 64 signatures; four -- 304 bytes of scratch -- doubles them)
#ifndef FRW_LONG_WAVES
#define FRW_LONG_WAVES 2
#endif
__global__ __launch_bounds__(WAVE, FRW_LONG_WAVES) void r1cs_long_rows_small_kernel(R1csDev r, size_t batch, const R1csView view,
                                                                    const uint32_t *__restrict__ zs, uint32_t *__restrict__ long_out)
{
    const R1csLongRow d = r.long_rows[blockIdx.x];
    const int lane = threadIdx.x;
    const size_t sig0 = (size_t)blockIdx.y * LONG_SIGS;
    uint64_t col[LONG_SIGS][NL29 + 1];                             // column 9 takes the carries of rows of any length
#pragma unroll
    for (int s = 0; s < LONG_SIGS; s++)
#pragma unroll
        for (int k = 0; k <= NL29; k++) col[s][k] = 0;
    for (uint32_t ch = 0; ch < d.num_chunks; ch++) {
        const size_t chunk = (size_t)d.first_chunk + ch;
        const uint32_t cidx = r.long_cidx[chunk * WAVE + lane];
        F29 c;
#pragma unroll
        for (int k = 0; k < NL29; k++) c.l[k] = r.long_coef[(chunk * NL29 + k) * WAVE + lane];
        uint32_t zv[LONG_SIGS];
        bool all_small = true;
#pragma unroll
        for (int s = 0; s < LONG_SIGS; s++) {
            const size_t sig = sig0 + s < batch ? sig0 + s : batch - 1;       // a ragged last group repeats the last signature
            zv[s] = zs[sig * r.num_long_vars + cidx];
            all_small &= zv[s] != 0xffffffffu;
        }
        if (!__all(all_small)) {
            // field products for the terms whose variable is large; they enter the columns as 32 (z c R)
            const uint32_t colv = r.long_col[chunk * WAVE + lane];
#pragma unroll
            for (int s = 0; s < LONG_SIGS; s++) {
                if (zv[s] == 0xffffffffu) {
                    const size_t sig = sig0 + s < batch ? sig0 + s : batch - 1;
                    const F29 prod = f29_mul(f29_unpack(fr_load(statement_vars(r, view, sig).at(colv))), c);
#pragma unroll
                    for (int k = 0; k < NL29; k++) col[s][k] += (uint64_t)prod.l[k] << 5;
                    zv[s] = 0;
                }
            }
        }
#pragma unroll
        for (int s = 0; s < LONG_SIGS; s++)
#pragma unroll
            for (int k = 0; k < NL29; k++) col[s][k] += (uint64_t)c.l[k] * zv[s];      // < 2^57 per term: 64 terms fit
        if ((ch & 31u) == 31u) {                                   // carry pass every 32 terms (rows of more than 2,048 terms)
#pragma unroll
            for (int s = 0; s < LONG_SIGS; s++)
#pragma unroll
                for (int k = 0; k < NL29; k++) { col[s][k + 1] += col[s][k] >> 29; col[s][k] &= M29; }
        }
    }
    F29 krrp;
#pragma unroll
    for (int k = 0; k < NL29; k++) krrp.l[k] = r.k_rrp[k];
#pragma unroll
    for (int s = 0; s < LONG_SIGS; s++) {
        // this lane's columns -> 29-bit limbs x[0..10]
        uint32_t x[11];
        uint64_t carry = 0;
#pragma unroll
        for (int k = 0; k <= NL29; k++) { const uint64_t t = col[s][k] + carry; x[k] = (uint32_t)t & M29; carry = t >> 29; }
        x[10] = (uint32_t)carry;
        // sum over the 64 lanes, a carry pass every second level (4 x 2^29 < 2^32)
#pragma unroll
        for (int level = 0; level < 6; level++) {
#pragma unroll
            for (int k = 0; k < 11; k++) x[k] += (uint32_t)__shfl_xor((int)x[k], 32 >> level, WAVE);
            if (level & 1) {
                uint32_t cy = 0;
#pragma unroll
                for (int k = 0; k < 10; k++) { const uint32_t t = x[k] + cy; x[k] = t & M29; cy = t >> 29; }
                x[10] += cy;
            }
        }
        const F29 sum = f29_redc_wide(x);                          // X / R' = sum(c z), < 2 p
        const F29 res = f29_canonical(f29_mul(sum, krrp));         // sum(c z) R
        if (lane == 0 && sig0 + s < batch) fr_store(long_row_out(r, d, view, long_out, sig0 + s), f29_pack(res));
    }
}

// (A/B, negative, round 4: a workgroup of four wavefronts taking one long row for 16 signatures, the row's coefficients -- 36 bytes a
// term, which every group of four signatures fetches again -- coming through LDS once per workgroup, double-buffered with one barrier
// per 64 terms: 725 us per 64 signatures against 469.  The refetches hit in cache; the barrier and two waves per SIMD cost more.)

// Scratch of the evaluation: the small values of the long rows' variables (batch x num_long_vars x 4 bytes) and, when
// the products are not kept (abc == nullptr), the long rows' results (batch x num_long x 32 bytes).  An aggregate's runs
// are evaluated one after the other on one stream: they share the scratch of the largest.
size_t r1cs_check_scratch_bytes(const R1csDev &r, size_t batch, bool with_abc)
{
    if (r.agg) {
        size_t need = 0;
        for (int g = 0; g < 2; g++) {
            if (!r.agg->set[g].count) continue;
            const size_t b = r1cs_check_scratch_bytes(*r.agg->set[g].base, r.agg->set[g].count, with_abc);
            need = b > need ? b : need;
        }
        return need;
    }
    if (!r.num_long) return 0;
    const size_t zs = (batch * (size_t)r.num_long_vars * sizeof(uint32_t) + 255) & ~(size_t)255;
    return zs + (with_abc ? 0 : batch * (size_t)r.num_long * 32);
}

namespace {
// one launch sequence over `batch` statements of the system `r` as `view` lays them out
hipError_t launch_view(const R1csDev &r, size_t batch, const R1csView &view, hipStream_t st, void *caller_scratch)
{
    if (batch == 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;
    const unsigned gx = (r.num_constraints + BLOCK - 1) / BLOCK;
    const dim3 egrid(gx > 64 ? 64 : gx, (unsigned)batch);
    const bool abc = view.abc != nullptr;
    // stream-ordered scratch: the small values of the long rows' variables, and (check only) the long rows' products
    uint32_t *zs = nullptr, *long_out = nullptr;
    bool scratch = true;
    const bool lent = caller_scratch != nullptr;
    if (r.num_long && lent) {
        zs = (uint32_t *)caller_scratch;
        if (!abc) long_out = (uint32_t *)((char *)caller_scratch + r1cs_check_scratch_bytes(r, batch, true));
    } else if (r.num_long) {
        scratch = hipMallocAsync((void **)&zs, batch * (size_t)r.num_long_vars * sizeof(uint32_t), st) == hipSuccess && zs;
        if (scratch && !abc) scratch = hipMallocAsync((void **)&long_out, batch * (size_t)r.num_long * 32, st) == hipSuccess && long_out;
        if (!scratch) (void)hipGetLastError();
    }
    if (abc || scratch) {
        if (r.num_long) {
            const dim3 grid(r.num_long, (unsigned)((batch + LONG_SIGS - 1) / LONG_SIGS));
            if (scratch) {
                hipLaunchKernelGGL(r1cs_zsmall_kernel, dim3((r.num_long_vars + BLOCK - 1) / BLOCK, (unsigned)batch), dim3(BLOCK), 0, st,
                                   r, batch, view, zs);
                hipLaunchKernelGGL(r1cs_long_rows_small_kernel, grid, dim3(WAVE), 0, st, r, batch, view, zs, long_out);
            } else {                                              // abc given, no scratch: every term a field product
                hipLaunchKernelGGL(r1cs_long_rows_kernel, grid, dim3(WAVE), 0, st, r, batch, view, long_out);
            }
        }
        if (r.flat_term) {
            if (abc) hipLaunchKernelGGL(r1cs_eval_flat_kernel<true>, egrid, dim3(BLOCK), 0, st, r, batch, view, long_out);
            else hipLaunchKernelGGL(r1cs_eval_flat_kernel<false>, egrid, dim3(BLOCK), 0, st, r, batch, view, long_out);
        } else if (abc) hipLaunchKernelGGL(r1cs_eval_kernel<true>, egrid, dim3(BLOCK), 0, st, r, batch, view, long_out);
        else hipLaunchKernelGGL(r1cs_eval_kernel<false>, egrid, dim3(BLOCK), 0, st, r, batch, view, long_out);
    } else {
        hipLaunchKernelGGL(r1cs_check_kernel, egrid, dim3(BLOCK), 0, st, r, batch, view);
    }
    const hipError_t e = hipGetLastError();
    if (zs && !lent) (void)hipFreeAsync(zs, st);
    if (long_out && !lent) (void)hipFreeAsync(long_out, st);
    return e;
}
}  // namespace

// `caller_scratch` (optional, at least r1cs_check_scratch_bytes): the QAP entry points lend part of their workspace, and
// frw_r1cs_eval_scratch_dev passes the caller's buffer -- then nothing is allocated and the call is capture-safe.  Without
// it a stream-ordered allocation is made for the duration of the call (and the slow kernels run should that fail).
// witness / instance / abc are [batch][W][4], [batch][I][4], [batch][3][C][4] of the system `r` -- for an aggregate the
// aggregate's own vectors, which every run reads in place.
hipError_t launch_r1cs_check(const R1csDev &r, size_t batch, const uint64_t *witness, const uint64_t *instance,
                             uint32_t *num_unsatisfied, uint64_t *abc, hipStream_t st, void *caller_scratch)
{
    if (batch == 0) return hipSuccess;
    if (batch > 65535) return hipErrorInvalidValue;
    if (num_unsatisfied) {
        hipError_t e = hipMemsetAsync(num_unsatisfied, 0, batch * sizeof(uint32_t), st);
        if (e != hipSuccess) return e;
    }
    const uint32_t *wit = (const uint32_t *)witness, *inst = (const uint32_t *)instance;
    const size_t W = r.num_witness, I = r.num_instance, C = r.num_constraints;
    if (!r.agg) {
        const R1csView v{wit, W * 8, inst, I * 8, inst, I * 8, (uint32_t *)abc, 3 * C, C, num_unsatisfied, 1, nullptr};
        return launch_view(r, batch, v, st, caller_scratch);
    }
    for (size_t b = 0; b < batch; b++) {
        const uint32_t *bw = wit + b * W * 8, *bi = inst + b * I * 8;
        uint32_t *ba = abc ? (uint32_t *)abc + b * 3 * C * 8 : nullptr;
        // one launch sequence per PARAMETER SET: its statements found through their offsets, wherever they stand
        for (int g = 0; g < 2; g++) {
            const R1csAggSet &set = r.agg->set[g];
            if (!set.count) continue;
            const R1csView v{bw, 0, bi, 0, bi, 0, ba, 0, C, num_unsatisfied ? num_unsatisfied + b : nullptr, 0, set.offs};
            const hipError_t e = launch_view(*set.base, set.count, v, st, caller_scratch);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

}  // namespace frw
