// frw_setup.hip -- the parts of a key's and a domain's SETUP that grow with the statement, on the device (round 5).
//
// examples/pok_sig.rs:30-31 of the reference calls Groth16::<Bls12_381>::circuit_specific_setup once per circuit; ark-groth16 0.3.0
// generator.rs (generate_parameters) then evaluates the QAP at the toxic point t -- u_i(t) = sum_rows A[row][i] L_row(t) and likewise
// v_i, w_i from B, C, with L_row(t) = zt w^row / (n (t - w^row)) (ark-poly's evaluate_all_lagrange_coefficients) -- and makes every
// query a fixed-base multiple of the generators.  Until round 4 the host did the first half (seconds for one Falcon circuit, a minute
// and 100 GB of host memory for the 1,024-statement aggregate of BASELINE configs[4], whose domain is 2^27), and also built the
// thirteen per-index factor tables of the witness map's transforms (frw_qap.hip) one sequential product at a time.  Here:
//   qap_table_kernel        a transform table: first x base^(e(i)) for every index i, from two small power tables of the base
//   setup_lagrange_kernel   L_i(t) for the whole domain: chunks of 64 indices per thread, Montgomery's trick inside a chunk, ONE
//                           Fermat inversion per chunk
//   setup_columns_kernel    the transposed sparse products: a thread per (statement, variable) walks the variable's column of the
//                           per-signature matrices (CSC) against the statement's slice of L -- the aggregate's matrices are block
//                           diagonal but for column 0, whose per-statement sums a last kernel adds up
//   setup_*_scalars_kernel  the scalars of a_query / b_query / l_query / gamma_abc / h_query as canonical integers, which the
//                           fixed-base kernels of frw_msm.hip turn into table rows in place
// Arithmetic: frw_fr29.h; every stored element is x R' (R' = 2^261) canonical, packed in 8 x 32 bits, unless said otherwise.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "frw_device.h"
#include "frw_fr29.h"

namespace frw {

namespace {
__device__ __forceinline__ F29 ld29(const uint32_t *p) { return f29_unpack(fr_load(p)); }
__device__ __forceinline__ void st29(uint32_t *p, const F29 &v) { fr_store(p, f29_pack(f29_canonical(v))); }      // v < 2 p
__device__ __forceinline__ F29 k29(const SetupConst &c)
{
    F29 r;
#pragma unroll
    for (int k = 0; k < NL29; k++) r.l[k] = c.l[k];
    return r;
}
// base^e R' (< 2 p) from lo[k] = base^k (k < 2^14) and hi[k] = base^(k 2^14)
__device__ __forceinline__ F29 pow_tab(const SetupPowTab &t, uint64_t e)
{
    return f29_mul(ld29(t.lo + (size_t)(e & (SETUP_POW_LO - 1)) * 8), ld29(t.hi + (size_t)(e >> SETUP_POW_LO_BITS) * 8));
}
// a^(p - 2): Fermat's inversion, 254 squarings and 130-odd products (once per 64 elements)
__device__ F29 f29_inv(const F29 &a)
{
    constexpr uint32_t P[8] = FRW_P32;
    static_assert(P[0] == 1u && P[1] == 0xffffffffu, "p - 2: the low word borrows from the second");
    F29 r = a;                                                      // bit 254 of p - 2 is set
#pragma nounroll
    for (int bit = 253; bit >= 0; bit--) {
        r = f29_mul(r, r);
        const int k = bit >> 5;
        const uint32_t w = k == 0 ? 0xffffffffu : k == 1 ? 0xfffffffeu : P[k];
        if ((w >> (bit & 31)) & 1u) r = f29_mul(r, a);
    }
    return r;
}
}  // namespace

// ---- transform tables ------------------------------------------------------------------------------------------------------------
// mode 0: e(i) = i (the scale tables: first x step^i); mode 1: the twist before the pass on the index bits [sh, sh + ts):
// e(i) = (i mod 2^sh) x bitrev_ts((i >> sh) mod 2^ts) x 2^(L - sh - ts)   (tools/dev/qap_fourstep_model.py)
__global__ __launch_bounds__(256) void qap_table_kernel(uint64_t n, SetupPowTab t, int mode, int sh, int ts, int L, SetupConst first, int has_first,
                                                        uint32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint64_t e = i;
    if (mode == 1) {
        const uint64_t low = i & (((uint64_t)1 << sh) - 1);
        const uint32_t r = (uint32_t)(i >> sh) & ((1u << ts) - 1u);
        e = (low * (uint64_t)(__brev(r) >> (32 - ts))) << (L - sh - ts);
    }
    F29 v = pow_tab(t, e);
    if (has_first) v = f29_mul(v, k29(first));
    st29(out + (size_t)i * 8, v);
}

hipError_t launch_qap_table(uint64_t n, const SetupPowTab &t, int mode, int sh, int ts, int L, const SetupConst *first, uint32_t *out, hipStream_t st)
{
    SetupConst f{};
    if (first) f = *first;
    hipLaunchKernelGGL(qap_table_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, n, t, mode, sh, ts, L, f, first ? 1 : 0, out);
    return hipGetLastError();
}

// ---- L_i(t) = c w^i / (t - w^i), c = zt / n, for the whole domain ------------------------------------------------------------------
constexpr int LAG_CHUNK = 64;
__global__ __launch_bounds__(64) void setup_lagrange_kernel(uint64_t n, SetupPowTab wt, SetupConst t_, SetupConst c_, SetupConst one_, uint32_t *__restrict__ lag)
{
    const uint64_t first = ((uint64_t)blockIdx.x * 64 + threadIdx.x) * LAG_CHUNK;
    if (first >= n) return;
    const int cnt = n - first < (uint64_t)LAG_CHUNK ? (int)(n - first) : LAG_CHUNK;
    const F29 t = k29(t_), c = k29(c_);
    // forward: the running product of the denominators, the prefix of index k parked in its slot
    F29 acc = k29(one_);
    for (int k = 0; k < cnt; k++) {
        const F29 wi = f29_reduce_4p(pow_tab(wt, first + k));      // (< 2 p already; the reduction keeps the bound explicit)
        const F29 den = f29_sub_kp<2>(t, wi);                      // t - w^i + 2 p, < 3 p
        fr_store(lag + (first + k) * 8, f29_pack(acc));            // < 2 p: fits the 256 bits
        acc = f29_mul(acc, den);
    }
    F29 inv = f29_inv(acc);
    for (int k = cnt - 1; k >= 0; k--) {
        const F29 wi = pow_tab(wt, first + k);
        const F29 den = f29_sub_kp<2>(t, wi);
        const F29 pre = ld29(lag + (first + k) * 8);
        st29(lag + (first + k) * 8, f29_mul(f29_mul(c, wi), f29_mul(inv, pre)));
        inv = f29_mul(inv, den);
    }
}

hipError_t launch_setup_lagrange(uint64_t n, const SetupPowTab &wt, const SetupConst &t, const SetupConst &c, const SetupConst &one, uint32_t *lag, hipStream_t st)
{
    const uint64_t threads = (n + LAG_CHUNK - 1) / LAG_CHUNK;
    hipLaunchKernelGGL(setup_lagrange_kernel, dim3((unsigned)((threads + 63) / 64)), dim3(64), 0, st, n, wt, t, c, one, lag);
    return hipGetLastError();
}

// ---- u, v, w: the QAP's polynomials at t, variable by variable -----------------------------------------------------------------------
// grid.y = the statements of one parameter set (wherever they sit in the aggregate: offs[s] = where statement s finds its witness
// variables, its public inputs and its rows); a thread per variable of the per-signature system -- but for variable 0, the constant one:
// it stands in 150,000 rows of A (every booleanity constraint (1 - b) b = 0), a chain no thread should walk alone (it was 0.2 s per
// launch: 94 s for the 516 runs of the 1,024-statement mix); setup_column0_kernel gives it a workgroup per (statement, matrix).
__global__ __launch_bounds__(256) void setup_columns_kernel(SetupRun run, uint64_t num_instance_all, size_t num_vars_all, const uint32_t *__restrict__ lag,
                                                            uint32_t *__restrict__ uvw /* [3][num_vars_all][8] */)
{
    const uint32_t c = blockIdx.x * 256 + threadIdx.x, s = blockIdx.y;
    if (c == 0 || c >= run.num_vars) return;
    const uint64_t wit_off = run.offs[4 * (size_t)s], pub_off = run.offs[4 * (size_t)s + 1], row_off = run.offs[4 * (size_t)s + 2];
    const uint32_t *rows = lag + (size_t)row_off * 8;
    for (int k = 0; k < 3; k++) {
        const SetupCsc &m = run.m[k];
        F29 acc;
#pragma unroll
        for (int j = 0; j < NL29; j++) acc.l[j] = 0;
        const uint32_t lo = m.col_ptr[c], hi = m.col_ptr[c + 1];
        for (uint32_t j = lo; j < hi; j++)
            acc = f29_reduce_4p(f29_add(acc, f29_mul(ld29(rows + (size_t)m.row[j] * 8), ld29(m.val + (size_t)j * 8))));
        uint32_t *dst;
        if (c < run.num_inst) dst = uvw + ((size_t)k * num_vars_all + pub_off + c) * 8;
        else dst = uvw + ((size_t)k * num_vars_all + num_instance_all + wit_off + (c - run.num_inst)) * 8;
        st29(dst, acc);
    }
}
// the constant one's column: grid (3 matrices, statements), the non-zeros dealt to 256 threads, a tree through LDS
__global__ __launch_bounds__(256) void setup_column0_kernel(SetupRun run, const uint32_t *__restrict__ lag, uint32_t *__restrict__ col0 /* [statements][3][8] */)
{
    __shared__ uint32_t lds[256 * NL29];
    const uint32_t k = blockIdx.x, s = blockIdx.y;
    const uint64_t row_off = run.offs[4 * (size_t)s + 2], stmt = run.offs[4 * (size_t)s + 3];
    const uint32_t *rows = lag + (size_t)row_off * 8;
    const SetupCsc &m = run.m[k];
    F29 acc;
#pragma unroll
    for (int j = 0; j < NL29; j++) acc.l[j] = 0;
    for (uint32_t j = m.col_ptr[0] + threadIdx.x; j < m.col_ptr[1]; j += 256)
        acc = f29_reduce_4p(f29_add(acc, f29_mul(ld29(rows + (size_t)m.row[j] * 8), ld29(m.val + (size_t)j * 8))));
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x >= stride && (int)threadIdx.x < 2 * stride)
            for (int j = 0; j < NL29; j++) lds[(threadIdx.x - stride) * NL29 + j] = acc.l[j];
        __syncthreads();
        if ((int)threadIdx.x < stride) {
            F29 o;
            for (int j = 0; j < NL29; j++) o.l[j] = lds[threadIdx.x * NL29 + j];
            acc = f29_reduce_4p(f29_add(acc, o));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) st29(col0 + ((size_t)stmt * 3 + k) * 8, acc);
}
// column 0 = the statements' sums; u_i += L_(C + i)(t) for the instance variables (r1cs_to_qap.rs: the input rows of A)
__global__ __launch_bounds__(256) void setup_columns_finish_kernel(uint32_t statements, uint64_t num_instance_all, uint64_t num_constraints_all, size_t num_vars_all,
                                                                   const uint32_t *__restrict__ lag, const uint32_t *__restrict__ col0, uint32_t *__restrict__ uvw)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= num_instance_all) return;
    if (i == 0) {
        for (int k = 0; k < 3; k++) {
            F29 acc;
#pragma unroll
            for (int j = 0; j < NL29; j++) acc.l[j] = 0;
            for (uint32_t s = 0; s < statements; s++) acc = f29_reduce_4p(f29_add(acc, ld29(col0 + ((size_t)s * 3 + k) * 8)));
            if (k == 0) acc = f29_reduce_4p(f29_add(acc, ld29(lag + (size_t)num_constraints_all * 8)));
            st29(uvw + (size_t)k * num_vars_all * 8, acc);
        }
        return;
    }
    uint32_t *u = uvw + (size_t)i * 8;
    st29(u, f29_reduce_4p(f29_add(ld29(u), ld29(lag + (size_t)(num_constraints_all + i) * 8))));
}

hipError_t launch_setup_columns(const SetupRun *runs, size_t num_runs, uint32_t statements, uint64_t num_instance_all, uint64_t num_constraints_all,
                                size_t num_vars_all, const uint32_t *lag, uint32_t *uvw, uint32_t *col0, hipStream_t st)
{
    for (size_t r = 0; r < num_runs; r++) {
        const SetupRun &run = runs[r];
        if (run.count == 0 || run.count > 65535) return hipErrorInvalidValue;
        hipLaunchKernelGGL(setup_columns_kernel, dim3((run.num_vars + 255) / 256, run.count), dim3(256), 0, st, run, num_instance_all, num_vars_all, lag, uvw);
        hipLaunchKernelGGL(setup_column0_kernel, dim3(3, run.count), dim3(256), 0, st, run, lag, col0);
    }
    hipLaunchKernelGGL(setup_columns_finish_kernel, dim3((unsigned)((num_instance_all + 255) / 256)), dim3(256), 0, st, statements, num_instance_all,
                       num_constraints_all, num_vars_all, lag, col0, uvw);
    return hipGetLastError();
}

// ---- the queries' scalars, canonical integers (what the fixed-base kernels take) ----------------------------------------------------
// x R' -> x: one product with the integer 1
__device__ __forceinline__ void store_plain(uint32_t *dst, const F29 &v)
{
    F29 one;
#pragma unroll
    for (int k = 0; k < NL29; k++) one.l[k] = k ? 0u : 1u;
    fr_store(dst, f29_pack(f29_canonical(f29_mul(v, one))));
}
// kind 0: in[first + i] itself (a_query: u; b_query: v); kind 1: (beta u + alpha v + w) x (ginv for an instance variable, dinv for a witness
// variable): gamma_abc / l_query
__global__ __launch_bounds__(256) void setup_var_scalars_kernel(int kind, uint64_t first, uint64_t count, uint64_t num_instance_all, size_t num_vars_all,
                                                                const uint32_t *__restrict__ uvw, int which, SetupConst alpha, SetupConst beta,
                                                                SetupConst ginv, SetupConst dinv, uint32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    const uint64_t col = first + i;
    if (kind == 0) {
        store_plain(out + (size_t)i * 8, ld29(uvw + ((size_t)which * num_vars_all + col) * 8));
        return;
    }
    const F29 u = ld29(uvw + (size_t)col * 8), v = ld29(uvw + (num_vars_all + col) * 8), w = ld29(uvw + (2 * num_vars_all + col) * 8);
    F29 x = f29_reduce_4p(f29_add(f29_mul(u, k29(beta)), f29_mul(v, k29(alpha))));
    x = f29_reduce_4p(f29_add(x, w));
    store_plain(out + (size_t)i * 8, f29_mul(x, k29(col < num_instance_all ? ginv : dinv)));
}
// h_query: (zt / delta) t^(first + i)
__global__ __launch_bounds__(256) void setup_h_scalars_kernel(uint64_t first, uint64_t count, SetupPowTab tt, SetupConst c, uint32_t *__restrict__ out)
{
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= count) return;
    store_plain(out + (size_t)i * 8, f29_mul(pow_tab(tt, first + i), k29(c)));
}

hipError_t launch_setup_var_scalars(int kind, uint64_t first, uint64_t count, uint64_t num_instance_all, size_t num_vars_all, const uint32_t *uvw, int which,
                                    const SetupConst &alpha, const SetupConst &beta, const SetupConst &ginv, const SetupConst &dinv, uint32_t *out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(setup_var_scalars_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, kind, first, count, num_instance_all, num_vars_all, uvw,
                       which, alpha, beta, ginv, dinv, out);
    return hipGetLastError();
}
hipError_t launch_setup_h_scalars(uint64_t first, uint64_t count, const SetupPowTab &tt, const SetupConst &c, uint32_t *out, hipStream_t st)
{
    if (count == 0) return hipSuccess;
    hipLaunchKernelGGL(setup_h_scalars_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, st, first, count, tt, c, out);
    return hipGetLastError();
}

// ---- diagnostics: p(t) for a polynomial in ark-ff's form (coefficient k at index k, x R, R = 2^256) and t R' -- chunks of 4,096
// coefficients by Horner's rule, the chunks' values weighted by powers from the table and added by one workgroup.  For tests that
// check h(t) zt = A(t) B(t) - C(t) or h_acc == (h(t) zt / delta) G1 on domains whose coefficients no host integer arithmetic can visit.
constexpr int EVAL_CHUNK = 4096;
__global__ __launch_bounds__(256) void poly_eval_chunks_kernel(uint64_t n, const uint32_t *__restrict__ coeffs, SetupConst t_, SetupPowTab tt, uint32_t *__restrict__ part)
{
    const uint64_t ch = (uint64_t)blockIdx.x * 256 + threadIdx.x, first = ch * EVAL_CHUNK;
    if (first >= n) return;
    const uint64_t last = first + EVAL_CHUNK < n ? first + EVAL_CHUNK : n;
    const F29 t = k29(t_);
    F29 acc;
#pragma unroll
    for (int j = 0; j < NL29; j++) acc.l[j] = 0;
    for (uint64_t k = last; k-- > first;) acc = f29_reduce_4p(f29_add(f29_mul(acc, t), ld29(coeffs + (size_t)k * 8)));    // (x R)(t R') / R' = x t R
    st29(part + (size_t)ch * 8, f29_mul(acc, pow_tab(tt, first)));
}
__global__ __launch_bounds__(256) void poly_eval_sum_kernel(uint64_t chunks, const uint32_t *__restrict__ part, uint32_t *__restrict__ out)
{
    __shared__ uint32_t lds[256 * NL29];
    F29 acc;
#pragma unroll
    for (int j = 0; j < NL29; j++) acc.l[j] = 0;
    for (uint64_t k = threadIdx.x; k < chunks; k += 256) acc = f29_reduce_4p(f29_add(acc, ld29(part + (size_t)k * 8)));
    for (int stride = 128; stride >= 1; stride >>= 1) {
        if ((int)threadIdx.x >= stride && (int)threadIdx.x < 2 * stride)
            for (int j = 0; j < NL29; j++) lds[(threadIdx.x - stride) * NL29 + j] = acc.l[j];
        __syncthreads();
        if ((int)threadIdx.x < stride) {
            F29 o;
            for (int j = 0; j < NL29; j++) o.l[j] = lds[threadIdx.x * NL29 + j];
            acc = f29_reduce_4p(f29_add(acc, o));
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) st29(out, acc);                            // p(t) R, ark-ff's form
}
hipError_t launch_poly_eval(uint64_t n, const uint32_t *coeffs, const SetupConst &t, const SetupPowTab &tt, uint32_t *part, uint32_t *out, hipStream_t st)
{
    const uint64_t chunks = (n + EVAL_CHUNK - 1) / EVAL_CHUNK;
    hipLaunchKernelGGL(poly_eval_chunks_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, n, coeffs, t, tt, part);
    hipLaunchKernelGGL(poly_eval_sum_kernel, dim3(1), dim3(256), 0, st, chunks, part, out);
    return hipGetLastError();
}
size_t poly_eval_scratch_bytes(uint64_t n) { return ((n + EVAL_CHUNK - 1) / EVAL_CHUNK) * 32; }

}  // namespace frw
