// frw_device.h -- constants and launcher prototypes shared by frw_kernels.hip and frw_capi.cpp.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

struct frw_msm;
struct frw_groth16_pk;

namespace frw {

constexpr uint32_t Q = 12289;          // falcon-rust MODULUS (gadgets/arithmetics.rs:5)
constexpr int WAVE = 64;               // CDNA wavefront
constexpr int BLOCK = 256;             // one workgroup = 4 wavefronts = one signature
constexpr int WAVES = BLOCK / WAVE;

constexpr int ST_OK = 0, ST_COEFF_RANGE = 1, ST_NORM_BOUND = 2, ST_DECODE = 3;    // == FRW_ST_* of include/frw.h

// stand-alone gadget kinds == FRW_G_* of include/frw.h
constexpr int G_LESS_THAN_Q = 0, G_MOD_Q = 1, G_ADD_MOD = 2, G_L2_ELEM = 3, G_NORM_512 = 4, G_NORM_1024 = 5;

// Per-device constant tables, built by frw_ctx_create.
struct Tables {
    uint16_t tw[1024];      // falcon-rust NTT_TABLE: 7^bitrev10(i) mod q   (misc.rs:72; script/ntt_param.sage:3-132)
    uint16_t itw[1024];     // 7^-bitrev10(i) mod q
    uint32_t ck[11][5];     // C_k = 2^k q^(k+1), 32-bit limbs (falcon_ntt.rs:31-39)
};

// FRW_ENC_COMPACT: one signature = its non-boolean witness elements as plain integers -- the 11 N that fit 32 bits as
// uint32_t (witness order: sig, v, then the b of S3, the b of S4, [prod, t, c] of S5, [r, sq] of S6), the 2 N quotients t
// of the two ntt_circuits as 5 x uint32_t little-endian limbs (S3's, then S4's) -- the 140 N + nb boolean elements as a bit
// array (witness order: S2, the boolean parts of S3, S4, S5, S6, then S7 in two words), and the 2 N instance values as
// uint32_t (without the leading one), then one status word (FRW_ST_*: a receiver of a gathered chunk has no other way to
// learn that a signature was rejected) and zero padding up to the 128-byte stride.  110 KB instead of 5.08 MB per
// Falcon-1024 signature; frw_expand_dev rebuilds the arkworks buffers (Montgomery form) from it.  Every byte of a record
// is written by the producer (the padding as zeros): the wire format is deterministic.
struct CompactLayout {
    size_t num_small, t_off, num_t, bits_off, bit_words, instance_off, num_instance, bytes;
    size_t seg_words;      // bit words of one enforce_less_than_q segment (27 N / 32)
    size_t status_off;     // the status word, right after the instance values
};
constexpr CompactLayout compact_layout(int logn)
{
    const size_t n = (size_t)1 << logn;
    CompactLayout c{};
    c.num_small = 11 * n;
    c.t_off = c.num_small * 4;
    c.num_t = 2 * n;
    c.bits_off = c.t_off + c.num_t * 20;
    c.seg_words = 27 * n / 32;
    c.bit_words = 4 * c.seg_words + n + 2;
    c.instance_off = c.bits_off + (c.bit_words * 4 + 15) / 16 * 16;
    c.num_instance = 2 * n;
    c.status_off = c.instance_off + c.num_instance * 4;
    c.bytes = (c.status_off + 4 + 127) / 128 * 128;
    return c;
}

// records the message frw_last_error() returns (thread-local) and maps the HIP error to FRW_E_HIP / FRW_E_OUT_OF_MEMORY
int record_hip_error(hipError_t e, const char *what);
void init_launch_config();
hipError_t launch_witness_ntt_verify_compact(const Tables *tab, int num_cu, int logn, size_t batch,
                                             const uint16_t *sig, const uint16_t *pk, const uint16_t *hm, void *compact,
                                             int32_t *status, hipStream_t st);
hipError_t launch_expand(int num_cu, int logn, size_t batch, const void *compact, uint64_t *wit, uint64_t *inst, hipStream_t st);
// grid / resident workgroups per CU / split flag the witness launcher would use for `batch` (diagnostics for bench.py)
void launch_shape_witness_ntt_verify(int num_cu, int logn, int enc, size_t batch, int out[4]);
hipError_t launch_witness_ntt_verify(const Tables *tab, int num_cu, int logn, int enc, size_t batch,
                                     const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                                     uint64_t *wit, uint64_t *inst, int32_t *status, hipStream_t st);
hipError_t launch_witness_dual_ntt_verify(const Tables *tab, int num_cu, int logn, int enc,
                                          size_t batch, const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                                          uint64_t *wit, uint64_t *inst, int32_t *status, hipStream_t st);
hipError_t launch_ntt_modq(const Tables *tab, int num_cu, int logn, int enc, size_t batch, const uint16_t *poly,
                           uint64_t *wit, uint16_t *ntt_out, int32_t *status, hipStream_t st);
hipError_t launch_gadget(int kind, int enc, size_t count, const void *a, const uint64_t *b, uint64_t *out,
                         int32_t *status, hipStream_t st);
hipError_t launch_hash_to_point(int logn, size_t batch, const uint8_t *nonces, const uint8_t *msgs, const uint64_t *msg_off,
                                uint16_t *hm, hipStream_t st);
hipError_t launch_decode_public_keys(int logn, size_t batch, const uint8_t *pk_bytes, uint16_t *pk, int32_t *status, hipStream_t st);
hipError_t launch_decode_signatures(int logn, size_t batch, const uint8_t *sig_bytes, size_t sig_len, uint16_t *sig,
                                    uint8_t *nonce_out, int32_t *status, hipStream_t st);
// R1CS matrices resident on the device (frw_r1cs_load): CSR, coefficients in Montgomery form (8 x u32)
struct R1csMatrixDev {
    const uint64_t *row_ptr;    // num_constraints + 1
    const uint32_t *col;        // nnz; < num_instance: instance variable (0 = the constant one), else witness
    const uint32_t *val;        // nnz x 8: c R, Montgomery form (the check-only kernel)
    const uint32_t *col_class;  // nnz: column | class << 30 (0 general, 1 coefficient +1, 2 coefficient -1)
    const uint32_t *val29;      // nnz x 8: c R' (R' = 2^261) packed, for f29_mul(z R, c R') = z c R
};
// Rows of at least R1CS_LONG_ROW terms (the 2 N un-reduced NTT outputs of the Falcon circuits are linear combinations
// of all N coefficients once arkworks' finalize() has inlined them) are not walked by one thread: a wavefront takes the
// row, 64 terms at a time.  Their coefficients are stored per chunk of 64 terms as c R' (R' = 2^261) in nine 29-bit
// limbs, one plane per limb (frw_fr29.h), padded with zero coefficients.
constexpr uint32_t R1CS_LONG_ROW = 128;
struct R1csLongRow { uint32_t matrix, row, first_chunk, num_chunks; };
struct R1csAgg;
struct R1csDev {
    uint32_t num_instance, num_witness, num_constraints;
    // host side only: non-null for an aggregate statement (frw_r1cs_load_aggregate), whose matrices are the block-diagonal
    // arrangement of the per-signature systems' and are never materialised -- see R1csAgg below; everything else is then unset
    const R1csAgg *agg;
    R1csMatrixDev a, b, c;
    const uint32_t *order;      // constraint rows by decreasing length
    uint32_t num_long;
    const R1csLongRow *long_rows;
    const uint32_t *long_col;   // [chunks][64]
    const uint32_t *long_coef;  // [chunks][9][64]
    const uint8_t *long_mask;   // [num_constraints]: bit m set = row is long in matrix m (0 = A, 1 = B, 2 = C)
    const uint32_t *long_slot;  // [3][num_constraints]: index into long_rows of (matrix, row), where long
    // the distinct variables the long rows read (the N signature coefficients, the N coefficients of v, ... -- small
    // integers in a Falcon witness): their plain values are extracted once per signature, and a long-row term whose
    // variable is below 2^28 is one multiply-add per limb instead of a field product
    uint32_t num_long_vars;
    const uint32_t *long_vars;  // [num_long_vars] column
    const uint32_t *long_cidx;  // [chunks][64] index into long_vars
    uint32_t k_rrp[9];          // R R' mod p (R = 2^256, R' = 2^261) as an integer, 29-bit limbs
    // The short rows once more, flattened for r1cs_eval_flat_kernel: position i (rows by decreasing length, as `order`) has ONE
    // header -- the row, how many terms it has in A, B, C (0 where it is long), the long mask -- and its terms of all three matrices
    // one after the other, a word each: column | coefficient index << 24 (0: +1, 1: -1, else into flat_coef, c R' packed like
    // val29: a Falcon circuit has 16 distinct coefficients).  A thread then needs three dependent loads for a row (header, terms,
    // variables) where the CSR walk needed a dozen -- and a third of the memory instructions, which is what the kernel waits for.  flat_term is null when a circuit does not fit (> 254 coefficients, a column
    // >= 2^24): r1cs_eval_kernel does the work then.
    const uint32_t *flat_head;  // [num_constraints][4]: row | (nA | nB << 8 | nC << 16 | long mask << 24) | first term / 4 | 0
    const uint32_t *flat_term;  // a row's terms start at a multiple of four words: a thread fetches them four at a time, as one 16-byte load
    const uint32_t *flat_coef;  // [flat_num_coefs][8]
    uint32_t flat_num_coefs;
};
constexpr uint32_t R1CS_FLAT_COEFS = 256;
// Where one launch of the evaluation kernels finds the statements ("signatures") of its batch and where their products go,
// in 32-bit words unless said otherwise.  The plain batch of the witness entry points and a run of statements inside an
// aggregate assignment differ in nothing else:
//   variable col >= I of statement s   wit + s wit_stride + 8 (col - I)
//   instance variable 1 <= col < I     inst + s inst_stride + 8 col
//   instance variable 0 (the one)      one + s one_stride                  (an aggregate has ONE constant for all statements)
//   (M z)_row, M = A, B, C             abc + 8 (s abc_sig_stride + m abc_mat_stride + row)
//   violated rows of statement s       added to flags[s flag_stride]       (an aggregate counts into one word)
struct R1csView {
    const uint32_t *wit;  size_t wit_stride;
    const uint32_t *inst; size_t inst_stride;
    const uint32_t *one;  size_t one_stride;
    uint32_t *abc;        size_t abc_sig_stride, abc_mat_stride;
    unsigned int *flags;  size_t flag_stride;
    // Statements that are NOT evenly spaced -- all the statements of one parameter set inside a MIXED aggregate, in ONE launch (round 5:
    // the 1,024-statement mix of BASELINE configs[4] is 516 runs of two statements; a launch sequence per run was 1,548 launches of 70
    // microseconds at the head of every proof): offs[3 s ..] = the elements before statement s's witness variables, before its public
    // inputs (its instance variable j >= 1 at inst + 8 (offs + j)) and the rows before its first row, in device memory; the strides
    // above are then unused (but for the matrices' stride).
    const uint64_t *offs;
};
// An aggregate statement: FalconNTTVerificationCircuit::generate_constraints run once per statement on ONE constraint system
// (host/frw_host.hpp FalconAggregateVerificationCircuit).  Its instance vector is [1, public inputs of statement 0, of
// statement 1, ...], its witness vector the concatenation of the statements' witness vectors, its constraint rows theirs in
// order -- so its matrices are block diagonal but for column 0, and A z, B z, C z are the statements' own products laid end
// to end: a run of consecutive statements of one parameter set is one launch of that set's kernels through an R1csView.
struct R1csAggRun {
    const R1csDev *base;        // the per-signature system of the run's parameter set
    uint32_t first, count;      // statements [first, first + count)
    uint64_t wit_off;           // elements before the run's first witness variable in the aggregate witness vector
    uint64_t pub_off;           // public inputs before the run's first one (its instance variable j >= 1 is aggregate variable pub_off + j)
    uint64_t row_off;           // constraint rows before the run's first one
};
// all the statements of one parameter set, wherever they stand in the aggregate: one launch of that set's kernels
struct R1csAggSet {
    const R1csDev *base;
    uint32_t count;
    const uint64_t *offs;       // device memory, [count][3]: R1csView::offs
};
struct R1csAgg {
    uint32_t num_statements;
    uint32_t num_runs;
    const R1csAggRun *runs;     // host memory
    R1csAggSet set[2];          // Falcon-512, Falcon-1024 (count 0: none)
};
size_t r1cs_check_scratch_bytes(const R1csDev &r, size_t batch, bool with_abc);
hipError_t launch_r1cs_check(const R1csDev &r, size_t batch, const uint64_t *witness, const uint64_t *instance,
                             uint32_t *num_unsatisfied, uint64_t *abc, hipStream_t st, void *scratch = nullptr);
// Tables of the QAP witness map's domain (frw_qap.hip), built by frw_r1cs_load.  Every entry is a field element times
// R' = 2^261 (frw_fr29.h): the per-index factors packed in 8 x 32 bits (they are < p), the 64-th roots in nine limbs.
constexpr int QAP_MAX_PASSES = 5;          // domains up to 2^30
constexpr int QAP_MIN_LOG_N = 14, QAP_MAX_LOG_N = 30;
// passes of a 2^L-point transform, lowest bits first; returns the number of passes (0: no schedule for this L)
inline int qap_pass_schedule(int L, int t[QAP_MAX_PASSES], int sh[QAP_MAX_PASSES])
{
    if (L < QAP_MIN_LOG_N || L > QAP_MAX_LOG_N) return 0;
    const int k = (L + 5) / 6;
    int deficit = 6 * k - L;                       // taken off the upper passes, the top one first, at most two each
    for (int i = 0; i < k; i++) t[i] = 6;
    for (int round = 0; round < 2 && deficit; round++)
        for (int i = k - 1; i >= 1 && deficit; i--) { t[i]--; deficit--; }
    if (deficit) return 0;
    for (int i = 0, s = 0; i < k; i++) { sh[i] = s; s += t[i]; }
    return k;
}
struct QapDev {
    int log_n;                    // domain = the 2^log_n-th roots of unity, 2^log_n >= num_constraints + num_instance
    const uint32_t *roots_fwd;    // w^(k n/64), k < 32, 12 words apart
    const uint32_t *roots_inv;    // w^-(k n/64)
    // The pass schedule: pass k works on the index bits [pass_sh[k], pass_sh[k] + pass_t[k]); pass 0 always takes the six lowest
    // bits, the others four to six each (qap_pass_schedule: 17 = 6 + 6 + 5, 18 = 6 + 6 + 6, 19 = 6 + 5 + 4 + 4, 20 = 6 + 5 + 5 + 4,
    // 22 = 6 + 6 + 5 + 5, 24 = 6 + 6 + 6 + 6, ...)
    int num_passes;
    int pass_t[QAP_MAX_PASSES], pass_sh[QAP_MAX_PASSES];
    const uint32_t *twist_fwd[QAP_MAX_PASSES - 1]; // [n][8]: what the forward pass k + 1 multiplies index i by on its way out
    const uint32_t *twist_inv[QAP_MAX_PASSES - 1]; // [n][8]: what the inverse pass k multiplies index i by for the NEXT pass, k + 1
    const uint32_t *scale_in;     // g^k / n                  (ifft's 1/n and coset_fft's distribute_powers, fused)
    const uint32_t *scale_in_a;   // 2^5 g^k / n              (for A z: the a b product divides by 2^261, the data carry 2^256)
    const uint32_t *scale_out;    // g^-k / (n (g^n - 1))     (ifft's 1/n, division by the vanishing polynomial, g^-k)
    // the six-transform quotient (launch_qap_quotient): psi = the 2n-th root of unity with psi^2 = w
    const uint32_t *scale_psi_in; // psi^k / n
    const uint32_t *scale_psi_out;// -16 psi^-k / n           (-1/2, and 32 for the data x data product in R' arithmetic)
    uint32_t sixteen_over_n[9];   // 16 / n, x R' in nine limbs
};
size_t qap_workspace_bytes_per_signature(const R1csDev &r, const QapDev &q);
// ---- setup on the device (frw_setup.hip): transform tables, the QAP at the toxic point, the queries' scalars ------------------------
// a field constant x R' (R' = 2^261) in nine 29-bit limbs, as a kernel argument
struct SetupConst { uint32_t l[9]; };
// base^e for any e below the domain size from two small tables: lo[k] = base^k (k < 2^14), hi[k] = base^(k 2^14) (k < 2^(L - 14)); x R' packed
constexpr int SETUP_POW_LO_BITS = 14;
constexpr uint64_t SETUP_POW_LO = (uint64_t)1 << SETUP_POW_LO_BITS;
struct SetupPowTab { const uint32_t *lo, *hi; };
// a per-signature matrix by columns: the non-zeros of variable c are [col_ptr[c], col_ptr[c + 1]): their rows and coefficients (c R' packed)
struct SetupCsc { const uint32_t *col_ptr, *row, *val; };
// ALL the statements of one parameter set inside an aggregate, wherever they stand (or a single circuit: one statement at zero)
struct SetupRun {
    SetupCsc m[3];
    uint32_t num_vars, num_inst, num_constraints;    // of the per-signature system: I + W, I (with the constant one), C
    uint32_t count;                                  // statements of this launch
    // [count][4], device memory: before statement s, the aggregate's witness variables, its public inputs (the statement's instance
    // variable j >= 1 is aggregate variable pub + j), its constraint rows; and the statement's index in the aggregate
    const uint64_t *offs;
};
hipError_t launch_qap_table(uint64_t n, const SetupPowTab &t, int mode, int sh, int ts, int L, const SetupConst *first, uint32_t *out, hipStream_t st);
hipError_t launch_setup_lagrange(uint64_t n, const SetupPowTab &wt, const SetupConst &t, const SetupConst &c, const SetupConst &one, uint32_t *lag, hipStream_t st);
hipError_t launch_setup_columns(const SetupRun *runs, size_t num_runs, uint32_t statements, uint64_t num_instance_all, uint64_t num_constraints_all,
                                size_t num_vars_all, const uint32_t *lag, uint32_t *uvw, uint32_t *col0, hipStream_t st);
hipError_t launch_setup_var_scalars(int kind, uint64_t first, uint64_t count, uint64_t num_instance_all, size_t num_vars_all, const uint32_t *uvw, int which,
                                    const SetupConst &alpha, const SetupConst &beta, const SetupConst &ginv, const SetupConst &dinv, uint32_t *out, hipStream_t st);
hipError_t launch_setup_h_scalars(uint64_t first, uint64_t count, const SetupPowTab &tt, const SetupConst &c, uint32_t *out, hipStream_t st);
// the multi-scalar multiplication side of a device-made key (frw_msm.hip): bare handles filled in place by fixed-base kernels
struct FixedBaseGen { uint32_t *g1, *g2; };        // [32][256] multiples of the published generators, as table rows
int fixed_base_gen_create(int device, FixedBaseGen *g);
void fixed_base_gen_free(FixedBaseGen *g);
int msm_alloc_bare(int device, int group /* 1: G1, 2: G2 */, int window_bits, size_t rows, uint64_t row_lo, ::frw_msm **out);
int msm_expand_tables(::frw_msm **m);       // a filled bare handle -> window tables over the same points (frees the bare one)
hipError_t msm_fill_fixed_base(::frw_msm *m, const FixedBaseGen &g, size_t first_row, size_t count, const uint32_t *d_scalars, hipStream_t st);
hipError_t fixed_base_ark_dev(const FixedBaseGen &g, int group, size_t count, const uint32_t *d_scalars, uint32_t *d_out, hipStream_t st);
void groth16_shard_range(uint64_t total, uint32_t rank, uint32_t world, uint64_t *lo, uint64_t *hi);
int groth16_pk_assemble(int device, uint64_t ni, uint64_t nw, uint64_t n, uint32_t rank, uint32_t world, ::frw_msm *h, ::frw_msm *a, ::frw_msm *b1,
                        ::frw_msm *l, ::frw_msm *b2, ::frw_groth16_pk **out);
hipError_t launch_poly_eval(uint64_t n, const uint32_t *coeffs, const SetupConst &t, const SetupPowTab &tt, uint32_t *part, uint32_t *out, hipStream_t st);
size_t poly_eval_scratch_bytes(uint64_t n);
hipError_t diag_valu_rates(int num_cu, void *scratch, double out[4], hipStream_t st);
hipError_t launch_qap_quotient(const R1csDev &r, const QapDev &q, size_t batch, const uint64_t *witness,
                               const uint64_t *instance, uint64_t *h, uint32_t *num_unsatisfied, void *workspace,
                               size_t workspace_bytes, hipStream_t st);
hipError_t launch_qap_witness_map(const R1csDev &r, const QapDev &q, size_t batch, const uint64_t *witness,
                                  const uint64_t *instance, uint64_t *h, uint32_t *num_unsatisfied, void *workspace,
                                  size_t workspace_bytes, hipStream_t st);
hipError_t launch_write_stream(void *buf, size_t bytes, size_t slab_bytes, int num_cu, hipStream_t st);
hipError_t launch_digest(const uint64_t *buf, size_t words, size_t items, uint64_t *out, hipStream_t st);

}  // namespace frw
