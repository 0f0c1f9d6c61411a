/*
 * frw.h -- C ABI of the MI355X Falcon-verification R1CS witness engine (libfrw.so).
 *
 * This is the drop-in boundary for the reference's hot path.  The reference
 * (zhenfeizhang/falcon-r1cs, all citations relative to its root) has no FFI of its own; the
 * path is the Rust generic API
 *     FalconNTTVerificationCircuit::generate_constraints   falcon-r1cs/src/circuits/falcon_ntt.rs:26-123
 *     NTTPolyVar::ntt_circuit                              falcon-r1cs/src/gadgets/poly.rs:104-159
 *     mod_q / add_mod                                      falcon-r1cs/src/gadgets/arithmetics.rs:105-149, :214-262
 *     enforce_less_than_q / is_less_than_6144 / enforce_less_than_norm_bound
 *                                                          falcon-r1cs/src/gadgets/range_proofs.rs:42-94, :289-333, :274-284
 *     l2_norm_var / enforce_decompose                      falcon-r1cs/src/gadgets/misc.rs:30-51, :9-24
 * whose only observable product is the pair of assignment vectors of the arkworks
 * ConstraintSystem (witness_assignment, instance_assignment).  Because the circuit is the same
 * for every signature of one parameter set, the replacement is a batch witness filler: a Rust
 * host keeps allocating variables and emitting constraints exactly as today and takes the
 * VALUES of all W witnesses and I instance variables of `batch` signatures from one call
 * (binding shown in INTEGRATION.md).
 *
 * Conventions
 *   - plain C types only; all buffers are caller-owned; no callbacks.
 *   - `logn` is 9 (Falcon-512) or 10 (Falcon-1024); the reference selects it with a cargo
 *     feature (falcon-r1cs/Cargo.toml:28-32), here it is a run-time argument.
 *   - polynomials are `uint16_t[batch][N]`, row-major, coefficients in [0, q), q = 12289
 *     (what falcon-rust's Polynomial::coeff() returns; falcon_ntt.rs:27-28,44).
 *   - a field element is 4 x uint64_t little-endian limbs (ark-ff Fp256 over the BLS12-381
 *     scalar field).  encoding FRW_ENC_MONTGOMERY stores x*2^256 mod p -- byte-identical to
 *     what arkworks keeps in witness_assignment; FRW_ENC_CANONICAL stores x itself.  The reference is generic
 *     over F: PrimeField but instantiates it with this one field everywhere (poly.rs:244, examples/pok_sig.rs:3);
 *     every witness value is an integer below 2^160, so FRW_ENC_CANONICAL and FRW_ENC_COMPACT are field-agnostic
 *     (a host on another >= 161-bit field converts them itself), only FRW_ENC_MONTGOMERY is BLS12-381-specific.
 *   - witness:  uint64_t[batch][W][4] in arkworks allocation order (layout: frw_layout()).
 *     instance: uint64_t[batch][I][4], I = 2N+1: [1, pk_ntt[0..N), hm_ntt[0..N)]
 *     (falcon_ntt.rs:63,67; public-input order as in examples/pok_sig.rs:38-45).
 *   - status:   int32_t[batch]: FRW_ST_OK, FRW_ST_COEFF_RANGE (an input coefficient >= q; that signature's
 *     witness and instance slots are zero-filled), FRW_ST_NORM_BOUND (l2 norm >= SIG_L2_BOUND; the witness IS written,
 *     with the truncated bit decomposition the reference assigns when its `#[cfg(not(test))]`
 *     panic is compiled out (range_proofs.rs:112-117,203-208), so the system is unsatisfied).
 *   - every function returns FRW_OK (0) or a negative FRW_E_* code; frw_strerror() names it.
 *     There is NO CPU fallback: without a usable HIP device every compute entry point fails
 *     with FRW_E_NO_DEVICE.
 *   - a context is bound to one HIP device and may be used by one host thread at a time;
 *     independent contexts may be used concurrently (the reference's ConstraintSystemRef is an
 *     Rc<RefCell<..>>, i.e. one circuit per thread as well).
 */
#ifndef FRW_H
#define FRW_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FRW_OK               0
#define FRW_E_INVALID_ARG   -1   /* bad logn / encoding / null pointer */
#define FRW_E_NO_DEVICE     -2   /* no HIP device, or device ordinal out of range */
#define FRW_E_HIP           -3   /* a HIP runtime call failed; see frw_last_error() */
#define FRW_E_OUT_OF_MEMORY -4
#define FRW_E_RANGE         -5   /* strict mode: at least one signature has status != FRW_ST_OK (the reference panics) */

#define FRW_ENC_CANONICAL   0
#define FRW_ENC_MONTGOMERY  1

#define FRW_ST_OK           0
#define FRW_ST_COEFF_RANGE  1
#define FRW_ST_NORM_BOUND   2
#define FRW_ST_DECODE       3   /* input preparation: malformed public key / signature encoding */

#define FRW_NUM_SEGMENTS    8

/* Witness layout = allocation order of falcon_ntt.rs:58-122, in units of field elements. */
typedef struct frw_layout {
    int32_t logn;
    int32_t n;                 /* N = 1 << logn */
    int32_t num_witness;       /* W = 153 N + {50 | 52}        (README.md:44,55: 156,724 / 78,386) */
    int32_t num_instance;      /* I = 2 N + 1                  (README.md:44,55: 2,049 / 1,025)   */
    int32_t num_constraints;   /* C = 159 N + {52 | 54}        (README.md:44,55: 162,870 / 81,460) */
    /* S0 sig[i]                          falcon_ntt.rs:58-59       N
     * S1 v[i]                            falcon_ntt.rs:71          N
     * S2 enforce_less_than_q(v[i])       falcon_ntt.rs:73-77       27 N
     * S3 mod_q blocks of ntt_circuit(sig) falcon_ntt.rs:88-89      29 N
     * S4 mod_q blocks of ntt_circuit(v)  falcon_ntt.rs:90-91       29 N
     * S5 pointwise product + add_mod     falcon_ntt.rs:94-111      30 N
     * S6 l2_norm_var over v || sig       falcon_ntt.rs:116-120     18 * 2N
     * S7 enforce_less_than_norm_bound    falcon_ntt.rs:122         50 | 52 */
    int32_t seg_off[FRW_NUM_SEGMENTS];
    int32_t seg_len[FRW_NUM_SEGMENTS];
} frw_layout_t;

typedef struct frw_ctx frw_ctx;

/* ---- structure --------------------------------------------------------------------------- */
int frw_layout(int logn, frw_layout_t *out);
const char *frw_strerror(int code);
/* text of the last failing HIP call on this thread ("" if none) */
const char *frw_last_error(void);

/* ---- device context ---------------------------------------------------------------------- */
int frw_device_count(void);
/* device = HIP ordinal >= 0.  Builds the twiddle / offset tables on that device. */
int frw_ctx_create(int device, frw_ctx **out);
void frw_ctx_destroy(frw_ctx *ctx);

/* ---- hot path, device-resident buffers (hipStream_t passed as void*; NULL = default stream) ----
 * Replaces one generate_constraints() call per signature (falcon_ntt.rs:26-123): fills the
 * witness and instance assignment of `batch` signatures.  All pointers are device pointers.
 * Asynchronous w.r.t. the host: returns after enqueueing on `stream`.
 * encoding FRW_ENC_COMPACT (see below): d_witness is the compact buffer, d_instance is ignored. */
int frw_witness_ntt_verify_dev(frw_ctx *ctx, int logn, size_t batch,
                               const uint16_t *d_sig, const uint16_t *d_pk, const uint16_t *d_hm,
                               int encoding, uint64_t *d_witness, uint64_t *d_instance,
                               int32_t *d_status, void *stream);

/* Replaces NTTPolyVar::ntt_circuit alone (poly.rs:104-159; the "ntt conversion" row of
 * examples/constraint_counts.rs:74-113): d_witness = uint64_t[batch][29 N][4] (the N mod_q
 * blocks [t, b, ltq(b)]), d_ntt_out = uint16_t[batch][N] (the b values == NTTPolynomial::from).
 * d_status: FRW_ST_OK or FRW_ST_COEFF_RANGE. */
int frw_ntt_modq_dev(frw_ctx *ctx, int logn, size_t batch, const uint16_t *d_poly, int encoding,
                     uint64_t *d_witness, uint16_t *d_ntt_out, int32_t *d_status, void *stream);

/* ---- hot path, host buffers ----------------------------------------------------------------
 * Same results through host memory (pageable, or page-locked from frw_host_alloc: faster): H2D of the inputs, the
 * kernels, D2H of the outputs, chunked so that any batch fits the device.  This is the call the reference's own
 * consumers make -- one signature per generate_constraints (examples/constraint_counts.rs:61-63, pok_sig.rs:24-32) --
 * so it is built for batch = 1 as much as for 10^5: device buffers, a page-locked staging buffer, streams and events
 * belong to the context and only grow (a call after the first allocates nothing; frw_diag_host_allocations counts,
 * frw_ctx_trim gives the memory back), and a batch that fits one chunk is one copy in, one launch, the copies out and
 * one synchronisation.  Calls on one context are serialised (a mutex); use one context per host thread for concurrency.
 * strict != 0 mirrors the reference's non-test build: returns FRW_E_RANGE if any status != FRW_ST_OK (outputs of those
 * signatures must not be used).  strict == 0 mirrors its cfg(test) build (see FRW_ST_NORM_BOUND above).
 * encoding FRW_ENC_COMPACT: `witness` receives batch x bytes_per_signature compact bytes, `instance` may be NULL. */
int frw_witness_ntt_verify(frw_ctx *ctx, int logn, size_t batch,
                           const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                           int encoding, uint64_t *witness, uint64_t *instance,
                           int32_t *status, int strict);

int frw_ntt_modq(frw_ctx *ctx, int logn, size_t batch, const uint16_t *poly, int encoding,
                 uint64_t *witness, uint16_t *ntt_out, int32_t *status);
/* number of device / page-locked allocations the host-buffer entry points of this context have made so far (it stops
 * growing once the largest batch shape has been seen), and release of everything they hold */
int frw_diag_host_allocations(frw_ctx *ctx, uint64_t *count);
int frw_ctx_trim(frw_ctx *ctx);

/* ---- compact encoding (no counterpart in the reference; for GPU-side consumers, the multi-GPU gather and PCIe) ----
 * 140 N + nb of the W witness elements of falcon_ntt.rs:58-122 are booleans, and all but 2 N of the others are integers
 * below 2^28.  FRW_ENC_COMPACT keeps, per signature (frw_compact_layout), the VALUES as plain integers:
 *   small     11 N x uint32_t   the non-boolean witness elements that fit 32 bits, in witness order:
 *                               sig[N], v[N], b of ntt_circuit(sig) [N], b of ntt_circuit(v) [N],
 *                               [prod, t, c] of the pointwise add_mod [3 N], [r, sq] of the 2 N l2-norm elements [4 N]
 *   t         2 N x 5 uint32_t  the mod_q quotients t = floor(a / q) of ntt_circuit(sig), then of ntt_circuit(v)
 *                               (132 / 146 bits; little-endian limbs)
 *   bits      uint32_t words    the boolean elements as a bit array in witness order (bit i of the array = bit i%32 of
 *                               word i/32): enforce_less_than_q(v[i]) 27 N bits, then the 27-bit enforce_less_than_q
 *                               blocks of the S3, S4, S5 segments (27 N each), the 16 booleans of every l2-norm element
 *                               (32 N), and the norm-bound block (50 | 52 bits, in two words of their own)
 *   instance  2 N x uint32_t    pk_ntt, hm_ntt (the leading constant one is implied)
 *   status    uint32_t          FRW_ST_* of this signature, then zeros up to the stride: a record tells its receiver (the
 *                               other end of an all-gather, say) that the signature was rejected
 * = 112,256 bytes per Falcon-1024 signature instead of 5,080,736.  frw_expand_dev / frw_expand_host rebuild, bit for bit,
 * the buffers frw_witness_ntt_verify_dev(..., FRW_ENC_MONTGOMERY, ...) writes (x -> x * 2^256 mod p for the values, 0 / the
 * Montgomery form of 1 for the booleans); a Rust host can equally build its Vec<Fr> with Fr::from(u64) / from limbs.
 * A signature with FRW_ST_COEFF_RANGE is all zeros but for its status word, and expands to what the direct entry point
 * leaves for it: zeros in the witness and in the instance vector (no leading one).  The producer writes every byte of a
 * record (padding as zeros), so equal inputs give equal bytes. */
#define FRW_ENC_COMPACT     2
typedef struct frw_compact_layout {
    int32_t logn, n;
    uint64_t bytes_per_signature;     /* stride of the compact buffer, a multiple of 128 */
    uint64_t small_off, num_small;    /* byte offset (0) and number (11 N) of the uint32_t values */
    uint64_t t_off, num_t;            /* byte offset and number (2 N) of the 5 x uint32_t quotients */
    uint64_t bits_off, num_bit_words; /* byte offset and number of uint32_t words of the bit array */
    uint64_t bit_seg_off[6];          /* first bit of S2, S3, S4, S5, S6, S7 booleans inside the bit array */
    uint64_t instance_off, num_instance_values;
    uint64_t status_off;              /* byte offset of the status word (= instance_off + 4 x num_instance_values) */
} frw_compact_layout_t;
int frw_compact_layout(int logn, frw_compact_layout_t *out);
/* d_compact: batch x bytes_per_signature bytes, 16-byte aligned.  Same statuses as frw_witness_ntt_verify_dev. */
int frw_witness_ntt_verify_compact_dev(frw_ctx *ctx, int logn, size_t batch,
                                       const uint16_t *d_sig, const uint16_t *d_pk, const uint16_t *d_hm,
                                       void *d_compact, int32_t *d_status, void *stream);
/* compact -> witness uint64_t[batch][W][4], instance uint64_t[batch][I][4] (FRW_ENC_MONTGOMERY bytes) */
int frw_expand_dev(frw_ctx *ctx, int logn, size_t batch, const void *d_compact,
                   uint64_t *d_witness, uint64_t *d_instance, void *stream);
/* The same expansion in host memory (no device; a format conversion: integers to Montgomery form, bits to 0 / 1 elements),
 * for a host that took the compact form over PCIe: frw_witness_ntt_verify(..., FRW_ENC_COMPACT, compact, NULL, status,
 * strict) moves 0.11 MB per Falcon-1024 signature instead of 5.08 MB.  Signatures are independent: callers may split
 * `batch` over threads. */
int frw_expand_host(int logn, size_t batch, const void *compact, uint64_t *witness, uint64_t *instance);

/* ---- the signed-split variant: FalconDualNTTVerificationCircuit (circuits/falcon_dual_ntt.rs:26-132) -----------
 * Same statement, signature and v split into non-negative (pos, neg) parts (gadgets/dual_poly.rs:15-31), four
 * ntt_circuits, two mod_q per NTT coefficient, squares without range checks (gadgets/misc.rs:55-65).
 * W = 186 N + 4 + {50|52}, I = 2 N + 1, C = 189 N + 10 + {50|52}.  Segment order (allocation order):
 *   0 sig.pos N | 1 sig.neg N | 2 pos*neg products N | 3 is_zero [is_not_equal, multiplier] 2
 *   4 v.pos N | 5 v.neg N | 6 products N | 7 is_zero 2
 *   8..11 mod_q blocks of ntt_circuit(sig.pos), (sig.neg), (v.pos), (v.neg), 29 N each
 *   12 per coefficient [sig_ntt.neg*pk_ntt, t, b, ltq(b)] [sig_ntt.pos*pk_ntt, t, b, ltq(b)], 60 N
 *   13 squares of v.pos, v.neg, sig.pos, sig.neg, 4 N | 14 norm bound
 * The signed lift uses the threshold q/2 = 6144 (falcon-rust's DualPolynomial; un-vendored). */
#define FRW_NUM_SEGMENTS_DUAL 15
typedef struct frw_layout_dual {
    int32_t logn, n, num_witness, num_instance, num_constraints;
    int32_t seg_off[FRW_NUM_SEGMENTS_DUAL];
    int32_t seg_len[FRW_NUM_SEGMENTS_DUAL];
} frw_layout_dual_t;

int frw_layout_dual(int logn, frw_layout_dual_t *out);
int frw_witness_dual_ntt_verify_dev(frw_ctx *ctx, int logn, size_t batch,
                                    const uint16_t *d_sig, const uint16_t *d_pk, const uint16_t *d_hm,
                                    int encoding, uint64_t *d_witness, uint64_t *d_instance,
                                    int32_t *d_status, void *stream);
int frw_witness_dual_ntt_verify(frw_ctx *ctx, int logn, size_t batch,
                                const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                                int encoding, uint64_t *witness, uint64_t *instance,
                                int32_t *status, int strict);

/* ---- R1CS matrix export (structure only; host, no GPU) --------------------------------------------------------
 * What a prover ingests after the hot path (examples/pok_sig.rs:30-32: Groth16 setup/prove call cs.to_matrices()):
 * A, B, C of the chosen circuit with every symbolic linear combination inlined.  Column j < I is instance variable
 * j (column 0 = the constant one), column I + k is witness k -- the same order as the witness/instance buffers
 * above.  File: "FRWR1CS1", u64 {I, W, C, nnz_A, nnz_B, nnz_C}, then per matrix CSR: u64 row_ptr[C+1],
 * u32 col[nnz], u64 value[nnz][4] (canonical little-endian).  counts (optional) receives the six header words. */
#define FRW_CIRCUIT_NTT       0   /* FalconNTTVerificationCircuit      circuits/falcon_ntt.rs      */
#define FRW_CIRCUIT_DUAL_NTT  1   /* FalconDualNTTVerificationCircuit  circuits/falcon_dual_ntt.rs */
int frw_r1cs_export(int circuit, int logn, const char *path, uint64_t *counts);

/* Batch satisfaction check on the device: the reference's `assert!(cs.is_satisfied())` (falcon_ntt.rs:159) for every
 * signature of an HBM-resident batch, against the matrices above (emitted from the gadget definitions, independently
 * of the witness kernels' closed form).  frw_r1cs_load builds them on the host (seconds) and uploads them once;
 * d_witness / d_instance are the buffers of the witness entry points with encoding FRW_ENC_MONTGOMERY;
 * d_num_unsatisfied[i] = number of constraint rows signature i violates (0 = satisfied).
 * (Diagnostics: with FRW_R1CS_NO_FLAT set in the environment at load time the handle's short rows are evaluated by the CSR walk
 * instead of from their flattened form -- the route of a circuit with more than 254 distinct coefficients; the tests run both.) */
typedef struct frw_r1cs frw_r1cs;
int frw_r1cs_load(int device, int circuit, int logn, frw_r1cs **out);
void frw_r1cs_free(frw_r1cs *r);
int frw_r1cs_check_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                       uint32_t *d_num_unsatisfied, void *stream);
/* Same pass, additionally writing the three matrix-vector products a prover's QAP witness map starts from (what
 * ark-groth16 computes on the CPU right after generate_constraints, examples/pok_sig.rs:32):
 * d_abc = uint64_t[batch][3][C][4] = A z, B z, C z per signature, Montgomery form, rows in constraint order.
 * frw_r1cs_check_dev and frw_r1cs_eval_dev take a stream-ordered scratch allocation (hipMallocAsync / hipFreeAsync on
 * `stream`) per call -- they are NOT stream-capture safe -- and run slower kernels should that allocation fail.  The
 * _scratch_ variant below is the same computation with the caller's scratch: no allocation, one fixed kernel sequence. */
int frw_r1cs_eval_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                      uint32_t *d_num_unsatisfied, uint64_t *d_abc, void *stream);
/* d_scratch: at least frw_r1cs_eval_scratch_bytes(r, batch, d_abc != NULL) bytes, 16-byte aligned (may be NULL when
 * that is 0).  d_abc may be NULL (check only). */
size_t frw_r1cs_eval_scratch_bytes(const frw_r1cs *r, size_t batch, int with_products);
int frw_r1cs_eval_scratch_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                              uint32_t *d_num_unsatisfied, uint64_t *d_abc, void *d_scratch, size_t scratch_bytes,
                              void *stream);

/* ---- QAP witness map (the step after the hot path in a Groth16 prover) ------------------------------------------
 * examples/pok_sig.rs:30-47 hands the circuit to Groth16::<Bls12_381>::prove; after generate_constraints the prover's
 * first step is ark-groth16 0.3.0's R1CStoQAP::witness_map (r1cs_to_qap.rs): the coefficients of
 *     h(X) = (A(X) B(X) - C(X)) / (X^n - 1)
 * over ark-poly's Radix2EvaluationDomain of size n = next_power_of_two(C + I) (2^17 for Falcon-512, 2^18 for
 * Falcon-1024), computed as  a = A z ++ instance values, b = B z, c = C z;  ifft, coset_fft (generator 7);
 * a o b - c;  / (7^n - 1);  coset_ifft.  frw_qap_witness_map_dev returns exactly that h for every signature of an HBM-resident
 * batch, whatever the witness -- by running the six-transform quotient below for all of them (the same h wherever the witness
 * satisfies the system) and then ark-groth16's seven transforms, as written, for just the signatures whose witness does not
 * (normally none: those launches find an empty list) -- from the buffers the witness entry points wrote (FRW_ENC_MONTGOMERY):
 *     d_h               uint64_t[batch][n][4], Montgomery form, coefficient k of h at index k (what the prover feeds,
 *                       after into_repr, to the MSM over pk.h_query; the last coefficient is zero for a satisfied system)
 *     d_num_unsatisfied optional uint32_t[batch]: constraint rows the witness violates (h is then not a quotient)
 *     d_workspace       at least workspace_bytes_per_signature bytes; the batch is processed in chunks of as many
 *                       signatures as fit.
 * Returns FRW_E_INVALID_ARG for a null pointer, a workspace smaller than one signature's, or a domain outside 2^14 .. 2^30
 * (the transforms run as passes of six, five or four radix-2 stages: 2^17 = 6 + 6 + 5 and 2^18 = 6 + 6 + 6 for the Falcon
 * circuits, 2^19 = 6 + 5 + 4 + 4 .. 2^24 = 6 + 6 + 6 + 6, 2^25 = 6 + 5 + 5 + 5 + 4 .. 2^27 = 6 + 6 + 5 + 5 + 5 for aggregate statements.
 * RUN AND TESTED: every domain 2^17 .. 2^27 (tests/test_gpu_qap.py, tests/test_gpu_aggregate.py) -- 2^27 is the 1,024-statement
 * aggregate of BASELINE configs[4], whose thirteen per-index factor tables (56 GB) are made on the device; the schedules of
 * 2^28 .. 2^30 come from the same rule and are checked by the exact-integer model of tests/test_qap_schedule.py only: a 2^28 domain's
 * tables alone are 120 GB, and the smallest statement that needs it is 1,600 Falcon-1024 verifications).  Stream-ordered: everything is enqueued on
 * `stream` and NOTHING is allocated -- the sparse products borrow the working arrays, idle at that point, as their
 * scratch -- so a call may be captured in a HIP graph.  A HIP failure is recorded for frw_last_error().
 * Precondition: every witness / instance element is canonical Montgomery form (limbs < p), which is what arkworks and
 * the witness entry points produce; the 29-bit evaluation path does not reduce its inputs, so an element >= p gives an
 * h that is wrong without an error code. */
typedef struct {
    int32_t log_domain_size;
    uint64_t domain_size;                      /* n */
    uint64_t num_constraints, num_instance;    /* C, I */
    uint64_t workspace_bytes_per_signature;    /* 3 C x 32 (A z, B z, C z) + 3 x 32 n (working arrays) + 64 (flags) */
} frw_qap_info_t;
int frw_qap_info(const frw_r1cs *r, frw_qap_info_t *out);
int frw_qap_witness_map_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                            uint64_t *d_h, uint32_t *d_num_unsatisfied, void *d_workspace, size_t workspace_bytes,
                            void *stream);
/* The same quotient with six transforms instead of seven, for witnesses that satisfy the system: with
 * a(X) b(X) = lo(X) + X^n hi(X), h = hi = ((a b mod X^n - 1) - (a b mod X^n + 1)) / 2 -- the first from the pointwise
 * products on the domain, the second from those on the coset psi H, psi^n = -1; C z is computed for d_num_unsatisfied but
 * not transformed.  d_num_unsatisfied[i] == 0  =>  d_h[i] is exactly frw_qap_witness_map_dev's (ark-groth16's) h.
 * Otherwise d_h[i] is still hi, which is NOT what ark-groth16 returns for an unsatisfied system (nor a quotient).
 * Same arguments and workspace as frw_qap_witness_map_dev. */
int frw_qap_quotient_dev(const frw_r1cs *r, size_t batch, const uint64_t *d_witness, const uint64_t *d_instance,
                         uint64_t *d_h, uint32_t *d_num_unsatisfied, void *d_workspace, size_t workspace_bytes,
                         void *stream);
/* Host buffers: witness uint64_t[batch][W][4], instance uint64_t[batch][I][4] (the constant one first) -- the bytes of
 * arkworks' witness_assignment / instance_assignment -- to h uint64_t[batch][n][4]; num_unsatisfied may be NULL.
 * Synchronous; 64 signatures in flight; the device buffers belong to the handle and only grow (a second call of the
 * same shape allocates nothing: frw_r1cs_diag_host_allocations).  Same canonical-limbs precondition as above. */
int frw_qap_witness_map(const frw_r1cs *r, size_t batch, const uint64_t *witness, const uint64_t *instance,
                        uint64_t *h, uint32_t *num_unsatisfied);
int frw_r1cs_diag_host_allocations(const frw_r1cs *r, uint64_t *count);

/* ---- an aggregate statement: many Falcon verifications, ONE constraint system, ONE proof ---------------------------------------
 * BASELINE configs[4] / SURVEY 8-f row 4.  The reference's falcon-aggregate-sig crate is an empty stub
 * (falcon-aggregate-sig/src/main.rs:1-3); the statement is defined the only way the reference's own circuits allow:
 * FalconNTTVerificationCircuit::generate_constraints (falcon_ntt.rs:26-123) run once per (pk, msg, sig) on one constraint
 * system, in order (host mirror: FalconAggregateVerificationCircuit in csrc/host/frw_host.hpp), proved by the flow of
 * examples/pok_sig.rs:30-47.  For statements i = 0 .. count-1 with parameter sets logn[i] (9 or 10, freely mixed):
 *     instance_assignment = [1, pk_ntt_0, hm_ntt_0, pk_ntt_1, hm_ntt_1, ...]            I = 1 + sum 2 N_i
 *     witness_assignment  = witness_0 ++ witness_1 ++ ...                              W = sum W_i
 *     constraints         = those of statement 0, then of statement 1, ...            C = sum C_i
 *     QAP domain          = next_power_of_two(C + I):  2^18 for 512 + 1024, 2^19 for two Falcon-1024, 2^20 for four, 2^22 for sixteen
 * frw_r1cs_load_aggregate returns a handle that every entry point taking a `frw_r1cs *` accepts (frw_r1cs_check_dev,
 * frw_r1cs_eval_dev, frw_qap_info, frw_qap_witness_map_dev, frw_qap_quotient_dev, frw_groth16_workspace_bytes,
 * frw_groth16_prove_dev), with d_witness = uint64_t[batch][W][4] and d_instance = uint64_t[batch][I][4] the aggregate's OWN
 * vectors (batch = number of aggregate statements of this shape, normally 1) -- which frw_aggregate_assign_dev makes from the
 * batches the witness entry points wrote: statement i takes the next unused signature of its parameter set's batch, so
 * d_witness_512 / d_instance_512 hold the Falcon-512 statements in order and d_witness_1024 / d_instance_1024 the Falcon-1024
 * ones (either pair may be NULL if the aggregate has no such statement).  The aggregate's matrices are never materialised: they
 * are the per-signature systems' blocks, and a run of consecutive statements of one parameter set is one launch of that set's
 * kernels reading the aggregate vectors in place.  d_num_unsatisfied counts the violated rows of the whole statement.
 * frw_aggregate_assign_dev -- preconditions it cannot check: the Falcon-512 batch holds at least count_logn9 signatures and the
 * Falcon-1024 batch count_logn10 (frw_r1cs_info), both written with the SAME encoding (the aggregate's constant one is copied from the
 * first statement's own instance vector) and none of them rejected (FRW_ST_COEFF_RANGE zero-fills a slot: its leading one too).
 * Device-to-device copies on `stream` only -- two per run of consecutive statements of one parameter set --, capture-safe.
 * frw_groth16_setup_r1cs is frw_groth16_setup for the system behind any handle; the proving key of an aggregate has one query
 * point per variable of the whole statement.  As window tables that is 3.6 KB of G1 tables x 3 and 7.2 KB of G2 per variable and
 * 1.8 KB per domain point of h_query: 53 GB for sixteen Falcon-1024 statements -- the fastest proofs, up to there.  Beyond, the key
 * keeps the points alone (FRW_KEY_BARE below: 112 / 224 bytes a point) and the sums run window by window over them: the 1,024 mixed
 * statements of BASELINE configs[4] (513 Falcon-512 + 511 Falcon-1024: C + I = 126.6 M, the 2^27 domain -- round 4 had written 2^28
 * and 480 GB here, both wrong) are 121.9 M variables, 83 GB of points, ONE proof in 0.6 s on one MI355X. */
typedef struct {
    uint64_t num_statements;                    /* 1 for the handles of frw_r1cs_load */
    uint64_t count_logn9, count_logn10;
    uint64_t num_instance, num_witness, num_constraints;    /* I (with the constant one), W, C of the whole system */
    int32_t log_domain_size;
    int32_t witness_map_on_device;              /* 0: the device has no transform schedule for this domain (2^14 .. 2^30 have) */
} frw_r1cs_info_t;
int frw_r1cs_load_aggregate(int device, size_t count, const int32_t *logn, frw_r1cs **out);
int frw_r1cs_info(const frw_r1cs *r, frw_r1cs_info_t *out);
int frw_aggregate_assign_dev(const frw_r1cs *aggregate, const uint64_t *d_witness_512, const uint64_t *d_instance_512,
                             const uint64_t *d_witness_1024, const uint64_t *d_instance_1024, uint64_t *d_witness,
                             uint64_t *d_instance, void *stream);

/* ---- multi-scalar multiplication over BLS12-381 G1 (the step after the witness map in a Groth16 prover) --------------
 * ark-groth16 0.3.0 prover.rs, create_proof_with_reduction_and_matrices (what examples/pok_sig.rs:30-47 runs):
 *     h_acc = VariableBaseMSM::multi_scalar_mul(&pk.h_query, &h_assignment)
 * and likewise over a_query / b_g1_query / l_query.  The bases are the proving key's: fixed per circuit.  frw_msm_g1_load
 * takes them once (host memory, ark-ff's bytes: per point x then y, each 6 x uint64_t little-endian limbs of the Fq element
 * in Montgomery form, all zero = the point at infinity) and builds a device table of 2^(16 j) P_i for the sixteen 16-bit
 * windows j (448 bytes per point); frw_msm_g1_dev then computes sum_i k_i P_i for every signature of a batch:
 *     d_scalars   uint64_t[batch][scalar_stride][4]; the first num_points elements of each signature's vector are used.
 *                 montgomery != 0: ark-ff's Fr Montgomery form (what frw_qap_witness_map_dev writes); 0: canonical
 *                 integers < r (what into_repr() gives; a 256-bit integer >= r is taken mod r, never out of bounds)
 *     d_out       uint64_t[batch][12]: the affine result in ark-ff's bytes (x, y; all zero = infinity) -- the bytes of the
 *                 G1Affine arkworks' into_affine() would hold
 *     d_workspace at least workspace_bytes_per_signature (frw_msm_info) bytes, 16-byte aligned; the batch is processed in
 *                 chunks of as many signatures as fit
 * Stream-ordered, allocates nothing.  Scalars that are zero cost nothing and scalars that are one are summed apart from the
 * buckets, so the witness-side sums of the prover (a_query, b_g1_query, l_query with a Falcon witness: 91 % of its elements
 * are 0 or 1) run as fast as h_acc, whose coefficients are uniform field elements.  Any other heavy repetition of ONE digit
 * (all scalars equal to 2, say) is computed correctly and still in parallel: a bucket above 1.5 x the mean size is cut into
 * equal work items, one thread each, whose sums a second kernel adds up.
 * frw_groth16_msm_h_dev is the call for h_acc: scalars = the first domain_size - 1 coefficients of every h as the witness
 * map left them (stride domain_size, Montgomery form); num_points must equal domain_size - 1. */
/* k_i G1 for `count` canonical scalars (uint64_t[count][4], < r) -> uint64_t[count][12] affine points in ark-ff's bytes, host
 * buffers: the FixedBaseMSM over the generator ark-groth16's generator.rs builds the proving key's queries with (h_query[i] =
 * (zt / delta) t^i G1, ...) -- here so that a key can be made on the device; 8-bit windows, 32 mixed additions per scalar. */
int frw_g1_fixed_base(int device, size_t count, const uint64_t *scalars, uint64_t *out);
/* The same in G2 (b_g2_query[i] = v_i(t) G2): uint64_t[count][24] out -- x.c0, x.c1, y.c0, y.c1 of the Fq2 coordinates. */
int frw_g2_fixed_base(int device, size_t count, const uint64_t *scalars, uint64_t *out);
typedef struct frw_msm frw_msm;
typedef struct {
    uint64_t num_points;
    int32_t window_bits, num_windows;          /* 16, 16; a _narrow handle: 8, 32 */
    uint64_t table_bytes;                      /* 16 x num_points x 112 (G2: 224); a _narrow handle: 32 x num_points x 112, and for up to 2^18
                                                * points as much again for the sums of the 255 non-empty subsets of every group of eight points */
    uint64_t workspace_bytes_per_signature;    /* sort keys (64 num_points bytes), the list of scalars equal to one (4 num_points),
                                                * 32,768 buckets + 131,072 work items + 4,096 partial sums x 240 bytes (G2: 464),
                                                * counters; a _narrow handle: 128 + 4 num_points bytes of keys and list, the partial
                                                * sums of its work items and of the ones */
} frw_msm_info_t;
int frw_msm_g1_load(int device, size_t num_points, const uint64_t *bases, frw_msm **out);
void frw_msm_free(frw_msm *m);
int frw_msm_info(const frw_msm *m, frw_msm_info_t *out);
int frw_msm_g1_dev(const frw_msm *m, size_t batch, const uint64_t *d_scalars, size_t scalar_stride, int montgomery,
                   uint64_t *d_out, void *d_workspace, size_t workspace_bytes, void *stream);
int frw_groth16_msm_h_dev(const frw_msm *m, size_t batch, const uint64_t *d_h, size_t domain_size, uint64_t *d_out,
                          void *d_workspace, size_t workspace_bytes, void *stream);
/* G2 (the prover's g2_b = MSM(b_g2_query, assignment) + ...): bases uint64_t[num_points][24], results uint64_t[batch][24]
 * (x.c0, x.c1, y.c0, y.c1; ark-ff's G2Affine limbs), everything else as for G1.  A handle serves the group it was loaded for;
 * the other group's _dev call returns FRW_E_INVALID_ARG.  Same algorithm over Fq2 (a G2 addition is three times a G1 one and
 * the kernel spills registers: fine for the 10^5 additions a witness-side sum needs, not tuned for more). */
int frw_msm_g2_load(int device, size_t num_points, const uint64_t *bases, frw_msm **out);
/* The same handles with 8-bit windows (32 of them: the table is twice as long; num_points <= 2^25): 128 buckets per signature
 * instead of 32,768, so nothing is spent on folding empty buckets.  For sums whose scalars are mostly zero, one or small -- a
 * witness as scalars (the sums over a_query, b_g1_query, b_g2_query, l_query: frw_groth16_pk_load uses these) -- this is three
 * times faster than the 16-bit tables; for dense 255-bit scalars (h) it is twice the additions and the 16-bit tables win.
 * A narrow handle of up to 2^18 points also holds, for every group of eight consecutive points, the sums of its 255 non-empty subsets:
 * the scalars equal to one (45 % of a Falcon witness, in long runs of booleans) then cost one addition per group instead of one each.
 * Every other call (frw_msm_info, frw_msm_g1_dev / _g2_dev, frw_msm_free) takes either kind of handle. */
int frw_msm_g1_load_narrow(int device, size_t num_points, const uint64_t *bases, frw_msm **out);
int frw_msm_g2_load_narrow(int device, size_t num_points, const uint64_t *bases, frw_msm **out);
int frw_msm_g2_dev(const frw_msm *m, size_t batch, const uint64_t *d_scalars, size_t scalar_stride, int montgomery,
                   uint64_t *d_out, void *d_workspace, size_t workspace_bytes, void *stream);
/* BARE handles (round 5): the points themselves and no window table -- 112 (G2: 224) bytes a point, up to 2^31 - 1 of them.  A sum then
 * runs window by window over the same rows (narrow == 0: sixteen 16-bit windows, the dense pipeline -- for uniform scalars such as h;
 * narrow != 0: thirty-two 8-bit windows sorted in one pass -- for a witness as scalars), the window sums put together by Horner's rule:
 * the bucket additions of the table path, one per point and non-zero digit, plus 255 point operations.  Results are the same affine
 * points, bit for bit.  frw_msm_info: table_bytes = num_points x 112 (224); one scalar vector is summed at a time, whatever the workspace.
 * narrow == 2: a dense bare handle on WIDE windows -- thirteen of 20 bits instead of sixteen of 16: 19 % fewer bucket additions (a bare
 * handle's windows cost no memory), a two-level counting sort whose passes write runs; frw_msm_info then says 20 and 13.  A dense bare
 * handle of 2^26 points and more takes wide windows by itself (the folds of 13 x 2^19 buckets are a fixed 14 ms: below that size they cost
 * more than three windows' additions save); on the 2^27-point sum of the 1,024-statement aggregate 423 -> 356 ms. */
int frw_msm_g1_load_bare(int device, size_t num_points, const uint64_t *bases, int narrow, frw_msm **out);
int frw_msm_g2_load_bare(int device, size_t num_points, const uint64_t *bases, int narrow, frw_msm **out);

/* ---- a whole Groth16 proof per signature (the rest of examples/pok_sig.rs:30-47) ---------------------------------------------------
 * ark-groth16 0.3.0 prover.rs, create_proof_with_reduction_and_matrices, for every signature of a resident batch of witnesses:
 * the witness map, the five multi-scalar multiplications (h_query; a_query, b_g1_query, l_query, b_g2_query with the
 * assignment), the blinding terms and the assembly of (A, B, C).  frw_groth16_pk_load takes the proving key's elements in
 * ark-ff's bytes (host memory; G1 points 12 x uint64_t, G2 points 24) and builds the five device tables once.
 *     rs        host, uint64_t[batch][2][4]: the blinding factors r, s of every proof, canonical integers < the group order
 *               (create_random_proof draws them; create_proof_no_zk passes zeros)
 *     d_proofs  uint64_t[batch][48]: A (G1Affine limbs, 12), B (G2Affine limbs, 24), C (12)
 *     d_num_unsatisfied  optional uint32_t[batch]: constraint rows the witness violates (the proof is then worthless)
 *     d_workspace  frw_groth16_workspace_bytes(pk, r, in_flight) bytes, 256-byte aligned; the batch runs in chunks that fit
 * d_witness / d_instance: what the witness entry points wrote (FRW_ENC_MONTGOMERY).  Ordered on `stream` -- the sort of the
 * scalars' digits runs there, the rest on streams of the key's own (the witness map and the sum over h_query on one, the three G1
 * witness-side sums as one chain of kernels followed by both scalar multiplications on another, G2 on a third), forked from and
 * joined back into `stream` by events -- except for the upload of `rs`, which is waited for before the call goes on (the array
 * may be short-lived).
 * Calls with one key may come from several host threads: they take turns putting their work on the key's streams (each with a
 * workspace of its own).  A call that returns an error has waited for whatever it had already started.
 * NOT stream-capture safe (frw_groth16_prove_rs_dev is): once per chunk the call waits on the host (hipStreamSynchronize on
 * `stream`) for the upload of r, s -- i.e. also for whatever `stream` held before -- and only then takes the
 * key's lock, so concurrent provers do not wait for each other's streams. */
typedef struct frw_groth16_pk frw_groth16_pk;
typedef struct {
    uint64_t num_instance, num_witness, domain_size;      /* I (with the constant one), W, n */
    const uint64_t *alpha_g1, *beta_g1, *delta_g1;         /* vk.alpha_g1, pk.beta_g1, pk.delta_g1 */
    const uint64_t *beta_g2, *delta_g2;                    /* vk.beta_g2, vk.delta_g2 */
    const uint64_t *a_query, *b_g1_query;                  /* [I + W][12] */
    const uint64_t *b_g2_query;                            /* [I + W][24] */
    const uint64_t *h_query;                               /* [n - 1][12] */
    const uint64_t *l_query;                               /* [W][12] */
} frw_groth16_pk_desc_t;
int frw_groth16_pk_load(int device, const frw_groth16_pk_desc_t *desc, frw_groth16_pk **out);
/* ---- keys beyond what window tables can hold, and keys in slices (round 5; BASELINE configs[4]: ONE proof for 1,024 statements) -------
 * The window tables above are a speed-up, not a requirement: 3.6 KB per variable and 1.8 KB per domain point make a sixteen-statement key
 * 53 GB.  A key of BARE handles keeps the points only (112 bytes a G1 point, 224 a G2 point: 83 GB for the 1,024-statement aggregate --
 * 513 Falcon-512 + 511 Falcon-1024, 121.9 M variables, the 2^27 domain) and its sums run window by window over the same rows: the same
 * number of bucket additions (one per point and non-zero digit), plus 255 point operations per sum to put the windows together.  Such a
 * key also knows in which rows b_g1_query / b_g2_query hold a point -- the variables some constraint has on its B side: 59 % of a Falcon
 * circuit's, with half of a witness's ones and none of its full-size values: a third of the additions -- and sums those two tables over them alone (an index
 * made when the key is put together, a second, short sort per proof: frw_diag_groth16_side_counts has the numbers).
 *   FRW_KEY_TABLES  window tables (the fastest proofs; what frw_groth16_pk_load and frw_groth16_setup make)
 *   FRW_KEY_BARE    the points only
 *   FRW_KEY_AUTO    tables up to FRW_KEY_AUTO_TABLE_VARIABLES variables (some sixteen Falcon-1024 statements: 50 GB), bare beyond
 * rank / world: a key in slices (bare handles only) -- this handle holds slice `rank` of `world` of every query: rows [z_lo, z_hi) of the
 * nv + 3 rows of a_query ++ [alpha, delta, O] and of the three tables laid out like it, and [h_lo, h_hi) of h_query (frw_groth16_pk_info;
 * the split is by equal counts, the first `total mod world` slices one longer).  A slice proves nothing by itself: every rank runs
 * frw_groth16_prove_partial_dev on the WHOLE witness (the witness map is recomputed by every rank: a tenth of the sums' time) with the SAME
 * blinding factors, which sums its slices of the five queries into FRW_GROTH16_PARTIAL_WORDS uint64_t -- A | B1' | L | H as G1Affine limbs (12
 * each) | B as G2Affine limbs (24), partial sums all, the blinding terms r delta_1, alpha_1, beta, s delta_2 inside those of the last rank;
 * the partial sums of all ranks in rank order (one all-gather of 576 bytes per rank) go to frw_groth16_prove_combine_dev on any rank, which
 * adds them up, applies s A + r B1' and writes the proof A | B | C -- byte for byte the proof of the whole key. */
#define FRW_KEY_AUTO    0
#define FRW_KEY_TABLES  1
#define FRW_KEY_BARE    2
#define FRW_KEY_AUTO_TABLE_VARIABLES 3000000
typedef struct {
    int32_t mode;               /* FRW_KEY_* */
    uint32_t rank, world;       /* world <= 1: the whole key */
} frw_groth16_key_opts_t;
typedef struct {
    int32_t mode;               /* FRW_KEY_TABLES or FRW_KEY_BARE */
    uint32_t rank, world;
    uint64_t z_lo, z_hi;        /* rows of the witness-side tables this handle holds, of num_instance + num_witness + 3 */
    uint64_t h_lo, h_hi;        /* rows of h_query, of domain_size - 1 */
    uint64_t key_bytes;         /* device memory of the five tables */
} frw_groth16_pk_info_t;
int frw_groth16_pk_load_opts(int device, const frw_groth16_pk_desc_t *desc, const frw_groth16_key_opts_t *opts, frw_groth16_pk **out);
int frw_groth16_pk_info(const frw_groth16_pk *pk, frw_groth16_pk_info_t *out);
/* one of the key's five tables as the frw_msm handle it is (borrowed: it goes with the key; never frw_msm_free it) -- for sums over a
 * single query, e.g. frw_groth16_msm_h_dev(frw_groth16_pk_query(pk, FRW_QUERY_H), ...) checked against (h(t) zt / delta) G1.  The
 * witness-side tables have num_instance + num_witness + 3 rows (see frw_groth16_key_opts_t); a slice of a key is a slice here too. */
#define FRW_QUERY_H   0
#define FRW_QUERY_A   1
#define FRW_QUERY_B1  2
#define FRW_QUERY_L   3
#define FRW_QUERY_B2  4
const frw_msm *frw_groth16_pk_query(const frw_groth16_pk *pk, int which);
/* ark-groth16 0.3.0 generator.rs generate_parameters for one of the Falcon circuits, with the toxic waste GIVEN:
 * toxic = uint64_t[5][4], canonical: alpha, beta, gamma, delta and the evaluation point t (circuit_specific_setup draws them
 * from its rng, and random generators of G1 / G2; here the published generators are used).  The QAP is evaluated at t on the
 * host (the circuit's matrices, Lagrange coefficients), the queries are fixed-base multiples made on the device, and the proving
 * key comes back loaded.  vk_out (optional): alpha_g1 (12) | beta_g2 (24) | gamma_g2 (24) | delta_g2 (24) | gamma_abc_g1 [I][12]
 * uint64_t -- the verifying key's elements in ark-ff's bytes.  FRW_E_INVALID_ARG if t lies in the domain or gamma / delta is 0.
 * A key whose toxic waste somebody knows proves nothing to anybody else; this is for tests, benchmarks and ceremonies that
 * combine contributions. */
int frw_groth16_setup(int device, int circuit, int logn, const uint64_t *toxic, frw_groth16_pk **pk_out, uint64_t *vk_out);
/* the same for the system behind a handle (a per-signature circuit or an aggregate statement), on the handle's device;
 * vk_out: 84 + 12 x num_instance uint64_t */
int frw_groth16_setup_r1cs(const frw_r1cs *r, const uint64_t *toxic, frw_groth16_pk **pk_out, uint64_t *vk_out);
/* The same with the kind of key chosen (frw_groth16_setup_r1cs is FRW_KEY_AUTO, whole).  A key of bare handles is made ON THE DEVICE end to
 * end: the Lagrange coefficients of the whole domain at t, the QAP's polynomials at t as transposed sparse products over the per-signature
 * matrices (block by block, in place), the queries' scalars, and every query point written straight into its table row by the fixed-base
 * kernels -- nothing of the statement's size exists in host memory but the verifying key's gamma_abc_g1 (the 1,024-statement key: 83 GB of
 * rows, 1.57 M points of gamma_abc_g1).  With world > 1 only this rank's slices are made (vk_out, if given, is the whole verifying key on
 * every rank). */
int frw_groth16_setup_r1cs_opts(const frw_r1cs *r, const uint64_t *toxic, const frw_groth16_key_opts_t *opts, frw_groth16_pk **pk_out,
                                uint64_t *vk_out);
void frw_groth16_pk_free(frw_groth16_pk *pk);
size_t frw_groth16_workspace_bytes(const frw_groth16_pk *pk, const frw_r1cs *r, size_t batch_in_flight);
int frw_groth16_prove_dev(const frw_groth16_pk *pk, const frw_r1cs *r, size_t batch, const uint64_t *d_witness,
                          const uint64_t *d_instance, const uint64_t *rs, uint64_t *d_proofs, uint32_t *d_num_unsatisfied,
                          void *d_workspace, size_t workspace_bytes, void *stream);
/* The same with the blinding factors in DEVICE memory (d_rs: uint64_t[batch][2][4], any 256-bit values: taken modulo the group
 * order; their split by the curve's endomorphism, done on the host for the call above until round 4, happens on the device for
 * both).  Nothing waits on the host: the call is ordered on `stream` throughout and MAY be captured into a HIP graph and replayed
 * (one capture = one chunk: give it the workspace of the whole batch; the key's streams join the capture through its events and
 * leave it before the call returns).  A replay reads the factors that are in d_rs THEN: refresh them between replays. */
int frw_groth16_prove_rs_dev(const frw_groth16_pk *pk, const frw_r1cs *r, size_t batch, const uint64_t *d_witness,
                             const uint64_t *d_instance, const uint64_t *d_rs, uint64_t *d_proofs, uint32_t *d_num_unsatisfied,
                             void *d_workspace, size_t workspace_bytes, void *stream);

/* A key in slices (see frw_groth16_key_opts_t).  d_witness / d_instance: the WHOLE statement's vectors, on every rank; rs: host,
 * uint64_t[batch][2][4], the same on every rank; d_partial: uint64_t[batch][FRW_GROTH16_PARTIAL_WORDS]; workspace as frw_groth16_prove_dev
 * (frw_groth16_workspace_bytes).  A whole key (world == 1) may be used too: its "partial" sums are the sums. */
#define FRW_GROTH16_PARTIAL_WORDS 72
#define FRW_GROTH16_COMBINE_WORKSPACE 4096
int frw_groth16_prove_partial_dev(const frw_groth16_pk *pk, const frw_r1cs *r, size_t batch, const uint64_t *d_witness,
                                  const uint64_t *d_instance, const uint64_t *rs, uint64_t *d_partial, uint32_t *d_num_unsatisfied,
                                  void *d_workspace, size_t workspace_bytes, void *stream);
/* d_partials: uint64_t[world][FRW_GROTH16_PARTIAL_WORDS] in rank order (device memory); rs: host, uint64_t[2][4]; d_proof: uint64_t[48];
 * d_workspace: FRW_GROTH16_COMBINE_WORKSPACE bytes, 256-byte aligned.  `pk` names the device (any rank's handle). */
int frw_groth16_prove_combine_dev(const frw_groth16_pk *pk, size_t world, const uint64_t *d_partials, const uint64_t *rs, uint64_t *d_proof,
                                  void *d_workspace, size_t workspace_bytes, void *stream);

/* ---- Groth16 verification (examples/pok_sig.rs:34-47: Groth16::verify(&vk, &public_inputs, &proof)) --------------------------------
 * ark-groth16 0.3.0 verifier.rs: prepare_verifying_key (e(alpha_g1, beta_g2), -gamma_g2, -delta_g2), prepare_inputs
 * (gamma_abc_g1[0] + sum x_i gamma_abc_g1[i]) and verify_proof: e(A, B) e(inputs, -gamma_g2) e(C, -delta_g2) == e(alpha_g1, beta_g2)
 * as one product of three Miller loops and one final exponentiation.  HOST code like the reference's (about 15 ms per proof on
 * one core; a batch runs one proof per host thread); no device is needed.
 *     vk          the layout frw_groth16_setup writes: alpha_g1 | beta_g2 | gamma_g2 | delta_g2 | gamma_abc_g1[num_instance]
 *                 (FRW_E_INVALID_ARG if a coordinate's limbs are not below the field modulus, a point is not on its curve or
 *                 not in the subgroup of order r -- every point, gamma_abc_g1 included: what ark's deserialiser checks)
 *     instance    uint64_t[batch][num_instance][4], encoding FRW_ENC_MONTGOMERY or FRW_ENC_CANONICAL: the instance vector AS THE
 *                 WITNESS ENTRY POINTS WRITE IT, i.e. the constant one first and then ark's public inputs (pk_ntt || hm_ntt)
 *     proofs      uint64_t[batch][48]: A | B | C as frw_groth16_prove_dev writes them
 *     flags       FRW_VERIFY_POINTS_ARE_CHECKED: skip the subgroup checks of A, B, C (ark checks them when it deserialises a proof;
 *                 these are raw limbs, so the check is made here unless the caller vouches for the points)
 *     accepted    int32_t[batch]: 1 the proof verifies, 0 it does not, -1 malformed: an instance value whose limbs (in either
 *                 encoding) are >= r, instance[0] != 1, a coordinate whose limbs are >= the field modulus (x + q is not accepted
 *                 for x), a point off its curve or outside the subgroup -- or, with FRW_VERIFY_POINTS_ARE_CHECKED, a point the
 *                 caller vouched for wrongly that drives the Miller loop into a vertical line */
#define FRW_VERIFY_POINTS_ARE_CHECKED 1
typedef struct frw_groth16_vk frw_groth16_vk;
int frw_groth16_vk_load(const uint64_t *vk, size_t num_instance, frw_groth16_vk **out);
/* flags: FRW_VK_POINTS_ARE_CHECKED -- the caller vouches for gamma_abc_g1 being on the curve and in the subgroup (a key it made itself a
 * moment ago: the 1,024-statement aggregate has 1.57 M of these points, a 255-bit ladder each); canonical limbs are still required and the
 * four fixed points are still checked. */
#define FRW_VK_POINTS_ARE_CHECKED 1
int frw_groth16_vk_load_opts(const uint64_t *vk, size_t num_instance, int flags, frw_groth16_vk **out);
void frw_groth16_vk_free(frw_groth16_vk *vk);
int frw_groth16_verify(const frw_groth16_vk *vk, size_t batch, const uint64_t *instance, int encoding, const uint64_t *proofs,
                       int flags, int32_t *accepted);
/* diagnostics for the parity tests: the verifier's pairing of one pair (g1: 12, g2: 24 uint64_t), written as the twelve
 * coefficients of 1, w, ..., w^11 in Fq[w] / (w^12 - 2 w^6 + 2), 6 uint64_t each in ark-ff's form.  The value is the CUBE of the
 * reduced ate pairing (frw_pairing.h says why that is as good). */
int frw_diag_pairing(const uint64_t *g1, const uint64_t *g2, uint64_t *out);

/* ---- input preparation (what the reference does with falcon-rust before any gadget runs) ---------------------
 * falcon_ntt.rs:27-28,44: sig_poly = Polynomial::from(&sig), pk_poly = Polynomial::from(&pk),
 * hm = Polynomial::from_hash_of_message(msg, sig.nonce()).  Formats are the Falcon specification's:
 *   public key  FRW_PK_LEN(logn)  = 1 + 14 N / 8 bytes: header 0x00 + logn, N x 14-bit coefficients (big-endian bits)
 *   signature   sig_len bytes (falcon.rs: 666 / 1280, the padded lengths): header 0x30 + logn, 40-byte nonce,
 *               compressed s2 (sign bit, 7 low bits, unary high part), zero padding
 *   hash        SHAKE256(nonce || msg) read as big-endian 16-bit words w, w < 5q accepted as w mod q
 * Messages are one byte blob plus batch+1 NON-DECREASING offsets (message i = msgs[msg_off[i] .. msg_off[i+1]));
 * frw_prepare_inputs returns FRW_E_INVALID_ARG for decreasing offsets, the _dev variant reads such a message as empty
 * and trusts the caller that msg_off[batch] does not exceed the blob.
 * status[i] = FRW_ST_DECODE for a malformed encoding (that signature must not be fed to the witness call). */
#define FRW_NONCE_LEN 40
#define FRW_PK_LEN(logn)  (1 + 14 * (1 << (logn)) / 8)
#define FRW_SIG_LEN(logn) ((logn) == 9 ? 666 : 1280)

int frw_hash_to_point_dev(frw_ctx *ctx, int logn, size_t batch, const uint8_t *d_nonces /* batch x 40 */,
                          const uint8_t *d_msgs, const uint64_t *d_msg_off /* batch + 1 */, uint16_t *d_hm, void *stream);
int frw_decode_public_keys_dev(frw_ctx *ctx, int logn, size_t batch, const uint8_t *d_pk_bytes, uint16_t *d_pk,
                               int32_t *d_status, void *stream);
int frw_decode_signatures_dev(frw_ctx *ctx, int logn, size_t batch, const uint8_t *d_sig_bytes, size_t sig_len,
                              uint16_t *d_sig, uint8_t *d_nonce_out /* batch x 40, may be NULL */, int32_t *d_status,
                              void *stream);
/* host buffers: decode + hash for `batch` (pk, msg, sig) triples -> the three coefficient vectors the witness
 * entry points take.  status[i]: FRW_ST_OK or FRW_ST_DECODE. */
int frw_prepare_inputs(frw_ctx *ctx, int logn, size_t batch, const uint8_t *pk_bytes, const uint8_t *sig_bytes,
                       size_t sig_len, const uint8_t *msgs, const uint64_t *msg_off, uint16_t *sig, uint16_t *pk,
                       uint16_t *hm, int32_t *status);

/* ---- stand-alone gadget blocks ---------------------------------------------------------------
 * The reference's gadget functions are also called outside the full circuit (its unit tests do; so can any
 * other circuit built from them).  One call fills the witness block of `count` independent gadget invocations,
 * `frw_gadget_block_len(kind)` field elements each, in the order the gadget allocates them:
 *   FRW_G_LESS_THAN_Q     enforce_less_than_q(a)            range_proofs.rs:42-94     a: uint64_t[count]            27
 *   FRW_G_MOD_Q           mod_q(a, q) -> [t, b, ltq(b)]     arithmetics.rs:105-149    a: uint32_t[count][5] (LE limbs, a < 2^160)  29
 *   FRW_G_ADD_MOD         add_mod(a, b, q) -> [t, c, ltq(c)] arithmetics.rs:214-262   a, b: uint64_t[count], a+b < 2^64  29
 *   FRW_G_L2_ELEM         one l2_norm_var element -> [a0..a13, w0, w1, r, sq]  misc.rs:35-47, range_proofs.rs:289-333
 *                                                                                     a: uint64_t[count], a <= q     18
 *   FRW_G_NORM_BOUND_512  enforce_less_than_norm_bound      range_proofs.rs:100-186   a: uint64_t[count]            50
 *   FRW_G_NORM_BOUND_1024                                   range_proofs.rs:192-272   a: uint64_t[count]            52
 * As in the reference, bit decompositions take the LOW bits of the value (to_bits_le().take(k)); a value that does
 * not fit yields the block the reference's cfg(test) build assigns (an unsatisfied system), not an error.
 * status[i] = FRW_ST_COEFF_RANGE only where the input is outside the documented domain.  b is ignored (may be NULL)
 * unless kind == FRW_G_ADD_MOD. */
#define FRW_G_LESS_THAN_Q       0
#define FRW_G_MOD_Q             1
#define FRW_G_ADD_MOD           2
#define FRW_G_L2_ELEM           3
#define FRW_G_NORM_BOUND_512    4
#define FRW_G_NORM_BOUND_1024   5

int frw_gadget_block_len(int kind);   /* elements per block, or FRW_E_INVALID_ARG */
int frw_gadget_dev(frw_ctx *ctx, int kind, size_t count, const void *d_a, const uint64_t *d_b, int encoding,
                   uint64_t *d_out, int32_t *d_status, void *stream);
int frw_gadget(frw_ctx *ctx, int kind, size_t count, const void *a, const uint64_t *b, int encoding,
               uint64_t *out, int32_t *status);

/* ---- utilities ---------------------------------------------------------------------------- */
/* Per-item digest of a device buffer of `items` x `words_per_item` uint64_t:
 * d_out[i] = sum_j splitmix64(buf[i][j] + j * 0x9E3779B97F4A7C15) mod 2^64.
 * Lets a host compare whole HBM-resident witness batches without copying them back. */
int frw_digest_dev(frw_ctx *ctx, const uint64_t *d_buf, size_t words_per_item, size_t items,
                   uint64_t *d_out, void *stream);

/* What a witness launch of `batch` signatures looks like on this device: out = {workgroups launched, resident
 * workgroups per CU the grid was sized for, CUs, number of signatures (the ragged tail beyond the last full round of
 * the grid, or a whole small batch) that are cut into five work items each}.  Launches whose batch is a multiple of
 * out[0] run fastest (static striding, no tail). */
int frw_diag_launch_shape(frw_ctx *ctx, int logn, int encoding, size_t batch, int32_t out[4]);

/* Roofline calibration: overwrites d_buf[0, bytes) with a compute-free write stream of the witness kernel's store
 * shape (workgroup-contiguous slabs of slab_bytes, 16 B per lane).  Timed by bench.py on the same device and
 * stream as the witness kernel to report the write bandwidth the device's HBM actually sustains. */
int frw_diag_write_stream_dev(frw_ctx *ctx, void *d_buf, size_t bytes, size_t slab_bytes, void *stream);

/* Synthetic, always-valid inputs (host side; stands where the reference's tests call
 * KeyPair::keygen + sign, falcon_ntt.rs:134-138): sig, v ~ rounded Gaussian(sigma_logn),
 * pk uniform in [0,q), hm := v + sig*pk mod (x^N+1, q); triples whose norm reaches the bound are
 * redrawn.  Counter-based: triple i depends only on (seed, first_index + i). */
int frw_synth_triples(int logn, size_t batch, uint64_t seed, uint64_t first_index,
                      uint16_t *sig, uint16_t *pk, uint16_t *hm);

/* Issue rates of the vector instructions a BLS12-381 Fr product is built from, measured on this device (what the QAP
 * transforms' roofline is priced with; bench.py): out[0] = v_add_u32, out[1] = v_mad_u64_u32, both in wave-instructions
 * per SIMD per microsecond with four waves per SIMD; out[2] = field products per second of the bare multiplier loop
 * (frw_fr29.h f29_mul, no loads, no butterflies) over the whole chip; out[3] = number of SIMDs.  Synchronous, ~10 ms. */
int frw_diag_valu_rates(frw_ctx *ctx, double out[4]);

/* p(t) on the device for a polynomial in device memory: d_coeffs uint64_t[n][4] (ark-ff's Montgomery form, coefficient k at index k -- an
 * h of frw_qap_witness_map_dev), t host, canonical; out: host uint64_t[4], canonical.  For checks such as h_acc == (h(t) zt / delta) G1 on
 * domains of 2^27 coefficients.  Synchronous; allocates its own small scratch. */
int frw_diag_poly_eval_dev(int device, uint64_t n, const uint64_t *d_coeffs, const uint64_t *t, uint64_t *out);

/* What the four witness-side sums of a key of BARE handles add up for the scalars d_z (device, uint64_t[rows of this key's slice][4],
 * Montgomery: z ++ [1, r, s] from row z_lo on): out[0] = the rows in which b_g1_query / b_g2_query hold a point (a variable that no
 * constraint has on its B side has the point at infinity there, and the sums over those two tables skip it: frw_groth16_pk.b_index);
 * out[1], out[2] = non-zero 8-bit digits of the scalars that are not one, and scalars equal to one, over all rows (a_query, l_query);
 * out[3], out[4] = the same over the rows of out[0] (b_g1_query, b_g2_query).  One point addition each.  The benchmark prices the
 * aggregate proof's roofline with these.  d_workspace: 256-byte aligned, frw_groth16_workspace_bytes(pk, r, 1) is enough.  Synchronous. */
int frw_diag_groth16_side_counts(const frw_groth16_pk *pk, const uint64_t *d_z, void *d_workspace, size_t workspace_bytes, void *stream,
                                 uint64_t *out);

/* Page-locked host memory.  Output buffers of the host-buffer entry points allocated here are filled by
 * asynchronous DMA that overlaps with the kernels of the next chunk (pageable buffers work too, slower). */
int frw_host_alloc(frw_ctx *ctx, size_t bytes, void **ptr);
int frw_host_free(frw_ctx *ctx /* may be NULL */, void *ptr);

/* thin wrappers so that a host without a HIP binding can own device memory */
int frw_malloc(frw_ctx *ctx, size_t bytes, void **d_ptr);
int frw_free(frw_ctx *ctx, void *d_ptr);
int frw_memcpy_h2d(frw_ctx *ctx, void *d_dst, const void *src, size_t bytes);
int frw_memcpy_d2h(frw_ctx *ctx, void *dst, const void *d_src, size_t bytes);
int frw_synchronize(frw_ctx *ctx, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* FRW_H */
