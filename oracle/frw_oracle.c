/*
 * ORACLE (test infrastructure only) -- closed-form CPU restatement, in plain C, of the witness the
 * reference's FalconNTTVerificationCircuit::generate_constraints assigns.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
 * It is the checker, never the product: the product path (falcon-r1cs_amd/csrc) does not link,
 * include or call anything here and fails loudly without a GPU.
 *
 * PARITY UNPINNED for full-witness values: the reference (Rust, arkworks 0.3 + falcon-rust) cannot
 * be built or run here and carries no golden witness vectors.  This restatement is pinned by
 *   (1) tests/test_oracle.py: bit-equality with oracle/falcon_gadgets.py (the gadget-by-gadget
 *       restatement executed against the arkworks front-end simulation, whose variable and
 *       constraint counts equal README.md:41-56 and whose constraint system is satisfied), and
 *   (2) the known-answer cases of the reference's gadget unit tests (see oracle/README.md).
 *
 * All file:line citations are relative to /root/reference/.
 */
#include <stdint.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define Q 12289u                       /* falcon-rust MODULUS; gadgets/arithmetics.rs:5 */
#define GEN 7u                         /* Falcon's 2048-th root of unity mod q (vrfy.c GMb) */

typedef unsigned __int128 u128;
typedef struct { uint64_t l[4]; } fr_t;                 /* ark-ff Fp256: 4 x u64 LE limbs */
typedef struct { uint64_t l[3]; } u192;                 /* ladder integers: <= 160 bits */

/* BLS12-381 scalar field (ark_ed_on_bls12_381::fq::Fq; gadgets/poly.rs:244) */
static const uint64_t FR_P[4]  = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const uint64_t FR_R[4]  = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
static const uint64_t FR_R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
#define FR_INV 0xfffffffeffffffffULL   /* -p^-1 mod 2^64 */

/* ---------------------------------------------------------------------------------------------
 * layout (allocation order of circuits/falcon_ntt.rs:58-122); offsets in witnesses
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int32_t logn, n;
    int32_t num_witness;       /* W = 153 N + {50|52} */
    int32_t num_instance;      /* I = 2 N + 1 (leading constant one included) */
    int32_t num_constraints;   /* C = 159 N + {52|54} */
    int32_t seg_off[8];        /* S0 sig, S1 v, S2 ltq(v), S3 mod_q(NTT sig), S4 mod_q(NTT v), S5 pointwise, S6 l2, S7 norm */
    int32_t seg_len[8];
} oracle_layout_t;

int frw_oracle_layout(int logn, oracle_layout_t *o)
{
    if (logn != 9 && logn != 10) return -1;
    int n = 1 << logn, nb = logn == 9 ? 50 : 52;
    int len[8] = {n, n, 27 * n, 29 * n, 29 * n, 30 * n, 36 * n, nb};
    int off = 0;
    o->logn = logn; o->n = n;
    for (int i = 0; i < 8; i++) { o->seg_off[i] = off; o->seg_len[i] = len[i]; off += len[i]; }
    o->num_witness = off;
    o->num_instance = 2 * n + 1;
    o->num_constraints = 159 * n + nb + 2;
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * tables: falcon-rust NTT_TABLE[i] = 7^bitrev10(i) mod q (script/ntt_param.sage:3-132), its
 * inverse, and the ladder offsets C_k = 2^k q^(k+1) (circuits/falcon_ntt.rs:31-39)
 * ------------------------------------------------------------------------------------------- */
static uint32_t TW[1024], ITW[1024];
static u192 CK[11];
static pthread_once_t tables_once = PTHREAD_ONCE_INIT;

static uint32_t powmod(uint32_t b, uint32_t e)
{
    uint32_t r = 1;
    while (e) { if (e & 1) r = (uint32_t)((uint64_t)r * b % Q); b = (uint32_t)((uint64_t)b * b % Q); e >>= 1; }
    return r;
}

static void u192_mul_small(u192 *r, const u192 *a, uint64_t s)
{
    u128 c = 0;
    for (int i = 0; i < 3; i++) { c += (u128)a->l[i] * s; r->l[i] = (uint64_t)c; c >>= 64; }
}

static void init_tables(void)
{
    for (uint32_t i = 0; i < 1024; i++) {
        uint32_t r = 0;
        for (int b = 0; b < 10; b++) if (i & (1u << b)) r |= 1u << (9 - b);
        TW[i] = powmod(GEN, r);
        ITW[i] = powmod(GEN, (2048 - r) % 2048);
    }
    u192 c = {{Q, 0, 0}};                                  /* C_0 = q */
    CK[0] = c;
    for (int k = 1; k <= 10; k++) { u192_mul_small(&c, &c, 2 * (uint64_t)Q); CK[k] = c; }
}

/* ---------------------------------------------------------------------------------------------
 * falcon-rust clear arithmetic used by circuits/falcon_ntt.rs:44-51
 * ------------------------------------------------------------------------------------------- */
/* NTTPolynomial::from(&Polynomial): Falcon mq_NTT schedule == the loop nest of gadgets/poly.rs:115-149 */
static void ntt_modq(uint32_t *a, int logn)
{
    int n = 1 << logn, t = n;
    for (int m = 1; m < n; m <<= 1) {
        int ht = t >> 1, j1 = 0;
        for (int i = 0; i < m; i++, j1 += t) {
            uint32_t s = TW[m + i];
            for (int j = j1; j < j1 + ht; j++) {
                uint32_t u = a[j], v = a[j + ht] * s % Q;
                a[j] = (u + v) % Q;
                a[j + ht] = (u + Q - v) % Q;
            }
        }
        t = ht;
    }
}

/* inverse of ntt_modq (Falcon mq_iNTT), incl. the 1/N scaling */
static void intt_modq(uint32_t *a, int logn)
{
    int n = 1 << logn, t = 1;
    for (int m = n; m > 1; m >>= 1) {
        int hm = m >> 1, dt = t << 1, j1 = 0;
        for (int i = 0; i < hm; i++, j1 += dt) {
            uint32_t s = ITW[hm + i];
            for (int j = j1; j < j1 + t; j++) {
                uint32_t u = a[j], v = a[j + t];
                a[j] = (u + v) % Q;
                a[j + t] = (u + Q - v) % Q * s % Q;
            }
        }
        t = dt;
    }
    uint32_t ninv = powmod((uint32_t)n, Q - 2);
    for (int i = 0; i < n; i++) a[i] = a[i] * ninv % Q;
}

/* ---------------------------------------------------------------------------------------------
 * ark-ff Fp256 encoding of witness_assignment / instance_assignment elements
 * ------------------------------------------------------------------------------------------- */
static void fr_mont_mul(fr_t *r, const uint64_t a[4], const uint64_t b[4])
{
    uint64_t t[6] = {0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 4; i++) {
        u128 c = 0;
        for (int j = 0; j < 4; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * FR_INV;
        c = (u128)m * FR_P[0] + t[0]; c >>= 64;
        for (int j = 1; j < 4; j++) { c += (u128)m * FR_P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
        c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
    }
    /* conditional subtract */
    uint64_t d[4]; u128 bw = 0;
    for (int j = 0; j < 4; j++) { u128 x = (u128)t[j] - FR_P[j] - (uint64_t)bw; d[j] = (uint64_t)x; bw = (x >> 64) & 1; }
    int ge = t[4] || !bw;
    for (int j = 0; j < 4; j++) r->l[j] = ge ? d[j] : t[j];
}

/* integer (<= 192 bits, < p) -> stored form: canonical limbs (encoding 0) or Montgomery x*R mod p (encoding 1) */
static void fr_encode(fr_t *r, const u192 *x, int encoding)
{
    uint64_t a[4] = {x->l[0], x->l[1], x->l[2], 0};
    if (encoding) fr_mont_mul(r, a, FR_R2);
    else memcpy(r->l, a, sizeof a);
}

static void fr_encode_small(fr_t *r, uint64_t x, int encoding)
{
    u192 v = {{x, 0, 0}};
    fr_encode(r, &v, encoding);
}

/* ---------------------------------------------------------------------------------------------
 * gadget witness blocks
 * ------------------------------------------------------------------------------------------- */
typedef struct { fr_t zero, one; int encoding; } enc_t;

static inline void put_bit(fr_t *w, const enc_t *e, unsigned bit) { *w = bit ? e->one : e->zero; }

/* gadgets/range_proofs.rs:42-94 enforce_less_than_q: 27 witnesses
 *   a0..a13 (:62-69), w0..w10 = kary_or(a0..a11) fold-left (:86), w11 = w10 & a12 (:84), w12 = w11 & a13 (:81-82) */
static fr_t *emit_ltq(fr_t *w, const enc_t *e, uint32_t a)
{
    for (int i = 0; i < 14; i++) put_bit(w++, e, (a >> i) & 1);
    unsigned acc = a & 1;
    for (int i = 1; i <= 11; i++) { acc |= (a >> i) & 1; put_bit(w++, e, acc); }
    unsigned w11 = acc & ((a >> 12) & 1);
    put_bit(w++, e, w11);
    put_bit(w++, e, w11 & ((a >> 13) & 1));
    return w;
}

/* gadgets/arithmetics.rs:105-149 mod_q on a ladder output: [t, b, ltq(b)] = 29 witnesses */
static fr_t *emit_mod_q(fr_t *w, const enc_t *e, const u192 *a, uint32_t *b_out)
{
    u192 t; uint64_t r = 0;
    for (int i = 2; i >= 0; i--) { u128 cur = ((u128)r << 64) | a->l[i]; t.l[i] = (uint64_t)(cur / Q); r = (uint64_t)(cur % Q); }
    fr_encode(w++, &t, e->encoding);                      /* t_var :137 */
    fr_encode_small(w++, r, e->encoding);                 /* b_var :138 */
    *b_out = (uint32_t)r;
    return emit_ltq(w, e, (uint32_t)r);                   /* :146 */
}

/* gadgets/poly.rs:113-149: the un-reduced butterfly ladder over the integers */
static void ladder(u192 *out, const uint32_t *in, int logn)
{
    int n = 1 << logn, t = n;
    for (int j = 0; j < n; j++) { out[j].l[0] = in[j]; out[j].l[1] = out[j].l[2] = 0; }
    for (int l = 0; l < logn; l++) {
        int m = 1 << l, ht = t >> 1, j1 = 0;
        const u192 *c = &CK[l + 1];
        for (int i = 0; i < m; i++, j1 += t) {
            uint64_t s = TW[m + i];
            for (int j = j1; j < j1 + ht; j++) {
                u192 u = out[j], v, x, y;
                u192_mul_small(&v, &out[j + ht], s);                       /* v = out[j+ht] * s   :136 */
                u128 cy = 0, bw = 0;
                for (int k = 0; k < 3; k++) {                              /* out[j] = u + v      :141 */
                    cy += (u128)u.l[k] + v.l[k]; x.l[k] = (uint64_t)cy; cy >>= 64;
                }
                cy = 0;
                for (int k = 0; k < 3; k++) {                              /* out[j+ht] = u + (C_{l+1} - v)  :137,:142 */
                    u128 d = (u128)c->l[k] - v.l[k] - (uint64_t)bw; bw = (d >> 64) & 1;
                    cy += (u128)u.l[k] + (uint64_t)d; y.l[k] = (uint64_t)cy; cy >>= 64;
                }
                out[j] = x; out[j + ht] = y;
            }
        }
        t = ht;
    }
}

/* gadgets/poly.rs:104-159 ntt_circuit: ladder, then mod_q per output in index order (:154-156) */
static fr_t *emit_ntt_circuit(fr_t *w, const enc_t *e, const uint32_t *in, int logn, uint32_t *b_out, u192 *scratch)
{
    int n = 1 << logn;
    ladder(scratch, in, logn);
    for (int k = 0; k < n; k++) w = emit_mod_q(w, e, &scratch[k], &b_out[k]);
    return w;
}

/* gadgets/misc.rs:30-51 l2_norm_var element + gadgets/range_proofs.rs:289-333 is_less_than_6144: 18 witnesses
 *   a0..a13, w0 = a11 & a12, w1 = nor(a13, w0), r = select(w1, e, q - e), sq = r * r */
static fr_t *emit_l2_elem(fr_t *w, const enc_t *e, uint32_t a, uint64_t *norm)
{
    for (int i = 0; i < 14; i++) put_bit(w++, e, (a >> i) & 1);
    unsigned w0 = ((a >> 11) & 1) & ((a >> 12) & 1);
    unsigned w1 = !((a >> 13) & 1) & !w0;
    put_bit(w++, e, w0);
    put_bit(w++, e, w1);
    /* modulus_var - e is a field subtraction; e < 2^16 so q - e may be negative only if e > q, which the
     * status word excludes: inputs are checked < q before this point */
    uint64_t r = w1 ? a : (uint64_t)Q - a;
    fr_encode_small(w++, r, e->encoding);
    fr_encode_small(w++, r * r, e->encoding);
    *norm += r * r;
    return w;
}

/* gadgets/range_proofs.rs:100-186 (falcon-512): 26 bits + 24 gates */
static fr_t *emit_norm_512(fr_t *w, const enc_t *e, uint64_t a)
{
    unsigned b[26], g[24];
    for (int i = 0; i < 26; i++) { b[i] = (a >> i) & 1; put_bit(w++, e, b[i]); }
    g[0] = b[19] | b[20]; for (int i = 1; i <= 4; i++) g[i] = g[i - 1] | b[20 + i];     /* kary_or a19..a24  :150 */
    g[5] = b[16] & b[17]; g[6] = g[5] & b[18];                                           /* kary_and a16..a18 :152 */
    g[7] = b[6] | b[7]; g[8] = g[7] | b[8]; g[9] = g[8] | b[9];                          /* kary_or a6..a9    :166 */
    g[10] = b[3] | b[4];                                                                 /* :170 */
    g[11] = b[1] & b[2];                                                                 /* :172 */
    g[12] = !g[10] & !g[11];          /* nor   :170-173 */
    g[13] = b[5] & !g[12];            /* :168 */
    g[14] = !g[9] & !g[13];           /* :166 */
    g[15] = b[10] & !g[14];           /* :164 */
    g[16] = !b[11] & !g[15];          /* :162 */
    g[17] = b[12] & !g[16];           /* :160 */
    g[18] = !b[13] & !g[17];          /* :158 */
    g[19] = b[14] & !g[18];           /* :156 */
    g[20] = !b[15] & !g[19];          /* :154 */
    g[21] = g[6] & !g[20];            /* :152 */
    g[22] = !g[4] & !g[21];           /* :150 */
    g[23] = b[25] & !g[22];           /* :148 */
    for (int i = 0; i < 24; i++) put_bit(w++, e, g[i]);
    return w;
}

/* gadgets/range_proofs.rs:192-272 (falcon-1024): 27 bits + 25 gates */
static fr_t *emit_norm_1024(fr_t *w, const enc_t *e, uint64_t a)
{
    unsigned b[27], g[25];
    for (int i = 0; i < 27; i++) { b[i] = (a >> i) & 1; put_bit(w++, e, b[i]); }
    g[0] = b[22] | b[23]; g[1] = g[0] | b[24]; g[2] = g[1] | b[25];                      /* :239 */
    g[3] = b[20] & b[21];                                                                /* :241 */
    g[4] = b[14] | b[15]; for (int i = 5; i <= 8; i++) g[i] = g[i - 1] | b[11 + i];      /* kary_or a14..a19 :243 */
    g[9] = b[9] | b[10];                                                                 /* :251 */
    g[10] = b[7] & b[8];                                                                 /* :253 */
    g[11] = b[5] | b[6];                                                                 /* :255 */
    g[12] = b[3] & b[4];                                                                 /* :257 */
    g[13] = b[1] | b[2];                                                                 /* :259 */
    g[14] = g[13] & g[12];            /* Not(w12).or(Not(w13)) -> and(w13, w12) :257-259 */
    g[15] = !g[11] & !g[14];          /* :255 */
    g[16] = g[10] & !g[15];           /* :253 */
    g[17] = !g[9] & !g[16];           /* :251 */
    g[18] = b[11] & !g[17];           /* :249 */
    g[19] = !b[12] & !g[18];          /* :247 */
    g[20] = b[13] & !g[19];           /* :245 */
    g[21] = !g[8] & !g[20];           /* :243 */
    g[22] = g[3] & !g[21];            /* :241 */
    g[23] = !g[2] & !g[22];           /* :239 */
    g[24] = b[26] & !g[23];           /* :237 */
    for (int i = 0; i < 25; i++) put_bit(w++, e, g[i]);
    return w;
}

static const uint64_t SIG_L2_BOUND[2] = {34034726ULL, 70265242ULL};   /* range_proofs.rs:104, :196 */

/* ---------------------------------------------------------------------------------------------
 * circuits/falcon_ntt.rs:26-123 for one (sig, pk, hm)
 * status: 0 ok | 1 some input coefficient >= q (nothing written) | 2 l2 norm >= bound (witness written with the
 * truncated bit decomposition, exactly what the reference assigns when its #[cfg(not(test))] panic is compiled out)
 * ------------------------------------------------------------------------------------------- */
static int witness_one(int logn, const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                       int encoding, fr_t *wit, fr_t *inst, u192 *scratch)
{
    int n = 1 << logn;
    uint32_t s[1024], v[1024], pkn[1024], hmn[1024], sn[1024], vn[1024], b_sig[1024], b_v[1024];
    enc_t e; e.encoding = encoding;
    memset(&e.zero, 0, sizeof e.zero);
    fr_encode_small(&e.one, 1, encoding);

    for (int i = 0; i < n; i++) {
        if (sig[i] >= Q || pk[i] >= Q || hm[i] >= Q) return 1;
        s[i] = sn[i] = sig[i]; pkn[i] = pk[i]; hmn[i] = hm[i];
    }
    ntt_modq(hmn, logn);                                         /* hm_ntt :45 */
    ntt_modq(pkn, logn);                                         /* pk_ntt :51 */
    ntt_modq(sn, logn);
    for (int i = 0; i < n; i++) vn[i] = (hmn[i] + Q - sn[i] * pkn[i] % Q) % Q;
    memcpy(v, vn, sizeof(uint32_t) * n);
    intt_modq(v, logn);                                          /* v = hm - sig*pk :48-49 */

    /* instance_assignment = [1, pk_ntt, hm_ntt]   :63, :67 */
    inst[0] = e.one;
    for (int i = 0; i < n; i++) fr_encode_small(&inst[1 + i], pkn[i], encoding);
    for (int i = 0; i < n; i++) fr_encode_small(&inst[1 + n + i], hmn[i], encoding);

    fr_t *w = wit;
    for (int i = 0; i < n; i++) fr_encode_small(w++, s[i], encoding);        /* S0 sig_poly_vars :58-59 */
    for (int i = 0; i < n; i++) fr_encode_small(w++, v[i], encoding);        /* S1 v_vars :71 */
    for (int i = 0; i < n; i++) w = emit_ltq(w, &e, v[i]);                   /* S2 :73-77 */
    w = emit_ntt_circuit(w, &e, s, logn, b_sig, scratch);                    /* S3 :88-89 */
    w = emit_ntt_circuit(w, &e, v, logn, b_v, scratch);                      /* S4 :90-91 */
    for (int i = 0; i < n; i++) {                                            /* S5 :94-111 */
        uint64_t prod = (uint64_t)b_sig[i] * pkn[i];                         /* sig_ntt[i] * pk_ntt[i] */
        uint64_t ab = b_v[i] + prod;                                         /* arithmetics.rs:238 */
        uint64_t c = ab % Q, t = (ab - c) / Q;                               /* :242-243 */
        fr_encode_small(w++, prod, encoding);
        fr_encode_small(w++, t, encoding);
        fr_encode_small(w++, c, encoding);
        w = emit_ltq(w, &e, (uint32_t)c);
    }
    uint64_t norm = 0;                                                       /* S6 :116-120, v first then sig */
    for (int i = 0; i < n; i++) w = emit_l2_elem(w, &e, v[i], &norm);
    for (int i = 0; i < n; i++) w = emit_l2_elem(w, &e, s[i], &norm);
    w = logn == 9 ? emit_norm_512(w, &e, norm) : emit_norm_1024(w, &e, norm);   /* S7 :122 */
    return norm >= SIG_L2_BOUND[logn - 9] ? 2 : 0;
}

/* ---------------------------------------------------------------------------------------------
 * public entry points (mirroring the product's C ABI in include/frw.h, CPU only)
 * ------------------------------------------------------------------------------------------- */
typedef struct {
    int logn, encoding; size_t lo, hi;
    const uint16_t *sig, *pk, *hm; uint64_t *wit, *inst; int32_t *status;
} job_t;

static void *worker(void *arg)
{
    job_t *j = (job_t *)arg;
    oracle_layout_t L; frw_oracle_layout(j->logn, &L);
    size_t n = (size_t)L.n;
    u192 *scratch = (u192 *)malloc(sizeof(u192) * n);
    for (size_t i = j->lo; i < j->hi; i++)
        j->status[i] = witness_one(j->logn, j->sig + i * n, j->pk + i * n, j->hm + i * n, j->encoding,
                                   (fr_t *)j->wit + i * (size_t)L.num_witness,
                                   (fr_t *)j->inst + i * (size_t)L.num_instance, scratch);
    free(scratch);
    return NULL;
}

/* witness: batch x W x 4 u64; instance: batch x (2N+1) x 4 u64; status: batch.  threads <= 1 -> single thread. */
int frw_oracle_witness_ntt_verify(int logn, size_t batch, const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                                  int encoding, uint64_t *witness, uint64_t *instance, int32_t *status, int threads)
{
    if ((logn != 9 && logn != 10) || (encoding != 0 && encoding != 1)) return -1;
    pthread_once(&tables_once, init_tables);
    if (threads < 1) threads = 1;
    if ((size_t)threads > batch) threads = batch ? (int)batch : 1;
    job_t *jobs = (job_t *)calloc((size_t)threads, sizeof(job_t));
    pthread_t *th = (pthread_t *)calloc((size_t)threads, sizeof(pthread_t));
    for (int t = 0; t < threads; t++) {
        job_t j = {logn, encoding, batch * (size_t)t / threads, batch * (size_t)(t + 1) / threads, sig, pk, hm, witness, instance, status};
        jobs[t] = j;
        if (threads == 1) worker(&jobs[t]); else pthread_create(&th[t], NULL, worker, &jobs[t]);
    }
    if (threads > 1) for (int t = 0; t < threads; t++) pthread_join(th[t], NULL);
    free(jobs); free(th);
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * circuits/falcon_dual_ntt.rs:26-132: the signed-split variant.
 * Layout (allocation order), N = 1 << logn, nb = 50 | 52, W = 186 N + 4 + nb:
 *   sig.pos N | sig.neg N | pos*neg products N | is_zero [is_not_equal, multiplier] 2      dual_poly.rs:20-29
 *   v.pos N | v.neg N | products N | is_zero 2                                            falcon_dual_ntt.rs:73
 *   mod_q blocks of ntt_circuit(sig.pos), (sig.neg), (v.pos), (v.neg): 4 x 29 N           :85-92, dual_poly.rs:47-48
 *   per i: [sig_ntt.neg*pk_ntt, mod_q block (29)] [sig_ntt.pos*pk_ntt, mod_q block (29)]   :95-116    60 N
 *   squares of v.pos, v.neg, sig.pos, sig.neg   4 N                                        :121-129, misc.rs:55-65
 *   norm bound nb                                                                          :131
 * ------------------------------------------------------------------------------------------- */
static fr_t *emit_dual_alloc(fr_t *w, const enc_t *e, const uint32_t *pos, const uint32_t *neg, int n)
{
    uint64_t acc = 0;
    for (int i = 0; i < n; i++) fr_encode_small(w++, pos[i], e->encoding);
    for (int i = 0; i < n; i++) fr_encode_small(w++, neg[i], e->encoding);
    for (int i = 0; i < n; i++) { uint64_t pr = (uint64_t)pos[i] * neg[i]; acc += pr; fr_encode_small(w++, pr, e->encoding); }
    /* AllocatedFp::is_neq: is_not_equal = (acc != 0); multiplier = acc^-1 if so, else 1.  A DualPolynomial has
     * pos[i]*neg[i] = 0 for every i, so acc = 0 on this path. */
    put_bit(w++, e, acc != 0);
    *w++ = e->one;
    return w;
}

static fr_t *emit_mod_q_small(fr_t *w, const enc_t *e, uint64_t a)
{
    u192 v = {{a, 0, 0}};
    uint32_t b;
    return emit_mod_q(w, e, &v, &b);
}

static int witness_dual_one(int logn, const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                            int encoding, fr_t *wit, fr_t *inst, u192 *scratch)
{
    int n = 1 << logn;
    uint32_t sp[1024], sn[1024], vp[1024], vn[1024], pkn[1024], hmn[1024], s[1024], v[1024];
    uint32_t b_sp[1024], b_sn[1024], b_vp[1024], b_vn[1024];
    enc_t e; e.encoding = encoding;
    memset(&e.zero, 0, sizeof e.zero);
    fr_encode_small(&e.one, 1, encoding);
    for (int i = 0; i < n; i++) {
        if (sig[i] >= Q || pk[i] >= Q || hm[i] >= Q) return 1;
        s[i] = sig[i]; pkn[i] = pk[i]; hmn[i] = hm[i];
        sp[i] = sig[i] < 6144 ? sig[i] : 0;                        /* DualPolynomial: signed split */
        sn[i] = sig[i] < 6144 ? 0 : Q - sig[i];
    }
    ntt_modq(hmn, logn); ntt_modq(pkn, logn); ntt_modq(s, logn);
    for (int i = 0; i < n; i++) v[i] = (hmn[i] + Q - s[i] * pkn[i] % Q) % Q;      /* v = hm - uh_pos + uh_neg  :48-50 */
    intt_modq(v, logn);
    for (int i = 0; i < n; i++) { vp[i] = v[i] < 6144 ? v[i] : 0; vn[i] = v[i] < 6144 ? 0 : Q - v[i]; }   /* :51 */

    inst[0] = e.one;
    for (int i = 0; i < n; i++) fr_encode_small(&inst[1 + i], pkn[i], encoding);
    for (int i = 0; i < n; i++) fr_encode_small(&inst[1 + n + i], hmn[i], encoding);

    fr_t *w = wit;
    w = emit_dual_alloc(w, &e, sp, sn, n);                                       /* :60-61 */
    w = emit_dual_alloc(w, &e, vp, vn, n);                                       /* :73 */
    w = emit_ntt_circuit(w, &e, sp, logn, b_sp, scratch);                        /* :85-90 */
    w = emit_ntt_circuit(w, &e, sn, logn, b_sn, scratch);
    w = emit_ntt_circuit(w, &e, vp, logn, b_vp, scratch);                        /* :91-92 */
    w = emit_ntt_circuit(w, &e, vn, logn, b_vn, scratch);
    for (int i = 0; i < n; i++) {                                                /* :95-116 */
        uint64_t prl = (uint64_t)b_sn[i] * pkn[i];
        fr_encode_small(w++, prl, encoding);
        w = emit_mod_q_small(w, &e, hmn[i] + b_vn[i] + prl);
        uint64_t prr = (uint64_t)b_sp[i] * pkn[i];
        fr_encode_small(w++, prr, encoding);
        w = emit_mod_q_small(w, &e, b_vp[i] + prr);
    }
    uint64_t norm = 0;                                                           /* :121-129 */
    const uint32_t *parts[4] = {vp, vn, sp, sn};
    for (int k = 0; k < 4; k++)
        for (int i = 0; i < n; i++) { uint64_t sq = (uint64_t)parts[k][i] * parts[k][i]; norm += sq; fr_encode_small(w++, sq, encoding); }
    w = logn == 9 ? emit_norm_512(w, &e, norm) : emit_norm_1024(w, &e, norm);    /* :131 */
    return norm >= SIG_L2_BOUND[logn - 9] ? 2 : 0;
}

int frw_oracle_dual_num_witness(int logn) { return 186 * (1 << logn) + 4 + (logn == 9 ? 50 : 52); }

int frw_oracle_witness_dual_ntt_verify(int logn, size_t batch, const uint16_t *sig, const uint16_t *pk, const uint16_t *hm,
                                       int encoding, uint64_t *witness, uint64_t *instance, int32_t *status)
{
    if ((logn != 9 && logn != 10) || (encoding != 0 && encoding != 1)) return -1;
    pthread_once(&tables_once, init_tables);
    size_t n = (size_t)1 << logn, W = (size_t)frw_oracle_dual_num_witness(logn), I = 2 * n + 1;
    u192 *scratch = (u192 *)malloc(sizeof(u192) * n);
    for (size_t i = 0; i < batch; i++)
        status[i] = witness_dual_one(logn, sig + i * n, pk + i * n, hm + i * n, encoding, (fr_t *)witness + i * W,
                                     (fr_t *)instance + i * I, scratch);
    free(scratch);
    return 0;
}

/* gadgets/poly.rs:104-159 alone (the reference's "ntt conversion" row, examples/constraint_counts.rs:74-113):
 * witness: batch x 29N x 4 u64 (the N mod_q blocks); ntt_out: batch x N (the b values == NTTPolynomial::from) */
int frw_oracle_ntt_modq(int logn, size_t batch, const uint16_t *poly, int encoding, uint64_t *witness, uint16_t *ntt_out)
{
    if ((logn != 9 && logn != 10) || (encoding != 0 && encoding != 1)) return -1;
    pthread_once(&tables_once, init_tables);
    size_t n = (size_t)1 << logn;
    enc_t e; e.encoding = encoding; memset(&e.zero, 0, sizeof e.zero); fr_encode_small(&e.one, 1, encoding);
    u192 *scratch = (u192 *)malloc(sizeof(u192) * n);
    uint32_t in[1024], b[1024];
    for (size_t i = 0; i < batch; i++) {
        for (size_t k = 0; k < n; k++) { in[k] = poly[i * n + k]; if (in[k] >= Q) { free(scratch); return -2; } }
        emit_ntt_circuit((fr_t *)witness + i * 29 * n, &e, in, logn, b, scratch);
        for (size_t k = 0; k < n; k++) ntt_out[i * n + k] = (uint16_t)b[k];
    }
    free(scratch);
    return 0;
}

/* clear-text helpers exposed for the tests */
int frw_oracle_ntt_clear(int logn, const uint16_t *in, uint16_t *out, int inverse)
{
    if (logn != 9 && logn != 10) return -1;
    pthread_once(&tables_once, init_tables);
    uint32_t a[1024]; int n = 1 << logn;
    for (int i = 0; i < n; i++) a[i] = in[i] % Q;
    if (inverse) intt_modq(a, logn); else ntt_modq(a, logn);
    for (int i = 0; i < n; i++) out[i] = (uint16_t)a[i];
    return 0;
}

/* ---------------------------------------------------------------------------------------------
 * digest used by the tests to compare whole device buffers with the oracle without copying them back
 * ------------------------------------------------------------------------------------------- */
static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ULL;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ULL;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBULL;
    return x ^ (x >> 31);
}

/* order-independent 64-bit digest of a u64 buffer: sum_i splitmix64(w_i + i * golden) */
uint64_t frw_oracle_digest(const uint64_t *w, size_t nwords)
{
    uint64_t h = 0;
    for (size_t i = 0; i < nwords; i++) h += splitmix64(w[i] + i * 0x9E3779B97F4A7C15ULL);
    return h;
}
