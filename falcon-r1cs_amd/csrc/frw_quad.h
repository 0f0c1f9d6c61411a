// frw_quad.h -- k P for ONE G1 point on FOUR lanes of a wavefront: the scalar multiplications s g_a and r (g1_b - s delta1) at the
// end of a Groth16 proof (ark-groth16 0.3.0 prover.rs; examples/pok_sig.rs:30-47 of the reference makes one proof per call).
//
// The point exists only when its sum is done, so the 128 doublings (+ ~96 additions) of k P are a chain nothing else can hide: for
// a proof made alone they were 3 of its 5.6 ms with one thread per scalar multiplication (2,500 dependent field products).  The
// formulas themselves are not serial: an XYZZ doubling is 9 products in THREE dependent levels, a full addition 14 in FOUR.  Here
// four lanes take one product each per level.  What a lane multiplies is a small linear form of earlier results
// (p1 + p2 + p3 + p4 - m1 - m2 - m3, the K q that keeps a difference positive being one of the p), and all results live in a
// register file of 16-word slots in LDS: a level = every lane reads the slots its two operands name, multiplies, writes one slot.
// The four lanes of a quad run in lockstep (one wavefront, same branches), LDS serves a wavefront's requests in order: all reads
// of a level happen before its writes, and the next level sees them -- no barrier.  The level programmes below are data
// (`Step`), the operand selection by lane is a select between compile-time constants; the same programme runs on the host with
// the four lanes one after the other (tests/cpp/test_fq29.cpp, tests/test_fq29_host.py).
//
// State: X, ZZ, ZZZ and Y as the PAIR (AY, BY), Y = AY + 4 q - BY: both formulas end with Y3 = a b - c d, and keeping the two
// products apart saves the level that would subtract them.  Bounds (units of q): X < 10, AY, BY, ZZ, ZZZ < 2 -- what the
// one-lane formulas of frw_fq29.h keep, so the result goes to the same consumers.
#pragma once
#include "frw_fq29.h"

namespace frw {
namespace quad {

constexpr int SLOT_WORDS = 16;
enum : uint32_t {
    ZERO, K4, K8,                              // constants: 0, 4 q, 8 q
    X, AY, BY, ZZ, ZZZ,                        // the running point
    PX, PEX, PBX, PY, PNY, PZZ, PZZZ,          // P, phi(P), P + phi(P): x | beta x | -(1 + beta) x;  y | -y;  ZZ, ZZZ shared
    T0, T1, T2, T3, T4, T5, T6, T7, T8, T9, T10,
    DUMP, NSLOTS,
    QX = 62, QY = 63                           // the addend's x and y: resolved per bit to PX / PEX / PBX and PY / PNY
};
enum : uint8_t { MUL = 0, LIN = 1, SQR = 2 };  // result = A B | A | A A

struct Lane { uint8_t ap[4], am[3], bp[2], bm, dst, kind; };
struct Step { Lane l[4]; };
constexpr Lane IDLE = {{ZERO, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZERO, ZERO}, ZERO, DUMP, MUL};

// dbl-2008-s-1 (a = 0) in three levels.  U = 2 Y, V = U^2, W = U V, S = X V, M = 3 X^2, X3 = M^2 - 2 S,
// Y3 = M (S - X3) - W Y = M (3 S - M^2) - W Y, ZZ3 = V ZZ, ZZZ3 = W ZZZ
inline constexpr Step D1 = {{
    {{AY, AY, K8, ZERO}, {BY, BY, ZERO}, {ZERO, ZERO}, ZERO, T0, SQR},                   // V = (2 Y)^2
    {{X, X, X, ZERO}, {ZERO, ZERO, ZERO}, {X, ZERO}, ZERO, T1, MUL},                     // M = (3 X) X
    IDLE, IDLE}};
inline constexpr Step D2 = {{
    {{AY, AY, K8, ZERO}, {BY, BY, ZERO}, {T0, ZERO}, ZERO, T2, MUL},                     // W = U V
    {{X, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {T0, ZERO}, ZERO, T3, MUL},              // S = X V
    {{T1, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZERO, ZERO}, ZERO, T4, SQR},           // M^2
    {{T0, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZZ, ZERO}, ZERO, ZZ, MUL}}};           // ZZ3 = V ZZ
inline constexpr Step D3 = {{
    {{T3, T3, T3, K4}, {T4, ZERO, ZERO}, {T1, ZERO}, ZERO, AY, MUL},                     // (3 S - M^2) M
    {{T2, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {AY, K4}, BY, BY, MUL},                 // W Y
    {{T2, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZZZ, ZERO}, ZERO, ZZZ, MUL},           // ZZZ3 = W ZZZ
    {{T4, K4, ZERO, ZERO}, {T3, T3, ZERO}, {ZERO, ZERO}, ZERO, X, LIN}}};                // X3 = M^2 - 2 S   (< 6 q)
// add-2008-s in four levels, the addend Q = (QX, QY, PZZ, PZZZ).  U1 = X ZZq, U2 = QX ZZ, S1 = Y ZZZq, S2 = QY ZZZ, P = U2 - U1,
// R = S2 - S1, PP = P^2, PPP = P PP, Q = U1 PP, X3 = R^2 - PPP - 2 Q, Y3 = R (Q - X3) - S1 PPP, ZZ3 = ZZ ZZq PP, ZZZ3 = ZZZ ZZZq PPP;
// Q - X3 = 3 Q + PPP - R^2 and 3 Q + PPP = (2 U1 + U2) PP: the fourth lane of level three, so that level four needs no X3
inline constexpr Step A1 = {{
    {{X, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {PZZ, ZERO}, ZERO, T0, MUL},             // U1
    {{QX, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZZ, ZERO}, ZERO, T1, MUL},             // U2
    {{AY, K4, ZERO, ZERO}, {BY, ZERO, ZERO}, {PZZZ, ZERO}, ZERO, T2, MUL},               // S1
    {{QY, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZZZ, ZERO}, ZERO, T3, MUL}}};          // S2
inline constexpr Step A2 = {{
    {{T1, K4, ZERO, ZERO}, {T0, ZERO, ZERO}, {ZERO, ZERO}, ZERO, T4, SQR},               // PP     (A = P: the zero test reads it)
    {{T3, K4, ZERO, ZERO}, {T2, ZERO, ZERO}, {ZERO, ZERO}, ZERO, T5, SQR},               // R^2    (A = R)
    {{ZZ, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {PZZ, ZERO}, ZERO, T6, MUL},            // ZZ ZZq
    {{ZZZ, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {PZZZ, ZERO}, ZERO, T7, MUL}}};        // ZZZ ZZZq
inline constexpr Step A3 = {{
    {{T1, K4, ZERO, ZERO}, {T0, ZERO, ZERO}, {T4, ZERO}, ZERO, T8, MUL},                 // PPP
    {{T0, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {T4, ZERO}, ZERO, T9, MUL},             // Q
    {{T6, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {T4, ZERO}, ZERO, ZZ, MUL},             // ZZ3
    {{T0, T0, T1, ZERO}, {ZERO, ZERO, ZERO}, {T4, ZERO}, ZERO, T10, MUL}}};              // 3 Q + PPP
inline constexpr Step A4 = {{
    {{T10, K4, ZERO, ZERO}, {T5, ZERO, ZERO}, {T3, K4}, T2, AY, MUL},                    // (Q - X3) R
    {{T2, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {T8, ZERO}, ZERO, BY, MUL},             // S1 PPP
    {{T7, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {T8, ZERO}, ZERO, ZZZ, MUL},            // ZZZ3
    {{T5, K8, ZERO, ZERO}, {T8, T9, T9}, {ZERO, ZERO}, ZERO, X, LIN}}};                  // X3 = R^2 - PPP - 2 Q   (< 10 q)
// the first addition: the running point is the identity and becomes Q (BY is zero whenever the point is the identity)
inline constexpr Step COPY = {{
    {{QX, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZERO, ZERO}, ZERO, X, LIN},
    {{QY, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZERO, ZERO}, ZERO, AY, LIN},
    {{PZZ, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZERO, ZERO}, ZERO, ZZ, LIN},
    {{PZZZ, ZERO, ZERO, ZERO}, {ZERO, ZERO, ZERO}, {ZERO, ZERO}, ZERO, ZZZ, LIN}}};

// how many operand positions a level really uses (the rest are ZERO in every lane and cost nothing)
constexpr int used(const Step &s, int which)
{
    int n = 0;
    for (int l = 0; l < 4; l++) {
        const Lane &L = s.l[l];
        if (which == 0) { for (int k = 0; k < 4; k++) if (L.ap[k] != ZERO && k + 1 > n) n = k + 1; }
        else if (which == 1) { for (int k = 0; k < 3; k++) if (L.am[k] != ZERO && k + 1 > n) n = k + 1; }
        else if (which == 2) { for (int k = 0; k < 2; k++) if (L.bp[k] != ZERO && k + 1 > n) n = k + 1; }
        else if (L.bm != ZERO) n = 1;
    }
    return n;
}
constexpr bool any_kind(const Step &s, uint8_t kind) { return s.l[0].kind == kind || s.l[1].kind == kind || s.l[2].kind == kind || s.l[3].kind == kind; }
constexpr bool all_kind(const Step &s, uint8_t kind) { return s.l[0].kind == kind && s.l[1].kind == kind && s.l[2].kind == kind && s.l[3].kind == kind; }
constexpr bool has_alias(const Step &s) { return s.l[0].ap[0] >= QX || s.l[1].ap[0] >= QX || s.l[2].ap[0] >= QX || s.l[3].ap[0] >= QX; }

__host__ __device__ __forceinline__ uint32_t pick(uint32_t lane, uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3)
{
    return lane == 0 ? c0 : lane == 1 ? c1 : lane == 2 ? c2 : c3;
}
__host__ __device__ __forceinline__ void slot_accumulate(const uint32_t *lds, uint32_t idx, int32_t (&acc)[NLQ], bool minus)
{
    const uint4 *v = (const uint4 *)(lds + idx * SLOT_WORDS);
    const uint4 a = v[0], b = v[1], c = v[2], d = v[3];
    const uint32_t w[NLQ] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w, c.x, c.y, c.z, c.w, d.x, d.y};
#pragma unroll
    for (int i = 0; i < NLQ; i++) acc[i] = minus ? acc[i] - (int32_t)w[i] : acc[i] + (int32_t)w[i];
}
__host__ __device__ __forceinline__ Fq29 slot_load(const uint32_t *lds, uint32_t idx)
{
    int32_t acc[NLQ];
#pragma unroll
    for (int i = 0; i < NLQ; i++) acc[i] = 0;
    slot_accumulate(lds, idx, acc, false);
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = (uint32_t)acc[i];
    return r;
}
__host__ __device__ __forceinline__ void slot_store(uint32_t *lds, uint32_t idx, const Fq29 &a)
{
    uint4 *v = (uint4 *)(lds + idx * SLOT_WORDS);
    v[0] = make_uint4(a.l[0], a.l[1], a.l[2], a.l[3]);
    v[1] = make_uint4(a.l[4], a.l[5], a.l[6], a.l[7]);
    v[2] = make_uint4(a.l[8], a.l[9], a.l[10], a.l[11]);
    v[3] = make_uint4(a.l[12], a.l[13], 0u, 0u);
}
__host__ __device__ __forceinline__ Fq29 finish_form(int32_t (&acc)[NLQ])
{
    Fq29 r;
#pragma unroll
    for (int i = 0; i < NLQ; i++) r.l[i] = (uint32_t)acc[i];
    (void)fq_normalise_signed(r);                 // four normalised slots at most on the plus side: no limb leaves 32 bits
    return r;
}

// one level for one lane: the product (or linear form) this lane owes, and its first operand (what the zero test of an addition reads)
template <const Step &S>
__host__ __device__ __forceinline__ Fq29 compute(const uint32_t *lds, uint32_t lane, uint32_t qx, uint32_t qy, Fq29 &A)
{
    constexpr int AP = used(S, 0), AM = used(S, 1), BP = used(S, 2), BM = used(S, 3);
    int32_t acc[NLQ];
#pragma unroll
    for (int i = 0; i < NLQ; i++) acc[i] = 0;
#pragma unroll
    for (int k = 0; k < AP; k++) {
        uint32_t idx = pick(lane, S.l[0].ap[k], S.l[1].ap[k], S.l[2].ap[k], S.l[3].ap[k]);
        if (k == 0 && has_alias(S)) idx = idx == QX ? qx : idx == QY ? qy : idx;
        slot_accumulate(lds, idx, acc, false);
    }
#pragma unroll
    for (int k = 0; k < AM; k++) slot_accumulate(lds, pick(lane, S.l[0].am[k], S.l[1].am[k], S.l[2].am[k], S.l[3].am[k]), acc, true);
    A = finish_form(acc);
    if (all_kind(S, LIN)) return A;
    Fq29 B = A;
    if (!all_kind(S, SQR)) {
#pragma unroll
        for (int i = 0; i < NLQ; i++) acc[i] = 0;
#pragma unroll
        for (int k = 0; k < BP; k++) slot_accumulate(lds, pick(lane, S.l[0].bp[k], S.l[1].bp[k], S.l[2].bp[k], S.l[3].bp[k]), acc, false);
        if (BM) slot_accumulate(lds, pick(lane, S.l[0].bm, S.l[1].bm, S.l[2].bm, S.l[3].bm), acc, true);
        B = finish_form(acc);
        if (any_kind(S, SQR)) {
            const bool sq = pick(lane, S.l[0].kind, S.l[1].kind, S.l[2].kind, S.l[3].kind) == SQR;
#pragma unroll
            for (int i = 0; i < NLQ; i++) B.l[i] = sq ? A.l[i] : B.l[i];
        }
    }
    Fq29 r = fq_mul(A, B);
    if (any_kind(S, LIN)) {
        const bool lin = pick(lane, S.l[0].kind, S.l[1].kind, S.l[2].kind, S.l[3].kind) == LIN;
#pragma unroll
        for (int i = 0; i < NLQ; i++) r.l[i] = lin ? A.l[i] : r.l[i];
    }
    return r;
}
template <const Step &S> __host__ __device__ __forceinline__ void store(uint32_t *lds, uint32_t lane, const Fq29 &r)
{
    slot_store(lds, pick(lane, S.l[0].dst, S.l[1].dst, S.l[2].dst, S.l[3].dst), r);
}

// the register file before the first bit: constants, the three addends (their coordinates brought below 2 q where the point may
// leave unchanged: k = 1, lambda, 1 + lambda), the identity as running point.  One lane writes it.
__host__ __device__ inline void setup(uint32_t *lds, const XyzzT<FqField> &base)
{
    const Fq29 one = fq_const(FQ29_ONE);
    slot_store(lds, ZERO, fq_zero());
    slot_store(lds, K4, fq_const(KQ29_4));
    slot_store(lds, K8, fq_const(make_kq29(8)));
    const Fq29 endo_x = fq_mul(base.x, fq_const(G1_ENDO_BETA29));                       // < 2 q
    slot_store(lds, PX, base.x);                                                        // < 10 q as its sum left it
    slot_store(lds, PEX, endo_x);
    slot_store(lds, PBX, fq_mul(fq_neg<16>(fq_add(base.x, endo_x)), one));              // P + phi(P) = (-(1 + beta) x, -y): same y, horizontal chord
    slot_store(lds, PY, fq_mul(base.y, one));
    slot_store(lds, PNY, fq_mul(fq_neg<16>(base.y), one));
    slot_store(lds, PZZ, base.zz);
    slot_store(lds, PZZZ, base.zzz);
    for (uint32_t s = X; s <= ZZZ; s++) slot_store(lds, s, fq_zero());
}
__host__ __device__ inline XyzzT<FqField> running_point(const uint32_t *lds, bool inf)
{
    XyzzT<FqField> m;
    m.x = slot_load(lds, X);
    m.y = fq_sub<4>(slot_load(lds, AY), slot_load(lds, BY));                             // < 6 q
    m.zz = slot_load(lds, ZZ);
    m.zzz = slot_load(lds, ZZZ);
    m.inf = inf;
    return m;
}
// the addition whose operands are equal or opposite (P = 0 in the formulas above): the complete one-lane formula, by one lane.
// (With the canonical split of a scalar below the group order -- k0 < lambda, what frw_msm.hip's groth16_split_kernel hands over -- the running
// point never meets an addend: 2 v = +-a mod r has no solution among the prefixes of such a pair.  The halves that do reach this
// branch, tests/test_fq29_host.py, are non-canonical; it is here so that the kernel is complete for ANY pair of 128-bit halves.)
__host__ __device__ inline bool add_degenerate(uint32_t *lds, uint32_t qx, uint32_t qy)
{
    XyzzT<FqField> q;
    q.x = slot_load(lds, qx); q.y = slot_load(lds, qy); q.zz = slot_load(lds, PZZ); q.zzz = slot_load(lds, PZZZ);
    q.inf = false;
    const XyzzT<FqField> r = pt_add(running_point(lds, false), q);
    slot_store(lds, X, r.x); slot_store(lds, AY, r.y); slot_store(lds, BY, fq_zero());
    slot_store(lds, ZZ, r.zz); slot_store(lds, ZZZ, r.zzz);
    return r.inf;
}

// k P, k = k0 + lambda k1 (128 bits each; frw_msm.hip groth16_split_kernel), on an executor that runs a level for the four lanes of a quad:
//   ex.template step<S>(qx, qy)            one level
//   ex.template step_test<S>(qx, qy, pz)   one level; pz = lane 0's first operand is zero (mod q)
//   ex.degenerate(qx, qy) -> inf           add_degenerate by one lane, its answer to all
// The device executor is one lane of four (frw_msm.hip); the host executor runs the lanes one after the other, all reads of a
// level before its writes.  Returns whether the result is the identity; the point is in the slots X, AY, BY, ZZ, ZZZ.
template <class Exec> __host__ __device__ inline bool scalar_mul(Exec &ex, const uint32_t (&k0)[4], const uint32_t (&k1)[4])
{
    bool inf = true;
#pragma nounroll
    for (int bit = 127; bit >= 0; bit--) {
        if (!inf) {
            ex.template step<D1>(ZERO, ZERO);
            ex.template step<D2>(ZERO, ZERO);
            ex.template step<D3>(ZERO, ZERO);
        }
        const uint32_t sel = ((k0[bit >> 5] >> (bit & 31)) & 1u) | (((k1[bit >> 5] >> (bit & 31)) & 1u) << 1);
        if (sel == 0) continue;
        const uint32_t qx = sel == 1 ? (uint32_t)PX : sel == 2 ? (uint32_t)PEX : (uint32_t)PBX, qy = sel == 3 ? (uint32_t)PNY : (uint32_t)PY;
        if (inf) {
            ex.template step<COPY>(qx, qy);
            inf = false;
            continue;
        }
        ex.template step<A1>(qx, qy);
        bool pz = false;
        ex.template step_test<A2>(qx, qy, pz);
        if (pz) {
            inf = ex.degenerate(qx, qy);
            continue;
        }
        ex.template step<A3>(qx, qy);
        ex.template step<A4>(qx, qy);
    }
    return inf;
}

// the four lanes one after the other (host: tests)
struct SerialExec {
    uint32_t *lds;
    template <const Step &S> void step(uint32_t qx, uint32_t qy)
    {
        Fq29 r[4], a;
        for (uint32_t lane = 0; lane < 4; lane++) r[lane] = compute<S>(lds, lane, qx, qy, a);
        for (uint32_t lane = 0; lane < 4; lane++) store<S>(lds, lane, r[lane]);
    }
    template <const Step &S> void step_test(uint32_t qx, uint32_t qy, bool &pz)
    {
        Fq29 r[4], a[4];
        for (uint32_t lane = 0; lane < 4; lane++) r[lane] = compute<S>(lds, lane, qx, qy, a[lane]);
        for (uint32_t lane = 0; lane < 4; lane++) store<S>(lds, lane, r[lane]);
        pz = fq_is_zero(a[0]);
    }
    bool degenerate(uint32_t qx, uint32_t qy) { return add_degenerate(lds, qx, qy); }
};

}  // namespace quad
}  // namespace frw
