// Host harness (shared library, ctypes) around the fourteen-limb Fq arithmetic and the XYZZ point formulas of the MSM
// kernels (falcon-r1cs_amd/csrc/frw_fq29.h), compiled with g++ through tests/cpp/hip_host.  tests/test_fq29_host.py
// drives it against Python integers and oracle/bls12_381.py.
#include <cstdint>
#include <cstring>
#include "frw_fq29.h"

using namespace frw;

static Fq29 load(const uint32_t *w) { Fq29 r; std::memcpy(r.l, w, sizeof r.l); return r; }
static void store(const Fq29 &a, uint32_t *w) { std::memcpy(w, a.l, sizeof a.l); }

extern "C" void t_fq(int op, const uint32_t *a, const uint32_t *b, uint32_t *out)
{
    switch (op) {
    case 0: store(fq_mul(load(a), load(b)), out); break;
    case 1: store(fq_add(load(a), load(b)), out); break;
    case 2: store(fq_sub<4>(load(a), load(b)), out); break;
    case 3: store(fq_sub<16>(load(a), load(b)), out); break;
    case 4: store(fq_sub<64>(load(a), load(b)), out); break;
    case 5: store(fq_canonical(load(a)), out); break;
    case 6: store(fq_reduce(load(a)), out); break;
    case 7: store(fq_from_ark(a), out); break;                 // a: 12 words
    case 8: fq_to_ark(load(a), out); break;                     // out: 12 words
    case 9: store(fq_inv(load(a)), out); break;
    case 10: out[0] = fq_is_zero(load(a)) ? 1u : 0u; break;
    case 11: store(fq_neg<4>(load(a)), out); break;
    case 12: { Fq29 t = fq_unpack(a); store(t, out); break; }   // a: 12 words
    case 13: fq_pack(load(a), out); break;
    case 14: store(fq_sqr(load(a)), out); break;
    case 15: store(fq_inv_fermat(load(a)), out); break;
    case 16: store(fq_inv_binary(load(a)), out); break;         // the one-round-at-a-time inverse (fq_inv's fallback)
    case 17: {                                                  // a: 12 canonical words; out[0..12): 1 / a mod q, out[12]: did the 31-round path finish
        uint32_t y[12], x[12];
        std::memcpy(y, a, sizeof y);
        out[12] = fq_inv_words31(y, x) ? 1u : 0u;
        std::memcpy(out, x, sizeof x);
        break;
    }
    case 18: out[0] = Q_NEG_INV31; break;
    }
}
// a b - c d with one reduction; which = 0: c < 4 q, 1: c < 16 q
extern "C" void t_fq_mul_sub(int which, const uint32_t *a, const uint32_t *b, const uint32_t *c, const uint32_t *d, uint32_t *out)
{
    if (which == 0) store(fq_mul_sub<4>(load(a), load(b), load(c), load(d)), out);
    else store(fq_mul_sub<16>(load(a), load(b), load(c), load(d)), out);
}

static G1Affine29 load_point(const uint32_t *w)                 // 24 words, ark-ff; zeros = infinity
{
    G1Affine29 p;
    uint32_t any = 0;
    for (int i = 0; i < 24; i++) any |= w[i];
    p.inf = any == 0;
    p.x = fq_from_ark(w);
    p.y = fq_from_ark(w + 12);
    return p;
}
static void store_point(const G1Xyzz &p, uint32_t *w)
{
    const G1Affine29 a = g1_to_affine(p);
    if (a.inf) { std::memset(w, 0, 96); return; }
    fq_to_ark(a.x, w);
    fq_to_ark(a.y, w + 12);
}

// op 0: p + q (mixed);  1: 2 p;  2: 2 p + 2 q through the general addition (both operands with ZZ != 1);
// 3: ((p + q) + p) + q, mixed, from the identity;  4: (2 p + q) - q  (the affine operand negated);
// 5: general addition of 2 p and 2 p (its doubling branch) ; 6: 2 p + (-(2 p))
extern "C" void t_g1(int op, const uint32_t *pw, const uint32_t *qw, uint32_t *out)
{
    const G1Affine29 p = load_point(pw), q = load_point(qw);
    G1Xyzz r = g1_identity();
    switch (op) {
    case 0: r = g1_add_affine(g1_from_affine(p), q); break;
    case 1: r = g1_double(g1_from_affine(p)); break;
    case 2: r = g1_add(g1_double(g1_from_affine(p)), g1_double(g1_from_affine(q))); break;
    case 3: r = g1_add_affine(g1_add_affine(g1_add_affine(g1_add_affine(r, p), q), p), q); break;
    case 4: {
        G1Affine29 nq = q;
        nq.y = fq_neg<4>(q.y);
        r = g1_add_affine(g1_add_affine(g1_double(g1_from_affine(p)), q), nq);
        break;
    }
    case 5: r = g1_add(g1_double(g1_from_affine(p)), g1_double(g1_from_affine(p))); break;
    case 6: {
        G1Xyzz d = g1_double(g1_from_affine(p)), n = d;
        n.y = fq_neg<16>(d.y);
        r = g1_add(d, n);
        break;
    }
    }
    store_point(r, out);
}

// ---- Fq2 and G2 ------------------------------------------------------------------------------------------------------------------
static Fq2_29 load2(const uint32_t *w) { Fq2_29 r; r.c0 = load(w); r.c1 = load(w + 14); return r; }
static void store2(const Fq2_29 &a, uint32_t *w) { store(a.c0, w); store(a.c1, w + 14); }
// a, b, out: 28 limbs (c0 then c1).  op 0 mul, 1 sqr, 2 inv, 3 is_zero (out[0])
extern "C" void t_fq2(int op, const uint32_t *a, const uint32_t *b, uint32_t *out)
{
    switch (op) {
    case 0: store2(fq2_mul(load2(a), load2(b)), out); break;
    case 1: store2(fq2_sqr(load2(a)), out); break;
    case 2: store2(Fq2Field::inv(load2(a)), out); break;
    case 3: out[0] = Fq2Field::is_zero(load2(a)) ? 1u : 0u; break;
    }
}
static G2Affine29 load_point2(const uint32_t *w)                // 48 words, ark-ff; zeros = infinity
{
    G2Affine29 p;
    uint32_t any = 0;
    for (int i = 0; i < 48; i++) any |= w[i];
    p.inf = any == 0;
    p.x = Fq2Field::from_ark(w);
    p.y = Fq2Field::from_ark(w + 24);
    return p;
}
static void store_point2(const G2Xyzz &p, uint32_t *w)
{
    const G2Affine29 a = pt_to_affine(p);
    if (a.inf) { std::memset(w, 0, 192); return; }
    Fq2Field::to_ark(a.x, w);
    Fq2Field::to_ark(a.y, w + 24);
}
// the same seven operations as t_g1, in G2; op 7: a chain of 40 mixed additions of q onto p (bounds must hold along it)
extern "C" void t_g2(int op, const uint32_t *pw, const uint32_t *qw, uint32_t *out)
{
    const G2Affine29 p = load_point2(pw), q = load_point2(qw);
    G2Xyzz r = pt_identity<Fq2Field>();
    switch (op) {
    case 0: r = pt_add_affine(pt_from_affine(p), q); break;
    case 1: r = pt_double(pt_from_affine(p)); break;
    case 2: r = pt_add(pt_double(pt_from_affine(p)), pt_double(pt_from_affine(q))); break;
    case 3: r = pt_add_affine(pt_add_affine(pt_add_affine(pt_add_affine(r, p), q), p), q); break;
    case 4: {
        G2Affine29 nq = q;
        nq.y = Fq2Field::neg<16>(q.y);
        r = pt_add_affine(pt_add_affine(pt_double(pt_from_affine(p)), q), nq);
        break;
    }
    case 5: r = pt_add(pt_double(pt_from_affine(p)), pt_double(pt_from_affine(p))); break;
    case 6: {
        G2Xyzz d = pt_double(pt_from_affine(p)), n = d;
        n.y = Fq2Field::neg<64>(d.y);
        r = pt_add(d, n);
        break;
    }
    case 7:
        r = pt_from_affine(p);
        for (int k = 0; k < 40; k++) r = k % 7 == 6 ? pt_double(r) : pt_add_affine(r, q);
        break;
    }
    store_point2(r, out);
}
// op 7 for G1 as well (through t_g1 would change its table): 40 steps from p with q
extern "C" void t_g1_chain(const uint32_t *pw, const uint32_t *qw, uint32_t *out)
{
    const G1Affine29 p = load_point(pw), q = load_point(qw);
    G1Xyzz r = g1_from_affine(p);
    for (int k = 0; k < 40; k++) r = k % 7 == 6 ? g1_double(r) : g1_add_affine(r, q);
    store_point(r, out);
}

// ---- k P on four lanes (frw_quad.h), the lanes run one after the other -------------------------------------------------------------
#include "frw_quad.h"
// p: 24 words ark-ff; pre: how many times p is doubled first (so that ZZ != 1, as a sum leaves it); k: k0 | k1, 4 words each
// (k = k0 + lambda k1); out: k (2^pre p), 24 words ark-ff.  Returns the number of slots the register file has | the number of additions that went through the complete one-lane formula << 8.
extern "C" int t_g1_quad_scale(const uint32_t *pw, int pre, const uint32_t *k, uint32_t *out)
{
    alignas(16) static uint32_t file[quad::NSLOTS * quad::SLOT_WORDS];
    G1Xyzz base = g1_from_affine(load_point(pw));
    for (int i = 0; i < pre; i++) base = g1_double(base);
    if (base.inf) { std::memset(out, 0, 96); return quad::NSLOTS; }
    quad::setup(file, base);
    struct Counting : quad::SerialExec {
        int degenerate_calls = 0;
        bool degenerate(uint32_t qx, uint32_t qy) { degenerate_calls++; return quad::SerialExec::degenerate(qx, qy); }
    } ex;
    ex.lds = file;
    uint32_t k0[4], k1[4];
    for (int i = 0; i < 4; i++) { k0[i] = k[i]; k1[i] = k[4 + i]; }
    const bool inf = quad::scalar_mul(ex, k0, k1);
    G1Xyzz r = quad::running_point(file, inf);
    if (inf) r = g1_identity();
    store_point(r, out);
    return quad::NSLOTS | ex.degenerate_calls << 8;
}
// the level programmes as data, for the structural checks of tests/test_fq29_host.py: out[step][lane][12] =
// ap[4] am[3] bp[2] bm dst kind; returns the number of steps
extern "C" int t_quad_programmes(uint8_t *out)
{
    const quad::Step *steps[] = {&quad::D1, &quad::D2, &quad::D3, &quad::A1, &quad::A2, &quad::A3, &quad::A4, &quad::COPY};
    int n = 0;
    for (const quad::Step *s : steps) {
        for (int l = 0; l < 4; l++) {
            const quad::Lane &L = s->l[l];
            uint8_t *o = out + (n * 4 + l) * 12;
            for (int k = 0; k < 4; k++) o[k] = L.ap[k];
            for (int k = 0; k < 3; k++) o[4 + k] = L.am[k];
            for (int k = 0; k < 2; k++) o[7 + k] = L.bp[k];
            o[9] = L.bm; o[10] = L.dst; o[11] = L.kind;
        }
        n++;
    }
    return n;
}
