// frw_host.hpp -- C++ host-side mirror of the reference's interface for the hot path, over the C ABI of
// include/frw.h.  Header-only.
//
// The reference is compiled code (Rust); its toolchain is absent from the build image, so the host layer above the
// C ABI is written in C++ with the reference's names, argument meaning and error behaviour:
//
//   reference (falcon-r1cs/src/...)                               here (namespace frw::host)
//   ---------------------------------------------------------------------------------------------
//   ark_relations::r1cs::ConstraintSystem / ConstraintSystemRef    ConstraintSystem / ConstraintSystemRef
//   ark_r1cs_std FpVar<F>, Boolean<F>, AllocationMode              FpVar, Boolean, AllocationMode
//   gadgets/misc.rs        enforce_decompose, l2_norm_var, ntt_param_var        same names
//   gadgets/range_proofs.rs enforce_less_than_q, is_less_than_6144,
//                           enforce_less_than_norm_bound                          same names
//   gadgets/arithmetics.rs mod_q, add_mod                                          same names
//   gadgets/poly.rs        PolyVar / NTTPolyVar ::alloc_vars, ::ntt_circuit        same names
//   circuits/falcon_ntt.rs FalconNTTVerificationCircuit::{build_circuit, generate_constraints}   same names
//
// Division of labour (BASELINE north_star): the host allocates variable indices and emits constraints exactly as
// arkworks does; it NEVER computes a witness value.  Every `new_witness_variable` takes its value from the engine
// (libfrw.so, HIP kernels): the full circuit from one frw_witness_ntt_verify call, a gadget called on its own from
// frw_gadget / frw_ntt_modq.  In setup mode values are F::one(), as in the reference (arithmetics.rs:121-125).
// Without an engine value in prove mode the gadget throws SynthesisError::AssignmentMissing -- there is no CPU path.
//
// What IS evaluated on the host: linear combinations (needed by is_satisfied and to hand a stand-alone gadget its
// input value), exactly the `value` bookkeeping ark-r1cs-std's AllocatedFp carries.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "../../../include/frw.h"

namespace frw::host {

// ---------------------------------------------------------------------------------------------------------------
// Fr: BLS12-381 scalar field element in ark-ff's Fp256 representation (Montgomery, 4 x u64 LE)
// ---------------------------------------------------------------------------------------------------------------
struct Fr {
    uint64_t l[4];
    using u128 = unsigned __int128;
    static constexpr uint64_t P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
    static constexpr uint64_t R1[4] = {0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL};
    static constexpr uint64_t R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
    static constexpr uint64_t INV = 0xfffffffeffffffffULL;

    static Fr zero() { return Fr{{0, 0, 0, 0}}; }
    static Fr one() { return Fr{{R1[0], R1[1], R1[2], R1[3]}}; }
    static Fr from_montgomery(const uint64_t *limbs) { Fr r; std::memcpy(r.l, limbs, 32); return r; }
    // canonical little-endian limbs (value < p) -> element
    static Fr from_canonical(const uint64_t limbs[4]) { Fr a; std::memcpy(a.l, limbs, 32); Fr r2{{R2[0], R2[1], R2[2], R2[3]}}; return a * r2; }
    static Fr from(uint64_t x) { uint64_t c[4] = {x, 0, 0, 0}; return from_canonical(c); }
    void to_canonical(uint64_t out[4]) const { Fr o{{1, 0, 0, 0}}; Fr r = (*this) * o; std::memcpy(out, r.l, 32); }
    bool is_zero() const { return !(l[0] | l[1] | l[2] | l[3]); }
    bool operator==(const Fr &o) const { return !std::memcmp(l, o.l, 32); }
    bool operator!=(const Fr &o) const { return !(*this == o); }

    Fr operator+(const Fr &o) const
    {
        Fr r; u128 c = 0;
        for (int i = 0; i < 4; i++) { c += (u128)l[i] + o.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
        r.reduce_once();
        return r;
    }
    Fr operator-(const Fr &o) const
    {
        Fr r; u128 b = 0;
        for (int i = 0; i < 4; i++) { u128 x = (u128)l[i] - o.l[i] - (uint64_t)b; r.l[i] = (uint64_t)x; b = (x >> 64) & 1; }
        if (b) { u128 c = 0; for (int i = 0; i < 4; i++) { c += (u128)r.l[i] + P[i]; r.l[i] = (uint64_t)c; c >>= 64; } }
        return r;
    }
    Fr operator-() const { return zero() - *this; }
    Fr operator*(const Fr &o) const
    {
        uint64_t t[6] = {0, 0, 0, 0, 0, 0};
        for (int i = 0; i < 4; i++) {
            u128 c = 0;
            for (int j = 0; j < 4; j++) { c += (u128)l[j] * o.l[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[4] = (uint64_t)c; t[5] = (uint64_t)(c >> 64);
            uint64_t m = t[0] * INV;
            c = (u128)m * P[0] + t[0]; c >>= 64;
            for (int j = 1; j < 4; j++) { c += (u128)m * P[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; }
            c += t[4]; t[3] = (uint64_t)c; t[4] = t[5] + (uint64_t)(c >> 64);
        }
        Fr r{{t[0], t[1], t[2], t[3]}};
        if (t[4]) { u128 b = 0; for (int i = 0; i < 4; i++) { u128 x = (u128)r.l[i] - P[i] - (uint64_t)b; r.l[i] = (uint64_t)x; b = (x >> 64) & 1; } }
        else r.reduce_once();
        return r;
    }
    Fr doubled() const { return *this + *this; }
    Fr pow(uint64_t e) const { Fr r = one(), b = *this; while (e) { if (e & 1) r = r * b; b = b * b; e >>= 1; } return r; }

private:
    void reduce_once()
    {
        uint64_t d[4]; u128 b = 0;
        for (int i = 0; i < 4; i++) { u128 x = (u128)l[i] - P[i] - (uint64_t)b; d[i] = (uint64_t)x; b = (x >> 64) & 1; }
        if (!b) std::memcpy(l, d, 32);
    }
};

// ---------------------------------------------------------------------------------------------------------------
// errors (ark_relations::r1cs::SynthesisError) and the engine handle
// ---------------------------------------------------------------------------------------------------------------
struct SynthesisError : std::runtime_error {
    enum Kind { AssignmentMissing, Unsatisfiable, Engine } kind;
    SynthesisError(Kind k, const std::string &what) : std::runtime_error(what), kind(k) {}
};

// RAII over frw_ctx; throws when no HIP device is usable (there is no CPU path).
class Engine {
public:
    explicit Engine(int device = 0)
    {
        int rc = frw_ctx_create(device, &ctx_);
        if (rc != FRW_OK) throw SynthesisError(SynthesisError::Engine, std::string("frw_ctx_create: ") + frw_strerror(rc) + "; " + frw_last_error());
    }
    ~Engine() { frw_ctx_destroy(ctx_); }
    Engine(const Engine &) = delete;
    Engine &operator=(const Engine &) = delete;
    frw_ctx *get() const { return ctx_; }
private:
    frw_ctx *ctx_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------
// ark_relations::r1cs::{Variable, LinearCombination, ConstraintSystem}
// ---------------------------------------------------------------------------------------------------------------
enum class VarKind : uint8_t { Zero, One, Instance, Witness, SymbolicLc };
struct Variable { VarKind kind; uint32_t index; };
inline Variable VarOne() { return {VarKind::One, 0}; }
using LinearCombination = std::vector<std::pair<Fr, Variable>>;
enum class AllocationMode { Constant, Input, Witness };

// ark_relations::r1cs::ConstraintMatrices: A, B, C with every symbolic LC inlined (what a prover ingests after
// cs.finalize() / cs.to_matrices(), examples/pok_sig.rs:30-32).  Column j < num_instance_variables is instance
// variable j (column 0 = the constant one); column num_instance_variables + k is witness k.  Rows are sorted by
// column, duplicate columns summed, zero coefficients dropped.
struct ConstraintMatrices {
    using Row = std::vector<std::pair<Fr, uint32_t>>;
    size_t num_instance_variables = 0, num_witness_variables = 0, num_constraints = 0;
    std::vector<Row> a, b, c;
    size_t non_zero(const std::vector<Row> &m) const { size_t n = 0; for (const auto &r : m) n += r.size(); return n; }
    // (A z) o (B z) == C z for z = instance || witness
    bool is_satisfied(const std::vector<Fr> &instance, const std::vector<Fr> &witness) const
    {
        auto dot = [&](const Row &r) { Fr acc = Fr::zero(); for (const auto &t : r) acc = acc + t.first * (t.second < instance.size() ? instance[t.second] : witness[t.second - instance.size()]); return acc; };
        for (size_t i = 0; i < num_constraints; i++) if (dot(a[i]) * dot(b[i]) != dot(c[i])) return false;
        return true;
    }
    // Binary file: "FRWR1CS1", u64 {num_instance, num_witness, num_constraints, nnz_a, nnz_b, nnz_c}, then per matrix
    // CSR: u64 row_ptr[C+1], u32 col[nnz], u64 value[nnz][4] (canonical little-endian limbs).
    bool write(const char *path) const
    {
        FILE *f = std::fopen(path, "wb");
        if (!f) return false;
        const uint64_t hdr[6] = {num_instance_variables, num_witness_variables, num_constraints, non_zero(a), non_zero(b), non_zero(c)};
        bool ok = std::fwrite("FRWR1CS1", 1, 8, f) == 8 && std::fwrite(hdr, 8, 6, f) == 6;
        for (const auto *m : {&a, &b, &c}) {
            std::vector<uint64_t> ptr{0};
            std::vector<uint32_t> col;
            std::vector<uint64_t> val;
            for (const Row &r : *m) {
                for (const auto &t : r) { uint64_t cl[4]; t.first.to_canonical(cl); col.push_back(t.second); val.insert(val.end(), cl, cl + 4); }
                ptr.push_back(col.size());
            }
            ok = ok && std::fwrite(ptr.data(), 8, ptr.size(), f) == ptr.size() && std::fwrite(col.data(), 4, col.size(), f) == col.size() &&
                 std::fwrite(val.data(), 8, val.size(), f) == val.size();
        }
        return std::fclose(f) == 0 && ok;
    }
};

class ConstraintSystem;
using ConstraintSystemRef = std::shared_ptr<ConstraintSystem>;

class ConstraintSystem {
public:
    static ConstraintSystemRef new_ref() { return std::make_shared<ConstraintSystem>(); }
    ConstraintSystem() { instance_assignment.push_back(Fr::one()); }

    // ---- mode / engine -------------------------------------------------------------------------------------
    void set_setup_mode(bool s) { setup_ = s; }
    bool is_in_setup_mode() const { return setup_; }
    void attach_engine(const Engine *e, bool strict = true) { engine_ = e; strict_ = strict; }
    const Engine *engine() const { return engine_; }
    bool strict() const { return strict_; }

    // Values the engine produced, in allocation order (Montgomery limbs).  Gadgets pop from here.
    void push_feed(const uint64_t *mont_limbs, size_t count)
    {
        for (size_t i = 0; i < count; i++) feed_.push_back(Fr::from_montgomery(mont_limbs + 4 * i));
    }
    size_t feed_remaining() const { return feed_.size() - feed_pos_; }
    Fr pop_feed(const char *who)
    {
        if (setup_) return Fr::one();
        if (feed_pos_ >= feed_.size())
            throw SynthesisError(SynthesisError::AssignmentMissing, std::string(who) + ": no engine value (AssignmentMissing; there is no CPU path)");
        return feed_[feed_pos_++];
    }

    // ---- allocation ----------------------------------------------------------------------------------------
    Variable new_input_variable(const Fr &v) { instance_assignment.push_back(v); return {VarKind::Instance, (uint32_t)instance_assignment.size() - 1}; }
    Variable new_witness_variable(const Fr &v) { witness_assignment.push_back(v); return {VarKind::Witness, (uint32_t)witness_assignment.size() - 1}; }
    Variable new_lc(LinearCombination lc) { lcs_.push_back(std::move(lc)); return {VarKind::SymbolicLc, (uint32_t)lcs_.size() - 1}; }
    void enforce_constraint(LinearCombination a, LinearCombination b, LinearCombination c)
    {
        a_.push_back(std::move(a)); b_.push_back(std::move(b)); c_.push_back(std::move(c));
    }

    size_t num_instance_variables() const { return instance_assignment.size(); }
    size_t num_witness_variables() const { return witness_assignment.size(); }
    size_t num_constraints() const { return a_.size(); }
    size_t num_linear_combinations() const { return lcs_.size(); }

    // ---- satisfaction --------------------------------------------------------------------------------------
    std::optional<size_t> which_is_unsatisfied()
    {
        if (setup_) throw SynthesisError(SynthesisError::AssignmentMissing, "is_satisfied in setup mode");
        lc_vals_.clear();
        lc_vals_.reserve(lcs_.size());
        for (const auto &lc : lcs_) lc_vals_.push_back(eval(lc));       // an LC only refers to earlier LCs
        for (size_t i = 0; i < a_.size(); i++)
            if (eval(a_[i]) * eval(b_[i]) != eval(c_[i])) return i;
        return std::nullopt;
    }
    bool is_satisfied() { return !which_is_unsatisfied().has_value(); }

    // ConstraintSystem::to_matrices() after inline_all_lcs(): structure only, valid in setup mode too
    ConstraintMatrices to_matrices() const
    {
        using Row = ConstraintMatrices::Row;
        const uint32_t ni = (uint32_t)instance_assignment.size();
        auto normalise = [](Row &r) {
            std::sort(r.begin(), r.end(), [](const auto &x, const auto &y) { return x.second < y.second; });
            Row out;
            for (const auto &t : r) {
                if (!out.empty() && out.back().second == t.second) out.back().first = out.back().first + t.first;
                else out.push_back(t);
            }
            out.erase(std::remove_if(out.begin(), out.end(), [](const auto &t) { return t.first.is_zero(); }), out.end());
            r.swap(out);
        };
        std::vector<Row> inl(lcs_.size());                     // expansion of every symbolic LC, in creation order
        auto expand = [&](const LinearCombination &lc) {
            Row r;
            for (const auto &t : lc) {
                switch (t.second.kind) {
                case VarKind::Zero: break;
                case VarKind::One: r.push_back({t.first, 0u}); break;
                case VarKind::Instance: r.push_back({t.first, t.second.index}); break;
                case VarKind::Witness: r.push_back({t.first, ni + t.second.index}); break;
                case VarKind::SymbolicLc:
                    for (const auto &u : inl[t.second.index]) r.push_back({t.first * u.first, u.second});
                    break;
                }
            }
            normalise(r);
            return r;
        };
        for (size_t i = 0; i < lcs_.size(); i++) inl[i] = expand(lcs_[i]);
        ConstraintMatrices m;
        m.num_instance_variables = instance_assignment.size();
        m.num_witness_variables = witness_assignment.size();
        m.num_constraints = a_.size();
        for (size_t i = 0; i < a_.size(); i++) { m.a.push_back(expand(a_[i])); m.b.push_back(expand(b_[i])); m.c.push_back(expand(c_[i])); }
        return m;
    }

    const std::vector<LinearCombination> &a() const { return a_; }
    const std::vector<LinearCombination> &b() const { return b_; }
    const std::vector<LinearCombination> &c() const { return c_; }

    std::vector<Fr> instance_assignment;   // [1, ...]   (public fields, as in ark-relations)
    std::vector<Fr> witness_assignment;

private:
    Fr value_of(Variable v) const
    {
        switch (v.kind) {
        case VarKind::Zero: return Fr::zero();
        case VarKind::One: return Fr::one();
        case VarKind::Instance: return instance_assignment[v.index];
        case VarKind::Witness: return witness_assignment[v.index];
        default: return lc_vals_[v.index];
        }
    }
    Fr eval(const LinearCombination &lc) const
    {
        Fr acc = Fr::zero();
        for (const auto &t : lc) acc = acc + t.first * value_of(t.second);
        return acc;
    }
    bool setup_ = false, strict_ = true;
    const Engine *engine_ = nullptr;
    std::vector<Fr> feed_;
    size_t feed_pos_ = 0;
    std::vector<LinearCombination> lcs_, a_, b_, c_;
    std::vector<Fr> lc_vals_;
};

// ---------------------------------------------------------------------------------------------------------------
// ark_r1cs_std::fields::fp::FpVar (Constant | Var(AllocatedFp))
// ---------------------------------------------------------------------------------------------------------------
class Boolean;

class FpVar {
public:
    static FpVar new_constant(const ConstraintSystemRef &, const Fr &c) { FpVar v; v.value_ = c; return v; }
    // caller-supplied value (a circuit INPUT chosen by the caller, as the reference's tests do with new_witness(|| Ok(a)))
    static FpVar new_witness(const ConstraintSystemRef &cs, const Fr &value) { return var(cs, value, cs->new_witness_variable(value)); }
    static FpVar new_input(const ConstraintSystemRef &cs, const Fr &value) { return var(cs, value, cs->new_input_variable(value)); }
    // gadget-internal witness: the value comes from the engine (or F::one() in setup mode)
    static FpVar new_witness_from_engine(const ConstraintSystemRef &cs, const char *who) { return new_witness(cs, cs->pop_feed(who)); }

    bool is_constant() const { return !cs_; }
    const Fr &value() const { return value_; }
    const ConstraintSystemRef &cs() const { return cs_; }
    Variable variable() const { return variable_; }

    // impl_ops!: Constant (+,-,*) Var produce symbolic LCs only; Var * Var allocates the product witness
    friend FpVar operator+(const FpVar &a, const FpVar &b)
    {
        if (a.is_constant() && b.is_constant()) return constant(a.value_ + b.value_);
        if (a.is_constant()) return b.add_constant(a.value_);
        if (b.is_constant()) return a.add_constant(b.value_);
        return var(a.cs_, a.value_ + b.value_, a.cs_->new_lc({{Fr::one(), a.variable_}, {Fr::one(), b.variable_}}));
    }
    friend FpVar operator-(const FpVar &a, const FpVar &b)
    {
        if (a.is_constant() && b.is_constant()) return constant(a.value_ - b.value_);
        if (b.is_constant()) return a.add_constant(-b.value_);
        if (a.is_constant()) return b.add_constant(-a.value_).negate();
        return var(a.cs_, a.value_ - b.value_, a.cs_->new_lc({{Fr::one(), a.variable_}, {-Fr::one(), b.variable_}}));
    }
    friend FpVar operator*(const FpVar &a, const FpVar &b)
    {
        if (a.is_constant() && b.is_constant()) return constant(a.value_ * b.value_);
        if (a.is_constant()) return b.mul_constant(a.value_);
        if (b.is_constant()) return a.mul_constant(b.value_);
        // AllocatedFp::mul: product witness + a * b = product
        FpVar prod = new_witness_from_engine(a.cs_, "FpVar*FpVar product");
        a.cs_->enforce_constraint({{Fr::one(), a.variable_}}, {{Fr::one(), b.variable_}}, {{Fr::one(), prod.variable_}});
        return prod;
    }
    FpVar double_() const
    {
        if (is_constant()) return constant(value_.doubled());
        return var(cs_, value_.doubled(), cs_->new_lc({{Fr::one(), variable_}, {Fr::one(), variable_}}));
    }
    FpVar negate() const { return var(cs_, -value_, cs_->new_lc({{-Fr::one(), variable_}})); }

    // EqGadget::enforce_equal: (self - other) * 1 = 0
    void enforce_equal(const FpVar &o) const
    {
        if (is_constant() && o.is_constant()) {
            if (value_ != o.value_) throw SynthesisError(SynthesisError::Unsatisfiable, "constant != constant");
            return;
        }
        const ConstraintSystemRef &cs = is_constant() ? o.cs_ : cs_;
        Variable x = is_constant() ? cs->new_lc({{value_, VarOne()}}) : variable_;
        Variable y = o.is_constant() ? cs->new_lc({{o.value_, VarOne()}}) : o.variable_;
        cs->enforce_constraint({{Fr::one(), x}, {-Fr::one(), y}}, {{Fr::one(), VarOne()}}, {});
    }

    Boolean is_zero() const;       // FieldVar::is_zero = is_eq(&zero): AllocatedFp::is_neq, 2 witnesses / 3 constraints
    static FpVar from_boolean(const Boolean &b);
    static FpVar conditionally_select(const Boolean &cond, const FpVar &t, const FpVar &f);

    static FpVar constant(const Fr &c) { FpVar v; v.value_ = c; return v; }
    static FpVar var(const ConstraintSystemRef &cs, const Fr &value, Variable variable)
    {
        FpVar v; v.cs_ = cs; v.value_ = value; v.variable_ = variable; return v;
    }

private:
    FpVar add_constant(const Fr &c) const
    {
        if (c.is_zero()) return *this;
        return var(cs_, value_ + c, cs_->new_lc({{Fr::one(), variable_}, {c, VarOne()}}));
    }
    FpVar mul_constant(const Fr &c) const { return var(cs_, value_ * c, cs_->new_lc({{c, variable_}})); }

    ConstraintSystemRef cs_;            // null => Constant
    Fr value_ = Fr::zero();
    Variable variable_{VarKind::Zero, 0};
};

// ---------------------------------------------------------------------------------------------------------------
// ark_r1cs_std::bits::boolean::Boolean (Is | Not | Constant) over AllocatedBool
// ---------------------------------------------------------------------------------------------------------------
class Boolean {
public:
    enum Kind { Is, Not, Constant };
    static Boolean constant(bool b) { Boolean r; r.kind_ = Constant; r.bit_ = b; return r; }
    static Boolean TRUE_() { return constant(true); }
    static Boolean FALSE_() { return constant(false); }

    // Boolean::new_witness: witness (value from the engine) + booleanity (1 - a) * a = 0
    static Boolean new_witness_from_engine(const ConstraintSystemRef &cs, const char *who)
    {
        Fr v = cs->pop_feed(who);
        Variable x = cs->new_witness_variable(v);
        cs->enforce_constraint({{Fr::one(), VarOne()}, {-Fr::one(), x}}, {{Fr::one(), x}}, {});
        return is(cs, x, v == Fr::one());
    }

    Kind kind() const { return kind_; }
    bool value() const { return kind_ == Constant ? bit_ : (kind_ == Is ? bit_ : !bit_); }
    const ConstraintSystemRef &cs() const { return cs_; }

    LinearCombination lc() const
    {
        if (kind_ == Constant) return bit_ ? LinearCombination{{Fr::one(), VarOne()}} : LinearCombination{};
        if (kind_ == Is) return {{Fr::one(), var_}};
        return {{Fr::one(), VarOne()}, {-Fr::one(), var_}};
    }
    Boolean not_() const
    {
        Boolean r = *this;
        if (kind_ == Constant) r.bit_ = !bit_; else r.kind_ = kind_ == Is ? Not : Is;
        return r;
    }
    // is_eq against a constant: self.xor(Constant(c)).not() -- no allocation
    Boolean is_eq(const Boolean &c) const
    {
        if (c.kind_ != Constant) throw std::logic_error("Boolean::is_eq: only constants are used on this path");
        return c.bit_ ? *this : not_();
    }

    Boolean and_(const Boolean &o) const
    {
        if (kind_ == Constant) return bit_ ? o : constant(false);
        if (o.kind_ == Constant) return o.bit_ ? *this : constant(false);
        if (kind_ == Is && o.kind_ == Not) return gate(*this, o, GateAndNot);
        if (kind_ == Not && o.kind_ == Is) return gate(o, *this, GateAndNot);
        if (kind_ == Not && o.kind_ == Not) return gate(*this, o, GateNor);
        return gate(*this, o, GateAnd);
    }
    Boolean or_(const Boolean &o) const
    {
        if (kind_ == Constant) return bit_ ? constant(true) : o;
        if (o.kind_ == Constant) return o.bit_ ? constant(true) : *this;
        if (kind_ == Is && o.kind_ == Is) return gate(*this, o, GateOr);
        // (a @ Is, b @ Not) | (b @ Not, a @ Is) | (b @ Not, a @ Not)  =>  !(!a & !b)
        const Boolean &a = kind_ == Is ? *this : o;
        const Boolean &b = kind_ == Is ? o : *this;
        return a.not_().and_(b.not_()).not_();
    }
    static Boolean kary_and(const std::vector<Boolean> &bits, size_t lo, size_t hi)
    {
        Boolean cur = bits[lo];
        for (size_t i = lo + 1; i < hi; i++) cur = cur.and_(bits[i]);
        return cur;
    }
    static Boolean kary_or(const std::vector<Boolean> &bits, size_t lo, size_t hi)
    {
        Boolean cur = bits[lo];
        for (size_t i = lo + 1; i < hi; i++) cur = cur.or_(bits[i]);
        return cur;
    }
    // enforce_equal(&Boolean::TRUE/FALSE): difference * 1 = 0
    void enforce_equal(const Boolean &c) const
    {
        if (c.kind_ != Constant) throw std::logic_error("Boolean::enforce_equal: only constants are used on this path");
        if (kind_ == Constant) {
            if (bit_ != c.bit_) throw SynthesisError(SynthesisError::AssignmentMissing, "false != true");
            return;
        }
        LinearCombination one_minus{{Fr::one(), VarOne()}, {-Fr::one(), var_}}, self{{Fr::one(), var_}};
        bool want_var_zero = (kind_ == Is) != c.bit_;      // Is==TRUE -> 1-a ; Is==FALSE -> a ; Not==TRUE -> a ; Not==FALSE -> 1-a
        cs_->enforce_constraint(want_var_zero ? self : one_minus, {{Fr::one(), VarOne()}}, {});
    }

private:
    enum GateKind { GateOr, GateAnd, GateAndNot, GateNor };
    static Boolean is(const ConstraintSystemRef &cs, Variable v, bool bit)
    {
        Boolean r; r.kind_ = Is; r.cs_ = cs; r.var_ = v; r.bit_ = bit; return r;
    }
    // AllocatedBool::{or, and, and_not, nor}: result witness WITHOUT booleanity check + one constraint.
    // x, y are taken as their underlying AllocatedBool (the Is/Not wrapper was resolved by the caller).
    static Boolean gate(const Boolean &x, const Boolean &y, GateKind g)
    {
        const ConstraintSystemRef &cs = x.cs_;
        Fr v = cs->pop_feed("boolean gate");
        Variable r = cs->new_witness_variable(v);
        LinearCombination X{{Fr::one(), x.var_}}, Y{{Fr::one(), y.var_}}, Rr{{Fr::one(), r}};
        LinearCombination nX{{Fr::one(), VarOne()}, {-Fr::one(), x.var_}}, nY{{Fr::one(), VarOne()}, {-Fr::one(), y.var_}},
            nR{{Fr::one(), VarOne()}, {-Fr::one(), r}};
        switch (g) {
        case GateOr: cs->enforce_constraint(nX, nY, nR); break;        // (1-a)(1-b) = 1-r
        case GateAnd: cs->enforce_constraint(X, Y, Rr); break;         // a b = r
        case GateAndNot: cs->enforce_constraint(X, nY, Rr); break;     // a (1-b) = r
        case GateNor: cs->enforce_constraint(nX, nY, Rr); break;       // (1-a)(1-b) = r
        }
        return is(cs, r, v == Fr::one());
    }
    Kind kind_ = Constant;
    bool bit_ = false;                 // Constant: the constant; Is/Not: value of the underlying AllocatedBool
    ConstraintSystemRef cs_;
    Variable var_{VarKind::Zero, 0};
};

inline FpVar FpVar::from_boolean(const Boolean &b)
{
    if (b.kind() == Boolean::Constant) return constant(b.value() ? Fr::one() : Fr::zero());
    return var(b.cs(), b.value() ? Fr::one() : Fr::zero(), b.cs()->new_lc(b.lc()));
}

// AllocatedFp::is_neq(c, v) with c = new_constant(0): is_not_equal (Boolean witness, with booleanity) and the
// multiplier (bare witness variable), then (c - v) * multiplier = is_not_equal and (c - v) * !is_not_equal = 0.
inline Boolean FpVar::is_zero() const
{
    if (is_constant()) return Boolean::constant(value_.is_zero());
    Variable zero = cs_->new_lc({{Fr::zero(), VarOne()}});
    Boolean is_not_equal = Boolean::new_witness_from_engine(cs_, "is_zero is_not_equal");
    Variable mult = cs_->new_witness_variable(cs_->pop_feed("is_zero multiplier"));
    LinearCombination diff{{Fr::one(), zero}, {-Fr::one(), variable_}};
    cs_->enforce_constraint(diff, {{Fr::one(), mult}}, is_not_equal.lc());
    cs_->enforce_constraint(diff, is_not_equal.not_().lc(), {});
    return is_not_equal.not_();
}

inline FpVar FpVar::conditionally_select(const Boolean &cond, const FpVar &t, const FpVar &f)
{
    if (cond.kind() == Boolean::Constant) return cond.value() ? t : f;
    const ConstraintSystemRef &cs = cond.cs();
    Variable tv = t.is_constant() ? cs->new_lc({{t.value(), VarOne()}}) : t.variable();
    Variable fv = f.is_constant() ? cs->new_lc({{f.value(), VarOne()}}) : f.variable();
    FpVar result = new_witness_from_engine(cs, "conditionally_select");
    // cond * (true - false) = result - false
    cs->enforce_constraint(cond.lc(), {{Fr::one(), tv}, {-Fr::one(), fv}}, {{Fr::one(), result.variable()}, {-Fr::one(), fv}});
    return result;
}

// ---------------------------------------------------------------------------------------------------------------
// falcon-rust stand-ins used by the path
// ---------------------------------------------------------------------------------------------------------------
constexpr uint32_t MODULUS = 12289;
inline uint64_t SIG_L2_BOUND(int logn) { return logn == 9 ? 34034726ULL : 70265242ULL; }

struct Polynomial {                      // coefficients in [0, q)
    std::vector<uint16_t> c;
    const std::vector<uint16_t> &coeff() const { return c; }
};
struct NTTPolynomial {
    std::vector<uint16_t> c;
    const std::vector<uint16_t> &coeff() const { return c; }
};
// falcon-rust PublicKey / Signature: the encoded bytes (Falcon spec 3.11.3-3.11.4; see include/frw.h)
struct PublicKey {
    std::vector<uint8_t> bytes;
    int logn() const { return bytes.empty() ? -1 : (int)bytes[0]; }
};
struct Signature {
    std::vector<uint8_t> bytes;
    // Signature::nonce(): the 40 bytes after the header (falcon_ntt.rs:44)
    std::vector<uint8_t> nonce() const { return std::vector<uint8_t>(bytes.begin() + 1, bytes.begin() + 1 + FRW_NONCE_LEN); }
};

inline uint32_t powmod_q(uint32_t b, uint32_t e) { uint64_t r = 1, x = b; while (e) { if (e & 1) r = r * x % MODULUS; x = x * x % MODULUS; e >>= 1; } return (uint32_t)r; }
// falcon-rust NTT_TABLE[i] = 7^bitrev10(i) mod q (script/ntt_param.sage:3-132)
inline uint32_t NTT_TABLE(uint32_t i) { uint32_t r = 0; for (int b = 0; b < 10; b++) if (i & (1u << b)) r |= 1u << (9 - b); return powmod_q(7, r); }

// ---------------------------------------------------------------------------------------------------------------
// engine requests for gadgets called on their own (the feed is empty): one frw_gadget / frw_ntt_modq call
// ---------------------------------------------------------------------------------------------------------------
namespace detail {
inline void require_engine(const ConstraintSystemRef &cs, const char *who)
{
    if (!cs->engine()) throw SynthesisError(SynthesisError::AssignmentMissing, std::string(who) + ": no engine attached (AssignmentMissing; there is no CPU path)");
}
inline void check(int rc, const char *who)
{
    if (rc != FRW_OK) throw SynthesisError(SynthesisError::Engine, std::string(who) + ": " + frw_strerror(rc) + "; " + frw_last_error());
}
// value must fit the gadget's documented input domain
inline bool fits_u64(const Fr &v, uint64_t &out) { uint64_t c[4]; v.to_canonical(c); out = c[0]; return !(c[1] | c[2] | c[3]); }

inline void feed_gadget(const ConstraintSystemRef &cs, int kind, const Fr &a, const Fr *b, const char *who)
{
    if (cs->is_in_setup_mode() || cs->feed_remaining()) return;       // inside a larger circuit: values already there
    require_engine(cs, who);
    const int blk = frw_gadget_block_len(kind);
    std::vector<uint64_t> out((size_t)blk * 4);
    int32_t st = 0;
    uint64_t ca[4];
    a.to_canonical(ca);
    if (kind == FRW_G_MOD_Q) {
        if (ca[3] | (ca[2] >> 32)) throw SynthesisError(SynthesisError::Engine, std::string(who) + ": input exceeds 160 bits");
        uint32_t limbs[5] = {(uint32_t)ca[0], (uint32_t)(ca[0] >> 32), (uint32_t)ca[1], (uint32_t)(ca[1] >> 32), (uint32_t)ca[2]};
        check(frw_gadget(cs->engine()->get(), kind, 1, limbs, nullptr, FRW_ENC_MONTGOMERY, out.data(), &st), who);
    } else {
        uint64_t av, bv = 0;
        if (!fits_u64(a, av) || (b && !fits_u64(*b, bv))) throw SynthesisError(SynthesisError::Engine, std::string(who) + ": input exceeds 64 bits");
        check(frw_gadget(cs->engine()->get(), kind, 1, &av, b ? &bv : nullptr, FRW_ENC_MONTGOMERY, out.data(), &st), who);
    }
    if (st != FRW_ST_OK) throw SynthesisError(SynthesisError::Engine, std::string(who) + ": input outside the gadget's domain");
    cs->push_feed(out.data(), (size_t)blk);
}
}  // namespace detail

// ---------------------------------------------------------------------------------------------------------------
// gadgets/misc.rs
// ---------------------------------------------------------------------------------------------------------------
// misc.rs:9-24
inline void enforce_decompose(const FpVar &a, const std::vector<Boolean> &bits)
{
    if (bits.empty()) throw std::invalid_argument("Invalid input length: 0");
    FpVar res = FpVar::from_boolean(bits.back());
    for (size_t i = bits.size() - 1; i-- > 0;) res = res.double_() + FpVar::from_boolean(bits[i]);
    res.enforce_equal(a);
}

// misc.rs:67-77
inline std::vector<FpVar> ntt_param_var(const ConstraintSystemRef &cs, int logn)
{
    std::vector<FpVar> res;
    for (uint32_t i = 0; i < (1u << logn); i++) res.push_back(FpVar::new_constant(cs, Fr::from(NTT_TABLE(i))));
    return res;
}

// ---------------------------------------------------------------------------------------------------------------
// gadgets/range_proofs.rs
// ---------------------------------------------------------------------------------------------------------------
namespace detail {
inline std::vector<Boolean> alloc_bits(const ConstraintSystemRef &cs, int n, const char *who)
{
    std::vector<Boolean> bits;
    for (int i = 0; i < n; i++) bits.push_back(Boolean::new_witness_from_engine(cs, who));
    return bits;
}
inline bool lt_u64(const Fr &v, uint64_t bound) { uint64_t x; return fits_u64(v, x) && x < bound; }
}  // namespace detail

// range_proofs.rs:42-94.  Strict contexts mirror the #[cfg(not(test))] panic at :57-60.
inline void enforce_less_than_q(const ConstraintSystemRef &cs, const FpVar &a)
{
    if (!cs->is_in_setup_mode() && cs->strict() && !detail::lt_u64(a.value(), MODULUS)) throw std::domain_error("Invalid input: value >= MODULUS");
    detail::feed_gadget(cs, FRW_G_LESS_THAN_Q, a.value(), nullptr, "enforce_less_than_q");
    std::vector<Boolean> b = detail::alloc_bits(cs, 14, "enforce_less_than_q bit");
    enforce_decompose(a, b);
    const Boolean F = Boolean::FALSE_();
    Boolean r13 = b[13].is_eq(F), r12 = b[12].is_eq(F);
    Boolean low = Boolean::kary_or(b, 0, 12).is_eq(F);
    r13.or_(r12.or_(low)).enforce_equal(Boolean::TRUE_());
}

// range_proofs.rs:289-333
inline Boolean is_less_than_6144(const ConstraintSystemRef &cs, const FpVar &a)
{
    std::vector<Boolean> b = detail::alloc_bits(cs, 14, "is_less_than_6144 bit");
    enforce_decompose(a, b);
    const Boolean F = Boolean::FALSE_();
    return b[13].is_eq(F).and_(b[12].is_eq(F).or_(b[11].is_eq(F))).is_eq(Boolean::TRUE_());
}

// range_proofs.rs:100-186
inline void enforce_less_than_norm_bound_512(const ConstraintSystemRef &cs, const FpVar &a)
{
    if (!cs->is_in_setup_mode() && cs->strict() && !detail::lt_u64(a.value(), SIG_L2_BOUND(9))) throw std::domain_error("Invalid input: norm >= SIG_L2_BOUND");
    detail::feed_gadget(cs, FRW_G_NORM_BOUND_512, a.value(), nullptr, "enforce_less_than_norm_bound_512");
    std::vector<Boolean> b = detail::alloc_bits(cs, 26, "norm bit");
    enforce_decompose(a, b);
    const Boolean F = Boolean::FALSE_();
    Boolean r25 = b[25].is_eq(F);
    Boolean k19 = Boolean::kary_or(b, 19, 25).is_eq(F);
    Boolean k16 = Boolean::kary_and(b, 16, 19).is_eq(F);
    Boolean r15 = b[15].is_eq(F), r14 = b[14].is_eq(F), r13 = b[13].is_eq(F), r12 = b[12].is_eq(F), r11 = b[11].is_eq(F), r10 = b[10].is_eq(F);
    Boolean k6 = Boolean::kary_or(b, 6, 10).is_eq(F);
    Boolean r5 = b[5].is_eq(F);
    Boolean k3 = Boolean::kary_or(b, 3, 5).is_eq(F);
    Boolean k1 = Boolean::kary_and(b, 1, 3).is_eq(F);
    Boolean x = k3.and_(k1);
    x = r5.or_(x); x = k6.and_(x); x = r10.or_(x); x = r11.and_(x); x = r12.or_(x); x = r13.and_(x);
    x = r14.or_(x); x = r15.and_(x); x = k16.or_(x); x = k19.and_(x); x = r25.or_(x);
    x.enforce_equal(Boolean::TRUE_());
}

// range_proofs.rs:192-272
inline void enforce_less_than_norm_bound_1024(const ConstraintSystemRef &cs, const FpVar &a)
{
    if (!cs->is_in_setup_mode() && cs->strict() && !detail::lt_u64(a.value(), SIG_L2_BOUND(10))) throw std::domain_error("Invalid input: norm >= SIG_L2_BOUND");
    detail::feed_gadget(cs, FRW_G_NORM_BOUND_1024, a.value(), nullptr, "enforce_less_than_norm_bound_1024");
    std::vector<Boolean> b = detail::alloc_bits(cs, 27, "norm bit");
    enforce_decompose(a, b);
    const Boolean F = Boolean::FALSE_();
    Boolean r26 = b[26].is_eq(F);
    Boolean k22 = Boolean::kary_or(b, 22, 26).is_eq(F);
    Boolean k20 = Boolean::kary_and(b, 20, 22).is_eq(F);
    Boolean k14 = Boolean::kary_or(b, 14, 20).is_eq(F);
    Boolean r13 = b[13].is_eq(F), r12 = b[12].is_eq(F), r11 = b[11].is_eq(F);
    Boolean k9 = Boolean::kary_or(b, 9, 11).is_eq(F);
    Boolean k7 = Boolean::kary_and(b, 7, 9).is_eq(F);
    Boolean k5 = Boolean::kary_or(b, 5, 7).is_eq(F);
    Boolean k3 = Boolean::kary_and(b, 3, 5).is_eq(F);
    Boolean k1 = Boolean::kary_or(b, 1, 3).is_eq(F);
    Boolean x = k3.or_(k1);
    x = k5.and_(x); x = k7.or_(x); x = k9.and_(x); x = r11.or_(x); x = r12.and_(x); x = r13.or_(x);
    x = k14.and_(x); x = k20.or_(x); x = k22.and_(x); x = r26.or_(x);
    x.enforce_equal(Boolean::TRUE_());
}

// range_proofs.rs:274-284 (the cargo feature becomes the run-time logn)
inline void enforce_less_than_norm_bound(const ConstraintSystemRef &cs, const FpVar &a, int logn)
{
    if (logn == 9) enforce_less_than_norm_bound_512(cs, a);
    else if (logn == 10) enforce_less_than_norm_bound_1024(cs, a);
    else throw std::invalid_argument("logn must be 9 or 10");
}

// misc.rs:30-51
inline FpVar l2_norm_var(const ConstraintSystemRef &cs, const std::vector<FpVar> &input, const FpVar &modulus_var)
{
    std::optional<FpVar> res;
    for (const FpVar &e : input) {
        detail::feed_gadget(cs, FRW_G_L2_ELEM, e.value(), nullptr, "l2_norm_var element");
        FpVar tmp = FpVar::conditionally_select(is_less_than_6144(cs, e), e, modulus_var - e);
        FpVar sq = tmp * tmp;
        res = res ? *res + sq : sq;
    }
    return *res;
}

// ---------------------------------------------------------------------------------------------------------------
// gadgets/arithmetics.rs
// ---------------------------------------------------------------------------------------------------------------
// arithmetics.rs:105-149
inline FpVar mod_q(const ConstraintSystemRef &cs, const FpVar &a, const FpVar &modulus_var)
{
    detail::feed_gadget(cs, FRW_G_MOD_Q, a.value(), nullptr, "mod_q");
    FpVar t_var = FpVar::new_witness_from_engine(cs, "mod_q t");
    FpVar b_var = FpVar::new_witness_from_engine(cs, "mod_q b");
    FpVar left = a - t_var * modulus_var;
    left.enforce_equal(b_var);
    enforce_less_than_q(cs, b_var);
    return b_var;
}

// arithmetics.rs:214-262
inline FpVar add_mod(const ConstraintSystemRef &cs, const FpVar &a, const FpVar &b, const FpVar &modulus_var)
{
    detail::feed_gadget(cs, FRW_G_ADD_MOD, a.value(), &b.value(), "add_mod");
    FpVar t_var = FpVar::new_witness_from_engine(cs, "add_mod t");
    FpVar c_var = FpVar::new_witness_from_engine(cs, "add_mod c");
    FpVar left = (a + b) - t_var * modulus_var;
    left.enforce_equal(c_var);
    enforce_less_than_q(cs, c_var);
    return c_var;
}

// ---------------------------------------------------------------------------------------------------------------
// gadgets/poly.rs
// ---------------------------------------------------------------------------------------------------------------
struct PolyVar {
    std::vector<FpVar> v;
    const std::vector<FpVar> &coeff() const { return v; }
    // poly.rs:195-211.  In prove mode the VALUES come from the engine feed when one is loaded (full circuit),
    // else from the caller's polynomial (an input chosen by the caller, as in the reference's tests).
    static PolyVar alloc_vars(const ConstraintSystemRef &cs, const Polynomial &poly, AllocationMode mode)
    {
        PolyVar r;
        for (uint16_t c : poly.coeff()) {
            Fr val = cs->is_in_setup_mode() ? Fr::one() : (cs->feed_remaining() && mode == AllocationMode::Witness ? cs->pop_feed("PolyVar") : Fr::from(c));
            r.v.push_back(mode == AllocationMode::Input ? FpVar::new_input(cs, val) : FpVar::new_witness(cs, val));
        }
        return r;
    }
};

struct NTTPolyVar {
    std::vector<FpVar> v;
    const std::vector<FpVar> &coeff() const { return v; }
    // poly.rs:47-63
    static NTTPolyVar alloc_vars(const ConstraintSystemRef &cs, const NTTPolynomial &poly, AllocationMode mode)
    {
        NTTPolyVar r;
        for (uint16_t c : poly.coeff()) {
            Fr val = cs->is_in_setup_mode() ? Fr::one() : Fr::from(c);
            r.v.push_back(mode == AllocationMode::Input ? FpVar::new_input(cs, val) : FpVar::new_witness(cs, val));
        }
        return r;
    }

    // poly.rs:104-159: the ladder builds symbolic LCs only; N x mod_q allocate.  Called on its own, the N mod_q
    // blocks come from one frw_ntt_modq call on the input polynomial's values.
    static NTTPolyVar ntt_circuit(const ConstraintSystemRef &cs, const PolyVar &input, const std::vector<FpVar> &const_vars,
                                  const std::vector<FpVar> &param, int logn)
    {
        const size_t N = (size_t)1 << logn;
        if (input.coeff().size() != N) throw std::invalid_argument("input length " + std::to_string(input.coeff().size()) + " is not N");
        if (!cs->is_in_setup_mode() && !cs->feed_remaining()) {
            detail::require_engine(cs, "ntt_circuit");
            std::vector<uint16_t> poly(N), out(N);
            for (size_t i = 0; i < N; i++) {
                uint64_t x;
                if (!detail::fits_u64(input.coeff()[i].value(), x) || x >= MODULUS) throw std::domain_error("ntt_circuit: coefficient >= MODULUS");
                poly[i] = (uint16_t)x;
            }
            std::vector<uint64_t> wit(29 * N * 4);
            int32_t st = 0;
            detail::check(frw_ntt_modq(cs->engine()->get(), logn, 1, poly.data(), FRW_ENC_MONTGOMERY, wit.data(), out.data(), &st), "frw_ntt_modq");
            cs->push_feed(wit.data(), 29 * N);
        }
        std::vector<FpVar> output = input.coeff();
        size_t t = N;
        for (int l = 0; l < logn; l++) {
            const size_t m = (size_t)1 << l, ht = t / 2;
            size_t j1 = 0;
            for (size_t i = 0; i < m; i++) {
                const FpVar &s = param[m + i];
                for (size_t j = j1; j < j1 + ht; j++) {
                    FpVar u = output[j];
                    FpVar v = output[j + ht] * s;
                    FpVar neg_v = const_vars[l + 1] - v;
                    output[j] = u + v;
                    output[j + ht] = u + neg_v;
                }
                j1 += t;
            }
            t = ht;
        }
        NTTPolyVar r;
        for (FpVar &e : output) r.v.push_back(mod_q(cs, e, const_vars[0]));
        return r;
    }
};

// falcon_ntt.rs:31-39: [q, 2 q^2, ..., 2^LOG_N q^(LOG_N+1)]
inline std::vector<FpVar> const_q_power_vars(const ConstraintSystemRef &cs, int logn)
{
    std::vector<FpVar> r;
    Fr q = Fr::from(MODULUS);
    for (int x = 1; x <= logn + 1; x++) r.push_back(FpVar::new_constant(cs, Fr::from(1ull << (x - 1)) * q.pow((uint64_t)x)));
    return r;
}

// ---------------------------------------------------------------------------------------------------------------
// circuits/falcon_ntt.rs
// ---------------------------------------------------------------------------------------------------------------
class FalconNTTVerificationCircuit {
public:
    // falcon_ntt.rs:15-17: build_circuit(pk, msg, sig).  The three coefficient vectors of :27-28,:44 are derived by
    // the engine (frw_prepare_inputs: decoders + SHAKE256 hash-to-point on the GPU) when generate_constraints runs.
    static FalconNTTVerificationCircuit build_circuit(PublicKey pk, std::vector<uint8_t> msg, Signature sig)
    {
        const int logn = pk.logn();
        if (logn != 9 && logn != 10) throw std::invalid_argument("public key header is not a Falcon-512/1024 key");
        FalconNTTVerificationCircuit c;
        c.logn_ = logn;
        c.pk_bytes_ = std::move(pk); c.msg_ = std::move(msg); c.sig_bytes_ = std::move(sig);
        c.from_bytes_ = true;
        return c;
    }

    // The same circuit from the coefficient vectors themselves (a caller that already holds Polynomial::from(&pk),
    // from_hash_of_message(..) and Polynomial::from(&sig), e.g. a batch prover after one frw_prepare_inputs call).
    static FalconNTTVerificationCircuit build_circuit(Polynomial pk, Polynomial hm, Polynomial sig, int logn)
    {
        const size_t N = (size_t)1 << logn;
        if (pk.c.size() != N || hm.c.size() != N || sig.c.size() != N) throw std::invalid_argument("input length is not N");
        FalconNTTVerificationCircuit c;
        c.pk_ = std::move(pk); c.hm_ = std::move(hm); c.sig_ = std::move(sig); c.logn_ = logn;
        return c;
    }

    // falcon_ntt.rs:26-123.  Prove mode: ONE engine call fills every witness and instance value; the structural
    // pass below allocates and constrains in the reference's order and pops them.
    void generate_constraints(const ConstraintSystemRef &cs) const
    {
        const int logn = logn_;
        const size_t N = (size_t)1 << logn;
        NTTPolynomial pk_ntt{std::vector<uint16_t>(N, 1)}, hm_ntt{std::vector<uint16_t>(N, 1)};
        Polynomial pk_ = this->pk_, hm_ = this->hm_, sig_ = this->sig_;
        if (from_bytes_) {
            pk_.c.assign(N, 0); hm_.c.assign(N, 0); sig_.c.assign(N, 0);
            if (!cs->is_in_setup_mode() && !preset_wit_) {       // falcon_ntt.rs:27-28,44 on the engine
                detail::require_engine(cs, "FalconNTTVerificationCircuit");
                const uint64_t off[2] = {0, msg_.size()};
                int32_t st = 0;
                detail::check(frw_prepare_inputs(cs->engine()->get(), logn, 1, pk_bytes_.bytes.data(), sig_bytes_.bytes.data(),
                                                 sig_bytes_.bytes.size(), msg_.empty() ? nullptr : msg_.data(), off, sig_.c.data(),
                                                 pk_.c.data(), hm_.c.data(), &st), "frw_prepare_inputs");
                if (st != FRW_ST_OK) throw std::domain_error("Invalid input: malformed public key or signature encoding");
            }
        }
        if (!cs->is_in_setup_mode()) {
            frw_layout_t L;
            frw_layout(logn, &L);
            std::vector<uint64_t> wit_buf, inst_buf;
            const uint64_t *wit = preset_wit_, *inst = preset_inst_;
            if (!wit) {                                   // no slice of a batched engine call was handed over: call now
                detail::require_engine(cs, "FalconNTTVerificationCircuit");
                wit_buf.resize((size_t)L.num_witness * 4);
                inst_buf.resize((size_t)L.num_instance * 4);
                int32_t st = 0;
                int rc = frw_witness_ntt_verify(cs->engine()->get(), logn, 1, sig_.c.data(), pk_.c.data(), hm_.c.data(),
                                                FRW_ENC_MONTGOMERY, wit_buf.data(), inst_buf.data(), &st, cs->strict() ? 1 : 0);
                if (rc == FRW_E_RANGE) throw std::domain_error("Invalid input: range check failed (status " + std::to_string(st) + ")");   // the reference panics
                detail::check(rc, "frw_witness_ntt_verify");
                if (st == FRW_ST_COEFF_RANGE) throw std::domain_error("Invalid input: coefficient >= MODULUS");
                wit = wit_buf.data();
                inst = inst_buf.data();
            }
            cs->push_feed(wit, (size_t)L.num_witness);
            // public inputs: canonical u16 values recovered from the engine's instance vector
            for (size_t i = 0; i < N; i++) {
                uint64_t c[4];
                Fr::from_montgomery(&inst[4 * (1 + i)]).to_canonical(c); pk_ntt.c[i] = (uint16_t)c[0];
                Fr::from_montgomery(&inst[4 * (1 + N + i)]).to_canonical(c); hm_ntt.c[i] = (uint16_t)c[0];
            }
        }
        std::vector<FpVar> consts = const_q_power_vars(cs, logn);                              // :31-39
        std::vector<FpVar> param_vars = ntt_param_var(cs, logn);                              // :40
        PolyVar sig_poly_vars = PolyVar::alloc_vars(cs, sig_, AllocationMode::Witness);       // :58-59
        NTTPolyVar pk_ntt_vars = NTTPolyVar::alloc_vars(cs, pk_ntt, AllocationMode::Input);   // :63
        NTTPolyVar hm_ntt_vars = NTTPolyVar::alloc_vars(cs, hm_ntt, AllocationMode::Input);   // :67
        PolyVar v_vars = PolyVar::alloc_vars(cs, Polynomial{std::vector<uint16_t>(N, 0)}, AllocationMode::Witness);   // :71 (v comes from the engine)
        for (const FpVar &e : v_vars.coeff()) enforce_less_than_q(cs, e);                     // :73-77
        NTTPolyVar sig_ntt_vars = NTTPolyVar::ntt_circuit(cs, sig_poly_vars, consts, param_vars, logn);   // :88-89
        NTTPolyVar v_ntt_vars = NTTPolyVar::ntt_circuit(cs, v_vars, consts, param_vars, logn);            // :90-91
        for (size_t i = 0; i < N; i++) {                                                      // :94-111
            FpVar prod = sig_ntt_vars.coeff()[i] * pk_ntt_vars.coeff()[i];
            hm_ntt_vars.coeff()[i].enforce_equal(add_mod(cs, v_ntt_vars.coeff()[i], prod, consts[0]));
        }
        std::vector<FpVar> both = v_vars.coeff();
        both.insert(both.end(), sig_poly_vars.coeff().begin(), sig_poly_vars.coeff().end());
        FpVar l2 = l2_norm_var(cs, both, consts[0]);                                          // :116-120
        enforce_less_than_norm_bound(cs, l2, logn);                                           // :122
        if (!cs->is_in_setup_mode() && cs->feed_remaining())
            throw SynthesisError(SynthesisError::Engine, "engine produced more values than the circuit allocates");
    }

    // A batch prover makes ONE frw_witness_ntt_verify(_dev) call for many signatures and hands circuit i its slice
    // (W and I field elements, Montgomery limbs); generate_constraints then makes no engine call of its own.
    void use_engine_output(const uint64_t *witness_slice, const uint64_t *instance_slice)
    {
        preset_wit_ = witness_slice;
        preset_inst_ = instance_slice;
    }

private:
    Polynomial pk_, hm_, sig_;
    PublicKey pk_bytes_;
    Signature sig_bytes_;
    std::vector<uint8_t> msg_;
    bool from_bytes_ = false;
    int logn_ = 10;
    const uint64_t *preset_wit_ = nullptr, *preset_inst_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------
// gadgets/dual_poly.rs, gadgets/misc.rs:55-65, circuits/falcon_dual_ntt.rs  (the signed-split variant)
// ---------------------------------------------------------------------------------------------------------------
struct DualPolynomial { Polynomial pos, neg; };

// misc.rs:55-65
inline FpVar l2_norm_var_without_range_check(const std::vector<FpVar> &input)
{
    std::optional<FpVar> res;
    for (const FpVar &e : input) { FpVar sq = e * e; res = res ? *res + sq : sq; }
    return *res;
}

struct DualPolyVar {
    PolyVar pos, neg;
    // dual_poly.rs:15-31
    static DualPolyVar alloc_vars(const ConstraintSystemRef &cs, const DualPolynomial &dual_poly, AllocationMode mode)
    {
        DualPolyVar r;
        r.pos = PolyVar::alloc_vars(cs, dual_poly.pos, mode);
        r.neg = PolyVar::alloc_vars(cs, dual_poly.neg, mode);
        FpVar acc = r.pos.coeff()[0] * r.neg.coeff()[0];
        for (size_t i = 1; i < r.pos.coeff().size(); i++) acc = acc + r.pos.coeff()[i] * r.neg.coeff()[i];
        acc.is_zero().enforce_equal(Boolean::TRUE_());
        return r;
    }
};

struct DualNTTPolyVar {
    NTTPolyVar pos, neg;
    // dual_poly.rs:41-51
    static DualNTTPolyVar ntt_circuit(const ConstraintSystemRef &cs, const DualPolyVar &input, const std::vector<FpVar> &const_vars,
                                      const std::vector<FpVar> &param, int logn)
    {
        DualNTTPolyVar r;
        r.pos = NTTPolyVar::ntt_circuit(cs, input.pos, const_vars, param, logn);
        r.neg = NTTPolyVar::ntt_circuit(cs, input.neg, const_vars, param, logn);
        return r;
    }
};

class FalconDualNTTVerificationCircuit {
public:
    // falcon_dual_ntt.rs:15-17, from the coefficient vectors falcon-rust derives (:27-28,:43)
    static FalconDualNTTVerificationCircuit build_circuit(Polynomial pk, Polynomial hm, Polynomial sig, int logn)
    {
        const size_t N = (size_t)1 << logn;
        if (pk.c.size() != N || hm.c.size() != N || sig.c.size() != N) throw std::invalid_argument("input length is not N");
        FalconDualNTTVerificationCircuit c;
        c.pk_ = std::move(pk); c.hm_ = std::move(hm); c.sig_ = std::move(sig); c.logn_ = logn;
        return c;
    }
    void use_engine_output(const uint64_t *witness_slice, const uint64_t *instance_slice) { preset_wit_ = witness_slice; preset_inst_ = instance_slice; }

    // falcon_dual_ntt.rs:26-132
    void generate_constraints(const ConstraintSystemRef &cs) const
    {
        const int logn = logn_;
        const size_t N = (size_t)1 << logn;
        NTTPolynomial pk_ntt{std::vector<uint16_t>(N, 1)}, hm_ntt{std::vector<uint16_t>(N, 1)};
        if (!cs->is_in_setup_mode()) {
            frw_layout_dual_t L;
            frw_layout_dual(logn, &L);
            std::vector<uint64_t> wit_buf, inst_buf;
            const uint64_t *wit = preset_wit_, *inst = preset_inst_;
            if (!wit) {
                detail::require_engine(cs, "FalconDualNTTVerificationCircuit");
                wit_buf.resize((size_t)L.num_witness * 4);
                inst_buf.resize((size_t)L.num_instance * 4);
                int32_t st = 0;
                int rc = frw_witness_dual_ntt_verify(cs->engine()->get(), logn, 1, sig_.c.data(), pk_.c.data(), hm_.c.data(),
                                                     FRW_ENC_MONTGOMERY, wit_buf.data(), inst_buf.data(), &st, cs->strict() ? 1 : 0);
                if (rc == FRW_E_RANGE) throw std::domain_error("Invalid input: range check failed (status " + std::to_string(st) + ")");
                detail::check(rc, "frw_witness_dual_ntt_verify");
                if (st == FRW_ST_COEFF_RANGE) throw std::domain_error("Invalid input: coefficient >= MODULUS");
                wit = wit_buf.data();
                inst = inst_buf.data();
            }
            cs->push_feed(wit, (size_t)L.num_witness);
            for (size_t i = 0; i < N; i++) {
                uint64_t c[4];
                Fr::from_montgomery(&inst[4 * (1 + i)]).to_canonical(c); pk_ntt.c[i] = (uint16_t)c[0];
                Fr::from_montgomery(&inst[4 * (1 + N + i)]).to_canonical(c); hm_ntt.c[i] = (uint16_t)c[0];
            }
        }
        const DualPolynomial zero{Polynomial{std::vector<uint16_t>(N, 0)}, Polynomial{std::vector<uint16_t>(N, 0)}};   // values come from the engine
        std::vector<FpVar> consts = const_q_power_vars(cs, logn);                                   // :31-39
        std::vector<FpVar> param_vars = ntt_param_var(cs, logn);                                   // :40
        DualPolyVar sig_poly_vars = DualPolyVar::alloc_vars(cs, zero, AllocationMode::Witness);    // :60-61
        NTTPolyVar pk_ntt_vars = NTTPolyVar::alloc_vars(cs, pk_ntt, AllocationMode::Input);        // :65
        NTTPolyVar hm_ntt_vars = NTTPolyVar::alloc_vars(cs, hm_ntt, AllocationMode::Input);        // :69
        DualPolyVar v_vars = DualPolyVar::alloc_vars(cs, zero, AllocationMode::Witness);           // :73
        DualNTTPolyVar sig_ntt_vars = DualNTTPolyVar::ntt_circuit(cs, sig_poly_vars, consts, param_vars, logn);   // :85-90
        DualNTTPolyVar v_ntt_vars = DualNTTPolyVar::ntt_circuit(cs, v_vars, consts, param_vars, logn);            // :91-92
        for (size_t i = 0; i < N; i++) {                                                           // :95-116
            FpVar left = mod_q(cs, hm_ntt_vars.coeff()[i] + v_ntt_vars.neg.coeff()[i] + sig_ntt_vars.neg.coeff()[i] * pk_ntt_vars.coeff()[i], consts[0]);
            FpVar right = mod_q(cs, v_ntt_vars.pos.coeff()[i] + sig_ntt_vars.pos.coeff()[i] * pk_ntt_vars.coeff()[i], consts[0]);
            left.enforce_equal(right);
        }
        std::vector<FpVar> all = v_vars.pos.coeff();                                               // :121-129
        for (const auto *part : {&v_vars.neg, &sig_poly_vars.pos, &sig_poly_vars.neg}) all.insert(all.end(), part->coeff().begin(), part->coeff().end());
        FpVar l2 = l2_norm_var_without_range_check(all);
        enforce_less_than_norm_bound(cs, l2, logn);                                                // :131
        if (!cs->is_in_setup_mode() && cs->feed_remaining())
            throw SynthesisError(SynthesisError::Engine, "engine produced more values than the circuit allocates");
    }

private:
    Polynomial pk_, hm_, sig_;
    int logn_ = 10;
    const uint64_t *preset_wit_ = nullptr, *preset_inst_ = nullptr;
};

// ---------------------------------------------------------------------------------------------------------------
// Aggregate statement.  The reference's falcon-aggregate-sig crate is an empty stub
// (falcon-aggregate-sig/src/main.rs:1-3), so the behaviour is defined here in the only way the reference's own
// circuits allow: ONE constraint system on which FalconNTTVerificationCircuit::generate_constraints runs once per
// statement, in order.  Its witness_assignment is then the concatenation of the per-signature witness vectors and its
// instance_assignment is [1, (pk_ntt, hm_ntt) of statement 0, of statement 1, ...] -- exactly the batch buffers of
// frw_witness_ntt_verify, so the whole aggregate witness costs one engine call per parameter set, whatever the count.
// ---------------------------------------------------------------------------------------------------------------
class FalconAggregateVerificationCircuit {
public:
    struct Statement { Polynomial pk, hm, sig; int logn; };

    static FalconAggregateVerificationCircuit build_circuit(std::vector<Statement> statements)
    {
        FalconAggregateVerificationCircuit c;
        c.st_ = std::move(statements);
        return c;
    }

    void generate_constraints(const ConstraintSystemRef &cs) const
    {
        // one batched engine call per parameter set (Falcon-512 and Falcon-1024 may be mixed)
        std::vector<uint64_t> wit[2], inst[2];
        std::vector<size_t> slot(st_.size());
        if (!cs->is_in_setup_mode()) {
            detail::require_engine(cs, "FalconAggregateVerificationCircuit");
            for (int logn = 9; logn <= 10; logn++) {
                const size_t N = (size_t)1 << logn;
                std::vector<uint16_t> sig, pk, hm;
                size_t count = 0;
                for (size_t i = 0; i < st_.size(); i++)
                    if (st_[i].logn == logn) {
                        sig.insert(sig.end(), st_[i].sig.c.begin(), st_[i].sig.c.end());
                        pk.insert(pk.end(), st_[i].pk.c.begin(), st_[i].pk.c.end());
                        hm.insert(hm.end(), st_[i].hm.c.begin(), st_[i].hm.c.end());
                        slot[i] = count++;
                    }
                if (!count) continue;
                if (sig.size() != count * N || pk.size() != count * N || hm.size() != count * N) throw std::invalid_argument("input length is not N");
                frw_layout_t L;
                frw_layout(logn, &L);
                wit[logn - 9].resize(count * (size_t)L.num_witness * 4);
                inst[logn - 9].resize(count * (size_t)L.num_instance * 4);
                std::vector<int32_t> status(count);
                int rc = frw_witness_ntt_verify(cs->engine()->get(), logn, count, sig.data(), pk.data(), hm.data(), FRW_ENC_MONTGOMERY,
                                                wit[logn - 9].data(), inst[logn - 9].data(), status.data(), cs->strict() ? 1 : 0);
                if (rc == FRW_E_RANGE) throw std::domain_error("Invalid input: a statement failed its range checks");
                detail::check(rc, "frw_witness_ntt_verify");
            }
        }
        for (size_t i = 0; i < st_.size(); i++) {
            const Statement &s = st_[i];
            FalconNTTVerificationCircuit one = FalconNTTVerificationCircuit::build_circuit(s.pk, s.hm, s.sig, s.logn);
            if (!cs->is_in_setup_mode()) {
                frw_layout_t L;
                frw_layout(s.logn, &L);
                one.use_engine_output(wit[s.logn - 9].data() + slot[i] * (size_t)L.num_witness * 4,
                                      inst[s.logn - 9].data() + slot[i] * (size_t)L.num_instance * 4);
            }
            one.generate_constraints(cs);
        }
    }

private:
    std::vector<Statement> st_;
};

}  // namespace frw::host
