// Tests of the C++ host mirror (falcon-r1cs_amd/csrc/host/frw_host.hpp), written after the reference's own tests:
//   gadgets/arithmetics.rs:311-373 (test_mod_q), :436-507 (test_add_mod)
//   gadgets/range_proofs.rs:343-418 (test_range_proof_mod_q), :420-504 (norm bound), :505-577 (half q)
//   gadgets/poly.rs:252-301 (test_ntt_mul_circuit)
//   circuits/falcon_ntt.rs:133-160 (test_ntt_verification_r1cs)
//   examples/constraint_counts.rs (count table)
// Modes:
//   structure                          CPU only: setup-mode counts == README.md:41-56
//   check <logn> <sig> <pk> <hm> <wit> <inst>   CPU only: a witness produced elsewhere (raw little-endian files)
//                                      satisfies the constraint system emitted here
//   gpu                                everything, values from the HIP engine
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <random>
#include <string>

#include "../../falcon-r1cs_amd/csrc/host/frw_host.hpp"

using namespace frw::host;

static int failures = 0;
#define EXPECT(cond)                                                                \
    do {                                                                            \
        if (!(cond)) { std::printf("  FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); failures++; } \
    } while (0)

static std::vector<uint16_t> ntt_clear(std::vector<uint16_t> a, int logn)     // independent of the engine
{
    size_t n = (size_t)1 << logn, t = n;
    for (size_t m = 1; m < n; m <<= 1) {
        size_t ht = t >> 1, j1 = 0;
        for (size_t i = 0; i < m; i++, j1 += t) {
            uint32_t s = NTT_TABLE((uint32_t)(m + i));
            for (size_t j = j1; j < j1 + ht; j++) {
                uint32_t u = a[j], v = a[j + ht] * s % MODULUS;
                a[j] = (uint16_t)((u + v) % MODULUS);
                a[j + ht] = (uint16_t)((u + MODULUS - v) % MODULUS);
            }
        }
        t = ht;
    }
    return a;
}

// ---------------------------------------------------------------------------------------------------------------
// examples/constraint_counts.rs: structure only (setup mode: no values, no engine)
// ---------------------------------------------------------------------------------------------------------------
static void count_table(int logn)
{
    const size_t N = (size_t)1 << logn;
    std::printf("|                 | # instance variables |      # witness |      #constraints |\n|---|---:|---:|---:|\n");
    {   // count_ntt_conversion_constraints, constraint_counts.rs:74-113
        auto cs = ConstraintSystem::new_ref();
        cs->set_setup_mode(true);
        auto param = ntt_param_var(cs, logn);
        auto consts = const_q_power_vars(cs, logn);
        PolyVar pv = PolyVar::alloc_vars(cs, Polynomial{std::vector<uint16_t>(N, 0)}, AllocationMode::Witness);
        size_t i0 = cs->num_instance_variables(), w0 = cs->num_witness_variables(), c0 = cs->num_constraints();
        NTTPolyVar::ntt_circuit(cs, pv, consts, param, logn);
        size_t di = cs->num_instance_variables() - i0, dw = cs->num_witness_variables() - w0, dc = cs->num_constraints() - c0;
        std::printf("|ntt conversion|\t%8zu |\t%8zu |\t%8zu |\n", di, dw, dc);
        EXPECT(di == 0 && dw == 29 * N && dc == 30 * N);                       // README.md:43,54
    }
    {   // count_verify_with_ntt_constraints, constraint_counts.rs:49-72
        auto cs = ConstraintSystem::new_ref();
        cs->set_setup_mode(true);
        Polynomial z{std::vector<uint16_t>(N, 0)};
        FalconNTTVerificationCircuit::build_circuit(z, z, z, logn).generate_constraints(cs);
        std::printf("|verify with ntt|\t%8zu |\t%8zu |\t%8zu |\n", cs->num_instance_variables(), cs->num_witness_variables(), cs->num_constraints());
        frw_layout_t L;
        frw_layout(logn, &L);
        EXPECT(cs->num_instance_variables() == (size_t)L.num_instance);
        EXPECT(cs->num_witness_variables() == (size_t)L.num_witness);
        EXPECT(cs->num_constraints() == (size_t)L.num_constraints);
        {   // count_verify_with_dual_ntt_constraints, constraint_counts.rs:115-138 (numbers not in the README)
            auto csd = ConstraintSystem::new_ref();
            csd->set_setup_mode(true);
            FalconDualNTTVerificationCircuit::build_circuit(z, z, z, logn).generate_constraints(csd);
            std::printf("|verify with dual ntt|\t%8zu |\t%8zu |\t%8zu |\n", csd->num_instance_variables(), csd->num_witness_variables(), csd->num_constraints());
            frw_layout_dual_t D;
            frw_layout_dual(logn, &D);
            EXPECT(csd->num_instance_variables() == (size_t)D.num_instance && csd->num_witness_variables() == (size_t)D.num_witness &&
                   csd->num_constraints() == (size_t)D.num_constraints);
        }
        const size_t want[2][3] = {{1025, 78386, 81460}, {2049, 156724, 162870}};     // README.md:55, :44
        EXPECT(cs->num_instance_variables() == want[logn - 9][0] && cs->num_witness_variables() == want[logn - 9][1] &&
               cs->num_constraints() == want[logn - 9][2]);
    }
}

// ---------------------------------------------------------------------------------------------------------------
template <typename T>
static std::vector<T> read_file(const char *path)
{
    std::ifstream f(path, std::ios::binary);
    std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
    std::vector<T> out(raw.size() / sizeof(T));
    std::memcpy(out.data(), raw.data(), out.size() * sizeof(T));
    return out;
}

static int check_external(int logn, char **files, bool dual)
{
    Polynomial sig{read_file<uint16_t>(files[0])}, pk{read_file<uint16_t>(files[1])}, hm{read_file<uint16_t>(files[2])};
    auto wit = read_file<uint64_t>(files[3]);
    auto inst = read_file<uint64_t>(files[4]);
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(nullptr, /*strict=*/false);
    if (dual) {
        auto circuit = FalconDualNTTVerificationCircuit::build_circuit(pk, hm, sig, logn);
        circuit.use_engine_output(wit.data(), inst.data());
        circuit.generate_constraints(cs);
    } else {
        auto circuit = FalconNTTVerificationCircuit::build_circuit(pk, hm, sig, logn);
        circuit.use_engine_output(wit.data(), inst.data());
        circuit.generate_constraints(cs);
    }
    auto bad = cs->which_is_unsatisfied();
    std::printf("constraints %zu witnesses %zu instances %zu -> %s", cs->num_constraints(), cs->num_witness_variables(),
                cs->num_instance_variables(), bad ? "UNSATISFIED" : "satisfied");
    if (bad) std::printf(" at constraint %zu", *bad);
    std::printf("\n");
    // instance vector handed over must be what the circuit allocated
    for (size_t i = 0; i < cs->instance_assignment.size(); i++)
        if (cs->instance_assignment[i] != Fr::from_montgomery(&inst[4 * i])) { std::printf("instance %zu differs\n", i); return 3; }
    return bad ? 2 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// GPU-backed tests
// ---------------------------------------------------------------------------------------------------------------
static void test_mod_q_case(const Engine &eng, uint64_t a, uint64_t b, bool satisfied)
{   // arithmetics.rs:311-338
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(&eng, false);
    FpVar a_var = FpVar::new_witness(cs, Fr::from(a));
    FpVar const_q_var = FpVar::new_constant(cs, Fr::from(MODULUS));
    size_t w0 = cs->num_witness_variables(), c0 = cs->num_constraints();
    FpVar b_var = mod_q(cs, a_var, const_q_var);
    EXPECT(cs->num_witness_variables() - w0 == 29 && cs->num_constraints() - c0 == 30);
    FpVar b_var2 = FpVar::new_witness(cs, Fr::from(b));
    b_var.enforce_equal(b_var2);
    EXPECT(cs->is_satisfied() == satisfied);
    EXPECT((b_var.value() == Fr::from(b)) == satisfied);
}

static void test_mod_q(const Engine &eng)
{
    std::printf("test_mod_q\n");
    test_mod_q_case(eng, 6, 6, true);                       // arithmetics.rs:346-360
    test_mod_q_case(eng, 0, 0, true);
    test_mod_q_case(eng, MODULUS, 0, true);
    test_mod_q_case(eng, MODULUS + 1, 1, true);
    test_mod_q_case(eng, 6, 7, false);
    test_mod_q_case(eng, 5, MODULUS - 1, false);
    std::mt19937_64 rng(1);
    for (int i = 0; i < 50; i++) {                          // :365-371
        uint64_t t = rng() % (1u << 30);
        test_mod_q_case(eng, t, t % MODULUS, true);
        test_mod_q_case(eng, t, (t + 1) % MODULUS, false);
    }
}

static void test_add_mod_case(const Engine &eng, uint64_t a, uint64_t b, uint64_t c, bool satisfied)
{   // arithmetics.rs:436-470
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(&eng, false);
    FpVar a_var = FpVar::new_witness(cs, Fr::from(a)), b_var = FpVar::new_witness(cs, Fr::from(b));
    FpVar const_q_var = FpVar::new_constant(cs, Fr::from(MODULUS));
    FpVar c_var = add_mod(cs, a_var, b_var, const_q_var);
    FpVar c_var2 = FpVar::new_witness(cs, Fr::from(c));
    c_var.enforce_equal(c_var2);
    EXPECT(cs->is_satisfied() == satisfied);
    EXPECT((c_var.value() == Fr::from(c)) == satisfied);
}

static void test_add_mod(const Engine &eng)
{
    std::printf("test_add_mod\n");
    test_add_mod_case(eng, 6, 36, 42, true);                // arithmetics.rs:480-494
    test_add_mod_case(eng, 0, 100, 100, true);
    test_add_mod_case(eng, 100, 0, 100, true);
    test_add_mod_case(eng, 5, MODULUS - 1, 4, true);
    test_add_mod_case(eng, 6, 7, 41, false);
    test_add_mod_case(eng, 5, MODULUS - 1, 3, false);
    std::mt19937_64 rng(2);
    for (int i = 0; i < 50; i++) {                          // :499-505
        uint64_t a = rng() % (1u << 30), b = rng() % (1u << 30);
        test_add_mod_case(eng, a, b, (a + b) % MODULUS, true);
        test_add_mod_case(eng, a, b, (a + b + 1) % MODULUS, false);
    }
}

template <typename G>
static void unary_case(const Engine &eng, G gadget, uint64_t value, bool satisfied)
{
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(&eng, false);
    FpVar a_var = FpVar::new_witness(cs, Fr::from(value));
    gadget(cs, a_var);
    bool got = cs->is_satisfied();
    if (got != satisfied) std::printf("  value %llu: satisfied=%d expected %d\n", (unsigned long long)value, got, satisfied);
    EXPECT(got == satisfied);
}

static void test_range_proofs(const Engine &eng)
{
    std::printf("test_range_proof_mod_q / norm_bound / half_q\n");
    auto ltq = [](const ConstraintSystemRef &cs, const FpVar &a) { enforce_less_than_q(cs, a); };
    for (auto [v, ok] : std::vector<std::pair<uint64_t, bool>>{{42, true}, {0, true}, {1 << 12, true}, {1 << 13, true}, {MODULUS - 1, true},
                                                              {MODULUS, false}, {MODULUS + 1, false}, {(uint64_t)MODULUS * 10000, false}})
        unary_case(eng, ltq, v, ok);                        // range_proofs.rs:365-389
    std::mt19937_64 rng(3);
    for (int i = 0; i < 100; i++) { uint64_t t = rng() % (1u << 15); unary_case(eng, ltq, t, t < MODULUS); }   // :394-398
    for (int logn : {9, 10}) {
        const uint64_t bound = SIG_L2_BOUND(logn);
        auto nb = [logn](const ConstraintSystemRef &cs, const FpVar &a) { enforce_less_than_norm_bound(cs, a, logn); };
        for (auto [v, ok] : std::vector<std::pair<uint64_t, bool>>{{42, true}, {0, true}, {1 << 25, true}, {1 << 24, true}, {bound - 1, true},
                                                                  {bound, false}, {bound + 1, false}, {1 << 27, false}, {1 << 26, logn == 10}})
            unary_case(eng, nb, v, ok);                     // range_proofs.rs:442-474
        for (int i = 0; i < 100; i++) { uint64_t t = rng() % (1u << 27); unary_case(eng, nb, t, t < bound); }
        for (uint64_t t = bound - 20; t < bound + 20; t++) unary_case(eng, nb, t, t < bound);
    }
    auto half = [](const ConstraintSystemRef &cs, const FpVar &a) {     // range_proofs.rs:505-520
        // is_less_than_6144 is only ever called from l2_norm_var on this path; ask the engine for its element block
        detail::feed_gadget(cs, FRW_G_L2_ELEM, a.value(), nullptr, "is_less_than_6144");
        is_less_than_6144(cs, a).enforce_equal(Boolean::TRUE_());
    };
    for (auto [v, ok] : std::vector<std::pair<uint64_t, bool>>{{42, true}, {0, true}, {6143, true}, {6144, false}, {6145, false}, {MODULUS, false}})
        unary_case(eng, half, v, ok);                       // :529-547
    for (int i = 0; i < 100; i++) { uint64_t t = rng() % MODULUS; unary_case(eng, half, t, t < 6144); }
}

static void test_ntt_mul_circuit(const Engine &eng, int logn)
{   // poly.rs:252-301
    std::printf("test_ntt_mul_circuit logn=%d\n", logn);
    const size_t N = (size_t)1 << logn;
    std::mt19937_64 rng(4 + logn);
    for (int rep = 0; rep < 3; rep++) {
        auto cs = ConstraintSystem::new_ref();
        cs->attach_engine(&eng);
        auto param_vars = ntt_param_var(cs, logn);
        auto const_power_q_vars = const_q_power_vars(cs, logn);
        Polynomial poly{std::vector<uint16_t>(N)};
        for (auto &c : poly.c) c = (uint16_t)(rng() % MODULUS);
        PolyVar poly_var = PolyVar::alloc_vars(cs, poly, AllocationMode::Witness);
        size_t w0 = cs->num_witness_variables(), c0 = cs->num_constraints();
        NTTPolyVar output_var = NTTPolyVar::ntt_circuit(cs, poly_var, const_power_q_vars, param_vars, logn);
        EXPECT(cs->num_witness_variables() - w0 == 29 * N && cs->num_constraints() - c0 == 30 * N);
        std::vector<uint16_t> output = ntt_clear(poly.c, logn);
        for (size_t i = 0; i < N; i++) EXPECT(output_var.coeff()[i].value() == Fr::from(output[i]));     // :292-297
        EXPECT(cs->is_satisfied());
    }
    bool threw = false;                                     // poly.rs:110-112: wrong length panics
    try {
        auto cs = ConstraintSystem::new_ref();
        cs->attach_engine(&eng);
        PolyVar shortv = PolyVar::alloc_vars(cs, Polynomial{std::vector<uint16_t>(N - 1, 1)}, AllocationMode::Witness);
        NTTPolyVar::ntt_circuit(cs, shortv, const_q_power_vars(cs, logn), ntt_param_var(cs, logn), logn);
    } catch (const std::invalid_argument &) { threw = true; }
    EXPECT(threw);
}

static void test_ntt_verification_r1cs(const Engine &eng, int logn)
{   // falcon_ntt.rs:133-160 (keygen + sign are replaced by the synthetic valid triple generator)
    std::printf("test_ntt_verification_r1cs logn=%d\n", logn);
    const size_t N = (size_t)1 << logn;
    for (uint64_t idx = 0; idx < 2; idx++) {
        Polynomial sig{std::vector<uint16_t>(N)}, pk{std::vector<uint16_t>(N)}, hm{std::vector<uint16_t>(N)};
        EXPECT(frw_synth_triples(logn, 1, 77, idx, sig.c.data(), pk.c.data(), hm.c.data()) == FRW_OK);
        auto cs = ConstraintSystem::new_ref();
        cs->attach_engine(&eng);
        auto falcon_circuit = FalconNTTVerificationCircuit::build_circuit(pk, hm, sig, logn);
        falcon_circuit.generate_constraints(cs);
        std::printf("  number of variables %zu %zu and constraints %zu\n", cs->num_instance_variables(), cs->num_witness_variables(), cs->num_constraints());
        auto bad = cs->which_is_unsatisfied();
        if (bad) std::printf("  unsatisfied at %zu\n", *bad);
        EXPECT(!bad);
        // the public inputs are the Falcon NTTs of pk and hm (falcon_ntt.rs:45,51,63,67)
        auto pk_ntt = ntt_clear(pk.c, logn), hm_ntt = ntt_clear(hm.c, logn);
        for (size_t i = 0; i < N; i++) {
            EXPECT(cs->instance_assignment[1 + i] == Fr::from(pk_ntt[i]));
            EXPECT(cs->instance_assignment[1 + N + i] == Fr::from(hm_ntt[i]));
        }
        // a corrupted public input must break the system
        cs->instance_assignment[3] = cs->instance_assignment[3] + Fr::one();
        EXPECT(!cs->is_satisfied());
    }
    // bad path: norm above the bound.  strict (non-test build) panics; permissive (cfg(test)) is unsatisfied.
    Polynomial sig{std::vector<uint16_t>(N, 3000)}, pk{std::vector<uint16_t>(N, 1)}, hm{std::vector<uint16_t>(N, 5)};
    bool threw = false;
    try {
        auto cs = ConstraintSystem::new_ref();
        cs->attach_engine(&eng, true);
        FalconNTTVerificationCircuit::build_circuit(pk, hm, sig, logn).generate_constraints(cs);
    } catch (const std::domain_error &) { threw = true; }
    EXPECT(threw);
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(&eng, false);
    FalconNTTVerificationCircuit::build_circuit(pk, hm, sig, logn).generate_constraints(cs);
    EXPECT(!cs->is_satisfied());
}

// test-side encoders (Falcon spec 3.11.2-3.11.4), to feed build_circuit(pk, msg, sig) with bytes
static std::vector<uint8_t> modq_encode(const std::vector<uint16_t> &c, int logn)
{
    std::vector<uint8_t> out{(uint8_t)logn};
    uint32_t acc = 0; int bits = 0;
    for (uint16_t x : c) { acc = (acc << 14) | x; bits += 14; while (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); } }
    return out;
}
static std::vector<uint8_t> comp_encode(const std::vector<uint16_t> &c, int logn, uint8_t nonce_byte)
{
    std::vector<uint8_t> out{(uint8_t)(0x30 + logn)};
    out.insert(out.end(), FRW_NONCE_LEN, nonce_byte);
    uint32_t acc = 0; int bits = 0;
    auto put = [&](uint32_t v, int n) { acc = (acc << n) | v; bits += n; while (bits >= 8) { bits -= 8; out.push_back((uint8_t)(acc >> bits)); } };
    for (uint16_t x : c) {
        const bool neg = x >= 6144;
        const uint32_t m = neg ? MODULUS - x : x;
        put(neg ? 1 : 0, 1); put(m & 127, 7);
        for (uint32_t k = 0; k < (m >> 7); k++) put(0, 1);
        put(1, 1);
    }
    if (bits) put(0, 8 - bits);
    out.resize((size_t)FRW_SIG_LEN(logn), 0);
    return out;
}

static void test_build_circuit_from_bytes(const Engine &eng, int logn)
{   // falcon_ntt.rs:15-17,27-28,44: build_circuit(pk, msg, sig) with encoded inputs; hash-to-point and decoding on the engine.
    // No Falcon signer exists here, so hm = H(nonce || msg) is unrelated to sig*pk and v is not short: in the permissive
    // (cfg(test)) build every constraint in front of the norm-bound gadget must hold (the norm does not even fit its
    // 26/27-bit decomposition, so the first failure is inside that gadget's last 52/54 constraints).
    std::printf("test_build_circuit_from_bytes logn=%d\n", logn);
    const size_t N = (size_t)1 << logn;
    std::vector<uint16_t> sig(N), pk(N), hm(N);
    EXPECT(frw_synth_triples(logn, 1, 5, 0, sig.data(), pk.data(), hm.data()) == FRW_OK);
    PublicKey pkb{modq_encode(pk, logn)};
    Signature sgb{comp_encode(sig, logn, 0x5a)};
    EXPECT(pkb.bytes.size() == (size_t)FRW_PK_LEN(logn) && sgb.bytes.size() == (size_t)FRW_SIG_LEN(logn));
    EXPECT(sgb.nonce().size() == FRW_NONCE_LEN && sgb.nonce()[0] == 0x5a);
    std::string text = "testing message";
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(&eng, false);
    FalconNTTVerificationCircuit::build_circuit(pkb, std::vector<uint8_t>(text.begin(), text.end()), sgb).generate_constraints(cs);
    auto bad = cs->which_is_unsatisfied();
    EXPECT(bad.has_value() && *bad >= cs->num_constraints() - (logn == 9 ? 52 : 54));
    // S0 must be the decoded signature
    for (size_t i = 0; i < N; i++) EXPECT(cs->witness_assignment[i] == Fr::from(sig[i]));
    // instance = NTT(pk) as decoded
    auto pk_ntt = ntt_clear(pk, logn);
    for (size_t i = 0; i < N; i++) EXPECT(cs->instance_assignment[1 + i] == Fr::from(pk_ntt[i]));
    // malformed signature -> the reference's decode would fail; here a domain error
    Signature broken = sgb;
    broken.bytes.back() = 1;
    bool threw = false;
    try {
        auto cs2 = ConstraintSystem::new_ref();
        cs2->attach_engine(&eng, false);
        FalconNTTVerificationCircuit::build_circuit(pkb, {}, broken).generate_constraints(cs2);
    } catch (const std::domain_error &) { threw = true; }
    EXPECT(threw);
}

static void test_dual_ntt_verification_r1cs(const Engine &eng, int logn)
{   // falcon_dual_ntt.rs:142-169
    std::printf("test_dual_ntt_verification_r1cs logn=%d\n", logn);
    const size_t N = (size_t)1 << logn;
    Polynomial sig{std::vector<uint16_t>(N)}, pk{std::vector<uint16_t>(N)}, hm{std::vector<uint16_t>(N)};
    EXPECT(frw_synth_triples(logn, 1, 78, 0, sig.c.data(), pk.c.data(), hm.c.data()) == FRW_OK);
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(&eng);
    FalconDualNTTVerificationCircuit::build_circuit(pk, hm, sig, logn).generate_constraints(cs);
    std::printf("  number of variables %zu %zu and constraints %zu\n", cs->num_instance_variables(), cs->num_witness_variables(), cs->num_constraints());
    auto bad = cs->which_is_unsatisfied();
    if (bad) std::printf("  unsatisfied at %zu\n", *bad);
    EXPECT(!bad);
    cs->witness_assignment[5] = cs->witness_assignment[5] + Fr::one();       // sig.pos[5] tampered
    EXPECT(!cs->is_satisfied());
}

static void test_aggregate_circuit(const Engine &eng)
{   // SURVEY 8-f row 4 / BASELINE configs[4] shape: mixed Falcon-512 / Falcon-1024 statements on ONE constraint system
    std::printf("test_aggregate_circuit\n");
    const int logns[5] = {10, 9, 9, 10, 9};
    std::vector<FalconAggregateVerificationCircuit::Statement> st;
    size_t want_w = 0, want_c = 0, want_i = 1;
    for (int k = 0; k < 5; k++) {
        const size_t N = (size_t)1 << logns[k];
        FalconAggregateVerificationCircuit::Statement s{Polynomial{std::vector<uint16_t>(N)}, Polynomial{std::vector<uint16_t>(N)},
                                                        Polynomial{std::vector<uint16_t>(N)}, logns[k]};
        EXPECT(frw_synth_triples(logns[k], 1, 99, (uint64_t)k, s.sig.c.data(), s.pk.c.data(), s.hm.c.data()) == FRW_OK);
        st.push_back(std::move(s));
        frw_layout_t L;
        frw_layout(logns[k], &L);
        want_w += (size_t)L.num_witness; want_c += (size_t)L.num_constraints; want_i += (size_t)L.num_instance - 1;
    }
    auto cs = ConstraintSystem::new_ref();
    cs->attach_engine(&eng);
    FalconAggregateVerificationCircuit::build_circuit(st).generate_constraints(cs);
    std::printf("  number of variables %zu %zu and constraints %zu\n", cs->num_instance_variables(), cs->num_witness_variables(), cs->num_constraints());
    EXPECT(cs->num_witness_variables() == want_w && cs->num_constraints() == want_c && cs->num_instance_variables() == want_i);
    EXPECT(cs->is_satisfied());
    // the aggregate witness is the concatenation of the per-statement witnesses: the third statement's first element is its sig[0]
    frw_layout_t L10, L9;
    frw_layout(10, &L10); frw_layout(9, &L9);
    EXPECT(cs->witness_assignment[(size_t)L10.num_witness + (size_t)L9.num_witness] == Fr::from(st[2].sig.c[0]));
    cs->witness_assignment[(size_t)L10.num_witness + 17] = cs->witness_assignment[(size_t)L10.num_witness + 17] + Fr::one();
    EXPECT(!cs->is_satisfied());
}

static void test_no_engine_is_assignment_missing()
{
    std::printf("test_no_engine_is_assignment_missing\n");
    auto cs = ConstraintSystem::new_ref();
    FpVar a = FpVar::new_witness(cs, Fr::from(6));
    bool threw = false;
    try { mod_q(cs, a, FpVar::new_constant(cs, Fr::from(MODULUS))); } catch (const SynthesisError &e) { threw = e.kind == SynthesisError::AssignmentMissing; }
    EXPECT(threw);
}

int main(int argc, char **argv)
{
    std::string mode = argc > 1 ? argv[1] : "structure";
    if (mode == "structure") {
        for (int logn : {10, 9}) { std::printf("Falcon-%d\n", 1 << logn); count_table(logn); }
        test_no_engine_is_assignment_missing();
    } else if ((mode == "check" || mode == "check-dual") && argc == 8) {
        return check_external(std::atoi(argv[2]), argv + 3, mode == "check-dual");
    } else if (mode == "gpu") {
        Engine eng(0);
        test_mod_q(eng);
        test_add_mod(eng);
        test_range_proofs(eng);
        for (int logn : {9, 10}) test_ntt_mul_circuit(eng, logn);
        for (int logn : {9, 10}) test_ntt_verification_r1cs(eng, logn);
        for (int logn : {9, 10}) test_build_circuit_from_bytes(eng, logn);
        for (int logn : {9, 10}) test_dual_ntt_verification_r1cs(eng, logn);
        test_aggregate_circuit(eng);
    } else {
        std::printf("usage: %s structure | gpu | check <logn> <sig> <pk> <hm> <witness> <instance>\n", argv[0]);
        return 64;
    }
    std::printf(failures ? "%d FAILURES\n" : "all ok\n", failures);
    return failures ? 1 : 0;
}
