// constraint_counts.cpp -- counterpart of the reference's falcon-r1cs/examples/constraint_counts.rs (BASELINE
// configs[0]): prints the "# instance variables | # witness | # constraints" table for the NTT conversion, the
// verify-with-ntt circuit and the dual-NTT circuit, and -- when a GPU is present -- fills every witness from the HIP
// engine and asserts cs.is_satisfied(), as the reference does after keygen + sign (constraint_counts.rs:49-72).
//
//   g++ -O2 -std=c++17 -o constraint_counts examples/constraint_counts.cpp -Lfalcon-r1cs_amd -lfrw -Wl,-rpath,$PWD/falcon-r1cs_amd
//   ./constraint_counts [9|10]          (the reference selects the parameter set with a cargo feature)
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <memory>

#include "../falcon-r1cs_amd/csrc/host/frw_host.hpp"

using namespace frw::host;

static void row(const char *name, size_t i, size_t w, size_t c) { std::printf("|%-24s|%22zu |%15zu |%18zu |\n", name, i, w, c); }

int main(int argc, char **argv)
{
    const int logn = argc > 1 ? std::atoi(argv[1]) : 10;
    if (logn != 9 && logn != 10) { std::fprintf(stderr, "usage: %s [9|10]\n", argv[0]); return 64; }
    const size_t N = (size_t)1 << logn;
    std::unique_ptr<Engine> eng;
    if (frw_device_count() > 0) eng = std::make_unique<Engine>(0);
    std::printf("Falcon-%zu%s\n", N, eng ? "  (witness values from the HIP engine)" : "  (no GPU: structure only, setup mode)");
    std::printf("|                        | # instance variables |      # witness |      #constraints |\n|---|---:|---:|---:|\n");

    Polynomial sig{std::vector<uint16_t>(N, 0)}, pk = sig, hm = sig;
    if (eng && frw_synth_triples(logn, 1, 1, 0, sig.c.data(), pk.c.data(), hm.c.data()) != FRW_OK) return 1;

    {   // count_ntt_conversion_constraints (constraint_counts.rs:74-113)
        auto cs = ConstraintSystem::new_ref();
        if (eng) cs->attach_engine(eng.get()); else cs->set_setup_mode(true);
        auto param = ntt_param_var(cs, logn);
        auto consts = const_q_power_vars(cs, logn);
        PolyVar pv = PolyVar::alloc_vars(cs, pk, AllocationMode::Witness);          // a uniformly random polynomial
        const size_t i0 = cs->num_instance_variables(), w0 = cs->num_witness_variables(), c0 = cs->num_constraints();
        NTTPolyVar::ntt_circuit(cs, pv, consts, param, logn);
        row("ntt conversion", cs->num_instance_variables() - i0, cs->num_witness_variables() - w0, cs->num_constraints() - c0);
        if (eng && !cs->is_satisfied()) { std::printf("ntt conversion: NOT satisfied\n"); return 1; }
    }
    auto run = [&](const char *name, auto circuit) {
        auto cs = ConstraintSystem::new_ref();
        if (eng) cs->attach_engine(eng.get()); else cs->set_setup_mode(true);
        const auto t0 = std::chrono::steady_clock::now();
        circuit.generate_constraints(cs);
        const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        row(name, cs->num_instance_variables(), cs->num_witness_variables(), cs->num_constraints());
        if (eng) {
            if (!cs->is_satisfied()) { std::printf("%s: NOT satisfied\n", name); std::exit(1); }
            std::printf("    satisfied; structure pass + one engine call: %.1f ms\n", ms);
        }
    };
    run("verify with ntt", FalconNTTVerificationCircuit::build_circuit(pk, hm, sig, logn));
    run("verify with dual ntt", FalconDualNTTVerificationCircuit::build_circuit(pk, hm, sig, logn));
    return 0;
}
