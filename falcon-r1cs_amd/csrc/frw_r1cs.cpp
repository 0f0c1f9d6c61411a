// frw_r1cs.cpp -- R1CS matrix export: the step right after the hot path for any real prover
// (examples/pok_sig.rs:30-32 hands the circuit to Groth16, which calls cs.finalize() / to_matrices()).
// Structure only: runs the host mirror (host/frw_host.hpp) in setup mode -- no values, no GPU -- inlines every
// symbolic linear combination and writes A, B, C in a small CSR file a prover can ingest without arkworks.
#include <stdint.h>
#include <exception>

#include "../../include/frw.h"
#include "host/frw_host.hpp"

extern "C" int frw_r1cs_export(int circuit, int logn, const char *path, uint64_t *counts /* 6 x u64, may be NULL */)
{
    using namespace frw::host;
    if ((logn != 9 && logn != 10) || !path || (circuit != FRW_CIRCUIT_NTT && circuit != FRW_CIRCUIT_DUAL_NTT)) return FRW_E_INVALID_ARG;
    try {
        const size_t N = (size_t)1 << logn;
        auto cs = ConstraintSystem::new_ref();
        cs->set_setup_mode(true);
        Polynomial z{std::vector<uint16_t>(N, 0)};
        if (circuit == FRW_CIRCUIT_NTT) FalconNTTVerificationCircuit::build_circuit(z, z, z, logn).generate_constraints(cs);
        else FalconDualNTTVerificationCircuit::build_circuit(z, z, z, logn).generate_constraints(cs);
        ConstraintMatrices m = cs->to_matrices();
        if (counts) {
            counts[0] = m.num_instance_variables; counts[1] = m.num_witness_variables; counts[2] = m.num_constraints;
            counts[3] = m.non_zero(m.a); counts[4] = m.non_zero(m.b); counts[5] = m.non_zero(m.c);
        }
        return m.write(path) ? FRW_OK : FRW_E_INVALID_ARG;
    } catch (const std::exception &) {
        return FRW_E_INVALID_ARG;
    }
}
