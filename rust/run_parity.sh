#!/usr/bin/env bash
# One command that turns "parity unpinned" into a yes or a no, on a machine with cargo, crates.io access AND an MI355X:
#
#     rust/run_parity.sh /path/to/a/checkout/of/zhenfeizhang/falcon-r1cs
#
# builds libfrw.so from this tree, links the two crates of rust/ into the reference's workspace (symlinks + two workspace
# members; the reference's sources are not touched) and runs the reference-side tests of falcon-r1cs-accel for both
# parameter sets: the engine's witness against arkworks' own witness_assignment for a falcon-rust signature
# (accelerated_circuit_is_satisfied_and_equals_the_cpu_witness) and the GPU witness map against ark-groth16's QAP identity
# (gpu_witness_map_satisfies_arkworks_qap_identity).  Neither cargo nor a network exists in the container this repository
# is built in, so this script has never run there.
set -euo pipefail
REF="${1:?usage: run_parity.sh <checkout of zhenfeizhang/falcon-r1cs>}"
HERE="$(cd "$(dirname "${BASH_SOURCE[0]}")" && pwd)"
ROOT="$(dirname "$HERE")"
command -v cargo >/dev/null || { echo "cargo not found: install a Rust toolchain first" >&2; exit 2; }
[ -x /opt/rocm/bin/hipcc ] || { echo "hipcc not found under /opt/rocm" >&2; exit 2; }
[ -f "$REF/falcon-r1cs/Cargo.toml" ] || { echo "$REF does not look like the reference workspace" >&2; exit 2; }
make -j"$(nproc)" -C "$ROOT/falcon-r1cs_amd/csrc"
export FRW_LIB_DIR="$ROOT/falcon-r1cs_amd"                      # rust/frw-sys/build.rs links libfrw.so from here
export LD_LIBRARY_PATH="$FRW_LIB_DIR:/opt/rocm/lib:${LD_LIBRARY_PATH:-}"
for crate in frw-sys falcon-r1cs-accel; do
    [ -e "$REF/$crate" ] || ln -s "$HERE/$crate" "$REF/$crate"
done
grep -q 'falcon-r1cs-accel' "$REF/Cargo.toml" || \
    sed -i 's|members = \[|members = [\n    "frw-sys",\n    "falcon-r1cs-accel",|' "$REF/Cargo.toml"
cd "$REF"
cargo test --release -p falcon-r1cs-accel -- --nocapture --test-threads 1
cargo test --release -p falcon-r1cs-accel --no-default-features --features falcon-512 -- --nocapture --test-threads 1
echo "parity pinned: the engine's witness equals arkworks' witness_assignment element for element, both parameter sets"
