#!/bin/bash
# The prover's profile set (profiles/README.md rows r0N_msm_*, r0N_groth16_*): per-kernel traces of the h_query sum and of whole
# proofs, and the plain timings at the other batch sizes.  Run on the GPU box from the repo root; outputs under gpurun_out/.
#   bash tools/refresh_prover_profiles.sh
set -e
R=$(pwd)
mkdir -p "$R/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/p_msm" -o msm --output-format csv -- python3 "$R/tools/time_msm.py" 10 64 5 > "$R/gpurun_out/p_msm.log" 2>&1
python3 "$R/tools/time_msm.py" 9 128 3 > "$R/gpurun_out/p_msm9.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/p_g16" -o g16 --output-format csv -- python3 "$R/tools/time_groth16.py" 10 64 5 > "$R/gpurun_out/p_g16.log" 2>&1
python3 "$R/tools/time_groth16.py" 9 128 3 > "$R/gpurun_out/p_g169.log" 2>&1
python3 "$R/tools/time_groth16.py" 10 1 5 > "$R/gpurun_out/p_g16b1.log" 2>&1
python3 "$R/tools/time_groth16.py" 10 16 5 > "$R/gpurun_out/p_g16b16.log" 2>&1
python3 "$R/tools/time_groth16.py" 10 128 3 > "$R/gpurun_out/p_g16b128.log" 2>&1
