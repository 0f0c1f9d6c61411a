#!/bin/bash
# The prover's profile set (profiles/README.md rows <tag>_qap_*, <tag>_msm_*, <tag>_groth16_*, <tag>_aggregate*): per-kernel traces of the
# witness map, of the h_query sum, of whole proofs (64 per call and one alone) and of ONE proof for a sixteen-statement aggregate, with
# the timelines of the last call of each, and the plain timings at the other batch sizes.  Run on the GPU box from the repo root (it is
# the last step of tools/profile_round.sh); everything lands in gpurun_out/profiles_<tag>/, ready to be copied into profiles/.
#   bash tools/refresh_prover_profiles.sh r04
set -e
TAG=${1:-r04}
R=$(pwd)
O="$R/gpurun_out/profiles_$TAG"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
export FRW_TIME_GRAPH=0      # the traces are of plain stream launches; the graph replay of a lone proof is timed below, without the profiler
trace() {   # trace <name> <program and arguments...>: kernel trace + stats as CSV, the program's own output beside it
    local name=$1; shift
    rocprofv3 --kernel-trace --stats -d "$R/gpurun_out/p_${TAG}_$name" -o "$name" --output-format csv -- python3 "$@" > "$O/${TAG}_${name}_timing.txt" 2>&1
    cp "$(find "$R/gpurun_out/p_${TAG}_$name" -name "${name}_kernel_stats.csv" | head -1)" "$O/${TAG}_${name}_kernel_stats.csv"
}
trace qap "$R/tools/time_qap.py" 10 256 5
trace msm "$R/tools/time_msm.py" 10 64 5
trace groth16 "$R/tools/time_groth16.py" 10 64 5
trace groth16_batch1 "$R/tools/time_groth16.py" 10 1 5
trace aggregate16 "$R/bench.py" --workload aggregate --aggregate 10x16 --steps 5
# BASELINE configs[4] as written: ONE proof for the 1,024 mixed statements (bare key, the 2^27 domain): setup, three proofs, verification
trace aggregate1024 "$R/tools/time_aggregate_large.py" 1024 2
python3 "$R/tools/kernel_timeline.py" "$(find "$R/gpurun_out/p_${TAG}_aggregate1024" -name "aggregate1024_kernel_trace.csv" | head -1)" 5 > "$O/${TAG}_aggregate1024_timeline.txt"
python3 "$R/tools/kernel_timeline.py" "$(find "$R/gpurun_out/p_${TAG}_groth16" -name "groth16_kernel_trace.csv" | head -1)" 0.3 > "$O/${TAG}_groth16_timeline_64.txt"
python3 "$R/tools/kernel_timeline.py" "$(find "$R/gpurun_out/p_${TAG}_groth16_batch1" -name "groth16_batch1_kernel_trace.csv" | head -1)" 0.08 > "$O/${TAG}_groth16_batch1_latency.txt"
python3 "$R/tools/kernel_timeline.py" "$(find "$R/gpurun_out/p_${TAG}_aggregate16" -name "aggregate16_kernel_trace.csv" | head -1)" 0.15 > "$O/${TAG}_aggregate16_timeline.txt"
{
    python3 "$R/tools/time_msm.py" 9 128 3
    python3 "$R/tools/time_groth16.py" 9 64 5
    python3 "$R/tools/time_groth16.py" 10 16 5
    python3 "$R/tools/time_groth16.py" 10 128 3
    FRW_TIME_GRAPH=1 python3 "$R/tools/time_groth16.py" 10 1 20
    python3 "$R/bench.py" --workload aggregate --aggregate 10x1,9x1,10x2 --steps 5
} > "$O/${TAG}_prover_other_sizes.txt" 2>&1
