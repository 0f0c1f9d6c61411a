#!/bin/bash
# Profiles of one round, on the GPU box, from ONE command:  bash tools/profile_round.sh r04
# Three rocprofv3 passes over the SAME bench.py command (kernel trace + stats; then WRITE_SIZE and FETCH_SIZE in passes of their own,
# as /opt/skills/guides/MI355X_MICROARCH.md prescribes), condensed into profiles/<tag>_* by tools/summarize_profiles.py; then the
# prover's set (witness map, h_query sum, whole proofs, the aggregate proof: tools/refresh_prover_profiles.sh).
set -e
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
# trace: the default benchmark (all three kernels of the JSON line run in it), without the CPU-side extras
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_trace -- python3 $REPO/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-r1cs-check --no-aggregate > $OUT/${TAG}_trace.json 2> $OUT/${TAG}_trace.log
# counters: one step is enough (2 launches of the dominant kernel + the secondary kernels)
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_w -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-r1cs-check --no-aggregate > /dev/null 2> $OUT/${TAG}_pmc_w.log
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_f -- python3 $REPO/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-r1cs-check --no-aggregate > /dev/null 2> $OUT/${TAG}_pmc_f.log
cd $REPO
python3 tools/summarize_profiles.py $TAG $OUT/${TAG}_trace $OUT/${TAG}_pmc_w $OUT/${TAG}_pmc_f 32768 10 witness_ntt_verify_kernel
python3 tools/summarize_profiles.py ${TAG}_verify512 $OUT/${TAG}_trace $OUT/${TAG}_pmc_w $OUT/${TAG}_pmc_f 8192 9 witness_ntt_verify_kernel
python3 tools/summarize_profiles.py ${TAG}_cfg2 $OUT/${TAG}_trace $OUT/${TAG}_pmc_w $OUT/${TAG}_pmc_f 4096 9 ntt_modq_kernel
# gpurun merges only gpurun_out/ back: leave copies of the summaries there
mkdir -p $OUT/profiles_$TAG && cp $REPO/profiles/${TAG}_*hbm_traffic.json $REPO/profiles/${TAG}_*kernel_stats.csv $OUT/profiles_$TAG/
cp $OUT/${TAG}_trace.json $OUT/profiles_$TAG/${TAG}_bench_under_rocprof.json
# the prover's tables and timelines, into the same directory
bash tools/refresh_prover_profiles.sh $TAG
