set -e
bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1 || { tail -30 gpurun_out/profile_round_r04.log; exit 1; }
tail -3 gpurun_out/profile_round_r04.log | cut -c1-300
python3 bench.py > gpurun_out/profiles_r04/r04_bench_n1.json 2> gpurun_out/bench_n1.log; tail -c 300 gpurun_out/profiles_r04/r04_bench_n1.json
