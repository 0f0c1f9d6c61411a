set -e
bash tools/profile_round.sh r04 > gpurun_out/profile_round_r04.log 2>&1 || { tail -30 gpurun_out/profile_round_r04.log; exit 1; }
python3 bench.py > gpurun_out/profiles_r04/r04_bench_n1.json 2> gpurun_out/bench_n1.log; tail -c 200 gpurun_out/profiles_r04/r04_bench_n1.json
python3 tools/soak_groth16.py 60 10 16 > gpurun_out/profiles_r04/r04_soak_groth16.txt 2>&1; tail -n 1 gpurun_out/profiles_r04/r04_soak_groth16.txt | cut -c1-250
