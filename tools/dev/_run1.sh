set -e
R=$(pwd); O=$R/gpurun_out/s9b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_aggregate.py -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -40 $O/gputests.txt; exit 1; }
tail -3 $O/gputests.txt
python3 tools/time_groth16.py 10 1 20 > $O/b1.txt 2>&1; tail -n 2 $O/b1.txt
python3 tools/time_groth16.py 10 2 20 > $O/b2.txt 2>&1; tail -n 2 $O/b2.txt
python3 tools/time_groth16.py 9 1 20 > $O/b1_512.txt 2>&1; tail -n 2 $O/b1_512.txt
python3 bench.py --workload aggregate --aggregate 10x16,10x1,9x1,10x2 --steps 5 > $O/agg.txt 2>&1; grep -o '"ms_per_proof": [0-9.]*' $O/agg.txt | head -5
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_s9b_agg -o aggregate16 --output-format csv -- python3 $R/bench.py --workload aggregate --aggregate 10x16 --steps 5 > $O/agg16_timing.txt 2>&1
python3 $R/tools/kernel_timeline.py "$(find $R/gpurun_out/p_s9b_agg -name 'aggregate16_kernel_trace.csv' | head -1)" 0.15 > $O/agg16_timeline.txt
head -1 $O/agg16_timeline.txt | cut -c1-150; grep "nmsm_ones_kernel<FqField>\|scale_quad" $O/agg16_timeline.txt | cut -c1-100
