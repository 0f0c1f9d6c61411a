set -e
R=$(pwd); O=$R/gpurun_out/s2b; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -30 $O/gputests.txt; exit 1; }
tail -3 $O/gputests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_s2b_b1 -o groth16_batch1 --output-format csv -- python3 $R/tools/time_groth16.py 10 1 5 > $O/b1_timing.txt 2>&1
python3 $R/tools/kernel_timeline.py "$(find $R/gpurun_out/p_s2b_b1 -name 'groth16_batch1_kernel_trace.csv' | head -1)" 0.05 > $O/b1_latency.txt
cd $R
python3 tools/time_groth16.py 10 1 10 > $O/b1_plain.txt 2>&1
python3 tools/time_groth16.py 10 64 5 > $O/b64_plain.txt 2>&1
python3 tools/time_groth16.py 10 16 5 > $O/b16_plain.txt 2>&1
tail -1 $O/b1_plain.txt $O/b64_plain.txt $O/b16_plain.txt
