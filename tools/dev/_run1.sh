set -e
R=$(pwd); O=$R/gpurun_out/s2h; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py tests/test_gpu_msm.py tests/test_gpu_aggregate.py -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -40 $O/gputests.txt; exit 1; }
tail -3 $O/gputests.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_s2h_b1 -o groth16_batch1 --output-format csv -- python3 $R/tools/time_groth16.py 10 1 5 > $O/b1_timing.txt 2>&1
python3 $R/tools/kernel_timeline.py "$(find $R/gpurun_out/p_s2h_b1 -name 'groth16_batch1_kernel_trace.csv' | head -1)" 0.04 > $O/b1_latency.txt
cd $R
python3 tools/time_groth16.py 10 1 20 > $O/b1_plain.txt 2>&1; tail -n 1 $O/b1_plain.txt
python3 tools/time_groth16.py 10 1 20 > $O/b1_plain2.txt 2>&1; tail -n 1 $O/b1_plain2.txt
python3 tools/time_groth16.py 10 2 20 > $O/b2_plain.txt 2>&1; tail -n 1 $O/b2_plain.txt
python3 tools/time_groth16.py 10 64 5 > $O/b64_plain.txt 2>&1; tail -n 1 $O/b64_plain.txt
python3 tools/time_groth16.py 9 1 20 > $O/b1_512.txt 2>&1; tail -n 1 $O/b1_512.txt
python3 tools/time_msm.py 10 1 10 > $O/msm1.txt 2>&1; tail -n 3 $O/msm1.txt
