set -e
R=$(pwd); O=$R/gpurun_out/s3e; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for v in sorted natural; do
  if [ $v = natural ]; then export FRW_DEV_FLAT_NATURAL=1; fi
  rocprofv3 --kernel-trace --stats -d $R/gpurun_out/p_s3e_$v -o qap --output-format csv -- python3 $R/tools/time_qap.py 10 64 20 > $O/qap_timing_$v.txt 2>&1
  cp "$(find $R/gpurun_out/p_s3e_$v -name 'qap_kernel_stats.csv' | head -1)" $O/qap_kernel_stats_$v.csv
  grep -h "witness map\|quotient" $O/qap_timing_$v.txt | tail -2
  grep "r1cs" $O/qap_kernel_stats_$v.csv | cut -d, -f1-4 | cut -c1-120
done
cd $R
FRW_DEV_FLAT_NATURAL=1 timeout -k 10 600 python -m pytest tests/test_gpu_qap.py -m gpu -x -q > $O/gputests_natural.txt 2>&1 || { tail -30 $O/gputests_natural.txt; exit 1; }
tail -2 $O/gputests_natural.txt
