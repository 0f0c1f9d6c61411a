set -e
R=$(pwd); O=$R/gpurun_out/s6a; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_groth16.py -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -60 $O/gputests.txt; exit 1; }
tail -3 $O/gputests.txt
python3 tools/time_groth16.py 10 1 20 > $O/b1.txt 2>&1 || { tail -30 $O/b1.txt; exit 1; }
tail -n 2 $O/b1.txt
python3 tools/time_groth16.py 9 1 20 > $O/b1_512.txt 2>&1; tail -n 2 $O/b1_512.txt
python3 tools/time_groth16.py 10 4 10 > $O/b4.txt 2>&1; tail -n 2 $O/b4.txt
python3 tools/time_groth16.py 10 64 5 > $O/b64.txt 2>&1; tail -n 1 $O/b64.txt
