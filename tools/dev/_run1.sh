set -e
R=$(pwd); O=$R/gpurun_out/s8a; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/gputests.txt 2>&1 || { tail -40 $O/gputests.txt; exit 1; }
tail -3 $O/gputests.txt
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.txt 2>&1; tail -n 1 $O/smoke.txt
