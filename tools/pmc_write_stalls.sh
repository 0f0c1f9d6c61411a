#!/bin/bash
# Where a write request waits on its way out of L2, for the witness kernel and for the compute-free calibration stream of
# the same bench.py run (two rocprofv3 --pmc passes, 4 TCC counters each):  bash tools/pmc_write_stalls.sh r02
set -e
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 0 --batch 32768 --chunk 32768 --no-cpu-baseline --no-r1cs-check --no-secondary"
rocprofv3 --pmc TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum --output-format csv -d $OUT/${TAG}_pmc_st1 -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_st1.log
rocprofv3 --pmc TCC_WRITE_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum TCC_IB_STALL_sum --output-format csv -d $OUT/${TAG}_pmc_st2 -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_st2.log
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1:3]
res = {"witness_ntt_verify_kernel<10, 1>": collections.OrderedDict(), "write_stream_kernel": collections.OrderedDict()}
for d in ("pmc_st1", "pmc_st2"):
    hits = glob.glob(os.path.join(out, "%s_%s" % (tag, d), "**", "*_counter_collection.csv"), recursive=True)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(max(hits, key=os.path.getmtime))):
        for k in res:
            if k in r["Kernel_Name"]:
                acc[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in acc.items():
        res[k][c] = sum(v) / len(v)
print("# one 32,768-signature launch (166.7 GB) of the witness kernel | the compute-free stream over the same 164.3 GB buffer")
print("%-40s %16s %16s" % ("counter (summed over the L2 channels)", "witness kernel", "write stream"))
for c in res["write_stream_kernel"]:
    print("%-40s %16.4g %16.4g" % (c, res["witness_ntt_verify_kernel<10, 1>"].get(c, float("nan")), res["write_stream_kernel"][c]))
PY
