#!/bin/bash
# SQ counters of the h_query sum's kernels (two rocprofv3 --pmc passes over tools/time_msm.py, counters only -- no tracing domains):
#   bash tools/pmc_msm.sh r04     -> gpurun_out/<tag>_msm_counters.txt
# What it answers: how many vector instructions msm_bucket_kernel issues per mixed addition (all of them, not only the field
# products' that the instruction-priced roofline counts), and how busy the vector ALU is while it runs.
set -e
TAG=${1:-r04}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $OUT/${TAG}_msm_pmc_mix -- python3 $REPO/tools/time_msm.py 10 64 1 > /dev/null 2> $OUT/${TAG}_msm_pmc_mix.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT/${TAG}_msm_pmc_act -- python3 $REPO/tools/time_msm.py 10 64 1 > /dev/null 2> $OUT/${TAG}_msm_pmc_act.log
cd $REPO
python3 - "$OUT" "$TAG" > $OUT/${TAG}_msm_counters.txt <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1:3]
per = collections.OrderedDict()
for d in ("msm_pmc_mix", "msm_pmc_act"):
    hits = glob.glob(os.path.join(out, "%s_%s" % (tag, d), "**", "*_counter_collection.csv"), recursive=True)
    path = max(hits, key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "msm_" in k:
            name = k.split("(")[0].replace("void frw::", "").replace("frw::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        per.setdefault(name, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
print("# SQ counters per launch (64 Falcon-1024 h vectors per call, 2^18 - 1 points), summed over the chip; tools/pmc_msm.sh")
for name, c in per.items():
    print(name)
    for k, v in sorted(c.items()):
        print("    %-22s %.4g" % (k, v))
b = per.get("msm_bucket_kernel<FqField, true>") or {}
if b.get("SQ_INSTS_VALU"):
    madds = 64 * 16 * 262143            # one per (signature, window, point) with a non-zero digit: an upper bound (digits are zero with probability 2^-16)
    print()
    print("msm_bucket_kernel<FqField, true>: %.0f vector instructions per wavefront per mixed addition of its 64 lanes"
          % (b["SQ_INSTS_VALU"] / (madds / 64.0)))
    print("  (the instruction-priced roofline of bench.py counts 3,668 multiplies + 877 others = 4,545 of them: the field products alone -- six products,")
    print("   two squares and one a b - c d with a shared reduction per mixed addition)")
    if b.get("SQ_ACTIVE_INST_VALU") and b.get("SQ_BUSY_CYCLES"):
        print("  SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES = %.3f   SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES = %.3f" % (
            b["SQ_ACTIVE_INST_VALU"] / b["SQ_WAVE_CYCLES"], b["SQ_WAIT_INST_ANY"] / b["SQ_WAVE_CYCLES"]))
PY
