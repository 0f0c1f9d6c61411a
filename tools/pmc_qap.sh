#!/bin/bash
# SQ counters of the QAP witness map's kernels (two rocprofv3 --pmc passes over tools/time_qap.py, counters only):
#   bash tools/pmc_qap.sh r02     -> gpurun_out/<tag>_qap_counters.txt
set -e
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/${TAG}_qap_pmc_mix -- python3 $REPO/tools/time_qap.py 10 64 1 > /dev/null 2> $OUT/${TAG}_qap_pmc_mix.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $OUT/${TAG}_qap_pmc_act -- python3 $REPO/tools/time_qap.py 10 64 1 > /dev/null 2> $OUT/${TAG}_qap_pmc_act.log
cd $REPO
python3 - "$OUT" "$TAG" > $OUT/${TAG}_qap_counters.txt <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1:3]
per = collections.OrderedDict()
for d in ("qap_pmc_mix", "qap_pmc_act"):
    hits = glob.glob(os.path.join(out, "%s_%s" % (tag, d), "**", "*_counter_collection.csv"), recursive=True)
    path = max(hits, key=os.path.getmtime)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "ntt_pass_kernel" in k or "r1cs_" in k:
            name = k.split("(")[0].replace("void frw::", "").replace("frw::", "")
            acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for name, cs in acc.items():
        per.setdefault(name, {}).update({c: sum(v) / len(v) for c, v in cs.items()})
print("# SQ counters per launch (64 Falcon-1024 signatures per call), summed over the chip; tools/pmc_qap.sh")
for name, c in per.items():
    print(name)
    wc = c.get("SQ_WAVE_CYCLES", 0)
    print("   waves %.0f  VALU %.3g  SALU %.3g  LDS %.3g  VMEM rd %.3g wr %.3g" % (c.get("SQ_WAVES", 0), c.get("SQ_INSTS_VALU", 0),
          c.get("SQ_INSTS_SALU", 0), c.get("SQ_INSTS_LDS", 0), c.get("SQ_INSTS_VMEM_RD", 0), c.get("SQ_INSTS_VMEM_WR", 0)))
    if wc:
        print("   wave cycles: issuing %.3f (VALU %.3f), waiting on an instruction %.3f, parked (waitcnt / barrier) %.3f; LDS conflict/active %.3f"
              % (c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_VALU", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc,
                 c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_LDS_BANK_CONFLICT", 0) / max(1.0, c.get("SQ_LDS_IDX_ACTIVE", 0))))
    if c.get("SQ_BUSY_CYCLES") and c.get("SQ_ACTIVE_INST_VALU"):
        print("   SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES = %.3f" % (c["SQ_ACTIVE_INST_VALU"] / c["SQ_BUSY_CYCLES"]))
PY
cat $OUT/${TAG}_qap_counters.txt
