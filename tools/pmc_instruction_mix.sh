#!/bin/bash
# Instruction mix and LDS behaviour of the witness kernel from SQ counters (one rocprofv3 --pmc pass each; counters only,
# no tracing): bash tools/pmc_instruction_mix.sh r02
set -e
TAG=${1:-r02}
REPO=$(cd "$(dirname "$0")/.." && pwd)
OUT=$REPO/gpurun_out
export TMPDIR=/tmp
cd /tmp
ARGS="--steps 1 --warmup 0 --batch 32768 --chunk 32768 --no-cpu-baseline --no-r1cs-check --no-secondary"
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $OUT/${TAG}_pmc_mix -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_mix.log
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/${TAG}_pmc_lds -- python3 $REPO/bench.py $ARGS > /dev/null 2> $OUT/${TAG}_pmc_lds.log
cd $REPO
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, os, sys, collections
out, tag = sys.argv[1:3]
res = collections.OrderedDict()
for d in ("pmc_mix", "pmc_lds"):
    hits = glob.glob(os.path.join(out, "%s_%s" % (tag, d), "**", "*_counter_collection.csv"), recursive=True)
    path = max(hits, key=os.path.getmtime)
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if "witness_ntt_verify_kernel<10, 1>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        res[k] = sum(v) / len(v)
sigs = 32768
stores = res.get("SQ_INSTS_VMEM_WR", 0)
print("# witness_ntt_verify_kernel<10,1>, one launch of %d signatures; SQ counters summed over the chip, per launch" % sigs)
for k, v in res.items():
    print("%-24s %18.0f" % (k, v))
if stores:
    print("wave-level store instructions per signature      %10.1f   (a witness is 4,898 KiB -> ~4,900 full 1 KiB stores)" % (stores / sigs))
    print("VALU instructions per store instruction          %10.2f   (whole kernel: NTTs, ladders, encodes and the tile writer)" % (res["SQ_INSTS_VALU"] / stores))
    print("SALU instructions per store instruction          %10.2f" % (res["SQ_INSTS_SALU"] / stores))
    print("LDS  instructions per store instruction          %10.2f" % (res["SQ_INSTS_LDS"] / stores))
if res.get("SQ_LDS_IDX_ACTIVE"):
    print("LDS bank-conflict cycles / LDS active cycles     %10.4f" % (res["SQ_LDS_BANK_CONFLICT"] / res["SQ_LDS_IDX_ACTIVE"]))
if res.get("SQ_WAVE_CYCLES"):
    print("wave cycles: issuing %.3f, waiting on an instruction %.3f, parked (waitcnt / barrier) %.3f"
          % (res["SQ_ACTIVE_INST_ANY"] / res["SQ_WAVE_CYCLES"], res["SQ_WAIT_INST_ANY"] / res["SQ_WAVE_CYCLES"], res["SQ_WAIT_ANY"] / res["SQ_WAVE_CYCLES"]))
PY
