#!/bin/bash
# SQ counters of every kernel of ONE proof for the 1,024-statement aggregate (and of its key's setup), kernels serialised by the profiler:
# two rocprofv3 --pmc passes over tools/time_aggregate_large.py 1024 1, counters only (no tracing domains).  On the GPU box, from anywhere:
#   bash tools/dev/pmc_sort.sh     -> gpurun_out/pmc_sort_mix/, gpurun_out/pmc_sort_act/ (*_counter_collection.csv)
# What it answered in round 5 (profiles/r05_witness_sorts.txt): nmsm_hist_bare_kernel's waves waited 95 % of their cycles -- for the one
# global counter of the list of ones.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_WAVES --output-format csv -d $R/gpurun_out/pmc_sort_mix -- python3 $R/tools/time_aggregate_large.py 1024 1 > /dev/null 2> $R/gpurun_out/pmc_sort_mix.log
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/gpurun_out/pmc_sort_act -- python3 $R/tools/time_aggregate_large.py 1024 1 > /dev/null 2> $R/gpurun_out/pmc_sort_act.log
