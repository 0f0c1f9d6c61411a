#!/usr/bin/env python3
"""Condense rocprofv3 output directories (gpurun_out/prof_*) into the tracked summaries under profiles/.

    python tools/summarize_profiles.py r01 gpurun_out/prof_trace gpurun_out/prof_pmc_w gpurun_out/prof_pmc_f

Writes profiles/<tag>_kernel_stats.csv (the --kernel-trace --stats table, verbatim) and
profiles/<tag>_hbm_traffic.json (per-launch HBM bytes of the witness kernel from the WRITE_SIZE / FETCH_SIZE
passes).  Corrections per /opt/skills/guides/MI355X_MICROARCH.md (HBM section): counters are in KiB; WRITE_SIZE is
exact for 16-B-per-lane streaming stores; FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (x2 applied;
the read side here is 0.1 % of the traffic, so the uncertainty does not matter).
"""
import csv
import glob
import hashlib
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def find(d, suffix):
    hits = glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True)
    return max(hits, key=os.path.getmtime) if hits else None      # gpurun merges, never deletes: newest wins


def pmc_avg(d, counter, kernel_substr):
    rows = [r for r in csv.DictReader(open(find(d, "_counter_collection.csv")))
            if r["Counter_Name"] == counter and kernel_substr in r["Kernel_Name"]]
    vals = [float(r["Counter_Value"]) for r in rows]
    grid = {r["Grid_Size"] for r in rows}
    return sum(vals) / len(vals), len(vals), sorted(grid)


def main():
    tag, trace, pmc_w, pmc_f = sys.argv[1:5]
    chunk = int(sys.argv[5]) if len(sys.argv) > 5 else 4096
    logn = int(sys.argv[6]) if len(sys.argv) > 6 else 10
    kname = sys.argv[7] if len(sys.argv) > 7 else "witness_ntt_verify_kernel"
    out = os.path.join(ROOT, "profiles")
    os.makedirs(out, exist_ok=True)
    if "_" not in tag:                         # the per-kernel tags (r02_verify512, ...) share the round's one trace
        shutil.copy(find(trace, "_kernel_stats.csv"), os.path.join(out, tag + "_kernel_stats.csv"))
    kern = "%s<%d, 1" % (kname, logn)          # prefix: the hot kernel carries a third template argument (SPLIT)
    w, nw, _ = pmc_avg(pmc_w, "WRITE_SIZE", kern)
    f, nf, _ = pmc_avg(pmc_f, "FETCH_SIZE", kern)
    stats = [r for r in csv.DictReader(open(find(trace, "_kernel_stats.csv"))) if kern in r["Name"]][0]
    summary = {
        "kernel": kern + ("" if kern.endswith(">") else ", ...>"), "logn": logn, "signatures_per_launch": chunk,
        # bench.py only quotes this summary while the kernel source is the one that was profiled
        "kernel_source_sha256_16": hashlib.sha256(open(os.path.join(ROOT, "falcon-r1cs_amd", "csrc", "frw_kernels.hip"), "rb").read()).hexdigest()[:16],
        "avg_launch_ns": float(stats["AverageNs"]), "calls": int(stats["Calls"]),
        "WRITE_SIZE_KiB_avg": w, "WRITE_SIZE_launches": nw, "FETCH_SIZE_KiB_avg_raw": f, "FETCH_SIZE_launches": nf,
        "write_bytes_per_launch": w * 1024, "read_bytes_per_launch": f * 1024 * 2,
        "hbm_bytes_per_launch": w * 1024 + f * 1024 * 2,
        "corrections": "KiB -> bytes; FETCH_SIZE x2 (gfx950 wide-read under-count); WRITE_SIZE exact",
    }
    json.dump(summary, open(os.path.join(out, tag + "_hbm_traffic.json"), "w"), indent=1)
    print(json.dumps(summary, indent=1))


if __name__ == "__main__":
    main()
