#!/usr/bin/env python3
"""The timeline of the LAST frw_groth16_prove_dev call in a rocprofv3 --kernel-trace CSV: every kernel of at least `min_ms`
milliseconds between the call's groth16_tails_kernel and its groth16_finish_kernel, start -> end in ms from the call's first
kernel, hardware queue, stream.     python3 tools/kernel_timeline.py <..._kernel_trace.csv> [min_ms = 0.3]"""
import csv
import sys


def main():
    path = sys.argv[1]
    min_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 0.3
    rows = sorted(csv.DictReader(open(path)), key=lambda r: int(r["Start_Timestamp"]))
    name = lambda r: r["Kernel_Name"].replace("frw::", "").replace("void ", "")
    tails = [i for i, r in enumerate(rows) if "groth16_tails_kernel" in r["Kernel_Name"]]
    ends = [i for i, r in enumerate(rows) if "groth16_finish_kernel" in r["Kernel_Name"]]
    if not tails or not ends:
        sys.exit("no frw_groth16_prove_dev call in this trace")
    lo, hi = tails[-1], ends[-1]
    t0 = int(rows[lo]["Start_Timestamp"])
    print("last frw_groth16_prove_dev call of %s: %.2f ms from its first kernel to the end of its last; kernels of %.2f ms and more"
          % (path.split("/")[-1], (int(rows[hi]["End_Timestamp"]) - t0) / 1e6, min_ms))
    print("   start ->      end (ms)  queue stream  kernel")
    for r in rows[lo:hi + 1]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if (e - s) / 1e6 >= min_ms:
            print("%8.2f -> %8.2f  q%-3s s%-3s %s" % ((s - t0) / 1e6, (e - t0) / 1e6, r.get("Queue_Id", "?"), r.get("Stream_Id", "?"), name(r)[:90]))


if __name__ == "__main__":
    main()
