#!/usr/bin/env python3
"""Reads a rocprofv3 --kernel-trace CSV and checks, per HSA queue (and per HIP stream when the column exists), that kernels do not
overlap in time: start[i+1] >= end[i] for consecutive dispatches.  Prints the overlaps (if any) with the kernel names.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/qo -- python3 tools/r6_side_stream_repro.py 2
    python tools/r6_queue_order_check.py gpurun_out/qo
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main(root):
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", root)
        return 1
    for path in files:
        rows = list(csv.DictReader(open(path)))
        print(path, len(rows), "dispatches; columns:", ", ".join(rows[0].keys()))
        for key in ("Queue_Id", "Stream_Id"):
            if key not in rows[0]:
                continue
            groups = defaultdict(list)
            for r in rows:
                groups[r[key]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], int(r.get("Dispatch_Id", 0))))
            for q, ks in sorted(groups.items()):
                ks.sort(key=lambda t: t[3] if t[3] else t[0])          # dispatch order
                bad = [(a, b) for a, b in zip(ks, ks[1:]) if b[0] < a[1]]
                print(f"  {key} {q}: {len(ks)} kernels, {len(bad)} overlapping consecutive pairs")
                for a, b in bad[:8]:
                    print(f"     {a[2]} [{a[0]}..{a[1]}]  then  {b[2]} starts {a[1] - b[0]} ns before it ended")
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/qo"))
