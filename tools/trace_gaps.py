"""Summarise a rocprofv3 kernel trace: per-iteration kernel-busy time vs wall time for the loop probe.
usage: trace_gaps.py <kernel_trace.csv> [n_last_kernels_window]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# find iterations by the adam kernel (one per iteration)
ad = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"] and "long" in r["Kernel_Name"]]
its = []
for a, b in zip(ad[:-1], ad[1:]):
    seg = rows[a + 1:b + 1]
    t0, t1 = int(rows[a]["End_Timestamp"]), int(seg[-1]["End_Timestamp"])
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
    its.append((t1 - t0, busy, len(seg), any("lpips" in r["Kernel_Name"] for r in seg)))
its = its[len(its) // 2:]
for same in (False, True):
    s = [x for x in its if x[3] == same]
    if s:
        print("same" if same else "val/train", "iters", len(s), "wall us %.1f" % (sum(x[0] for x in s) / len(s) / 1e3),
              "busy us %.1f" % (sum(x[1] for x in s) / len(s) / 1e3), "launches %.1f" % (sum(x[2] for x in s) / len(s)))
# per-kernel share inside val/train iterations
from collections import defaultdict
acc, cnt = defaultdict(float), defaultdict(int)
n = 0
for a, b in list(zip(ad[:-1], ad[1:]))[len(ad) // 2:]:
    seg = rows[a + 1:b + 1]
    if any("lpips" in r["Kernel_Name"] for r in seg):
        continue
    n += 1
    for r in seg:
        k = r["Kernel_Name"][:70]
        acc[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"]); cnt[k] += 1
for k in sorted(acc, key=acc.get, reverse=True)[:25]:
    print(f"{k:70s} {cnt[k]/n:6.1f} x {acc[k]/cnt[k]/1e3:7.1f} us = {acc[k]/n/1e3:7.1f}")
