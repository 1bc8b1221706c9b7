"""Where the idle time of an iteration sits: MEDIAN gap (next kernel's start - this kernel's end) per position in a
'val'/'train' iteration of the loop, from a rocprofv3 kernel trace (the median: one multi-millisecond stall of the profiled process
in one iteration otherwise shows up as tens of microseconds at one site; the mean is printed beside it when it differs).
usage: trace_gap_sites.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ad = [i for i, r in enumerate(rows) if ("adam_kernel" in r["Kernel_Name"] and "long" in r["Kernel_Name"]) or "adam_pack_kernel" in r["Kernel_Name"]]
segs = [rows[a:b + 1] for a, b in zip(ad[:-1], ad[1:])]
segs = [s for s in segs if not any("lpips" in r["Kernel_Name"] for r in s) and any("conv3x3" in r["Kernel_Name"] for r in s)]
segs = segs[len(segs) // 3:]
L = max(set(len(s) for s in segs), key=[len(s) for s in segs].count)       # the modal launch count
segs = [s for s in segs if len(s) == L]
print(f"{len(segs)} iterations of {L - 1} launches")
def short(n):
    n = n.replace("void ", "").replace("npp::", "")
    return n.split("(")[0][:44]
tot_gap = tot_busy = 0.0
for i in range(L - 1):
    gaps = sorted((int(s[i + 1]["Start_Timestamp"]) - int(s[i]["End_Timestamp"])) / 1e3 for s in segs)
    durs = sorted((int(s[i + 1]["End_Timestamp"]) - int(s[i + 1]["Start_Timestamp"])) / 1e3 for s in segs)
    gap, dur, mean = gaps[len(gaps) // 2], durs[len(durs) // 2], sum(gaps) / len(gaps)
    tot_gap += gap; tot_busy += dur
    note = f"   (mean gap {mean:.1f} us, max {gaps[-1]:.0f})" if abs(mean - gap) > 2.0 else ""
    print(f"{i:3d} {short(segs[0][i]['Kernel_Name']):44s} -> {short(segs[0][i + 1]['Kernel_Name']):44s} gap {gap:6.1f} us   next runs {dur:6.1f} us{note}")
print(f"total gap {tot_gap:.1f} us, busy {tot_busy:.1f} us")
