"""Timeline of ONE 'same' iteration from a rocprofv3 kernel trace of tools/r4_same_probe.py: every launch with its queue, start offset and
duration (the contextual chain and the LPIPS branch run on two streams).  usage: r4_same_timeline.py <kernel_trace.csv> [iteration from the end]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
back = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ad = [i for i, r in enumerate(rows) if "adam_pack_kernel" in r["Kernel_Name"]]
a, b = ad[-back - 1], ad[-back]
t0 = int(rows[a]["End_Timestamp"])
qs = {}
for r in rows[a + 1:b + 1]:
    q = qs.setdefault(r["Queue_Id"], len(qs))
    n = r["Kernel_Name"].replace("void ", "").replace("npp::", "").split("(")[0][:46]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    print(f"q{q} {'    ' * q}{(s - t0) / 1e3:8.1f} +{(e - s) / 1e3:6.1f}  {n}")
print(f"iteration: {(int(rows[b]['End_Timestamp']) - t0) / 1e3:.1f} us")
