#!/usr/bin/env python3
"""Mean per-launch value of every counter in rocprofv3 counter_collection CSVs, for npp:: kernels.
usage: pmc_table.py <csv> [<csv> ...]"""
import csv, re, sys
from collections import defaultdict
tab = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    per = defaultdict(float); meta = {}
    for r in csv.DictReader(open(path)):
        m = re.search(r"npp::(\w+(?:<[^>]*>)?)", r["Kernel_Name"])
        if not m: continue
        key = (r["Dispatch_Id"], r["Counter_Name"])
        per[key] += float(r["Counter_Value"]); meta[key] = m.group(1)
    for (d, c), v in per.items():
        tab[meta[(d, c)]][c].append(v)
for k in sorted(tab):
    print(k)
    for c in sorted(tab[k]):
        v = tab[k][c]
        print(f"   {c:36s} {sum(v)/len(v):16.1f}   (n={len(v)})")
