#!/usr/bin/env python3
"""Every dispatch (our kernels AND whatever else reaches the queue: torch fills / copies, blits) between two consecutive
npp::mlp_fwd_kernel launches of the profiled loop, with the idle time in front of each: r6_iteration_all_kernels.py <results.db> [which]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
which = int(sys.argv[2]) if len(sys.argv) > 2 else 300
rows = db.execute("select name, start, end from kernels order by start").fetchall()
fw = [i for i, r in enumerate(rows) if "mlp_fwd_kernel<2" in r[0] or "mlp_fwd_kernel<1" in r[0] or "mlp_fwd_kernel<true" in r[0]]
a, b = fw[which], fw[which + 1]
prev_end = rows[a - 1][2]
tot_gap = 0.0
for r in rows[a - 1:b + 1]:
    gap = (r[1] - prev_end) / 1e3
    tot_gap += max(gap, 0.0)
    print(f"gap {gap:7.2f} us  run {(r[2] - r[1]) / 1e3:7.2f} us  {r[0][:100]}")
    prev_end = max(prev_end, r[2])
print(f"iteration wall {(rows[b][1] - rows[a][1]) / 1e3:.1f} us, idle in front of launches {tot_gap:.1f} us, {b - a} dispatches")
