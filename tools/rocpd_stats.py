#!/usr/bin/env python3
"""Per-kernel totals from a rocprofv3 rocpd database (the default output format): rocpd_stats.py <results.db> [top]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
top = int(sys.argv[2]) if len(sys.argv) > 2 else 25
cur = db.cursor()
rows = cur.execute("select name, count(*), avg(duration)/1e3, sum(duration)/1e3, grid_x, grid_y, grid_z, workgroup_x from kernels "
                   "group by name, grid_x, grid_y, grid_z order by 4 desc").fetchall()
tot = sum(r[3] for r in rows)
for r in rows[:top]:
    print(f"{r[1]:6d} x {r[2]:8.1f} us = {r[3] / 1e3:8.2f} ms {100 * r[3] / tot:5.1f}%  wgs=({r[4] // max(1, r[7])},{r[5]},{r[6]})  {r[0][:90]}")
print(f"total {tot / 1e3:.2f} ms in {sum(r[1] for r in rows)} launches")
