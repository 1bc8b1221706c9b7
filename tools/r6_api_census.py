#!/usr/bin/env python3
"""HIP API calls per loop iteration (rocprofv3 --hip-trace database): r6_api_census.py <results.db> <iterations>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
rows = db.execute("select name, count(*) from regions group by name order by 2 desc").fetchall() if "regions" in tabs else []
for name, cnt in rows[:40]:
    print(f"{cnt / n:9.2f} per iteration  {cnt:8d}  {name}")
try:
    for r in db.execute("select name, count(*), avg(end-start)/1e3 from memory_copies group by name"):
        print("memory copy:", r)
except Exception as e:
    print("memory copies:", e)
