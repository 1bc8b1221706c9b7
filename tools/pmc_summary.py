#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into profiles/<name>.json.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>

Per kernel (npp:: kernels only): mean counter value per launch, converted to bytes the way
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: both counters are in KiB-units of 1024 B
as emitted by rocprofv3; FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled.
"""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"npp::(\w+(?:<[^>]*>)?)", name)
    return "npp::" + m.group(1) if m else None


def collect(path, counter):
    acc = defaultdict(list)
    # rocprofv3 emits one row per (dispatch, counter[, dimension]); sum the rows of one dispatch
    per = defaultdict(float)
    names = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        if k is None:
            continue
        per[r["Dispatch_Id"]] += float(r["Counter_Value"])
        names[r["Dispatch_Id"]] = k
    for d, v in per.items():
        acc[names[d]].append(v)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def main():
    fetch, nf = collect(sys.argv[1], "FETCH_SIZE")
    write, nw = collect(sys.argv[2], "WRITE_SIZE")
    out = {"note": "mean per launch; bytes = KiB*1024, FETCH_SIZE doubled (gfx950 correction)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, 0.0) * 1024 * 2
        w = write.get(k, 0.0) * 1024
        out["kernels"][k] = {"launches": nf.get(k, nw.get(k, 0)), "fetch_bytes": f, "write_bytes": w,
                             "hbm_bytes": f + w, "FETCH_SIZE_raw_KiB": fetch.get(k, 0.0),
                             "WRITE_SIZE_raw_KiB": write.get(k, 0.0)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out["kernels"].items():
        print(f"{k:44s} fetch {v['fetch_bytes']/1e6:9.2f} MB  write {v['write_bytes']/1e6:9.2f} MB")


if __name__ == "__main__":
    main()
