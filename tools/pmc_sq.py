#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc SQ_* passes (any number of counter_collection.csv files) into one JSON: per npp:: kernel the
mean of every counter per launch (summed over the counter's dimensions), mean duration, and the derived MFMA figures.

usage: pmc_sq.py <out.json> <counter_collection.csv> [...]

Units (MI355X_MICROARCH.md, 'Per-instruction cycle constants'): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles summed over waves; SQ_VALU_MFMA_BUSY_CYCLES counts cycles summed over SIMDs' matrix pipes;
SQ_BUSY_CYCLES counts cycles summed over shader engines (x 32 SEs... reported per dimension, summed here)."""
import csv, json, re, sys
from collections import defaultdict


def short(name):
    m = re.search(r"npp::(\w+(?:<[^>]*>)?)", name)
    return "npp::" + m.group(1) if m else None


def main():
    out_path, files = sys.argv[1], sys.argv[2:]
    per = defaultdict(lambda: defaultdict(float))      # (file, dispatch) -> counter -> value
    meta = {}
    for fi, path in enumerate(files):
        for r in csv.DictReader(open(path)):
            k = short(r["Kernel_Name"])
            if k is None:
                continue
            key = (fi, r["Dispatch_Id"])
            per[key][r["Counter_Name"]] += float(r["Counter_Value"])
            meta[key] = (k, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3, int(r["Grid_Size"]), int(r["Workgroup_Size"]))
    agg = defaultdict(lambda: defaultdict(list))
    dur = defaultdict(list)
    for key, cs in per.items():
        k, us, _, _ = meta[key]
        dur[k].append(us)
        for c, v in cs.items():
            agg[k][c].append(v)
    out = {"note": "mean per launch over all launches of the profiled run; profiled launches run slower than un-profiled ones",
           "kernels": {}}
    for k in sorted(agg):
        d = {c: sum(v) / len(v) for c, v in agg[k].items()}
        d["launches"] = len(dur[k]) // max(1, len(files))
        d["mean_us_profiled"] = sum(dur[k]) / len(dur[k])
        if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d and d["GRBM_GUI_ACTIVE"] > 0:
            # GRBM_GUI_ACTIVE is summed over the 8 XCDs; 1024 SIMDs (256 CUs x 4) each own one matrix pipe
            cyc = d["GRBM_GUI_ACTIVE"] / 8.0
            d["mfma_pipe_busy_frac"] = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
            # GRBM_GUI_ACTIVE / 8 / wall reads high on short dispatches (MI355X_MICROARCH.md 'DVFS give-back': within 3 % of the
            # in-kernel clock only from ~10 ms, high below ~0.3 ms): reported only where it means something
            if d["mean_us_profiled"] >= 300.0:
                d["effective_clock_GHz"] = cyc / (d["mean_us_profiled"] * 1e3)
        if "SQ_WAVE_CYCLES" in d and d["SQ_WAVE_CYCLES"] > 0:
            for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_VALU",
                      "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM"):
                if c in d:
                    d[c + "_frac_of_wave_cycles"] = d[c] / d["SQ_WAVE_CYCLES"]
        out["kernels"][k] = d
    json.dump(out, open(out_path, "w"), indent=1)
    for k, d in out["kernels"].items():
        clk = f"clk {d['effective_clock_GHz']:.2f} GHz" if "effective_clock_GHz" in d else "clk   n/a    "
        print(f"{k:40s} {d['mean_us_profiled']:8.1f} us  mfma_busy {d.get('mfma_pipe_busy_frac', float('nan')):.3f}  "
              f"{clk}  wait_any {d.get('SQ_WAIT_ANY_frac_of_wave_cycles', float('nan')):.2f}  "
              f"wait_inst {d.get('SQ_WAIT_INST_ANY_frac_of_wave_cycles', float('nan')):.2f}  active {d.get('SQ_ACTIVE_INST_ANY_frac_of_wave_cycles', float('nan')):.2f}")


if __name__ == "__main__":
    main()
