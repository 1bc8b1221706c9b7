#!/bin/bash
# kernel stats of the stacked loop: tools/r5_stack_prof.sh <M> -> gpurun_out/stack_prof_M<M>/kernel_stats.csv
M=${1:-8}
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/stack_prof_M$M; rm -rf $out; mkdir -p $out
timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/tools/r5_stack_prof.py $M 200 > $out/log.txt 2>&1
cp $(find $out/raw -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
rm -rf $out/raw
python3 - $out/kernel_stats.csv <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:32]:
    print(f"{r['Name'][:80]:80s} {int(r['Calls']):6d} x {float(r['AverageNs'])/1e3:8.2f} us  {100*float(r['TotalDurationNs'])/tot:5.1f} %")
PY
tail -2 $out/log.txt
