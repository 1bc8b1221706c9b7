#!/bin/bash
# per-kernel average durations of the bench's timed loop for several trees, one rocprofv3 kernel-trace run each:
#   tools/r4_kstats.sh <outdir> <tree> [<tree> ...]     (tree = a directory holding bench.py; "." = this repo)
OUT=$1; shift
export TMPDIR=/tmp
ROOT=$(pwd)
for d in "$@"; do
  tag=$(echo $d | tr '/.' '__')
  mkdir -p $ROOT/$OUT/$tag
  (cd /tmp && timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $ROOT/$OUT/$tag -o k -- python3 $ROOT/$d/bench.py --no-cpu-baseline --no-psnr --no-extras --steps 120 --warmup 10 > $ROOT/$OUT/$tag/log.txt 2>&1)
  echo "== $d"; f=$(find $ROOT/$OUT/$tag -name '*kernel_stats.csv' | head -1)
  python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    print(f"{r['Name'][:70]:70s} {int(r['Calls']):6d} {float(r['AverageNs'])/1e3:8.2f} us  tot {float(r['TotalDurationNs'])/1e6:8.2f} ms")
PY
done
