#!/bin/bash
# usage: tools/r4_kstats2.sh <outdir> <python script> [args...]  -- rocprofv3 kernel stats of a python script, top kernels printed
set -e
out=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 "$@" > $out/out.log 2>&1
grep -v "^[EW]2026" $out/out.log | grep -v Warning | grep -v "ref = _Trunk" | tail -3
python3 - $out <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:6]:
    print(r['Name'][:50].ljust(50), r['Calls'], round(float(r['AverageNs'])/1e3,2))
PY
