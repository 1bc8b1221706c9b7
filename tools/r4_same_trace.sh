#!/bin/bash
# kernel-trace timeline of a 'same' iteration: tools/r4_same_trace.sh  -> gpurun_out/same_timeline.txt
set -e
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/same_trace; rm -rf $out; mkdir -p $out
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/r4_same_probe.py same 60 > $out/out.log 2>&1
f=$(find $out -name "*kernel_trace.csv" | head -1)
python3 $GRAFT_REPO_ROOT/tools/r4_same_timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/same_timeline.txt
rm -rf $out
