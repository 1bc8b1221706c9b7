#!/bin/bash
# kernel stats + the per-position kernel sequence of 'val' iterations of the bench: tools/r4_final_profile.sh -> gpurun_out/final_prof/
set -e
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/final_prof; rm -rf $out; mkdir -p $out
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-psnr --no-extras --steps 400 > $out/bench.json 2> $out/bench.err
cp $(find $out/raw -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/trace_gap_sites.py $(find $out/raw -name "*kernel_trace.csv" | head -1) > $out/sequence.txt
rm -rf $out/raw
