#!/bin/bash
# per-position kernel sequence + kernel stats of the bench loop under an environment setting: tools/r5_seq.sh <tag> [VAR=val ...]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/$tag; rm -rf $out; mkdir -p $out
for kv in "$@"; do export "$kv"; done
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/raw -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-psnr --no-extras --steps 400 > $out/bench.json 2> $out/bench.err
cp $(find $out/raw -name "*kernel_stats.csv" | head -1) $out/kernel_stats.csv
python3 $GRAFT_REPO_ROOT/tools/trace_gap_sites.py $(find $out/raw -name "*kernel_trace.csv" | head -1) > $out/sequence.txt
rm -rf $out/raw
