#!/bin/bash
# Collect the per-round evidence on the GPU box:  tools/profile_round.sh r01
# (kernel-trace stats and the two PMC passes are separate rocprofv3 runs, as the pool requires)
set -u
R=${1:-r01}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --no-psnr --no-extras --steps 60 --warmup 10"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $B > $OUT/stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o k -- python3 $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o k -- python3 $B > $OUT/write.log 2>&1
cd $ROOT
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
cp $S $OUT/${R}_bench_c2_kernel_stats.csv
python3 tools/pmc_summary.py $F $W $OUT/${R}_pmc_hbm_summary.json
head -12 $OUT/${R}_bench_c2_kernel_stats.csv | cut -c1-150
