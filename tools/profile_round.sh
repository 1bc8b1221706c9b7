#!/bin/bash
# Collect the per-round evidence on the GPU box:  tools/profile_round.sh r02
# (kernel-trace stats, the FETCH/WRITE PMC passes and the SQ (MFMA utilisation) passes are separate rocprofv3 runs, as the
# pool requires: no --pmc together with trace domains)
set -u
R=${1:-r03}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
# the profiled program: the bench's timed loop, or NPP_PROFILE_CMD (e.g. "tools/r5_stack_prof.py 8 60" for the stacked loop)
# (--windows 0: only the pre-drawn-pool loop -- under rocprofv3 the sampling loop is host-bound, its gaps would fill the sequence table)
B="$ROOT/${NPP_PROFILE_CMD:-bench.py --no-cpu-baseline --no-psnr --no-extras --steps 100 --warmup 10 --windows 0}"
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o k -- python3 $B > $OUT/stats.log 2>&1
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o k -- python3 $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o k -- python3 $B > $OUT/write.log 2>&1
echo "hbm pmc done"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq1 -o k -- python3 $B > $OUT/sq1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/sq2 -o k -- python3 $B > $OUT/sq2.log 2>&1
echo "sq pmc done"
cd $ROOT
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
T=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
cp $S $OUT/${R}_bench_c2_kernel_stats.csv
python3 tools/pmc_summary.py $F $W $OUT/${R}_pmc_hbm_summary.json > $OUT/hbm_table.txt
python3 tools/pmc_sq.py $OUT/${R}_pmc_sq_summary.json $(find $OUT/sq1 $OUT/sq2 -name '*counter_collection.csv') > $OUT/sq_table.txt
python3 tools/trace_gap_sites.py $T > $OUT/${R}_iteration_kernel_sequence.txt
head -14 $OUT/${R}_bench_c2_kernel_stats.csv | cut -c1-150
