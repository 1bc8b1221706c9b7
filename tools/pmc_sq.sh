#!/bin/bash
# SQ counter passes of the bench's kernels on the GPU box:  tools/pmc_sq.sh <tag> [lib.so]
# (counter collection alone: no --kernel-trace / --stats in the same run, as the pool requires)
set -u
R=${1:-sq}
LIB=${2:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/$R
mkdir -p $OUT
export TMPDIR=/tmp
[ -n "$LIB" ] && export NPP_LIB_PATH=$ROOT/$LIB
B="$ROOT/bench.py --no-cpu-baseline --no-psnr --no-extras --steps 30 --warmup 10"
cd /tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $OUT/p1 -o k -- python3 $B > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/p2 -o k -- python3 $B > $OUT/p2.log 2>&1
cd $ROOT
python3 tools/pmc_sq.py $OUT/${R}_pmc_sq_summary.json $(find $OUT/p1 $OUT/p2 -name '*counter_collection.csv')
