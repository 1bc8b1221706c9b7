set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out/r04h; mkdir -p $OUT; export TMPDIR=/tmp
B="$ROOT/bench.py --no-cpu-baseline --no-psnr --no-extras --steps 60 --warmup 10"
cd /tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o k -- python3 $B > $OUT/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o k -- python3 $B > $OUT/write.log 2>&1
cd $ROOT
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -1)
W=$(find $OUT/write -name '*counter_collection.csv' | head -1)
python3 tools/pmc_summary.py $F $W $OUT/r04_pmc_hbm_summary.json > $OUT/hbm_table.txt
rm -rf $OUT/fetch $OUT/write
