set -e
cd /tmp && export TMPDIR=/tmp
for v in 1 0; do
  out=$GRAFT_REPO_ROOT/gpurun_out/seq_fwdfold$v; rm -rf $out; mkdir -p $out
  NPP_POOL_FOLD_FWD=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/r4_same_probe.py val 120 > $out/out.log 2>&1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/trace_gap_sites.py $f > $GRAFT_REPO_ROOT/gpurun_out/seq_fwdfold$v.txt
  rm -rf $out
done
