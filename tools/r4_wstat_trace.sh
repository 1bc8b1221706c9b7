set -e
cd /tmp && export TMPDIR=/tmp
for v in 0 1; do
  out=$GRAFT_REPO_ROOT/gpurun_out/same_trace_ws$v; rm -rf $out; mkdir -p $out
  NPP_CONV_WSTAT=$v timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/r4_same_probe.py same 60 > $out/out.log 2>&1
  f=$(find $out -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/r4_same_timeline.py $f > $GRAFT_REPO_ROOT/gpurun_out/same_timeline_ws$v.txt
  rm -rf $out
done
