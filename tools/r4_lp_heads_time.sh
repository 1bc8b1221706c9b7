#!/bin/bash
# average duration of the grouped LPIPS heads launch in 'same' iterations, per env setting: tools/r4_lp_heads_time.sh "ENV=.." ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/lp_heads; rm -rf $out; mkdir -p $out
  env $v timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/r4_same_probe.py same 100 > $out/out.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  echo "$v $(grep -E 'lpips_multi|trunk_grad_in' $f | cut -d, -f1-4 | tr '\n' ' ')"
  rm -rf $out
done
