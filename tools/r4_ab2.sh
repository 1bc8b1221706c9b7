#!/bin/bash
run() { (cd $1 && env $2 timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['all_kernels_us_in_sequence']; print('$3', round(d['ms_per_step'],4), r['mlp_fwd_train'], r['mlp_bwd_chain'], r['mlp_wgrad'], round(d['mlp_only_step']['ms_per_step'],4))"); }
for rep in 1 2 3; do
  run _scratch/r3 "A=1" "r3      "
  run . "A=1" "current "
  run . "NPP_LIB_PATH=$PWD/build_ab/libnpp_r3fwd.so" "r3fwd   "
done
