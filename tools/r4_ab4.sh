#!/bin/bash
# same-box A/B of the complete iteration over several trees: tools/r4_ab4.sh <tree> [<tree> ...]   (each a built checkout, e.g. _scratch/r3 .)
run() { (cd $1 && timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']['all_kernels_us_in_sequence']; print('$1'.ljust(14), round(d['ms_per_step'],4), r['mlp_fwd_train'], r['pixel_loss'], r['mlp_bwd_chain'], r['mlp_wgrad'], r['adam+repack'])"); }
for rep in 1 2 3; do
  for t in "$@"; do run $t; done
done
