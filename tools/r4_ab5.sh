#!/bin/bash
# same-box A/B of built library variants (tools/build_variant.sh): tools/r4_ab5.sh default nopp ...   ("default" = the in-tree library)
run() { if [ "$1" = default ]; then e=""; else e="NPP_LIB_PATH=build_ab/libnpp_$1.so"; fi
  env $e timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; q=r['all_kernels_us_in_sequence']; print('$1'.ljust(10), round(d['ms_per_step'],4), 'frac', round(r['frac'],4), 'in-iteration', r.get('all_kernels_us_in_iteration'), 'mlp-only wgrad', q['mlp_wgrad'])"; }
for rep in 1 2 3; do for t in "$@"; do run $t; done; done
