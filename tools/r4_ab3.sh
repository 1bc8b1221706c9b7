#!/bin/bash
# same-box A/B of env switches of the CURRENT tree: tools/r4_ab3.sh "NPP_X=0" "NPP_X=1" ...
run() { env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 400 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],4), d['patch_loss_kernels_us'])"; }
for rep in 1 2 3; do for v in "$@"; do run "$v"; done; done
