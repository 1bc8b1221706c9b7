#!/bin/bash
# same-box A/B of process environments over the device-only loop: tools/r6_env_ab.sh "A=1" "B=2" ...   (use "X=" for the default)
run() { env $1 timeout -k 10 200 python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 400 --windows 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$1', round(d['config']['device_only_ms_per_step'],4))"; }
for rep in 1 2 3; do for v in "$@"; do run "$v"; done; done
