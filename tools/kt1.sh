#!/bin/bash
for lib in "$@"; do
  NPP_LIB_PATH=${lib:+$PWD/$lib} python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 20 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', d['roofline']['all_kernels_us'])"
done
