#!/bin/bash
# per-kernel timings of several builds, interleaved in one gpurun call: tools/kt.sh lib1.so lib2.so ...
for rep in 1 2; do
  for lib in "$@"; do
    NPP_LIB_PATH=${lib:+$PWD/$lib} python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 40 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['ms_per_step'],4), round(d['mlp_only_step']['ms_per_step'],4), d['roofline']['all_kernels_us'])"
  done
done
