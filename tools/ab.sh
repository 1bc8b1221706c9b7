#!/bin/bash
# A/B several builds of libnpp_hip.so inside ONE gpurun call (box-to-box clocks differ by ~5 %):
#   tools/ab.sh build_ab/libnpp_a.so build_ab/libnpp_b.so ...     ("" = the in-tree library)
for rep in 1 2 3; do
  for lib in "$@"; do
    NPP_LIB_PATH=${lib:+$PWD/$lib} python bench.py --no-cpu-baseline --no-psnr --no-extras --steps 100 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e6,2), round(d['ms_per_step'],4), d['roofline']['all_kernels_us'])"
  done
done
