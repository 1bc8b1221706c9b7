#!/bin/bash
# MLP-only step wall time of several builds, interleaved: tools/r3_ab_wall.sh "" build_ab/libnpp_x.so ...
for rep in 1 2; do
  for lib in "$@"; do
    NPP_LIB_PATH=${lib:+$PWD/$lib} python tools/r3_step_wall.py 2>/dev/null
    NPP_FUSED_REPACK=0 NPP_LIB_PATH=${lib:+$PWD/$lib} python tools/r3_step_wall.py 2>/dev/null
  done
done
