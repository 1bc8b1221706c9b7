#!/bin/bash
# in-sequence kernel times of several library builds, interleaved, one process each: tools/r3_variants.sh "" build_ab/libnpp_x.so ...
for rep in 1 2; do
  for lib in "$@"; do
    echo "== ${lib:-in-tree} (rep $rep)"
    NPP_LIB_PATH=${lib:+$PWD/$lib} R3_ROWS=${R3_ROWS:-8192,16384,26624} python tools/r3_l3_probe.py 2>/dev/null | grep -v "^rows"
  done
done
