#!/bin/bash
# Build an A/B variant of libnpp_hip.so with extra compile flags:  tools/build_variant.sh <name> "<-DFOO=1 ...>"
# -> build_ab/libnpp_<name>.so (git-ignored, travels with gpurun).  Use with tools/ab.sh / NPP_LIB_PATH.
set -e
NAME=$1; shift
FLAGS="$*"
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PKG="$ROOT/learning-continuous-implicit-representation-for-near-periodic-patterns_amd"
OUT="$ROOT/build_ab/$NAME"
mkdir -p "$OUT"
pids=()
for s in "$PKG"/csrc/*.hip; do
  o="$OUT/$(basename "${s%.hip}").o"
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -I "$ROOT/include" -I "$PKG/csrc" -Wall -Wno-unused-function \
      -ffp-contract=off $FLAGS -c "$s" -o "$o" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/build_ab/libnpp_$NAME.so" "$OUT"/*.o
echo "built build_ab/libnpp_$NAME.so"
