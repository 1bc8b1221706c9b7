#!/bin/bash
# per-kernel times of the value-only contextual core at the ranking's size for built library variants (build_ab/libnpp_<v>.so):
#   tools/r6_cx_big_variants.sh default nomfma ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/cxv; rm -rf $out; mkdir -p $out
  if [ "$v" = default ]; then unset NPP_LIB_PATH; else export NPP_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/libnpp_$v.so; fi
  timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/r6_cx_big_probe.py > $out/out.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" <<'PY'
import csv, sys
print(sys.argv[1] + ": " + " | ".join("%s %.0f us" % (r["Name"].replace("npp::", "").split("(")[0], float(r["AverageNs"]) / 1e3)
                                      for r in csv.DictReader(open(sys.argv[2])) if "cx_" in r["Name"]))
PY
  rm -rf $out
done
