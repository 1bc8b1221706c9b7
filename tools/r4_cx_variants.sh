#!/bin/bash
# per-kernel times of the contextual core alone (tools/cx_probe.py) for built library variants: tools/r4_cx_variants.sh default cx_nomfma ...
cd /tmp && export TMPDIR=/tmp
for v in "$@"; do
  if [ "$v" = default ]; then e="NPP_X=0"; else e="NPP_LIB_PATH=$GRAFT_REPO_ROOT/build_ab/libnpp_$v.so"; fi
  out=$GRAFT_REPO_ROOT/gpurun_out/cxv; rm -rf $out; mkdir -p $out
  env $e timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/tools/cx_probe.py > $out/out.log 2>&1
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 - "$v" "$f" <<'PY'
import csv, sys
print(sys.argv[1] + ": " + " | ".join("%s %.1f" % (r["Name"].replace("npp::", "").split("(")[0], float(r["AverageNs"]) / 1e3)
                                      for r in csv.DictReader(open(sys.argv[2])) if "cx_" in r["Name"]))
PY
  rm -rf $out
done
