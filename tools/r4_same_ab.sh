#!/bin/bash
# same-box A/B of env switches on iterations of one patch source: tools/r4_same_ab.sh <source> "ENV=.." "ENV=.." ...
src=$1; shift
for rep in 1 2 3; do for v in "$@"; do echo -n "$v  "; env $v timeout -k 10 200 python tools/r4_same_probe.py $src 400 2>/dev/null | grep "ms per"; done; done
