#!/bin/bash
# same-box A/B of two library builds on the conv micro-benchmark at the a2d2 sizes: bash tools/ab_conv.sh <other .so>
cd $GRAFT_REPO_ROOT
A="--scans 6 --points 131072 --beams 64 --front --order plan --reps 10"
run() {  # label, lib
  for cfg in "1 96 2232" "2 96 2232" "4 128 2142" "8 256 2144" "16 256 2144"; do
    set -- $1 $2 $cfg
    LATTE_HIP_LIB=$2 timeout -k 10 150 python -u tools/conv_bench.py $A --stride $3 --channels $4 --variants $5 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$1', r['stride'], r['C'], r['V'], r['variant'], r['us'], r['tflops'])" || return 1
  done
}
run other $GRAFT_REPO_ROOT/$1 && run tree $GRAFT_REPO_ROOT/latte_amd/liblatte_hip.so && run other $GRAFT_REPO_ROOT/$1 && run tree $GRAFT_REPO_ROOT/latte_amd/liblatte_hip.so
