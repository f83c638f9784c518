#!/bin/bash
# same-box A/B of two builds of liblatte_hip.so: bash tools/ab_bench.sh <other .so> [rounds]
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${2:-2}); do
  LATTE_HIP_LIB=$GRAFT_REPO_ROOT/$1 timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('other', round(d['ms_per_step'],2), d['roofline']['achieved'])"
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tree ', round(d['ms_per_step'],2), d['roofline']['achieved'])"
done
