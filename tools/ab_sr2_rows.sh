#!/bin/bash
# same-box A/B of the row count from which conv workgroups take two 16-row strips per wave (default 100 000):
#   bash tools/ab_sr2_rows.sh <workload> <threshold> [<threshold> ...]
cd $GRAFT_REPO_ROOT
W=$1; shift
for i in 1 2; do
  for t in "$@"; do
    LATTE_DIAG_CONV_SR2_ROWS=$t timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$W sr2_rows=$t', round(d['ms_per_step'],2), 'conv TF', d['roofline']['achieved'])" || exit 1
  done
done
