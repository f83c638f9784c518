#!/bin/bash
# same-box A/B: per-class / global quantiles by radix selection (default) against the radix-sort path (LT_QUANTILE_SORT=1)
cd $GRAFT_REPO_ROOT
W=${1:-a2d2_kitti_128k}
for i in 1 2 3; do
  for e in 1 0; do
    LT_QUANTILE_SORT=$e timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sort=$e', round(d['ms_per_step'],2), 'conv TF', d['roofline']['achieved'], 'host', d.get('host_busy_ms_per_step'))" || exit 1
  done
done
