#!/bin/bash
# same-box A/B of the one-call backbone pass (lt_net_forward / lt_net_backward) against the module-by-module path:
# bash tools/ab_netexec.sh [rounds] [workload]
cd $GRAFT_REPO_ROOT
W=${2:+--workload $2}
for i in $(seq 1 ${1:-2}); do
  for e in 0 1; do
    LATTE_NET_EXEC=$e timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 $W 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('net_exec=$e', round(d['ms_per_step'],2), 'host busy', d.get('host_busy_ms_per_step'), 'conv TF', d['roofline']['achieved'], 'launches', d['roofline']['launches'])" || exit 1
  done
done
