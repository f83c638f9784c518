#!/bin/bash
# same-box A/B of the side stream's priority (diagnostic LATTE_DIAG_SIDE_PRIORITY: -1 high = default, 0 normal, 1 low)
cd $GRAFT_REPO_ROOT
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else '?')"
for p in -1 0 1 -1 0 1; do
  LATTE_DIAG_SIDE_PRIORITY=$p timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('side priority $p', round(d['ms_per_step'],2), 'host busy', d.get('host_busy_ms_per_step'), 'conv TF', d['roofline']['achieved'])" || exit 1
done
