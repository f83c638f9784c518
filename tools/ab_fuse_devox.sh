#!/bin/bash
# same-box A/B: devoxelize folded into the point branch's apply kernel (LT_NET_FUSE_DEVOX=0 switches it off)
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for e in 0 1; do
    LT_NET_FUSE_DEVOX=$e timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fuse_devox=$e', round(d['ms_per_step'],2), 'conv TF', d['roofline']['achieved'])" || exit 1
  done
done
