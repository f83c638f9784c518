#!/bin/bash
# same-box A/B: the network's dense layers (K = 1, identity plan) on the LDS-staged-weights kernel vs the split-j kernel
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do
  for e in 0 1; do
    LT_CONV_LS_DENSE=$e timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ls_dense=$e', round(d['ms_per_step'],2), 'conv TF', r['achieved'], r['launches'], 'dense', r.get('dense_layers_through_the_conv_kernel',{}).get('avg_launch_us'), r.get('dense_layers_through_the_conv_kernel',{}).get('tflops'))" || exit 1
  done
done
