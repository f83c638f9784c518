#!/bin/bash
# same-box A/B of two whole trees (library + Python): scratch/prev_tree (the last commit, exported by hand:
#   bash tools/build_prev.sh && git archive HEAD latte_amd bench.py oracle include | tar -x -C scratch/prev_tree
#   && cp scratch/ab/liblatte_prev.so scratch/prev_tree/latte_amd/liblatte_hip.so) against the working tree
cd $GRAFT_REPO_ROOT
W=${1:-a2d2_kitti_128k}
for i in 1 2 3; do
  for t in scratch/prev_tree .; do
    (cd $t && timeout -k 10 200 python bench.py --workload $W --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tree=$t', round(d['ms_per_step'],2), 'conv TF', d['roofline']['achieved'], 'host', d.get('host_busy_ms_per_step'))") || exit 1
  done
done
