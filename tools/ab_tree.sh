#!/bin/bash
# same-box A/B of the working tree against scratch/prev_tree (latte_amd/ + bench.py of an earlier commit with a library
# of its own): for changes on the Python side.  Prepare in the build container:
#   rm -rf scratch/prev_tree && mkdir -p scratch/prev_tree && git archive <commit> latte_amd bench.py | tar -x -C scratch/prev_tree
#   cp <library of that commit> scratch/prev_tree/latte_amd/liblatte_hip.so
cd $GRAFT_REPO_ROOT
for i in $(seq 1 ${1:-3}); do
  (cd scratch/prev_tree && timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('prev', round(d['ms_per_step'],2), d['roofline']['achieved'], d.get('host_issue_ms_per_step'))") || exit 1
  timeout -k 10 200 python bench.py --no-cpu-baseline --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('tree', round(d['ms_per_step'],2), d['roofline']['achieved'], d.get('host_issue_ms_per_step'))" || exit 1
done
