#!/bin/bash
# What the side stream's work costs the main stream (diagnostics, not legitimate: structures of the last prefetch reused):
# LATTE_DIAG_KEEP_INDEX_CACHE = 1 (everything), prepare (the backbone's index structures only), geometry (HR geometry only)
cd $GRAFT_REPO_ROOT
for i in 1 2; do
for e in 0 1 prepare geometry; do
  if [ $e = 0 ]; then unset LATTE_DIAG_KEEP_INDEX_CACHE; else export LATTE_DIAG_KEEP_INDEX_CACHE=$e; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('keep_cache=$e', round(d['ms_per_step'],2), 'host busy', d.get('host_busy_ms_per_step'), 'conv TF', d['roofline']['achieved'])" || exit 1
done
done
