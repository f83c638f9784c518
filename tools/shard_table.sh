cd $GRAFT_REPO_ROOT
for n in 8 4 2; do
  timeout -k 10 300 python bench.py --workload multiseq_64 --emulate-shard $n --no-cpu-baseline --no-hbm-classes --no-bf16x3-line --no-from-raw --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('shard $n', round(d['ms_per_step'],2), 'host busy', d.get('host_busy_ms_per_step'), 'issue', d.get('host_issue_ms_per_step'), 'conv TF', d['roofline']['achieved'], d['config'].get('voxels_per_gpu'))" || exit 1
done
