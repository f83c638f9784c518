for i in 1 2; do for m in 0 1 2; do LT_CONV_LS=$m timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('LS=$m', round(d['ms_per_step'],3), d['host_issue_ms_per_step'], d['roofline']['achieved'])"; done; done
