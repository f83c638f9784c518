for sw in 5e-3 5e-4 5e-5; do for pf in "" "--no-prefetch"; do for i in 1 2; do LATTE_DIAG_SWITCH_INTERVAL=$sw timeout -k 10 200 python -u bench.py --workload synthia_kitti_32k_b1 $pf --steps 30 --warmup 5 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('sw=$sw pf=$pf', round(d['ms_per_step'],3), d['host_issue_ms_per_step'])"; done; done; done
