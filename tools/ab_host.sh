# host-side A/B: step time and host issue time on the GPU-bound headline and on the launch-bound one-scan workload
for wl in a2d2_kitti_128k synthia_kitti_32k_b1 nuscenes_34k_w5; do for i in 1 2; do timeout -k 10 200 python -u bench.py --workload $wl --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$wl', round(d['ms_per_step'],3), d['host_issue_ms_per_step'], d['roofline']['achieved'])"; done; done
