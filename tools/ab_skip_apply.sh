# upper bound of "normalise on load in the consumer conv": the a2d2 step with the BatchNorm apply pass of every residual
# block's first conv simply left out (wrong results), against the normal step, same box
for i in 1 2; do for m in "" 1; do LATTE_DIAG_SKIP_FIRST_APPLY=$m timeout -k 10 200 python -u bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw 2>gpurun_out/ab_skip.err | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('skip_first_apply=$m', round(d['ms_per_step'],3), d['host_issue_ms_per_step'], d['roofline']['achieved'])"; done; done
