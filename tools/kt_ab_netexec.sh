#!/bin/bash
# rocprofv3 kernel-trace stats of the bench with and without the one-call backbone pass, per-kernel totals side by side
# (on the GPU box): bash tools/kt_ab_netexec.sh
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/kt_ab
for e in 0 1; do
  export LATTE_NET_EXEC=$e
  timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ktab_$e -o kt -- python $R/bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --no-kernel-events > $R/gpurun_out/kt_ab/log_$e.txt 2>&1
  cp /tmp/ktab_$e/kt_kernel_stats.csv $R/gpurun_out/kt_ab/stats_$e.csv
done
python - <<PY
import csv, collections
def load(f):
    d = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = r["Name"].split("(")[0][:60]
        a = d.setdefault(k, [0, 0.0]); a[0] += int(r["Calls"]); a[1] += float(r["TotalDurationNs"]) / 1e6
    return d
a, b = load("$R/gpurun_out/kt_ab/stats_0.csv"), load("$R/gpurun_out/kt_ab/stats_1.csv")
keys = sorted(set(a) | set(b), key=lambda k: -abs(b.get(k, [0, 0])[1] - a.get(k, [0, 0])[1]))
print(f"{'kernel':60s} {'calls0':>7s} {'ms0':>9s} {'calls1':>7s} {'ms1':>9s} {'diff':>8s}")
for k in keys[:40]:
    x, y = a.get(k, [0, 0.0]), b.get(k, [0, 0.0])
    print(f"{k:60s} {x[0]:7d} {x[1]:9.2f} {y[0]:7d} {y[1]:9.2f} {y[1]-x[1]:8.2f}")
print("total", sum(v[1] for v in a.values()), sum(v[1] for v in b.values()))
PY
