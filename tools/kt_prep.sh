#!/bin/bash
# stand-alone kernel times of SPVCNN.prepare() (index structures of the backbone) on an idle GPU, per kernel and per run
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_prep -o kt -- python $R/tools/prep_only.py a2d2_kitti_128k 10 > $R/gpurun_out/kt_prep.log 2>&1
python - <<PY
import csv
rows = list(csv.DictReader(open("/tmp/kt_prep/kt_kernel_stats.csv")))
tot = 0.0
print("per prepare() call (10 calls traced, a2d2_kitti_128k):")
for r in rows:
    ms = float(r["TotalDurationNs"]) / 1e6 / 10
    tot += ms
    if ms > 0.01:
        print(f"  {r['Name'].split('(')[0][:70]:70s} {int(r['Calls'])/10:7.1f} launches {1e3*ms:9.1f} us")
print("total kernel time per call", round(tot, 3), "ms")
PY
