#!/bin/bash
# kernel trace of the default bench -> which of OUR kernels the rocPRIM sorts / runtime fills / copies follow (tools/sort_sites.py)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_sites -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --no-kernel-events > $R/gpurun_out/kt_sites.log 2>&1
python $R/tools/sort_sites.py /tmp/kt_sites/kt_kernel_trace.csv --steps 3 > $R/gpurun_out/sort_sites_a2d2.txt 2>&1
