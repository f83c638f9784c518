#!/bin/bash
# kernel trace of the default bench -> tools/side_overlap.py: which side-stream kernels the main stream's excess time sits under
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/kt_ov -o kt -- python $R/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --no-kernel-events > $R/gpurun_out/kt_overlap.log 2>&1
python $R/tools/side_overlap.py /tmp/kt_ov/kt_kernel_trace.csv --steps 6 > $R/gpurun_out/side_overlap_a2d2.txt 2>&1
