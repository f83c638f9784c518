#!/bin/bash
# per-queue kernel time of the emulated 8-way shard of multiseq_64 (rocprofv3 kernel trace -> tools/stream_busy.py)
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_s8 -o kt -- python $R/bench.py --workload multiseq_64 --emulate-shard 8 --steps 5 --warmup 2 --no-cpu-baseline --no-hbm-classes --no-bf16x3-line --no-from-raw --no-kernel-events > $R/gpurun_out/kt_s8.log 2>&1
python $R/tools/stream_busy.py /tmp/kt_s8/kt_kernel_trace.csv --steps 3 > $R/gpurun_out/stream_busy_shard8.txt 2>&1
