#!/bin/bash
# split-k (0 = current default) vs split-j (2000+) conv kernels per U-Net level
cd $GRAFT_REPO_ROOT
O=gpurun_out/cb_sj; mkdir -p $O; rm -f $O/*.log
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 16 --channels 128,256 --order plan --variants 0,2148,2144,2248,2244,2242,2142 --reps 10 > $O/s16.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 8 --channels 64,128,256 --order plan --variants 0,2244,2242,2144,2142 --reps 10 > $O/s8.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 4 --channels 64,128 --order plan --variants 0,2244,2242,2144,2142 --reps 10 > $O/s4.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 2 --channels 32 --order plan --variants 0,2222,2122 --reps 10 > $O/s2a.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 2 --channels 96 --order plan --variants 0,2233,2232,2132,2133 --reps 10 > $O/s2b.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 1 --channels 32 --order plan --variants 0,2222,2122 --reps 10 > $O/s1a.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 1 --channels 96 --order plan --variants 0,2233,2232,2132,2133 --reps 10 > $O/s1b.log 2>&1
cat $O/*.log | grep -v '^{' | grep -v amdgpu.ids | tail -5
grep -h '^{' $O/*.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['stride'], r['C'], r['variant'], r['us'], r['tflops'])"
