#!/bin/bash
# diagnostics: split-j conv with cache-resident weights (4xxxx), rows (5xxxx) or both (6xxxx) at the a2d2_kitti_128k sizes
cd $GRAFT_REPO_ROOT
O=gpurun_out/cb_resident; mkdir -p $O; rm -f $O/*.log
A="--scans 6 --points 131072 --beams 64 --front --order plan --reps 10"
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 1 --channels 96 --variants 2232,42232,52232,62232 > $O/s1_96.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 2 --channels 96 --variants 2232,42232,52232,62232 > $O/s2_96.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 4 --channels 128 --variants 2142,42142,52142,62142 > $O/s4_128.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 8 --channels 256 --variants 2144,42144,52144,62144 > $O/s8_256.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 1 --channels 32 --variants 2222,42222,52222,62222 > $O/s1_32.log 2>&1
grep -h '^{' $O/*.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['stride'], r['C'], r['V'], r['variant'], r['us'], r['tflops'])"
