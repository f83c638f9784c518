#!/bin/bash
# split-j shape sweep per U-Net level at the a2d2_kitti_128k sizes (6 x 131 072-point front scans)
cd $GRAFT_REPO_ROOT
O=gpurun_out/cb_a2d2; mkdir -p $O; rm -f $O/*.log
A="--scans 6 --points 131072 --beams 64 --front --order plan --reps 10"
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 1 --channels 96 --variants 0,2132,2232,2432,2133,2233,2136,2236 > $O/s1_96.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 2 --channels 96 --variants 0,2132,2232,2432,2133,2233,2136 > $O/s2_96.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 4 --channels 128 --variants 0,2142,2242,2144,2244,2148 > $O/s4_128.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 8 --channels 256 --variants 0,2142,2242,2144,2244,2148,2248 > $O/s8_256.log 2>&1 &&
timeout -k 10 150 python -u tools/conv_bench.py $A --stride 16 --channels 256 --variants 0,2144,2148,2248 > $O/s16_256.log 2>&1
cat $O/*.log | grep -v '^{' | grep -v amdgpu.ids | tail -5
grep -h '^{' $O/*.log | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['stride'], r['C'], r['V'], r['variant'], r['us'], r['tflops'])"
