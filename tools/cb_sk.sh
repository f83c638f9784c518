#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/cb_sk; mkdir -p $O; rm -f $O/*.log
V="0,1244,21244,1148,21148,1248,21248,1444,21444"
V96="0,1434,21434,1234,21234"
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 16 --channels 128,256 --order plan --variants $V --reps 5 > $O/s16.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 8 --channels 128,256 --order plan --variants $V --reps 5 > $O/s8.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 4 --channels 128 --order plan --variants $V --reps 5 > $O/s4.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 2 --channels 96 --order plan --variants $V96 --reps 5 > $O/s2.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 1 --channels 96 --order plan --variants $V96 --reps 5 > $O/s1.log 2>&1
echo done
