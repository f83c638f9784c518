#!/bin/bash
# split-k sweep on the bench workload's levels: bash tools/cb_sk.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/cb_sk; mkdir -p $O; rm -f $O/*.log
V="0,1144,1148,1184,1224,1228,1244,1248,1284,1288,1424,1444"
V96="0,1124,1134,1164,1234,1238,1264,1268,1224,1228,1434,1438,1464"
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 16 --channels 128,256 --order plan --variants $V --reps 5 > $O/s16.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 8 --channels 128,256 --order plan --variants $V --reps 5 > $O/s8.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 4 --channels 64,128 --order plan --variants $V --reps 5 > $O/s4.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 2 --channels 96 --order plan --variants $V96 --reps 5 > $O/s2.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 1 --channels 96 --order plan --variants $V96 --reps 5 > $O/s1.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 2 --channels 32 --order plan --variants 0,1122,1124,1128,1222,1224,1228,1214,1218 --reps 5 > $O/s2b.log 2>&1 &&
timeout -k 10 120 python -u tools/conv_bench.py --scans 6 --stride 1 --channels 32 --order plan --variants 0,1122,1124,1128,1222,1224,1228,1214,1218 --reps 5 > $O/s1b.log 2>&1
echo done
