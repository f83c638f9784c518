#!/bin/bash
# conv at the a2d2 sizes with the level's rows stored in hash order (as is) / Morton / lexicographic order: time and
# HBM-side fetch bytes (rocprofv3 --pmc FETCH_SIZE, doubled per the guide)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
A="--scans 6 --points 131072 --beams 64 --front --order plan --reps 10"
for cfg in "1 96" "2 96" "8 256"; do
  set -- $cfg
  for st in asis morton lex; do
    timeout -k 10 150 python -u $R/tools/conv_bench.py $A --stride $1 --channels $2 --storage $st 2>/dev/null | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print('$st', r['stride'], r['C'], r['V'], r['us'], r['tflops'])" || exit 1
    rm -rf /tmp/pfs; timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pfs -o pf -- python $R/tools/conv_bench.py $A --stride $1 --channels $2 --storage $st --reps 2 > /dev/null 2>&1
    python - <<PY
import csv
tot = n = 0
for r in csv.DictReader(open("/tmp/pfs/pf_counter_collection.csv")):
    if r["Counter_Name"] == "FETCH_SIZE" and "spconv" in r["Kernel_Name"]:
        tot += float(r["Counter_Value"]); n += 1
print("   $st fetch MB per conv launch", round(2 * tot / max(n, 1) * 1024 / 1e6, 1), "launches", n)
PY
  done
done
