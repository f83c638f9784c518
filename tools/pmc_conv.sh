set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/conv_bench.py --scans 6 --channels ${1:-256} --reps 3 --order plan --variant ${2:-0} > $R/gpurun_out/pmc1.log 2>&1
ls /tmp/pmc1
python - <<'PY'
import csv, collections, glob
f = glob.glob('/tmp/pmc1/*counter_collection.csv')[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'][:60]
    agg[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVE_CYCLES': cnt[k] += 1
for k, d in agg.items():
    if 'spconv_rf' in k or 'spconv_plan' in k:
        print(k, 'dispatches', cnt[k])
        for c, v in sorted(d.items()): print('   ', c, v / max(cnt[k], 1))
PY
