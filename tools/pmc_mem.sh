# memory-side counters of the conv kernel in the micro-benchmark: bash tools/pmc_mem.sh <channels> <variant> <stride>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pm1 /tmp/pm2
A="python $R/tools/conv_bench.py --scans 6 --channels ${1:-256} --reps 3 --order plan --variant ${2:-0} --stride ${3:-1}"
timeout -s KILL 120 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm1 -o p -- $A > $R/gpurun_out/pm1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm2 -o p -- $A > $R/gpurun_out/pm2.log 2>&1
python - <<'PY'
import csv, collections, glob
for d in ('/tmp/pm1', '/tmp/pm2'):
    f = glob.glob(d + '/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    first = None
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if first is None: first = r['Counter_Name']
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == first: cnt[k] += 1
    for k, dd in agg.items():
        if 'spconv_rf' in k:
            print(k, 'dispatches', cnt[k])
            for c, v in sorted(dd.items()): print('   ', c, v / max(cnt[k], 1))
PY
