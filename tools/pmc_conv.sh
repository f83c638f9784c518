# SQ counters of the conv kernel in the micro-benchmark: bash tools/pmc_conv.sh <channels> <variant> <stride>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmc1 /tmp/pmc2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/conv_bench.py --scans 6 --channels ${1:-256} --reps 3 --order plan --variant ${2:-0} --stride ${3:-1} > $R/gpurun_out/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc2 -o p -- python $R/tools/conv_bench.py --scans 6 --channels ${1:-256} --reps 3 --order plan --variant ${2:-0} --stride ${3:-1} > $R/gpurun_out/pmc2.log 2>&1
python - <<'PY'
import csv, collections, glob
for d in ('/tmp/pmc1', '/tmp/pmc2'):
    f = glob.glob(d + '/*counter_collection.csv')[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    first = None
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'][:60]
        if first is None: first = r['Counter_Name']
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == first: cnt[k] += 1
    for k, dd in agg.items():
        if 'spconv_rf' in k or 'spconv_plan' in k:
            print(k, 'dispatches', cnt[k])
            for c, v in sorted(dd.items()): print('   ', c, v / max(cnt[k], 1))
PY
