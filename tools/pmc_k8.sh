# SQ counters of the 2^3 transposed conv kernels (tools/k8_bench.py) at one shape: bash tools/pmc_k8.sh <stride,Cin,Cout> <tag>
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
SH=${1:-1,96,96}; TAG=${2:-k8}
OUT=$R/gpurun_out/pmc_${TAG}_${SH//,/_}.txt
rm -rf /tmp/pmc1 /tmp/pmc2
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/k8_bench.py --only up --shape $SH --reps 3 > $R/gpurun_out/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM --output-format csv -d /tmp/pmc2 -o p -- python $R/tools/k8_bench.py --only up --shape $SH --reps 3 > $R/gpurun_out/pmc2.log 2>&1
python - > $OUT <<'PY'
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
        if 'spconv' in k:
            print(k, 'dispatches', cnt[k])
            for c, v in sorted(dd.items()): print('   ', c, v / max(cnt[k], 1))
PY
cat $OUT
