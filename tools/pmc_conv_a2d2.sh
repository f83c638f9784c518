# SQ counters of the default conv kernel at the a2d2_kitti_128k sizes (6 x 131 072-point front scans):
#   bash tools/pmc_conv_a2d2.sh <channels> <stride> <tag> [variant]     -> gpurun_out/pmc_<tag>_c<channels>_s<stride>.txt
# Two passes (the SQ block takes 8 counters per pass).  Per-dispatch averages of the conv kernels.
set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
C=${1:-96}; S=${2:-1}; TAG=${3:-r03}; V=${4:-0}
OUT=$R/gpurun_out/pmc_${TAG}_c${C}_s${S}.txt
rm -rf /tmp/pmc1 /tmp/pmc2
ARGS="--scans 6 --points 131072 --front --beams 64 --channels $C --reps 3 --order plan --variant $V --stride $S"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS SQ_INSTS_MFMA --output-format csv -d /tmp/pmc1 -o p -- python $R/tools/conv_bench.py $ARGS > $R/gpurun_out/pmc1.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc2 -o p -- python $R/tools/conv_bench.py $ARGS > $R/gpurun_out/pmc2.log 2>&1
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
tail -3 $R/gpurun_out/pmc1.log >> $OUT
cat $OUT
