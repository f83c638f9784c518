#!/bin/bash
# rocprofv3 evidence for the round: kernel-trace stats of the bench command, then two PMC passes (HBM read / write bytes).
# usage (on the GPU box): bash tools/profile_round.sh r01
set -e
TAG=${1:-r02}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
mkdir -p $R/gpurun_out/prof_$TAG
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -o kt -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw > $R/gpurun_out/prof_$TAG/kt.log 2>&1
cp /tmp/kt_$TAG/kt_kernel_stats.csv $R/gpurun_out/prof_$TAG/${TAG}_kernel_stats.csv
python $R/tools/stream_busy.py /tmp/kt_$TAG/kt_kernel_trace.csv --steps 3 > $R/gpurun_out/prof_$TAG/${TAG}_stream_busy_a2d2.txt 2>&1 || true
grep -v "^[WEI]2026" $R/gpurun_out/prof_$TAG/kt.log | tail -1 > $R/gpurun_out/prof_$TAG/${TAG}_bench_under_rocprof.json
# the conv family = kernels named k_spconv_*; the network's dense layers (K = 1 over an identity plan) run the same code
# under the name k_dense_sj and are not counted
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pf_$TAG -o pf -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --no-kernel-events > $R/gpurun_out/prof_$TAG/pf.log 2>&1
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/pw_$TAG -o pw -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --no-kernel-events > $R/gpurun_out/prof_$TAG/pw.log 2>&1
python - <<PY
import csv, collections, json
out = {}
for tag, f in (("FETCH_SIZE", "/tmp/pf_$TAG/pf_counter_collection.csv"), ("WRITE_SIZE", "/tmp/pw_$TAG/pw_counter_collection.csv")):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != tag:
            continue
        k = r["Kernel_Name"].split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    out[tag] = {k: {"dispatches": v[0], "sum": v[1], "per_dispatch": v[1] / v[0]} for k, v in agg.items()}
json.dump(out, open("$R/gpurun_out/prof_$TAG/${TAG}_pmc_hbm.json", "w"), indent=1)
conv = [k for k in out["FETCH_SIZE"] if "spconv" in k]
n = sum(out["FETCH_SIZE"][k]["dispatches"] for k in conv)
f = sum(out["FETCH_SIZE"][k]["sum"] for k in conv)
w = sum(out["WRITE_SIZE"].get(k, {"sum": 0})["sum"] for k in conv)
print("conv dispatches", n, "FETCH_SIZE/dispatch", f / n, "WRITE_SIZE/dispatch", w / n)
json.dump({"workload": "a2d2_kitti_128k", "hbm_bytes_per_launch": round((2 * f / n + w / n) * 1024), "fetch_size_kb_per_launch_raw": f / n,
           "write_size_kb_per_launch": w / n,
           "correction": "FETCH_SIZE doubled (gfx950 reports half the bytes of 16-B/lane reads, MI355X_MICROARCH.md #HBM); "
                         "WRITE_SIZE as reported; KB -> bytes x1024",
           "dispatches": n,
           "source": "profiles/$TAG" + "_pmc_hbm.json (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes, "
                     "bench.py --steps 2 --warmup 1)"},
          open("$R/gpurun_out/prof_$TAG/${TAG}_conv_traffic.json", "w"), indent=1)
PY
# last, and allowed to fail: MFMA pipe utilisation of the conv kernels
timeout -s KILL 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d /tmp/pm_$TAG -o pm -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --no-hbm-classes --no-bf16x3-line --no-from-raw --no-kernel-events > $R/gpurun_out/prof_$TAG/pm.log 2>&1 || { echo "MFMA-busy pass failed"; exit 0; }
python - <<PY
import csv, collections, json
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for r in csv.DictReader(open("/tmp/pm_$TAG/pm_counter_collection.csv")):
    k = r["Kernel_Name"].split("(")[0]
    if "spconv" not in k:
        continue
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
out = {"note": "MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs): the busy counter counts "
               "cycles summed over the SIMDs (32 per v_mfma_f32_16x16x4_f32), GRBM_GUI_ACTIVE is summed over the 8 XCDs "
               "(MI355X_MICROARCH.md, cycle-constants table); zero-padded rows of a strip are busy cycles too", "kernels": {}}
tb = tg = 0.0
for k, d in agg.items():
    b, g = d["SQ_VALU_MFMA_BUSY_CYCLES"], d["GRBM_GUI_ACTIVE"]
    tb += b; tg += g
    out["kernels"][k] = {"dispatches": cnt[k], "mfma_busy_cycles": b, "insts_mfma": d["SQ_INSTS_MFMA"], "gui_active": g,
                         "mfma_utilisation": round(b / (g / 8 * 1024), 4) if g else None}
out["all_conv_kernels_mfma_utilisation"] = round(tb / (tg / 8 * 1024), 4) if tg else None
json.dump(out, open("$R/gpurun_out/prof_$TAG/${TAG}_mfma_busy.json", "w"), indent=1)
print("MFMA utilisation (conv kernels):", out["all_conv_kernels_mfma_utilisation"])
PY
