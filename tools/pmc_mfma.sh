#!/bin/bash
# usage (GPU box): tools/pmc_mfma.sh <tag> <python script> [args]   — MFMA counters per kernel (own --pmc pass)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/$tag -- python3 "$@" > gpurun_out/$tag/run.log 2>&1
echo rc=$?
python3 - <<PY
import csv, glob, collections, statistics as st
f = glob.glob("gpurun_out/$tag/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:34]
    acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
    acc[name]["dur_ns"].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
print(f"{'kernel':36s} {'calls':>6s} {'us':>8s} {'MFMA GF/launch':>15s} {'TF/s':>7s} {'mfma_busy/cu_busy':>18s}")
for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]['dur_ns'])):
    mops = st.median(v.get("SQ_INSTS_VALU_MFMA_MOPS_F64", [0]))
    if mops == 0: continue
    dur = st.median(v["dur_ns"]) / len(set(v.keys()) - {"dur_ns"}) * (len(set(v.keys()) - {"dur_ns"}))
    n = len(v["SQ_INSTS_VALU_MFMA_MOPS_F64"])
    dur = st.median(v["dur_ns"])
    busy = st.median(v.get("SQ_VALU_MFMA_BUSY_CYCLES", [0])); cu = st.median(v.get("SQ_BUSY_CU_CYCLES", [1]))
    print(f"{k:36s} {n:6d} {dur/1e3:8.2f} {mops*512/1e9:15.4f} {mops*512/dur/1e3:7.2f} {busy/max(cu,1):18.3f}")
PY
