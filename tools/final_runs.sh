#!/bin/bash
# usage (GPU box): tools/final_runs.sh <round tag>  — everything profiles/ quotes, in one call
tag=$1
cd $GRAFT_REPO_ROOT
bash tools/collect_profiles.sh $tag "builder run, round ${tag#r} (final build)" > gpurun_out/${tag}_collect.log 2>&1
cp gpurun_out/${tag}_pmc_traffic_lips4096_m512.json gpurun_out/${tag}_pmc_traffic_oxide16384_m1024.json profiles/   # (the bench line quotes the traffic of THIS build: same csrc sha)
python bench.py > gpurun_out/${tag}_bench_line.json 2> gpurun_out/${tag}_bench_line.err
python examples/md_nvt_config5.py --steps 300 > gpurun_out/${tag}_md_config5_16384.log 2>&1
python examples/md_nvt_config5.py --steps 1000 > gpurun_out/${tag}_md_config5_16384_1000steps.log 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/${tag}_bench_line_20steps.json 2>/dev/null
(python tools/update_bench.py 32 1024 2; python tools/update_bench.py 32 256 2) > gpurun_out/${tag}_update_bench_16384.log 2>&1
for cfg in "8 128" "16 512" "25 1024" "32 1024" "32 512"; do set -- $cfg
  python bench.py --atoms-side $1 --inducing $2 --steps 100 --warmup 10 --no-cpu-baseline --no-big-wall 2>/dev/null | tail -1
done > gpurun_out/${tag}_sweep.jsonl
tail -5 gpurun_out/${tag}_md_config5_16384.log
cat gpurun_out/${tag}_update_bench_16384.log | grep -v amdgpu
python - <<PY
import json
for l in open("gpurun_out/${tag}_sweep.jsonl"):
    d=json.loads(l); r=d["roofline"]
    print(d["config"]["atoms"], d["config"]["inducing"], round(d["ms_per_step"]*1e3,1), round(d["value"]/1e6,1), r["kernel"], round(r["frac"],3), r.get("knm_TFs"), r.get("wcov_TFs"), r.get("desc_hbm_frac"))
PY
tail -12 gpurun_out/${tag}_collect.log
