#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <round tag, e.g. r02> <source note>
# kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes (separate, as MI355X_MICROARCH.md prescribes), MFMA pass
tag=$1; note=$2
cd $GRAFT_REPO_ROOT
bash tools/prof.sh ${tag}_final > gpurun_out/${tag}_final_kstats.txt 2>&1
bash tools/pmc.sh ${tag}_pmc_fetch FETCH_SIZE --no-big-wall > gpurun_out/${tag}_pmc_fetch.txt 2>&1
bash tools/pmc.sh ${tag}_pmc_write WRITE_SIZE --no-big-wall > gpurun_out/${tag}_pmc_write.txt 2>&1
f=$(ls gpurun_out/${tag}_pmc_fetch/*/*counter_collection.csv | head -1)
w=$(ls gpurun_out/${tag}_pmc_write/*/*counter_collection.csv | head -1)
bash tools/pmc.sh ${tag}_pmc_sq "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY" > gpurun_out/${tag}_pmc_sq.txt 2>&1
bash tools/pmc.sh ${tag}_pmc_lds "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES" > gpurun_out/${tag}_pmc_lds.txt 2>&1
q=$(ls gpurun_out/${tag}_pmc_sq/*/*counter_collection.csv | head -1)
qt=$(ls gpurun_out/${tag}_pmc_sq/*/*kernel_trace.csv | head -1)
python3 tools/pmc_json.py $f $w gpurun_out/${tag}_pmc_traffic_lips4096_m512.json "$note" $q $qt > /dev/null
cp gpurun_out/${tag}_pmc_sq.txt gpurun_out/${tag}_pmc_sq_summary.txt; tail -n +2 gpurun_out/${tag}_pmc_lds.txt >> gpurun_out/${tag}_pmc_sq_summary.txt
bash tools/pmc_mfma.sh ${tag}_pmc_mfma bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-big-wall > gpurun_out/${tag}_pmc_mfma_summary.txt 2>&1
# the 16384-atom / 1024-inducing frame of bench.py's roofline_16384: FETCH_SIZE / WRITE_SIZE passes of tools/big_frame_steps.py
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p gpurun_out/${tag}_pmc16k_$c
  (cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/${tag}_pmc16k_$c -- python3 tools/big_frame_steps.py 20 > gpurun_out/${tag}_pmc16k_$c/run.log 2>&1)
done
PMC_WORKLOAD="4-species oxide 16384 atoms / 1024 inducing, 1 GPU (tools/big_frame_steps.py)" python3 tools/pmc_json.py \
  $(ls gpurun_out/${tag}_pmc16k_FETCH_SIZE/*/*counter_collection.csv | head -1) $(ls gpurun_out/${tag}_pmc16k_WRITE_SIZE/*/*counter_collection.csv | head -1) \
  gpurun_out/${tag}_pmc_traffic_oxide16384_m1024.json "$note" > /dev/null
cp $(ls gpurun_out/${tag}_final/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats_lips4096_m512.csv
cat gpurun_out/${tag}_final_kstats.txt | tail -12
cat gpurun_out/${tag}_pmc_mfma_summary.txt | tail -8
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_pmc_traffic_lips4096_m512.json'))
for k,v in d['kernels'].items(): print(k, {a: round(b/1e6,2) for a,b in v.items()}, 'MB')"
