#!/bin/bash
# usage (GPU box): tools/collect_profiles.sh <round tag, e.g. r02> <source note>
# kernel trace + stats, FETCH_SIZE / WRITE_SIZE passes (separate, as MI355X_MICROARCH.md prescribes), MFMA pass
tag=$1; note=$2
cd $GRAFT_REPO_ROOT
bash tools/prof.sh ${tag}_final > gpurun_out/${tag}_final_kstats.txt 2>&1
bash tools/pmc.sh ${tag}_pmc_fetch FETCH_SIZE > gpurun_out/${tag}_pmc_fetch.txt 2>&1
bash tools/pmc.sh ${tag}_pmc_write WRITE_SIZE > gpurun_out/${tag}_pmc_write.txt 2>&1
f=$(ls gpurun_out/${tag}_pmc_fetch/*/*counter_collection.csv | head -1)
w=$(ls gpurun_out/${tag}_pmc_write/*/*counter_collection.csv | head -1)
python3 tools/pmc_json.py $f $w gpurun_out/${tag}_pmc_traffic_lips4096_m512.json "$note" > /dev/null
bash tools/pmc_mfma.sh ${tag}_pmc_mfma bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${tag}_pmc_mfma_summary.txt 2>&1
cp $(ls gpurun_out/${tag}_final/*/*kernel_stats.csv | head -1) gpurun_out/${tag}_kernel_stats_lips4096_m512.csv
cat gpurun_out/${tag}_final_kstats.txt | tail -12
cat gpurun_out/${tag}_pmc_mfma_summary.txt | tail -8
python3 -c "
import json; d=json.load(open('gpurun_out/${tag}_pmc_traffic_lips4096_m512.json'))
for k,v in d['kernels'].items(): print(k, {a: round(b/1e6,2) for a,b in v.items()}, 'MB')"
