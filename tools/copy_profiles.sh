#!/bin/bash
# usage (build container, after gpurun merged gpurun_out/): tools/copy_profiles.sh <round tag>
# copies what tools/collect_profiles.sh left under gpurun_out/ into profiles/ (the tracked evidence)
tag=$1
cd "$(dirname "$0")/.."
cp gpurun_out/${tag}_kernel_stats_lips4096_m512.csv profiles/
cp "$(ls gpurun_out/${tag}_pmc_fetch/*/*counter_collection.csv | head -1)" profiles/${tag}_pmc_FETCH_SIZE_lips4096_m512.csv
cp "$(ls gpurun_out/${tag}_pmc_write/*/*counter_collection.csv | head -1)" profiles/${tag}_pmc_WRITE_SIZE_lips4096_m512.csv
cp gpurun_out/${tag}_pmc_traffic_lips4096_m512.json gpurun_out/${tag}_pmc_mfma_summary.txt gpurun_out/${tag}_pmc_sq_summary.txt profiles/
[ -f gpurun_out/${tag}_pmc_traffic_oxide16384_m1024.json ] && cp gpurun_out/${tag}_pmc_traffic_oxide16384_m1024.json profiles/
[ -f gpurun_out/${tag}_bench_line_20steps.json ] && tail -1 gpurun_out/${tag}_bench_line_20steps.json > profiles/${tag}_bench_line_builder_run_20steps.json
[ -f gpurun_out/${tag}_md_config5_16384_1000steps.log ] && cp gpurun_out/${tag}_md_config5_16384_1000steps.log profiles/
[ -f gpurun_out/${tag}_sweep.jsonl ] && cp gpurun_out/${tag}_sweep.jsonl profiles/${tag}_size_sweep.jsonl
[ -f gpurun_out/${tag}_bench_line.json ] && tail -1 gpurun_out/${tag}_bench_line.json > profiles/${tag}_bench_line_builder_run.json
[ -f gpurun_out/${tag}_md_config5_16384.log ] && cp gpurun_out/${tag}_md_config5_16384.log profiles/
[ -f gpurun_out/${tag}_update_bench_16384.log ] && cp gpurun_out/${tag}_update_bench_16384.log profiles/
ls -la profiles/${tag}_*
