#!/bin/bash
# usage (GPU box): tools/bm_sweep.sh — bench.py at several sizes under the tile-shape overrides (see README)
for kd in 16,16 32,32; do
  for sz in "" "--atoms-side 8 --inducing 128" "--atoms-side 32 --inducing 512"; do
    echo "KD=$kd $sz: $(SGPR_GEMM_KD=$kd python bench.py --no-cpu-baseline $sz --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,1), d['roofline']['stage_us'], d['roofline'].get('gemm_TFLOPs'))")"
  done
done
