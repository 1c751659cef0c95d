python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5
for i in 1 2; do python bench.py --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,1), d['roofline']['stage_us'])"; done
