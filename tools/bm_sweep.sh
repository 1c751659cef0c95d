python -m pytest tests/test_hip_calculator.py tests/test_hip_parity.py -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; grep -E "passed|failed|error" gpurun_out/pytest_gpu.log | tail -5
for sz in "" "--atoms-side 32 --inducing 512" "--atoms-side 48 --inducing 512"; do
  echo "SWEEP $sz: $(python bench.py --no-cpu-baseline $sz --steps 100 --warmup 10 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1e3,1), round(d['value']/1e6,1), d['roofline']['kernel'], round(d['roofline']['frac'],2), d['roofline']['stage_us'], d['roofline'].get('gemm_TFLOPs'))")"
done
