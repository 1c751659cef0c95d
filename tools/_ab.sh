python -m pytest tests/test_hip_paths.py tests/test_hip_parity.py -m gpu -x -q 2>&1 | tail -3
for w in 4 8 4 8; do SGPR_GEMM_WAVES=$w python bench.py --steps 300 --warmup 30 --no-big-wall 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print('waves',$w,'ms',round(d['ms_per_step'],5),r['stage_us'])"; done
python tools/stamps_wcov.py 2>&1 | grep -A8 "== knm"
