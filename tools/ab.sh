#!/bin/bash
# usage (GPU box): tools/ab.sh "<bench args A>" "<bench args B>" [reps]: ms_per_step of two bench configurations, interleaved
A=$1; B=$2; R=${3:-3}
for r in $(seq $R); do
  for v in A B; do
    args=$A; [ $v = B ] && args=$B
    python3 bench.py --steps 300 --warmup 20 --no-cpu-baseline $args 2>/dev/null | grep "^{" | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v [$args]', 'ms_per_step=%.4f' % d['ms_per_step'], 'rebuilds', d['config'].get('list_rebuilds_in_timed_steps'), {k: round(v, 1) for k, v in d['roofline'].get('stage_us', {}).items()})"
  done
done
