#!/bin/bash
# usage (GPU box): tools/ab_sizes.sh "<ENV=.. ENV=..>" ... — bench stage times over frame sizes under each environment
cd $GRAFT_REPO_ROOT
SIZES=${SIZES:-"18,512 20,512 22,512 22,1024 25,1024 28,1024"}
for envs in "$@"; do
  for cfg in $SIZES; do
    side=${cfg%,*}; m=${cfg#*,}
    echo -n "[$envs] side $side m $m: "
    env $envs python3 bench.py --atoms-side $side --inducing $m --steps 50 --warmup 10 --md-steps 0 --no-cpu-baseline --no-big-wall 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print(round(d['ms_per_step_resident_frames']*1e3,1), 'knm', r['stage_us']['gemm_knm'], r['knm_frac'], 'wcov', r['stage_us'].get('gemm_w_covloss'), r['wcov_frac'])"
  done
done
