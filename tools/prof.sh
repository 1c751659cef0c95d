#!/bin/bash
# usage (on the GPU box, via gpurun): tools/prof.sh <tag> [bench args]
tag=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$tag -- python3 bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-big-wall "$@" > gpurun_out/$tag/run.log 2>&1
echo rc=$?
python3 tools/kstats.py gpurun_out/$tag 90
