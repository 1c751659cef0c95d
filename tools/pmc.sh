#!/bin/bash
# usage: tools/pmc.sh <tag> "<counters>" [bench args]
tag=$1; shift; ctrs=$1; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/$tag
rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d gpurun_out/$tag -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-big-wall "$@" > gpurun_out/$tag/run.log 2>&1
echo rc=$?
python3 tools/pmcsum.py gpurun_out/$tag
