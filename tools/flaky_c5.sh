#!/bin/bash
# tools/flaky_c5.sh <label> <runs> [ENV=... ...]: the two-rank config-5 run repeated; prints the failures (diagnostic)
label=$1; runs=$2; shift 2
fails=0
for i in $(seq 1 $runs); do
  env "$@" HSA_ENABLE_IPC_MODE_LEGACY=0 SGPR_PEER_TIMEOUT_MS=8000 timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 400)) examples/md_nvt_config5.py --steps 60 --m-seed 1016 > gpurun_out/fl_${label}_$i.log 2>&1
  rc=$?
  if [ $rc -ne 0 ]; then fails=$((fails+1)); echo "$label run $i rc=$rc: $(grep -v 'amdgpu.ids\|hostname\|Gloo' gpurun_out/fl_${label}_$i.log | grep 'SgprError\|AssertionError' | head -2 | cut -c1-220)"; else rm -f gpurun_out/fl_${label}_$i.log; fi
done
echo "$label: $fails failures of $runs"
