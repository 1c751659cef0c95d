#!/bin/bash
# tools/ipc_probe.sh <world> [len] [iters] [fine]: `world` processes on device 0 exchanging through hipIpc-mapped buffers
cd "$(dirname "$0")"
W=${1:-2}; D=$(mktemp -d)
export HSA_ENABLE_IPC_MODE_LEGACY=0
pids=()
for r in $(seq 0 $((W-1))); do timeout 120 ./ipc_probe $r $W $D ${2:-16395} ${3:-400} ${4:-1} & pids+=($!); done
rc=0; for p in "${pids[@]}"; do wait $p || rc=$?; done
rm -rf $D; echo "ipc_probe world $W: rc $rc"; exit $rc
