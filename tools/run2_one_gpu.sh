#!/bin/bash
# two ranks on ONE GPU with the host-staged collective: everything of the N>1 path except RCCL itself
cd $GRAFT_REPO_ROOT
timeout 250 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 20 --warmup 3 --collective torch --no-cpu-baseline 2>&1 | tail -5
