#!/usr/bin/env python3
"""Can two ranks share ONE GPU under RCCL?  (probe for a two-rank test of the native collective on a one-GPU box)"""
import multiprocessing as mp
import os
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, q_id, q_out):
    def bomb():
        print(f"[probe] rank {rank}: stuck in comm_init / all-reduce for 40 s", flush=True)
        os._exit(3)
    t = threading.Timer(40.0, bomb)
    t.daemon = True
    t.start()
    import numpy as np
    from autoforce_amd import SGPRModel
    mdl = SGPRModel(3, 3, 4, 6.0, species=[14], device=0)
    if rank == 0:
        uid = mdl.comm_unique_id()
        q_id.put(uid)
    else:
        uid = q_id.get(timeout=30)
    try:
        mdl.comm_init(uid, rank, 2)
        print(f"[probe] rank {rank}: comm_init ok", flush=True)
        import torch
        buf = torch.full((8,), float(rank + 1), dtype=torch.float64, device="cuda:0")
        from autoforce_amd import _lib
        _lib.check(_lib.load().sgpr_comm_allreduce(mdl.handle, buf.data_ptr(), 8, 0, None))
        torch.cuda.synchronize()
        print(f"[probe] rank {rank}: all-reduce -> {buf.cpu().numpy()[:2]}", flush=True)
    except Exception as e:  # noqa: BLE001
        print(f"[probe] rank {rank}: failed: {e}", flush=True)
    t.cancel()
    os._exit(0)


if __name__ == "__main__":
    mp.set_start_method("spawn")
    q_id, q_out = mp.Queue(), mp.Queue()
    ps = [mp.Process(target=worker, args=(r, q_id, q_out)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join(60)
    while not q_out.empty():
        print(q_out.get())
    print("exit codes", [p.exitcode for p in ps])
