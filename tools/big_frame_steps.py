#!/usr/bin/env python3
"""K eager predict steps of the 16384-atom 4-species oxide frame with 1024 inducing LCEs (BASELINE configs[4] size) —
the program tools/collect_profiles.sh puts under `rocprofv3 --pmc` for profiles/rNN_pmc_traffic_oxide16384_m1024.json
(the frame bench.py's `roofline_16384` is measured on).  usage: big_frame_steps.py [steps]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    from autoforce_amd import SGPRModel, _lib
    from autoforce_amd.workloads import inducing_from_frame, oxide
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    numbers, pos, cell, pbc = oxide(seed=0)
    mdl = SGPRModel(3, 3, 4, 6.0, species=sorted(set(int(z) for z in numbers)), device=0)
    n2, p2, c2, b2 = oxide(seed=1)
    mdl.set_inducing(inducing_from_frame(mdl, n2, p2, c2, b2, 1024, seed=1))
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(64, 1024)), rng.normal(size=64))
    mdl.set_weights(rng.normal(size=1024), choli=mdl.choli, vscale=mdl.make_vscale())
    mdl.predict(numbers, pos, cell, pbc)  # binds the system
    lib, h, N = _lib.load(), mdl.handle, len(numbers)
    dev = torch.device("cuda", 0)
    pos_d = torch.from_numpy(pos).to(dev)
    cell_d = torch.from_numpy(np.ascontiguousarray(cell, np.float64)).to(dev)
    packed = torch.zeros(int(lib.sgpr_packed_len(N)), dtype=torch.float64, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(steps):
        _lib.check(lib.sgpr_step_dev(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), sp))
        torch.cuda.synchronize(dev)
    print("steps", steps, "energy", float(packed[4 * N]))
    mdl.close()


if __name__ == "__main__":
    main()
