#!/usr/bin/env python3
"""data_push of one 16384-atom frame against m inducing LCEs (the rows of add_1atoms_fast's trial): workload for a kernel trace."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel, workloads  # noqa: E402

m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
numbers, pos, cell, pbc = workloads.oxide_ordered((32, 32, 16), seed=0, sigma=0.08)
species = sorted(set(int(z) for z in numbers))
mdl = SGPRModel(3, 3, 4, 6.0, species=species)
n2, p2, c2, b2 = workloads.oxide_ordered((32, 32, 16), seed=1, sigma=0.08)
mdl.set_inducing(workloads.inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=1, noise=0.0))
for k in range(3):
    t0 = time.time()
    mdl.data_push(numbers, pos + 0.01 * k, cell, pbc, 6)
    print(f"data_push {k}: {1e3 * (time.time() - t0):.1f} ms")
    mdl.data_pop(-1)
mdl.close()
