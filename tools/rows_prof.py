"""Rows of a 16384-atom frame for 1024 columns (data_push into the resident matrix): wall per call; run under
rocprofv3 --kernel-trace --stats for the per-kernel split.  usage: python3 tools/rows_prof.py [calls=3]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel
from autoforce_amd.workloads import inducing_from_frame, lips
calls = int(sys.argv[1]) if len(sys.argv) > 1 else 3
numbers, pos, cell, pbc = lips((32, 32, 16), seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
n2, p2, c2, b2 = lips((32, 32, 16), seed=1)
mdl.set_inducing(inducing_from_frame(mdl, n2, p2, c2, b2, 1024, seed=1))
ts = []
for _ in range(calls + 1):
    t = time.perf_counter(); mdl.data_push(numbers, pos, cell, pbc, 6); ts.append(time.perf_counter() - t); mdl.data_pop(-1)
print("data_push(16384 atoms, 1024 columns) ms:", " ".join(f"{1e3 * t:.2f}" for t in ts), mdl.solve_info())
