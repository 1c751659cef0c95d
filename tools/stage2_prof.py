"""Twenty second-stage solves (sgpr_resolve: the banded 2m x m QR) at m = 1024 on a 16384-atom frame: the workload of a
kernel trace of the second stage alone.  usage: python3 tools/stage2_prof.py [m=1024]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel
from autoforce_amd.workloads import inducing_from_frame, lips
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
fr = lips((32, 32, 16), seed=0)
mdl.set_inducing(inducing_from_frame(mdl, *fr, m, seed=1))
mdl.data_push(*fr, 6)
rows = mdl.data_info()[1]
Y = np.random.default_rng(0).normal(size=rows)
mdl.data_solve(Y, noise=0.01)
t0 = time.perf_counter()
for k in range(20):
    mdl.resolve(noise=0.01 + 0.001 * k)
print(f"resolve m={m}: {1e3 * (time.perf_counter() - t0) / 20:.2f} ms each")
