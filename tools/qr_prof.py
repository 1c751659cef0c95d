"""Two refits from the resident design matrix (16384-atom frames): the workload of tools/prof-style kernel traces
of the QR kernels.  usage: python3 tools/qr_prof.py [m=1024] [frames=2]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel
from autoforce_amd.workloads import inducing_from_frame, lips
m = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nfr = int(sys.argv[2]) if len(sys.argv) > 2 else 2
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
frames = [lips((32, 32, 16), seed=k) for k in range(nfr)]
mdl.set_inducing(inducing_from_frame(mdl, *frames[0], m, seed=1))
for fr in frames:
    mdl.data_push(*fr, 6)
rows = mdl.data_info()[1]
Y = np.random.default_rng(0).normal(size=rows)
for _ in range(3):
    t0 = time.perf_counter(); mdl.data_solve(Y, noise=0.01); print(f"data_solve rows={rows} m={m}: {1e3*(time.perf_counter()-t0):.2f} ms")
