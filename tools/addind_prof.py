"""One appended LCE with F stored 16384-atom frames (sgpr_add_inducing: K_mm border + the new column's rows for every
stored frame): the workload of a kernel trace.  usage: python3 tools/addind_prof.py [frames=4] [m=512]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel
from autoforce_amd.workloads import inducing_from_frame, lips
nfr = int(sys.argv[1]) if len(sys.argv) > 1 else 4
m = int(sys.argv[2]) if len(sys.argv) > 2 else 512
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
numbers, pos, cell, pbc = lips((32, 32, 16), seed=0)
rng = np.random.default_rng(0)
frames = [(numbers, pos + 0.05 * rng.normal(size=pos.shape), cell, pbc) for k in range(nfr)]  # one system, as in an MD run
X = inducing_from_frame(mdl, *frames[0], m + 8, seed=1)
mdl.set_inducing(X[:m])
for fr in frames:
    mdl.data_push(*fr, 6)
ts = []
for k in range(6):
    t = time.perf_counter(); mdl.add_inducing(X[m + k]); ts.append(time.perf_counter() - t)
print(f"add_inducing with {nfr} stored frames, m = {m}: ms", " ".join(f"{1e3 * t:.2f}" for t in ts))
