"""sgpr_kernel_rows at 4096 atoms / 512 inducing columns: wall time of the call (host arrays in and out)."""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from autoforce_amd.workloads import lips, inducing_from_frame
from autoforce_amd import SGPRModel
numbers, pos, cell, pbc = lips(16, seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
n2, p2, c2, b2 = lips(16, seed=1)
X = inducing_from_frame(mdl, n2, p2, c2, b2, 512, seed=1)
mdl.set_inducing(X)
ts = []
for _ in range(6):
    t = time.perf_counter(); Ke, Kf, Kv = mdl.kernel_rows(numbers, pos, cell, pbc); ts.append(time.perf_counter() - t)
print("kernel_rows(4096 atoms, 512 columns) ms:", " ".join(f"{1e3*t:.2f}" for t in ts))
print("species of the columns:", np.bincount([x.number for x in X]).nonzero()[0], np.bincount([x.number for x in X])[[3, 15, 16]])
ts = []
for _ in range(4):
    t = time.perf_counter(); mdl.data_push(numbers, pos, cell, pbc, 6); ts.append(time.perf_counter() - t); mdl.data_pop(-1)
print("data_push (rows computed into the resident matrix, nothing crosses PCIe) ms:", " ".join(f"{1e3*t:.2f}" for t in ts))
