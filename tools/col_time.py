import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from autoforce_amd.workloads import lips, inducing_from_frame
from autoforce_amd import SGPRModel
numbers, pos, cell, pbc = lips(16, seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
n2, p2, c2, b2 = lips(16, seed=1)
X = inducing_from_frame(mdl, n2, p2, c2, b2, 512, seed=1)
mdl.set_inducing(X)
mdl.kernel_rows(numbers, pos, cell, pbc)
for q in (511, 0, 200, 511, 511):
    t = time.time(); mdl.kernel_columns(numbers, pos, cell, pbc, q, 1); print(f"column {q}: {1e3*(time.time()-t):.2f} ms")
t = time.time(); mdl.kernel_columns(numbers, pos, cell, pbc, 0, 8); print(f"8 columns: {1e3*(time.time()-t):.2f} ms")
