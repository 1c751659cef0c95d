import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from autoforce_amd.workloads import lips, inducing_from_frame
from autoforce_amd import SGPRModel
numbers, pos, cell, pbc = lips(16, seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
n2, p2, c2, b2 = lips(16, seed=1)
X = inducing_from_frame(mdl, n2, p2, c2, b2, 512, seed=1)
mdl.set_inducing(X)
rng = np.random.default_rng(0)
K = rng.normal(size=(49180, 512)); Y = rng.normal(size=len(K))
mdl.solve(K, Y)
for _ in range(3):
    t = time.time(); mdl.solve(K, Y); print(f"solve {1e3*(time.time()-t):.1f} ms")
for _ in range(3):
    t = time.time(); mdl.resolve(0.02); print(f"resolve {1e3*(time.time()-t):.1f} ms")
