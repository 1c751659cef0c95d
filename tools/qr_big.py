import sys, time, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import oracle as orc
from autoforce_amd.workloads import lips, inducing_from_frame
from autoforce_amd import SGPRModel
numbers, pos, cell, pbc = lips(16, seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
n2, p2, c2, b2 = lips(16, seed=1)
rng = np.random.default_rng(0)
for m, rows in ((96, 9000), (200, 30001), (512, 49180), (512, 120000)):
    X = inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=1)
    mdl.set_inducing(X)
    K = rng.normal(size=(rows, m)); Y = rng.normal(size=rows)
    mdl.solve(K, Y)
    t = time.time(); mu = mdl.solve(K, Y); dt = time.time() - t
    msg = f"m={m} rows={rows}: {dt*1e3:.1f} ms"
    if rows <= 50000:
        ref = orc.regression(mdl.M, K, Y, noise0=0.01)
        msg += f"  err vs oracle {np.abs(mu - ref['mu']).max() / np.abs(ref['mu']).max():.2e}"
    print(msg, flush=True)
