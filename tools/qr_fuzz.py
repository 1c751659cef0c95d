import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from oracle import oracle as orc
from test_hip_parity import load, model_from_fixture
from autoforce_amd import Local
g = load("g5_big40")
base = model_from_fixture(g)
X = list(base.X)
rng = np.random.default_rng(0)
worst = 0
for m in (1, 2, 5, 12):
    mdl = base.scratch(); mdl.set_inducing(X[:m])
    M = mdl.M
    for rows in (0, 1, 3, m, m + 1, 17, 40, 130, 700, 1500):
        K = rng.normal(size=(rows, m)); Y = rng.normal(size=rows)
        mu = mdl.solve(K, Y, noise=0.01)
        ref = orc.regression(M, K, Y, noise0=0.01)
        err = np.abs(mu - ref["mu"]).max() / max(np.abs(ref["mu"]).max(), 1e-300) if rows else np.abs(mu).max()
        mu2 = mdl.resolve(noise=0.05); ref2 = orc.regression(M, K, Y, noise0=0.05)
        err2 = np.abs(mu2 - ref2["mu"]).max() / max(np.abs(ref2["mu"]).max(), 1e-300) if rows else np.abs(mu2).max()
        worst = max(worst, err, err2)
        if max(err, err2) > 1e-8: print("BAD", m, rows, err, err2)
print("worst", worst)
