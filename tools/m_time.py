import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from autoforce_amd.workloads import lips, inducing_from_frame
from autoforce_amd import SGPRModel
numbers, pos, cell, pbc = lips(int(sys.argv[1]) if len(sys.argv) > 1 else 26, seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
X = inducing_from_frame(mdl, numbers, pos, cell, pbc, 24, seed=1)
mdl.set_inducing(X)
def tm(label, f):
    t0 = time.perf_counter(); r = f(); print(f"{label:40s} {1e3*(time.perf_counter()-t0):8.3f} ms"); return r
tm("M (fresh)", lambda: mdl.M)
Ke, Kf, Kv = tm("kernel_rows", lambda: mdl.kernel_rows(numbers, pos, cell, pbc))
tm("M after rows", lambda: mdl.M)
tm("M again", lambda: mdl.M)
K = np.concatenate([Ke[None], Kf, Kv]); Y = np.random.default_rng(0).normal(size=len(K))
tm("solve", lambda: mdl.solve(K, Y))
tm("M after solve", lambda: mdl.M)
tm("M again", lambda: mdl.M)
tm("set_weights", lambda: mdl.set_weights(mdl.mu, choli=mdl.choli))
tm("M after set_weights", lambda: mdl.M)
tm("predict", lambda: mdl.predict(numbers, pos, cell, pbc))
tm("M after predict", lambda: mdl.M)
tm("M again", lambda: mdl.M)
tm("make_vscale", lambda: mdl.make_vscale())
tm("M after vscale", lambda: mdl.M)
