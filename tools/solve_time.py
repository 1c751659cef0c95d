import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from autoforce_amd.workloads import lips, inducing_from_frame
from autoforce_amd import SGPRModel
numbers, pos, cell, pbc = lips(16, seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
n2, p2, c2, b2 = lips(16, seed=1)
for m in (128, 512):
    X = inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=1)
    mdl.set_inducing(X)
    t = time.time(); Ke, Kf, Kv = mdl.kernel_rows(numbers, pos, cell, pbc); t1 = time.time() - t
    t = time.time(); Ke, Kf, Kv = mdl.kernel_rows(numbers, pos, cell, pbc); t1 = time.time() - t
    print(f"m={m}: kernel_rows(4096 atoms) {t1*1e3:.1f} ms")
    t = time.time(); c = mdl.kernel_columns(numbers, pos, cell, pbc, m - 1, 1); print(f"   one column {1e3*(time.time()-t):.2f} ms")
    rng = np.random.default_rng(0)
    for nfr in (1, 4):
        K = np.concatenate([Ke[None]] * nfr + [Kf] * nfr + [Kv] * nfr)
        Y = rng.normal(size=len(K))
        mdl.solve(K, Y)
        t = time.time(); mdl.solve(K, Y); dt = time.time() - t
        print(f"   solve rows={len(K)} m={m}: {dt*1e3:.1f} ms  ({2*len(K)*m*m/dt/1e9:.1f} GF/s)")
    t = time.time(); mdl.add_inducing(X[0]); print(f"   add_inducing rebuild {1e3*(time.time()-t):.2f} ms"); mdl.remove_inducing(-1)
    K = np.concatenate([Ke[None]] * 4 + [Kf] * 4 + [Kv] * 4); Y = rng.normal(size=len(K))
    mdl.solve(K, Y, noise=0.01); ref = mdl.mu.copy()
    t = time.time(); mu2 = mdl.resolve(noise=0.02); print(f"   resolve (cached factor) m={m}: {1e3*(time.time()-t):.1f} ms")
    mu3 = mdl.resolve(noise=0.01); print("   resolve(0.01) vs solve(0.01):", np.abs(mu3 - ref).max() / np.abs(ref).max())
