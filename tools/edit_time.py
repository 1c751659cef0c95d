#!/usr/bin/env python3
"""Wall time of the inducing-set edit entry points through the Python API (frame bound, as inside an MD run):
full set_inducing, incremental add_inducing, remove_inducing(-1), M download, and the solve that follows.
usage: python3 tools/edit_time.py [atoms_side=16] [m=512]"""
import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from autoforce_amd.workloads import lips, inducing_from_frame
from autoforce_amd import SGPRModel
side = int(sys.argv[1]) if len(sys.argv) > 1 else 16
m = int(sys.argv[2]) if len(sys.argv) > 2 else 512
numbers, pos, cell, pbc = lips(side, seed=0)
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
n2, p2, c2, b2 = lips(side, seed=1)
X = inducing_from_frame(mdl, n2, p2, c2, b2, m + 4, seed=1)


def t(f, n=3):
    best = 1e9
    for _ in range(n):
        t0 = time.perf_counter(); f(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


mdl.set_inducing(X[:m])
rng = np.random.default_rng(0)
K, Y = rng.normal(size=(256, m)), rng.normal(size=256)
mdl.solve(K, Y)
mdl.set_weights(mdl.mu, choli=mdl.choli, vscale=mdl.make_vscale())
mdl.predict(numbers, pos, cell, pbc)  # a frame is bound: edits also resize the per-frame work arrays
print(f"{len(numbers)} atoms bound, m = {m}")
print(f"  set_inducing (full rebuild)      {t(lambda: mdl.set_inducing(X[:m])):8.2f} ms")
mdl.solve(K, Y)
t0 = time.perf_counter(); mdl.add_inducing(X[m]); ta = time.perf_counter() - t0
t0 = time.perf_counter(); M = mdl.M; tm = time.perf_counter() - t0
K1 = np.concatenate([K, rng.normal(size=(256, 1))], axis=1)
t0 = time.perf_counter(); mdl.solve(K1, Y); ts = time.perf_counter() - t0
t0 = time.perf_counter(); mdl.remove_inducing(-1); tr = time.perf_counter() - t0
t0 = time.perf_counter(); mdl.solve(K, Y); ts2 = time.perf_counter() - t0
print(f"  add_inducing (bordered)          {ta * 1e3:8.2f} ms")
print(f"  M download                       {tm * 1e3:8.2f} ms")
print(f"  solve after add (cached factor)  {ts * 1e3:8.2f} ms   (256 rows)")
print(f"  remove_inducing(-1)              {tr * 1e3:8.2f} ms")
print(f"  solve after pop (cached factor)  {ts2 * 1e3:8.2f} ms")
mdl.set_inducing(X[:m])
t0 = time.perf_counter(); mdl.solve(K, Y); ts3 = time.perf_counter() - t0
print(f"  solve after a rebuild (ladder)   {ts3 * 1e3:8.2f} ms")
