#!/usr/bin/env python3
"""Large column selections on the kept factorisation (debugging aid): m0 LCEs -> keep m1 in argsort-of-row-sum order,
refit through the kept reflectors, compare with a from-scratch model."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel, workloads  # noqa: E402

m0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1416
m1 = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 3
push_after = len(sys.argv) > 4 and sys.argv[4] == "push"
shape = (8, 8, 8)
numbers, pos, cell, pbc = workloads.oxide_ordered(shape, seed=0)
species = sorted(set(int(z) for z in numbers))
mdl = SGPRModel(3, 3, 4, 6.0, species=species)
X = []
k = 0
while len(X) < m0:
    n2, p2, c2, b2 = workloads.oxide_ordered(shape, seed=100 + k, sigma=0.05 + 0.01 * (k % 5))
    X += workloads.inducing_from_frame(mdl, n2, p2, c2, b2, min(400, m0 - len(X)), seed=200 + k, noise=0.0)
    k += 1
mdl.set_inducing(X)
rng = np.random.default_rng(0)
frames = []
for f in range(nframes + 1):
    n2, p2, c2, b2 = workloads.oxide_ordered(shape, seed=300 + f, sigma=0.08)
    frames.append((n2, p2, c2, b2))
for fr in frames[:nframes]:
    mdl.data_push(*fr, 6)
rows = mdl.data_info()[1]
Y = rng.normal(size=rows + 1 + 3 * len(numbers) + 6)
mu = mdl.data_solve(Y[:rows], noise=0.02)
print("first solve:", mdl.solve_info(), "finite", np.isfinite(mu).all(), "rows", rows)


def check(tag, Yv):
    mu = mdl.data_solve(Yv, noise=0.02).copy()
    info = mdl.solve_info()
    ref = mdl.scratch()
    ref.set_inducing(mdl.X)
    for fr in frames[:mdl.data_info()[0]]:
        ref.data_push(*fr, 6)
    want = ref.data_solve(Yv, noise=0.02)
    p0, p1 = ref.data_matvec(want), mdl.data_matvec(mu)
    print(f"{tag}: m={mdl.m} {info} finite={np.isfinite(mu).all()} fit_err={np.abs(p0 - p1).max() / np.abs(p0).max():.3e} "
          f"choli_err={np.abs(ref.choli - mdl.choli).max() / np.abs(ref.choli).max():.3e} kmm_equal={np.array_equal(ref.M, mdl.M)}", flush=True)
    ref.close()


order = np.argsort(mdl.M.sum(axis=1), kind="stable").tolist()[:m1]
mdl.select_inducing(order)
if push_after:
    mdl.data_push(*frames[nframes], 6)
    check("select + push", Y)
else:
    check("select", Y[:rows])
mdl.add_inducing(X[0].__class__(X[0].number, X[0]._b, X[0]._r + 0.01))
check("append", Y if push_after else Y[:rows])
mdl.close()
