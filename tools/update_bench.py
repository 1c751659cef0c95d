#!/usr/bin/env python3
"""BASELINE config 5, the update path in isolation: 16384-atom LiPS frames stored in the resident training set,
m inducing LCEs, then the calls one on-the-fly update makes (add_1inducing trial = add + refit [+ pop + refit],
add_1atoms_fast = push + refit [+ pop + refit]).
usage: python3 tools/update_bench.py [side=32 -> 32x32x16 = 16384 atoms, else side^3] [m=1024] [frames=2]"""
import os, sys, time
import numpy as np
import scipy.optimize  # (the noise search imports it: keep that out of the timings)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel
from autoforce_amd.posterior import PosteriorPotential
from autoforce_amd.sgprio import Frame
from autoforce_amd.workloads import inducing_from_frame, lips

side = int(sys.argv[1]) if len(sys.argv) > 1 else 32
m = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
nfr = int(sys.argv[3]) if len(sys.argv) > 3 else 2
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
rng = np.random.default_rng(0)
frames = []
for k in range(nfr + 1):
    numbers, pos, cell, pbc = lips((32, 32, 16) if side == 32 else side, seed=k)
    N = len(numbers)
    frames.append(Frame(numbers, pos, cell, pbc, float(rng.normal()), 0.1 * rng.normal(size=(N, 3)), 0.01 * rng.normal(size=6)))
N = frames[0].natoms
X = inducing_from_frame(mdl, *frames[0].system(), m + 1, seed=1)
post = PosteriorPotential(mdl)


def tm(label, f, n=1):
    t0 = time.perf_counter()
    for _ in range(n):
        r = f()
    dt = (time.perf_counter() - t0) / n
    print(f"  {label:58s} {1e3 * dt:9.2f} ms", flush=True)
    return r


print(f"{N} atoms per frame, {nfr} stored frames ({sum(1 + 3 * f.natoms + f.nv for f in frames[:nfr])} rows), m = {m}")
tm("set_data (all rows of all frames + first refit)", lambda: post.set_data(frames[:nfr], X[:m]))
tm("make_munu (refit from the resident matrix)", post.make_munu, 3)
tm("add_inducing (1 LCE: K_mm border + 1 column per frame + refit)", lambda: post.add_inducing(X[m]))
tm("pop_1inducing (+ refit)", post.pop_1inducing)
tm("add_1inducing trial (add, refit, energy_of x2, [pop, refit])", lambda: post.add_1inducing(X[m], 1e9))
tm("add_data (1 frame: rows for m columns + refit)", lambda: post.add_data([frames[nfr]]))
tm("pop_1data (+ refit)", post.pop_1data)
tm("add_1atoms_fast trial (push, refit, 2 products, [pop, refit])", lambda: post.add_1atoms_fast(frames[nfr], 1e9, 1e9))
tm("make_munu(algo=3) (noise search: 1 force-only factor + batched scans)", lambda: post.make_munu(algo=3, noise_f=0.05))
# downsize(lii=True) at the size limit: the m LCEs with the smallest K_mm row sums, in argsort order (a permutation)
post.add_inducing(X[m])
tm("downsize(lii=True) m+1 -> m (select through the kept reflectors + refit)", lambda: post.downsize(1e9, m, first=True, lii=True))
print("   ", mdl.solve_info())
post.add_inducing(X[m])
tm("remove an LCE in the middle (+ refit)", lambda: (post.select_inducing([i for i in range(len(post.X)) if i != 7])))
print("   ", mdl.solve_info())
os.environ["SGPR_SOLVE_TIMING"] = "1"
if os.environ.get("SGPR_PROFILE_OPT"):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); post.make_munu(algo=3, noise_f=0.05); post.add_1inducing(X[m], 1e9); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
