"""Large-size check of the kept factorisation (multi-level trees): appended / popped columns, the force-only fit and the
row append against a model that factors from scratch.  usage: python3 tools/keep_check.py [m0=48]"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel
from autoforce_amd.workloads import inducing_from_frame, lips
m0 = int(sys.argv[1]) if len(sys.argv) > 1 else 48
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
frames = [lips((32, 32, 16), seed=k) for k in range(3)]
X = inducing_from_frame(mdl, *frames[0], m0 + 40, seed=1)
mdl.set_inducing(X[:m0])
for fr in frames[:2]:
    mdl.data_push(*fr, 6)
rng = np.random.default_rng(0)
nrow = [1 + 3 * len(f[0]) + 6 for f in frames]
Yall = rng.normal(size=sum(nrow))
worst = 0.0


def check(tag, we=True):
    global worst
    rows = mdl.data_info()[1]
    Y = Yall[:rows]
    got = mdl.data_solve(Y, noise=0.02, with_energies=we).copy()
    ref = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
    ref.set_inducing(mdl.X)
    for fr in frames[:mdl.data_info()[0]]:
        ref.data_push(*fr, 6)
    ref.qr_keep_off = True
    want = ref.data_solve(Y, noise=0.02, with_energies=we)
    K = None
    v = rng.normal(size=len(got))
    a, b = mdl.data_matvec(got), mdl.data_matvec(want)
    err = np.abs(a - b).max() / np.abs(b).max()
    worst = max(worst, err)
    print(f"{tag:50s} m={len(got):4d} rows={rows}: rel. difference of K mu {err:.2e}", flush=True)
    ref.close()


check("full factorisation")
for k in range(5):
    mdl.add_inducing(X[m0 + k]); check(f"append {k}")
check("force-only, full", we=False)
mdl.add_inducing(X[m0 + 5]); check("force-only after an append", we=False); check("append 5")
mdl.remove_inducing(-1); check("pop"); check("force-only after a pop", we=False)
mdl.data_push(*frames[2], 6); check("row append (frame pushed)"); check("row append, force-only", we=False)
mdl.data_pop(-1); check("frame popped: kept slot")
for _ in range(8):
    mdl.remove_inducing(-1)
check("pops into the full factorisation")
mdl.add_inducing(X[m0 + 7]); check("append below the original count")
print("worst", worst)
