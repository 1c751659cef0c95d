#!/usr/bin/env python3
"""Candidate lists under strained cells, randomised: triclinic cells, random strain and thermal walks with occasional jumps and
wraps; every step the handle with kept candidates must equal, bit for bit, a handle that rebuilds its lists every step.
usage: python3 tools/fuzz_npt.py [seeds=6] [steps=120]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from autoforce_amd import _lib  # noqa: E402
from test_hip_parity import load, model_from_fixture  # noqa: E402

nseeds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 120
tot_fast = tot_slow = 0
for seed in range(nseeds):
    name = ["g5_mixed64", "g5_tric24", "g5_si32", "g5_big40", "g5_bigtric36"][seed % 5]
    g = load(name)
    fast, slow = model_from_fixture(g), model_from_fixture(g)
    for mdl in (fast, slow):
        mdl.set_weights(g["mu"], choli=g["choli"])
    _lib.check(_lib.load().sgpr_set_option(slow.handle, b"skin_milliangstrom", 0))
    rng = np.random.default_rng(500 + seed)
    pos, cell = g["positions"].copy(), g["cell"].copy()
    N = len(pos)
    amp = [1e-4, 5e-4, 2e-3][seed % 3]
    for step in range(steps):
        strain = np.eye(3) + amp * rng.normal(size=(3, 3))
        frac = np.linalg.solve(cell.T, pos.T).T
        cell = cell @ strain
        pos = frac @ cell + 0.012 * rng.normal(size=pos.shape)
        if rng.random() < 0.03:
            pos[rng.integers(N)] += rng.normal(size=3) * 0.6
        if rng.random() < 0.02 and np.all(g["pbc"]):
            pos[rng.integers(N)] += cell[rng.integers(3)]
        a = fast.predict(g["numbers"], pos, cell, g["pbc"], cov=True)
        b = slow.predict(g["numbers"], pos, cell, g["pbc"], cov=True)
        for k in ("energy", "forces", "stress", "beta", "cov"):
            if not np.array_equal(np.asarray(a[k]), np.asarray(b[k])):
                print(f"seed {seed} ({name}) step {step}: {k} differs by {np.abs(np.asarray(a[k]) - np.asarray(b[k])).max():.3e}")
                raise SystemExit(1)
        for x, y in zip(fast.neighbors(N), slow.neighbors(N)):
            assert np.array_equal(x, y), (seed, step)
    print(f"seed {seed} ({name}, strain {amp:g}/step): ok, rebuilds {fast.list_rebuilds()} of {steps} (every-step handle: {slow.list_rebuilds()})", flush=True)
    tot_fast += fast.list_rebuilds(); tot_slow += slow.list_rebuilds()
    fast.close(); slow.close()
print(f"all ok; rebuilds {tot_fast} vs {tot_slow}")
