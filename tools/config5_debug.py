#!/usr/bin/env python3
"""Small-size replay of examples/md_nvt_config5.py that checks the edited model against a from-scratch one after
EVERY update step and stops at the first disagreement / non-finite number (debugging aid)."""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("c5", os.path.join(ROOT, "examples", "md_nvt_config5.py"))
c5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(c5)
from autoforce_amd import workloads  # noqa: E402

shape = tuple(int(x) for x in (sys.argv[1:4] or [8, 8, 8]))
m_seed = int(sys.argv[4]) if len(sys.argv) > 4 else 120
cap = int(sys.argv[5]) if len(sys.argv) > 5 else 128
steps = int(sys.argv[6]) if len(sys.argv) > 6 else 60
n_exceed = int(sys.argv[7]) if len(sys.argv) > 7 else 8
calc, teacher, (numbers, pos, cell, pbc), vel0 = workloads.config5_preseeded(shape, m_seed, cap, n_exceed)
np.random.seed(1)
model = calc.model
eng = model.engine
import autoforce_amd.posterior as P
orig = P.PosteriorPotential.make_munu
trace = []


def wrapped(self, *a, **k):
    r = orig(self, *a, **k)
    e = self.engine
    ok = np.isfinite(e.mu).all() and all(np.isfinite(v) for v in self.mean.weights.values()) and \
        all(np.isfinite(v) or v == np.inf for v in e._vscale.values())
    trace.append((len(e.X), len(self.data), e.solve_info(), bool(ok), float(e.sigma or 0), e.ridge, float(np.abs(e.mu).max()),
                  dict(e._vscale), dict(self.mean.weights), self._noise))
    if not ok:
        print("NON-FINITE mu after make_munu:", trace[-6:], flush=True)
        raise SystemExit(1)
    return r


P.PosteriorPotential.make_munu = wrapped
for step, E, T, wall, p, v in workloads.langevin_nvt(calc, numbers, pos, cell, pbc, steps, 600.0, 1.0, 0.1, vel=vel0):
    b = np.asarray(calc.get_covloss())
    pct = np.percentile(b, [50, 90, 99, 99.9, 100]) / calc.ediff
    print(f"{step:4d} E={E:12.5f} T={T:7.1f} size={calc.size} upd={int(calc.updated)} covloss={calc.covlog[:9]} ediff={calc.ediff:.3g} "
          f"wall={1e3 * wall:7.1f} refits={len(trace)} beta/ediff p50,90,99,99.9,max={np.round(pct, 3).tolist()} {eng.solve_info()}", flush=True)
    if not np.isfinite(E):
        print("non-finite energy; last refits:", trace[-8:])
        print("mean:", model.mean, "vscale:", eng._vscale, "mu finite:", np.isfinite(eng.mu).all())
        break
    if calc.updated and (cap <= 256 or os.environ.get("C5_VERIFY")):
        res = dict(calc=calc)
        try:
            print("     verify:", c5.verify(res, tol_choli=1e-8, tol_fit=1e-6))
        except AssertionError as ex:
            print("     VERIFY FAILED:", ex)
            print("     last refits:", trace[-8:])
            break
    trace.clear()
