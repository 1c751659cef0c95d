#!/usr/bin/env python3
"""Long device MD runs at 4096 / 512: 20000 Langevin steps (600 K) and 20000 velocity-Verlet steps from where they end —
temperature held, total energy conserved to the integrator's order, no capacity flag, no drift of the step time."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoforce_amd.ase_shim import kB
from autoforce_amd.workloads import FS, MASS, fit_to_teacher, lips

numbers, pos, cell, pbc = lips(16, seed=0)
N = len(numbers)
mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
fit_to_teacher(mdl, numbers, pos, cell, pbc)
mass = np.array([MASS[int(z)] for z in numbers])
rng = np.random.default_rng(0)
v0 = rng.normal(size=(N, 3)) * np.sqrt(kB * 600.0 / mass[:, None])
mdl.md_begin(numbers, pos, cell, pbc, mass, v0, dt=FS, friction=1e-2, kT=kB * 600.0, seed=5)
for block in range(10):
    t0 = time.perf_counter()
    sc, code = mdl.md_run(2000, None)
    dt = time.perf_counter() - t0
    T = sc[:, 12] / (3 * N * kB)
    print(f"Langevin block {block}: {1e6 * dt / len(sc):.1f} us/step, code {code}, T {T.mean():.0f} K (last {T[-1]:.0f}), E {sc[-1, 0]:.3f}", flush=True)
    assert code == 0 and len(sc) == 2000 and not sc[:, 10].any()
st = mdl.md_state(0, results=True)
mdl.md_end()
mdl.md_begin(numbers, st["positions"], cell, pbc, mass, st["velocities"], dt=FS, friction=0.0, kT=0.0)
tot = []
for block in range(10):
    sc, code = mdl.md_run(2000, None)
    H = sc[:, 0] + 0.5 * sc[:, 12]
    tot.append(H)
    print(f"NVE block {block}: code {code}, H {H[0]:.4f} .. {H[-1]:.4f}, spread {H.max() - H.min():.4f} eV, T {sc[-1, 12] / (3 * N * kB):.0f} K", flush=True)
    assert code == 0 and not sc[:, 10].any()
H = np.concatenate(tot)
print(f"NVE 20000 steps: drift {H[-100:].mean() - H[:100].mean():+.4f} eV of {abs(H.mean()):.1f} ({(H[-100:].mean() - H[:100].mean()) / N * 1e3:+.4f} meV/atom), rms fluctuation {H.std():.4f} eV")
