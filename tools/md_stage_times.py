#!/usr/bin/env python3
"""Per-kernel times of a device MD step (sgpr_md_run) next to those of a resident-frames step (sgpr_step_dev_next):
hip events on the launch stream, LiPS 4096 / 512."""
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
mdl.md_begin(numbers, pos, cell, pbc, mass, v0, dt=FS, friction=float(sys.argv[1]) if len(sys.argv) > 1 else 1e-3, kT=kB * 600.0)
mdl.md_run(300, rng.normal(size=(300, N, 3)))
mdl.profile(True)
acc = {}
for _ in range(20):
    mdl.md_run(8, rng.normal(size=(8, N, 3)))
    for k, v in mdl.stage_times().items():
        acc.setdefault(k, []).append(v)
mdl.profile(False)
print({k: round(1e3 * float(np.median(v)), 2) for k, v in acc.items()})
for K in (400, 400):
    noise = rng.normal(size=(K, N, 3))
    r0 = mdl.list_rebuilds()
    t0 = time.perf_counter()
    sc, code = mdl.md_run(K, noise)
    dt = time.perf_counter() - t0
    print(f"{K} steps: {1e6 * dt / len(sc):.1f} us/step, rebuilds {mdl.list_rebuilds() - r0}, T {sc[:, 12].mean() / (3 * N * kB):.0f} K, code {code}")
t0 = time.perf_counter(); sc, code = mdl.md_run(400, None); dt = time.perf_counter() - t0
print(f"no noise upload (NVE continuation): {1e6 * dt / len(sc):.1f} us/step")
