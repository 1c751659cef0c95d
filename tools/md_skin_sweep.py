#!/usr/bin/env python3
"""Device MD loop (LiPS 4096 / 512, fitted weights, 600 K) against the Verlet skin of the neighbour candidates."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from autoforce_amd import _lib
from autoforce_amd.ase_shim import kB
from autoforce_amd.workloads import FS, MASS, fit_to_teacher, lips
numbers, pos, cell, pbc = lips(16, seed=0)
N = len(numbers)
mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
fit_to_teacher(mdl, numbers, pos, cell, pbc)
mass = np.array([MASS[int(z)] for z in numbers])
v0 = np.random.default_rng(0).normal(size=(N, 3)) * np.sqrt(kB * 600.0 / mass[:, None])
lib = _lib.load()
for skin in (400, 500, 600, 700, 800, 1000):
    _lib.check(lib.sgpr_set_option(mdl.handle, b"skin_milliangstrom", skin))
    mdl.md_begin(numbers, pos, cell, pbc, mass, v0, dt=FS, friction=float(os.environ.get("SWEEP_FRICTION", "0.02")), kT=kB * 600.0, seed=5)
    mdl.md_run(300, None)
    best = None
    for rep in range(3):
        r0 = mdl.list_rebuilds(); t0 = time.perf_counter()
        sc, code = mdl.md_run(400, None)
        dt = time.perf_counter() - t0
        us = 1e6 * dt / len(sc)
        best = us if best is None else min(best, us)
    print(f"skin {skin / 1000:.2f} A: {best:.1f} us/step, rebuilds {mdl.list_rebuilds() - r0} per 400 steps, T {sc[:, 12].mean() / (3 * N * kB):.0f} K, dims {mdl.dims['nn_max']}", flush=True)
