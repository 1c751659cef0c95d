#!/usr/bin/env python3
"""The device MD loop alone, for `rocprofv3 --kernel-trace --stats -- python3 tools/md_prof.py`: per-kernel times of MD steps
(every step from the forces of the one before) to set beside those of the resident-frames bench."""
import os, sys
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
v0 = np.random.default_rng(0).normal(size=(N, 3)) * np.sqrt(kB * 600.0 / mass[:, None])
mdl.md_begin(numbers, pos, cell, pbc, mass, v0, dt=FS, friction=1e-3, kT=kB * 600.0, seed=11)
for _ in range(5):
    sc, code = mdl.md_run(400, None)
    print(len(sc), code, mdl.list_rebuilds())
