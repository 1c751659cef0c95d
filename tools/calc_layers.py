"""Wall time of one synchronised step through the layers of the drop-in surface: the bare C call (pointers prepared once),
SGPRModel.predict (numpy conversions + fresh output arrays), ActiveCalculator through atoms.get_forces()."""
import os, sys, time, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from autoforce_amd import _lib
from autoforce_amd.workloads import lips
from autoforce_amd.calculator import ActiveCalculator
from autoforce_amd.ase_shim import Atoms

numbers, pos, cell, pbc = lips(16, seed=0)
mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
N = len(numbers)
rng = np.random.default_rng(0)
lib = _lib.load()
nz, cz, pz = _lib.i32(numbers), _lib.f64(cell), _lib.i32(np.asarray(pbc, bool).astype(np.int32))
F, s, b, E = np.empty((N, 3)), np.zeros(6), np.empty(N), C.c_double(0)
def med(f, n=300):
    t = []
    for _ in range(n):
        x = pos + 0.012 * rng.normal(size=pos.shape)
        t0 = time.perf_counter(); f(x); t.append(time.perf_counter() - t0)
    return np.median(t[20:]) * 1e6
def raw(x):
    lib.sgpr_compute(mdl._h, N, _lib.ptr(nz), _lib.ptr(x), _lib.ptr(cz), _lib.ptr(pz), 0, 1, C.addressof(E), _lib.ptr(F), _lib.ptr(s), _lib.ptr(b), None)
print("bare sgpr_compute            %.1f us" % med(raw))
print("SGPRModel.predict            %.1f us" % med(lambda x: mdl.predict(numbers, x, cell, pbc)))
print("SGPRModel.predict_view       %.1f us" % med(lambda x: mdl.predict_view(numbers, x, cell, pbc)))
calc = ActiveCalculator(covariance=mdl, logfile=None)
atoms = Atoms(numbers, pos.copy(), cell, pbc); atoms.calc = calc
def viacalc(x):
    atoms.positions = x
    atoms.get_forces()
print("atoms.get_forces()           %.1f us" % med(viacalc))
