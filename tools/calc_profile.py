"""cProfile of ActiveCalculator.calculate() on the bench frame (numpy in / numpy out, one synchronised call per step):
where the host-side time of the SURVEY 8(d) metric goes."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from autoforce_amd.workloads import lips
from autoforce_amd.calculator import ActiveCalculator
from autoforce_amd.ase_shim import Atoms

numbers, pos, cell, pbc = lips(16, seed=0)
mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
calc = ActiveCalculator(covariance=mdl, logfile=None)
atoms = Atoms(numbers, pos.copy(), cell, pbc)
atoms.calc = calc
rng = np.random.default_rng(0)
def step():
    atoms.positions = atoms.positions + 0.012 * rng.normal(size=pos.shape)
    atoms.get_forces()
for _ in range(20): step()
t = []
for _ in range(200):
    t0 = time.perf_counter(); step(); t.append(time.perf_counter() - t0)
print("median step (incl. the position update) %.1f us" % (np.median(t) * 1e6))
pr = cProfile.Profile(); pr.enable()
for _ in range(500): step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
