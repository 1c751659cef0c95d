"""Machine-learning accelerated relaxation from the command line — theforce/cl/relax.py in this package's terms:

    python -m autoforce_amd.cl.relax -i start.xyz -o relaxed.xyz      # keywords from ./ARGS

The reference minimises with an optimizer of `ase.optimize` (`algo = 'BFGS'` by default, cl/relax.py:50-54) around the
active calculator — the model learns on the way, `calculate()` being what it is inside MD — and then CONFIRMS the minimum
(cl/relax.py:59-70): as long as `update_data(try_fake=False)` accepts the exact labels of the current structure, the model
is updated and the relaxation continues from there.  With ASE installed this driver uses ASE's optimizers (and
`UnitCellFilter` for `cell = True`) exactly as the reference does.  ASE is a dependency of the reference, not part of it,
and is absent from the build image: without it two of its optimizers are restated here from their published algorithms
(ase/optimize/bfgs.py, ase/optimize/fire.py, ASE 3.22) for positions only — `BFGS` (dense Hessian, H0 = 70 eV/A^2, the
step taken through the eigen-decomposition with |omega|, the longest atomic step scaled down to 0.2 A) and `FIRE`.
Structures and the trajectory are extended XYZ."""
import argparse

import numpy as np

from . import gen_active_calc, get_default_args, read_args, update_args
from ..sgprio import Frame, format_extxyz
from .md import read_structure


def force_max(forces):
    """The convergence measure of ase.optimize.Optimizer.converged: the largest atomic force."""
    return float(np.sqrt((np.asarray(forces) ** 2).sum(axis=1).max())) if len(forces) else 0.0


class BFGS:
    """ase/optimize/bfgs.py restated: quasi-Newton on the positions with a dense 3N x 3N Hessian."""

    def __init__(self, atoms, maxstep=0.2, alpha=70.0):
        self.atoms, self.maxstep, self.alpha = atoms, maxstep, alpha
        self.initialize()

    def initialize(self):
        self.H, self.pos0, self.forces0 = None, None, None

    def update(self, pos, forces):
        if self.H is None:
            self.H = np.eye(len(pos)) * self.alpha
            return
        dpos = pos - self.pos0
        if np.abs(dpos).max() < 1e-7:   # (same configuration again: nothing learnt)
            return
        dforces = forces - self.forces0
        a = dpos @ dforces
        dg = self.H @ dpos
        b = dpos @ dg
        self.H -= np.outer(dforces, dforces) / a + np.outer(dg, dg) / b

    def step(self, forces):
        pos = self.atoms.get_positions()
        f = np.asarray(forces, float).reshape(-1)
        self.update(pos.reshape(-1), f)
        omega, V = np.linalg.eigh(self.H)
        dpos = (V @ ((f @ V) / np.fabs(omega))).reshape(-1, 3)
        longest = np.sqrt((dpos ** 2).sum(axis=1)).max()
        if longest >= self.maxstep:
            dpos *= self.maxstep / longest
        self.pos0, self.forces0 = pos.reshape(-1).copy(), f.copy()
        self.atoms.set_positions(pos + dpos)


class FIRE:
    """ase/optimize/fire.py restated (Bitzek et al., PRL 97, 170201): damped dynamics with an adaptive time step."""

    def __init__(self, atoms, dt=0.1, maxstep=0.2, dtmax=1.0, nmin=5, finc=1.1, fdec=0.5, astart=0.1, fa=0.99):
        self.atoms = atoms
        self.dt0, self.maxstep, self.dtmax, self.nmin, self.finc, self.fdec, self.astart, self.fa = dt, maxstep, dtmax, nmin, finc, fdec, astart, fa
        self.initialize()

    def initialize(self):
        self.v, self.dt, self.a, self.nsteps = None, self.dt0, self.astart, 0

    def step(self, forces):
        f = np.asarray(forces, float)
        if self.v is None:
            self.v = np.zeros_like(f)
        else:
            vf = np.vdot(f, self.v)
            if vf > 0.0:
                self.v = (1.0 - self.a) * self.v + self.a * f / np.sqrt(np.vdot(f, f)) * np.sqrt(np.vdot(self.v, self.v))
                if self.nsteps > self.nmin:
                    self.dt = min(self.dt * self.finc, self.dtmax)
                    self.a *= self.fa
                self.nsteps += 1
            else:
                self.v[:] = 0.0
                self.a = self.astart
                self.dt *= self.fdec
                self.nsteps = 0
        self.v += self.dt * f
        dr = self.dt * self.v
        norm = np.sqrt(np.vdot(dr, dr))
        if norm > self.maxstep:
            dr = self.maxstep * dr / norm
        self.atoms.set_positions(self.atoms.get_positions() + dr)


BUILT_IN = {"BFGS": BFGS, "FIRE": FIRE}


class _Runner:
    """`dyn.irun(fmax)` / `dyn.run(fmax)` / `dyn.initialize()` of an ase.optimize optimizer around the built-in steppers."""

    def __init__(self, atoms, algo, trajectory, master, max_steps=100000):
        self.atoms, self.opt, self.master, self.max_steps = atoms, BUILT_IN[algo](atoms), master, max_steps
        self.out = open(trajectory, "w") if (trajectory and master) else None
        self.nsteps = 0

    def initialize(self):
        self.opt.initialize()

    def _dump(self, forces):
        if self.out is not None:
            a = self.atoms
            self.out.writelines(format_extxyz(Frame(a.numbers, a.positions, a.cell, a.pbc, a.calc.results.get("energy"), forces, None)))
            self.out.flush()
        if self.master:
            print(f"{type(self.opt).__name__}: {self.nsteps:4d}  energy {self.atoms.calc.results.get('energy', float('nan')):.6f}  fmax {force_max(forces):.4f}")

    def irun(self, fmax):
        forces = self.atoms.get_forces()
        self._dump(forces)
        yield False
        while force_max(forces) >= fmax and self.nsteps < self.max_steps:
            self.opt.step(forces)
            self.nsteps += 1
            forces = self.atoms.get_forces()
            self._dump(forces)
            yield False
        yield True

    def run(self, fmax):
        for _ in self.irun(fmax):
            pass
        return force_max(self.atoms.get_forces()) < fmax


def _optimizer(atoms, algo, cell, mask, trajectory, master):
    try:
        from ase import optimize
        from ase.constraints import UnitCellFilter
    except ImportError:
        if cell:
            raise NotImplementedError("cell = True is ase.constraints.UnitCellFilter around an ase.optimize optimizer "
                                      "(cl/relax.py:46-49): install ASE; without it positions only")
        if algo not in BUILT_IN:
            raise NotImplementedError(f"algo = '{algo}' is an ase.optimize class: install ASE; without it: {sorted(BUILT_IN)}")
        return _Runner(atoms, algo, trajectory, master)
    filtered = UnitCellFilter(atoms, mask=mask) if cell else atoms          # cl/relax.py:46-49
    return getattr(optimize, algo)(filtered, trajectory=trajectory, master=master)


def relax(atoms, fmax=0.01, cell=False, mask=None, algo="BFGS", trajectory="relax.xyz", rattle=0.02, clear_hist=False, confirm=True,
          calc=None, seed=None):
    """The keywords of theforce/cl/relax.py::relax (same names and defaults; the trajectory is extended XYZ unless ASE
    writes it).  Returns the number of exact (teacher) calculations the run asked for."""
    rng = np.random.default_rng(seed)
    numbers = np.asarray(atoms.numbers)
    calc = gen_active_calc(species=sorted(set(int(z) for z in numbers))) if calc is None else calc
    load1 = calc.size[0]
    master = calc.rank == 0
    if rattle:
        atoms.set_positions(atoms.get_positions() + rng.normal(scale=rattle, size=(len(numbers), 3)))   # atoms.rattle(rattle)
    atoms.calc = calc
    dyn = _optimizer(atoms, algo, cell, mask, trajectory, master)
    for _ in dyn.irun(fmax):
        if calc.updated and clear_hist:
            dyn.initialize()
    load2 = calc.size[0]
    if calc.active and confirm:                                             # cl/relax.py:59-70
        while True:
            load2 += 1
            if calc.update_data(try_fake=False):
                calc.update(data=False)
                calc.results.clear()
                if clear_hist:
                    dyn.initialize()
                dyn.run(fmax=fmax)
            else:
                break
        ml = ("ML", calc.results["energy"], calc.results["forces"])
        exact = ("Ab initio", *calc._test())
        for method, energy, forces in (ml, exact):
            forces = np.asarray(forces)
            if master:
                print(f"\n    relaxation result ({method}):\n    energy:      {energy}\n    force (rms): {np.sqrt(np.mean(forces ** 2))}\n"
                      f"    force (max): {abs(forces).max()}\n")
    if master:
        print(f"\tTotal number of Ab initio calculations: {load2 - load1}\n")
    return load2 - load1


def main(argv=None):
    ap = argparse.ArgumentParser(description="Machine Learning accelerated relaxation")
    ap.add_argument("-i", "--input", default="POSCAR.xyz", help="the initial coordinates of the atoms (extended XYZ)")
    ap.add_argument("-o", "--output", default="CONTCAR.xyz", help="the final coordinates of the atoms (extended XYZ)")
    a = ap.parse_args(argv)
    from ..ase_shim import Atoms
    fr = read_structure(a.input)
    atoms = Atoms(fr.numbers, fr.positions, fr.cell, fr.pbc)
    kwargs = get_default_args(relax)
    kwargs.pop("calc", None)
    update_args(kwargs, read_args())
    relax(atoms, **kwargs)
    with open(a.output, "w") as f:
        f.writelines(format_extxyz(Frame(atoms.numbers, atoms.positions, atoms.cell, atoms.pbc, None, None, None)))


if __name__ == "__main__":
    main()
