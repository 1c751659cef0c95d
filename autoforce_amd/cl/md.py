"""Machine-learning molecular dynamics from the command line — theforce/cl/md.py in this package's terms:

    python -m autoforce_amd.cl.md -i start.xyz -o final.xyz        # keywords from ./ARGS

Langevin dynamics (the reference's `dynamics = 'Langevin'`, cl/md.py:117-128) and the reference's DEFAULT — `dynamics =
'NPT'` without a bulk modulus: Nose-Hoover NVT, ase.md.npt.NPT(pfactor=None, ttime=tdamp fs), cl/md.py:17, :131-166 — run with
positions and velocities in device memory (ActiveCalculator.run_md); NPT with a bulk modulus (a cell that moves, cl/md.py:147-166)
is the same ASE integrator restated in `autoforce_amd/npt.py` (Melchionna's combined Nose-Hoover / Parrinello-Rahman scheme,
`mask`, `iso`, `stress`, `pdamp`, the `ml_filter` wrapper) driving calculate() once per step, as ASE's object does in the
reference.  Structures are read and written as extended XYZ (ASE's own format; without ASE no other reader exists
here), the trajectory likewise (`trajectory = 'md.xyz'`)."""
import argparse

import numpy as np

from . import gen_active_calc, get_default_args, read_args, update_args
from ..ase_shim import kB
from ..sgprio import Frame, format_extxyz, parse_extxyz
from ..workloads import MASS


def read_frames(path, r=None):
    """The frames of an extended-XYZ file that ase.io.read(path, r) would return: r = None or an index: that ONE frame (None:
    the last), r = 'start:stop:step': the slice."""
    frames = _all_frames(path)
    if r is None:
        return [frames[-1]]
    r = str(r)
    if ":" not in r:
        return [frames[int(r)]]
    parts = [int(p) if p.strip() else None for p in (r.split(":") + ["", ""])[:3]]
    return frames[slice(*parts)]


def read_structure(path, index=-1):
    """Frame `index` of an extended-XYZ file."""
    return _all_frames(path)[index]


def _all_frames(path):
    lines = open(path).read().splitlines()
    frames, k = [], 0
    while k < len(lines):
        if not lines[k].strip():
            k += 1
            continue
        n = int(lines[k].split()[0])
        frames.append(parse_extxyz(lines[k:k + n + 2]))
        k += n + 2
    if not frames:
        raise ValueError(f"{path}: no frames")
    return frames


def init_velocities(numbers, masses, temperature, rng, cm0=True):
    """util/aseutil.py:11-20: Maxwell-Boltzmann momenta (ase.md.velocitydistribution: p = xi sqrt(m kB T)), then the
    centre-of-mass momentum removed (Stationary).  (ZeroRotation is left out for periodic cells, where it has no meaning.)"""
    xi = rng.standard_normal((len(numbers), 3))
    p = xi * np.sqrt(masses * kB * temperature)[:, None]
    if cm0:
        p -= p.sum(0) * (masses / masses.sum())[:, None]
    return p / masses[:, None]


def _scale_cell(atoms, cell, factor):
    """atoms.set_cell(factor * cell, scale_atoms=True): the atoms keep their fractional coordinates."""
    old = np.array(getattr(atoms.cell, "array", atoms.cell), float)
    frac = np.linalg.solve(old.T, np.asarray(atoms.positions, float).T).T
    atoms.cell = factor * np.asarray(cell, float)
    atoms.positions = frac @ (factor * np.asarray(cell, float))


def manual_steps(atoms, calc, eps, rng, eps2=0.0, npt=False):
    """cl/md.py:175-196: a rattled copy — and, before a run whose cell moves, an expanded and a shrunk copy — is shown to an
    active calculator before the run."""
    calc._logpref = "#"
    calc.log("manual steps:")
    calc.log(f"rattle: {eps}")
    saved = atoms.positions.copy()
    atoms.calc = calc
    if eps > 0.0:
        atoms.positions = saved + rng.normal(scale=eps, size=saved.shape)
        atoms.get_potential_energy()
    if npt and eps2 > 0.0:
        cell = np.array(getattr(atoms.cell, "array", atoms.cell), float)
        calc.log(f"expand: {(1. + eps2)}*cell")
        _scale_cell(atoms, cell, 1.0 + eps2)
        atoms.get_potential_energy()
        calc.log(f"shrink: {(1. - eps2)}*cell")
        _scale_cell(atoms, cell, 1.0 - eps2)
        atoms.get_potential_energy()
        _scale_cell(atoms, cell, 1.0)
    atoms.positions = saved
    calc._logpref = ""


def npt_dynamics(atoms, calc, dt, tem, bulk_modulus, stress, mask, iso, tdamp, pdamp, ml_filter):
    """cl/md.py:131-166 with a bulk modulus: ase.md.npt.NPT(atoms, dt fs, temperature_K, externalstress = stress GPa, ttime =
    tdamp fs, pfactor = (pdamp fs)^2 * bulk_modulus GPa, mask) on the upper-triangular cell (`configure_cell`, :169-172;
    an all-zero cell would need ASE's `center(vacuum=6)`: a periodic cell is asked for here), `iso` = no traceless strain."""
    from ..npt import GPA, NPT, FilterDeltas, make_cell_upper_triangular
    from ..workloads import FS
    cell = np.array(getattr(atoms.cell, "array", atoms.cell), float)
    if np.allclose(cell, 0.0) or abs(np.linalg.det(cell)) < 1e-12:
        raise ValueError("NPT needs a three-dimensional cell (the reference puts a cluster into a box with 6 A of vacuum around it, "
                         "cl/md.py:169-172: give the box; a slab needs a finite third vector)")
    v = atoms.get_velocities()
    pos, cell_ut, R = make_cell_upper_triangular(atoms.positions, cell)
    if not np.array_equal(cell_ut, cell):
        atoms.cell = cell_ut
        atoms.positions = pos
        if v is not None:
            atoms.set_velocities(np.asarray(v) @ R)
    md_atoms = FilterDeltas(atoms, shrink=ml_filter) if ml_filter else atoms
    dyn = NPT(md_atoms, dt * FS, tem, externalstress=stress * GPA, ttime=tdamp * FS, pfactor=(pdamp * FS) ** 2 * bulk_modulus * GPA,
              mask=mask)
    if iso:
        dyn.set_fraction_traceless(0.0)
    return dyn


def md(atoms, calc=None, dynamics="NPT", dt=None, tem=300.0, picos=100, trajectory="md.xyz", loginterval=1, append=False,
       rattle=0.0, friction=1e-3, eps_pos=0.05, seed=None, bulk_modulus=None, stress=0.0, mask=None, iso=False, tdamp=25, pdamp=100,
       ml_filter=0.8, eps_cell=0.05):
    """The keywords of theforce/cl/md.py::md (same names, same defaults: `dynamics = 'NPT'` with `bulk_modulus = None` is
    Nose-Hoover NVT with the damping time `tdamp` fs).
    picos > 0: pico-seconds per temperature; picos < 0: -picos steps (cl/md.py:100)."""
    rng = np.random.default_rng(seed)
    numbers = np.asarray(atoms.numbers)
    calc = gen_active_calc(species=sorted(set(int(z) for z in numbers))) if calc is None else calc
    atoms.calc = calc
    if calc.active:
        manual_steps(atoms, calc, eps_pos, rng, eps_cell, npt=bool(bulk_modulus))
    if rattle:
        atoms.positions = atoms.positions + rng.normal(scale=rattle, size=atoms.positions.shape)
    temperatures = list(tem) if hasattr(tem, "__iter__") else [tem]
    if calc.rank == 0:
        print(f"MD temperatures: {temperatures}")
    if getattr(atoms, "_masses", "ase") is None:
        atoms._masses = np.array([MASS[int(z)] for z in numbers])
    masses = np.asarray(atoms.get_masses(), float)
    v = atoms.get_velocities()
    if v is None or np.allclose(v, 0.0):
        atoms.set_velocities(init_velocities(numbers, masses, temperatures[0], rng))
    if dt is None:
        dt = 0.25 if (numbers == 1).any() else 1.0          # cl/md.py:70-74
    if dynamics.upper() not in ("LANGEVIN", "NPT"):
        raise ValueError(f"dynamics = {dynamics!r}: 'NPT' or 'Langevin' (cl/md.py:84-101)")
    moving_cell = dynamics.upper() == "NPT" and bool(bulk_modulus)
    tdamp_fs = float(tdamp) if dynamics.upper() == "NPT" else None
    out = open(trajectory, "a" if append else "w") if (trajectory and calc.rank == 0) else None
    for T in temperatures:
        steps = int(picos * 1000 / dt) if picos > 0 else int(-picos)
        if moving_cell:
            # the barostat's integrator around calculate(), one call per step (the log line, the covloss gate and the model
            # updates are calculate()'s own, as in the reference)
            dyn = npt_dynamics(atoms, calc, dt, T, bulk_modulus, stress, mask, iso, tdamp, pdamp, ml_filter)
            for step, energy, temperature, wall in dyn.run(steps):
                if out is not None and loginterval and step % loginterval == 0:
                    out.writelines(format_extxyz(Frame(numbers, atoms.positions, atoms.cell, atoms.pbc, energy, None, None)))
            continue
        for step, energy, temperature, updated, wall in calc.run_md(atoms, steps, T, dt_fs=dt, friction=friction, rng=None,
                                                                    seed=int(rng.integers(1, 2 ** 62)), sync_every=loginterval or None,
                                                                    tdamp_fs=tdamp_fs):
            if out is not None and loginterval and step % loginterval == 0:
                # (the state lives on the device: run_md brings the positions back at the steps a trajectory wants them)
                out.writelines(format_extxyz(Frame(numbers, atoms.positions, atoms.cell, atoms.pbc, energy, None, None)))
    if out is not None:
        out.close()
    return atoms


def main(argv=None):
    ap = argparse.ArgumentParser(description="Machine Learning Molecular Dynamics (MLMD)")
    ap.add_argument("-i", "--input", default="POSCAR.xyz", help="the initial coordinates of the atoms (extended XYZ)")
    ap.add_argument("-o", "--output", default="CONTCAR.xyz", help="the final coordinates of the atoms (extended XYZ)")
    a = ap.parse_args(argv)
    from ..ase_shim import Atoms
    fr = read_structure(a.input)
    atoms = Atoms(fr.numbers, fr.positions, fr.cell, fr.pbc)
    kwargs = get_default_args(md)
    kwargs.pop("calc", None)
    update_args(kwargs, read_args())
    md(atoms, **kwargs)
    with open(a.output, "w") as f:
        f.writelines(format_extxyz(Frame(atoms.numbers, atoms.positions, atoms.cell, atoms.pbc, None, None, None)))


if __name__ == "__main__":
    main()
