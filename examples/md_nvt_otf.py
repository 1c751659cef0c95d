#!/usr/bin/env python3
"""NVT (Langevin) molecular dynamics with on-the-fly SGPR learning on the MI355X — the shape of
BASELINE.json's config 5 (16384 atoms, up to 1024 inducing LCEs, inducing-set
updates inside a 1000-step NVT run) with a stand-in teacher.

    python examples/md_nvt_otf.py --side 32 32 16 --steps 200 --max-inducing 1024

* calculator: autoforce_amd.calculator.ActiveCalculator (the reference's ActiveCalculator surface);
* teacher: a smooth pair potential evaluated in numpy on the device's own neighbour list (a real
  run passes any ASE calculator: VASP, GPAW, ...);
* integrator: BAOAB Langevin with the parameters of the reference's driver (cl/md.py:31,70-74: dt = 1 fs, friction 1e-3,
  T = 600 K; Maxwell-Boltzmann start as util/aseutil.py:11-20) on the device loop (ActiveCalculator.run_md: the state stays
  in HBM between model updates); --host-loop: the same scheme in numpy, one calculate() per step.  With ASE installed,
  ase.md.langevin.Langevin drives the same calculator.
The system is an ordered two-species rocksalt-type lattice (2.72 A nearest-neighbour distance,
rattled): its environments repeat, so the seed set stays small — a random alloy would make every
LCE unique under the reference's 0.95 similarity seed rule (active.py:631-654).
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from autoforce_amd import SGPRModel  # noqa: E402
from autoforce_amd.ase_shim import Atoms, kB  # noqa: E402
from autoforce_amd.calculator import ActiveCalculator  # noqa: E402

FS = 0.09822694788464063  # ase.units.fs: 1 fs in A sqrt(amu/eV)
MASS = {3: 6.94, 8: 15.999, 9: 18.998, 11: 22.99, 12: 24.305, 15: 30.974, 16: 32.06, 17: 35.45, 40: 91.224, 57: 138.905}


class PairTeacher:
    """phi(r) = eps [(1 - e^{-a (r - r0)})^2 - 1] (1 - (r/rc)^2)^2, pairs from the device neighbour list."""
    implemented_properties = ["energy", "forces", "stress", "free_energy"]

    def __init__(self, species, rc=5.0, eps=0.25, a=1.4, r0=2.8, device=0):
        self.rc, self.eps, self.a, self.r0 = rc, eps, a, r0
        self.nl = SGPRModel(3, 3, 4, rc, species=species, device=device)  # used for its neighbour list only
        self.calls, self.seconds, self.results, self._key = 0, 0.0, {}, None

    def calculate(self, atoms):
        t0 = time.time()
        self.calls += 1
        pos = np.asarray(atoms.positions, float)
        cell = np.asarray(getattr(atoms.cell, "array", atoms.cell), float)
        N = len(pos)
        self.nl.predict(atoms.numbers, pos, cell, atoms.pbc, beta=False)
        ptr, j, off = self.nl.neighbors(N)
        i = np.repeat(np.arange(N), np.diff(ptr))
        d = pos[j] - pos[i] + off @ cell
        r = np.linalg.norm(d, axis=1)
        x = np.exp(-self.a * (r - self.r0))
        m_, dm = self.eps * ((1 - x) ** 2 - 1), self.eps * 2 * (1 - x) * self.a * x
        s = 1 - (r / self.rc) ** 2
        phi, dphi = m_ * s * s, dm * s * s - m_ * 4 * s * r / self.rc ** 2
        g = (dphi / r)[:, None] * d
        F = np.zeros_like(pos)
        np.add.at(F, i, g)
        vir = 0.5 * np.einsum("pa,pb->ab", d, g)
        stress = (vir / abs(np.linalg.det(cell)))[[0, 1, 2, 1, 0, 0], [0, 1, 2, 2, 2, 1]]
        self.results = dict(energy=0.5 * phi.sum(), forces=F, stress=stress, free_energy=0.5 * phi.sum())
        self.seconds += time.time() - t0

    def get_property(self, name, atoms=None):
        key = None if atoms is None else atoms.positions.tobytes()
        if atoms is not None and key != self._key:
            self.calculate(atoms)
            self._key = key
        return self.results[name]


def rocksalt(shape, spacing=2.72, sigma=0.05, seed=0, species=(3, 9)):
    rng = np.random.default_rng(seed)
    g = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).reshape(-1, 3)
    numbers = np.where(g.sum(1) % 2 == 0, species[0], species[1]).astype(np.int32)
    pos = g * spacing + sigma * rng.normal(size=(len(g), 3))
    return numbers, pos, np.diag([n * spacing for n in shape]).astype(float), np.array([True] * 3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--side", type=int, nargs=3, default=[8, 8, 8], help="lattice sites per direction (even numbers)")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--temperature", type=float, default=600.0)
    ap.add_argument("--dt", type=float, default=1.0, help="fs")
    ap.add_argument("--friction", type=float, default=1e-3, help="1/(ASE time unit), as cl/md.py")
    ap.add_argument("--max-inducing", type=float, default=1024)
    ap.add_argument("--max-data", type=float, default=10)
    ap.add_argument("--ediff", type=float, default=0.086)
    ap.add_argument("--fdiff", type=float, default=0.129)
    ap.add_argument("--ioptim", type=int, default=1)
    ap.add_argument("--out", default="gpurun_out/md_otf")
    ap.add_argument("--host-loop", action="store_true", help="integrate in numpy, one calculate() per step (the driver of rounds 1-3); "
                    "default: ActiveCalculator.run_md — positions and velocities stay in device memory between model updates")
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    species = [3, 9]
    numbers, pos, cell, pbc = rocksalt(args.side, species=species)
    N = len(numbers)
    teacher = PairTeacher(species)
    calc = ActiveCalculator(calculator=teacher, kernel_kw=dict(species=species), logfile=f"{args.out}/active.log",
                            tape=f"{args.out}/model.sgpr", pckl=f"{args.out}/model.npz", ediff=args.ediff,
                            fdiff=args.fdiff, max_inducing=args.max_inducing, max_data=args.max_data, ioptim=args.ioptim)
    for f in (f"{args.out}/model.sgpr",):
        if os.path.exists(f):
            os.remove(f)
    rng = np.random.default_rng(1)
    np.random.seed(1)  # sample_rand_lces draws from numpy's global generator (as the reference's does, active.py:656-690)
    mass = np.array([MASS[int(z)] for z in numbers])[:, None]
    kT = kB * args.temperature
    vel = rng.normal(size=(N, 3)) * np.sqrt(kT / mass)
    vel -= (mass * vel).sum(0) / mass.sum()
    dt = args.dt * FS
    c1 = np.exp(-args.friction * dt)
    c2 = np.sqrt(1 - c1 * c1)

    def forces(p, v):
        at = Atoms(numbers, p, cell, pbc, velocities=v, masses=mass[:, 0])
        at.calc = calc
        return at.get_forces(), at.get_potential_energy()

    t_all = time.time()
    rows = []
    if not args.host_loop:
        # the same scheme on the device loop: the run halts where a step's largest covloss asks for sampling, that step goes
        # through calculate() (teacher, updates, log), and the run goes on with the new model
        at = Atoms(numbers, pos, cell, pbc, velocities=vel, masses=mass[:, 0])
        tcalls, tsec = teacher.calls, teacher.seconds
        for step, E, T, updated, wall in calc.run_md(at, args.steps, args.temperature, dt_fs=args.dt, friction=args.friction, rng=None,
                                                     chunk=64, seed=1):
            if step == 0:
                print(f"# {N} atoms, seed model {calc.size}, first step {time.time() - t_all:.2f} s (teacher {teacher.seconds:.2f} s)")
            else:
                rows.append((step, wall, teacher.seconds - tsec, teacher.calls - tcalls, calc.size, updated))
                print(f"{step:5d} E={E:14.6f} T={T:7.1f} size={calc.size} wall={wall * 1e3:8.1f} ms "
                      f"teacher={1e3 * (teacher.seconds - tsec):7.1f} ms", flush=True)
            tcalls, tsec = teacher.calls, teacher.seconds
    else:
        F, E = forces(pos, vel)
        print(f"# {N} atoms, seed model {calc.size}, first step {time.time() - t_all:.2f} s (teacher {teacher.seconds:.2f} s)")
    for step in range(1, args.steps + 1 if args.host_loop else 0):
        t0 = time.time()
        tcalls, tsec = teacher.calls, teacher.seconds
        vel += 0.5 * dt * F / mass
        pos = pos + 0.5 * dt * vel
        vel = c1 * vel + c2 * np.sqrt(kT / mass) * rng.normal(size=(N, 3))
        pos = pos + 0.5 * dt * vel
        F, E = forces(pos, vel)
        vel += 0.5 * dt * F / mass
        T = float((mass * vel ** 2).sum() / (3 * N * kB))
        wall = time.time() - t0
        rows.append((step, wall, teacher.seconds - tsec, teacher.calls - tcalls, calc.size, calc.updated))
        print(f"{step:5d} E={E:14.6f} T={T:7.1f} covloss={calc.covlog[:8]:>8s} size={calc.size} "
              f"wall={wall * 1e3:8.1f} ms teacher={1e3 * (teacher.seconds - tsec):7.1f} ms", flush=True)
    total = time.time() - t_all
    quiet = [r[1] for r in rows if r[3] == 0 and not r[5]]
    upd = [r[1] - r[2] for r in rows if r[5]]
    print(f"# {args.steps} steps in {total:.1f} s; teacher calls {teacher.calls} ({teacher.seconds:.1f} s); final size {calc.size}")
    if quiet:
        print(f"# prediction-only steps: {len(quiet)}, median {1e3 * np.median(quiet):.2f} ms/step "
              f"(calculator + host integrator) = {N / np.median(quiet):.3g} atom*steps/s")
    if upd:
        print(f"# model-update steps: {len(upd)}, median {1e3 * np.median(upd):.1f} ms excluding the teacher")


if __name__ == "__main__":
    main()
