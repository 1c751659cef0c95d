"""The moving-cell integrator of the command line (autoforce_amd/npt.py: ase.md.npt.NPT restated — theforce/cl/md.py:131-166
runs it around the calculator when a bulk modulus is given).  ASE is absent from the build image, so the restatement is
pinned by what the published scheme guarantees and by its NVT limit:
  * without a barostat it is the Nose-Hoover recurrence of workloads.nose_hoover_nvt (the device loop's host twin);
  * the extended system's conserved quantity (ASE's get_gibbs_free_energy) holds along a run with both thermostat and
    barostat;
  * the cell answers an external pressure, `mask` freezes cell components, `iso` keeps the shape;
  * the rotation that makes the cell upper triangular (util/aseutil.py:61-71) is rigid."""
import numpy as np
import pytest

import active_common as ac
from autoforce_amd.ase_shim import Atoms, kB
from autoforce_amd.npt import GPA, NPT, FilterDeltas, make_cell_upper_triangular
from autoforce_amd.workloads import FS, MASS, nose_hoover_nvt
from helpers import PairTeacher


def _system(seed=0, temperature=300.0, a=2.9):
    rng = np.random.default_rng(seed)
    sites = np.array([[i, j, k] for i in range(3) for j in range(3) for k in range(3)], float) * a
    numbers = np.array(([3, 9] * 14)[:27])
    cell = np.diag([3 * a] * 3)
    pos = sites + 0.05 * rng.normal(size=sites.shape)
    mass = np.array([MASS[int(z)] for z in numbers])
    v = rng.normal(size=pos.shape) * np.sqrt(kB * temperature / mass)[:, None]
    v -= (mass[:, None] * v).sum(0) / mass.sum()
    v -= (mass[:, None] * v).sum(0) / mass[:, None] / len(mass)   # (… and the mean momentum ASE's NPT removes: nothing left)
    return numbers, pos, cell, mass, v


def _atoms(numbers, pos, cell, mass, v, calc):
    at = Atoms(numbers, pos, cell, True, velocities=v, masses=mass)
    at.calc = calc
    return at


def test_upper_triangular_rotation_is_rigid():
    rng = np.random.default_rng(1)
    cell = rng.normal(size=(3, 3)) + 4 * np.eye(3)
    if np.linalg.det(cell) < 0:
        cell[0] *= -1
    pos = rng.random((20, 3)) @ cell
    p2, c2, R = make_cell_upper_triangular(pos, cell)
    assert c2[1, 0] == c2[2, 0] == c2[2, 1] == 0.0 and c2[2, 2] > 0 and c2[1, 1] > 0
    np.testing.assert_allclose(R @ R.T, np.eye(3), atol=1e-14)
    assert abs(np.linalg.det(R) - 1.0) < 1e-14
    np.testing.assert_allclose(np.linalg.det(c2), np.linalg.det(cell), rtol=1e-13)
    d = np.linalg.norm(pos[:, None] - pos[None], axis=2)
    np.testing.assert_allclose(np.linalg.norm(p2[:, None] - p2[None], axis=2), d, atol=1e-12)
    np.testing.assert_allclose(np.linalg.solve(c2.T, p2.T), np.linalg.solve(cell.T, pos.T), atol=1e-12)   # same fractional coordinates
    # a cell that already has the form is left alone (bit for bit)
    p3, c3, R3 = make_cell_upper_triangular(p2, c2)
    np.testing.assert_allclose(c3, c2, atol=1e-13)


def test_without_a_barostat_it_is_the_nose_hoover_twin():
    numbers, pos, cell, mass, v = _system()
    calc = PairTeacher(rc=4.0)
    dyn = NPT(_atoms(numbers, pos, cell, mass, v, calc), 1.0 * FS, 300.0, ttime=25.0 * FS, pfactor=None)
    ref = nose_hoover_nvt(PairTeacher(rc=4.0), numbers, pos, cell, [True] * 3, 30, 300.0, 1.0, 25.0, vel=v)
    for (k, E, T, _), (kr, Er, Tr, _, xr, vr, zeta, zint) in zip(dyn.run(30), ref):
        assert k == kr
        np.testing.assert_allclose(dyn.atoms.positions, xr, rtol=0, atol=1e-10)
        assert abs(E - Er) < 1e-9 and abs(dyn.zeta - zeta) < 1e-12 and abs(dyn.zeta_integrated - zint) < 1e-12
        if k:
            np.testing.assert_allclose(dyn.atoms.get_velocities(), vr, rtol=0, atol=1e-10)
        np.testing.assert_array_equal(np.asarray(dyn.atoms.cell), cell)


@pytest.mark.parametrize("iso", [False, True])
def test_the_extended_energy_is_conserved_and_the_cell_answers_pressure(iso):
    numbers, pos, cell, mass, v = _system(temperature=300.0, a=2.15)      # (near the teacher's equilibrium volume)
    runs = {}
    for P, dt in ((0.0, 1.0), (3.0, 1.0), (0.0, 0.5)):
        at = _atoms(numbers, pos, cell, mass, v, PairTeacher(rc=4.0))
        dyn = NPT(at, dt * FS, 300.0, externalstress=P * GPA, ttime=25.0 * FS, pfactor=(75.0 * FS) ** 2 * 40.0 * GPA)
        if iso:
            dyn.set_fraction_traceless(0.0)
        G, V, H = [], [], []
        for k, E, T, _ in dyn.run(int(600 / dt)):
            G.append(dyn.get_gibbs_free_energy())
            V.append(at.get_volume())
            H.append(E + dyn.kinetic_energy())
        runs[P, dt] = (np.ptp(G), np.mean(V[int(200 / dt):]))
        # thermostat and barostat move the particles' energy by electron volts; the extended system's conserved quantity
        # (ASE's get_gibbs_free_energy) fluctuates by less than a hundredth of that and does not drift
        assert np.ptp(G) < 0.012 * np.ptp(H), (np.ptp(G), np.ptp(H))
        assert abs(G[-1] - G[0]) < 0.25 * np.ptp(G) + 1e-3
        c = np.asarray(at.cell)
        assert c[1, 0] == c[2, 0] == c[2, 1] == 0.0
        if iso:   # the shape is kept: h stays a multiple of the start cell
            np.testing.assert_allclose(c / c[0, 0], cell / cell[0, 0], atol=1e-12)
        else:
            assert abs(c[0, 0] - c[1, 1]) > 1e-6 and abs(c[0, 1]) > 1e-8      # (thermal noise moves the components apart)
    assert runs[3.0, 1.0][1] < runs[0.0, 1.0][1] - 5.0           # 3 GPa squeeze the cell (A^3, of 270)
    # the fluctuation of the conserved quantity is the integrator's: first order in the time step (centred momenta, staggered
    # strain rate), so half the step halves it
    assert 0.35 < runs[0.0, 0.5][0] / runs[0.0, 1.0][0] < 0.65, runs


def test_mask_freezes_cell_components():
    numbers, pos, cell, mass, v = _system()
    at = _atoms(numbers, pos, cell, mass, v, PairTeacher(rc=4.0))
    dyn = NPT(at, 1.0 * FS, 300.0, externalstress=1.0 * GPA, ttime=25.0 * FS, pfactor=(75.0 * FS) ** 2 * 40.0 * GPA, mask=(0, 0, 1))
    for _ in dyn.run(60):
        pass
    c = np.asarray(at.cell)
    assert c[2, 2] != cell[2, 2]
    c2 = c.copy()
    c2[2, 2] = cell[2, 2]
    np.testing.assert_array_equal(c2, cell)


def test_filter_deltas_spreads_a_model_update():
    """calculator/active.py:46-73: the jump of an update is subtracted at once and handed back over the following calls."""
    class Calc:
        deltas = None

    class A:
        calc = Calc()
        f = np.zeros((2, 3))

        def get_forces(self):
            return self.f.copy()

        def get_stress(self):
            return np.zeros(6)

    a = A()
    fa = FilterDeltas(a, shrink=0.5)
    np.testing.assert_array_equal(fa.get_forces(), 0.0)
    a.f = a.f + 0.4                                  # the model changed: every force jumps by 0.4 ...
    a.calc.deltas = dict(forces=np.full((2, 3), 0.4), stress=np.zeros(6), energy=0.0)
    np.testing.assert_allclose(fa.get_forces(), 0.4 - 0.2)      # ... of which half is held back,
    a.calc.deltas = None
    np.testing.assert_allclose(fa.get_forces(), 0.4 - 0.1)      # a quarter, ...
    np.testing.assert_allclose(fa.get_forces(), 0.4 - 0.05)
    a.calc.deltas = dict(forces=np.full((2, 3), 10.0), stress=np.zeros(6), energy=0.0)
    np.testing.assert_allclose(fa.get_forces(), 0.4 - 1.0)      # clamped to 1 eV/A
    assert len(fa.get_stress()) == 6 and fa.calc is a.calc


def test_md_driver_with_a_bulk_modulus_moves_the_cell(tmp_path, monkeypatch):
    """`md(dynamics='NPT', bulk_modulus=…)` end to end on the CPU engine: manual expand / shrink steps, the cell made upper
    triangular, one calculate() — one log line — per step, a trajectory whose lattice changes."""
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.cl.md import md, read_frames
    from helpers import OracleModel
    monkeypatch.chdir(tmp_path)
    np.random.seed(7)
    rng0, numbers, pos, cell = ac.start(0)
    calc = ActiveCalculator(engine=OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=PairTeacher(rc=4.0),
                            logfile="active.log", pckl=None, tape=None, **ac.KW)
    atoms = Atoms(numbers, pos, cell, True)
    md(atoms, calc=calc, dynamics="NPT", bulk_modulus=30.0, stress=0.5, tem=300.0, picos=-8, trajectory="npt.xyz", loginterval=2,
       tdamp=25, pdamp=100, seed=3, iso=True)
    log = open("active.log").read()
    assert "expand: 1.05*cell" in log and "shrink: 0.95*cell" in log
    assert calc.step >= 9 and calc.size[1] > 2
    frames = read_frames("npt.xyz", ":")
    assert len(frames) == 5
    cells = np.array([f.cell for f in frames])
    assert np.abs(cells[-1] - cells[0]).max() > 1e-6 and not np.array_equal(np.asarray(atoms.cell), cell)
    np.testing.assert_allclose(np.asarray(atoms.cell) / np.asarray(atoms.cell)[0, 0], cell / cell[0, 0], atol=1e-12)   # iso
