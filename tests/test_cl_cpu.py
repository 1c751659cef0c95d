"""The command-line layer (autoforce_amd/cl, after theforce/cl): ARGS parsing with the reference's syntax, keyword
splitting between the calculator and the driver, structure IO, velocity initialisation, and the `md` driver end to end on
the CPU engine (which has no device loop: run_md falls back to the host integrator)."""
import os

import numpy as np
import pytest

import active_common as ac
from autoforce_amd.ase_shim import Atoms, kB
from autoforce_amd.calculator import ActiveCalculator, kcal_mol
from helpers import OracleModel, PairTeacher


def test_args_file_syntax(tmp_path, monkeypatch):
    from autoforce_amd import cl
    monkeypatch.chdir(tmp_path)
    (tmp_path / "ARGS").write_text("# a comment\ncovariance = None   # model\nediff = 2*kcal_mol\ntem = [300., 600.]\n\npicos = -5\n"
                                   "kernel_kw = {'lmax': 3, 'species': [3, 9]}\nmax_inducing = inf\ndynamics = 'Langevin'\n")
    args = cl.read_args()
    assert args["ediff"] == 2 * kcal_mol and args["tem"] == [300.0, 600.0] and args["picos"] == -5
    assert args["kernel_kw"]["species"] == [3, 9] and args["max_inducing"] == float("inf") and args["covariance"] is None
    kw = cl.update_args(cl.get_default_args(ActiveCalculator.__init__), args)
    assert kw["ediff"] == 2 * kcal_mol and "tem" not in kw and kw["max_inducing"] == float("inf")
    from autoforce_amd.cl import md as mdmod
    dk = cl.update_args(cl.get_default_args(mdmod.md), args)
    assert dk["tem"] == [300.0, 600.0] and dk["picos"] == -5 and dk["friction"] == 1e-3 and "ediff" not in dk
    with pytest.raises(Exception):
        (tmp_path / "ARGS").write_text("x = __import__('os').system('true')\n")
        cl.read_args()


def test_velocities_and_structure_io(tmp_path):
    from autoforce_amd.cl.md import init_velocities, read_structure
    from autoforce_amd.sgprio import Frame, format_extxyz
    rng = np.random.default_rng(0)
    numbers = np.array([3, 9] * 200)
    masses = np.where(numbers == 3, 6.94, 18.998)
    v = init_velocities(numbers, masses, 500.0, rng)
    T = (masses[:, None] * v ** 2).sum() / (3 * len(numbers) * kB)
    assert abs(T - 500.0) < 40.0 and np.abs((masses[:, None] * v).sum(0)).max() < 1e-10
    pos = rng.random((400, 3)) * 10
    cell = np.diag([10.0, 11.0, 12.0])
    p = tmp_path / "two.xyz"
    with open(p, "w") as f:
        f.writelines(format_extxyz(Frame(numbers, pos, cell, True, None, None, None)))
        f.writelines(format_extxyz(Frame(numbers, pos + 1.0, cell, True, -3.5, None, None)))
    last = read_structure(str(p))
    np.testing.assert_array_equal(last.numbers, numbers)
    np.testing.assert_allclose(last.positions, pos + 1.0, rtol=0, atol=1e-13)
    np.testing.assert_allclose(read_structure(str(p), 0).positions, pos, rtol=0, atol=1e-13)


def test_md_driver_on_the_cpu_engine(tmp_path, monkeypatch):
    """`md` end to end: an active calculator on the CPU engine, two temperatures, five steps each (picos < 0), a
    trajectory every second step; the log has one line per step and the model has grown."""
    from autoforce_amd.cl.md import md, read_structure
    monkeypatch.chdir(tmp_path)
    np.random.seed(7)
    rng0, numbers, pos, cell = ac.start(0)
    calc = ActiveCalculator(engine=OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=PairTeacher(rc=4.0),
                            logfile="active.log", pckl=None, tape=None, **ac.KW)
    atoms = Atoms(numbers, pos, cell, True)
    md(atoms, calc=calc, dynamics="Langevin", tem=[300.0, 400.0], picos=-5, trajectory="md.xyz", loginterval=2, friction=0.02, seed=3)
    assert calc.size[1] > 2 and calc.step >= 12
    steps = [ln for ln in open("active.log").read().splitlines() if len(ln.split()) >= 6 and ln.split()[2].isdigit()
             and ln.split()[3].lstrip("-").replace(".", "", 1).replace("e-", "", 1).isdigit()]
    assert len(steps) >= 12
    frames = open("md.xyz").read().count("Lattice=")
    assert frames == 2 * 3      # steps 0, 2, 4 of each temperature
    assert read_structure("md.xyz").natoms == len(numbers)
    # the reference's default dynamics: 'NPT' without a bulk modulus = Nose-Hoover NVT (cl/md.py:17, :131-166) — on this
    # engine through the host loop (workloads.nose_hoover_nvt); with a bulk modulus the cell moves: tests/test_npt_cpu.py
    step0 = calc.step
    md(atoms, calc=calc, picos=-4, tem=300.0, tdamp=25, trajectory=None, seed=3)
    assert calc.step >= step0 + 5
    # atoms that carry ASE constraints are not for this loop (neither integrator knows them): said so, not ignored
    atoms.constraints = [object()]
    with pytest.raises(NotImplementedError, match="constraints"):
        next(calc.run_md(atoms, 2, 300.0))
    atoms.constraints = []


def test_built_in_bfgs_and_fire_follow_their_published_rules():
    """The optimizers restated in cl/relax.py for images without ASE.  BFGS on an exactly quadratic surface E = x^T A x / 2:
    the first step is the force over H0 = 70 eV/A^2 (scaled to the 0.2 A limit when longer), the Hessian update is the BFGS
    formula, and the minimum is reached to 1e-8 eV/A in a few more steps than dimensions; FIRE stops every time the power
    F.v turns negative and gets there too."""
    from autoforce_amd.cl.relax import BFGS, FIRE, force_max

    class Quad:
        def __init__(self, A, x):
            self.A, self.x = A, x.copy()
        def get_positions(self):
            return self.x.copy()
        def set_positions(self, p):
            self.x = np.asarray(p, float).copy()
        def forces(self):
            return -(self.A @ self.x.reshape(-1)).reshape(-1, 3)

    rng = np.random.default_rng(3)
    n = 4
    B = rng.normal(size=(3 * n, 3 * n))
    A = B @ B.T / (3 * n) + 20.0 * np.eye(3 * n)
    x0 = 0.05 * rng.normal(size=(n, 3))
    q = Quad(A, x0)
    opt = BFGS(q)
    f0 = q.forces()
    opt.step(f0)
    want = f0 / 70.0
    longest = np.sqrt((want ** 2).sum(1)).max()
    np.testing.assert_allclose(q.x - x0, want * min(1.0, 0.2 / longest), rtol=1e-12, atol=1e-15)
    steps = 1
    while force_max(q.forces()) > 1e-8 and steps < 60:
        opt.step(q.forces())
        steps += 1
    assert force_max(q.forces()) <= 1e-8 and steps <= 40, steps
    # the secant equation of the update: the new Hessian maps the last displacement onto the change of the gradient
    q.set_positions(q.x + 0.01 * rng.normal(size=q.x.shape))
    dpos, dgrad = q.x.reshape(-1) - opt.pos0, -(q.forces().reshape(-1)) + opt.forces0
    opt.update(q.x.reshape(-1), q.forces().reshape(-1))
    np.testing.assert_allclose(opt.H @ dpos, dgrad, rtol=1e-9, atol=1e-12)
    q = Quad(A, x0)
    fire = FIRE(q)
    for k in range(4000):
        if force_max(q.forces()) < 1e-6:
            break
        fire.step(q.forces())
    assert force_max(q.forces()) < 1e-6, k


def test_relax_driver_on_the_cpu_engine(tmp_path, monkeypatch):
    """`relax` end to end on the CPU engine: an active calculator learns while the structure is minimised, the run stops at
    fmax, the confirmation loop (cl/relax.py:59-70) asks the teacher for exact labels until `update_data(try_fake=False)`
    declines, and the relaxed structure's EXACT forces are small too."""
    from autoforce_amd.cl.relax import force_max, relax
    monkeypatch.chdir(tmp_path)
    np.random.seed(11)
    rng0, numbers, pos, cell = ac.start(0)
    teacher = PairTeacher(rc=4.0)
    calc = ActiveCalculator(engine=OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=teacher, logfile="active.log", pckl=None,
                            tape=None, **ac.KW)
    atoms = Atoms(numbers, pos, cell, True)
    n_exact = relax(atoms, fmax=0.1, algo="BFGS", trajectory="relax.xyz", rattle=0.02, calc=calc, seed=5)
    assert n_exact >= 1 and calc.size[0] >= 1
    assert force_max(calc.results["forces"]) < 0.1
    e_exact, f_exact = calc._test()
    assert force_max(f_exact) < 0.35, force_max(f_exact)    # (the model is only as good as ediff / fdiff ask)
    assert open("relax.xyz").read().count("Lattice=") >= 2
    with pytest.raises(NotImplementedError):
        relax(atoms, cell=True, calc=calc)
    with pytest.raises(NotImplementedError):
        relax(atoms, algo="LBFGS", calc=calc)


def test_train_and_test_drivers(tmp_path, monkeypatch):
    """cl/train.py and cl/test.py: a learner's tape and its labelled frames (extended XYZ) train a fresh model through
    include_tape / include_data, with `-r` read as the reference reads it; the test driver evaluates frames without a
    teacher and writes energies and forces back; single_point is the last frame."""
    from autoforce_amd.cl import md as mdmod
    from autoforce_amd.cl import test as testmod
    from autoforce_amd.cl import train as trainmod
    from autoforce_amd.sgprio import Frame, format_extxyz
    monkeypatch.chdir(tmp_path)
    (tmp_path / "src").mkdir()
    calc, teacher, trace = ac.run(OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), tmp_path / "src", steps=4)
    # labelled frames as a trajectory file
    with open("frames.xyz", "w") as f:
        for tr in trace:
            at = tr[5]
            lab = Atoms(at.numbers, at.positions, at.cell, True)
            lab.calc = PairTeacher(rc=4.0)
            f.writelines(format_extxyz(Frame(at.numbers, at.positions, at.cell, at.pbc, lab.get_potential_energy(), lab.get_forces(), lab.get_stress())))
    assert len(mdmod.read_frames("frames.xyz", "::2")) == 2 and len(mdmod.read_frames("frames.xyz", "1")) == 1
    assert len(mdmod.read_frames("frames.xyz", None)) == 1 and len(mdmod.read_frames("frames.xyz", "1:")) == 3
    fresh = ActiveCalculator(engine=OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=None, logfile="train.log", pckl=None,
                             tape=None, **ac.KW)
    fresh._calc = object()   # "active" without a live teacher: the labels come from the files
    trainmod.train(str(tmp_path / "src" / "model.sgpr"), r="1", calc=fresh)
    n_tape = fresh.size
    assert n_tape[0] == 1 and n_tape[1] >= 2
    trainmod.train("frames.xyz", r="::", calc=fresh)
    assert fresh.size[0] >= n_tape[0] and fresh.size[1] >= n_tape[1]
    with pytest.raises(RuntimeError, match="integer"):
        trainmod.train(str(tmp_path / "src" / "model.sgpr"), r="::2", calc=fresh)
    # evaluation only
    probe = ActiveCalculator(covariance=fresh.model, logfile=None)
    res = testmod.test("frames.xyz", r="::2", o="test.xyz", calc=probe)
    assert len(res) == 2 and open("test.xyz").read().count("Lattice=") == 2
    back = mdmod.read_frames("test.xyz", "0")[0]
    np.testing.assert_allclose(back.forces, res[0][1], rtol=0, atol=1e-7)
    e_last, f_last = testmod.single_point("frames.xyz", "sp.xyz", calc=probe)
    assert abs(e_last - mdmod.read_frames("sp.xyz")[0].energy) < 1e-7
    (tmp_path / "ARGS").write_text("calculator = 'PAIR'\n")
    with pytest.raises(RuntimeError, match="calculator = None"):
        testmod.test("frames.xyz")
    # cl/offline.py: the same walk over stored frames with a live teacher — the model learns from what it is unsure of
    from autoforce_amd.cl.offline import offline
    learner = ActiveCalculator(engine=OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=PairTeacher(rc=4.0), logfile=None,
                               pckl=None, tape=None, **ac.KW)
    res = offline("frames.xyz", r="::", o="offline.xyz", calc=learner)
    assert len(res) == 4 and learner.size[0] >= 1 and open("offline.xyz").read().count("Lattice=") == 4
    (tmp_path / "ARGS").write_text("calculator = None\n")
    with pytest.raises(RuntimeError, match="set a calculator"):
        offline("frames.xyz")


def test_init_model_driver(tmp_path, monkeypatch):
    """cl/init_model.py: rattled copies of one structure seed and grow a model; the trajectory holds them with results."""
    from autoforce_amd.cl.init_model import init_model
    from autoforce_amd.cl.md import read_frames
    monkeypatch.chdir(tmp_path)
    np.random.seed(5)
    rng0, numbers, pos, cell = ac.start(0)
    calc = ActiveCalculator(engine=OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=PairTeacher(rc=4.0), logfile=None, pckl=None,
                            tape=None, **ac.KW)
    init_model(Atoms(numbers, pos, cell, True), samples=3, rattle=0.05, trajectory="init.xyz", calc=calc, seed=1)
    assert calc.size[0] >= 1 and calc.size[1] >= 2 and calc.step == 3
    frames = read_frames("init.xyz", "::")
    assert len(frames) == 3 and frames[0].forces is not None and np.abs(frames[0].positions - pos).max() > 1e-3
