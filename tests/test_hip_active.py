"""GPU: the inducing-set edit entry points, k(loc, X), and the on-the-fly learner on the HIP
engine — against the reference's fitted states (g8), against the oracle engine driven through the
same control flow, and run-to-run."""
import numpy as np
import pytest

import active_common as ac
from test_hip_parity import load, model_from_fixture

pytestmark = pytest.mark.gpu


def hip_engine():
    from autoforce_amd import SGPRModel
    return SGPRModel(3, 3, 4, 4.5, species=ac.SPECIES)


def test_edit_sequence_against_reference():
    from autoforce_amd import SGPRModel
    g = load("g5_big40")
    ac.check_g8_edit_sequence(SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                        species=g["species"].tolist()))


def test_acceptance_rules_against_reference():
    """g11: the accept / reject decisions of the reference's own add_1inducing / add_1atoms_fast
    (regression/gppotential.py:898-982), with every kernel row, K_mm edit and solve on the device."""
    from autoforce_amd import SGPRModel
    g = load("g5_big40")
    ac.check_g11_acceptance(SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                      species=g["species"].tolist()))


def test_bcm_against_reference():
    """g12: the reference's own committee combination (calculator/active_bcm.py:589-633, forces by autograd
    through the weighted energy) against the device committee."""
    from autoforce_amd import SGPRModel
    g = load("g5_big40")
    ac.check_g12_bcm(lambda: SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                       species=g["species"].tolist()))


@pytest.mark.parametrize("name", ["g5_mixed64", "g5_big40"])
def test_edit_entry_points_equal_rebuild(name):
    """sgpr_add/remove/select_inducing leave the library in the state sgpr_set_inducing builds for
    the same list; sgpr_kernel_columns are columns of sgpr_kernel_rows; sgpr_kernel_local is the
    row of K_mm the LCE would get."""
    g = load(name)
    full = model_from_fixture(g)
    X = list(full.X)
    M = full.M
    rows = full.kernel_rows(g["numbers"], g["positions"], g["cell"], g["pbc"])
    m = len(X)
    inc = full.scratch()
    for x in X[:3]:
        inc.add_inducing(x)
    inc.set_inducing(X[:5])
    for x in X[5:]:
        inc.add_inducing(x)
    np.testing.assert_array_equal(inc.M, M)
    for a, b in zip(inc.kernel_rows(g["numbers"], g["positions"], g["cell"], g["pbc"]), rows):
        np.testing.assert_array_equal(a, b)
    cols = inc.kernel_columns(g["numbers"], g["positions"], g["cell"], g["pbc"], 2, 3)
    np.testing.assert_array_equal(cols[0], rows[0][2:5])
    np.testing.assert_array_equal(cols[1], rows[1][:, 2:5])
    np.testing.assert_array_equal(cols[2], rows[2][:, 2:5])
    # k(loc, X) for a member = its K_mm row; k(loc, loc) = the diagonal
    k, kxx = inc.kernel_local(X[4])
    np.testing.assert_allclose(k, M[4], rtol=1e-12, atol=1e-14)
    assert abs(kxx - M[4, 4]) < 1e-12
    inc.remove_inducing(-1)
    np.testing.assert_array_equal(inc.M, M[:-1, :-1])
    inc.remove_inducing(0)
    np.testing.assert_array_equal(inc.M, M[1:-1, 1:-1])
    keep = [m - 3, 0, 2]
    inc.select_inducing(keep)
    sel = [k + 1 for k in keep]
    np.testing.assert_array_equal(inc.M, M[np.ix_(sel, sel)])
    assert [x is X[i] for x, i in zip(inc.X, sel)] == [True] * 3
    # errors
    from autoforce_amd import Local, SgprError
    with pytest.raises(SgprError):
        inc.remove_inducing(7)
    with pytest.raises(SgprError):
        inc.add_inducing(Local(99, [], np.zeros((0, 3))))
    with pytest.raises(SgprError):
        inc.kernel_columns(g["numbers"], g["positions"], g["cell"], g["pbc"], 2, 5)
    inc.close()
    full.close()


@pytest.mark.parametrize("name", ["g5_mixed64", "g5_big40"])
def test_bordered_cholesky_after_edits(name):
    """sgpr_add_inducing borders the cached Cholesky factor and its inverse (one row of K_mm, O(m^2)),
    sgpr_remove_inducing(-1) deletes the row again: the solves that follow must agree with a model that
    was set up from scratch and refactored (the reference refactors at every edit, gppotential.py:745-791)."""
    g = load(name)
    full = model_from_fixture(g)
    X = list(full.X)
    m = len(X)
    rng = np.random.default_rng(5)
    rows = full.kernel_rows(g["numbers"], g["positions"], g["cell"], g["pbc"])
    K = np.concatenate([rows[0][None], rows[1], rows[2]])
    Y = rng.normal(size=len(K))
    inc = full.scratch()
    inc.set_inducing(X[:m - 3])
    inc.solve(K[:, :m - 3], Y)                       # factor cached: the edits below border it
    for k in range(m - 3, m):
        inc.add_inducing(X[k])
        ref = full.scratch()
        ref.set_inducing(X[:k + 1])
        mu_i, mu_r = inc.solve(K[:, :k + 1], Y), ref.solve(K[:, :k + 1], Y)
        assert inc.ridge == ref.ridge
        np.testing.assert_allclose(inc.choli, ref.choli, rtol=0, atol=1e-9 * np.abs(ref.choli).max())
        pred_i, pred_r = K[:, :k + 1] @ mu_i, K[:, :k + 1] @ mu_r
        np.testing.assert_allclose(pred_i, pred_r, rtol=0, atol=1e-8 * np.abs(pred_r).max())
        ref.close()
    inc.remove_inducing(-1)
    ref = full.scratch()
    ref.set_inducing(X[:m - 1])
    inc.solve(K[:, :m - 1], Y); ref.solve(K[:, :m - 1], Y)
    np.testing.assert_allclose(inc.choli, ref.choli, rtol=0, atol=1e-9 * np.abs(ref.choli).max())
    np.testing.assert_array_equal(inc.M, ref.M)
    # and the evaluator built on the edited model predicts like the one built from scratch
    for mdl in (inc, ref):
        mdl.set_weights(mdl.mu, choli=mdl.choli, vscale=mdl.make_vscale())
    a = inc.predict(g["numbers"], g["positions"], g["cell"], g["pbc"])
    b = ref.predict(g["numbers"], g["positions"], g["cell"], g["pbc"])
    assert abs(a["energy"] - b["energy"]) <= 1e-8 * max(1.0, abs(b["energy"]))
    np.testing.assert_allclose(a["forces"], b["forces"], rtol=0, atol=1e-8 * np.abs(b["forces"]).max())
    np.testing.assert_allclose(a["beta"], b["beta"], rtol=0, atol=1e-6)
    inc.close(); ref.close(); full.close()


def test_learning_loop_matches_oracle_engine(tmp_path):
    """Same scenario, same host logic, two engines: the HIP library and the CPU oracle.  The
    sampled sets must coincide step by step and the predictions agree to solver precision."""
    from helpers import OracleModel
    (tmp_path / "hip").mkdir()
    (tmp_path / "cpu").mkdir()
    c1, t1, tr1 = ac.run(hip_engine(), tmp_path / "hip")
    c2, t2, tr2 = ac.run(OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), tmp_path / "cpu")
    assert [t[0] for t in tr1] == [t[0] for t in tr2]
    assert t1.calls == t2.calls
    for a, b in zip(tr1, tr2):
        assert abs(a[1] - b[1]) < 1e-6
        assert np.abs(a[2] - b[2]).max() < 1e-6
        assert a[4] == b[4]
    for x, y in zip(c1.model.X, c2.model.X):
        assert x.number == y.number
        # same environment; the two neighbour-list builders list it in different orders
        ox, oy = np.lexsort(np.round(x._r, 9).T), np.lexsort(np.round(y._r, 9).T)
        np.testing.assert_allclose(x._r[ox], y._r[oy], rtol=0, atol=1e-12)
        np.testing.assert_array_equal(x._b[ox], y._b[oy])
    assert c1.size[0] >= 2 and c1.size[1] > tr1[0][0][1]


def test_learning_loop_is_reproducible(tmp_path):
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    _, _, tr1 = ac.run(hip_engine(), tmp_path / "a", steps=4)
    _, _, tr2 = ac.run(hip_engine(), tmp_path / "b", steps=4)
    for a, b in zip(tr1, tr2):
        assert a[0] == b[0] and a[1] == b[1]
        np.testing.assert_array_equal(a[2], b[2])


def test_saved_learner_resumes(tmp_path):
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    calc, teacher, trace = ac.run(hip_engine(), tmp_path, steps=3)
    c2 = ActiveCalculator(covariance=str(tmp_path / "model.npz"), logfile=None)
    assert c2.size == calc.size
    at = trace[-1][5]
    probe = Atoms(at.numbers, at.positions, at.cell, True)
    probe.calc = c2
    np.testing.assert_allclose(probe.get_forces(), trace[-1][2], rtol=0, atol=1e-10)


def test_bcm_committee_on_device(tmp_path):
    """Two members trained on different trajectories, combined on the device with the weights of
    active_bcm.py:589-633; agrees with the same committee built on the oracle engine."""
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator_bcm import BCMActiveCalculator
    from helpers import OracleModel
    res = {}
    for tag, make in (("hip", hip_engine), ("cpu", lambda: OracleModel(3, 3, 4, 4.5, species=ac.SPECIES))):
        (tmp_path / (tag + "a")).mkdir()
        (tmp_path / (tag + "b")).mkdir()
        ca, _, tra = ac.run(make(), tmp_path / (tag + "a"), steps=3, tape=False)
        cb, _, _ = ac.run(make(), tmp_path / (tag + "b"), steps=3, seed=3, tape=False)
        at = tra[-1][5]
        bcm = BCMActiveCalculator(covariance=cb.model, kernel_model_dict={"a": ca.model}, logfile=None)
        p = Atoms(at.numbers, at.positions, at.cell, True)
        p.calc = bcm
        res[tag] = (p.get_potential_energy(), p.get_forces(), bcm.get_covloss_total(), bcm.bcm_weights)
    assert abs(res["hip"][0] - res["cpu"][0]) < 1e-6
    assert np.abs(res["hip"][1] - res["cpu"][1]).max() < 1e-6
    np.testing.assert_allclose(res["hip"][2], res["cpu"][2], rtol=0, atol=1e-5)
    for k in res["hip"][3]:
        assert abs(res["hip"][3][k] - res["cpu"][3][k]) < 1e-5


def test_hyperparameter_search_against_the_reference_on_device():
    """g14: the reference's own _regression(optimize=True) (scipy BFGS) against make_munu(algo=3) on the device."""
    from autoforce_amd import SGPRModel
    g = load("g5_big40")
    ac.check_g14_hpo(SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]), species=g["species"].tolist()))


@pytest.mark.parametrize("units", ["metal", "real"])
def test_lammps_fix_external_callback_on_the_device(units):
    """cl/lmp.py:42-71 with the HIP engine behind the calculator: forces in LAMMPS' tag order and units, energy, the
    virial in LAMMPS' component order — against the CPU oracle on the same model."""
    import test_lammps_bridge as tl
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.lammps_bridge import NKTV2P, FixExternalBridge, convert
    from helpers import OracleEngine
    g = load("g5_tric24")
    cell = np.triu(g["cell"])
    mdl = model_from_fixture(g)
    mdl.set_weights(g["mu"], choli=g["choli"])
    calc = ActiveCalculator(covariance=mdl, logfile=None)
    zs = sorted(set(g["numbers"].tolist()))
    types = np.array([zs.index(z) + 1 for z in g["numbers"]])
    lmp = tl.FakeLammps(g["positions"], types, cell, 1.0)
    bridge = FixExternalBridge(lmp, calc, units, {k + 1: z for k, z in enumerate(zs)}, "autoforce")
    N = len(types)
    tag = np.random.default_rng(0).permutation(N) + 1
    fext = np.zeros((N, 3))
    oracle = OracleEngine(g)
    for step, shift in enumerate((0.0, 0.01)):
        lmp.x = g["positions"] + shift
        bridge(None, step, N, tag, None, fext)
        ref = oracle.predict(g["numbers"], g["positions"] + shift, cell, [True] * 3)
        fmax = np.abs(ref["forces"]).max()
        np.testing.assert_allclose(fext, convert(ref["forces"][tag - 1], "force", "ASE", units), rtol=0,
                                   atol=1e-9 * float(convert(fmax, "force", "ASE", units)))
        assert lmp.energy[0] == "autoforce"
        assert abs(lmp.energy[1] - float(convert(ref["energy"], "energy", "ASE", units))) <= 1e-10 * abs(float(convert(1.0, "energy", "ASE", units)))
        vol = abs(np.linalg.det(cell))
        want = -convert(ref["stress"], "pressure", "ASE", units) / (NKTV2P[units] / vol)
        np.testing.assert_allclose(lmp.virial[1], want[[0, 1, 2, 5, 4, 3]], rtol=0, atol=1e-8 * np.abs(want).max())
    mdl.close()


def test_relaxation_driver_on_the_device(tmp_path, monkeypatch):
    """autoforce_amd.cl.relax (theforce/cl/relax.py) around the HIP engine: the model learns while BFGS minimises, the run
    ends below fmax, the confirmation loop ends when update_data(try_fake=False) declines, and the teacher's own forces on
    the relaxed structure are small too.  (The same driver on the CPU engine: tests/test_cl_cpu.py; the two engines under
    the same host logic: test_learning_loop_matches_oracle_engine.)"""
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.cl.relax import force_max, relax
    from helpers import PairTeacher
    monkeypatch.chdir(tmp_path)
    np.random.seed(11)
    rng0, numbers, pos, cell = ac.start(0)
    calc = ActiveCalculator(engine=hip_engine(), calculator=PairTeacher(rc=4.0), logfile=None, pckl=None, tape=None, **ac.KW)
    atoms = Atoms(numbers, pos, cell, True)
    n_exact = relax(atoms, fmax=0.1, algo="BFGS", trajectory="relax.xyz", rattle=0.02, calc=calc, seed=5)
    assert force_max(calc.results["forces"]) < 0.1 and n_exact >= 1 and calc.size[0] >= 1
    e_exact, f_exact = calc._test()
    assert force_max(f_exact) < 0.35, force_max(f_exact)
    assert open("relax.xyz").read().count("Lattice=") >= 2
