"""Host logic of the on-the-fly learner driven through the CPU
oracle engine: PosteriorPotential bookkeeping, acceptance rules, tape and model files.
The same scenario runs on the HIP engine in tests/test_hip_active.py."""
import os

import numpy as np
import pytest

import active_common as ac
from autoforce_amd.ase_shim import Atoms
from autoforce_amd.calculator import ActiveCalculator, Switch
from autoforce_amd.posterior import Frame, PosteriorPotential
from autoforce_amd.sgprio import SgprIO
from helpers import OracleEngine, OracleModel, PairTeacher, load


def engine():
    return OracleModel(3, 3, 4, 4.5, species=ac.SPECIES)


def test_learning_loop(tmp_path):
    calc, teacher, trace = ac.run(engine(), tmp_path)
    sizes = [t[0] for t in trace]
    assert sizes[0][0] == 1 and sizes[0][1] >= 2          # seeded from the first frame
    assert sizes[-1][1] > sizes[0][1] and sizes[-1][0] >= 2  # LCEs and data were sampled later on
    assert all(a[1] <= b[1] for a, b in zip(sizes, sizes[1:]))
    # the model tracks the teacher on the frames it has just seen
    at = trace[-1][5]
    ref = Atoms(at.numbers, at.positions, at.cell, True)
    ref.calc = teacher
    assert np.abs(trace[-1][2] - ref.get_forces()).max() < 0.3
    assert abs(trace[-1][1] - ref.get_potential_energy()) < 0.05
    # every step's covloss ended below the sampling threshold (that is what update_inducing enforces)
    assert all(float(t[3]) < 1.5 * ac.KW["ediff"] for t in trace)
    # deltas are reported exactly on the steps (> 0) where the model changed
    changed = [b[0] != a[0] for a, b in zip(trace, trace[1:])]
    assert [t[4] for t in trace[1:]] == changed
    log = open(tmp_path / "active.log").read()
    for token in ("seed size:", "added indu:", "added data:", "fit error (mean,mae):", "exact energy:", "DF:"):
        assert token in log
    # results stay a fixed point: asking again does not call the engine or the teacher
    n = teacher.calls
    trace[-1][5].get_forces()
    assert teacher.calls == n


def test_wildcard_species_table_grows_on_demand(tmp_path):
    """kernel_kw without `species` = the reference's default wildcard kernel (active.py:28-38): the dense
    table is laid out from the species met; the run is the fixed-table run."""
    (tmp_path / "a").mkdir(); (tmp_path / "b").mkdir()
    _, _, fixed = ac.run(engine(), tmp_path / "a", steps=4)
    calc, _, wild = ac.run(OracleModel(3, 3, 4, 4.5, species=[0]), tmp_path / "b", steps=4, wildcard=True)
    assert sorted(calc.engine.species) == ac.SPECIES
    assert [t[0] for t in wild] == [t[0] for t in fixed]
    for a, b in zip(wild, fixed):
        assert abs(a[1] - b[1]) <= 1e-10 * max(1.0, abs(b[1]))
        np.testing.assert_allclose(a[2], b[2], rtol=0, atol=1e-9)
    assert "species table -> [3, 9]" in open(tmp_path / "b" / "active.log").read()


def test_tape_replay_and_build(tmp_path):
    calc, teacher, trace = ac.run(engine(), tmp_path)
    blocks = SgprIO(str(tmp_path / "model.sgpr")).read()
    kinds = [k for k, _ in blocks]
    assert kinds.count("atoms") == calc.size[0] and kinds.count("local") >= calc.size[1]
    assert kinds[0] == "atoms"  # the seed frame comes first (active.py:618-623)
    # frames on the tape carry the teacher's labels
    fr = next(o for k, o in blocks if k == "atoms")
    at = Atoms(fr.numbers, fr.positions, fr.cell, fr.pbc)
    at.calc = PairTeacher(rc=4.0)
    assert abs(at.get_potential_energy() - fr.energy) < 1e-9
    assert np.abs(at.get_forces() - fr.forces).max() < 1e-9
    # 1. build(): one-shot model from the calculator's own tape
    sub = tmp_path / "b"
    sub.mkdir()
    os.link(tmp_path / "model.sgpr", sub / "model.sgpr")
    c2 = ActiveCalculator(engine=engine(), logfile=str(sub / "active.log"), tape=str(sub / "model.sgpr"),
                          pckl=str(sub / "model.npz"), **ac.KW)
    c2.build()
    assert c2.size == (kinds.count("atoms"), kinds.count("local"))
    last = trace[-1][5]
    probe = Atoms(last.numbers, last.positions, last.cell, True)
    probe.calc = c2
    assert np.abs(probe.get_forces() - trace[-1][2]).max() < 0.35  # same data, different fit history
    with pytest.raises(RuntimeError, match="already exists"):
        c2.build()
    # 2. include_tape(): replay through the acceptance rules of a fresh learner
    sub3 = tmp_path / "c"
    sub3.mkdir()
    c3 = ActiveCalculator(engine=engine(), calculator=None, logfile=str(sub3 / "active.log"),
                          tape=str(sub3 / "own.sgpr"), pckl=None, **ac.KW)
    c3._calc = object()  # "active" without a live teacher: labels come from the tape
    c3.include_tape(str(tmp_path / "model.sgpr"))
    assert c3.size[0] >= 1 and 2 <= c3.size[1] <= kinds.count("local")
    with pytest.raises(RuntimeError, match="own .sgpr tape"):
        c3.include_tape(str(sub3 / "own.sgpr"))


def test_model_file_roundtrip(tmp_path):
    from autoforce_amd.modelio import load_model
    calc, teacher, trace = ac.run(engine(), tmp_path, steps=3, tape=False)
    post = load_model(str(tmp_path / "model.npz"), engine=engine())
    assert (post.ndata, len(post.X)) == calc.size
    np.testing.assert_allclose(post.Kf, calc.model.Kf, rtol=0, atol=1e-12)
    np.testing.assert_array_equal(post.mu, calc.model.mu)
    assert post.mean.weights == calc.model.mean.weights
    c2 = ActiveCalculator(covariance=post, logfile=None)
    at = trace[-1][5]
    probe = Atoms(at.numbers, at.positions, at.cell, True)
    probe.calc = c2
    np.testing.assert_allclose(probe.get_forces(), trace[-1][2], rtol=0, atol=1e-10)
    assert abs(probe.get_potential_energy() - trace[-1][1]) < 1e-10


def _frames(n, seed=5):
    rng, numbers, pos, cell = ac.start(seed)
    teacher = PairTeacher(rc=4.0)
    out = []
    for _ in range(n):
        pos = pos + 0.08 * rng.normal(size=pos.shape)
        at = Atoms(numbers, pos, cell, True)
        at.calc = teacher
        out.append(Frame(numbers, pos, cell, True, at.get_potential_energy(), at.get_forces(), at.get_stress()))
    return out


def _locals(fr, idx, rc=4.5):
    from oracle import oracle as orc
    from autoforce_amd.model import Local
    ptr, j, off = orc.neighbors(fr.positions, fr.cell, fr.pbc, rc)
    out = []
    for k in idx:
        a, b = ptr[k], ptr[k + 1]
        out.append(Local(fr.numbers[k], fr.numbers[j[a:b]], fr.positions[j[a:b]] - fr.positions[k] + off[a:b] @ fr.cell))
    return out


def test_posterior_edits_equal_rebuilds():
    """add/pop/select/downsize keep (Ke, Kf, Kv, M, mu) identical to a model built from scratch on
    the same data and inducing set (gppotential.py:730-842)."""
    frames = _frames(3)
    locs = _locals(frames[0], range(0, 18, 2))

    def fresh(data, X):
        p = PosteriorPotential(engine())
        p.set_data(data, X)
        return p

    def same(p, q):
        for name in ("Ke", "Kf", "Kv"):
            np.testing.assert_allclose(getattr(p, name), getattr(q, name), rtol=0, atol=1e-12)
        np.testing.assert_allclose(p.M, q.M, rtol=0, atol=1e-13)
        np.testing.assert_allclose(p.K @ p.mu, q.K @ q.mu, rtol=0, atol=1e-8)

    p = fresh(frames[:1], locs[:5])
    p.add_inducing(locs[5])
    p.add_data([frames[1]])
    p.add_inducing(locs[6])
    same(p, fresh(frames[:2], locs[:7]))
    p.pop_1inducing()
    p.popfirst_1inducing()
    same(p, fresh(frames[:2], locs[1:6]))
    p.add_data([frames[2]])
    p.popfirst_1data()
    same(p, fresh(frames[1:3], locs[1:6]))
    p.select_inducing([4, 0, 2])
    same(p, fresh(frames[1:3], [locs[5], locs[1], locs[3]]))
    # downsize(lii): keeps the m LCEs with the smallest K_mm row sums, oldest data go first
    p = fresh(frames, locs)
    order = np.argsort(p.M.sum(axis=1), kind="stable")[:4].tolist()
    ch1, ch2 = p.downsize(2, 4, first=True, lii=True)
    assert ch1 == 1 and ch2 == order
    same(p, fresh(frames[1:], [locs[i] for i in order]))


def test_acceptance_rules():
    frames = _frames(3)
    locs = _locals(frames[0], range(18))
    p = PosteriorPotential(engine())
    p.set_data(frames[:1], locs[:4])
    # an LCE already in the set changes nothing -> refused (and the duplicate makes K_mm singular)
    assert p.add_1inducing(locs[0], 1e-3)[0] == 0 and len(p.X) == 4
    # a far threshold refuses, a zero threshold accepts
    assert p.add_1inducing(locs[9], 1e3)[0] == 0 and len(p.X) == 4
    added, de = p.add_1inducing(locs[9], 0.0)
    assert added == 1 and de > 0 and len(p.X) == 5
    # data: the frame it was fitted on adds nothing; a new frame with generous thresholds is refused,
    # with tight ones accepted
    assert p.add_1atoms_fast(frames[0], 1e-4, 1e-4)[0] == 0 and p.ndata == 1
    assert p.add_1atoms_fast(frames[2], 1e3, 1e3)[0] == 0 and p.ndata == 1
    assert p.add_1atoms_fast(frames[2], 1e-6, 1e-6)[0] == 1 and p.ndata == 2
    # leakage of an inducing LCE is ~0, of a new one positive
    assert abs(p.leakage(locs[0])) < 1e-6 and p.leakage(locs[13]) > 1e-4


def test_rejected_trials_restore_what_a_refit_would_give():
    """A rejected add_1inducing / add_1atoms_fast hands back the fitted state saved before the trial; the reference
    refits instead (gppotential.py:898-982) — for exactly the same model.  Both must leave the same posterior."""
    frames = _frames(3)
    locs = _locals(frames[0], range(18))

    class NoRestore:  # the same engine without the shortcut: PosteriorPotential falls back to pop + refit
        def __init__(self, eng):
            self._e = eng

        def __getattr__(self, name):
            if name in ("restore_weights", "snapshot_weights"):
                raise AttributeError(name)
            return getattr(self._e, name)

    posts = []
    for wrap in (lambda e: e, NoRestore):
        p = PosteriorPotential(wrap(engine()))
        p.set_data(frames[:2], locs[:6])
        assert p.add_1inducing(locs[9], 1e3)[0] == 0          # rejected inducing trial
        assert p.add_1atoms_fast(frames[2], 1e3, 1e3)[0] == 0  # rejected data trial
        posts.append(p)
    a, b = posts
    assert len(a.X) == len(b.X) == 6 and a.ndata == b.ndata == 2
    np.testing.assert_allclose(a.mu, b.mu, rtol=0, atol=1e-9 * np.abs(b.mu).max())
    np.testing.assert_allclose(a.choli, b.choli, rtol=0, atol=1e-9 * np.abs(b.choli).max())
    np.testing.assert_allclose(a._stats, b._stats, rtol=1e-8, atol=1e-12)
    assert a.mean.weights.keys() == b.mean.weights.keys()
    for z in a.mean.weights:
        assert abs(a.mean.weights[z] - b.mean.weights[z]) < 1e-10
    assert a.engine._vscale.keys() == b.engine._vscale.keys()
    for z in a.engine._vscale:
        assert abs(a.engine._vscale[z] - b.engine._vscale[z]) <= 1e-9 * abs(b.engine._vscale[z])


def test_hpo_meets_noise_target():
    """make_munu(algo=3): after the noise search the force-fit MAE of the forces-only fit sits at
    noise_f (gppotential.py:1265-1300), and the mean offsets solve their least-squares problem."""
    frames = _frames(2)
    locs = _locals(frames[0], range(18)) + _locals(frames[1], range(0, 18, 2))
    p = PosteriorPotential(engine())
    p.set_data(frames, locs)
    # the reachable force MAE spans 0.0143 (noise -> 0) .. 0.0177 (noise -> 1) on this set; the
    # search is local (BFGS from the current noise), so start it on the slope
    p._noise["all"] = 1.0
    p.make_munu(algo=3, noise_f=0.016)
    mu = p._solve(with_energies=False)
    _, f, _ = p.targets()
    assert abs(np.abs(p.Kf @ mu - f).mean() - 0.016) < 1e-4
    nat = np.array([fr.natoms for fr in frames], float)
    res = (np.array([fr.energy for fr in frames]) - p.Ke @ mu - [p.mean(fr.counts()) for fr in frames]) / nat
    A = np.array([[fr.counts()[z] for z in ac.SPECIES] for fr in frames], float) / nat[:, None]
    assert np.abs(A.T @ res).max() < 1e-10  # normal equations of the per-atom energy residual


def test_switch_thresholds():
    s = Switch([0.1, 2.0, 0.2, 5.0, 0.4])
    assert (s(1.0), s(3.0), s(9.0)) == (0.1, 0.2, 0.4)
    with pytest.raises(RuntimeError, match="not ordered"):
        Switch([0.1, 5.0, 0.2, 2.0, 0.3])
    calc = ActiveCalculator(engine=engine(), logfile=None, ediff=[0.1, 2.0, 0.2])
    calc.maximum_force = 1.0
    assert calc.ediff == 0.1 and calc.ediff_ub == 0.1
    calc.maximum_force = 3.0
    assert calc.ediff == 0.2


def test_tape_golden_blocks():
    """`.sgpr` text written by the reference's own writer (tests/golden/g10_tape.sgpr) parses to
    the arrays it was written from, and our writer reproduces the reference's `local` text
    byte for byte (io/sgprio.py:16-22)."""
    from autoforce_amd.sgprio import format_lce
    path = os.path.join(os.path.dirname(__file__), "golden", "g10_tape.sgpr")
    want = load("g10_tape")
    blocks = SgprIO(path).read()
    locs = [o for k, o in blocks if k == "local"]
    assert len(locs) == int(want["n_local"])
    for q, loc in enumerate(locs):
        assert loc.number == int(want["loc_z"][q])
        a, b = want["loc_ptr"][q], want["loc_ptr"][q + 1]
        np.testing.assert_array_equal(loc._b, want["loc_b"][a:b])
        np.testing.assert_allclose(loc._r, want["loc_r"][a:b], rtol=0, atol=5e-9)  # {:16.8f}
    text = open(path).read()
    for loc in locs:
        assert "start: local\n" + "".join(format_lce(loc)) + "end: local\n" in text
    fr = next(o for k, o in blocks if k == "atoms")
    np.testing.assert_array_equal(fr.numbers, want["at_numbers"])
    np.testing.assert_allclose(fr.positions, want["at_positions"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(fr.forces, want["at_forces"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(fr.stress, want["at_stress"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(fr.cell, want["at_cell"], rtol=0, atol=1e-12)
    assert abs(fr.energy - float(want["at_energy"])) < 1e-12 and fr.pbc.tolist() == want["at_pbc"].tolist()
    params = next(o for k, o in blocks if k == "params")
    assert params == {"ediff": 0.086, "fdiff": 0.129}


def test_edit_sequence_against_reference():
    g = load("g5_big40")
    ac.check_g8_edit_sequence(OracleModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                          species=g["species"].tolist()))


def _al_worker(rank, world, port, tmp, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import pathlib

    import torch.distributed as dist
    import active_common as ac2
    from helpers import OracleModel as OM
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    d = pathlib.Path(tmp) / f"rank{rank}"
    d.mkdir()
    calc, teacher, trace = ac2.run(OM(3, 3, 4, 4.5, species=ac2.SPECIES), d, steps=4, tape=(rank == 0),
                                   process_group=dist.group.WORLD)
    q.put((rank, [t[0] for t in trace], [t[1] for t in trace], trace[-1][2], calc.model.mu))
    dist.barrier()
    dist.destroy_process_group()


def test_frames_without_stress_labels(tmp_path):
    """A teacher without stress (clusters, molecules): the frame contributes energy and force rows only,
    through add / pop / column edits and the model file."""
    from autoforce_amd.modelio import load_model, save_model
    g = load("g5_big40")
    eng = OracleEngine(g)
    rng = np.random.default_rng(3)
    N = len(g["numbers"])
    fr = [Frame(g["numbers"], g["positions"] + 0.02 * k, g["cell"], g["pbc"], float(rng.normal()), rng.normal(size=(N, 3)),
                None if k == 1 else rng.normal(size=6) * 0.01) for k in range(3)]
    p = PosteriorPotential(eng)
    p.Ke, p.Kf, p.Kv = (np.zeros((0, eng.m)),) * 3
    p.add_data(fr)
    assert p.Kv.shape == (12, eng.m) and len(p.targets()[2]) == 12 and p.K.shape[0] == len(np.concatenate(p.targets()))
    x = p.X[3]
    p.add_inducing(x)
    assert p.Kv.shape == (12, eng.m) and p.Ke.shape == (3, eng.m)
    p.pop_1data()       # frame 2 (with stress)
    assert p.Kv.shape[0] == 6
    p.pop_1data()       # frame 1 (without)
    assert p.Kv.shape[0] == 6 and p.ndata == 1
    p.add_data([fr[1]])
    path = str(tmp_path / "m.npz")
    save_model(path, p)
    q = load_model(path, engine=OracleModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                             species=g["species"].tolist()))
    assert q.data[1].stress is None and q.Kv.shape == p.Kv.shape
    np.testing.assert_allclose(q.K, p.K, rtol=0, atol=1e-12)
    with pytest.raises(ValueError, match="energy and forces"):
        p.add_data([Frame(g["numbers"], g["positions"], g["cell"], g["pbc"], 1.0, None, None)])


def test_acceptance_rules_against_reference():
    """g11: accept / reject decisions of the reference's own sampler code on the CPU engine."""
    g = load("g5_big40")
    ac.check_g11_acceptance(OracleModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                        species=g["species"].tolist()))


def test_bcm_against_reference():
    """g12: the reference's own committee combination (active_bcm.py:589-633) on the CPU engine."""
    g = load("g5_big40")
    ac.check_g12_bcm(lambda: OracleModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                         species=g["species"].tolist()))


def test_learning_loop_world2_gloo(tmp_path):
    """Atoms sharded over two ranks (one all-reduce per evaluation, LCEs handed out by their
    owner, rank 0's solve broadcast): every rank takes the same decisions as a single process."""
    import torch.multiprocessing as mp
    (tmp_path / "single").mkdir()
    _, _, ref = ac.run(engine(), tmp_path / "single", steps=4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 7) % 500)
    procs = [ctx.Process(target=_al_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, sizes, energies, forces, mu in got:
        assert sizes == [t[0] for t in ref]
        np.testing.assert_allclose(energies, [t[1] for t in ref], rtol=0, atol=1e-8)
        np.testing.assert_allclose(forces, ref[-1][2], rtol=0, atol=1e-8)
    np.testing.assert_array_equal(got[0][4], got[1][4])


def test_bcm_committee(tmp_path):
    """active_bcm.py:589-633, :885-894: weights from each member's worst covloss, member-wise
    minimum as the sampling uncertainty, initiate_bcm freezes the live model."""
    from autoforce_amd.calculator_bcm import BCMActiveCalculator
    (tmp_path / "a").mkdir()
    calc_a, teacher, tr_a = ac.run(engine(), tmp_path / "a", steps=3, tape=False)
    (tmp_path / "b").mkdir()
    calc_b, _, tr_b = ac.run(engine(), tmp_path / "b", steps=3, seed=3, tape=False)
    at = tr_a[-1][5]
    probe = lambda: Atoms(at.numbers, at.positions, at.cell, True)  # noqa: E731
    outs = []
    for c in (calc_a, calc_b):
        p = probe()
        p.calc = ActiveCalculator(covariance=c.model, logfile=None)
        outs.append((p.get_potential_energy(), p.get_forces(), p.get_stress(), p.calc.get_covloss().copy()))
    bcm = BCMActiveCalculator(covariance=calc_b.model, kernel_model_dict={"a": calc_a.model}, logfile=None)
    p = probe()
    p.calc = bcm
    e, f, s = p.get_potential_energy(), p.get_forces(), p.get_stress()
    scale = []
    for o in outs:
        cm = o[3].max()
        scale.append((-np.log(cm) if cm < 1 else 0.0) / cm)
    w = np.array(scale) / np.sum(scale)
    assert abs(e - (w[0] * outs[0][0] + w[1] * outs[1][0])) < 1e-12
    np.testing.assert_allclose(f, w[0] * outs[0][1] + w[1] * outs[1][1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(s, w[0] * outs[0][2] + w[1] * outs[1][2], rtol=0, atol=1e-14)
    np.testing.assert_array_equal(bcm.get_covloss_total(), np.minimum(outs[0][3], outs[1][3]))
    # model a saw this frame: it carries the larger weight
    assert bcm.bcm_weights["a"] > bcm.bcm_weights["live"]
    # a learning committee: freeze the live model, the next frame seeds a new member
    (tmp_path / "c").mkdir()
    live = BCMActiveCalculator(engine=engine(), calculator=teacher, logfile=str(tmp_path / "c" / "active.log"),
                               pckl=str(tmp_path / "c" / "model"), tape=str(tmp_path / "c" / "model"), **ac.KW)
    p = probe()
    p.calc = live
    p.get_forces()
    assert live.size[0] == 1 and os.path.isfile(tmp_path / "c" / "model_1.npz")
    live.initiate_bcm()
    assert live.size == (0, 0) and len(live.model_dict) == 1
    q = Atoms(at.numbers, at.positions + 0.05, at.cell, True)
    q.calc = live
    q.get_forces()
    assert live.size[0] == 1 and os.path.isfile(tmp_path / "c" / "model_2.npz")
    assert os.path.isfile(tmp_path / "c" / "model_2.sgpr") and set(live.bcm_weights) == {str(tmp_path / "c" / "model_1"), "live"}
    # restart: members on disk are picked up again
    again = BCMActiveCalculator(engine=engine(), logfile=None, pckl=str(tmp_path / "c" / "model"), member_engine=engine)
    assert again.pckl_id == 2 and len(again.model_dict) == 1


def test_hyperparameter_search_against_the_reference():
    g = load("g5_big40")
    ac.check_g14_hpo(OracleModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]),
                                 species=g["species"].tolist()))


def _bcm_worker(rank, world, port, tmp, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import torch.distributed as dist
    from autoforce_amd.ase_shim import Atoms as A
    from autoforce_amd.calculator_bcm import BCMActiveCalculator
    from autoforce_amd.modelio import load_model
    from helpers import OracleModel as OM
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = lambda: OM(3, 3, 4, 4.5, species=[3, 9])  # noqa: E731
    members = {k: load_model(os.path.join(tmp, f"{k}.npz"), engine=eng()) for k in ("a", "b")}
    live = load_model(os.path.join(tmp, "c.npz"), engine=eng())
    bcm = BCMActiveCalculator(covariance=live, kernel_model_dict=members, logfile=None, process_group=dist.group.WORLD,
                              members_over_ranks=True)
    fr = np.load(os.path.join(tmp, "frame.npz"))
    at = A(fr["numbers"], fr["positions"], fr["cell"], True)
    at.calc = bcm
    calls0 = [m.engine.calls for m in members.values()] + [live.engine.calls]
    e, f, s = at.get_potential_energy(), at.get_forces(), at.get_stress()
    calls = [m.engine.calls - c for m, c in zip(list(members.values()) + [live], calls0)]
    q.put((rank, e, f, s, bcm.get_covloss().copy(), dict(bcm.bcm_weights), calls))
    dist.barrier()
    dist.destroy_process_group()


def test_bcm_one_member_per_rank_world2_gloo(tmp_path):
    """members_over_ranks: three members on two ranks, every member evaluated by ONE rank, unsharded; energies, forces,
    stress, weights and the member-wise minimum covloss equal the single-process committee's on every rank."""
    import torch.multiprocessing as mp
    from autoforce_amd.calculator_bcm import BCMActiveCalculator
    from autoforce_amd.modelio import save_model
    calcs = []
    for k, seed in (("a", 0), ("b", 3), ("c", 5)):
        (tmp_path / k).mkdir()
        c, teacher, tr = ac.run(engine(), tmp_path / k, steps=3, seed=seed, tape=False)
        save_model(str(tmp_path / f"{k}.npz"), c.model)
        calcs.append((c, tr))
    at = calcs[0][1][-1][5]
    np.savez(tmp_path / "frame.npz", numbers=at.numbers, positions=at.positions, cell=at.cell)
    ref = BCMActiveCalculator(covariance=calcs[2][0].model, kernel_model_dict={"a": calcs[0][0].model, "b": calcs[1][0].model},
                              logfile=None)
    p = Atoms(at.numbers, at.positions, at.cell, True)
    p.calc = ref
    e0, f0, s0 = p.get_potential_energy(), p.get_forces(), p.get_stress()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + 91) % 500)
    procs = [ctx.Process(target=_bcm_worker, args=(r, 2, port, str(tmp_path), q)) for r in range(2)]
    for pr in procs:
        pr.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for pr in procs:
        pr.join(timeout=60)
        assert pr.exitcode == 0
    for rank, e, f, s, beta, w, calls in got:
        assert abs(e - e0) < 1e-10
        np.testing.assert_allclose(f, f0, rtol=0, atol=1e-10)
        np.testing.assert_allclose(s, s0, rtol=0, atol=1e-12)
        np.testing.assert_allclose(beta, ref.get_covloss_total(), rtol=0, atol=1e-12)
        assert set(w) == set(ref.bcm_weights) and all(abs(w[k] - ref.bcm_weights[k]) < 1e-12 for k in w)
        # members a, c (indices 0, 2) on rank 0, member b on rank 1: nobody evaluates somebody else's member
        assert [c > 0 for c in calls] == ([True, False, True] if rank == 0 else [False, True, False]), (rank, calls)


def test_side_files_of_an_uncertain_frame_and_of_the_test_keyword(tmp_path, monkeypatch):
    """active.py:495-499: a prediction-only calculator whose largest covloss exceeds ediff appends the frame to
    `active_uncertain`; active.py:678-704 (`test = n`): every n steps an active calculator writes the frame with the teacher's
    results to `active_FP` and with the model's to `active_ML` (ase.io.Trajectory with ASE; extended XYZ here) and logs the errors."""
    from autoforce_amd.cl.md import read_frames
    monkeypatch.chdir(tmp_path)
    calc, teacher, trace = ac.run(OracleModel(3, 3, 4, 4.5, species=ac.SPECIES), tmp_path, steps=6, tape=False, test=2)
    fp, ml = read_frames(str(tmp_path / "active_FP.xyz"), ":"), read_frames(str(tmp_path / "active_ML.xyz"), ":")
    assert len(fp) == len(ml) >= 1
    log = open(tmp_path / "active.log").read()
    assert log.count("errors (test):") == len(fp) and "testing energy:" in log
    np.testing.assert_array_equal(fp[-1].positions, ml[-1].positions)
    assert fp[-1].forces is not None and ml[-1].forces is not None and fp[-1].energy != ml[-1].energy
    # the same model without a teacher: every frame whose covloss is above the threshold goes to active_uncertain
    passive = ActiveCalculator(covariance=calc.model, calculator=None, logfile=str(tmp_path / "passive.log"), ediff=1e-6)
    rng = np.random.default_rng(3)
    at = trace[-1][5]
    for k in range(3):
        at2 = Atoms(at.numbers, at.positions + 0.2 * rng.normal(size=at.positions.shape), at.cell, True)
        at2.calc = passive
        at2.get_potential_energy()
    unc = read_frames(str(tmp_path / "active_uncertain.xyz"), ":")
    assert len(unc) == 3 and unc[0].natoms == len(at.numbers) and unc[0].energy is None
