"""GPU tests of the fused last kernel (finalize + binning of the next step) and of the device-resident MD loop built on
it: the resident-frames form against plain steps bit for bit, the device integrator against the host loop
(workloads.langevin_nvt around the same library) bit for bit with the same noise stream, the covloss gate halting
on the device exactly where the host loop's calculator would update (calculator/active.py:492-499), and velocity
Verlet conserving energy."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


class _PredictCalc:
    """The library behind the three ASE getters (what ActiveCalculator.calculate does on a prediction-only step)."""
    implemented_properties = ["energy", "forces", "stress", "free_energy"]

    def __init__(self, mdl):
        self.mdl, self.calls, self.betas = mdl, 0, []
        self._key, self.results = None, {}

    def get_property(self, name, atoms=None):
        key = atoms.positions.tobytes()
        if key != self._key:
            out = self.mdl.predict(atoms.numbers, atoms.positions, atoms.cell, atoms.pbc)
            self.results = dict(energy=out["energy"], forces=out["forces"], stress=out["stress"], free_energy=out["energy"])
            self.betas.append(float(out["beta"].max()))
            self._key = key
            self.calls += 1
        return self.results[name]


def _model(side=8, m=48, scale=0.02, seed=1):
    from autoforce_amd import SGPRModel
    from autoforce_amd.workloads import inducing_from_frame, lips
    numbers, pos, cell, pbc = lips(side, seed=0)
    species = sorted(set(int(z) for z in numbers))
    mdl = SGPRModel(3, 3, 4, 6.0, species=species)
    n2, p2, c2, b2 = lips(side, seed=seed)
    mdl.set_inducing(inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=seed))
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(64, m)), rng.normal(size=64))
    mdl.set_weights(scale * rng.normal(size=m), choli=mdl.choli, vscale=mdl.make_vscale())
    return mdl, (numbers, pos, cell, pbc)


def test_resident_frames_with_the_next_step_binned_by_the_last_kernel():
    """sgpr_step_dev_next over a batch of resident frames (the bench's timed region) against sgpr_step_dev on the same
    frames: every packed result bit for bit, one binning launch for the whole batch, and a call that breaks the chain
    (other positions than announced) is served correctly."""
    import ctypes as C
    import torch
    from autoforce_amd import _lib
    mdl, (numbers, pos, cell, pbc) = _model()
    lib = _lib.load()
    N = len(numbers)
    rng = np.random.default_rng(5)
    frames = [pos]
    for _ in range(40):
        frames.append(frames[-1] + 0.012 * rng.normal(size=pos.shape))
    mdl.predict(numbers, pos, cell, pbc)  # binds, sizes the capacities
    r0 = mdl.list_rebuilds()
    dev = torch.device("cuda:0")
    fr = torch.tensor(np.stack(frames), device=dev)
    cl = torch.tensor(cell, device=dev)
    plen = 4 * N + 11
    out_a = torch.zeros((len(frames), plen), dtype=torch.float64, device=dev)
    out_b = torch.zeros_like(out_a)
    h = mdl.handle
    st = None
    for k in range(len(frames)):
        _lib.check(lib.sgpr_step_dev(h, fr[k].data_ptr(), cl.data_ptr(), out_a[k].data_ptr(), st))
    _lib.check(lib.sgpr_sync_check(h, st))
    r1 = mdl.list_rebuilds()
    mdl.profile(True)
    for k in range(len(frames)):
        nxt = fr[k + 1].data_ptr() if k + 1 < len(frames) else None
        _lib.check(lib.sgpr_step_dev_next(h, fr[k].data_ptr(), cl.data_ptr(), out_b[k].data_ptr(), nxt, st))
        if k == 20:  # a step in the middle of the chain: its last kernel has binned the next frame, no binning launch of its own
            _lib.check(lib.sgpr_sync_check(h, st))
            stages = list(mdl.stage_times())
            assert "finalize_bin_next" in stages and "neighbor_bin" not in stages, stages
    _lib.check(lib.sgpr_sync_check(h, st))
    mdl.profile(False)
    a, b = out_a.cpu().numpy(), out_b.cpu().numpy()
    assert np.array_equal(a, b), np.abs(a - b).max()
    r2 = mdl.list_rebuilds()
    assert r2 - r1 <= (r1 - r0) + 2 and r2 - r1 < len(frames) // 2, (r0, r1, r2)  # the same walk: the same (few) rebuilds, now decided in the fused kernel
    # the chain broken on purpose: frame 3 announced, frame 7 passed
    _lib.check(lib.sgpr_step_dev_next(h, fr[2].data_ptr(), cl.data_ptr(), out_b[2].data_ptr(), fr[3].data_ptr(), st))
    _lib.check(lib.sgpr_step_dev_next(h, fr[7].data_ptr(), cl.data_ptr(), out_b[7].data_ptr(), fr[8].data_ptr(), st))
    _lib.check(lib.sgpr_step_dev_next(h, fr[8].data_ptr(), cl.data_ptr(), out_b[8].data_ptr(), None, st))
    _lib.check(lib.sgpr_sync_check(h, st))
    b = out_b.cpu().numpy()
    assert np.array_equal(a[[2, 7, 8]], b[[2, 7, 8]])
    mdl.close()


def test_device_langevin_equals_the_host_loop_bit_for_bit():
    from autoforce_amd.workloads import langevin_nvt, langevin_nvt_device
    mdl, (numbers, pos, cell, pbc) = _model()
    steps = 70  # (several chunks of launches, candidate rebuilds included: thermal motion crosses skin / 2 within ~40 fs)
    calc = _PredictCalc(mdl)
    host = [(s, E, T, w, p.copy(), v.copy()) for s, E, T, w, p, v in
            langevin_nvt(calc, numbers, pos, cell, pbc, steps, temperature=600.0, dt_fs=1.0, friction=0.05, seed=3)]
    dev = list(langevin_nvt_device(mdl, numbers, pos, cell, pbc, steps, temperature=600.0, dt_fs=1.0, friction=0.05, seed=3, chunk=32))
    assert len(dev) == len(host) == steps + 1
    for (s0, E0, T0, _, _, _), (s1, E1, T1, bmax), b0 in zip(host, dev, calc.betas):
        assert s0 == s1
        assert E0 == E1, (s0, E0, E1)             # same positions -> the same step, bit for bit
        assert abs(T0 - T1) <= 1e-12 * T0          # (the sum over atoms runs in another order)
        assert bmax == b0
    # the state the device ends in = the host loop's last yield
    st = mdl.md_state(results=True)
    assert np.array_equal(st["positions"], host[-1][4])
    assert np.array_equal(st["velocities"], host[-1][5])
    assert mdl.list_rebuilds() > 1
    mdl.close()


def test_velocity_verlet_is_second_order_and_repeats():
    """friction = 0: velocity Verlet.  The total energy error over the same physical time falls fourfold when the step is
    halved (a first-order slip — a force that does not belong to its positions, a half kick applied twice — would halve
    it at best), the potential energy itself moves far more than the total, and a repeated run gives the same bits."""
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS
    mdl, (numbers, pos, cell, pbc) = _model(scale=0.05)
    N = len(numbers)
    mass = np.array([MASS[int(z)] for z in numbers])
    rng = np.random.default_rng(0)
    vel = rng.normal(size=(N, 3)) * np.sqrt(kB * 300.0 / mass[:, None])

    def run(dt_fs, n):
        mdl.md_begin(numbers, pos, cell, pbc, mass, vel, dt=dt_fs * FS, friction=0.0, kT=0.0)
        sc, code = mdl.md_run(n + 1, None, final=True)
        assert code == 0 and len(sc) == n + 1
        return sc, mdl.md_state(results=True)

    sc1, _ = run(1.0, 30)
    sc2, st2 = run(0.5, 60)
    sc4, _ = run(0.25, 120)
    err = [np.abs(sc[:, 0] + 0.5 * sc[:, 12] - (sc[0, 0] + 0.5 * sc[0, 12])).max() for sc in (sc1, sc2, sc4)]
    assert 2.8 < err[0] / err[1] < 5.5 and 2.8 < err[1] / err[2] < 5.5, err
    assert np.ptp(sc4[:, 0]) > 30 * err[2], (np.ptp(sc4[:, 0]), err)
    sc2b, st2b = run(0.5, 60)
    assert np.array_equal(sc2, sc2b) and np.array_equal(st2["positions"], st2b["positions"])
    assert np.array_equal(st2["velocities"], st2b["velocities"])
    mdl.close()


def test_covloss_gate_halts_on_the_device_at_the_step_the_host_would_update():
    """ediff between the largest covloss of the first and of a later configuration: the run must stop exactly at the
    first evaluation that reaches it, with that configuration, its forces and its covloss intact, and go on — through
    the same trajectory — once the handler has dealt with it."""
    from autoforce_amd.workloads import langevin_nvt, langevin_nvt_device
    mdl, (numbers, pos, cell, pbc) = _model()
    steps = 45
    calc = _PredictCalc(mdl)
    # (the generator hands out its own velocity array and goes on updating it in place: copies)
    host = [(s, E, T, w, p.copy(), v.copy()) for s, E, T, w, p, v in
            langevin_nvt(calc, numbers, pos, cell, pbc, steps, temperature=900.0, dt_fs=1.0, friction=0.05, seed=4)]
    b = np.array(calc.betas)
    later = np.nonzero(b > b[:3].max())[0]
    assert len(later), "the covloss never exceeds its starting value on this walk; pick another seed"
    k_halt = int(later[0])
    ediff = 0.5 * (b[:k_halt].max() + b[k_halt])
    seen = []

    def on_halt(model, state):
        seen.append(state)

    dev = list(langevin_nvt_device(mdl, numbers, pos, cell, pbc, steps, temperature=900.0, dt_fs=1.0, friction=0.05, seed=4,
                                   ediff=ediff, chunk=64, on_halt=on_halt))
    # every later evaluation above the (unchanged) threshold halts as well: the first one is what is pinned here
    assert len(seen) >= 1
    st = seen[0]
    assert np.array_equal(st["positions"], host[k_halt][4])
    assert np.array_equal(st["velocities"], host[k_halt][5])
    assert st["beta"].max() == b[k_halt] >= ediff
    ref = mdl.predict(numbers, host[k_halt][4], cell, pbc)
    # (predict() just rebuilt the lists of that frame from scratch: same bits)
    assert np.array_equal(st["forces"], ref["forces"]) and st["energy"] == ref["energy"]
    # the trajectory is the host's, halts or not
    assert [d[1] for d in dev] == [h[1] for h in host]
    mdl.close()


def test_on_the_fly_learning_on_the_device_equals_the_host_loop(tmp_path):
    """ActiveCalculator.run_md (state on the device, the covloss gate halts it, calculate() updates the model) against
    the host loop around the same calculator class (workloads.langevin_nvt: one calculate() per step, as an ASE
    integrator drives it): the same model updates at the same steps, the same log — energies, temperatures,
    covlosses, sizes — line by line, and the same final state bit for bit."""
    import re
    import active_common as ac
    from autoforce_amd import SGPRModel
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.workloads import langevin_nvt
    from helpers import PairTeacher
    steps = 60
    logs, finals, sizes, upd = [], [], [], []
    for mode in ("host", "device"):
        np.random.seed(1234)
        rng0, numbers, pos, cell = ac.start(0)
        d = tmp_path / mode
        d.mkdir()
        calc = ActiveCalculator(engine=SGPRModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=PairTeacher(rc=4.0),
                                logfile=str(d / "active.log"), pckl=None, tape=None, **ac.KW)
        vel = 0.02 * np.random.default_rng(3).normal(size=pos.shape)
        if mode == "host":
            out = []
            for st, E, T, _, p, v in langevin_nvt(calc, numbers, pos, cell, True, steps, 300.0, 1.0, 0.02, vel=vel,
                                                  rng=np.random.default_rng(9)):
                out.append((st, E, T, bool(calc.updated)))
                last = (p.copy(), v.copy())
        else:
            at = Atoms(numbers, pos, cell, True, velocities=vel)
            assert calc.md_on_device_ok() or calc._needs_seed()
            out = list(calc.run_md(at, steps, 300.0, dt_fs=1.0, friction=0.02, rng=np.random.default_rng(9), chunk=16))
            last = (at.positions.copy(), at.get_velocities())
        txt = open(d / "active.log").read().splitlines()
        logs.append([re.sub(r"^\S+ \S+ ", "", ln) for ln in txt])   # (drop the time stamps)
        finals.append(last)
        sizes.append(calc.size)
        upd.append([o[0] for o in out if o[3]])
        assert len(out) == steps + 1
        energies = [o[1] for o in out]
        if mode == "host":
            e_host = energies
        else:
            first = next((k for k, (a, b) in enumerate(zip(energies, e_host)) if a != b), None)
            diff = next(((a, b) for a, b in zip(logs[0], logs[1]) if a.split(" ")[:2] != b.split(" ")[:2]), None)
            assert first is None, (first, upd, sizes, diff)
        calc.engine.close()
    assert sizes[0] == sizes[1] and sizes[0][1] > 2
    assert upd[0] == upd[1] and len(upd[0]) >= 2, upd     # the model was updated several times, at the same steps
    assert len(upd[0]) < steps // 2                           # ... and most steps ran without the host
    assert len(logs[0]) == len(logs[1])
    num = re.compile(r"^(\d+) (\S+) (\S+) (\S+) $")
    for a, b in zip(logs[0], logs[1]):
        ma, mb = num.match(a), num.match(b)
        if ma and mb:   # a step's line: energy and covloss bit for bit, the temperature to the order of its sum
            assert ma.group(1) == mb.group(1) and ma.group(2) == mb.group(2) and ma.group(4) == mb.group(4), (a, b)
            assert abs(float(ma.group(3)) - float(mb.group(3))) <= 1e-12 * float(ma.group(3)), (a, b)
        else:
            assert a == b
    assert np.array_equal(finals[0][0], finals[1][0]) and np.array_equal(finals[0][1], finals[1][1])


def test_deviates_drawn_on_the_device():
    """A seeded run (the integrator draws its own deviates: Philox4x32-10 + Box-Muller on (seed, configuration, atom,
    component)): standard normal to sampling accuracy; the host loop fed the SAME deviates (SGPRModel.md_deviates)
    reproduces it bit for bit; and the trajectory does not depend on how the run is cut into calls."""
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS, langevin_nvt
    mdl, (numbers, pos, cell, pbc) = _model()
    N = len(numbers)
    mass = np.array([MASS[int(z)] for z in numbers])
    vel = np.random.default_rng(2).normal(size=(N, 3)) * np.sqrt(kB * 600.0 / mass[:, None])
    steps = 40

    def run(cuts):
        mdl.md_begin(numbers, pos, cell, pbc, mass, vel, dt=FS, friction=0.05, kT=kB * 600.0, seed=77)
        out = []
        for k, n in enumerate(cuts):
            sc, code = mdl.md_run(n, None, final=(k == len(cuts) - 1))
            assert code == 0 and len(sc) == n
            out.append(sc)
        return np.concatenate(out), mdl.md_state(results=True)

    sc_a, st_a = run([steps + 1])
    xi = mdl.md_deviates(0, steps)
    n = xi.size   # 61440 deviates: four standard errors of each moment
    assert abs(xi.mean()) < 4.0 / np.sqrt(n) and abs(xi.std() - 1.0) < 4.0 / np.sqrt(2 * n) and abs((xi ** 3).mean()) < 4.0 * np.sqrt(6.0 / n)
    assert abs((xi ** 4).mean() - 3.0) < 4.0 * np.sqrt(96.0 / n) and np.abs(np.corrcoef(xi[:-1].ravel(), xi[1:].ravel())[0, 1]) < 4.0 / np.sqrt(n)
    assert np.abs(xi).max() > 3.5    # (the tails are there)
    assert not np.array_equal(xi, mdl.md_deviates(1, steps))      # (the counter matters)
    sc_b, st_b = run([7, 1, 16, 17])                                # 41 evaluations in four calls
    assert np.array_equal(sc_a[:, :13], sc_b[:, :13])
    assert np.array_equal(st_a["positions"], st_b["positions"]) and np.array_equal(st_a["velocities"], st_b["velocities"])

    class Rows:   # a Generator stand-in that deals the device's rows to the host loop
        def __init__(self):
            self.k = 0

        def normal(self, size):
            self.k += 1
            return xi[self.k - 1]

    calc = _PredictCalc(mdl)
    host = [(s, E, T, w, p.copy(), v.copy()) for s, E, T, w, p, v in
            langevin_nvt(calc, numbers, pos, cell, pbc, steps, 600.0, 1.0, 0.05, vel=vel, rng=Rows())]
    assert [h[1] for h in host] == sc_a[:, 0].tolist()
    assert np.array_equal(host[-1][4], st_a["positions"]) and np.array_equal(host[-1][5], st_a["velocities"])
    mdl.close()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_frames_with_the_next_step_binned(world):
    """A sharded rank's step over resident frames (scatter-form reverse pass, fixed-point force sums) with the next
    frame binned by its last kernel, rank by rank on this one device: every packed partial result bit for bit equal to
    sgpr_step_dev's, and the partial sums add up to the unsharded step."""
    import torch
    from autoforce_amd import _lib
    mdl, (numbers, pos, cell, pbc) = _model()
    lib, h = _lib.load(), mdl.handle
    N = len(numbers)
    rng = np.random.default_rng(6)
    frames = [pos]
    for _ in range(24):
        frames.append(frames[-1] + 0.015 * rng.normal(size=pos.shape))
    dev = torch.device("cuda:0")
    fr = torch.tensor(np.stack(frames), device=dev)
    cl = torch.tensor(cell, device=dev)
    plen = 4 * N + 11
    whole = mdl.predict(numbers, frames[-1], cell, pbc)
    tot = np.zeros(plen)
    for r in range(world):
        mdl.predict(numbers, pos, cell, pbc, rank=r, world=world)   # binds the share, sizes the capacities
        out = []
        for use_next in (False, True):
            o = torch.zeros((len(frames), plen), dtype=torch.float64, device=dev)
            for k in range(len(frames)):
                nxt = fr[k + 1].data_ptr() if (use_next and k + 1 < len(frames)) else None
                _lib.check(lib.sgpr_step_dev_next(h, fr[k].data_ptr(), cl.data_ptr(), o[k].data_ptr(), nxt, None))
            _lib.check(lib.sgpr_sync_check(h, None))
            out.append(o.cpu().numpy())
        assert np.array_equal(out[0], out[1]), (r, np.abs(out[0] - out[1]).max())
        tot += out[1][-1]
    fmax = np.abs(whole["forces"]).max()
    assert np.abs(tot[:3 * N].reshape(N, 3) - whole["forces"]).max() <= 1e-10 * fmax
    assert abs(tot[4 * N] - whole["energy"]) <= 1e-10 * max(1.0, abs(whole["energy"]))
    mdl.close()


def test_command_line_md_on_the_device(tmp_path, monkeypatch):
    """autoforce_amd.cl.md (theforce/cl/md.py's keywords): Langevin on the device loop, a trajectory frame every fifth
    step brought back from device memory, the ARGS file read with the reference's syntax."""
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.cl import get_default_args, read_args, update_args
    from autoforce_amd.cl.md import md, read_structure
    monkeypatch.chdir(tmp_path)
    (tmp_path / "ARGS").write_text("dynamics = 'Langevin'\ntem = 600.   # K\npicos = -20\nloginterval = 5\nfriction = 0.02\nseed = 3\n")
    mdl, (numbers, pos, cell, pbc) = _model()
    calc = ActiveCalculator(covariance=mdl, logfile=str(tmp_path / "active.log"))
    assert calc.md_on_device_ok()
    atoms = Atoms(numbers, pos, cell, pbc)
    kw = update_args(get_default_args(md), read_args())
    kw.pop("calc")
    md(atoms, calc=calc, **kw)
    lines = open("md.xyz").read().splitlines()
    assert sum("Lattice=" in ln for ln in lines) == 5         # steps 0, 5, 10, 15, 20
    frames = [read_structure("md.xyz", k).positions for k in range(5)]
    for a, b in zip(frames, frames[1:]):
        d = np.abs(b - a).max()
        assert 1e-3 < d < 0.5, d                               # five thermal steps apart
    np.testing.assert_allclose(frames[-1], atoms.positions, rtol=0, atol=1e-13)
    log = [ln for ln in open(tmp_path / "active.log").read().splitlines()]
    assert sum(1 for ln in log if len(ln.split()) == 6 and ln.split()[2].isdigit()) >= 21   # one line per step
    mdl.close()


def test_handles_give_back_their_device_memory():
    """A handle owns its device buffers by name (no destructors): after a predict step, a device MD run with a halt, a
    refit and `close()` the free memory of the device is what it was, handle after handle — the MD state, the tile tables
    and the mapped host words of the run included (sgpr_destroy)."""
    import torch
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS

    def cycle():
        mdl, (numbers, pos, cell, pbc) = _model(side=16, m=48)   # 4096 atoms: the MD state alone is 1.3 MB
        mdl.predict(numbers, pos, cell, pbc)
        mass = np.array([MASS[int(z)] for z in numbers])
        mdl.md_begin(numbers, pos, cell, pbc, mass, np.zeros_like(pos), dt=FS, friction=1e-3, kT=kB * 300.0, seed=3)
        mdl.md_run(12, None)
        mdl.md_end()
        mdl.close()

    cycle()  # (the first two handles also pay for pools of the runtime: +48 MB once, with the second handle's streams)
    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(6):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (4 << 20), (free0, free1)   # (six leaked MD states would be 8 MB; the driver hands out 2 MB pages)


def test_a_capacity_outgrown_inside_the_device_loop_is_resized_and_the_run_goes_on():
    """A cluster that contracts: the neighbour counts grow past the capacity the lists were allocated with WHILE the device
    loop runs.  The step concerned flags the overflow, the run halts there (code 2), the next call re-sizes and repeats the
    step — and the trajectory is the host loop's, bit for bit, as if nothing had happened."""
    from autoforce_amd import SGPRModel
    from autoforce_amd.workloads import FS, inducing_from_frame, langevin_nvt, langevin_nvt_device, lips
    numbers, pos, cell, pbc = lips(8, seed=0)
    pos = (pos - pos.mean(0)) * (3.3 / 2.72)                 # 512 atoms, 3.3 A apart: ~25 neighbours inside 6 A
    cell = np.eye(3) * 400.0
    pbc = np.array([False, False, False])
    species = sorted(set(int(z) for z in numbers))
    mdl = SGPRModel(3, 3, 4, 6.0, species=species)
    mdl.set_inducing(inducing_from_frame(mdl, numbers, pos, cell, pbc, 24, seed=1))
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(40, 24)), rng.normal(size=40))
    mdl.set_weights(1e-4 * rng.normal(size=24), choli=mdl.choli, vscale=mdl.make_vscale())   # (nearly ballistic motion)
    steps = 60
    vel = -pos * (0.30 / (steps * FS))                        # homogeneous contraction by 30 % over the run: ~75 neighbours
    nn0 = np.diff(mdl.neighbors(len(numbers))[0]).max() if mdl.predict(numbers, pos, cell, pbc) else 0
    calc = _PredictCalc(mdl)
    host = [(s, E, p.copy(), v.copy()) for s, E, T, w, p, v in
            langevin_nvt(calc, numbers, pos, cell, pbc, steps, temperature=0.0, dt_fs=1.0, friction=0.0, seed=3, vel=vel)]
    nn1 = np.diff(mdl.neighbors(len(numbers))[0]).max()
    assert nn0 < 64 < nn1, (nn0, nn1)                         # the run crosses the initial capacity of 64 neighbours
    # a fresh handle (capacities as allocated for the START frame) runs the same trajectory on the device
    dev_mdl = SGPRModel(3, 3, 4, 6.0, species=species)
    dev_mdl.set_inducing(mdl.X)
    dev_mdl.set_weights(mdl.mu, choli=mdl.choli, vscale=mdl.make_vscale())
    codes = []
    real_run = dev_mdl.md_run
    def spy(*a, **k):
        sc, code = real_run(*a, **k)
        codes.append(code)
        return sc, code
    dev_mdl.md_run = spy
    dev = list(langevin_nvt_device(dev_mdl, numbers, pos, cell, pbc, steps, temperature=0.0, dt_fs=1.0, friction=0.0, seed=3, vel=vel, chunk=16))
    assert 2 in codes, codes                                  # at least one call ended on an outgrown capacity
    assert len(dev) == len(host) == steps + 1
    for (s0, E0, _, _), (s1, E1, T1, bmax) in zip(host, dev):
        assert s0 == s1 and E0 == E1, (s0, E0, E1)
    st = dev_mdl.md_state(results=True)
    assert np.array_equal(st["positions"], host[-1][2]) and np.array_equal(st["velocities"], host[-1][3])
    mdl.close(); dev_mdl.close()


def test_device_nose_hoover_equals_the_host_twin_bit_for_bit_and_survives_halts():
    """Nose-Hoover NVT (the reference's default dynamics: ase.md.npt.NPT with pfactor = None, cl/md.py:17, :131-166) inside the
    step's last kernel against workloads.nose_hoover_nvt around the same library: positions, centred velocities, zeta and
    its integral after 60 steps bit for bit, however the run is batched; a covloss halt in the middle hands back exactly the
    host loop's configuration and the run goes on along the same trajectory."""
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS, nose_hoover_nvt
    mdl, (numbers, pos, cell, pbc) = _model()
    N, steps, T, tdamp = len(numbers), 60, 700.0, 20.0
    mass = np.array([MASS[int(z)] for z in numbers])
    rng = np.random.default_rng(8)
    vel = rng.normal(size=(N, 3)) * np.sqrt(kB * T / mass[:, None])
    calc = _PredictCalc(mdl)
    host = [(s, E, Tk, p.copy(), v.copy(), z, zi) for s, E, Tk, w, p, v, z, zi in
            nose_hoover_nvt(calc, numbers, pos, cell, pbc, steps, temperature=T, dt_fs=1.0, tdamp_fs=tdamp, vel=vel)]
    b = np.array(calc.betas)

    def begin():
        mdl.md_begin(numbers, pos, cell, pbc, mass, vel, dt=1.0 * FS, friction=0.0, kT=kB * T, ttime=tdamp * FS)

    begin()
    rows = []
    for n in (7, 1, 20, 33):
        sc, code = mdl.md_run(n, None, final=(len(rows) + n == steps + 1))
        assert code == 0 and len(sc) == n
        rows.extend(sc)
    sc = np.array(rows)
    assert len(sc) == steps + 1
    assert [r[0] for r in sc] == [h[1] for h in host]                         # energies: same positions, same step
    assert np.array_equal(sc[:, 14], np.array([h[5] for h in host]))           # zeta
    assert np.array_equal(sc[:, 15], np.array([h[6] for h in host]))           # its integral
    np.testing.assert_allclose(sc[:, 12] / (3 * N * kB), [h[2] for h in host], rtol=1e-12)
    st = mdl.md_state(results=True)
    assert np.array_equal(st["positions"], host[-1][3]) and np.array_equal(st["velocities"], host[-1][4])
    assert np.array_equal(st["velocities_pre"], host[-2][4])                   # what the integrator held when it asked for F
    # the thermostat acts: zeta leaves zero, the temperature stays near the target
    assert np.abs(sc[:, 14]).max() > 0 and abs(np.mean(sc[20:, 12]) / (3 * N * kB) - T) < 0.25 * T
    # a halt in the middle
    later = np.nonzero(b > b[:3].max())[0]
    assert len(later), "the covloss never exceeds its starting value on this walk"
    k = int(later[0])
    ediff = 0.5 * (b[:k].max() + b[k])
    begin()
    sc1, code = mdl.md_run(steps + 1, None, ediff=ediff, final=True)
    assert code == 1 and len(sc1) == k + 1
    sth = mdl.md_state(results=True)
    assert np.array_equal(sth["positions"], host[k][3]) and np.array_equal(sth["velocities"], host[k][4])
    assert np.array_equal(sth["velocities_pre"], host[k - 1][4] if k else vel)
    sc2, code = mdl.md_run(steps + 1 - k, None, ediff=0.0, final=True)         # the halted configuration again, then on
    assert code == 0 and [r[0] for r in sc2] == [h[1] for h in host[k:]]
    assert np.array_equal(sc2[:, 14], np.array([h[5] for h in host[k:]]))
    st2 = mdl.md_state(results=True)
    assert np.array_equal(st2["positions"], host[-1][3]) and np.array_equal(st2["velocities"], host[-1][4])
    mdl.close()


def test_nose_hoover_conserves_its_extended_energy():
    """E + KE + 1.5 N kT (ttime zeta)^2 + 3 (N - 1) kT int zeta dt (ase.md.npt.NPT.get_gibbs_free_energy without the cell
    terms) over 1500 steps: its drift is a small fraction of what the thermostat moves in and out of the system."""
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS, fit_to_teacher
    mdl, (numbers, pos, cell, pbc) = _model(scale=0.05)
    fit_to_teacher(mdl, numbers, pos, cell, pbc)
    mdl.set_weights(mdl.mu, choli=mdl.choli, vscale=mdl.make_vscale())
    N, T, tdamp = len(numbers), 600.0, 25.0
    mass = np.array([MASS[int(z)] for z in numbers])
    rng = np.random.default_rng(1)
    vel = rng.normal(size=(N, 3)) * np.sqrt(kB * 300.0 / mass[:, None])   # starts cold: the thermostat has work to do
    mdl.md_begin(numbers, pos, cell, pbc, mass, vel, dt=1.0 * FS, friction=0.0, kT=kB * T, ttime=tdamp * FS)
    rows = []
    while len(rows) < 1500:
        sc, code = mdl.md_run(min(500, 1500 - len(rows)), None)
        assert code in (0, 2)
        rows.extend(sc)
    sc = np.array(rows)
    ttime, kT = tdamp * FS, kB * T
    thermostat = 1.5 * N * kT * (ttime * sc[:, 14]) ** 2 + 3 * (N - 1) * kT * sc[:, 15]
    H = sc[:, 0] + 0.5 * sc[:, 12] + thermostat
    Tk = sc[:, 12] / (3 * N * kB)
    assert abs(Tk[500:].mean() - T) < 0.08 * T, Tk[500:].mean()
    assert np.ptp(thermostat) > 0.5 * 1.5 * N * kB * 300.0            # it pumped a good part of the missing kinetic energy in
    assert np.abs(H - H[0]).max() < 0.02 * np.ptp(thermostat), (np.abs(H - H[0]).max(), np.ptp(thermostat))
    mdl.close()


def test_run_md_nose_hoover_on_the_device_equals_the_host_loop(tmp_path):
    """ActiveCalculator.run_md(tdamp_fs=...) — the reference CLI's default dynamics — with the state on the device against
    the host loop (workloads.nose_hoover_nvt around calculate()) of the same calculator class: same updates at the same
    steps, same energies, same final state."""
    import active_common as ac
    from autoforce_amd import SGPRModel
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    from helpers import PairTeacher
    from autoforce_amd.workloads import nose_hoover_nvt

    def run(device):
        np.random.seed(5)
        rng0, numbers, pos, cell = ac.start(0)
        d = tmp_path / ("dev" if device else "host")
        d.mkdir()
        calc = ActiveCalculator(engine=SGPRModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=PairTeacher(rc=4.0),
                                logfile=str(d / "active.log"), pckl=None, tape=None, **ac.KW)
        at = Atoms(numbers, pos, cell, True)
        vel = np.random.default_rng(3).normal(size=pos.shape) * 0.02
        at.set_velocities(vel)
        steps = 40
        if device:
            out = [(s, E, bool(u)) for s, E, T, u, w in calc.run_md(at, steps, 300.0, dt_fs=1.0, tdamp_fs=20.0, chunk=16)]
            return out, at.positions.copy(), at.get_velocities(), calc.size
        out = []
        numbers_, pos_, cell_, pbc_ = calc._system(at)
        for s, E, T, w, p, v, z, zi in nose_hoover_nvt(calc, numbers_, pos_, cell_, pbc_, steps, 300.0, 1.0, 20.0, vel=vel):
            out.append((s, E, bool(calc.updated)))
            last = (p.copy(), v.copy())
        return out, last[0], last[1], calc.size

    dev, xd, vd, sized = run(True)
    host, xh, vh, sizeh = run(False)
    assert sized == sizeh and sized[1] > 2
    assert [d[0] for d in dev] == [h[0] for h in host]
    np.testing.assert_allclose([d[1] for d in dev], [h[1] for h in host], rtol=0, atol=1e-9)
    assert [d[2] for d in dev] == [h[2] for h in host]
    np.testing.assert_allclose(xd, xh, rtol=0, atol=1e-9)
    np.testing.assert_allclose(vd, vh, rtol=0, atol=1e-9)


def test_device_loop_survives_other_frames_on_the_handle_between_its_calls():
    """A model update between two sgpr_md_run calls evaluates OTHER frames on the same handle (the training rows of stored
    frames, trial models): the handle is then bound to a system of another size or composition.  The run binds its own
    system back (and sgpr_md_state reads the run's own permutation): the trajectory is the uninterrupted one, bit for bit."""
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS, lips
    mdl, (numbers, pos, cell, pbc) = _model()
    N = len(numbers)
    mass = np.array([MASS[int(z)] for z in numbers])
    vel = np.random.default_rng(6).normal(size=(N, 3)) * np.sqrt(kB * 600.0 / mass[:, None])

    def begin():
        mdl.md_begin(numbers, pos, cell, pbc, mass, vel, dt=1.0 * FS, friction=0.02, kT=kB * 600.0, seed=5)

    begin()
    sc, _ = mdl.md_run(30, None, final=True)
    ref = mdl.md_state(results=True)
    begin()
    sc1, _ = mdl.md_run(12, None)
    # another system on the handle: fewer atoms, and one of the same size with another composition
    n2, p2, c2, b2 = lips(6, seed=3)
    mdl.predict(n2, p2, c2, b2)
    shuffled = np.random.default_rng(1).permutation(numbers)
    mdl.predict(shuffled, pos, cell, pbc)
    mid = mdl.md_state()                                  # (while the handle is bound elsewhere)
    sc2, _ = mdl.md_run(18, None, final=True)
    st = mdl.md_state(results=True)
    assert np.array_equal(np.concatenate([sc1, sc2])[:, 0], sc[:, 0])
    assert np.array_equal(st["positions"], ref["positions"]) and np.array_equal(st["velocities"], ref["velocities"])
    begin()
    mdl.md_run(12, None)
    assert np.array_equal(mdl.md_state()["positions"], mid["positions"])
    mdl.close()


def test_one_off_entry_points_give_their_device_memory_back():
    """sgpr_jitcholesky and sgpr_md_deviates work in buffers of their own: given back on every way out."""
    import torch
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS
    mdl, (numbers, pos, cell, pbc) = _model()
    mass = np.array([MASS[int(z)] for z in numbers])
    mdl.md_begin(numbers, pos, cell, pbc, mass, None, dt=FS, friction=1e-3, kT=kB * 300.0, seed=3)
    rng = np.random.default_rng(0)
    A = rng.normal(size=(700, 700))
    A = A @ A.T + 700 * np.eye(700)

    def cycle():
        mdl.jitcholesky(A)
        mdl.md_deviates(0, 64)

    cycle()
    torch.cuda.synchronize()
    free0 = torch.cuda.mem_get_info()[0]
    for _ in range(8):
        cycle()
    torch.cuda.synchronize()
    free1 = torch.cuda.mem_get_info()[0]
    assert free0 - free1 < (4 << 20), (free0, free1)   # (eight leaked calls were 80 MB)
    with pytest.raises(Exception):
        mdl.jitcholesky(-np.eye(300))                   # the error path, too
    torch.cuda.synchronize()
    assert free0 - torch.cuda.mem_get_info()[0] < (4 << 20)
    mdl.close()


def test_npt_with_a_moving_cell_around_the_device_calculator():
    """cl/md.py:131-166 with a bulk modulus: the barostat's integrator (autoforce_amd/npt.py, ASE's NPT restated) around the
    device predict path — forces AND stress of every step feed the next cell.  The cell changes every step; the candidate
    lists are kept while the strain since their build allows it (neighbor.hip's affine rule): against a handle that rebuilds
    its lists at every step the trajectory — positions, cell, strain rate — is the same bits, with a handful of rebuilds
    instead of one per step; the extended system's conserved quantity holds."""
    from autoforce_amd import _lib
    from autoforce_amd.ase_shim import Atoms, kB
    from autoforce_amd.npt import GPA, NPT
    from autoforce_amd.workloads import FS, MASS
    steps = 60
    out = {}
    for skin0 in (False, True):
        mdl, (numbers, pos, cell, pbc) = _model()
        if skin0:
            _lib.check(_lib.load().sgpr_set_option(mdl.handle, b"skin_milliangstrom", 0))
        mass = np.array([MASS[int(z)] for z in numbers])
        cell0 = cell.copy()
        rng = np.random.default_rng(3)
        v = rng.normal(size=pos.shape) * np.sqrt(kB * 600.0 / mass)[:, None]
        at = Atoms(numbers, pos, cell, pbc, velocities=v, masses=mass)
        at.calc = _PredictCalc(mdl)
        dyn = NPT(at, 1.0 * FS, 600.0, externalstress=1.0 * GPA, ttime=25.0 * FS, pfactor=(100.0 * FS) ** 2 * 30.0 * GPA)
        r0 = mdl.list_rebuilds()
        G = [dyn.get_gibbs_free_energy() for _ in dyn.run(steps)]
        out[skin0] = (at.positions.copy(), np.asarray(at.cell).copy(), dyn.eta.copy(), dyn.zeta, mdl.list_rebuilds() - r0, np.array(G),
                      at.calc.calls)
        mdl.close()
    fast, slow = out[False], out[True]
    for a, b in zip(fast[:4], slow[:4]):
        np.testing.assert_array_equal(a, b)
    assert slow[4] >= steps and fast[4] <= steps // 4, (fast[4], slow[4])
    assert fast[6] == steps + 1                                  # one evaluation per configuration
    assert np.abs(fast[1] - cell0).max() > 1e-4                  # the cell has moved
    assert np.ptp(fast[5]) < 0.05 * max(abs(fast[5][0]), 1.0)


def test_command_line_npt_with_a_bulk_modulus_on_the_device_calculator(tmp_path, monkeypatch):
    """autoforce_amd.cl.md with `dynamics = 'NPT'` and a `bulk_modulus` (theforce/cl/md.py:131-166 with a moving cell): the
    restated ASE integrator drives ActiveCalculator.calculate() on the HIP engine once per step — one log line per step, a
    trajectory whose lattice changes, `iso` keeping the shape, the triclinic start cell rotated to upper-triangular form with
    the interatomic distances (hence the energy) untouched."""
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.cl import get_default_args, read_args, update_args
    from autoforce_amd.cl.md import md, read_frames
    monkeypatch.chdir(tmp_path)
    (tmp_path / "ARGS").write_text("dynamics = 'NPT'\nbulk_modulus = 25.\nstress = 0.2   # GPa\niso = True\ntem = 500.\npicos = -12\n"
                                   "loginterval = 4\ntdamp = 25\npdamp = 100\nml_filter = 0.8\nseed = 3\n")
    mdl, (numbers, pos, cell, pbc) = _model()
    # a rigidly rotated copy of the frame: the same physics in a cell that is NOT upper triangular
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]]) @ \
        np.array([[1.0, 0.0, 0.0], [0.0, np.cos(0.2), -np.sin(0.2)], [0.0, np.sin(0.2), np.cos(0.2)]])
    calc = ActiveCalculator(covariance=mdl, logfile=str(tmp_path / "active.log"))
    e0 = mdl.predict(numbers, pos, cell, pbc)["energy"]
    atoms = Atoms(numbers, pos @ R.T, cell @ R.T, pbc)
    kw = update_args(get_default_args(md), read_args())
    kw.pop("calc")
    md(atoms, calc=calc, **kw)
    frames = read_frames("md.xyz", ":")
    assert len(frames) == 4                                      # steps 0, 4, 8, 12
    c0 = np.asarray(frames[0].cell)
    assert c0[1, 0] == c0[2, 0] == c0[2, 1] == 0.0 and abs(abs(np.linalg.det(c0)) - abs(np.linalg.det(cell))) < 1e-9 * abs(np.linalg.det(cell))
    assert abs(frames[0].energy - e0) < 1e-9 * max(1.0, abs(e0))  # the rotation changed nothing the model sees
    c1 = np.asarray(atoms.cell)
    assert np.abs(c1 - c0).max() > 1e-6                          # the cell moved ...
    np.testing.assert_allclose(c1 / c1[2, 2], c0 / c0[2, 2], atol=1e-12)   # ... and kept its shape (iso)
    log = [ln for ln in open(tmp_path / "active.log").read().splitlines()]
    assert sum(1 for ln in log if len(ln.split()) == 6 and ln.split()[2].isdigit()) >= 13   # one line per step
    mdl.close()
