"""GPU: the library's own exchange between ranks (sgpr_peer_*: hipIpc-mapped receive buffers, one push per peer, local sum
in rank order) with 2 and 4 PROCESSES on the one GPU of the test box — RCCL refuses that, this exchange does not — and the
device MD loop sharded over it.

Reference semantics: the ranks' partial sums are combined by all-reduces of zero-padded arrays, i.e. an all-gather and a sum
(calculator/active.py:562, :600-602, :770-777; _mpi4py.py:47-56); every rank then holds every force, and the integrator of
cl/md.py:117-128 runs replicated on all of them.  Checked here:
  * a sharded step through the exchange = the single-process step in scatter form BIT FOR BIT in forces and covloss on every
    rank (the fixed-point force sums travel and are added as integers), energy / stress to rounding — and = the gather
    form of the default single-process path to 1e-10;
  * the sharded device MD loop: positions and velocities after 40 Langevin steps bit for bit those of the single process, on
    every rank; the covloss gate halts every rank at the same evaluation; runs cut into different batches agree;
  * the free-standing all-reduce (SUM, MAX) over the exchange."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

STEPS = 40


def _otf_worker(rank, world, port, tmp, q, tdamp):
    """On-the-fly learning with the MD state on the devices of `world` ranks (ActiveCalculator.run_md over the exchange)."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import pathlib
    import torch.distributed as dist
    import active_common as ac
    from autoforce_amd import SGPRModel
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.watchdog import Watchdog
    from helpers import PairTeacher
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SGPR_PEER_TIMEOUT_MS"] = "20000"
    with Watchdog(f"sharded on-the-fly MD, rank {rank} of {world}", seconds=240, rank=rank):
        if world > 1:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        np.random.seed(1234)
        rng0, numbers, pos, cell = ac.start(0)
        d = pathlib.Path(tmp) / f"w{world}r{rank}"
        d.mkdir()
        calc = ActiveCalculator(engine=SGPRModel(3, 3, 4, 4.5, species=ac.SPECIES), calculator=PairTeacher(rc=4.0),
                                logfile=str(d / "active.log"), pckl=None, tape=None,
                                process_group=dist.group.WORLD if world > 1 else None, **ac.KW)
        at = Atoms(numbers, pos, cell, True, velocities=0.02 * np.random.default_rng(3).normal(size=pos.shape))
        out = [(s, E, bool(u)) for s, E, T, u, w in calc.run_md(at, 40, 300.0, dt_fs=1.0, friction=0.02, seed=7, chunk=16, tdamp_fs=tdamp)]
        on_device = calc.md_on_device_ok()
        q.put((rank, out, at.positions.copy(), at.get_velocities(), calc.size, on_device, int(getattr(calc.engine, "peer_world", 1))))
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()


def _build(side=8, m=48, scale=0.02, seed=1, scatter=False):
    from autoforce_amd import SGPRModel, _lib
    from autoforce_amd.workloads import inducing_from_frame, lips
    numbers, pos, cell, pbc = lips(side, seed=0)
    species = sorted(set(int(z) for z in numbers))
    mdl = SGPRModel(3, 3, 4, 6.0, species=species)
    if scatter:
        _lib.check(_lib.load().sgpr_set_option(mdl.handle, b"reverse_scatter", 1))
    n2, p2, c2, b2 = lips(side, seed=seed)
    mdl.set_inducing(inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=seed))
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(64, m)), rng.normal(size=64))
    mdl.set_weights(scale * rng.normal(size=m), choli=mdl.choli, vscale=mdl.make_vscale())
    return mdl, (numbers, pos, cell, pbc)


def _md(mdl, system, steps, batches, ediff=0.0):
    """`steps` evaluations of a seeded Langevin run cut into `batches`; returns the final state and the scalars."""
    from autoforce_amd.ase_shim import kB
    from autoforce_amd.workloads import FS, MASS
    numbers, pos, cell, pbc = system
    masses = np.array([MASS[int(z)] for z in numbers])
    rng = np.random.default_rng(3)
    vel = rng.normal(size=pos.shape) * np.sqrt(kB * 600.0 / masses)[:, None]
    mdl.md_begin(numbers, pos, cell, pbc, masses, vel, dt=1.0 * FS, friction=0.02, kT=kB * 600.0, seed=11)
    rows, code, left = [], 0, steps
    sizes = iter(list(batches) + [100] * 8)
    n = next(sizes)
    while left > 0:
        sc, code = mdl.md_run(min(n, left), None, ediff=ediff, final=False)
        rows.append(sc)
        left -= len(sc)
        if code == 1:      # the covloss gate
            break
        if code == 0:      # (2: a neighbour capacity was outgrown on some rank — the next call re-sizes and repeats the evaluation)
            n = next(sizes)
    st = mdl.md_state(results=False)
    return st["positions"], st["velocities_pre"], np.concatenate(rows), code


def _worker(rank, world, port, q, ediff):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import torch
    import torch.distributed as dist
    from autoforce_amd import _lib
    from autoforce_amd.watchdog import Watchdog
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SGPR_PEER_TIMEOUT_MS"] = "20000"   # (several processes share the one GPU of the test box)
    with Watchdog(f"peer exchange, rank {rank} of {world}", seconds=240, rank=rank):
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mdl, system = _build()
        numbers, pos, cell, pbc = system
        N = len(numbers)
        blobs = [None] * world
        dist.all_gather_object(blobs, mdl.peer_export(rank, world, 7 * N + 11))
        mdl.peer_attach(blobs)
        dist.barrier()
        out = mdl.predict(numbers, pos, cell, pbc, rank=rank, world=world)
        # a second frame through the warm path, and the free-standing all-reduce
        pos2 = pos + 0.01 * np.random.default_rng(4).normal(size=pos.shape)
        out2 = mdl.predict(numbers, pos2, cell, pbc, rank=rank, world=world)
        buf = torch.arange(1000, dtype=torch.float64, device="cuda:0") * (rank + 1)
        lib = _lib.load()
        _lib.check(lib.sgpr_comm_allreduce(mdl.handle, buf.data_ptr(), buf.numel(), 0, None))
        _lib.check(lib.sgpr_sync_check(mdl.handle, None))
        s_sum = buf.cpu().numpy().copy()
        # (an odd number of words at an address that is not 16-byte aligned, with guard words either side)
        big = torch.arange(-1, 1001, dtype=torch.float64, device="cuda:0") * (rank + 1)
        buf = big[1:1000]
        assert buf.data_ptr() % 16 == 8 and buf.numel() == 999
        _lib.check(lib.sgpr_comm_allreduce(mdl.handle, buf.data_ptr(), buf.numel(), 1, None))
        _lib.check(lib.sgpr_sync_check(mdl.handle, None))
        s_max = big.cpu().numpy().copy()
        x, v, sc, code = _md(mdl, system, STEPS, [3 + rank, 7, 100], ediff=0.0)   # (every rank cuts the run differently)
        xh, vh, sch, codeh = _md(mdl, system, STEPS, [100], ediff=ediff)
        q.put((rank, out["forces"], out["beta"], out["energy"], out["stress"], out2["forces"], s_sum, s_max, x, v, sc, xh, vh, sch,
               codeh))
        dist.barrier()
        mdl.peer_destroy()
        dist.destroy_process_group()


@pytest.fixture(scope="module")
def single():
    """The single-process twins: scatter form (what the sharded run must equal bit for bit) and the default gather form."""
    mdl, system = _build(scatter=True)
    numbers, pos, cell, pbc = system
    ref = mdl.predict(numbers, pos, cell, pbc)
    pos2 = pos + 0.01 * np.random.default_rng(4).normal(size=pos.shape)
    ref2 = mdl.predict(numbers, pos2, cell, pbc)
    x, v, sc, _ = _md(mdl, system, STEPS, [100])
    # the gate: a threshold the largest covloss first reaches in the middle of the run
    bmax = sc[:, 11]
    ediff = float(bmax[5 + int(np.argmax(bmax[5:STEPS - 3]))])
    first = int(np.argmax(bmax >= ediff))
    xh, vh, sch, codeh = _md(mdl, system, STEPS, [100], ediff=ediff)
    assert codeh == 1 and len(sch) == first + 1
    mdl.close()
    g, _ = _build(scatter=False)
    gat = g.predict(numbers, pos, cell, pbc)
    xg, vg, scg, _ = _md(g, system, STEPS, [100])
    g.close()
    return dict(ref=ref, ref2=ref2, x=x, v=v, sc=sc, ediff=ediff, xh=xh, vh=vh, sch=sch, gat=gat, xg=xg, vg=vg)


@pytest.mark.parametrize("world", [2, 4, 8])
def test_exchange_and_sharded_md_equal_the_single_process_bit_for_bit(world, single):
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29700 + ((os.getpid() + 13 * world) % 250)
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, single["ediff"])) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref, ref2 = single["ref"], single["ref2"]
    fmax = np.abs(ref["forces"]).max()
    for rank, F, beta, E, stress, F2, s_sum, s_max, x, v, sc, xh, vh, sch, codeh in got:
        # one sharded step: forces and covloss bit for bit the single-process scatter form, on every rank
        np.testing.assert_array_equal(F, ref["forces"])
        np.testing.assert_array_equal(beta, ref["beta"])
        np.testing.assert_array_equal(F2, ref2["forces"])
        assert abs(E - ref["energy"]) <= 1e-12 * max(1.0, abs(ref["energy"]))
        np.testing.assert_allclose(stress, ref["stress"], rtol=0, atol=1e-12 * np.abs(ref["stress"]).max())
        # ... and the default (gather-form) single-process path to rounding
        np.testing.assert_allclose(F, single["gat"]["forces"], rtol=0, atol=1e-10 * fmax)
        # free-standing all-reduce
        np.testing.assert_array_equal(s_sum, np.arange(1000.0) * sum(range(1, world + 1)))
        np.testing.assert_array_equal(s_max[1:1000], np.arange(999.0) * world)
        assert s_max[0] == -(rank + 1.0) and s_max[1000] == 999.0 * (rank + 1) and s_max[1001] == 1000.0 * (rank + 1)
        # the sharded MD loop: the single process's trajectory, whatever the batching
        assert len(sc) == STEPS
        np.testing.assert_array_equal(x, single["x"])
        np.testing.assert_array_equal(v, single["v"])
        np.testing.assert_array_equal(sc[:, 11], single["sc"][:, 11])       # largest covloss per evaluation
        np.testing.assert_array_equal(sc[:, 12], single["sc"][:, 12])       # kinetic energy
        np.testing.assert_allclose(sc[:, 0], single["sc"][:, 0], rtol=0, atol=1e-11 * max(1.0, np.abs(single["sc"][:, 0]).max()))
        # the covloss gate: every rank halts at the evaluation the single process halts at, with its state
        assert codeh == 1 and len(sch) == len(single["sch"])
        np.testing.assert_array_equal(xh, single["xh"])
        np.testing.assert_array_equal(vh, single["vh"])
    # every rank holds the same bits (energies included: the sums run in rank order everywhere)
    for t in got[1:]:
        assert t[3] == got[0][3]
        np.testing.assert_array_equal(t[4], got[0][4])
        np.testing.assert_array_equal(t[10], got[0][10])
    # the gather-form trajectory of the default single-process loop agrees to rounding-level drift over 40 steps
    np.testing.assert_allclose(got[0][8], single["xg"], rtol=0, atol=1e-8)


def test_scatter_form_md_on_a_single_rank_equals_the_gather_form_to_rounding(single):
    """The single-rank scatter-form loop (the sharded run's twin) against the default loop: same physics, forces that differ
    in the last bits (fixed-point sums against a shuffle tree)."""
    np.testing.assert_allclose(single["x"], single["xg"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(single["v"], single["vg"], rtol=0, atol=1e-8)


def test_bench_two_ranks_over_the_exchange_on_one_gpu():
    """`bench.py --gpus 2` launched as the driver launches it, both ranks on the one GPU of the test box, the ranks' partial
    sums combined by the library's own exchange (`--collective ipc`; RCCL refuses two ranks on a device): the resident-frames
    pipeline AND the sharded device MD loop run end to end, rank 0 prints ONE line whose `value` is the MD loop's.  (The
    numbers mean nothing on a shared GPU.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29950 + ((os.getpid() + 5) % 40)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", SGPR_PEER_TIMEOUT_MS="20000")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10",
                          "--warmup", "3", "--collective", "ipc", "--no-cpu-baseline", "--no-big-wall", "--md-steps", "40"],
                         capture_output=True, text=True, timeout=900, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["collective"] == "ipc" and d["scaling"] == "strong"
    assert d["value_is"].startswith("md_loop") and d["md_loop"]["steps"] == 40
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert len(d["per_rank"]) == 2 and all(r["local_atoms"] == 2048 for r in d["per_rank"])
    assert d["allreduce_us"] is not None and d["allreduce_us"] > 0


@pytest.mark.parametrize("tdamp", [None, 20.0])
def test_on_the_fly_learning_with_the_md_state_on_two_ranks(tmp_path, tdamp):
    """BASELINE configs[4] in small: on-the-fly inducing-set updates inside NVT MD with the atoms SHARDED — the MD state on the
    devices of two ranks (both on the one GPU of the test box), the covloss gate halting both at the same step, the model
    update on the host path of every rank, Langevin and Nose-Hoover — against the single process: the same updates at the
    same steps, energies and the final state to rounding (the single process sums its forces in the gather form)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    res = {}
    for world in (1, 2):
        q = ctx.Queue()
        port = 29850 + ((os.getpid() + 17 * world + (0 if tdamp is None else 5)) % 90)
        procs = [ctx.Process(target=_otf_worker, args=(r, world, port, str(tmp_path), q, tdamp)) for r in range(world)]
        for p in procs:
            p.start()
        res[world] = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
        for p in procs:
            p.join(timeout=60)
            assert p.exitcode == 0
    ref = res[1][0]
    assert ref[4][1] > 2 and sum(1 for o in ref[1] if o[2]) >= 2         # the model grew, several updates
    for rank, out, x, v, size, on_device, peer_world in res[2]:
        assert on_device and peer_world == 2                              # the device loop ran sharded over the exchange
        assert size == ref[4]
        assert [o[0] for o in out] == [o[0] for o in ref[1]] and [o[2] for o in out] == [o[2] for o in ref[1]]
        np.testing.assert_allclose([o[1] for o in out], [o[1] for o in ref[1]], rtol=0, atol=1e-7)
        np.testing.assert_allclose(x, ref[2], rtol=0, atol=1e-7)
        np.testing.assert_allclose(v, ref[3], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(res[2][0][2], res[2][1][2])            # both ranks end in the same bits
    np.testing.assert_array_equal(res[2][0][3], res[2][1][3])


def _timeout_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import torch.distributed as dist
    from autoforce_amd import _lib
    from autoforce_amd.watchdog import Watchdog
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["SGPR_PEER_TIMEOUT_MS"] = "20000"
    with Watchdog(f"peer time-out, rank {rank} of {world}", seconds=240, rank=rank):
        dist.init_process_group("gloo", rank=rank, world_size=world)
        mdl, (numbers, pos, cell, pbc) = _build()
        N = len(numbers)

        def attach():
            blobs = [None] * world
            dist.all_gather_object(blobs, mdl.peer_export(rank, world, 7 * N + 11))
            mdl.peer_attach(blobs)
            dist.barrier()
        attach()
        ok = mdl.predict(numbers, pos, cell, pbc, rank=rank, world=world)          # the checked path
        ok2 = mdl.predict_view(numbers, pos, cell, pbc, rank=rank, world=world)    # the warm path
        same = np.array_equal(ok["forces"], ok2["forces"])
        dist.barrier()
        # rank 1 leaves the step; rank 0's WARM call must fail (not return the sums of the exchange before)
        msgs = []
        os.environ["SGPR_PEER_TIMEOUT_MS"] = "1000" if rank == 0 else "20000"
        attach()                                          # (the time-out is read at export: rank 0 now gives up after 1 s)
        mdl.predict(numbers, pos, cell, pbc, rank=rank, world=world)
        mdl.predict_view(numbers, pos, cell, pbc, rank=rank, world=world)
        dist.barrier()
        if rank == 0:
            for call in (mdl.predict_view, mdl.predict, mdl.predict_view):
                try:
                    out = call(numbers, pos + 0.01, cell, pbc, rank=rank, world=world)
                    msgs.append("returned")
                except _lib.SgprError as exc:
                    msgs.append(str(exc))
        dist.barrier()
        # a time-out is permanent until the exchange is built again: export + attach on every rank, then steps work
        attach()
        again = mdl.predict(numbers, pos, cell, pbc, rank=rank, world=world)
        again2 = mdl.predict_view(numbers, pos, cell, pbc, rank=rank, world=world)
        q.put((rank, same, msgs, np.array_equal(again["forces"], ok["forces"]), np.array_equal(again2["forces"], ok["forces"])))
        dist.barrier()
        mdl.peer_destroy()
        dist.destroy_process_group()


def test_a_rank_that_leaves_the_step_fails_the_warm_call_of_its_peer():
    """A peer that never pushes: the waiting rank's bounded wait gives up, and the call — the WARM path of sgpr_compute /
    sgpr_compute_view included, which reads only the step's own overflow word — fails with the time-out instead of returning
    the sums of the previous exchange of the same parity; so does every later call, until the exchange is exported and
    attached again (then steps are the old bits)."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + (os.getpid() % 90)
    procs = [ctx.Process(target=_timeout_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, same, msgs, again, again2 in got:
        assert same and again and again2
    msgs = got[0][2]
    assert len(msgs) == 3 and all("timed out" in m for m in msgs), msgs
