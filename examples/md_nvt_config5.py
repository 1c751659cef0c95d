#!/usr/bin/env python3
"""BASELINE config 5 as worded: on-the-fly inducing-set updates at the size limit inside NVT MD — 16384-atom
4-species oxide (ordered: `autoforce_amd.workloads.oxide_ordered`), the model pre-seeded to ~1000 inducing LCEs,
max_inducing = 1024, Langevin 600 K, 1 fs.  Every update step past the limit ends in downsize(lii=True)
(calculator/active.py:963-969 -> regression/gppotential.py:829-832), which the library follows incrementally:
K_mm factor per species block, kept first-stage QR through the reflectors of R1[:, idx].

    python examples/md_nvt_config5.py --steps 250            # prints one line per step and a summary
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 examples/md_nvt_config5.py
                                                             # the config as worded: atoms sharded over 8 ranks, the MD state
                                                             # on every device (the library's own exchange)

`run()` is also what tests/test_hip_config5.py drives.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from autoforce_amd import workloads  # noqa: E402


def run(steps=250, shape=(32, 32, 16), m_seed=1000, max_inducing=1024, friction=0.1, temperature=600.0, n_exceed=8,
        verbose=False, stop_after_downsizes=None, min_steps=0, device_md=True, host_rng=False, process_group=None, device=0):
    """friction: per ASE time unit.  The reference CLI's default, 1e-3 (cl/md.py:31), is a 10-ps coupling: invisible
    in a few hundred fs.  The default here (0.1: 0.1 ps) lets a short run show whether the thermostat HOLDS the
    temperature while the model is edited under it."""
    t_setup = time.time()
    extra = {} if process_group is None else dict(process_group=process_group)
    calc, teacher, (numbers, pos, cell, pbc), vel = workloads.config5_preseeded(shape, m_seed, max_inducing, n_exceed, device=device,
                                                                                temperature=temperature, friction=friction, **extra)
    np.random.seed(1)
    model = calc.model
    stats = dict(downsizes=0, downsize_ms=[], routes=[], pending=False)
    inner, inner_munu = model.downsize, model.make_munu

    def counted(*a, **k):
        t0 = time.time()
        ch1, ch2 = inner(*a, **k)
        if ch2:
            stats["downsizes"] += 1
            stats["downsize_ms"].append(1e3 * (time.time() - t0))
            stats["pending"] = True
        return ch1, ch2

    def munu(*a, **k):
        out = inner_munu(*a, **k)
        if stats["pending"]:  # the refit that follows a downsize: which route did its first stage take?
            stats["routes"].append(model.engine.solve_info())
            stats["pending"] = False
        return out

    model.downsize, model.make_munu = counted, munu
    setup_s = time.time() - t_setup
    rows = []
    last = None

    def host_loop():
        for step, E, T, wall, p, v in workloads.langevin_nvt(calc, numbers, pos, cell, pbc, steps, temperature, 1.0, friction, vel=vel):
            yield step, E, T, wall, p, bool(calc.updated)

    def device_loop():
        # the same scheme with positions and velocities in device memory (host_rng: also the same random stream as
        # workloads.langevin_nvt, drawn from default_rng(1) here and uploaded; default: drawn on the device): prediction-only steps never leave the device, a step whose covloss reaches the
        # sampling threshold is handed to calculate() (ActiveCalculator.run_md); the steps of a device batch share its wall time
        from autoforce_amd.ase_shim import Atoms
        at = Atoms(numbers, pos, cell, pbc, velocities=vel, masses=np.array([workloads.MASS[int(z)] for z in numbers]))
        rng = np.random.default_rng(1) if host_rng else None   # None: the deviates are drawn on the device
        for step, E, T, upd, wall in calc.run_md(at, steps, temperature, dt_fs=1.0, friction=friction, rng=rng, chunk=64, seed=1):
            yield step, E, T, wall, at.positions, upd

    dbg = os.environ.get("C5_DEBUG")   # diagnostic: fingerprints of the model after every update, per rank
    for step, E, T, wall, p, updated in (device_loop() if device_md else host_loop()):
        if dbg and updated:
            eng = model.engine
            eng._choli, eng._choli_on_device = None, True
            ch = eng.choli
            print(f"[c5dbg] rank {os.environ.get('RANK', '0')} step {step} m {eng.m} mu {float(np.sum(eng.mu)):.17g} "
                  f"choli {float(np.abs(ch).sum()):.17g} M {float(np.abs(eng.M).sum()):.17g} E {E:.17g}", flush=True)
        rows.append(dict(step=step, E=E, T=T, wall=wall, size=calc.size, updated=updated, covloss=calc.covlog,
                         teacher_s=teacher.seconds, downsizes=stats["downsizes"], info=model.engine.solve_info()))
        if len(rows) > 1:
            rows[-1]["teacher_ms"] = 1e3 * (rows[-1]["teacher_s"] - rows[-2]["teacher_s"])
        else:
            rows[-1]["teacher_ms"] = 1e3 * teacher.seconds
        if verbose:
            r = rows[-1]
            print(f"{step:5d} E={E:14.6f} T={T:7.1f} covloss={str(r['covloss'])[:8]:>8s} size={r['size']} upd={int(r['updated'])} "
                  f"wall={1e3 * wall:8.1f} ms teacher={r['teacher_ms']:7.1f} ms downsizes={r['downsizes']}  {r['info']}", flush=True)
        last = p
        if stop_after_downsizes and stats["downsizes"] >= stop_after_downsizes and step >= min_steps:
            break
    if device_md and getattr(model.engine, "_md", None) is not None:   # (the configuration the device holds now: the end of the last batch)
        last = model.engine.md_state()["positions"]
    return dict(calc=calc, teacher=teacher, system=(numbers, last, cell, pbc), rows=rows, stats=stats, setup_s=setup_s)


def verify(res, tol_choli=1e-9, tol_fit=1e-7):
    """The edited model against the same model set up and factored from scratch on the device: K_mm bit for bit,
    choli, and the fit (K mu) of the same targets at the same noise."""
    model = res["calc"].model
    eng = model.engine
    ref = eng.scratch()
    ref.set_inducing(eng.X)
    out = {}
    out["kmm_equal"] = bool(np.array_equal(eng.M, ref.M))
    for fr in model.data:
        ref.data_push(*fr.system(), fr.nv)
    Y = model._store_targets()
    noise = 1.0 / (1.0 + np.exp(-model._noise["all"]))
    mu_ref = ref.data_solve(Y, with_energies=True, noise=noise)
    mu_now = eng.data_solve(Y, with_energies=True, noise=noise)
    out["route"] = eng.solve_info()
    c0, c1 = ref.choli, eng.choli
    out["choli_err"] = float(np.abs(c0 - c1).max() / np.abs(c0).max())
    p0, p1 = ref.data_matvec(mu_ref), eng.data_matvec(mu_now)
    out["fit_err"] = float(np.abs(p0 - p1).max() / np.abs(p0).max())
    out["ridge"] = (ref.ridge, eng.ridge)
    ref.close()
    assert out["kmm_equal"], "K_mm differs from a rebuild"
    assert out["choli_err"] <= tol_choli, out
    assert out["fit_err"] <= tol_fit, out
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=250)
    ap.add_argument("--side", type=int, nargs=3, default=[32, 32, 16])
    ap.add_argument("--m-seed", type=int, default=1000)
    ap.add_argument("--max-inducing", type=int, default=1024)
    ap.add_argument("--friction", type=float, default=0.1)
    ap.add_argument("--n-exceed", type=int, default=8, help="ediff = the n-th largest covloss of a held-out equilibrated frame")
    ap.add_argument("--host-loop", action="store_true", help="integrate in numpy, one calculate() per step (the round-3 driver)")
    ap.add_argument("--host-rng", action="store_true", help="device loop with numpy's deviates uploaded (the host loop's trajectory, bit for bit)")
    args = ap.parse_args()
    t0 = time.time()
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    group, device = None, 0
    if world > 1:   # launched by torch.distributed.run: one rank per GPU (several per GPU when there are fewer GPUs than ranks)
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        group, device = dist.group.WORLD, int(os.environ.get("LOCAL_RANK", "0")) % max(torch.cuda.device_count(), 1)
    res = run(args.steps, tuple(args.side), args.m_seed, args.max_inducing, args.friction, n_exceed=args.n_exceed, verbose=rank == 0,
              device_md=not args.host_loop, host_rng=args.host_rng, process_group=group, device=device)
    if rank != 0:
        verify(res)
        dist.barrier()
        return
    rows = res["rows"]
    upd = [1e3 * r["wall"] - r["teacher_ms"] for r in rows[1:] if r["updated"]]
    capped = [1e3 * r["wall"] - r["teacher_ms"] for r in rows[1:] if r["updated"] and r["size"][1] >= args.max_inducing]
    quiet = [1e3 * r["wall"] for r in rows[1:] if not r["updated"]]
    print(f"# set-up {res['setup_s']:.1f} s, {len(rows) - 1} steps in {time.time() - t0 - res['setup_s']:.1f} s; final size {res['calc'].size}; "
          f"downsizes {res['stats']['downsizes']} (median {np.median(res['stats']['downsize_ms'] or [0]):.1f} ms each)")
    if upd:
        print(f"# model-update steps: {len(upd)}, median {np.median(upd):.1f} ms excluding the teacher; at m = max_inducing: "
              f"{len(capped)} steps, median {np.median(capped or [0]):.1f} ms")
    if quiet:
        print(f"# prediction-only steps: {len(quiet)}, median {np.median(quiet):.2f} ms")
    T = np.array([r["T"] for r in rows])
    print(f"# temperature over the last 100 steps: mean {T[-100:].mean():.1f} K, min {T[-100:].min():.1f}, max {T[-100:].max():.1f}")
    print("# against a from-scratch model:", verify(res))
    if world > 1:
        eng = res["calc"].engine
        how = ("the library's own exchange" if getattr(eng, "peer_world", 1) == world else
               "RCCL" if getattr(eng, "comm_world", 1) == world else "host-staged")
        print(f"# {world} ranks: collective = {how}; MD state on the devices: {res['calc'].md_on_device_ok()}")
        dist.barrier()


if __name__ == "__main__":
    main()
