"""CPU tests of the host side: sharding rule, the reference's Distributer policy, the calculator
surface (ASE protocol, errors, log line) and the world_size=2 path over gloo with an
oracle-backed engine standing in for the GPU."""
import os
import sys

import numpy as np
import pytest

from conftest import ROOT
from helpers import OracleEngine, load


def test_shard_indices_partition_and_balance():
    from autoforce_amd.sharding import rank_of_atoms, shard_indices
    rng = np.random.default_rng(0)
    numbers = rng.choice([3, 15, 16], size=4096, p=[0.375, 0.125, 0.5])
    species = [3, 15, 16]
    for world in (1, 2, 3, 4, 8):
        parts = [shard_indices(numbers, species, r, world) for r in range(world)]
        allidx = np.sort(np.concatenate(parts))
        np.testing.assert_array_equal(allidx, np.arange(len(numbers)))  # a partition
        sizes = [len(p) for p in parts]
        assert max(sizes) - min(sizes) <= 1
        for z in species:  # per-species balance, the point of the reference's Distributer
            c = [int((numbers[p] == z).sum()) for p in parts]
            assert max(c) - min(c) <= 1
        ranks = rank_of_atoms(numbers, species, world)
        for r in range(world):
            np.testing.assert_array_equal(np.sort(np.nonzero(ranks == r)[0]), np.sort(parts[r]))


@pytest.mark.parametrize("ws", [1, 2, 4, 8])
def test_distributer_policy_matches_reference(ws):
    """theforce/descriptor/atoms.py:235-246 via the golden rank maps."""
    from autoforce_amd.sharding import distributer_ranks
    g = load("g9_distributer")
    ranks, loads, total = distributer_ranks(g["numbers"], ws)
    np.testing.assert_array_equal(ranks, g[f"ranks_{ws}"])
    ranks2, _, _ = distributer_ranks(g["numbers"][::-1], ws, loads, total)
    np.testing.assert_array_equal(ranks2, g[f"ranks2_{ws}"])


def test_unknown_species_raises():
    from autoforce_amd.sharding import species_slots
    with pytest.raises(ValueError, match="79"):
        species_slots([3, 79], [3, 16])


def _atoms(g):
    from autoforce_amd.ase_shim import Atoms
    return Atoms(g["numbers"], g["positions"], g["cell"], g["pbc"])


def test_calculator_single_process_matches_golden(tmp_path):
    from autoforce_amd.calculator import ActiveCalculator
    g = load("g5_tric24")
    eng = OracleEngine(g)
    calc = ActiveCalculator(engine=eng, logfile=str(tmp_path / "active.log"))
    atoms = _atoms(g)
    atoms.calc = calc
    e = atoms.get_potential_energy()
    f = atoms.get_forces()
    s = atoms.get_stress()
    assert eng.calls == 1  # ASE caching contract: one calculate() serves all three getters
    assert abs(e - float(g["energy"])) < 1e-11
    assert np.abs(f - g["forces"]).max() <= 1e-9 * np.abs(g["forces"]).max()
    assert np.abs(s - g["stress"]).max() <= 1e-9 * np.abs(g["stress"]).max()
    assert calc.results["free_energy"] == calc.results["energy"]
    assert calc.size == (0, len(g["ind_z"])) and calc.step == 1
    np.testing.assert_allclose(calc.cov, g["cov"], rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(calc.get_covloss(), g["covloss"], rtol=0, atol=1e-6)
    # moving an atom invalidates the cache
    atoms.positions[0, 0] += 0.01
    atoms.get_forces()
    assert eng.calls == 2
    # per-step log line "date time step energy temperature covloss" (active.py:519-523,1129-1134)
    lines = open(tmp_path / "active.log").read().strip().splitlines()
    assert lines[0].endswith("active calculator says Hello!")
    size_at = next(k for k, ln in enumerate(lines) if "model size:" in ln)
    step_line = lines[size_at + 1].split()
    assert int(step_line[2]) == 0 and abs(float(step_line[3]) - float(g["energy"])) < 1e-9
    assert abs(float(step_line[5]) - np.max(g["covloss"])) < 1e-6


def test_calculator_errors():
    from autoforce_amd.calculator import ActiveCalculator
    g = load("g5_tric24")

    from helpers import OracleModel
    calc = ActiveCalculator(engine=OracleModel(species=[3, 16]), logfile=None)
    atoms = _atoms(g)
    atoms.calc = calc
    with pytest.raises(RuntimeError, match="you forgot to assign a DFT calculator"):
        atoms.get_potential_energy()
    # no `species` in kernel_kw = the reference's wildcard kernel (table laid out from the frames met): an
    # empty device model is created, which on a box without a GPU fails loudly (no CPU fallback)
    from autoforce_amd import SgprError
    with pytest.raises(SgprError) as e:
        ActiveCalculator(covariance=None, logfile=None)
    assert e.value.code == -2


def _worker(rank, world, port, name, tmp, q):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests")]
    import torch.distributed as dist
    from helpers import OracleEngine, load
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = load(name)
    eng = OracleEngine(g, mean={int(g["species"][0]): 0.5})
    calc = ActiveCalculator(engine=eng, process_group=dist.group.WORLD, logfile=os.path.join(tmp, "active.log"))
    atoms = Atoms(g["numbers"], g["positions"], g["cell"], g["pbc"])
    atoms.calc = calc
    e = atoms.get_potential_energy()
    f = atoms.get_forces()
    s = atoms.get_stress()
    q.put((rank, e, f, s, calc.get_covloss(), calc.cov))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name", ["g5_mixed64", "g5_cluster16"])
def test_calculator_world2_gloo(name, tmp_path):
    """Two processes, atoms sharded, one all-reduce: every rank ends with the full result."""
    import torch.multiprocessing as mp
    g = load(name)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, name, str(tmp_path), q)) for r in range(2)]
    for p in procs:
        p.start()
    got = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    nz0 = int((g["numbers"] == g["species"][0]).sum())
    for rank, e, f, s, beta, cov in got:
        assert abs(e - (float(g["energy"]) + 0.5 * nz0)) < 1e-10
        assert np.abs(f - g["forces"]).max() <= 1e-9 * np.abs(g["forces"]).max()
        assert np.abs(s - g["stress"]).max() <= 1e-9 * max(np.abs(g["stress"]).max(), 1e-12)
        ok = np.isfinite(g["covloss"])
        np.testing.assert_allclose(beta[ok], g["covloss"][ok], rtol=0, atol=1e-6)
    # each rank holds only its own rows of cov; together they give K_nm
    np.testing.assert_allclose(got[0][5] + got[1][5], g["cov"], rtol=1e-10, atol=1e-13)
    # only rank 0 writes the log
    lines = open(tmp_path / "active.log").read().strip().splitlines()
    assert len(lines) == 5  # hello, kernel, settings, model size, one step


def test_watchdog_ends_a_stuck_rank():
    """autoforce_amd.watchdog: a rank blocked inside a collective set-up is reported and ended with exit code 3 (never a
    re-exec); a call that returns in time is left alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, time; sys.path.insert(0, %r); from autoforce_amd.watchdog import Watchdog\n"
            "with Watchdog('quick', seconds=5, rank=1):\n    pass\n"
            "with Watchdog('ncclCommInitRank', seconds=0.3, rank=1):\n    time.sleep(20)\n") % root
    p = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60)
    assert p.returncode == 3
    assert "rank 1" in p.stderr and "ncclCommInitRank" in p.stderr


def test_largest_covloss_is_the_head_of_the_reference_argsort():
    """update_inducing offers the first atom, in descending order of covloss, that is neither chosen nor ignored
    (calculator/active.py:851-856 walks a full argsort for it); the masked argmax that replaces the sort picks the same
    atom — ties to the lowest index, everything excluded -> the tail of the order — on random cases."""
    from autoforce_amd.calculator import ActiveCalculator

    class Stub:
        ignore = []

    rng = np.random.default_rng(0)
    for _ in range(3000):
        n = int(rng.integers(1, 12))
        beta = rng.choice([0.0, 0.1, 0.5, 0.5, 0.9, 1.0, np.inf], size=n).astype(float)
        chosen = [int(c) for c in rng.choice(n, size=int(rng.integers(0, n + 1)), replace=False)]
        Stub.ignore = [int(c) for c in rng.choice(n, size=int(rng.integers(0, 2)), replace=False)]
        order = np.argsort(-beta, kind="stable")
        want = next((int(i) for i in order if int(i) not in chosen and int(i) not in Stub.ignore), int(order[-1]))
        assert ActiveCalculator._largest_covloss(Stub, beta, chosen) == want
