"""GPU: the sharded calculator on the HIP engine with TWO ranks (both on the one GPU of the test box).

Two ways of combining the ranks' partial sums are exercised.  (1) The library's own exchange through hipIpc-mapped
buffers (SGPR_COLLECTIVE=auto, the default): it runs with two ranks on one device.  (2) SGPR_COLLECTIVE=rccl: RCCL refuses
two ranks on one device (ncclCommInitRank: invalid usage), so the native communicator cannot be built here — which is
exactly the start-up failure a multi-GPU job must survive: every rank agrees that it failed
(calculator.py::_attach_native_comm) and the run continues with the host-side all-reduce of the packed partial
sums.  Either way the on-the-fly learning loop goes through everything a sharded run does — initiate_model and
get_unique_lces (whole-frame, replicated evaluations between sharded ones), LCEs handed out by their owner, rank 0's
solve broadcast — and must take the decisions of a single process, with its energies and forces."""
import os

import numpy as np
import pytest

import active_common as ac

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, tmp, q, mode):
    import sys
    os.environ["SGPR_COLLECTIVE"] = mode
    os.environ["SGPR_PEER_TIMEOUT_MS"] = "20000"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path[:0] = [root, os.path.join(root, "tests")]
    import pathlib

    import torch.distributed as dist
    import active_common as ac2
    from autoforce_amd import SGPRModel
    from autoforce_amd.watchdog import Watchdog
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    with Watchdog("two-rank learning loop", seconds=240, rank=rank):
        dist.init_process_group("gloo", rank=rank, world_size=world)
        d = pathlib.Path(tmp) / f"rank{rank}"
        d.mkdir()
        eng = SGPRModel(3, 3, 4, 4.5, species=ac2.SPECIES)
        calc, teacher, trace = ac2.run(eng, d, steps=4, tape=(rank == 0), process_group=dist.group.WORLD)
        log = open(d / "active.log").read() if rank == 0 else ""
        q.put((rank, [t[0] for t in trace], [t[1] for t in trace], trace[-1][2], calc.model.mu,
               int(getattr(calc.engine, "comm_world", 1)), log, int(getattr(calc.engine, "peer_world", 1))))
        dist.barrier()
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["auto", "rccl"])
def test_sharded_learning_loop_two_ranks_one_gpu(tmp_path, mode):
    import torch.multiprocessing as mp
    from autoforce_amd import SGPRModel
    (tmp_path / "single").mkdir()
    _, _, ref = ac.run(SGPRModel(3, 3, 4, 4.5, species=ac.SPECIES), tmp_path / "single", steps=4)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + ((os.getpid() + (11 if mode == "auto" else 211)) % 500)
    procs = [ctx.Process(target=_worker, args=(r, 2, port, str(tmp_path), q, mode)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, sizes, energies, forces, mu, comm_world, log, peer_world in got:
        assert sizes == [t[0] for t in ref]
        np.testing.assert_allclose(energies, [t[1] for t in ref], rtol=0, atol=1e-8)
        np.testing.assert_allclose(forces, ref[-1][2], rtol=0, atol=1e-8)
        if mode == "auto":
            assert comm_world == 2 and peer_world == 2   # the library's own exchange: two ranks on one device are fine
        else:
            # one device for two ranks: the RCCL communicator cannot exist, and every rank knows
            assert comm_world == 1 and peer_world == 1
    if mode == "rccl":
        assert "host-side all-reduce instead" in got[0][6]
    np.testing.assert_array_equal(got[0][4], got[1][4])


def test_bench_two_ranks_on_one_gpu_dry_run():
    """`bench.py --gpus 2` launched as the driver launches it (torch.distributed.run, one rank per process), both ranks on
    the one GPU of the test box with the host-side collective (`--collective torch`: RCCL refuses two ranks on a device):
    the sharded path of the benchmark runs end to end and rank 0 prints ONE line with the whole-job value, `scaling`
    "strong" (the frame does not grow with N) and one `per_rank` record per rank.  (The numbers mean nothing on a shared GPU.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    port = 29600 + ((os.getpid() + 37) % 300)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                          "127.0.0.1", "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "10",
                          "--warmup", "3", "--collective", "torch", "--no-cpu-baseline", "--no-big-wall", "--md-steps", "0"],
                         capture_output=True, text=True, timeout=600, cwd=root, env=env)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 10 and d["warmup"] == 3 and d["scaling"] == "strong"
    assert abs(d["value"] - 4096 / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert len(d["per_rank"]) == 2 and sorted(r["rank"] for r in d["per_rank"]) == [0, 1]
    assert all(r["local_atoms"] == 2048 for r in d["per_rank"])
    assert "x2" in d["config"]["parallelism"]
