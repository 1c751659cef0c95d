#!/usr/bin/env python3
"""bench.py — MD-step throughput of the SGPR predict hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one frame: device neighbour list -> SeSoap descriptors
-> K_nm -> energy, forces, virial -> covloss (calculator/active.py:425-502 of the reference),
with positions already resident in HBM.  Workload at N=1: BASELINE configs[2] — 4096-atom
"LiPS" (3 species), 512 inducing points, lmax=nmax=3, eta=4, rc=6 A, fp64.
N > 1: atoms are dealt to ranks (per-species round robin, the reference's Distributer); every
rank evaluates its share and ONE exchange of the packed [F | beta | E | virial] buffer combines them
(the reference's four MPI all-reduces, calculator/active.py:562,601,602,777).  The exchange is issued
by libsgpr_hip itself on the step's stream: by default its own all-gather over hipIpc-mapped buffers
(sgpr_peer_attach: one hop on the xGMI mesh, summed in rank order — the same bits for every N; it also
runs with several ranks on ONE GPU), else one RCCL all-reduce (sgpr_comm_init; --collective native).
No PyTorch op on the step.  torch.distributed (gloo, CPU) only carries the handles / the 128-byte
communicator id at start-up and the barrier / max-over-ranks around the timed region.  The frame is the
same for every N, so scaling is "strong".

torch is used for device memory and that bootstrap only; the numerics are libsgpr_hip.so
(hand-written HIP) called through its C ABI.

Reported besides `value` (device-resident pipeline, K steps enqueued back to back):
`calculate_wall`: median wall time of one ActiveCalculator.calculate() on the same frame — numpy
positions in, numpy results out, synchronised every call — SURVEY.md section 8(d)'s definition.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
MIN_TIMED_MS = 50.0       # the --steps batch is repeated until this much has been timed; the median batch is reported
FP64_MFMA_PEAK_TF = 78.6  # BASELINE.md §4 (MI355X FP64 matrix peak)


def csrc_sha():
    """Hash of the kernel sources: a committed PMC summary is only quoted while it describes THESE kernels."""
    import glob
    import hashlib
    hsh = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "autoforce_amd", "csrc", "*.hip")) +
                    glob.glob(os.path.join(ROOT, "autoforce_amd", "csrc", "*.inc")) +
                    glob.glob(os.path.join(ROOT, "autoforce_amd", "csrc", "*.h"))):
        hsh.update(open(f, "rb").read())
    return hsh.hexdigest()[:16]


def latest_profile(pattern):
    import glob
    c = sorted(glob.glob(os.path.join(ROOT, "profiles", pattern)))
    return c[-1] if c else None


def build_model(device, numbers, pos, cell, pbc, m, workload_seed=1):
    from autoforce_amd import SGPRModel
    from autoforce_amd.workloads import inducing_from_frame, lips
    species = sorted(set(int(z) for z in numbers))
    mdl = SGPRModel(3, 3, 4, 6.0, species=species, device=device)
    n2, p2, c2, b2 = lips(round(len(numbers) ** (1 / 3)), seed=workload_seed)
    X = inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=workload_seed)
    mdl.set_inducing(X)
    # regression state: choli from the device solve on a small synthetic data block, mu ~ N(0,1)
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(64, m)), rng.normal(size=64))
    mu = rng.normal(size=m)
    mdl.set_weights(mu, choli=mdl.choli, vscale=mdl.make_vscale())
    return mdl


def survey_step_bytes(N, nn, D, m):
    """SURVEY.md section 8(d), literally: neighbour gather forward + backward (44 B per pair each), force
    scatter (24 B per pair), p^ write + read and W write + read (4 x 8ND), P^m read, K_nm write + read,
    force write."""
    return N * nn * 44 * 2 + N * nn * 24 + 4 * (8 * N * D) + 8 * m * D + 2 * (8 * N * m) + 24 * N


def algorithmic_bytes(N, nn, D, m, cs):
    """The same terms split by the kernel of THIS formulation that moves them (D = packed row length,
    cs = doubles of c per atom), plus what the formulation adds: the 40-B bin record per atom, the list
    (8 B), pair record (32 B) and pair gradient (32 B) per pair, c per atom."""
    return {
        "neighbor_bin": N * (24 + 24 + 40),
        "list_forward": N * nn * 44 + 8 * N * D + N * nn * (8 + 32) + 8 * N * cs,
        "gemm_knm": 8 * N * D + 8 * m * D + 8 * N * m,
        "gemm_w_covloss": 8 * N * m + 8 * m * D + 8 * N * D + 8 * N * m + 8 * m * m,
        "gemm_w": 8 * N * m + 8 * m * D + 8 * N * D,
        "descriptor_rev": 2 * 8 * N * D + 8 * N * cs + N * nn * 32 + N * nn * 32 + 24 * N,
        # (the covloss tiles ride in the reverse kernel's launch: K and choli read once more, row sums out)
        "descriptor_rev_covloss": 2 * 8 * N * D + 8 * N * cs + N * nn * 32 + N * nn * 32 + 24 * N + 8 * N * m + 8 * m * m,
        "finalize": N * nn * 32 + 24 * N + 32 * N,
        # the last kernel also opens the next step: next positions in, sorted copy + bin record out
        "finalize_bin_next": N * nn * 32 + 24 * N + 32 * N + N * (24 + 24 + 40),
    }


def algorithmic_flops(atom_z, ind_z, Dpad, world=1):
    """Dense fp64 flops the block-diagonal (species-sorted) formulation needs per launch of the two GEMM
    kernels: K_nm = P^n.P^m^T and W = Aw.P^m are 2 n_s m_s Dpad per species block; the covloss product
    K.choli^T is triangular inside a block: n_s m_s^2."""
    knm = w = cov = 0.0
    for z in np.unique(ind_z):
        n_s = float((np.asarray(atom_z) == z).sum()) / world
        m_s = float((np.asarray(ind_z) == z).sum())
        knm += 2.0 * n_s * m_s * Dpad
        w += 2.0 * n_s * m_s * Dpad
        cov += n_s * m_s * m_s
    return {"gemm_knm": knm, "gemm_w_covloss": w + cov, "gemm_w": w, "covloss": cov}


def calculate_wall_big(device, sigma):
    """BASELINE configs[4] size through the drop-in surface: 16384-atom 4-species oxide, 1024 inducing LCEs — the
    calculate() wall (numpy in, numpy out, synchronised) next to the device time of the same step."""
    import torch
    from autoforce_amd import SGPRModel, _lib
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    from autoforce_amd.workloads import inducing_from_frame, oxide
    numbers, pos, cell, pbc = oxide(seed=0)
    species = sorted(set(int(z) for z in numbers))
    mdl = SGPRModel(3, 3, 4, 6.0, species=species, device=device)
    n2, p2, c2, b2 = oxide(seed=1)
    mdl.set_inducing(inducing_from_frame(mdl, n2, p2, c2, b2, 1024, seed=1))
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(64, 1024)), rng.normal(size=64))
    mdl.set_weights(rng.normal(size=1024), choli=mdl.choli, vscale=mdl.make_vscale())
    calc = ActiveCalculator(covariance=mdl, logfile=None)
    atoms = Atoms(numbers, pos.copy(), cell, pbc)
    atoms.calc = calc
    walls = []
    for it in range(35):
        atoms.positions = atoms.positions + sigma * rng.normal(size=pos.shape)
        tc = time.perf_counter()
        atoms.get_forces()
        walls.append(time.perf_counter() - tc)
    w = float(np.median(walls[5:]))
    # the same step device-resident: K steps enqueued back to back
    lib, h, N = _lib.load(), mdl.handle, len(numbers)
    dev = torch.device("cuda", device)
    pos_d = torch.from_numpy(atoms.positions).to(dev)
    cell_d = torch.from_numpy(cell).to(dev)
    packed = torch.zeros(int(lib.sgpr_packed_len(N)), dtype=torch.float64, device=dev)
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    for _ in range(3):
        _lib.check(lib.sgpr_step_dev(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), sp))
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(30):
        _lib.check(lib.sgpr_step_dev(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), sp))
    torch.cuda.synchronize(dev)
    dstep = (time.perf_counter() - t0) / 30
    # roofline of the same frame: per-kernel hip-event times of eager steps, the dominant kernel against its bound
    mdl.profile(True)
    acc = {}
    for _ in range(16):
        _lib.check(lib.sgpr_step_dev(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), sp))
        torch.cuda.synchronize(dev)
        for k, v in mdl.stage_times().items():
            acc.setdefault(k, []).append(v)
    mdl.profile(False)
    stage_ms = {k: float(np.median(v)) for k, v in acc.items() if len(v) >= 8}
    dims = mdl.dims
    ptr, _, _ = mdl.neighbors(N)
    nn_mean = float(ptr[-1]) / N
    ab = algorithmic_bytes(N, nn_mean, dims["Dc"], 1024, dims["S"] * 64)
    af = algorithmic_flops(numbers, [x.number for x in mdl.X], dims["Dpad"], 1)
    dom = max((k for k in stage_ms if k in ab), key=lambda k: stage_ms[k])
    dom_s = stage_ms[dom] * 1e-3
    if dom in af:
        head = {"bound": "mfma", "achieved": af[dom] / dom_s / 1e12, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                "frac": af[dom] / dom_s / 1e12 / FP64_MFMA_PEAK_TF, "algorithmic_flops": af[dom]}
    else:
        head = {"bound": "hbm", "achieved": ab[dom] / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ab[dom] / dom_s / 1e9 / HBM_PEAK_GBS}
    traffic, traffic_source = None, "null: no PMC pass was committed for this frame"
    tpath = latest_profile("r*_pmc_traffic_oxide16384_m1024.json")
    if tpath:
        tj = json.load(open(tpath))
        if tj.get("csrc_sha") == csrc_sha():
            traffic = tj["kernels"].get(dom, {}).get("fetch_x2_plus_write")
            traffic_source = f"profiles/{os.path.basename(tpath)} ({tj.get('source', 'builder run')})"
        else:
            traffic_source = f"null: profiles/{os.path.basename(tpath)} was collected on other kernel sources"
    roof = {"kernel": dom, **head, "traffic": traffic, "traffic_source": traffic_source, "algorithmic_bytes": ab[dom],
            "avg_launch_us": stage_ms[dom] * 1e3,
            "timing": "hip events on the launch stream, 16 eager steps (marker overhead ~1 us per stage not subtracted)",
            "stage_us": {k: round(v * 1e3, 2) for k, v in stage_ms.items()},
            "gemm_TFLOPs": {k: round(af[k] / (stage_ms[k] * 1e-3) / 1e12, 2) for k in af if stage_ms.get(k)},
            "gemm_frac": {k: round(af[k] / (stage_ms[k] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF, 4) for k in af if stage_ms.get(k)},
            "hbm_GBs": {k: round(ab[k] / (stage_ms[k] * 1e-3) / 1e9, 1) for k in ab if stage_ms.get(k)},
            "mean_neighbors": round(nn_mean, 2)}
    mdl.close()
    return {"median_ms": w * 1e3, "atom_steps_per_s": N / w, "device_step_ms": dstep * 1e3, "wall_over_device": w / dstep,
            "atoms": N, "inducing": 1024, "calls": len(walls) - 5, "roofline": roof}


def cpu_baseline(numbers, pos, cell, pbc, mdl, mu, sample_atoms, min_seconds=12.0):
    """Times the CPU oracle (C/OpenMP restatement of the reference path, pinned by the golden
    vectors) on a bounded sample of the SAME frame: descriptors + K_nm + reverse pass + covloss for
    `sample_atoms` atoms (forces scatter to all atoms), all host cores."""
    from oracle import oracle as orc
    species = np.array(mdl.species, np.int32)
    X = mdl.X
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(3, 3, 6.0, species, ind_z, ind_ptr, np.concatenate([x._b for x in X]),
                                       np.concatenate([x._r for x in X]))
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < min_seconds:
        # the whole path on the CPU: linked-cell neighbour list of the frame, then descriptors + K_nm +
        # reverse pass + covloss of the first `sample_atoms` atoms' environments
        ptr, j, off = orc.neighbors_cells(pos, cell, pbc, 6.0)
        ptr_s = ptr.copy()
        ptr_s[sample_atoms + 1:] = ptr_s[sample_atoms]
        orc.frame(3, 3, 6.0, 4.0, species, numbers, pos, cell, (ptr_s, j, off), ind_z, nnm, Pm, mu, choli=mdl.choli,
                  want_p=False)
        reps += 1
        dt = time.perf_counter() - t0
    return sample_atoms * reps / dt, dt, reps, orc.num_threads()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--atoms-side", type=int, default=16, help="simple-cubic sites per edge (16 -> 4096 atoms)")
    ap.add_argument("--inducing", type=int, default=512)
    ap.add_argument("--graph", action="store_true", help="replay the step from a captured HIP graph")
    ap.add_argument("--walk-sigma", type=float, default=0.012,
                    help="the atoms move every step, as in MD: Gaussian random walk, this sigma per component per step "
                         "(A; 600 K thermal velocities are 0.007-0.015 A per 1 fs step here); 0 = static frame")
    ap.add_argument("--walk-frames", type=int, default=64, help="frames of the walk (traversed forth and back)")
    ap.add_argument("--skin", type=float, default=0.5, help="Verlet skin of the neighbour candidates, A (0 = rebuild every step)")
    ap.add_argument("--overlap", type=int, default=0, help="1: covloss GEMM on a side stream next to the reverse pass")
    ap.add_argument("--fuse-next", type=int, default=1, help="0: every step bins for itself (sgpr_step_dev), 6 launches per step")
    ap.add_argument("--md-steps", type=int, default=400, help="steps of the device-resident Langevin loop behind `md_loop` (0: skip)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-big-wall", action="store_true", help="skip the calculate() wall time of the 16384-atom / 1024 frame")
    ap.add_argument("--collective", default="auto", choices=["auto", "ipc", "native", "torch"],
                    help="auto (default): the library's own exchange through hipIpc-mapped buffers (all-gather of the ranks' "
                         "partial sums + local sum in rank order: deterministic, also runs with several ranks on one device, the "
                         "MD loop runs sharded on it), else RCCL, else host staged; ipc / native (= RCCL all-reduce on the step's "
                         "stream) / torch (torch.distributed all_reduce of the packed buffer, host staged): only that one")
    ap.add_argument("--cpu-sample", type=int, default=0, help="atoms in the CPU-baseline sample (0 = auto)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from autoforce_amd import _lib
    from autoforce_amd.workloads import lips

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (libsgpr_hip has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()  # dry runs may put several ranks on one GPU (gloo only)
    torch.cuda.set_device(local_rank)
    from autoforce_amd.watchdog import Watchdog
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("NCCL_DEBUG", "WARN")  # RCCL's own diagnostics of a failed start-up reach stderr
        # CPU-side group: carries the communicator id at start-up and the barriers / max around the timed
        # region; the step's collective is the library's own (RCCL over xGMI)
        dist.init_process_group("gloo", rank=rank, world_size=world)

    numbers, pos, cell, pbc = lips(args.atoms_side, seed=0)
    N, m = len(numbers), args.inducing
    mdl = build_model(local_rank, numbers, pos, cell, pbc, m)
    lib = _lib.load()
    h = mdl.handle
    _lib.check(lib.sgpr_set_option(h, b"graph", 1 if args.graph else 0))
    _lib.check(lib.sgpr_set_option(h, b"overlap", args.overlap))
    _lib.check(lib.sgpr_set_option(h, b"skin_milliangstrom", int(round(1000 * args.skin))))

    dev = torch.device("cuda", local_rank)
    # MD-like input: a Gaussian random walk away from the frame and back (closed loop of 2W - 2 frames, resident
    # in HBM), so that the atoms move every step and the neighbour candidates are rebuilt at the rate an MD run
    # would rebuild them; step k reads frame k of the loop (a pointer, no op on the step)
    W = max(args.walk_frames, 1) if args.walk_sigma > 0 else 1
    wrng = np.random.default_rng(7)
    walk = [pos]
    for _ in range(W - 1):
        walk.append(walk[-1] + args.walk_sigma * wrng.normal(size=pos.shape))
    loop = walk + walk[-2:0:-1]
    frames_d = torch.from_numpy(np.stack(loop)).to(dev)
    nframes = len(loop)
    pos_d = frames_d[0]
    cell_d = torch.from_numpy(cell).to(dev)
    packed = torch.zeros(int(lib.sgpr_packed_len(N)), dtype=torch.float64, device=dev)
    _lib.check(lib.sgpr_bind_system(h, N, _lib.ptr(_lib.i32(numbers)), _lib.ptr(_lib.i32(pbc.astype(np.int32))),
                                    rank, world))
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)
    backend = "none"
    if world > 1 and args.collective in ("auto", "ipc"):
        # the library's own exchange: every rank exports its receive buffers, the handles travel over the CPU group,
        # everybody maps everybody's; the outcome is agreed on (MIN over ranks) before anyone steps
        ok, blob = 1, None
        try:
            blob = mdl.peer_export(rank, world, 7 * N + 11)
        except Exception as exc:  # noqa: BLE001
            print(f"[bench] rank {rank}: sgpr_peer_export failed ({exc})", file=sys.stderr, flush=True)
            ok = 0
        blobs = [None] * world
        dist.all_gather_object(blobs, blob)
        if ok and all(b is not None for b in blobs):
            try:
                mdl.peer_attach(blobs)
            except Exception as exc:  # noqa: BLE001
                print(f"[bench] rank {rank}: sgpr_peer_attach failed ({exc})", file=sys.stderr, flush=True)
                ok = 0
        else:
            ok = 0
        t = torch.tensor([ok])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 1:
            # one small exchange with a known answer on every rank before the bench relies on it (a peer whose stores are
            # not seen is a time-out here, once, not a lost measurement)
            with Watchdog("self-test of the exchange between the ranks", seconds=120, rank=rank):
                if not mdl.peer_selftest(rank, world):
                    print(f"[bench] rank {rank}: the exchange's self-test failed", file=sys.stderr, flush=True)
                    ok = 0
            t = torch.tensor([ok])
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            if blob is not None:
                mdl.peer_destroy()
        else:
            backend = "ipc"
    native = world > 1 and backend == "none" and args.collective in ("auto", "native")
    if native:
        # every rank takes the same branch: the outcome of the attempt is agreed on before anyone steps
        ok = 1
        try:
            box = [mdl.comm_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0)
            # a rank that dies before it gets here would leave the others inside ncclCommInitRank for ever: the
            # watchdog names the stuck rank and ends the process with exit code 3 (never a re-exec)
            with Watchdog("sgpr_comm_init (ncclCommInitRank)", seconds=180, rank=rank):
                mdl.comm_init(box[0], rank, world)  # collective: ncclCommInitRank on every rank
        except Exception as exc:  # noqa: BLE001 - report and fall back rather than lose the measurement
            print(f"[bench] rank {rank}: native RCCL communicator failed ({exc}); host-staged all-reduce instead",
                  file=sys.stderr, flush=True)
            ok = 0
        t = torch.tensor([ok])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        if int(t.item()) == 0:
            native = False
            if ok:
                mdl.comm_destroy()
        else:
            backend = "rccl"
    native = backend != "none"   # the library combines the ranks' partial sums itself

    frame_ptr = [frames_d[k].data_ptr() for k in range(nframes)]
    counter = [0]

    def step():
        # with a communicator attached the step ends with the all-reduce of `packed`, same stream.
        # The frames are resident, so the step can name the one that follows: its last kernel then bins it and the next
        # step starts with the list filter (single rank; 5 launches per step instead of 6)
        k = counter[0] % nframes
        counter[0] += 1
        if args.fuse_next:   # (sharded ranks too: the scatter-form last kernel bins the next frame, the all-reduce follows it)
            _lib.check(lib.sgpr_step_dev_next(h, frame_ptr[k], cell_d.data_ptr(), packed.data_ptr(), frame_ptr[(k + 1) % nframes], sp))
        else:
            _lib.check(lib.sgpr_step_dev(h, frame_ptr[k], cell_d.data_ptr(), packed.data_ptr(), sp))
        if world > 1 and not native:
            t = packed.cpu()
            dist.all_reduce(t)
            packed.copy_(t)

    warm_batches = []
    with Watchdog("warm-up steps (the first collective of every rank)", seconds=300, rank=rank):
        for _ in range(max(args.warmup, 2)):
            step()
        _lib.check(lib.sgpr_sync_check(h, sp))
        # ... and on until the step time has settled (clocks, caches and the first list rebuilds of a fresh process take
        # tens of steps; --warmup is the minimum).  Every rank runs the same number of batches (the decision is rank 0's).
        for _ in range(80):
            torch.cuda.synchronize(dev)
            tb = time.perf_counter()
            for _ in range(20):
                step()
            torch.cuda.synchronize(dev)
            warm_batches.append((time.perf_counter() - tb) / 20)
            # (at least ten batches — the clock of a fresh process keeps rising for some 150 steps, slowly enough for three
            # batches in a row to agree long before it has settled; then three within 1.5 % of each other, none of them more
            # than 1.5 % above the fastest batch so far)
            stop = (len(warm_batches) >= 10 and max(warm_batches[-3:]) <= 1.015 * min(warm_batches[-3:])
                    and max(warm_batches[-3:]) <= 1.015 * min(warm_batches))
            if world > 1:
                t = torch.tensor([1 if stop else 0])
                dist.broadcast(t, src=0)
                stop = bool(t.item())
            if stop:
                break
        _lib.check(lib.sgpr_sync_check(h, sp))

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    def max_over_ranks(x):
        if world > 1:
            t = torch.tensor([x], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return x

    def batches_of(first_dt):
        """A batch = exactly --steps steps between two fences (the contract's timed region).  One batch of 20 steps is
        under 2 ms: the batch is REPEATED until >= MIN_TIMED_MS of steps have been timed and the MEDIAN batch is reported
        (min / max beside it).  The count follows from the first batch's max-over-ranks time: the same on every rank."""
        return int(min(400, max(1, np.ceil(MIN_TIMED_MS * 1e-3 / max(first_dt, 1e-9)))))

    def resident_batch():
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        return max_over_ranks(time.perf_counter() - t0)

    rb0 = C.c_int64(0)
    _lib.check(lib.sgpr_get_list_rebuilds(h, C.addressof(rb0)))
    with Watchdog("timed steps", seconds=max(600, int(0.05 * args.steps)), rank=rank):
        res_dts = [resident_batch()]
        for _ in range(batches_of(res_dts[0]) - 1):
            res_dts.append(resident_batch())
    rb1 = C.c_int64(0)
    _lib.check(lib.sgpr_get_list_rebuilds(h, C.addressof(rb1)))
    _lib.check(lib.sgpr_sync_check(h, sp))
    dt = float(np.median(res_dts))
    ms_per_step = dt / args.steps * 1e3
    value = N * args.steps / dt
    out_host = packed.cpu().numpy()  # the reduced result of the last timed step

    # ---- per-kernel durations: HIP events on the launch stream, same steps run eagerly
    mdl.profile(True)
    acc = {}
    nprof = max(min(args.steps, 50), 30)
    for _ in range(nprof):
        step()
        torch.cuda.synchronize(dev)
        for k, v in mdl.stage_times().items():
            acc.setdefault(k, []).append(v)
    mdl.profile(False)
    # (medians: one stalled launch in fifty — a rebuild step, a clock change — would otherwise sit in the mean of its stage)
    stage_ms = {k: float(np.median(v)) for k, v in acc.items() if len(v) >= nprof // 2}
    # An event-to-event interval holds one launch's marker/dispatch overhead besides the kernel.
    # Calibration: the same steps without the markers (and without the collective) take t_plain;
    # the markers therefore cost (sum of intervals - t_plain) / n_stages per stage.
    torch.cuda.synchronize(dev)
    tp = time.perf_counter()
    for _ in range(nprof):
        step()
    torch.cuda.synchronize(dev)
    t_plain_ms = (time.perf_counter() - tp) / nprof * 1e3
    overhead_ms = max((sum(stage_ms.values()) - t_plain_ms) / max(len(stage_ms), 1), 0.0)
    stage_ms = {k: max(v - overhead_ms, 0.0) for k, v in stage_ms.items()}
    dims = mdl.dims
    # N > 1: every rank's own stage times (a curve can only be read if one sees whose share is the long one) and the
    # collective alone: the packed buffer all-reduced back to back, hip events on the step's stream
    per_rank, allreduce_us = None, None
    if world > 1:
        if native:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            scratch = packed.clone()
            nar = 50
            for _ in range(5):
                _lib.check(lib.sgpr_comm_allreduce(h, scratch.data_ptr(), scratch.numel(), 0, sp))
            e0.record(stream)
            for _ in range(nar):
                _lib.check(lib.sgpr_comm_allreduce(h, scratch.data_ptr(), scratch.numel(), 0, sp))
            e1.record(stream)
            torch.cuda.synchronize(dev)
            allreduce_us = e0.elapsed_time(e1) / nar * 1e3
        mine = {"rank": rank, "local_atoms": (N - rank + world - 1) // world, "step_us": round(t_plain_ms * 1e3, 2),
                "allreduce_us": None if allreduce_us is None else round(allreduce_us, 2),
                **{k + "_us": round(v * 1e3, 2) for k, v in stage_ms.items()}}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    # PCIe-inclusive rate of the host-array entry point (never `value`): numpy in, numpy out
    host_rate = None
    calc_wall = None
    calc_wall_big = None
    if world == 1:
        mdl.predict(numbers, pos, cell, pbc, beta=True)
        th = time.perf_counter()
        nh = 20
        for _ in range(nh):
            mdl.predict(numbers, pos, cell, pbc, beta=True)
        host_rate = N * nh / (time.perf_counter() - th)
        # wall time of one calculate() (NL + descriptors + K_nm + E/F/stress + covloss) through
        # the drop-in surface, median of >= 50 calls after 5 warm-ups; the atoms move every call by the same
        # random-walk step as above, so nothing is served from ASE's result cache and the neighbour candidates
        # are rebuilt at the MD rate
        from autoforce_amd.ase_shim import Atoms
        from autoforce_amd.calculator import ActiveCalculator
        calc = ActiveCalculator(covariance=mdl, logfile=None)
        atoms = Atoms(numbers, pos.copy(), cell, pbc)
        atoms.calc = calc
        rng = np.random.default_rng(5)
        walls = []
        for it in range(55):
            atoms.positions = atoms.positions + (args.walk_sigma or 1e-3) * rng.normal(size=pos.shape)
            tc = time.perf_counter()
            atoms.get_forces()
            walls.append(time.perf_counter() - tc)
        w = float(np.median(walls[5:]))
        calc_wall = {"median_ms": w * 1e3, "atom_steps_per_s": N / w, "calls": len(walls) - 5,
                     "device_step_ms": t_plain_ms, "wall_over_device": w * 1e3 / t_plain_ms,
                     "what": "ActiveCalculator.calculate() wall, numpy in / numpy out, one synchronised call per step"}
        if not args.no_big_wall:
            calc_wall_big = calculate_wall_big(local_rank, args.walk_sigma or 1e-3)

    # ---- an MD LOOP: step k + 1 starts from the forces of step k.  Langevin NVT (600 K, 1 fs, friction as cl/md.py:31)
    # with positions and velocities in device memory (sgpr_md_*): the wall time of K dependent steps, noise upload
    # included.  (The bench model's weights are random: scaled by 0.01 for this loop so that the forces are of the size a
    # fitted model gives — the work per step does not depend on their values.)
    md_loop, md_head = None, None
    # (several ranks: the loop runs sharded over the library's own exchange — every rank integrates all atoms from the
    # summed forces —; with RCCL or the host-staged collective only the resident-frames figure exists)
    if args.md_steps > 0 and (world == 1 or backend == "ipc"):
        from autoforce_amd.ase_shim import kB
        from autoforce_amd.workloads import FS, MASS, fit_to_teacher
        snap = mdl.snapshot_weights()
        choli0 = mdl.choli
        fit_to_teacher(mdl, numbers, pos, cell, pbc, device=local_rank)
        mdl.set_weights(mdl.mu, choli=mdl.choli, vscale=mdl.make_vscale())
        mrng = np.random.default_rng(11)
        mass = np.array([MASS[int(z)] for z in numbers])
        v0 = mrng.normal(size=(N, 3)) * np.sqrt(kB * 600.0 / mass[:, None])
        K = args.md_steps
        mdl.md_begin(numbers, pos, cell, pbc, mass, v0, dt=1.0 * FS, friction=1e-3, kT=kB * 600.0, seed=11)
        # the synthetic workload first: 200 steps in which the start lattice relaxes under the fitted model and the
        # neighbour capacities take their size (part of building the state, like the model fit); then the contract's
        # W untimed warm-up steps, then exactly K timed ones
        MD_EQUILIBRATION = 200
        mdl.md_run(MD_EQUILIBRATION, None)
        if args.warmup > 0:
            mdl.md_run(args.warmup, None)

        def md_timed(noise, K=K):
            rows, resizes = [], 0
            tm = time.perf_counter()
            while sum(len(r) for r in rows) < K and resizes <= 8:
                d0 = sum(len(r) for r in rows)
                sc, code = mdl.md_run(K - d0, None if noise is None else noise[d0:])
                rows.append(sc)
                resizes += code == 2   # a neighbour capacity outgrown on the way: the next call re-sizes and goes on
            return time.perf_counter() - tm, np.concatenate(rows), resizes

        # THE HEADLINE: exactly --steps dependent MD steps between two fences, max over the ranks — one BATCH; batches are
        # repeated (each continues the run where the one before stopped) until >= MIN_TIMED_MS have been timed, and the
        # median batch is what `value` / `ms_per_step` quote
        def md_batch():
            fence()
            th = time.perf_counter()
            _, sc_b, rz_b = md_timed(None, args.steps)
            fence()
            return max_over_ranks(time.perf_counter() - th), len(sc_b) == args.steps and rz_b == 0

        md_dts = []
        d0, ok0 = md_batch()
        for _ in range(batches_of(d0) - 1 if ok0 else 0):
            md_dts.append(md_batch())
        md_dts = [d0] * bool(ok0) + [d for d, ok in md_dts if ok]   # (a batch that outgrew a capacity is not a sample)
        if md_dts:
            dt_md = float(np.median(md_dts))
            md_head = {"ms_per_step": dt_md / args.steps * 1e3, "value": N * args.steps / dt_md, "batches": len(md_dts),
                       "ms_per_step_min": min(md_dts) / args.steps * 1e3, "ms_per_step_max": max(md_dts) / args.steps * 1e3,
                       "ms_per_step_first_batch": d0 / args.steps * 1e3}
        rbm0 = mdl.list_rebuilds()
        tmd, sc, resizes = md_timed(None)                       # deviates drawn on the device
        rbm1 = mdl.list_rebuilds()
        if world == 1:
            t_rows, sc_rows, _ = md_timed(mrng.normal(size=(K, N, 3)))   # numpy's deviates, uploaded (the host loop's stream)
        else:
            t_rows, sc_rows = tmd, sc
        T_md = sc[:, 12] / (3 * N * kB)
        md_loop = {"steps": int(len(sc)), "capacity_resizes": int(resizes), "ms_per_step": tmd / max(len(sc), 1) * 1e3,
                   "atom_steps_per_s": N * len(sc) / tmd, "list_rebuilds": int(rbm1 - rbm0),
                   "ms_per_step_with_uploaded_deviates": t_rows / max(len(sc_rows), 1) * 1e3,
                   "temperature_K_mean": float(T_md.mean()) if len(sc) else None,
                   "energy_first_last": [float(sc[0, 0]), float(sc[-1, 0])] if len(sc) else None,
                   "largest_covloss_first_last": [float(sc[0, 11]), float(sc[-1, 11])] if len(sc) else None,
                   "what": "sgpr_md_run: BAOAB Langevin 600 K, 1 fs, friction 1e-3 (cl/md.py:31), state resident in HBM, "
                           "integrator + binning of the next step inside the step's last kernel (5 launches per step), the "
                           "Langevin deviates drawn on the device (counter-based); wall time of K dependent steps, the host reads "
                           "16 scalars per step; weights fitted to workloads.PairTeacher on the start frame (a model whose "
                           "forces hold the lattice together; the work per step does not depend on the weights)"}
        mdl.set_weights(snap["mu"], choli=choli0, vscale=snap["vscale"] or None)

    result = None
    if rank == 0:
        cnt = (N - rank + world - 1) // world
        ptr, _, _ = mdl.neighbors(N)
        nn_mean = float(ptr[-1]) / max(cnt, 1)
        Dc = dims["Dc"]
        # algorithmic bytes of THIS formulation: packed rows (Dc) and only this rank's atoms
        cs = dims["S"] * 64
        ab = algorithmic_bytes(cnt, nn_mean, Dc, m, cs)
        rev_key = "descriptor_rev_covloss" if "descriptor_rev_covloss" in stage_ms else "descriptor_rev"
        w_key = "gemm_w" if "gemm_w" in stage_ms else "gemm_w_covloss"
        dom = max((k for k in stage_ms if k in ab), key=lambda k: stage_ms[k])
        dom_s = stage_ms[dom] * 1e-3
        # HBM traffic of the dominant kernel: PMC counters cannot be read inside this run (rocprofv3 --pmc
        # is a separate pass); the committed summary of the builder's own pass over this command is quoted
        # with its source, or null
        traffic, traffic_source = None, None
        sha = csrc_sha()
        tpath = latest_profile(f"r*_pmc_traffic_lips{N}_m{m}.json")
        pmc = None
        if world == 1 and tpath:
            tj = json.load(open(tpath))
            if tj.get("csrc_sha") == sha:
                pmc = tj
                traffic = tj["kernels"].get(dom, {}).get("fetch_x2_plus_write")
                traffic_source = f"profiles/{os.path.basename(tpath)} ({tj.get('source', 'builder run')}; same kernel sources: csrc sha {sha})"
            else:
                traffic_source = (f"null: profiles/{os.path.basename(tpath)} was collected on other kernel sources "
                                  f"(csrc sha {tj.get('csrc_sha')} then, {sha} now)")
        af = algorithmic_flops(numbers, [x.number for x in mdl.X], dims["Dpad"], world)
        if dom in af:  # a GEMM leads: price it against the dense fp64 MFMA peak
            Dd = dims["D"] * dims["S"] ** 2
            dense = {"gemm_knm": 2.0 * cnt * m * Dd, "gemm_w": 2.0 * cnt * m * Dd, "gemm_w_covloss": 2.0 * cnt * m * Dd + float(m) * m * cnt}
            head = {"bound": "mfma", "achieved": af[dom] / dom_s / 1e12, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": af[dom] / dom_s / 1e12 / FP64_MFMA_PEAK_TF, "algorithmic_flops": af[dom],
                    "dense_equiv_flops": dense.get(dom),
                    "flops_note": "algorithmic_flops counts what the block-diagonal formulation needs: K_nm, W and L^-1 are zero "
                                  "outside a species block (atoms and inducing points are species-sorted) and rows are packed "
                                  f"{Dd} -> {dims['Dpad']} doubles (p[u,v,l] is symmetric in u,v); dense_equiv_flops is SURVEY 8(d)'s "
                                  "formula for the same product on dense unpacked operands (2NmD + m^2 N): pricing the launch with "
                                  "it would exceed the fp64 MFMA peak, so frac uses the block count"}
        else:
            head = {"bound": "hbm", "achieved": ab[dom] / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ab[dom] / dom_s / 1e9 / HBM_PEAK_GBS}
        roof = {
            "kernel": dom,
            **head,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "algorithmic_bytes": ab[dom],
            "avg_launch_us": stage_ms[dom] * 1e3,
            "timing": f"hip events on the launch stream, {nprof} eager steps after the timed region, minus the "
                      f"per-stage marker overhead ({overhead_ms * 1e3:.2f} us = (sum of intervals - marker-free step time) / stages)",
            # scalar keys (a parser that drops nested objects keeps these): per-kernel time, the descriptor kernels
            # against the HBM roofline with their algorithmic bytes, the GEMMs against the fp64 MFMA peak
            **{k + "_us": round(v * 1e3, 2) for k, v in stage_ms.items()},
            "desc_fwd_GBs": round(ab["list_forward"] / (stage_ms["list_forward"] * 1e-3) / 1e9, 1) if stage_ms.get("list_forward") else None,
            "desc_rev_GBs": round(ab[rev_key] / (stage_ms[rev_key] * 1e-3) / 1e9, 1) if stage_ms.get(rev_key) else None,
            "desc_hbm_frac": round((ab["list_forward"] + ab[rev_key]) /
                                   ((stage_ms.get("list_forward", 0) + stage_ms.get(rev_key, 0)) * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
            if stage_ms.get("list_forward") and stage_ms.get(rev_key) else None,
            "desc_valu_frac": (pmc or {}).get("desc_valu_frac"),
            "desc_valu_note": (pmc or {}).get("desc_valu_note"),
            "knm_TFs": round(af["gemm_knm"] / (stage_ms["gemm_knm"] * 1e-3) / 1e12, 2) if stage_ms.get("gemm_knm") else None,
            "wcov_TFs": round(af[w_key] / (stage_ms[w_key] * 1e-3) / 1e12, 2) if stage_ms.get(w_key) else None,
            "gemm_phase_us": round(sum(stage_ms.get(k, 0.0) for k in ("gemm_knm", "gemm_w", "gemm_w_covloss")) * 1e3, 2),
            "knm_frac": round(af["gemm_knm"] / (stage_ms["gemm_knm"] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF, 4) if stage_ms.get("gemm_knm") else None,
            "wcov_frac": round(af[w_key] / (stage_ms[w_key] * 1e-3) / 1e12 / FP64_MFMA_PEAK_TF, 4) if stage_ms.get(w_key) else None,
            "stage_us": {k: round(v * 1e3, 2) for k, v in stage_ms.items()},
            "gemm_TFLOPs": {k: round(af[k] / (stage_ms[k] * 1e-3) / 1e12, 2) for k in af if stage_ms.get(k)},
            "hbm_GBs": {k: round(ab[k] / (stage_ms[k] * 1e-3) / 1e9, 1) for k in ab if stage_ms.get(k)},
            "step_bytes_this_formulation": sum(v for k, v in ab.items() if k in stage_ms),
            "step_bytes_survey_formula": survey_step_bytes(cnt, nn_mean, dims["D"] * dims["S"] ** 2, m),
            "step_GBs_survey_formula": survey_step_bytes(cnt, nn_mean, dims["D"] * dims["S"] ** 2, m) / (ms_per_step * 1e-3) / 1e9,
            "pass_GBs_packed": sum(v for k, v in ab.items() if k in stage_ms) / (sum(stage_ms[k] for k in ab if k in stage_ms) * 1e-3) / 1e9,
        }
        result = {
            "metric": "MD-step atoms*steps/sec (SGPR predict: NL + descriptors + K_nm + E/F/stress + covloss); `value`: a Langevin "
                      "MD loop on the device, every step from the forces of the one before (sgpr_md_run; --steps dependent steps "
                      "between two fences); `value_resident_frames`: steps over independent frames resident in HBM, enqueued "
                      "back to back; `md_loop`: the same MD loop over --md-steps steps; `value_calculate_wall`: SURVEY 8(d)'s "
                      "t_step = wall of one synchronised ActiveCalculator.calculate()"
                      + ("" if md_head else "  [no device MD loop in this configuration: `value` is the resident-frames figure]"),
            "value": md_head["value"] if md_head else value,
            "unit": "atom*steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": md_head["ms_per_step"] if md_head else ms_per_step,
            "value_is": "md_loop (dependent steps)" if md_head else "resident frames",
            "timed": {"batch_steps": args.steps, "min_timed_ms": MIN_TIMED_MS,
                      "what": "a batch = exactly --steps steps between two fences (barrier + synchronize), max over ranks; repeated "
                              "until min_timed_ms have been timed; value / ms_per_step = the MEDIAN batch",
                      "md_loop": md_head, "resident_frames": {"batches": len(res_dts), "ms_per_step_min": min(res_dts) / args.steps * 1e3,
                                                              "ms_per_step_max": max(res_dts) / args.steps * 1e3,
                                                              "ms_per_step_first_batch": res_dts[0] / args.steps * 1e3}},
            "value_resident_frames": value,
            "ms_per_step_resident_frames": ms_per_step,
            "collective": backend if world > 1 else None,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"LiPS {N} atoms (3 species), {m} inducing points, lmax=nmax=3, eta=4, rc=6.0",
                "atoms": N, "inducing": m, "mean_neighbors": round(nn_mean, 2), "max_neighbors": dims["nn_max"],
                "packed_row": Dc, "graph": bool(args.graph), "launches_per_step": (5 if args.fuse_next else 6) if world == 1 else len(stage_ms),
                "input": (f"closed Gaussian random walk, sigma {args.walk_sigma} A per component per step, {nframes} frames "
                          f"resident in HBM" if nframes > 1 else "static frame"),
                "neighbor_skin_A": args.skin,
                "md_state": ("the timed MD steps continue a run of 200 equilibration steps (the start lattice relaxing under the "
                             "fitted model: part of building the state) + the W warm-up steps") if md_head else None,
                "list_rebuilds_in_timed_steps": int(rb1.value - rb0.value),
                "host_array_path_atom_steps_per_s": host_rate,
                "parallelism": f"atoms sharded x{world}, " + (
                    "single rank: no collective" if world == 1 else
                    f"the library's own exchange per step: all-gather of {3 * N + 4 * ((N + world - 1) // world) + 11} words of partial sums into hipIpc-mapped "
                    "peer buffers + local sum in rank order" if backend == "ipc" else
                    f"one RCCL all-reduce of {len(out_host)} doubles per step, issued by libsgpr_hip on the step stream"
                    if backend == "rccl" else f"torch.distributed all-reduce of {len(out_host)} doubles, host staged"),
            },
            "roofline": roof,
            "per_rank": per_rank,
            "allreduce_us": allreduce_us,
            "value_calculate_wall": None if calc_wall is None else calc_wall["atom_steps_per_s"],
            "value_md_loop": None if md_loop is None else md_loop["atom_steps_per_s"],
            "md_loop": md_loop,
            "warmup_batches_ms": [round(1e3 * b, 4) for b in warm_batches],
            "calculate_wall": calc_wall,
            "calculate_wall_ms": None if calc_wall is None else calc_wall["median_ms"],
            "roofline_16384": None if calc_wall_big is None else calc_wall_big.pop("roofline", None),
            "calculate_wall_16384": calc_wall_big,
            "energy": float(out_host[4 * N]),
            "max_force": float(np.abs(out_host[:3 * N]).max()),
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or N
            v, cdt, reps, cores = cpu_baseline(numbers, pos, cell, pbc, mdl, mdl.mu, sample)
            result["cpu_baseline"] = {
                "value": v, "unit": "atom*steps/s", "cores": cores, "kind": "port",
                "sample": f"{reps} passes over {sample} of {N} atoms of the same frame (linked-cell neighbour list + "
                          f"descriptors + K_nm + reverse pass + covloss), {cdt:.1f} s, "
                          f"oracle/sgpr_oracle.c with OpenMP on {cores} threads",
            }
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    mdl.close()


if __name__ == "__main__":
    main()
