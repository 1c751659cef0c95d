#!/usr/bin/env python3
"""bench.py — MD-step throughput of the SGPR predict hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" = one pass of the hot path over one frame: device neighbour list -> SeSoap descriptors
-> K_nm -> energy, forces, virial -> covloss (calculator/active.py:425-502 of the reference),
with positions already resident in HBM.  Workload at N=1: BASELINE configs[2] — 4096-atom
"LiPS" (3 species), 512 inducing points, lmax=nmax=3, eta=4, rc=6 A, fp64.
N > 1: atoms are dealt to ranks (per-species round robin, the reference's Distributer); every
rank evaluates its share and ONE RCCL all-reduce of the packed [F | beta | E | virial] buffer
combines them (the reference's four MPI all-reduces, calculator/active.py:562,601,602,777).
The frame is the same for every N, so scaling is "strong".

torch is used for device memory, streams and torch.distributed only; the numerics are
libsgpr_hip.so (hand-written HIP) called through its C ABI.
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
FP64_MFMA_PEAK_TF = 78.6  # BASELINE.md §4 (MI355X FP64 matrix peak)


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


def algorithmic_bytes(N, nn, D, m):
    """Per-kernel algorithmic HBM bytes of one step: the SURVEY.md §8d step formula
    N nn 44 x2 + N nn 24 + 4 (8 N D) + 8 m D + 2 (8 N m) + 24 N, split by the kernel that moves each term
    (+ the list the build kernel writes and the triangular factor the covloss product reads)."""
    return {
        "neighbor_bin": N * (24 + 24 + 16),
        "neighbor_build": N * nn * 8 + N * 24,
        "descriptor_fwd": N * nn * 44 + 8 * N * D,
        "gemm_knm": 8 * N * D + 8 * m * D + 8 * N * m,
        "gemm_w_covloss": 8 * N * m + 8 * m * D + 8 * N * D + 8 * N * m + 8 * m * m,
        "descriptor_dc": 8 * N * D,
        "descriptor_pair": N * nn * 44 + N * nn * 24 + 24 * N,
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
    return {"gemm_knm": knm, "gemm_w_covloss": w + cov}


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
    ptr, j, off = mdl.neighbors(len(numbers))  # device neighbour list of the benchmark frame
    # restrict to the first `sample_atoms` atoms' environments
    ptr_s = ptr.copy()
    ptr_s[sample_atoms + 1:] = ptr_s[sample_atoms]
    reps, dt = 0, 0.0
    t0 = time.perf_counter()
    while dt < min_seconds:
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
    ap.add_argument("--overlap", type=int, default=0, help="1: covloss GEMM on a side stream next to the reverse pass")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo for single-GPU dry runs)")
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
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    numbers, pos, cell, pbc = lips(args.atoms_side, seed=0)
    N, m = len(numbers), args.inducing
    mdl = build_model(local_rank, numbers, pos, cell, pbc, m)
    lib = _lib.load()
    h = mdl.handle
    _lib.check(lib.sgpr_set_option(h, b"graph", 1 if args.graph else 0))
    _lib.check(lib.sgpr_set_option(h, b"overlap", args.overlap))

    dev = torch.device("cuda", local_rank)
    pos_d = torch.from_numpy(pos).to(dev)
    cell_d = torch.from_numpy(cell).to(dev)
    packed = torch.zeros(int(lib.sgpr_packed_len(N)), dtype=torch.float64, device=dev)
    _lib.check(lib.sgpr_bind_system(h, N, _lib.ptr(_lib.i32(numbers)), _lib.ptr(_lib.i32(pbc.astype(np.int32))),
                                    rank, world))
    stream = torch.cuda.current_stream(dev)
    sp = C.c_void_p(stream.cuda_stream)

    def step():
        _lib.check(lib.sgpr_step_dev(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), sp))
        if world > 1:
            dist.all_reduce(packed)

    for _ in range(max(args.warmup, 2)):
        step()
    _lib.check(lib.sgpr_sync_check(h, sp))

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    _lib.check(lib.sgpr_sync_check(h, sp))
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    ms_per_step = dt / args.steps * 1e3
    value = N * args.steps / dt
    out_host = packed.cpu().numpy()  # the reduced result of the last timed step

    # ---- per-kernel durations: HIP events on the launch stream, same steps run eagerly
    mdl.profile(True)
    acc = {}
    nprof = min(args.steps, 50)
    for _ in range(nprof):
        step()
        torch.cuda.synchronize(dev)
        for k, v in mdl.stage_times().items():
            acc[k] = acc.get(k, 0.0) + v
    mdl.profile(False)
    stage_ms = {k: v / nprof for k, v in acc.items()}
    # An event-to-event interval holds one launch's marker/dispatch overhead besides the kernel.
    # Calibration: the same steps without the markers (and without the collective) take t_plain;
    # the markers therefore cost (sum of intervals - t_plain) / n_stages per stage.
    torch.cuda.synchronize(dev)
    tp = time.perf_counter()
    for _ in range(nprof):
        _lib.check(lib.sgpr_step_dev(h, pos_d.data_ptr(), cell_d.data_ptr(), packed.data_ptr(), sp))
    torch.cuda.synchronize(dev)
    t_plain_ms = (time.perf_counter() - tp) / nprof * 1e3
    overhead_ms = max((sum(stage_ms.values()) - t_plain_ms) / max(len(stage_ms), 1), 0.0)
    stage_ms = {k: max(v - overhead_ms, 0.0) for k, v in stage_ms.items()}
    dims = mdl.dims
    # PCIe-inclusive rate of the host-array entry point (never `value`): numpy in, numpy out
    host_rate = None
    if world == 1:
        mdl.predict(numbers, pos, cell, pbc, beta=True)
        th = time.perf_counter()
        nh = 20
        for _ in range(nh):
            mdl.predict(numbers, pos, cell, pbc, beta=True)
        host_rate = N * nh / (time.perf_counter() - th)

    result = None
    if rank == 0:
        cnt = (N - rank + world - 1) // world
        ptr, _, _ = mdl.neighbors(N)
        nn_mean = float(ptr[-1]) / max(cnt, 1)
        Dc = dims["Dc"]
        # algorithmic bytes of THIS formulation: packed rows (Dc) and only this rank's atoms
        ab = algorithmic_bytes(cnt, nn_mean, Dc, m)
        ab_survey = algorithmic_bytes(cnt, nn_mean, dims["D"] * dims["S"] ** 2, m)
        dom = max((k for k in stage_ms if k in ab), key=lambda k: stage_ms[k])
        dom_s = stage_ms[dom] * 1e-3
        # dense flops actually required by the block-diagonal packed formulation
        traffic = None
        tpath = os.path.join(ROOT, "profiles", f"r01_pmc_traffic_lips{N}_m{m}.json")
        if world == 1 and os.path.exists(tpath):  # PMC counters come from separate rocprofv3 --pmc passes
            traffic = json.load(open(tpath))["kernels"].get(dom, {}).get("fetch_x2_plus_write")
        af = algorithmic_flops(numbers, [x.number for x in mdl.X], dims["Dpad"], world)
        if dom in af:  # a GEMM leads: price it against the dense fp64 MFMA peak
            head = {"bound": "mfma", "achieved": af[dom] / dom_s / 1e12, "peak": FP64_MFMA_PEAK_TF, "unit": "TFLOP/s",
                    "frac": af[dom] / dom_s / 1e12 / FP64_MFMA_PEAK_TF, "algorithmic_flops": af[dom]}
        else:
            head = {"bound": "hbm", "achieved": ab[dom] / dom_s / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": ab[dom] / dom_s / 1e9 / HBM_PEAK_GBS}
        roof = {
            "kernel": dom,
            **head,
            "traffic": traffic,
            "algorithmic_bytes": ab[dom],
            "avg_launch_us": stage_ms[dom] * 1e3,
            "timing": f"hip events on the launch stream, {nprof} eager steps after the timed region, minus the "
                      f"per-stage marker overhead ({overhead_ms * 1e3:.2f} us = (sum of intervals - marker-free step time) / stages)",
            "stage_us": {k: round(v * 1e3, 2) for k, v in stage_ms.items()},
            "gemm_TFLOPs": {k: round(af[k] / (stage_ms[k] * 1e-3) / 1e12, 2) for k in af if stage_ms.get(k)},
            "hbm_GBs": {k: round(ab[k] / (stage_ms[k] * 1e-3) / 1e9, 1) for k in ab if stage_ms.get(k)},
            "step_bytes_packed_layout": sum(ab.values()),
            "step_bytes_survey_formula": sum(ab_survey.values()),
            "pass_GBs_packed": sum(ab.values()) / (sum(stage_ms[k] for k in ab if k in stage_ms) * 1e-3) / 1e9,
        }
        result = {
            "metric": "MD-step atoms*steps/sec (SGPR predict: NL + descriptors + K_nm + E/F/stress + covloss)",
            "value": value,
            "unit": "atom*steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"LiPS {N} atoms (3 species), {m} inducing points, lmax=nmax=3, eta=4, rc=6.0",
                "atoms": N, "inducing": m, "mean_neighbors": round(nn_mean, 2), "max_neighbors": dims["nn_max"],
                "packed_row": Dc, "graph": bool(args.graph),
                "host_array_path_atom_steps_per_s": host_rate,
                "parallelism": f"atoms sharded x{world}, one RCCL all-reduce of {len(out_host)} doubles per step",
            },
            "roofline": roof,
            "energy": float(out_host[4 * N]),
            "max_force": float(np.abs(out_host[:3 * N]).max()),
        }
        if world == 1 and not args.no_cpu_baseline:
            sample = args.cpu_sample or N
            v, cdt, reps, cores = cpu_baseline(numbers, pos, cell, pbc, mdl, mdl.mu, sample)
            result["cpu_baseline"] = {
                "value": v, "unit": "atom*steps/s", "cores": cores, "kind": "port",
                "sample": f"{reps} passes over {sample} of {N} atoms of the same frame (descriptors + K_nm + reverse "
                          f"pass + covloss; neighbour list taken from the device), {cdt:.1f} s, "
                          f"oracle/sgpr_oracle.c with OpenMP on {cores} threads",
            }
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    mdl.close()


if __name__ == "__main__":
    main()
