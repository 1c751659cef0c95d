#!/usr/bin/env python3
"""Randomised differential test of the incremental model edits at larger sizes than tests/test_hip_data.py: random walks over
append / pop / remove-at-index / argsort-ordered downsize by a large batch / random selections / frame push and pop / new
targets / force-only fits, the fit compared after every step with a model set up and factored from scratch.
usage: python3 tools/fuzz_edits.py [seed0=0] [seeds=4] [m0=220] [steps=40]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel, workloads  # noqa: E402

seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
nseeds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
m0 = int(sys.argv[3]) if len(sys.argv) > 3 else 220
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 40
shape = (6, 6, 6)
worst = 0.0
for seed in range(seed0, seed0 + nseeds):
    rng = np.random.default_rng(1000 + seed)
    numbers, pos, cell, pbc = workloads.oxide_ordered(shape, seed=seed)
    species = sorted(set(int(z) for z in numbers))
    mdl = SGPRModel(3, 3, 4, 6.0, species=species)
    pool = []
    k = 0
    while len(pool) < m0 + 200:
        n2, p2, c2, b2 = workloads.oxide_ordered(shape, seed=50 + 7 * seed + k, sigma=0.04 + 0.02 * (k % 4))
        pool += workloads.inducing_from_frame(mdl, n2, p2, c2, b2, 150, seed=90 + k, noise=0.0)
        k += 1
    rng.shuffle(pool)
    mdl.set_inducing(pool[:m0])
    nxt = [m0]
    frames, targets = [], []

    def push():
        n2, p2, c2, b2 = workloads.oxide_ordered(shape, seed=int(rng.integers(1 << 30)), sigma=0.07)
        nv = int(rng.choice([0, 6]))
        mdl.data_push(n2, p2, c2, b2, nv)
        frames.append(((n2, p2, c2, b2), nv))
        targets.append(rng.normal(size=1 + 3 * len(n2) + nv))

    for _ in range(3):
        push()

    def check(tag, with_energies=True):
        global worst
        Y = np.concatenate(targets)
        got = mdl.data_solve(Y, noise=0.02, with_energies=with_energies).copy()
        info = mdl.solve_info()
        ref = mdl.scratch()
        ref.set_inducing(mdl.X)
        for fr, nv in frames:
            ref.data_push(*fr, nv)
        want = ref.data_solve(Y, noise=0.02, with_energies=with_energies)
        assert np.array_equal(ref.M, mdl.M), tag
        assert ref.ridge == mdl.ridge, (tag, ref.ridge, mdl.ridge)
        p0, p1 = ref.data_matvec(want), mdl.data_matvec(got)
        err = float(np.abs(p0 - p1).max() / max(np.abs(p0).max(), 1e-300))
        cerr = float(np.abs(ref.choli - mdl.choli).max() / np.abs(ref.choli).max())
        worst = max(worst, err)
        ok = np.isfinite(got).all() and err < 1e-6 and cerr < 1e-5  # (K_mm of near-duplicate LCEs is ill-conditioned: choli to ~cond x eps)
        if not ok:
            print(f"seed {seed} {tag}: m={mdl.m} frames={len(frames)} fit_err={err:.2e} choli_err={cerr:.2e} {info}", flush=True)
            raise SystemExit(1)
        ref.close()

    check("start")
    for step in range(steps):
        op = rng.choice(["add", "add", "add8", "pop", "remove", "lii_big", "lii_small", "select", "push", "popdata", "popfirstdata", "targets", "force_only"])
        m = mdl.m
        if op == "add":
            mdl.add_inducing(pool[nxt[0] % len(pool)]); nxt[0] += 1
        elif op == "add8":
            for _ in range(8):
                mdl.add_inducing(pool[nxt[0] % len(pool)]); nxt[0] += 1
        elif op == "pop" and m > 20:
            mdl.remove_inducing(-1)
        elif op == "remove" and m > 20:
            mdl.remove_inducing(int(rng.integers(m)))
        elif op == "lii_big" and m > 60:
            keep = int(m * rng.uniform(0.55, 0.8))
            mdl.select_inducing(np.argsort(mdl.M.sum(axis=1), kind="stable")[:keep].tolist())
        elif op == "lii_small" and m > 20:
            mdl.select_inducing(np.argsort(mdl.M.sum(axis=1), kind="stable")[:m - int(rng.integers(1, 4))].tolist())
        elif op == "select" and m > 20:
            mdl.select_inducing(rng.permutation(m)[:m - int(rng.integers(0, 3))].tolist())
        elif op == "push" and len(frames) < 5:
            push()
        elif op == "popdata" and len(frames) > 1:
            mdl.data_pop(-1); frames.pop(); targets.pop()
        elif op == "popfirstdata" and len(frames) > 1:
            mdl.data_pop(0); frames.pop(0); targets.pop(0)
        elif op == "targets":
            j = int(rng.integers(len(targets)))
            targets[j] = targets[j] + 0.1 * rng.normal(size=len(targets[j]))
        elif op == "force_only":
            check(f"step {step} force_only", with_energies=False)
        check(f"step {step} {op}")
    print(f"seed {seed}: ok, final m = {mdl.m}, frames = {len(frames)}", flush=True)
    mdl.close()
print(f"all ok; worst fit difference {worst:.2e}")
