"""Synthetic workloads of BASELINE.json (recipes: SURVEY.md §8d).  Host-side numpy only.

The reference ships no structure files for these systems, so frames are generated from fixed
seeds: simple-cubic sites, shuffled species, Gaussian rattle with a minimum-distance rejection.
"""
import numpy as np

from .model import Local


def _rattled_lattice(shape, spacing, sigma, rng, dmin=1.6):
    g = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).reshape(-1, 3) * spacing
    pos = g.astype(float)
    cell = np.diag([n * spacing for n in shape]).astype(float)
    # rattle with rejection: a displacement is redrawn until no lattice neighbour (pre-rattle
    # distance = spacing) can come closer than dmin; cheap sufficient test per atom
    out = np.empty_like(pos)
    lim = (spacing - dmin) / 2.0
    for i in range(len(pos)):
        while True:
            d = sigma * rng.normal(size=3)
            if np.linalg.norm(d) <= lim:
                break
        out[i] = pos[i] + d
    return out, cell


def lips(n_side=16, seed=0, sigma=0.15):
    """C3/C4: "LiPS" 16^3 = 4096 simple-cubic sites, 2.72 A, 1536 Li / 512 P / 2048 S."""
    rng = np.random.default_rng(seed)
    dims = tuple(n_side) if np.ndim(n_side) else (n_side,) * 3  # (32, 32, 16): the 16384 atoms of config 5
    N = int(np.prod(dims))
    nLi, nP = 3 * N // 8, N // 8
    numbers = rng.permutation(np.array([3] * nLi + [15] * nP + [16] * (N - nLi - nP))).astype(np.int32)
    pos, cell = _rattled_lattice(dims, 2.72, sigma, rng)
    return numbers, pos, cell, np.array([True, True, True])


def si_diamond(reps=(2, 2, 1), seed=0, sigma=0.05):
    """C1: diamond Si a=5.431, 8-atom cubic x (2,2,1) = 32 atoms; L_z = 5.431 < rc: atoms meet their own images."""
    rng = np.random.default_rng(seed)
    a = 5.431
    base = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0], [.25, .25, .25], [.25, .75, .75],
                     [.75, .25, .75], [.75, .75, .25]]) * a
    cells = np.stack(np.meshgrid(*[np.arange(n) for n in reps], indexing="ij"), -1).reshape(-1, 3) * a
    pos = (cells[:, None, :] + base[None]).reshape(-1, 3) + sigma * rng.normal(size=(len(cells) * 8, 3))
    cell = np.diag([n * a for n in reps]).astype(float)
    return np.full(len(pos), 14, np.int32), pos, cell, np.array([True, True, True])


def li_bcc(reps=(8, 4, 4), seed=0, sigma=0.10):
    """C2: bcc Li a=3.49, 2-atom cubic x reps = 256 atoms."""
    rng = np.random.default_rng(seed)
    a = 3.49
    base = np.array([[0, 0, 0], [0.5, 0.5, 0.5]]) * a
    cells = np.stack(np.meshgrid(*[np.arange(n) for n in reps], indexing="ij"), -1).reshape(-1, 3) * a
    pos = (cells[:, None, :] + base[None]).reshape(-1, 3) + sigma * rng.normal(size=(len(cells) * 2, 3))
    cell = np.diag([n * a for n in reps]).astype(float)
    return np.full(len(pos), 3, np.int32), pos, cell, np.array([True, True, True])


def oxide(shape=(32, 32, 16), seed=0, sigma=0.15):
    """C5: 16384 sites, 4 species 2:1:1:4 (Li, Zr, La, O)."""
    rng = np.random.default_rng(seed)
    N = int(np.prod(shape))
    counts = [2 * N // 8, N // 8, N // 8]
    z = [3] * counts[0] + [40] * counts[1] + [57] * counts[2]
    z += [8] * (N - len(z))
    numbers = rng.permutation(np.array(z)).astype(np.int32)
    pos, cell = _rattled_lattice(shape, 2.72, sigma, rng)
    return numbers, pos, cell, np.array([True, True, True])


def inducing_from_frame(model, numbers, pos, cell, pbc, m, seed, noise=0.05):
    """m LCEs drawn species-proportionally from a frame (+ noise), using the MODEL's own device
    neighbour list (the product path; no oracle involved)."""
    rng = np.random.default_rng(seed)
    N = len(numbers)
    model.predict(numbers, pos, cell, pbc, beta=False)  # builds the neighbour list on the device
    ptr, j, off = model.neighbors(N)
    numbers = np.asarray(numbers)
    picks = []
    zs, cnt = np.unique(numbers, return_counts=True)
    quota = np.floor(cnt / N * m).astype(int)
    quota[np.argmax(cnt)] += m - quota.sum()
    for z, q in zip(zs, quota):
        picks.extend(rng.choice(np.nonzero(numbers == z)[0], size=q, replace=False).tolist())
    X = []
    rc = model.cutoff
    for a in picks:
        s = slice(int(ptr[a]), int(ptr[a + 1]))
        r = pos[j[s]] - pos[a] + off[s].astype(float) @ cell
        r = r + noise * rng.normal(size=r.shape)
        keep = np.linalg.norm(r, axis=1) < rc - 1e-3
        X.append(Local(int(numbers[a]), numbers[j[s]][keep], r[keep]))
    return X
