#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (read-only)
from /root/reference in the build container.  Nothing from the reference is copied: the
fixtures are inputs + the reference's outputs (numpy arrays).

Run (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/gen/make_golden.py

The reference needs `ase` and `mpi4py`, both absent here; tests/golden/gen/stubs/ holds
empty stand-ins (our own code) for the handful of names touched at import time.  The
periodic neighbour list (ASE's job in the reference, descriptor/atoms.py:348-363) is
replaced by the brute-force builder below; it defines the pair rule |r| < rc, bothways,
no self-interaction but self-images kept (SURVEY.md §8c "parity unpinned at this boundary").

Reference call sites exercised (file:line under /root/reference/theforce):
  descriptor/ylm.py:113-225            Ylm.forward (incl. the near-z shear, :10-23)
  descriptor/sesoap.py:161-260         SeSoap.forward (value + analytic grad)
  descriptor/soap.py:488-525           the repo's only numeric KAT (AbsSeriesSoap)
  descriptor/atoms.py:365-382          Local construction, r = xyz[n]-xyz[a]+off@cell
  similarity/universal.py:109-122      get_func, similarity/similarity.py:94-103 lone_atoms
  regression/gppotential.py:63-84      EnergyForceKernel.energy_energy
  calculator/active.py:548-611         E, autograd forces, cell gradient, stress
  calculator/active.py:781-804         covloss; gppotential.py:644-649 vscale
  regression/algebra.py:29-47          jitcholesky
  regression/gppotential.py:1204-1339  _regression(optimize=False)
  descriptor/atoms.py:228-246          Distributer
"""
import os
import sys

sys.dont_write_bytecode = True  # never write __pycache__ into the read-only reference tree
from types import SimpleNamespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.dirname(HERE)
sys.path[:0] = [os.path.join(HERE, "stubs"), "/root/reference"]

import torch  # noqa: E402
import theforce  # noqa: E402,F401  (sets fp64 default)
from theforce.descriptor.atoms import Distributer, Local  # noqa: E402
from theforce.descriptor.cutoff import PolyCut  # noqa: E402
from theforce.descriptor.sesoap import DefaultRadii, SeSoap  # noqa: E402
from theforce.descriptor.soap import AbsSeriesSoap  # noqa: E402
from theforce.descriptor.ylm import Ylm  # noqa: E402
from theforce.regression.algebra import jitcholesky  # noqa: E402
from theforce.regression.gppotential import (  # noqa: E402
    AutoMean,
    EnergyForceKernel,
    _regression,
)
from theforce.regression.kernel import White  # noqa: E402
from theforce.similarity.sesoap import SeSoapKernel  # noqa: E402

torch.set_num_threads(8)


# ----------------------------------------------------------------------------- helpers
def brute_force_nl(pos, cell, pbc, rc):
    """All (i, j, off) with |x_j - x_i + off@cell| < rc, (j,off) != (i,0). Sorted by (i, j, off)."""
    n = len(pos)
    cell = np.asarray(cell, float)
    if np.abs(np.linalg.det(cell)) > 1e-12:
        inv = np.linalg.inv(cell)
        heights = 1.0 / np.linalg.norm(inv, axis=0)  # V / |a_j x a_k|
    else:
        heights = np.full(3, np.inf)
    frac_span = np.zeros(3)
    if np.all(np.isfinite(heights)):
        f = pos @ np.linalg.inv(cell)
        frac_span = f.max(0) - f.min(0)
    nmax = [
        int(np.ceil(rc / heights[k] + frac_span[k])) if pbc[k] else 0 for k in range(3)
    ]
    shifts = [
        (a, b, c)
        for a in range(-nmax[0], nmax[0] + 1)
        for b in range(-nmax[1], nmax[1] + 1)
        for c in range(-nmax[2], nmax[2] + 1)
    ]
    I, J, O = [], [], []
    for s in shifts:
        t = np.array(s, float) @ cell
        d = pos[None, :, :] + t - pos[:, None, :]
        r = np.sqrt((d**2).sum(-1))
        m = r < rc
        if s == (0, 0, 0):
            m &= ~np.eye(n, dtype=bool)
        i, j = np.nonzero(m)
        I.append(i)
        J.append(j)
        O.append(np.tile(np.array(s, np.int32), (len(i), 1)))
    I = np.concatenate(I)
    J = np.concatenate(J)
    O = np.concatenate(O)
    order = np.lexsort((O[:, 2], O[:, 1], O[:, 0], J, I))
    I, J, O = I[order], J[order], O[order]
    ptr = np.zeros(n + 1, np.int64)
    np.add.at(ptr, I + 1, 1)
    ptr = np.cumsum(ptr)
    return ptr, J.astype(np.int32), O.astype(np.int32)


def make_kernel(lmax=3, nmax=3, eta=4, rc=6.0):
    kern = SeSoapKernel(lmax, nmax, eta, rc, radii=DefaultRadii())
    return kern, EnergyForceKernel([kern])


def dense_p(kern, loc, species):
    """p-hat of a Local as dense [S,S,D] over the model species table (zeros for absent).
    The COO block with indices (ab[0], ab[1]) = (beta, alpha) holds sum_m c[alpha,n]c*[beta,n']
    flattened [n,n',l] (sesoap.py:165-171,195-203); we store out[ab[0]-slot, ab[1]-slot]."""
    v = kern.saved(loc, "value")
    S = len(species)
    D = kern.descriptor.dim
    out = np.zeros((S, S, D))
    if v is None:
        return out
    v = v.coalesce()
    idx = v.indices().numpy()
    val = v.values().detach().numpy()
    zs = list(species)
    for k in range(idx.shape[1]):
        out[zs.index(idx[0, k]), zs.index(idx[1, k])] = val[k]
    return out


def frame_outputs(name, numbers, pos, cell, pbc, ind, mu, lmax=3, nmax=3, eta=4, rc=6.0):
    """Push one frame through the reference path; `ind` = list of (Zc, Znbr[], r[][3])."""
    kern, efk = make_kernel(lmax, nmax, eta, rc)
    ptr, J, O = brute_force_nl(pos, cell, pbc, rc)
    xyz = torch.tensor(pos, requires_grad=True)
    lll = torch.tensor(cell, requires_grad=True)
    locs = []
    N = len(numbers)
    for a in range(N):
        n = J[ptr[a] : ptr[a + 1]].astype(np.int64)
        off = O[ptr[a] : ptr[a + 1]]
        cells = (torch.from_numpy(off[..., None].astype(float)) * lll).sum(dim=1)
        r = xyz[n] - xyz[a] + cells  # descriptor/atoms.py:367-368
        loc = Local(a, n, numbers[a], numbers[n], r, off, efk.kernels, dont_save_grads=True)
        loc.natoms = N
        locs.append(loc)
    X = []
    for zc, zn, rr in ind:
        k = len(zn)
        loc = Local(
            0,
            np.arange(1, k + 1),
            int(zc),
            np.asarray(zn, dtype=np.int64),
            torch.tensor(np.asarray(rr, float).reshape(k, 3)),
            None,
            efk.kernels,
            dont_save_grads=True,
        )
        X.append(loc)
    cov = efk(locs, X)
    M = efk(X, X).detach()
    mu_t = torch.tensor(mu)
    E = (cov @ mu_t).sum()
    F = -torch.autograd.grad(E, xyz, retain_graph=True, allow_unused=True)[0]
    (dcell,) = torch.autograd.grad(E, lll, allow_unused=True)
    if dcell is None:
        dcell = torch.zeros_like(lll)
    # active.py:604-610
    stress1 = -(F[:, None] * xyz[..., None]).sum(dim=0)
    stress2 = (dcell[:, None] * lll[..., None]).sum(dim=0)
    # ase Atoms.get_volume raises ValueError only for a rank-deficient cell -> volume = -2 (active.py:606-609)
    vol = abs(np.linalg.det(cell)) if abs(np.linalg.det(cell)) > 0 else -2.0
    stress = ((stress1 + stress2).detach().numpy() / vol).flat[[0, 4, 8, 5, 2, 1]]
    L, ridge = jitcholesky(M)
    choli = L.inverse().contiguous()
    # active.py:781-804 (normalized kernel) + gppotential.py:644-649
    b = choli @ cov.detach().t()
    c = (b * b).sum(dim=0)
    beta = (1 - c).clamp(min=0.0).sqrt()
    indz = np.array([z for z, _, _ in ind])
    mm = mu_t * (M @ mu_t)
    species = sorted(set(int(z) for z in numbers) | set(int(z) for z in indz)
                     | set(int(z) for _, zn, _ in ind for z in zn))
    vscale = {int(z): float(mm[torch.from_numpy(indz == z)].sum() / (indz == z).sum())
              for z in set(indz.tolist())}
    vs = np.array([vscale.get(int(z), np.inf) for z in numbers])
    with np.errstate(invalid="ignore"):
        covloss = beta.numpy() * np.sqrt(vs)
    P = np.stack([dense_p(kern, l, species) for l in locs])
    Pm = np.stack([dense_p(kern, l, species) for l in X])
    ind_ptr = np.cumsum([0] + [len(zn) for _, zn, _ in ind])
    out = dict(
        lmax=lmax, nmax=nmax, eta=eta, rc=rc,
        numbers=np.asarray(numbers, np.int32), positions=pos, cell=cell,
        pbc=np.asarray(pbc, bool), species=np.asarray(species, np.int32),
        nl_ptr=ptr, nl_j=J, nl_off=O,
        ind_z=indz.astype(np.int32), ind_ptr=ind_ptr.astype(np.int64),
        ind_nbr_z=np.concatenate([np.asarray(zn, np.int32) for _, zn, _ in ind] + [np.zeros(0, np.int32)]),
        ind_nbr_r=np.concatenate([np.asarray(rr, float).reshape(-1, 3) for _, _, rr in ind] + [np.zeros((0, 3))]),
        mu=mu, p=P, p_ind=Pm, cov=cov.detach().numpy(), M=M.numpy(),
        energy=float(E), forces=F.numpy(), dcell=dcell.numpy(), stress=stress,
        L=L.numpy(), ridge=float(ridge), choli=choli.numpy(), beta=beta.numpy(),
        vscale_z=np.array(sorted(vscale), np.int32),
        vscale=np.array([vscale[z] for z in sorted(vscale)]), covloss=covloss,
    )
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **out)
    print(f"{name}: N={N} m={len(ind)} <nn>={len(J)/N:.1f} E={float(E):.6f} |F|max={abs(F).max():.4f} "
          f"ridge={float(ridge):.2e} sumF={abs(F.sum(0)).max():.1e}")
    return out


def env_of(pos, cell, pbc, rc, a, numbers):
    ptr, J, O = brute_force_nl(pos, cell, pbc, rc)
    n = J[ptr[a] : ptr[a + 1]]
    r = pos[n] - pos[a] + O[ptr[a] : ptr[a + 1]].astype(float) @ cell
    return int(numbers[a]), np.asarray(numbers)[n], r


def inducing_from(rng, numbers, pos, cell, pbc, rc, m, noise=0.05):
    """m LCEs drawn (species-proportionally, without replacement) from a frame, + noise."""
    idx = rng.choice(len(numbers), size=m, replace=False)
    ind = []
    for a in idx:
        zc, zn, r = env_of(pos, cell, pbc, rc, a, numbers)
        r = r + noise * rng.normal(size=r.shape)
        keep = np.linalg.norm(r, axis=1) < rc - 1e-3  # keep the env strictly inside rc
        ind.append((zc, zn[keep], r[keep]))
    return ind


# ----------------------------------------------------------------------------- G1 Ylm
def g1_ylm():
    rng = np.random.default_rng(101)
    out = {}
    for lmax in (2, 3, 4):
        a = rng.normal(size=(40, 3)) * np.array([1.0, 1.0, 1.0])
        b = a.copy()
        b[:6, 0] = 1e-3 * rng.normal(size=6) * np.abs(b[:6, 2])  # within the 0.01 cone
        b[:6, 1] = 1e-3 * rng.normal(size=6) * np.abs(b[:6, 2])
        b[5] = [0.0, 0.0, -1.3]  # exactly on -z
        for tag, v in (("plain", a), ("nearz", b)):
            xyz = torch.tensor(v)
            Y, dY = Ylm(lmax)(xyz, grad=True)
            out[f"l{lmax}_{tag}_xyz"] = v
            out[f"l{lmax}_{tag}_Y"] = Y.numpy()
            out[f"l{lmax}_{tag}_dY"] = dY.numpy()
    np.savez_compressed(os.path.join(OUT, "g1_ylm.npz"), **out)
    print("g1_ylm done")


# ----------------------------------------------------------------------------- G2 SeSoap
def g2_sesoap():
    rng = np.random.default_rng(202)
    out = {}
    cases = []

    def env(nn, zs, scale=2.2):
        r = rng.normal(size=(nn, 3)) * scale
        d = np.linalg.norm(r, axis=1)
        r[d > 5.9] *= (5.5 / d[d > 5.9])[:, None]
        r[d < 0.8] *= (1.0 / d[d < 0.8])[:, None]
        z = rng.choice(zs, size=nn)
        return r, z.astype(np.int64)

    cases.append(("s1", 3, 3) + env(40, [3]))
    cases.append(("s2h", 3, 3) + env(30, [1, 8]))
    cases.append(("s3", 3, 3) + env(45, [3, 15, 16]))
    r, z = env(20, [3, 15, 16])
    r[3] = [1e-4, -2e-4, 2.5]
    cases.append(("s3_nearz", 3, 3, r, z))
    r, z = env(1, [14])
    cases.append(("single", 3, 3, r, z))
    r, z = env(12, [14, 8])
    r[0] = r[0] / np.linalg.norm(r[0]) * 6.5  # beyond the cutoff -> zero weight
    cases.append(("beyond", 3, 3, r, z))
    cases.append(("l2n2", 2, 2) + env(25, [3, 16]))
    cases.append(("l4n2", 4, 2) + env(25, [3, 16]))
    cases.append(("l3n4", 3, 4) + env(25, [29]))
    names = []
    for name, lmax, nmax, r, z in cases:
        s = SeSoap(lmax, nmax, PolyCut(6.0), radii=DefaultRadii())
        ab, p, _, dp, _ = s(torch.tensor(r), torch.tensor(z), grad=True, sparse_tensor=False)
        names.append(name)
        out[name + "_lmax"] = lmax
        out[name + "_nmax"] = nmax
        out[name + "_r"] = r
        out[name + "_z"] = z.astype(np.int32)
        out[name + "_ab"] = ab.numpy().astype(np.int32)  # (2, S^2): rows (ab[0], ab[1])
        out[name + "_p"] = p.numpy()  # [S^2, D]
        out[name + "_dp"] = dp.numpy()  # [S^2, D, nn, 3]
        # un-normalised as well (pins nnl separately from the norm)
        ab2, p2, _ = s(torch.tensor(r), torch.tensor(z), grad=False, normalize=False, sparse_tensor=False)
        out[name + "_p_raw"] = p2.numpy()
        # vector-Jacobian product by autograd (the force path's derivative, active.py:587-599)
        G = rng.normal(size=tuple(p.shape))
        x = torch.tensor(r, requires_grad=True)
        _, pa, _ = s(x, torch.tensor(z), grad=False, sparse_tensor=False)
        (pa * torch.tensor(G)).sum().backward()
        out[name + "_G"] = G
        out[name + "_vjp"] = x.grad.numpy()
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g2_sesoap.npz"), **out)
    print("g2_sesoap done:", names)


# ----------------------------------------------------------------------------- G3 SubSeSoap
def g3_subsesoap():
    """The fixed-species descriptor of the `species=[...]` kernels (descriptor/sesoap.py:263-391;
    calculator/active.py:31-38) on the g2 environments: dense p[S,S,n,n',l] over the sorted
    species list of each case, plus one case whose table holds a species absent from the
    environment and one where a neighbour species is outside the table (it is ignored)."""
    from theforce.descriptor.sesoap import SubSeSoap
    g2 = np.load(os.path.join(OUT, "g2_sesoap.npz"))
    out, names = {}, []
    for name in g2["names"]:
        lmax, nmax = int(g2[name + "_lmax"]), int(g2[name + "_nmax"])
        r, z = g2[name + "_r"], g2[name + "_z"].astype(np.int64)
        tables = {"": sorted(set(z.tolist()))}
        if name == "s3":
            tables["_extra"] = [3, 8, 15, 16]       # 8 never occurs
            tables["_drop"] = [3, 16]               # P neighbours are outside the table
        for tag, table in tables.items():
            s = SubSeSoap(lmax, nmax, PolyCut(6.0), table, radii=DefaultRadii())
            p = s(torch.tensor(r), torch.tensor(z), grad=False)
            key = name + tag
            names.append(key)
            out[key + "_case"] = str(name)
            out[key + "_table"] = np.array(table, np.int32)
            out[key + "_p"] = p.numpy().reshape(len(table), len(table), nmax + 1, nmax + 1, lmax + 1)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(OUT, "g3_subsesoap.npz"), **out)
    print("g3_subsesoap done:", names)



# ----------------------------------------------------------------------------- G13 fixed-species KERNEL values
def g13_subsesoap_kernel():
    """K-level fixture of the `species=[...]` kernels: calculator/active.py:28-38 builds one SubSeSoapKernel per
    central species (similarity/sesoap.py:27-43 = HeterogeneousSoapKernel with DotProd()**exponent on the SubSeSoap
    descriptor, similarity/heterosoap.py:37-71) and regression/gppotential.py:63-84 sums them.  Environments: the g2
    cases as LCEs with several central species, one whose neighbours include a species outside the table (dropped
    silently, descriptor/sesoap.py:343-346), and two lone atoms (similarity/similarity.py:94-103 adds its term once
    PER KERNEL of the list).  Output: K[i][j] = sum over the kernels of kern(loc_i, loc_j)."""
    from theforce.similarity.sesoap import SubSeSoapKernel
    g2 = np.load(os.path.join(OUT, "g2_sesoap.npz"))
    table = [3, 15, 16]
    kerns = [SubSeSoapKernel(3, 3, 4, 6.0, z, table, radii=DefaultRadii()) for z in table]
    efk = EnergyForceKernel(kerns)
    envs = []
    for case, zc in (("s3", 3), ("s3", 15), ("s3", 16), ("s3_nearz", 16), ("s1", 3), ("s3_nearz", 3)):
        r, z = g2[case + "_r"], g2[case + "_z"].astype(np.int64)
        envs.append((zc, z, r))
    rng = np.random.default_rng(1313)
    r, z = g2["s3_r"] + 0.05 * rng.normal(size=g2["s3_r"].shape), g2["s3_z"].astype(np.int64).copy()
    z[::7] = 8                                     # oxygen neighbours: not in the table
    envs.append((15, z, r))
    envs.append((3, np.zeros(0, np.int64), np.zeros((0, 3))))   # lone atoms
    envs.append((3, np.zeros(0, np.int64), np.zeros((0, 3))))
    envs.append((16, np.zeros(0, np.int64), np.zeros((0, 3))))
    locs = []
    for k, (zc, z, r) in enumerate(envs):
        n = len(z)
        loc = Local(0, np.arange(1, n + 1), zc, z, torch.tensor(r).reshape(n, 3), None, efk.kernels, dont_save_grads=True)
        loc.natoms = n + 1
        locs.append(loc)
    K = efk(locs, locs).detach().numpy()
    out = dict(table=np.array(table, np.int32), lmax=3, nmax=3, eta=4.0, rc=6.0, K=K, n_env=len(envs),
               zc=np.array([e[0] for e in envs], np.int32),
               ptr=np.concatenate([[0], np.cumsum([len(e[1]) for e in envs])]).astype(np.int64),
               nbr_z=np.concatenate([e[1] for e in envs]).astype(np.int32),
               nbr_r=np.concatenate([e[2].reshape(-1, 3) for e in envs]))
    np.savez_compressed(os.path.join(OUT, "g13_subsesoap_kernel.npz"), **out)
    print("g13_subsesoap_kernel done:", K.shape, "lone-atom block:\n", K[-3:, -3:])

# ----------------------------------------------------------------------------- KAT
def kat_absseries():
    """descriptor/soap.py:488-525: inputs and the target tensor, plus what the reference
    computes for them here (full precision)."""
    xyz = np.array(
        [
            [0.175, 0.884, -0.87, 0.354, -0.082, 3.1],
            [-0.791, 0.116, 0.19, -0.832, 0.184, 0.0],
            [0.387, 0.761, 0.655, -0.528, 0.973, 0.0],
        ]
    ).T.copy()
    target = np.array(
        [
            [[0.36174603, 0.39013356, 0.43448023], [0.39013356, 0.42074877, 0.46857549], [0.43448023, 0.46857549, 0.5218387]],
            [[0.2906253, 0.30558356, 0.33600938], [0.30558356, 0.3246583, 0.36077952], [0.33600938, 0.36077952, 0.40524778]],
            [[0.16241845, 0.18307552, 0.20443194], [0.18307552, 0.22340802, 0.26811937], [0.20443194, 0.26811937, 0.34109511]],
        ]
    )
    s = AbsSeriesSoap(2, 2, PolyCut(3.0))
    p, dp = s(torch.tensor(xyz))
    p = p.permute(2, 0, 1).numpy()  # [l, n, n']
    assert np.allclose(p, target, rtol=1e-5, atol=1e-8)
    np.savez_compressed(os.path.join(OUT, "kat_absseries.npz"), xyz=xyz, target_lnn=target, p_lnn=p,
                        dp=dp.numpy(), lmax=2, nmax=2, rc=3.0, unit=1.0)
    print("kat_absseries: reference reproduces its own target")


# ----------------------------------------------------------------------------- frames
def rattle(rng, pos, sigma):
    return pos + sigma * rng.normal(size=pos.shape)


def frames():
    # C1: 32-atom diamond Si, 8-atom cubic x (2,2,1): L_z = 5.431 < rc -> self images
    rng = np.random.default_rng(0)
    a = 5.431
    basis = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0],
                      [.25, .25, .25], [.25, .75, .75], [.75, .25, .75], [.75, .75, .25]]) * a
    pos = np.concatenate([basis + np.array([i, j, 0]) * a for i in range(2) for j in range(2)])
    cell = np.diag([2 * a, 2 * a, a])
    pos = rattle(rng, pos, 0.05)
    numbers = np.full(32, 14)
    rng1 = np.random.default_rng(1)
    pos2 = rattle(rng1, pos, 0.10)
    ind = inducing_from(rng1, numbers, pos2, cell, [True] * 3, 6.0, 16)
    mu = np.random.default_rng(2).normal(size=16)
    frame_outputs("g5_si32", numbers, pos, cell, [True] * 3, ind, mu)

    # 3 species incl. H (radius 0.5), cubic 4^3 sites at 2.6 A
    rng = np.random.default_rng(10)
    g = np.stack(np.meshgrid(*[np.arange(4)] * 3, indexing="ij"), -1).reshape(-1, 3) * 2.6
    pos = rattle(rng, g.astype(float), 0.15)
    numbers = rng.permutation(np.array([1] * 16 + [8] * 24 + [40] * 24))
    cell = np.eye(3) * 4 * 2.6
    rng1 = np.random.default_rng(11)
    pos2 = rattle(rng1, g.astype(float), 0.2)
    ind = inducing_from(rng1, numbers, pos2, cell, [True] * 3, 6.0, 24)
    mu = np.random.default_rng(12).normal(size=24)
    frame_outputs("g5_mixed64", numbers, pos, cell, [True] * 3, ind, mu)

    # triclinic, 2 species, positions deliberately outside the cell as well
    rng = np.random.default_rng(20)
    cell = np.array([[7.1, 0.0, 0.0], [2.3, 6.4, 0.0], [-1.1, 1.9, 8.2]])
    f = rng.random((24, 3)) * 1.4 - 0.2
    pos = f @ cell
    # push apart atoms that are too close
    for _ in range(200):
        ptr, J, O = brute_force_nl(pos, cell, [True] * 3, 1.7)
        if len(J) == 0:
            break
        i = np.repeat(np.arange(24), np.diff(ptr))
        d = pos[J] - pos[i] + O.astype(float) @ cell
        np.add.at(pos, i, -0.15 * d / np.linalg.norm(d, axis=1, keepdims=True))
    numbers = rng.choice([3, 16], size=24)
    rng1 = np.random.default_rng(21)
    ind = inducing_from(rng1, numbers, rattle(rng1, pos, 0.1), cell, [True] * 3, 6.0, 12)
    mu = np.random.default_rng(22).normal(size=12)
    frame_outputs("g5_tric24", numbers, pos, cell, [True] * 3, ind, mu)

    # non-periodic cluster with a lone atom; inducing set has a lone LCE of the same species
    rng = np.random.default_rng(30)
    pos = rng.normal(size=(14, 3)) * 2.0
    for _ in range(200):
        ptr, J, O = brute_force_nl(pos, np.zeros((3, 3)), [False] * 3, 1.5)
        if len(J) == 0:
            break
        i = np.repeat(np.arange(14), np.diff(ptr))
        d = pos[J] - pos[i]
        np.add.at(pos, i, -0.15 * d / np.linalg.norm(d, axis=1, keepdims=True))
    pos = np.concatenate([pos, [[30.0, 0.0, 0.0], [0.0, -40.0, 0.0]]])  # two lone atoms
    numbers = np.array([10] * 7 + [18] * 7 + [10, 18])
    cell = np.zeros((3, 3))
    rng1 = np.random.default_rng(31)
    ind = inducing_from(rng1, numbers, rattle(rng1, pos, 0.1), cell, [False] * 3, 6.0, 8)
    ind.append((10, np.zeros(0, np.int64), np.zeros((0, 3))))  # lone Ne LCE
    mu = np.random.default_rng(32).normal(size=9)
    frame_outputs("g5_cluster16", numbers, pos, cell, [False] * 3, ind, mu)

    # mixed pbc slab (pbc in x,y only), 2 species, with a near-z neighbour pair (shear quirk)
    rng = np.random.default_rng(40)
    g = np.stack(np.meshgrid(np.arange(3), np.arange(3), np.arange(2), indexing="ij"), -1).reshape(-1, 3) * 2.9
    pos = rattle(rng, g.astype(float), 0.1)
    pos[1] = pos[0] + np.array([1e-3, -2e-3, 2.9])  # atom 1 almost exactly above atom 0
    numbers = rng.choice([29, 47], size=18)
    cell = np.diag([8.7, 8.7, 12.0])
    rng1 = np.random.default_rng(41)
    ind = inducing_from(rng1, numbers, rattle(rng1, pos, 0.1), cell, [True, True, False], 6.0, 10)
    mu = np.random.default_rng(42).normal(size=10)
    frame_outputs("g5_slab18_nearz", numbers, pos, cell, [True, True, False], ind, mu)

    # cells at least 2*rc wide in every periodic direction: no atom appears twice in a neighbour list.
    # Needed for the training-row fixtures: the reference's get_leftgrad scatters with `g[j] += f`
    # (similarity/universal.py:148), which drops contributions when j repeats (periodic self images).
    rng = np.random.default_rng(60)
    cell = np.eye(3) * 12.6
    pos = rng.random((40, 3)) * 12.6
    for _ in range(300):
        ptr, J, O = brute_force_nl(pos, cell, [True] * 3, 1.9)
        if len(J) == 0:
            break
        i = np.repeat(np.arange(40), np.diff(ptr))
        d = pos[J] - pos[i] + O.astype(float) @ cell
        np.add.at(pos, i, -0.15 * d / np.linalg.norm(d, axis=1, keepdims=True))
    numbers = rng.choice([3, 16], size=40)
    rng1 = np.random.default_rng(61)
    ind = inducing_from(rng1, numbers, rattle(rng1, pos, 0.1), cell, [True] * 3, 6.0, 12)
    mu = np.random.default_rng(62).normal(size=12)
    frame_outputs("g5_big40", numbers, pos, cell, [True] * 3, ind, mu)

    rng = np.random.default_rng(70)
    cell = np.array([[13.0, 0.0, 0.0], [3.0, 13.2, 0.0], [-2.0, 2.5, 13.5]])
    pos = rng.random((36, 3)) @ cell
    for _ in range(300):
        ptr, J, O = brute_force_nl(pos, cell, [True] * 3, 1.9)
        if len(J) == 0:
            break
        i = np.repeat(np.arange(36), np.diff(ptr))
        d = pos[J] - pos[i] + O.astype(float) @ cell
        np.add.at(pos, i, -0.15 * d / np.linalg.norm(d, axis=1, keepdims=True))
    numbers = rng.choice([8, 40, 1], size=36)
    rng1 = np.random.default_rng(71)
    ind = inducing_from(rng1, numbers, rattle(rng1, pos, 0.1), cell, [True] * 3, 6.0, 10)
    mu = np.random.default_rng(72).normal(size=10)
    frame_outputs("g5_bigtric36", numbers, pos, cell, [True] * 3, ind, mu)

    # small-basis variant (lmax=2, nmax=2, eta=2, rc=4.5) on the Si frame
    rng = np.random.default_rng(0)
    pos = np.concatenate([basis + np.array([i, j, 0]) * a for i in range(2) for j in range(2)])
    pos = rattle(rng, pos, 0.05)
    cell = np.diag([2 * a, 2 * a, a])
    numbers = np.full(32, 14)
    rng1 = np.random.default_rng(51)
    ind = inducing_from(rng1, numbers, rattle(rng1, pos, 0.1), cell, [True] * 3, 4.5, 8)
    mu = np.random.default_rng(52).normal(size=8)
    frame_outputs("g5_si32_l2n2", numbers, pos, cell, [True] * 3, ind, mu, lmax=2, nmax=2, eta=2, rc=4.5)


# ----------------------------------------------------------------------------- G7 regression
def g7_regression():
    out = {}
    # jitcholesky: PD, and rank-deficient (ladder)
    rng = np.random.default_rng(70)
    A = rng.normal(size=(12, 12))
    Mpd = A @ A.T + 0.5 * np.eye(12)
    L, ridge = jitcholesky(torch.tensor(Mpd))
    out["chol_pd_M"], out["chol_pd_L"], out["chol_pd_ridge"] = Mpd, L.numpy(), float(ridge)
    B = rng.normal(size=(12, 4))
    Msd = B @ B.T  # rank 4
    Msd[5] = Msd[3]
    Msd[:, 5] = Msd[:, 3]  # exact duplicate row/col (duplicate inducing point)
    L, ridge = jitcholesky(torch.tensor(Msd))
    out["chol_sd_M"], out["chol_sd_L"], out["chol_sd_ridge"] = Msd, L.numpy(), float(ridge)
    ones = np.ones((30, 30))  # algebra.py:218-224 uses the all-ones matrix
    L, ridge = jitcholesky(torch.tensor(ones))
    out["chol_ones_L"], out["chol_ones_ridge"] = L.numpy(), float(ridge)

    # _regression on a duck-typed model
    rng = np.random.default_rng(71)
    m, nd = 10, 3
    natoms = [5, 7, 4]
    Z = [rng.choice([3, 16], size=n) for n in natoms]
    C = rng.normal(size=(m, 6))
    M = C @ C.T / 6 + 0.3 * np.eye(m)
    Ke = rng.normal(size=(nd, m))
    Kf = rng.normal(size=(3 * sum(natoms), m))
    Kv = rng.normal(size=(6 * nd, m))
    en = rng.normal(size=nd) * 3
    fr = [rng.normal(size=(n, 3)) for n in natoms]
    st = [rng.normal(size=6) * 0.01 for _ in natoms]
    vol = [100.0 + 10 * i for i in range(nd)]
    w = {3: 0.3, 16: -0.2}

    class _A:
        def __init__(self, i):
            self.target_forces = torch.tensor(fr[i])
            self.target_stress = torch.tensor(st[i])
            self._v = vol[i]
            self._z = Z[i]

        def get_volume(self):
            return self._v

        def counts(self):
            u, c = np.unique(self._z, return_counts=True)
            return {int(a): int(b) for a, b in zip(u, c)}

    class _D(list):
        target_energy = torch.tensor(en)

        @property
        def natoms(self):
            return natoms

        def counts(self):
            tot = {}
            for a in self:
                for z, c in a.counts().items():
                    tot[z] = tot.get(z, 0) + c
            return tot

    data = _D([_A(i) for i in range(nd)])
    mean = AutoMean()
    mean.set_data(data)
    for z in w:
        mean.weights[z] = torch.tensor(w[z])

    def gp_mean(dat, forces=False):
        return torch.stack([mean(a) for a in dat])

    ns = SimpleNamespace(
        ignore_forces=False, M=torch.tensor(M), Ke=torch.tensor(Ke), Kf=torch.tensor(Kf), Kv=torch.tensor(Kv),
        X=[SimpleNamespace(number=3)] * m, data=data,
        gp=SimpleNamespace(noise=White(signal=0.01, requires_grad=False), mean=gp_mean), mean=mean,
    )
    ns.K = torch.cat([ns.Ke, ns.Kf, ns.Kv])
    _regression(ns, optimize=False)
    out.update(reg_M=M, reg_Ke=Ke, reg_Kf=Kf, reg_Kv=Kv, reg_energies=en,
               reg_forces=np.concatenate([f.reshape(-1) for f in fr]),
               reg_virial=np.concatenate([s * v for s, v in zip(st, vol)]),
               reg_mean=np.array([float(mean(a)) for a in data]),
               reg_noise0=0.01, reg_mu=ns.mu.numpy(), reg_choli=ns.choli.numpy(),
               reg_ridge=float(ns.ridge), reg_sigma=float(ns.scaled_noise["all"]))
    np.savez_compressed(os.path.join(OUT, "g7_regression.npz"), **out)
    print("g7_regression done; sigma =", ns.scaled_noise["all"], "ridge(sd) =", out["chol_sd_ridge"],
          "ridge(ones) =", out["chol_ones_ridge"])


# ----------------------------------------------------------------------------- G9 Distributer
def g9_distributer():
    rng = np.random.default_rng(90)
    numbers = rng.choice([3, 15, 16], size=50, p=[0.375, 0.125, 0.5])
    out = {"numbers": numbers.astype(np.int32)}
    for ws in (1, 2, 4, 8):
        d = Distributer(ws)
        a = SimpleNamespace(numbers=numbers, ranks=None)
        d(a)
        out[f"ranks_{ws}"] = np.array(a.ranks, np.int32)
        # second frame on the same (loaded) distributer, as in consecutive MD steps w/o unload
        b = SimpleNamespace(numbers=numbers[::-1], ranks=None)
        d(b)
        out[f"ranks2_{ws}"] = np.array(b.ranks, np.int32)
    np.savez_compressed(os.path.join(OUT, "g9_distributer.npz"), **out)
    print("g9_distributer done")


# ----------------------------------------------------------------------------- training rows
def kernel_rows(name):
    """K_e, K_f, K_v of one data frame against the inducing set, by the reference's ANALYTIC
    gradient path (similarity/universal.py:109-183 get_func / get_leftgrad / get_virial through
    regression/gppotential.py:63-84): Ke = sum_i k(i,q); Kf = -d(sum_i k(i,q))/dx; Kv = sum r (x) dk/dr."""
    g = np.load(os.path.join(OUT, name + ".npz"))
    kern, efk = make_kernel(int(g["lmax"]), int(g["nmax"]), int(g["eta"]), float(g["rc"]))
    numbers, pos, cell = g["numbers"], g["positions"], g["cell"]
    ptr, J, O = g["nl_ptr"], g["nl_j"], g["nl_off"]
    N = len(numbers)
    xyz = torch.tensor(pos)
    lll = torch.tensor(cell)
    locs = []
    for a in range(N):
        n = J[ptr[a]:ptr[a + 1]].astype(np.int64)
        assert len(set(n.tolist())) == len(n), "repeated neighbour: the reference's g[j] += f would drop terms"
        off = O[ptr[a]:ptr[a + 1]]
        r = xyz[n] - xyz[a] + (torch.from_numpy(off[..., None].astype(float)) * lll).sum(dim=1)
        loc = Local(a, n, numbers[a], numbers[n], r, off, efk.kernels, dont_save_grads=False)
        loc.natoms = N
        locs.append(loc)
    X = []
    ip = g["ind_ptr"]
    for q, zc in enumerate(g["ind_z"]):
        k = int(ip[q + 1] - ip[q])
        X.append(Local(0, np.arange(1, k + 1), int(zc), g["ind_nbr_z"][ip[q]:ip[q + 1]].astype(np.int64),
                       torch.tensor(g["ind_nbr_r"][ip[q]:ip[q + 1]].reshape(k, 3)), None, efk.kernels, True))
    m = len(X)
    Ke = efk.base_kerns(locs, X, "func").detach().numpy().sum(0)                       # [m]
    lg = efk.base_kerns(locs, X, "leftgrad").detach().numpy().reshape(N, 3 * N, m).sum(0)
    Kf = -lg                                                                            # [3N, m]
    Kv = efk.base_kerns(locs, X, "virial").detach().numpy().reshape(N, 6, m).sum(0)     # [6, m]
    np.savez_compressed(os.path.join(OUT, name.replace("g5_", "g6_rows_") + ".npz"), Ke=Ke, Kf=Kf, Kv=Kv)
    print(f"{name}: rows done; |Kf|max={abs(Kf).max():.3f}")


# ----------------------------------------------------------------------------- G8 inducing-set edits
def _ref_locals(g, efk):
    numbers, pos, cell = g["numbers"], g["positions"], g["cell"]
    ptr, J, O = g["nl_ptr"], g["nl_j"], g["nl_off"]
    N = len(numbers)
    xyz, lll = torch.tensor(pos), torch.tensor(cell)
    locs = []
    for a in range(N):
        n = J[ptr[a]:ptr[a + 1]].astype(np.int64)
        off = O[ptr[a]:ptr[a + 1]]
        r = xyz[n] - xyz[a] + (torch.from_numpy(off[..., None].astype(float)) * lll).sum(dim=1)
        loc = Local(a, n, numbers[a], numbers[n], r, off, efk.kernels, dont_save_grads=False)
        loc.natoms = N
        locs.append(loc)
    X = []
    ip = g["ind_ptr"]
    for q, zc in enumerate(g["ind_z"]):
        k = int(ip[q + 1] - ip[q])
        X.append(Local(0, np.arange(1, k + 1), int(zc), g["ind_nbr_z"][ip[q]:ip[q + 1]].astype(np.int64),
                       torch.tensor(g["ind_nbr_r"][ip[q]:ip[q + 1]].reshape(k, 3)), None, efk.kernels, True))
    return locs, X


def g8_edits(name="g5_big40"):
    """The states a model passes through under add_inducing / pop_1inducing / popfirst_1inducing /
    select_inducing (gppotential.py:745-813, :1037-1046), each one refitted by the reference's
    _regression on the reference's own K_e/K_f/K_v and K_mm of that inducing subset.  Labels are a
    fixed synthetic target so that every state fits the same data."""
    g = np.load(os.path.join(OUT, name + ".npz"))
    kern, efk = make_kernel(int(g["lmax"]), int(g["nmax"]), int(g["eta"]), float(g["rc"]))
    locs, X = _ref_locals(g, efk)
    N, numbers = len(g["numbers"]), g["numbers"]
    rng = np.random.default_rng(80)
    energy = float(rng.normal()) * 2.0
    forces = rng.normal(size=(N, 3)) * 0.3
    stress = rng.normal(size=6) * 0.01
    vol = abs(np.linalg.det(g["cell"]))
    states = [list(range(8)), list(range(9)), list(range(10)), list(range(9)), list(range(1, 9)), [5, 1, 8, 3]]
    out = dict(frame=name, energy=energy, forces=forces, stress=stress, n_states=len(states))

    class _A:
        target_forces = torch.tensor(forces)
        target_stress = torch.tensor(stress)

        def get_volume(self):
            return vol

        def counts(self):
            u, c = np.unique(numbers, return_counts=True)
            return {int(a): int(b) for a, b in zip(u, c)}

    class _D(list):
        target_energy = torch.tensor([energy])
        natoms = [N]

        def counts(self):
            return self[0].counts()

    data = _D([_A()])
    for k, idx in enumerate(states):
        Xs = [X[i] for i in idx]
        m = len(Xs)
        M = efk(Xs, Xs).detach()
        Ke = efk.base_kerns(locs, Xs, "func").detach().sum(0).view(1, m)
        Kf = -efk.base_kerns(locs, Xs, "leftgrad").detach().view(N, 3 * N, m).sum(0)
        Kv = efk.base_kerns(locs, Xs, "virial").detach().view(N, 6, m).sum(0)
        mean = AutoMean()
        mean.set_data(data)
        ns = SimpleNamespace(
            ignore_forces=False, M=M, Ke=Ke, Kf=Kf, Kv=Kv, X=[SimpleNamespace(number=int(x.number)) for x in Xs],
            data=data, gp=SimpleNamespace(noise=White(signal=0.01, requires_grad=False),
                                          mean=lambda dat, forces=False: torch.stack([mean(a) for a in dat])),
            mean=mean)
        ns.K = torch.cat([Ke, Kf, Kv])
        _regression(ns, optimize=False)
        out[f"idx_{k}"] = np.array(idx, np.int32)
        out[f"M_{k}"] = M.numpy()
        out[f"mu_{k}"] = ns.mu.numpy()
        out[f"pred_{k}"] = (ns.K @ ns.mu).numpy()
        out[f"ridge_{k}"] = float(ns.ridge)
        out[f"sigma_{k}"] = float(ns.scaled_noise["all"])
    np.savez_compressed(os.path.join(OUT, "g8_edits.npz"), **out)
    print("g8_edits done:", [len(s) for s in states], "ridges", [out[f"ridge_{k}"] for k in range(len(states))])




# ----------------------------------------------------------------------------- G14 hyper-parameter search
def g14_hpo(name="g5_big40"):
    """_regression(optimize=True, noise_f=...) of the reference itself (gppotential.py:1265-1335: scipy BFGS through
    torch autograd on (MAE_f(noise) - noise_f)^2, then on the mean weights) on a case where the objective is NOT
    flat: forces that the model fits to a MAE between 0.044 (small noise) and 0.078 (noise -> 1), noise_f = 0.06 in
    between — the minimiser is the root MAE_f = noise_f, whatever the search that finds it."""
    g = np.load(os.path.join(OUT, name + ".npz"))
    kern, efk = make_kernel(int(g["lmax"]), int(g["nmax"]), int(g["eta"]), float(g["rc"]))
    locs, X = _ref_locals(g, efk)
    N, numbers = len(g["numbers"]), g["numbers"]
    idx = list(range(10))
    Xs = [X[i] for i in idx]
    m = len(Xs)
    M = efk(Xs, Xs).detach()
    Ke = efk.base_kerns(locs, Xs, "func").detach().sum(0).view(1, m)
    Kf = -efk.base_kerns(locs, Xs, "leftgrad").detach().view(N, 3 * N, m).sum(0)
    Kv = efk.base_kerns(locs, Xs, "virial").detach().view(N, 6, m).sum(0)
    rng = np.random.default_rng(140)
    mu_true = rng.normal(size=m)
    forces = (Kf.numpy() @ mu_true).reshape(N, 3) + 0.05 * rng.normal(size=(N, 3))
    vol = abs(np.linalg.det(g["cell"]))
    stress = (Kv.numpy() @ mu_true) / vol + 1e-4 * rng.normal(size=6)
    energy = float(Ke.numpy() @ mu_true) + 0.7 * N   # an offset for the mean to find
    noise_f = 0.06

    class _A:
        target_forces = torch.tensor(forces)
        target_stress = torch.tensor(stress)

        def get_volume(self):
            return vol

        def counts(self):
            u, c = np.unique(numbers, return_counts=True)
            return {int(a): int(b) for a, b in zip(u, c)}

    class _D(list):
        target_energy = torch.tensor([energy])
        natoms = [N]

        def counts(self):
            return self[0].counts()

    data = _D([_A()])
    mean = AutoMean()
    mean.set_data(data)
    ns = SimpleNamespace(
        ignore_forces=False, M=M, Ke=Ke, Kf=Kf, Kv=Kv, X=[SimpleNamespace(number=int(x.number)) for x in Xs],
        data=data, gp=SimpleNamespace(noise=White(signal=0.01, requires_grad=False),
                                      mean=lambda dat, forces=False: torch.stack([mean(a) for a in dat])),
        mean=mean)
    ns.K = torch.cat([Ke, Kf, Kv])
    # the shape of the objective, for the record: the force-only fit's MAE_f over a scan of the noise (numpy
    # restatement of make_mu(), gppotential.py:1245-1263, used for this diagnostic curve only)
    from theforce.regression.gppotential import to_0_1, to_inf_inf
    Lc = np.linalg.cholesky(M.numpy())
    Kfv = np.concatenate([Kf.numpy(), Kv.numpy()])
    Yfv = np.concatenate([forces.reshape(-1), stress * vol, np.zeros(m)])
    scale = float(np.diag(M.numpy()).mean() * 0.99)
    scan = []
    for x in np.linspace(-9.0, 6.0, 31):
        sig = 1.0 / (1.0 + np.exp(-x)) * scale
        mu_x = np.linalg.lstsq(np.concatenate([Kfv, sig * Lc.T]), Yfv, rcond=None)[0]
        scan.append((float(x), float(np.abs(Kf.numpy() @ mu_x - forces.reshape(-1)).mean())))
    x_start = 0.5   # a continuing run: the search starts from the noise the previous refit ended with
    ns._noise = {"all": torch.tensor(x_start)}
    _regression(ns, optimize=True, noise_f=noise_f)
    mae = float((ns.Kf @ ns.mu - torch.tensor(forces).view(-1)).abs().mean())
    out = dict(frame=name, idx=np.array(idx, np.int32), energy=energy, forces=forces, stress=stress, noise_f=noise_f,
               noise_logit_start=x_start,
               noise_logit=float(ns._noise["all"]), noise=float(to_0_1(ns._noise["all"])), sigma=float(ns.scaled_noise["all"]),
               mu=ns.mu.detach().numpy(), pred=(ns.K @ ns.mu).detach().numpy(), mae_f=mae, ridge=float(ns.ridge),
               mean_z=np.array(sorted(mean.weights), np.int32),
               mean_w=np.array([float(mean.weights[z]) for z in sorted(mean.weights)]), scan=np.array(scan))
    np.savez_compressed(os.path.join(OUT, "g14_hpo.npz"), **out)
    print("scan:", [(round(a, 1), round(b, 4)) for a, b in scan])
    print("g14_hpo done: noise", out["noise"], "sigma", out["sigma"], "MAE_f", mae, "(target", noise_f, ") mean", out["mean_w"],
          "scan MAE range", min(s_[1] for s_ in scan), max(s_[1] for s_ in scan))

# ----------------------------------------------------------------------------- G11 acceptance rules
def g11_acceptance(name="g5_big40"):
    """The decisions of the on-the-fly sampler, taken by the reference's OWN code
    (regression/gppotential.py:898-953 add_1atoms_fast, :955-982 add_1inducing, with its add_inducing /
    pop_1inducing / add_data / pop_1data :730-800 and _regression :1204-1339), on a duck-typed model:
    only the kernel plumbing (which Locals a frame consists of) is supplied here; every number and every
    accept/reject comes out of the reference functions.  Candidate LCEs carry `counts()` (one atom of
    their species): the reference evaluates `self.mean(loc)` inside `self(loc)` (:1121-1136)."""
    from theforce.descriptor.atoms import LocalsData
    from theforce.regression.gppotential import PosteriorPotential

    g = np.load(os.path.join(OUT, name + ".npz"))
    kern, efk = make_kernel(int(g["lmax"]), int(g["nmax"]), int(g["eta"]), float(g["rc"]))
    rc = float(g["rc"])
    numbers, cell, pbc = g["numbers"], g["cell"], g["pbc"]
    N = len(numbers)
    vol = abs(np.linalg.det(cell))
    rng = np.random.default_rng(110)

    def teacher(pos):
        """a smooth synthetic label: harmonic pull towards the golden frame + pairwise terms are not
        needed; what matters is that labels are a deterministic function of the positions."""
        d = pos - g["positions"]
        e = 0.5 * 0.8 * float((d * d).sum()) - 3.0
        return e, -0.8 * d + 0.05 * np.sin(3.0 * pos), 0.002 * np.array([1.0, -0.5, 0.25, 0.1, -0.2, 0.3]) * (1.0 + float(np.abs(d).sum()))

    # (LocalsData accepts exact `Local` instances only: the method is attached to the class, in this process)
    Local.counts = lambda self: {int(self.number): 1}
    _Loc = Local

    def locals_of(pos, grad=False):
        ptr, J, O = brute_force_nl(pos, cell, pbc, rc)
        xyz = torch.tensor(pos, requires_grad=grad)
        lll = torch.tensor(cell)
        locs = []
        for a in range(N):
            n = J[ptr[a]:ptr[a + 1]].astype(np.int64)
            off = O[ptr[a]:ptr[a + 1]]
            r = xyz[n] - xyz[a] + (torch.from_numpy(off[..., None].astype(float)) * lll).sum(dim=1)
            loc = _Loc(a, n, numbers[a], numbers[n], r, off, efk.kernels, dont_save_grads=grad)
            loc.natoms = N
            locs.append(loc)
        return locs, xyz

    class _Frame:
        is_distributed = False

        def __init__(self, pos):
            self.positions = pos
            self.locs, _ = locals_of(pos)
            e, f, s = teacher(pos)
            self.energy, self.forces, self.stress = e, f, s
            self.target_energy = torch.tensor([e])
            self.target_forces = torch.tensor(f)
            self.target_stress = torch.tensor(s)
            self.natoms = N

        def includes_species(self, species):
            return True

        def get_volume(self):
            return vol

        def counts(self):
            u, c = np.unique(numbers, return_counts=True)
            return {int(a): int(b) for a, b in zip(u, c)}

        def __len__(self):
            return N

    class _Data(list):
        is_distributed = False

        @property
        def X(self):
            return self

        @property
        def natoms(self):
            return [fr.natoms for fr in self]

        @property
        def target_energy(self):
            return torch.cat([fr.target_energy for fr in self])

        def counts(self):
            tot = {}
            for fr in self:
                for z, c in fr.counts().items():
                    tot[z] = tot.get(z, 0) + c
            return tot

    def as_locals(obj):
        """frames -> their Locals grouped per frame; Locals / LocalsData -> one group per Local"""
        if isinstance(obj, (_Frame,)):
            return [obj.locs]
        if isinstance(obj, Local):
            return [[obj]]
        return [grp for o in obj for grp in as_locals(o)]

    def kern_call(first, second, cov="energy_energy"):
        rows = as_locals(first)
        cols = [x for grp in as_locals(second) for x in grp]
        m = len(cols)
        out = []
        for grp in rows:
            n = len(grp)
            if cov == "energy_energy":
                out.append(efk.base_kerns(grp, cols, "func").sum(0).view(1, m))
            elif cov == "forces_energy":
                out.append(-efk.base_kerns(grp, cols, "leftgrad").view(n, -1, m).sum(0))
            elif cov == "virial_energy":
                out.append(efk.base_kerns(grp, cols, "virial").view(n, 6, m).sum(0))
            else:
                raise NotImplementedError(cov)
        return torch.cat(out) if out else torch.zeros(0, m)

    class _Model:
        ignore_forces = False
        has_target_forces = False
        is_distributed = False

        def __init__(self):
            self.mean_obj = AutoMean()
            self.gp = SimpleNamespace(species=sorted(set(int(z) for z in numbers)), kern=kern_call,
                                      noise=White(signal=0.01, requires_grad=False), method_caching=False,
                                      clear_cached=lambda *a, **k: None, parametric=self.mean_obj,
                                      mean=lambda dat, forces=False: torch.stack([self.mean_obj(a) for a in dat]))
            self.data = _Data()
            self.X = LocalsData([])
            self.M = torch.empty(0, 0)
            self.Ke, self.Kf, self.Kv = torch.empty(0, 0), torch.empty(0, 0), torch.empty(0, 0)
            self.ridge = 0.0

        mean = property(lambda self: self.mean_obj)
        K = property(lambda self: torch.cat([self.Ke, self.Kf, self.Kv], dim=0))

        def make_munu(self, *a, **k):
            if len(self.X) == 0 or len(self.data) == 0:
                return
            self.mean_obj.set_data(self.data)
            _regression(self, optimize=False)

        def __call__(self, *a, **k):
            return PosteriorPotential.forward(self, *a, **k)

        add_inducing = PosteriorPotential.add_inducing
        pop_1inducing = PosteriorPotential.pop_1inducing
        add_data = PosteriorPotential.add_data
        pop_1data = PosteriorPotential.pop_1data
        add_1inducing = PosteriorPotential.add_1inducing
        add_1atoms_fast = PosteriorPotential.add_1atoms_fast

    mdl = _Model()
    pos0 = g["positions"] + 0.03 * rng.normal(size=(N, 3))
    fr0 = _Frame(pos0)
    # the inducing candidates: environments of rattled copies of the frame
    cand_pos = g["positions"] + 0.12 * rng.normal(size=(N, 3))
    cand_locs, _ = locals_of(cand_pos)
    order = rng.permutation(N)[:14]
    out = dict(frame=name, n_cand=len(order), cand_pos=cand_pos, cand_atoms=order.astype(np.int32), pos0=pos0)
    # seed: first data frame (no inducing yet -> appended bare), first two candidates
    added, de, df = mdl.add_1atoms_fast(fr0, 0.05, 0.1, None, None, False)
    assert added == 1
    events = []  # kind, index, added, de, df, m, n_data, ridge, threshold 1, threshold 2
    for k, a in enumerate(order):
        loc = cand_locs[int(a)]
        ediff = (0.01, 0.3, 0.6)[k % 3]  # thresholds chosen so that both outcomes occur
        _ediff = ediff if len(mdl.X) > 1 else float(torch.finfo().eps)  # add_ninducing, :1004
        added, de = mdl.add_1inducing(loc, _ediff, detach=False)
        events.append((0, int(a), int(added), float(de), 0.0, len(mdl.X), len(mdl.data), float(mdl.ridge), _ediff, 0.0))
    # data candidates: further rattled frames through add_1atoms_fast with the calculator's cov
    frames_pos = [g["positions"] + s * rng.normal(size=(N, 3)) for s in (0.02, 0.15, 0.04, 0.3)]
    out["frames_pos"] = np.stack(frames_pos)
    for k, pos in enumerate(frames_pos):
        fr = _Frame(pos)
        locs_g, xyz = locals_of(pos, grad=True)
        cov = efk(locs_g, list(mdl.X))
        ediff_tot, fdiff = ((0.05, 0.1), (0.05, 3.0), (0.05, 0.1), (0.05, 6.0))[k]
        added, de, df = mdl.add_1atoms_fast(fr, ediff_tot, fdiff, xyz, cov, False)
        events.append((1, k, int(added), float(de), float(df), len(mdl.X), len(mdl.data), float(mdl.ridge), ediff_tot, fdiff))
        out[f"mu_after_frame_{k}"] = mdl.mu.detach().numpy().copy()
    out["events"] = np.array(events, float)
    out["teacher_e"] = np.array([teacher(p_)[0] for p_ in [pos0] + frames_pos])
    out["teacher_f"] = np.stack([teacher(p_)[1] for p_ in [pos0] + frames_pos])
    out["teacher_s"] = np.stack([teacher(p_)[2] for p_ in [pos0] + frames_pos])
    np.savez_compressed(os.path.join(OUT, "g11_acceptance.npz"), **out)
    print("g11_acceptance done:")
    for e in events:
        print("   ", "indu" if e[0] == 0 else "data", int(e[1]), "added" if e[2] else "rejected", f"de={e[3]:.3e} df={e[4]:.3e}",
              f"size=({int(e[6])},{int(e[5])}) ridge={e[7]:.1e}")


# ----------------------------------------------------------------------------- G12 committee (BCM)
def g12_bcm(name="g5_big40"):
    """The committee prediction of BCMActiveCalculator.update_results (calculator/active_bcm.py:589-633,
    with its get_covloss_with_model :842-858, get_covloss :860-885 and grads :640-663) run on a duck-typed
    calculator: two members = two inducing subsets of the golden frame's set with their own mu / choli /
    _vscale; energy = sum_k scale_k E_k / sum_k scale_k, scale_k = -ln(covmax_k) / covmax_k, and forces /
    stress by torch.autograd through that weighted energy."""
    import types
    from theforce.calculator.active_bcm import BCMActiveCalculator

    g = np.load(os.path.join(OUT, name + ".npz"))
    kern, efk = make_kernel(int(g["lmax"]), int(g["nmax"]), int(g["eta"]), float(g["rc"]))
    numbers, pos, cell, pbc = g["numbers"], g["positions"], g["cell"], g["pbc"]
    ptr, J, O = g["nl_ptr"], g["nl_j"], g["nl_off"]
    N = len(numbers)
    xyz = torch.tensor(pos, requires_grad=True)
    lll = torch.tensor(cell, requires_grad=True)
    locs = []
    for a in range(N):
        n = J[ptr[a]:ptr[a + 1]].astype(np.int64)
        off = O[ptr[a]:ptr[a + 1]]
        r = xyz[n] - xyz[a] + (torch.from_numpy(off[..., None].astype(float)) * lll).sum(dim=1)
        loc = Local(a, n, numbers[a], numbers[n], r, off, efk.kernels, dont_save_grads=True)
        loc.natoms = N
        locs.append(loc)
    ip = g["ind_ptr"]
    X = []
    for q, zc in enumerate(g["ind_z"]):
        k = int(ip[q + 1] - ip[q])
        X.append(Local(0, np.arange(1, k + 1), int(zc), g["ind_nbr_z"][ip[q]:ip[q + 1]].astype(np.int64),
                       torch.tensor(g["ind_nbr_r"][ip[q]:ip[q + 1]].reshape(k, 3)), None, efk.kernels, True))
    rng = np.random.default_rng(120)
    m = len(X)
    subsets = {"a": list(range(0, m, 2)), "live": list(range(1, m, 2)) + [0]}
    out = dict(frame=name)
    members = {}
    for key, idx in subsets.items():
        Xs = [X[i] for i in idx]
        M = efk(Xs, Xs).detach()
        L, ridge = jitcholesky(M)
        choli = L.inverse().contiguous()
        mu = torch.tensor(rng.normal(size=len(idx)) * 0.5)
        indz = np.array([int(x.number) for x in Xs])
        mm = mu * (M @ mu)  # gppotential.py:644-649
        vscale = {int(z): mm[torch.from_numpy(indz == z)].sum() / int((indz == z).sum()) for z in set(indz.tolist())}
        w = {int(z): float(rng.normal() * 0.1) for z in set(int(q) for q in numbers)}
        mean = (lambda ww: (lambda atoms: torch.tensor(sum(ww[int(z)] for z in atoms.numbers))))(w)
        members[key] = SimpleNamespace(choli=choli, mu=mu, _vscale=vscale, mean=mean, cov=efk(locs, Xs))
        out[f"{key}_idx"] = np.array(idx, np.int32)
        out[f"{key}_mu"] = mu.numpy()
        out[f"{key}_choli"] = choli.numpy()
        out[f"{key}_ridge"] = float(ridge)
        out[f"{key}_vscale_z"] = np.array(sorted(vscale), np.int32)
        out[f"{key}_vscale"] = np.array([float(vscale[z]) for z in sorted(vscale)])
        out[f"{key}_mean_z"] = np.array(sorted(w), np.int32)
        out[f"{key}_mean_w"] = np.array([w[z] for z in sorted(w)])
    vol = abs(np.linalg.det(cell))
    duck = SimpleNamespace(
        model_dict={"a": members["a"]}, K_sm={"a": members["a"].cov}, model=members["live"], cov=members["live"].cov,
        atoms=SimpleNamespace(numbers=numbers, is_distributed=False, xyz=xyz, lll=lll, get_volume=lambda: vol),
        gather=lambda x: x, normalized=True, results={}, maximum_force=None, log=lambda *a, **k: None)
    for fn in ("update_results", "get_covloss_with_model", "get_covloss", "grads"):
        setattr(duck, fn, types.MethodType(getattr(BCMActiveCalculator, fn), duck))
    energy, covloss_max = duck.update_results()
    out.update(energy=float(energy), forces=np.asarray(duck.results["forces"]), stress=np.asarray(duck.results["stress"]),
               covloss_max=float(covloss_max),
               covloss_a=duck.get_covloss_with_model("a", duck.K_sm["a"]).detach().numpy(),
               covloss_live=duck.get_covloss().detach().numpy())
    np.savez_compressed(os.path.join(OUT, "g12_bcm.npz"), **out)
    print(f"g12_bcm done: E={float(energy):.6f} |F|max={np.abs(out['forces']).max():.4f} covmax: a={out['covloss_a'].max():.4f} "
          f"live={out['covloss_live'].max():.4f}")


# ----------------------------------------------------------------------------- G10 .sgpr tape
def g10_tape():
    """Text produced by the reference's own tape writer (io/sgprio.py:16-22, :67-90) for three
    LCEs and a params block.  The `atoms` block cannot be produced here (the reference delegates
    it to ase.io's extended-XYZ writer and ASE is absent): it is written below in that format's
    documented layout (Lattice / Properties / energy / stress 3x3 / pbc, species + pos + forces)."""
    from theforce.io.sgprio import SgprIO
    path = os.path.join(OUT, "g10_tape.sgpr")
    if os.path.exists(path):
        os.remove(path)
    g = np.load(os.path.join(OUT, "g5_mixed64.npz"))
    kern, efk = make_kernel()
    ip = g["ind_ptr"]
    tape = SgprIO(path)
    rng = np.random.default_rng(100)
    numbers = np.array([3, 15, 16, 16], np.int32)
    pos = rng.uniform(0, 5, size=(4, 3))
    cell = np.array([[6.0, 0, 0], [0.5, 6.5, 0], [0, 0.25, 7.0]])
    forces = rng.normal(size=(4, 3))
    stress = rng.normal(size=6) * 0.01
    energy = -12.345678901234
    sym = {3: "Li", 15: "P", 16: "S"}
    voigt = [(0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1)]
    s33 = np.zeros((3, 3))
    for v, (i, j) in zip(stress, voigt):
        s33[i, j] = s33[j, i] = v
    with open(path, "a") as f:
        f.write("\nstart: atoms\n4\n")
        f.write('Lattice="{}" Properties=species:S:1:pos:R:3:forces:R:3 energy={!r} stress="{}" pbc="T T F"\n'.format(
            " ".join(repr(float(v)) for v in cell.reshape(-1)), energy, " ".join(repr(float(v)) for v in s33.reshape(-1))))
        for z, p, fo in zip(numbers, pos, forces):
            f.write("{:<2s}      {:16.8f} {:16.8f} {:16.8f} {:16.8f} {:16.8f} {:16.8f}\n".format(sym[int(z)], *p, *fo))
        f.write("end: atoms\n")
    loc_z, loc_ptr, loc_b, loc_r = [], [0], [], []
    for q in (0, 5, 11):
        k = int(ip[q + 1] - ip[q])
        b = g["ind_nbr_z"][ip[q]:ip[q + 1]].astype(np.int64)
        r = g["ind_nbr_r"][ip[q]:ip[q + 1]].reshape(k, 3)
        loc = Local(0, np.arange(1, k + 1), int(g["ind_z"][q]), b, torch.tensor(r), None, efk.kernels, True)
        tape.write(loc)
        loc_z.append(int(g["ind_z"][q]))
        loc_b.append(b)
        loc_r.append(r)
        loc_ptr.append(loc_ptr[-1] + k)
    tape.write_params(ediff=0.086, fdiff=0.129)
    # what the %16.8f text can hold of the positions / forces
    np.savez_compressed(os.path.join(OUT, "g10_tape.npz"), n_local=3, loc_z=np.array(loc_z, np.int32),
                        loc_ptr=np.array(loc_ptr, np.int64), loc_b=np.concatenate(loc_b).astype(np.int32),
                        loc_r=np.concatenate(loc_r), at_numbers=numbers, at_positions=np.round(pos, 8),
                        at_forces=np.round(forces, 8), at_stress=stress, at_cell=cell, at_energy=energy,
                        at_pbc=np.array([True, True, False]))
    print("g10_tape done:", path)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "kat", "frames", "g7", "g9", "rows", "g8", "g10"]
    if "g1" in which:
        g1_ylm()
    if "g2" in which:
        g2_sesoap()
    if "g3" in which:
        g3_subsesoap()
    if "kat" in which:
        kat_absseries()
    if "frames" in which:
        frames()
    if "g7" in which:
        g7_regression()
    if "g9" in which:
        g9_distributer()
    if "rows" in which:
        for nm in ("g5_big40", "g5_bigtric36", "g5_cluster16"):
            kernel_rows(nm)
    if "g8" in which:
        g8_edits()
    if "g11" in which:
        g11_acceptance()
    if "g12" in which:
        g12_bcm()
    if "g10" in which:
        g10_tape()
    if "g13" in which:
        g13_subsesoap_kernel()
    if "g14" in which:
        g14_hpo()
