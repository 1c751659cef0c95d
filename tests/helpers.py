"""Test-only helpers: an oracle-backed engine with the calculator's engine interface, so the host
logic (sharding, packing, all-reduce, logging) can be exercised on CPU with gloo."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import oracle as orc
from autoforce_amd.sharding import shard_indices


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


def OracleEngine(g, mean=None):
    """An OracleModel filled from a golden frame fixture (inducing LCEs, mu, choli, vscale)."""
    from autoforce_amd.model import Local
    eng = OracleModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]), species=g["species"].tolist())
    ptr = g["ind_ptr"]
    eng.set_inducing([Local(int(z), g["ind_nbr_z"][ptr[q]:ptr[q + 1]], g["ind_nbr_r"][ptr[q]:ptr[q + 1]])
                      for q, z in enumerate(g["ind_z"])])
    eng.set_weights(g["mu"], mean=mean, vscale=dict(zip(g["vscale_z"].tolist(), g["vscale"].tolist())),
                    choli=g["choli"])
    return eng


class OracleModel:
    """The CPU oracle behind SGPRModel's interface (set/add/remove/select inducing, kernel rows,
    solve, predict ...), so that PosteriorPotential and the ActiveCalculator control flow can be
    exercised without a GPU.  Test infrastructure only."""

    def __init__(self, lmax=3, nmax=3, exponent=4, cutoff=6.0, species=None, radii=None, device=0):
        self.lmax, self.nmax, self.exponent, self.cutoff = int(lmax), int(nmax), float(exponent), float(cutoff)
        self.species = [int(z) for z in species]
        self.radii = orc.default_radii(self.species) if radii is None else np.asarray(radii, float)
        self.device = device
        self.X, self.mu, self.choli, self.ridge, self.sigma = [], None, None, 0.0, None
        self.mean = {z: 0.0 for z in self.species}
        self._vscale = {}
        self._Pm, self._nnm = np.zeros((0, 1)), np.zeros(0, np.int32)
        self._nl = None
        self.calls = 0

    def with_species(self, species):
        new = OracleModel(self.lmax, self.nmax, self.exponent, self.cutoff, species)
        new.mean.update(self.mean)
        new._vscale = dict(self._vscale)
        if self.X:
            new.set_inducing(self.X)
            if self.mu is not None:
                new.set_weights(self.mu, mean=self.mean, vscale=self._vscale or None, choli=self.choli)
        new.ridge, new.sigma = self.ridge, self.sigma
        return new

    # ---- inducing set
    def _csr(self, X):
        ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])]).astype(np.int64)
        z = np.concatenate([x._b for x in X] + [np.zeros(0, np.int32)]).astype(np.int32)
        r = np.concatenate([x._r for x in X] + [np.zeros((0, 3))])
        return np.array([x.number for x in X], np.int32), ptr, z, r

    def set_inducing(self, X):
        self.X = list(X)
        if self.X:
            self._Pm, self._nnm = orc.inducing_descriptors(self.lmax, self.nmax, self.cutoff, self.species,
                                                           *self._csr(self.X), radii=self.radii)
        else:
            self._Pm, self._nnm = np.zeros((0, 1)), np.zeros(0, np.int32)
        self.mu = self.choli = None

    def add_inducing(self, loc):
        self.set_inducing(self.X + [loc])

    def remove_inducing(self, index=-1):
        X = list(self.X)
        del X[index]
        self.set_inducing(X)

    def select_inducing(self, indices):
        self.set_inducing([self.X[int(i)] for i in indices])

    @property
    def m(self):
        return len(self.X)

    @property
    def _ind_z(self):
        return np.array([x.number for x in self.X], np.int32)

    @property
    def M(self):
        return orc.kernel_matrix(self._ind_z, self._nnm, self._Pm, self._ind_z, self._nnm, self._Pm, self.exponent)

    @property
    def M_rowsum(self):
        return self.M.sum(axis=1)

    def kernel_local(self, loc):
        P, nn = orc.inducing_descriptors(self.lmax, self.nmax, self.cutoff, self.species, *self._csr([loc]),
                                         radii=self.radii)
        z = np.array([loc.number], np.int32)
        k = orc.kernel_matrix(z, nn, P, self._ind_z, self._nnm, self._Pm, self.exponent)[0] if self.m else np.zeros(0)
        return k, float(orc.kernel_matrix(z, nn, P, z, nn, P, self.exponent)[0, 0])

    # ---- rows
    def kernel_rows(self, numbers, positions, cell, pbc):
        return self.kernel_columns(numbers, positions, cell, pbc, 0, self.m)

    def kernel_columns(self, numbers, positions, cell, pbc, q_first, q_count):
        nl = orc.neighbors(positions, cell, pbc, self.cutoff)
        sel = slice(q_first, q_first + q_count)
        return orc.kernel_rows(self.lmax, self.nmax, self.cutoff, self.exponent, self.species, numbers, positions, cell,
                               nl, self._ind_z[sel], self._nnm[sel], self._Pm[sel])

    # ---- weights
    def solve(self, K, Y, noise=0.01):
        self._last_K, self._last_Y = np.asarray(K, float).reshape(-1, self.m), np.asarray(Y, float)
        ref = orc.regression(self.M, self._last_K, self._last_Y, noise0=noise)
        if ref is None:
            raise RuntimeError("cholesky was not successful!")
        self.mu, self.choli, self.ridge, self.sigma = ref["mu"], ref["choli"], ref["ridge"], ref["sigma"]
        self.make_vscale()
        return self.mu

    def resolve(self, noise=0.01):
        return self.solve(self._last_K, self._last_Y, noise=noise)

    def make_vscale(self):
        vs = orc.vscale(self.M, self.mu, self._ind_z, np.array(self.species, np.int32))
        self._vscale = {z: float(v) for z, v in zip(self.species, vs) if np.isfinite(v)}
        return self._vscale

    def set_weights(self, mu, mean=None, vscale=None, choli=None):
        self.mu = np.asarray(mu, float).copy()
        if mean is not None:
            self.mean.update({int(z): float(w) for z, w in mean.items()})
        if vscale is not None:
            self._vscale = {int(z): float(v) for z, v in vscale.items()}
        self.choli = None if choli is None else np.asarray(choli, float).copy()

    def snapshot_weights(self):
        return dict(mu=None if self.mu is None else self.mu.copy(), choli=None if self.choli is None else self.choli.copy(),
                    ridge=self.ridge, sigma=self.sigma, vscale=dict(self._vscale), mean=dict(self.mean))

    def restore_weights(self, snap):
        """The end of a rejected trial: the state saved before it (SGPRModel.restore_weights)."""
        self.mu, self.choli = snap["mu"].copy(), snap["choli"].copy()
        self.ridge, self.sigma = snap["ridge"], snap["sigma"]
        self._vscale = dict(snap["vscale"])
        self.mean.update(snap["mean"])

    # ---- evaluation
    def predict(self, numbers, positions, cell, pbc, rank=0, world=1, cov=False, beta=True):
        self.calls += 1
        numbers = np.asarray(numbers, np.int32)
        N = len(numbers)
        ptr, j, off = orc.neighbors(positions, cell, pbc, self.cutoff)
        self._nl = (ptr, j, off)
        mine = np.zeros(N, bool)
        mine[shard_indices(numbers, self.species, rank, world)] = True
        keep = np.repeat(mine, np.diff(ptr))
        ptr2 = np.concatenate([[0], np.cumsum(np.diff(ptr) * mine)]).astype(np.int64)
        self._nl_mine = (ptr2, j[keep], off[keep])
        if self.m == 0 or self.mu is None:
            return dict(energy=0.0, forces=np.zeros((N, 3)), stress=np.zeros(6), beta=np.zeros(N) if beta else None,
                        cov=np.zeros((N, self.m)) if cov else None)
        out = orc.frame(self.lmax, self.nmax, self.cutoff, self.exponent, self.species, numbers, positions, cell,
                        self._nl_mine, self._ind_z, self._nnm, self._Pm, self.mu, choli=self.choli, radii=self.radii,
                        want_p=False)
        K = out["cov"] * mine[:, None]
        self._last_cov = K
        e = float((K @ self.mu).sum())
        if rank == 0:
            e += sum(self.mean.get(int(z), 0.0) for z in numbers)
        b = None
        if beta and out["beta"] is not None:
            b = out["beta"] * np.sqrt([self._vscale.get(int(z), np.inf) for z in numbers])
            b[~mine] = 0.0
        elif beta:
            b = np.zeros(N)
        return dict(energy=e, forces=out["forces"], stress=out["stress"], beta=b, cov=K if cov else None)

    def last_cov(self, N):
        return self._last_cov

    def neighbors(self, N):
        return self._nl_mine

    def scratch(self):
        return OracleModel(self.lmax, self.nmax, self.exponent, self.cutoff, self.species, self.radii)

    def close(self):
        pass


class PairTeacher:
    """A smooth two-body 'ab initio' stand-in with analytic forces and stress (ASE-calculator
    protocol: get_property(name, atoms)).  phi(r) = eps[(1 - e^{-a(r - r0)})^2 - 1] * (1 - (r/rc)^2)^2."""
    implemented_properties = ["energy", "forces", "stress", "free_energy"]

    def __init__(self, rc=4.0, eps=0.4, a=1.3, r0=2.6):
        self.rc, self.eps, self.a, self.r0 = rc, eps, a, r0
        self.calls = 0
        self.results = {}
        self._key = None

    def _phi(self, r):
        x = np.exp(-self.a * (r - self.r0))
        m, dm = self.eps * ((1 - x) ** 2 - 1), self.eps * 2 * (1 - x) * self.a * x
        s = (1 - (r / self.rc) ** 2)
        c, dc = s * s, -4 * s * r / self.rc ** 2
        return m * c, dm * c + m * dc

    def calculate(self, atoms):
        self.calls += 1
        pos, cell = np.asarray(atoms.positions, float), np.asarray(getattr(atoms.cell, "array", atoms.cell), float)
        ptr, j, off = orc.neighbors(pos, cell, np.asarray(atoms.pbc, bool), self.rc)
        i = np.repeat(np.arange(len(pos)), np.diff(ptr))
        d = pos[j] - pos[i] + off @ cell
        r = np.linalg.norm(d, axis=1)
        phi, dphi = self._phi(r)
        g = (dphi / r)[:, None] * d  # d phi / d r_vec for each directed pair
        F = np.zeros_like(pos)
        np.add.at(F, i, g)  # both directions are listed: each gets half of the pair energy
        vir = 0.5 * np.einsum("pa,pb->ab", d, g)
        vol = abs(np.linalg.det(cell))
        stress = (vir / vol)[[0, 1, 2, 1, 0, 0], [0, 1, 2, 2, 2, 1]] if vol > 0 else np.zeros(6)
        self.results = dict(energy=0.5 * phi.sum(), forces=F, stress=stress, free_energy=0.5 * phi.sum())

    def get_property(self, name, atoms=None):
        key = None if atoms is None else (atoms.positions.tobytes(), np.asarray(atoms.cell).tobytes())
        if atoms is not None and key != self._key:
            self.calculate(atoms)
            self._key = key
        return self.results[name]
