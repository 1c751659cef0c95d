"""Test-only helpers: an oracle-backed engine with the calculator's engine interface, so the host
logic (sharding, packing, all-reduce, logging) can be exercised on CPU with gloo."""
import os

import numpy as np

from conftest import GOLDEN
from oracle import oracle as orc
from autoforce_amd.sharding import shard_indices


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


class OracleEngine:
    """Same contract as SGPRModel.predict(..., rank, world): PARTIAL sums over this rank's atoms
    (mean term on rank 0 only, beta/cov zero outside the share)."""

    def __init__(self, g, mean=None):
        self.g = g
        self.lmax, self.nmax = int(g["lmax"]), int(g["nmax"])
        self.eta, self.rc = float(g["eta"]), float(g["rc"])
        self.species = [int(z) for z in g["species"]]
        self.Pm, self.nnm = orc.inducing_descriptors(self.lmax, self.nmax, self.rc, self.species, g["ind_z"],
                                                     g["ind_ptr"], g["ind_nbr_z"], g["ind_nbr_r"])
        self.m = len(g["ind_z"])
        self.mean = mean or {}
        self.calls = 0

    def predict(self, numbers, positions, cell, pbc, rank=0, world=1, cov=False, beta=True):
        self.calls += 1
        g = self.g
        N = len(numbers)
        ptr, j, off = orc.neighbors(positions, cell, pbc, self.rc)
        mine = np.zeros(N, bool)
        mine[shard_indices(numbers, self.species, rank, world)] = True
        counts = np.diff(ptr) * mine
        keep = np.repeat(mine, np.diff(ptr))
        ptr2 = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
        out = orc.frame(self.lmax, self.nmax, self.rc, self.eta, self.species, numbers, positions, cell,
                        (ptr2, j[keep], off[keep]), g["ind_z"], self.nnm, self.Pm, g["mu"], choli=g["choli"])
        # atoms outside the share have empty environments in this call: drop their lone-atom
        # kernel entries and beta
        K = out["cov"] * mine[:, None]
        e = float((K @ g["mu"]).sum())
        if rank == 0:
            e += sum(self.mean.get(int(z), 0.0) for z in numbers)
        vs = dict(zip(g["vscale_z"].tolist(), g["vscale"].tolist()))
        b = out["beta"] * np.sqrt([vs.get(int(z), np.inf) for z in numbers]) * mine
        b[~mine] = 0.0
        return dict(energy=e, forces=out["forces"], stress=out["stress"], beta=b, cov=K)
