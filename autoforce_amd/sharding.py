"""Atom -> rank assignment of the sharded predict path (host logic, numpy only).

The reference deals atoms to MPI ranks with `Distributer` (theforce/descriptor/atoms.py:235-246):
each atom goes to the rank with the smallest (total load, per-species load, rank id), i.e. a
per-species round robin.  libsgpr_hip uses the closed form of that policy for a fresh frame:
stable-sort the atoms by species slot and give rank r the sorted atoms r, r+world, r+2*world, ...
(equal totals +-1, equal per-species loads +-1).  `distributer_ranks` is the reference's stateful
algorithm itself, kept for callers that want identical rank maps across consecutive frames.
"""
import numpy as np


def species_slots(numbers, species):
    table = {int(z): k for k, z in enumerate(species)}
    try:
        return np.array([table[int(z)] for z in numbers], dtype=np.int32)
    except KeyError as e:
        raise ValueError(f"atomic number {e.args[0]} is not in the model's species table") from None


def sorted_order(numbers, species):
    """perm[g] = caller index of the g-th atom in species-sorted order (stable)."""
    return np.argsort(species_slots(numbers, species), kind="stable").astype(np.int32)


def shard_indices(numbers, species, rank, world):
    """Caller indices of the atoms rank `rank` of `world` evaluates (same rule as sgpr_bind_system)."""
    perm = sorted_order(numbers, species)
    return perm[rank::world]


def rank_of_atoms(numbers, species, world):
    perm = sorted_order(numbers, species)
    ranks = np.empty(len(perm), np.int32)
    ranks[perm] = np.arange(len(perm)) % world
    return ranks


def distributer_ranks(numbers, world_size, loads=None, total=None):
    """Distributer.__call__ (theforce/descriptor/atoms.py:235-246), stateful form."""
    loads = {} if loads is None else loads
    total = [0] * world_size if total is None else total
    ranks = []
    for z in numbers:
        z = int(z)
        if z not in loads:
            loads[z] = [0] * world_size
        rank = min(range(world_size), key=lambda r: (total[r], loads[z][r], r))
        ranks.append(rank)
        loads[z][rank] += 1
        total[rank] += 1
    return np.array(ranks, np.int32), loads, total


def pack_partial(out, N):
    """[F(3N) | beta(N) | E | virial-as-stress(6)] as one vector for a single all-reduce(SUM):
    the reference's four collectives (calculator/active.py:562,601,602,777) fused."""
    v = np.zeros(4 * N + 7)
    v[:3 * N] = np.asarray(out["forces"], float).reshape(-1)
    if out.get("beta") is not None:
        v[3 * N:4 * N] = out["beta"]
    v[4 * N] = out["energy"]
    v[4 * N + 1:] = out["stress"]
    return v


def unpack_total(v, N):
    return dict(forces=v[:3 * N].reshape(N, 3).copy(), beta=v[3 * N:4 * N].copy(), energy=float(v[4 * N]),
                stress=v[4 * N + 1:4 * N + 7].copy())
