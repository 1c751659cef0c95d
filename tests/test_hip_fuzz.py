"""Randomised differential test of the whole predict path against the pinned oracle: random
triclinic cells (some thinner than the cutoff), every periodicity pattern, one to four species
(hydrogen with its 0.5 length unit among them), atoms sitting exactly on a neighbour's z axis
(the shear quirk of descriptor/ylm.py:10-23), lone atoms in open directions."""
import numpy as np
import pytest

from test_hip_paths import build, compare

pytestmark = pytest.mark.gpu

SPECIES_POOL = [1, 3, 8, 14, 16, 29]


@pytest.mark.parametrize("seed", range(16))
def test_random_system_matches_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    S = int(rng.integers(1, 5))
    species = sorted(rng.choice(SPECIES_POOL, size=S, replace=False).tolist())
    n = int(rng.integers(6, 70))
    rc = float(rng.choice([3.5, 4.5, 5.0]))
    # a cell with edges between 0.7 rc and 2.6 rc and a random shear
    L = rng.uniform(0.7, 2.6, size=3) * rc
    cell = np.diag(L) + np.tril(rng.uniform(-0.25, 0.25, size=(3, 3)) * L.min(), -1)
    pbc = [bool(b) for b in rng.integers(0, 2, size=3)]
    frac = rng.random((n, 3))
    pos = frac @ cell
    from oracle import oracle as orc
    for _ in range(200):  # push apart close pairs (periodic images included)
        ptr, j, off = orc.neighbors(pos, cell, pbc, 1.3)
        if len(j) == 0:
            break
        i = np.repeat(np.arange(n), np.diff(ptr))
        d = pos[j] - pos[i] + off.astype(float) @ cell
        np.add.at(pos, i, -0.25 * d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9))
    if seed % 3 == 0 and n > 8:  # one atom straight above another: the shear branch
        pos[1] = pos[0] + np.array([0.0, 0.0, 1.9])
    numbers = rng.choice(species, size=n).astype(np.int32)
    eta = float(rng.choice([2.0, 4.0]))
    m = int(min(n, rng.integers(3, 14)))
    mdl, nl = build(3, 3, eta, rc, species, numbers, pos, cell, pbc, m, seed + 7)
    compare(mdl, 3, 3, eta, rc, numbers, pos, cell, pbc, nl, tol=2e-8)
    mdl.close()


@pytest.mark.parametrize("seed", range(4))
def test_random_training_rows_match_oracle(seed):
    """K_e / K_f / K_v of random periodic systems (exact reverse pass per column) against the oracle."""
    from oracle import oracle as orc
    rng = np.random.default_rng(2000 + seed)
    species = sorted(rng.choice(SPECIES_POOL, size=int(rng.integers(1, 4)), replace=False).tolist())
    n, rc = int(rng.integers(10, 40)), 4.0
    L = rng.uniform(0.9, 2.2, size=3) * rc
    cell = np.diag(L) + np.tril(rng.uniform(-0.2, 0.2, size=(3, 3)) * L.min(), -1)
    pbc = [True, True, bool(seed % 2)]
    pos = rng.random((n, 3)) @ cell
    for _ in range(200):
        ptr, j, off = orc.neighbors(pos, cell, pbc, 1.3)
        if len(j) == 0:
            break
        i = np.repeat(np.arange(n), np.diff(ptr))
        d = pos[j] - pos[i] + off.astype(float) @ cell
        np.add.at(pos, i, -0.25 * d / np.maximum(np.linalg.norm(d, axis=1, keepdims=True), 1e-9))
    numbers = rng.choice(species, size=n).astype(np.int32)
    mdl, nl = build(3, 3, 4.0, rc, species, numbers, pos, cell, pbc, min(n, 9), seed)
    X = mdl.X
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(3, 3, rc, np.array(species, np.int32), ind_z, ind_ptr,
                                       np.concatenate([x._b for x in X]), np.concatenate([x._r for x in X]))
    want = orc.kernel_rows(3, 3, rc, 4.0, np.array(species, np.int32), numbers, pos, cell, nl, ind_z, nnm, Pm)
    got = mdl.kernel_rows(numbers, pos, cell, pbc)
    np.testing.assert_allclose(got[0], want[0], rtol=1e-10, atol=1e-13)
    for a, b in zip(got[1:], want[1:]):
        assert np.abs(a - b).max() <= 1e-8 * max(np.abs(b).max(), 1e-12)
    mdl.close()
