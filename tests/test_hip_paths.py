"""GPU tests that force the less-travelled code paths of the HIP library, each against the pinned CPU
oracle on identical inputs: 5 species (the 8-slot kernel instantiation), lmax=nmax=4, environments
with more than 64 neighbours (multi-tile descriptor passes, neighbour capacity growth), a cell much
smaller than the cutoff (many periodic images, more than 64 bins in the sweep), a dense cluster in
ONE bin (bin-capacity growth), degenerate inputs (no atoms, no inducing set)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def random_frame(rng, n, box, species, dmin=1.4, pbc=True, cell=None):
    pos = rng.random((n, 3)) * box
    cell = np.eye(3) * box if cell is None else cell
    from oracle import oracle as orc
    for _ in range(300):  # push apart close pairs
        ptr, j, off = orc.neighbors(pos, cell, [pbc] * 3, dmin)
        if len(j) == 0:
            break
        i = np.repeat(np.arange(n), np.diff(ptr))
        d = pos[j] - pos[i] + off.astype(float) @ cell
        np.add.at(pos, i, -0.2 * d / np.linalg.norm(d, axis=1, keepdims=True))
    numbers = rng.choice(species, size=n).astype(np.int32)
    return numbers, pos, cell


def build(lmax, nmax, eta, rc, species, numbers, pos, cell, pbc, m, seed, radii=None):
    from autoforce_amd import Local, SGPRModel
    from oracle import oracle as orc
    rng = np.random.default_rng(seed)
    ptr, j, off = orc.neighbors(pos, cell, pbc, rc)
    idx = rng.choice(len(numbers), size=m, replace=False)
    X = []
    for a in idx:
        s = slice(ptr[a], ptr[a + 1])
        r = pos[j[s]] - pos[a] + off[s].astype(float) @ cell + 0.03 * rng.normal(size=(ptr[a + 1] - ptr[a], 3))
        keep = np.linalg.norm(r, axis=1) < rc - 1e-3
        X.append(Local(int(numbers[a]), numbers[j[s]][keep], r[keep]))
    mdl = SGPRModel(lmax, nmax, eta, rc, species=species, radii=radii)
    mdl.set_inducing(X)
    return mdl, (ptr, j, off)


def compare(mdl, lmax, nmax, eta, rc, numbers, pos, cell, pbc, nl, tol=1e-8, radii=None):
    from oracle import oracle as orc
    X = mdl.X
    species = np.array(mdl.species, np.int32)
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(lmax, nmax, rc, species, ind_z, ind_ptr,
                                       np.concatenate([x._b for x in X]), np.concatenate([x._r for x in X]), radii=radii)
    M = orc.kernel_matrix(ind_z, nnm, Pm, ind_z, nnm, Pm, eta)
    np.testing.assert_allclose(mdl.M, M, rtol=1e-9, atol=1e-12)
    L, ridge = orc.jitcholesky(M)
    choli = orc.tril_inverse(L)
    mu = np.random.default_rng(9).normal(size=len(X))
    mdl.set_weights(mu, choli=choli)
    out = mdl.predict(numbers, pos, cell, pbc, cov=True)
    N = len(numbers)
    p, j, off = mdl.neighbors(N)
    i = np.repeat(np.arange(N), np.diff(p))
    i0 = np.repeat(np.arange(N), np.diff(nl[0]))
    got = set(map(tuple, np.column_stack([i, j, off]).tolist()))
    want = set(map(tuple, np.column_stack([i0, nl[1], nl[2]]).tolist()))
    assert got == want
    ref = orc.frame(lmax, nmax, rc, eta, species, numbers, pos, cell, nl, ind_z, nnm, Pm, mu, choli=choli, radii=radii)
    np.testing.assert_allclose(out["cov"], ref["cov"], rtol=1e-9, atol=1e-12)
    assert abs(out["energy"] - ref["energy"]) <= 1e-9 * max(1.0, abs(ref["energy"]))
    assert np.abs(out["forces"] - ref["forces"]).max() <= tol * np.abs(ref["forces"]).max()
    assert np.abs(out["stress"] - ref["stress"]).max() <= tol * max(np.abs(ref["stress"]).max(), 1e-12)
    np.testing.assert_allclose(out["beta"], ref["beta"], rtol=0, atol=3e-6)
    return out


def test_five_species_uses_the_8_slot_kernels():
    rng = np.random.default_rng(1)
    species = [1, 6, 7, 8, 16]
    numbers, pos, cell = random_frame(rng, 96, 9.5, species)
    mdl, nl = build(3, 3, 4.0, 5.0, species, numbers, pos, cell, [True] * 3, 20, 2)
    compare(mdl, 3, 3, 4.0, 5.0, numbers, pos, cell, [True] * 3, nl)
    mdl.close()


@pytest.mark.parametrize("lmax,nmax,nspec", [(4, 4, 6), (2, 3, 5), (3, 4, 8), (4, 2, 7), (2, 2, 8), (3, 2, 5), (2, 4, 6), (4, 3, 5),
                                             (3, 3, 10), (3, 3, 16)])
def test_more_than_four_species_for_every_lmax_nmax(lmax, nmax, nspec):
    """The reference's kernels take any species (its table is 120 wide, descriptor/sesoap.py:134) and any (lmax, nmax)
    (similarity/sesoap.py:10-24).  Here every pair of {2,3,4}^2 is compiled for up to eight species slots: frames with five to
    eight species against the oracle, K_mm included, and the training rows (the one-column-per-wave form: the sixteen-column
    kernel is compiled for up to four slots).  The reference's default pair (3, 3) goes on to sixteen slots (ten and sixteen
    species here: descriptor rows of 8320 doubles, the reverse kernel one atom per workgroup)."""
    rng = np.random.default_rng(100 * lmax + 10 * nmax + nspec)
    species = [1, 6, 7, 8, 16, 3, 9, 15, 11, 12, 13, 14, 17, 19, 20, 29][:nspec]
    numbers, pos, cell = random_frame(rng, 112, 10.0, species)
    mdl, nl = build(lmax, nmax, 4.0, 5.0, species, numbers, pos, cell, [True] * 3, 24, 2)
    compare(mdl, lmax, nmax, 4.0, 5.0, numbers, pos, cell, [True] * 3, nl)
    from oracle import oracle as orc
    X = mdl.X
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(lmax, nmax, 5.0, np.array(species, np.int32), ind_z, ind_ptr,
                                       np.concatenate([x._b for x in X]), np.concatenate([x._r for x in X]))
    Ke, Kf, Kv = mdl.kernel_rows(numbers, pos, cell, [True] * 3)
    ref = orc.kernel_rows(lmax, nmax, 5.0, 4.0, np.array(species, np.int32), numbers, pos, cell, nl, ind_z, nnm, Pm)
    np.testing.assert_allclose(Ke, ref[0], rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(Kf, ref[1], rtol=0, atol=1e-8 * np.abs(ref[1]).max())
    np.testing.assert_allclose(Kv, ref[2], rtol=0, atol=1e-8 * np.abs(ref[2]).max())
    mdl.close()


def test_non_integer_exponent_and_custom_radii():
    """eta = 2.5 takes the pow() branch of the kernel epilogue (integer exponents use repeated
    multiplication); radii other than the defaults exercise the per-species length unit
    (descriptor/sesoap.py:84-99 allows any table) where r/u is not exact."""
    rng = np.random.default_rng(11)
    species = [3, 8, 15]
    radii = np.array([0.8, 1.3, 1.0])
    numbers, pos, cell = random_frame(rng, 80, 9.0, species)
    mdl, nl = build(3, 3, 2.5, 5.0, species, numbers, pos, cell, [True] * 3, 18, 5, radii=radii)
    compare(mdl, 3, 3, 2.5, 5.0, numbers, pos, cell, [True] * 3, nl, radii=radii)
    mdl.close()


def test_lmax4_nmax4():
    rng = np.random.default_rng(3)
    species = [13, 8]
    numbers, pos, cell = random_frame(rng, 60, 8.4, species)
    mdl, nl = build(4, 4, 2.0, 5.0, species, numbers, pos, cell, [True] * 3, 12, 4)
    compare(mdl, 4, 4, 2.0, 5.0, numbers, pos, cell, [True] * 3, nl)
    mdl.close()


def test_more_than_64_neighbours_and_capacity_growth():
    """rc = 7.5 on a dense frame: ~110 neighbours per atom (two descriptor tiles, list capacity grows
    from its initial 64)."""
    rng = np.random.default_rng(5)
    species = [3, 16]
    numbers, pos, cell = random_frame(rng, 120, 12.0, species, dmin=1.8)
    mdl, nl = build(3, 3, 4.0, 7.5, species, numbers, pos, cell, [True] * 3, 16, 6)
    assert np.diff(nl[0]).max() > 64
    compare(mdl, 3, 3, 4.0, 7.5, numbers, pos, cell, [True] * 3, nl)
    assert mdl.dims["maxnn"] >= np.diff(nl[0]).max()
    mdl.close()


def test_tiny_cell_many_images():
    """3-atom triclinic cell with heights ~2.6 A and rc = 6: every neighbour is a periodic image,
    the sweep covers 5x5x5 = 125 (> 64) image bins, lists hold > 256 entries."""
    species = [29]
    cell = np.array([[2.7, 0.0, 0.0], [0.4, 2.8, 0.0], [0.3, -0.5, 2.9]])
    pos = np.array([[0.1, 0.2, 0.1], [1.4, 1.5, 1.2], [2.2, 0.3, 2.0]])
    numbers = np.array([29, 29, 29], np.int32)
    from autoforce_amd import Local, SGPRModel
    from oracle import oracle as orc
    nl = orc.neighbors(pos, cell, [True] * 3, 6.0)
    assert np.diff(nl[0]).min() > 100
    rng = np.random.default_rng(7)
    X = []
    for a in range(3):
        s = slice(nl[0][a], nl[0][a + 1])
        r = pos[nl[1][s]] - pos[a] + nl[2][s].astype(float) @ cell + 0.02 * rng.normal(size=(nl[0][a + 1] - nl[0][a], 3))
        keep = np.linalg.norm(r, axis=1) < 6.0 - 1e-3
        X.append(Local(29, numbers[nl[1][s]][keep], r[keep]))
    mdl = SGPRModel(3, 3, 4.0, 6.0, species=species)
    mdl.set_inducing(X)
    compare(mdl, 3, 3, 4.0, 6.0, numbers, pos, cell, [True] * 3, nl)
    mdl.close()


def test_dense_cluster_in_one_bin():
    """150 atoms, no periodicity: one bin holds everything (bin capacity grows from 64)."""
    rng = np.random.default_rng(11)
    species = [79, 47]
    numbers, pos, cell = random_frame(rng, 150, 14.0, species, dmin=2.2, pbc=False)
    zero = np.zeros((3, 3))
    mdl, nl = build(3, 3, 4.0, 6.0, species, numbers, pos, zero, [False] * 3, 14, 12)
    compare(mdl, 3, 3, 4.0, 6.0, numbers, pos, zero, [False] * 3, nl)
    mdl.close()


def test_slab_cell_with_zero_open_vector():
    """cell = [a, b, 0] with pbc = TTF: the device completes the open direction (ase complete_cell) and keeps
    the in-plane images; same numbers as with an explicit out-of-plane vector and as the oracle.  A zero
    vector along a periodic direction is an error."""
    from autoforce_amd import SgprError
    rng = np.random.default_rng(7)
    inplane = np.array([[7.0, 0.3, 0.0], [-0.5, 6.6, 0.0]])
    cell0 = np.vstack([inplane, np.zeros(3)])
    cell1 = np.vstack([inplane, [0.0, 0.0, 40.0]])
    pos = rng.uniform(0, 1, (30, 3)) @ np.vstack([inplane, [0.0, 0.0, 3.0]])
    numbers = rng.choice([3, 16], 30).astype(np.int32)
    mdl, X = build(3, 3, 4.0, 6.0, [3, 16], numbers, pos, cell1, [True, True, False], m=8, seed=5)
    mdl.set_weights(rng.normal(size=8))
    a = mdl.predict(numbers, pos, cell0, [True, True, False])
    b = mdl.predict(numbers, pos, cell1, [True, True, False])
    assert abs(a["energy"] - b["energy"]) <= 1e-12 * max(1.0, abs(b["energy"]))
    np.testing.assert_allclose(a["forces"], b["forces"], rtol=0, atol=1e-12 * np.abs(b["forces"]).max())
    assert mdl.neighbors(30)[0][-1] > 30 * 20
    with pytest.raises(SgprError):
        mdl.predict(numbers, pos, cell0, [True, True, True])
    mdl.close()


@pytest.mark.parametrize("name", ["g5_mixed64", "g5_si32", "g5_tric24"])
def test_candidate_lists_are_reused_without_changing_anything(name):
    """Verlet candidates (|r| < rc + skin, rebuilt on the device when an atom has moved more than skin/2 or the
    cell changed): over an MD-like walk — small moves, one jump, a cell strain, a wrap through the cell — every
    step gives bit for bit what a model that rebuilds its lists every step gives, and the same neighbour list."""
    from autoforce_amd import _lib
    from test_hip_parity import load, model_from_fixture
    g = load(name)
    fast, slow = model_from_fixture(g), model_from_fixture(g)
    _lib.check(_lib.load().sgpr_set_option(slow.handle, b"skin_milliangstrom", 0))
    rng = np.random.default_rng(11)
    pos, cell = g["positions"].copy(), g["cell"].copy()
    N = len(pos)
    for step in range(14):
        if step in (1, 2, 3, 4, 5, 6, 9, 10, 12):
            pos = pos + 0.03 * rng.normal(size=pos.shape)          # thermal-size moves: lists are reused
        elif step == 7:
            pos[rng.integers(N)] += np.array([0.9, -0.4, 0.2])      # one atom jumps: rebuild
        elif step == 8:
            cell = cell @ (np.eye(3) + 0.002 * rng.normal(size=(3, 3)))  # strained cell: rebuild
        elif step == 11 and g["pbc"].all():
            pos[0] = pos[0] + cell[0]                                # wrapped through the cell: rebuild
        a = fast.predict(g["numbers"], pos, cell, g["pbc"], cov=True)
        b = slow.predict(g["numbers"], pos, cell, g["pbc"], cov=True)
        for k in ("energy", "forces", "stress", "beta", "cov"):
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=f"step {step}: {k}")
        for x, y in zip(fast.neighbors(N), slow.neighbors(N)):
            np.testing.assert_array_equal(x, y)
    fast.close(); slow.close()


@pytest.mark.parametrize("shape", ["64,64;32,32;8", "32,32;32,32;8", "32,32;16,16;4", "16,16;16,16;8", "r64", "r64x3"])
def test_gemm_tile_shapes_agree_bit_for_bit(shape, monkeypatch):
    """The three products of a step (K_nm, W, covloss) run on 32x64 tiles with 16-deep LDS stages; the 64-row form and the
    32-deep stages stay compiled in for K_mm, dense launches and the forked path.  Every form accumulates a dot product
    over k in the same order (one MFMA k-step of 4 after the other), so K_nm, forces, stress and beta must not move
    by a single bit when the tile tables are built for another shape (diagnostic overrides SGPR_GEMM_BM / SGPR_GEMM_KD),
    nor when a K_nm tile is shared by four waves (two 16 x 16 blocks each) instead of eight (SGPR_GEMM_WAVES), nor on the
    64 x 64 eight-wave tiles that large frames take ("r64": SGPR_GEMM_64 forces them here; "r64x3": on three register stage
    sets, three workgroups per CU), nor on the 16 x 64 half tiles (one wave per SIMD) that launches of fewer tiles than CUs take —
    this frame's default; the other shapes are forced against it."""
    rng = np.random.default_rng(31)
    species = [3, 15, 16]
    numbers, pos, cell = random_frame(rng, 300, 16.0, species)
    pbc = [True] * 3
    outs = []
    for bm, kd, waves in ((None, None, None), shape.split(";") if not shape.startswith("r") else (shape, "", "")):
        if bm is None:
            for k in ("SGPR_GEMM_BM", "SGPR_GEMM_KD", "SGPR_GEMM_WAVES", "SGPR_GEMM_64", "SGPR_GEMM_WGS64"):
                monkeypatch.delenv(k, raising=False)
        elif bm.startswith("r"):
            monkeypatch.setenv("SGPR_GEMM_64", "1,1")
            monkeypatch.setenv("SGPR_GEMM_WGS64", "3" if bm == "r64x3" else "2")   # (three register sets, three workgroups per CU)
        else:
            monkeypatch.setenv("SGPR_GEMM_BM", bm)
            monkeypatch.setenv("SGPR_GEMM_KD", kd)
            monkeypatch.setenv("SGPR_GEMM_WAVES", waves)
        mdl, nl = build(3, 3, 4, 6.0, species, numbers, pos, cell, pbc, 90, seed=5)
        mu = np.random.default_rng(9).normal(size=len(mdl.X))
        mdl.solve(np.random.default_rng(2).normal(size=(40, len(mdl.X))), np.random.default_rng(3).normal(size=40))
        mdl.set_weights(mu, choli=mdl.choli, vscale=mdl.make_vscale())
        outs.append(mdl.predict(numbers, pos, cell, pbc, cov=True))
        mdl.close()
    a, b = outs
    # (the energy is a sum of per-tile partials: its grouping, not its terms, follows the tile shape)
    assert abs(a["energy"] - b["energy"]) <= 4e-16 * abs(a["energy"]) * np.sqrt(len(numbers))
    for key in ("forces", "stress", "beta", "cov"):
        assert np.array_equal(a[key], b[key]), key


def test_degenerate_inputs():
    from autoforce_amd import SGPRModel, SgprError
    mdl = SGPRModel(3, 3, 4.0, 6.0, species=[14])
    # no atoms
    out = mdl.predict(np.zeros(0, np.int32), np.zeros((0, 3)), np.eye(3) * 5, [True] * 3)
    assert out["energy"] == 0.0 and out["forces"].shape == (0, 3)
    # atoms but no inducing set: zero energy and forces, lists still built
    pos = np.array([[0.0, 0, 0], [1.5, 1.5, 1.5]])
    out = mdl.predict(np.array([14, 14], np.int32), pos, np.eye(3) * 4.0, [True] * 3)
    assert out["energy"] == 0.0 and np.all(out["forces"] == 0.0)
    p, j, off = mdl.neighbors(2)
    assert p[-1] > 0
    with pytest.raises(SgprError):
        mdl.set_weights(np.zeros(0))
    mdl.close()
    with pytest.raises(SgprError) as e:
        SGPRModel(5, 3, 4.0, 6.0, species=[14])  # not compiled in
    assert e.value.code == -6


def test_npt_walk_reuses_candidates_under_strain():
    """A barostat strains the cell a little EVERY step (cl/md.py:147-150).  The candidate lists stay valid while the
    non-affine displacement of every atom, u_i = x_i - x_i0 h0^-1 h, stays below (sigma_min(h0^-1 h) (rc + skin) - rc) / 2
    (neighbor.hip): over a 40-step walk with a 1e-4 random strain and thermal-size moves per step the lists are rebuilt
    a handful of times, not 40, and every step gives bit for bit what a handle that rebuilds its lists every step
    gives — same pairs, same order, same sums."""
    from autoforce_amd import _lib
    from test_hip_parity import load, model_from_fixture
    g = load("g5_mixed64")
    fast, slow = model_from_fixture(g), model_from_fixture(g)
    _lib.check(_lib.load().sgpr_set_option(slow.handle, b"skin_milliangstrom", 0))
    rng = np.random.default_rng(23)
    pos, cell = g["positions"].copy(), g["cell"].copy()
    N = len(pos)
    steps = 40
    for step in range(steps):
        strain = np.eye(3) + 1e-4 * rng.normal(size=(3, 3))
        frac = np.linalg.solve(cell.T, pos.T).T
        cell = cell @ strain
        pos = frac @ cell + 0.012 * rng.normal(size=pos.shape)   # affine move with the cell + thermal-size noise
        a = fast.predict(g["numbers"], pos, cell, g["pbc"], cov=True)
        b = slow.predict(g["numbers"], pos, cell, g["pbc"], cov=True)
        for k in ("energy", "forces", "stress", "beta", "cov"):
            np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]), err_msg=f"step {step}: {k}")
        for x, y in zip(fast.neighbors(N), slow.neighbors(N)):
            np.testing.assert_array_equal(x, y)
    assert slow.list_rebuilds() >= steps
    assert fast.list_rebuilds() <= 6, fast.list_rebuilds()
    # a strain beyond the bound (here 5 %: sigma_min * 6.5 < 6) must rebuild, and still agree
    cell = cell @ (np.eye(3) * 1.05)
    pos = pos * 1.05
    before = fast.list_rebuilds()
    a = fast.predict(g["numbers"], pos, cell, g["pbc"])
    b = slow.predict(g["numbers"], pos, cell, g["pbc"])
    assert fast.list_rebuilds() == before + 1
    for k in ("energy", "forces", "stress", "beta"):
        np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]))
    fast.close(); slow.close()


@pytest.mark.parametrize("side,m", [(8, 48), (16, 512)])
def test_fused_gemm_launch_equals_the_three_launches(side, m, monkeypatch):
    """The K_nm, W and covloss products of a step go out as ONE launch whose W / covloss tiles wait, panel by panel, on
    the K_nm tiles ahead of them in the tile list (gemm_tile.inc, EPI_FUSED).  Against the same handle with the option off
    (K_nm launch, then the grouped W + covloss launch): forces, stress and covloss bit for bit on every frame of a walk —
    a consumer that started before its producers had written through would show here — the energy to the rounding of
    its per-tile partials (four per K_nm tile instead of eight), and the stage table shows one GEMM stage."""
    import ctypes as C
    import torch
    from autoforce_amd import _lib
    from test_hip_md import _model
    monkeypatch.setenv("SGPR_GEMM_HALF", "0")   # (the fused launch exists for the 32-row forms; a small frame would take half tiles)
    mdl, (numbers, pos, cell, pbc) = _model(side=side, m=m)
    lib, h, N = _lib.load(), mdl.handle, len(numbers)
    rng = np.random.default_rng(8)
    frames = [pos]
    for _ in range(60):
        frames.append(frames[-1] + 0.01 * rng.normal(size=pos.shape))
    mdl.predict(numbers, pos, cell, pbc)
    dev = torch.device("cuda:0")
    fr, cl = torch.tensor(np.stack(frames), device=dev), torch.tensor(cell, device=dev)
    outs = []
    for fused in (1, 0, 1):
        _lib.check(lib.sgpr_set_option(h, b"gemm_fused", fused))
        out = torch.zeros((len(frames), 4 * N + 11), dtype=torch.float64, device=dev)
        for rep in range(3):   # (the same frames three times over: every repetition must land on the same bits)
            for k in range(len(frames)):
                nxt = fr[k + 1].data_ptr() if k + 1 < len(frames) else None
                _lib.check(lib.sgpr_step_dev_next(h, fr[k].data_ptr(), cl.data_ptr(), out[k].data_ptr(), nxt, None))
            _lib.check(lib.sgpr_sync_check(h, None))
            outs.append(out.cpu().numpy().copy())
        mdl.profile(True)
        mdl.predict(numbers, pos, cell, pbc)
        stages = list(mdl.stage_times())
        mdl.profile(False)
        assert ("gemm_fused" in stages) == bool(fused) and ("gemm_knm" in stages) != bool(fused), stages
    ref = outs[3]   # the three-launch form
    for o in outs:
        e = 4 * N   # packed: [F (3N) | covloss (N) | E | virial (9) | overflow]
        assert np.array_equal(np.delete(o, e, axis=1), np.delete(ref, e, axis=1)), np.abs(o - ref).max()
        assert np.all(np.abs(o[:, e] - ref[:, e]) <= 4e-16 * np.abs(ref[:, e]) * np.sqrt(N) + 1e-13)
    assert np.array_equal(outs[0], outs[1]) and np.array_equal(outs[0], outs[2]) and np.array_equal(outs[0], outs[6])
    mdl.close()


def test_bench_line_keeps_its_contract():
    """`python bench.py` as the driver runs it (fewer steps, a small CPU sample): ONE JSON line on stdout with the metric
    BASELINE.json names, the whole-job value, and the `roofline` / `cpu_baseline` objects with every field the contract
    lists; `frac` = achieved / peak, the value consistent with ms_per_step."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5",
                          "--md-steps", "60", "--no-big-wall", "--cpu-sample", "256"], capture_output=True, text=True,
                         timeout=900, cwd=root)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    base = json.load(open(os.path.join(root, "BASELINE.json")))
    assert base["metric"].startswith("MD-step atoms") and d["metric"].startswith("MD-step atoms*steps/sec")   # BASELINE's metric
    assert d["unit"] == "atom*steps/s"
    for key in ("value", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
                "data", "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5 and d["higher_is_better"] is True
    assert d["dtype"] == "f64" and d["data"].startswith("synthetic") and "workload" in d["config"] and "model" not in d["config"]
    atoms = 4096
    assert abs(d["value"] - atoms / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in r, key
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    c = d["cpu_baseline"]
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in c, key
    assert c["kind"] in ("reference", "port") and c["value"] > 0 and c["cores"] >= 1
    assert d["config"]["launches_per_step"] == 5 and d["value_md_loop"] > 0 and d["value_calculate_wall"] > 0
    # the headline is the dependent MD loop (every step from the forces of the one before); the resident-frames pipeline is
    # reported beside it and is the faster of the two
    assert d["value_is"].startswith("md_loop") and d["value_resident_frames"] > 0 and d["ms_per_step_resident_frames"] > 0
    assert abs(d["value_resident_frames"] - atoms / (d["ms_per_step_resident_frames"] * 1e-3)) <= 1e-6 * d["value_resident_frames"]
    assert d["value"] <= 1.05 * d["value_resident_frames"]
    # the timed region: --steps is the BATCH, repeated until >= 50 ms have been timed; the median batch is the headline
    t = d["timed"]
    assert t["batch_steps"] == 20 and t["min_timed_ms"] >= 50.0
    for leg, ms in ((t["md_loop"], d["ms_per_step"]), (t["resident_frames"], d["ms_per_step_resident_frames"])):
        assert leg["batches"] >= 3 and leg["batches"] * 20 * ms >= 0.5 * t["min_timed_ms"]
        assert leg["ms_per_step_min"] <= ms <= leg["ms_per_step_max"]
    # the flop count behind frac is the block-diagonal one; the dense formula of SURVEY 8(d) is quoted beside it
    if r["bound"] == "mfma":
        assert r["dense_equiv_flops"] > r["algorithmic_flops"] > 0
