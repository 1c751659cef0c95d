"""GPU tests of the calculator surface, the device-resident step (HIP graph replay), model IO and
the full-size BASELINE workload (4096 atoms / 512 inducing) against the oracle and through
size-independent properties."""
import ctypes as C
import os

import numpy as np
import pytest

from helpers import load


def pair_set(ptr, j, off):
    i = np.repeat(np.arange(len(ptr) - 1), np.diff(ptr))
    return set(zip(i.tolist(), np.asarray(j).tolist(), map(tuple, np.asarray(off).tolist())))

pytestmark = pytest.mark.gpu


def model_from_fixture(g):
    from autoforce_amd import Local, SGPRModel
    mdl = SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]), species=g["species"].tolist())
    ptr = g["ind_ptr"]
    mdl.set_inducing([Local(int(z), g["ind_nbr_z"][ptr[q]:ptr[q + 1]], g["ind_nbr_r"][ptr[q]:ptr[q + 1]])
                      for q, z in enumerate(g["ind_z"])])
    vs = dict(zip(g["vscale_z"].tolist(), g["vscale"].tolist()))
    mdl.set_weights(g["mu"], vscale=vs, choli=g["choli"])
    return mdl


def test_calculator_on_hip(tmp_path):
    from autoforce_amd.ase_shim import Atoms
    from autoforce_amd.calculator import ActiveCalculator
    g = load("g5_mixed64")
    calc = ActiveCalculator(covariance=model_from_fixture(g), logfile=str(tmp_path / "active.log"))
    atoms = Atoms(g["numbers"], g["positions"], g["cell"], g["pbc"])
    atoms.calc = calc
    assert abs(atoms.get_potential_energy() - float(g["energy"])) < 1e-10
    assert np.abs(atoms.get_forces() - g["forces"]).max() <= 1e-9 * np.abs(g["forces"]).max()
    assert np.abs(atoms.get_stress() - g["stress"]).max() <= 1e-9 * np.abs(g["stress"]).max()
    # calc.cov (active.py:464) is fetched from the device only when somebody looks at it
    assert calc._cov is None
    np.testing.assert_allclose(calc.cov, g["cov"], rtol=1e-10, atol=1e-13)
    ok = np.isfinite(g["covloss"])
    np.testing.assert_allclose(calc.get_covloss()[ok], g["covloss"][ok], rtol=0, atol=1e-6)
    assert calc.step == 1 and calc.size == (0, 24)


def test_model_io_roundtrip(tmp_path):
    from autoforce_amd.modelio import load_model, save_model
    g = load("g5_tric24")
    mdl = model_from_fixture(g)
    ref = mdl.predict(g["numbers"], g["positions"], g["cell"], g["pbc"])
    path = str(tmp_path / "model.npz")
    save_model(path, mdl)
    mdl.close()
    m2 = load_model(path).engine
    out = m2.predict(g["numbers"], g["positions"], g["cell"], g["pbc"])
    for k in ("energy", "forces", "stress", "beta"):
        np.testing.assert_array_equal(np.asarray(out[k]), np.asarray(ref[k]))
    m2.close()


def test_device_step_matches_host_step_and_replays():
    """sgpr_step_dev (device pointers, HIP-graph replay) gives bit-identical packed results to
    sgpr_compute, also after the atoms moved (same graph, new positions in the same buffer)."""
    import torch
    from autoforce_amd import _lib
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    lib = _lib.load()
    N = len(g["numbers"])
    dev = torch.device("cuda", 0)
    pos = torch.from_numpy(g["positions"].copy()).to(dev)
    cell = torch.from_numpy(g["cell"].copy()).to(dev)
    packed = torch.zeros(int(lib.sgpr_packed_len(N)), dtype=torch.float64, device=dev)
    _lib.check(lib.sgpr_bind_system(mdl.handle, N, _lib.ptr(_lib.i32(g["numbers"])),
                                    _lib.ptr(_lib.i32(g["pbc"].astype(np.int32))), 0, 1))
    sp = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rng = np.random.default_rng(0)
    for it in range(4):  # 1st eager+checked, 2nd captures, 3rd/4th replay
        if it >= 2:
            pos += torch.from_numpy(0.02 * rng.normal(size=(N, 3))).to(dev)
        _lib.check(lib.sgpr_step_dev(mdl.handle, pos.data_ptr(), cell.data_ptr(), packed.data_ptr(), sp))
        _lib.check(lib.sgpr_sync_check(mdl.handle, sp))
        got = packed.cpu().numpy().copy()
        ref = mdl.predict(g["numbers"], pos.cpu().numpy(), g["cell"], g["pbc"])
        # predict() re-binds nothing (same system) but leaves the graph valid
        np.testing.assert_array_equal(got[:3 * N].reshape(N, 3), ref["forces"])
        np.testing.assert_array_equal(got[3 * N:4 * N], ref["beta"])
        assert got[4 * N] == ref["energy"]
        s = np.zeros(6)
        lib.sgpr_stress_from_virial(_lib.ptr(got[4 * N + 1:4 * N + 10].copy()), _lib.ptr(g["cell"].copy()), _lib.ptr(s))
        np.testing.assert_array_equal(s, ref["stress"])
    mdl.close()


def _lips_model(n_side, m, seed=1):
    from autoforce_amd import SGPRModel
    from autoforce_amd.workloads import inducing_from_frame, lips
    numbers, pos, cell, pbc = lips(n_side, seed=0)
    mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
    n2, p2, c2, b2 = lips(n_side, seed=seed)
    mdl.set_inducing(inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=seed))
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(64, m)), rng.normal(size=64))
    mu = rng.normal(size=m)
    mdl.set_weights(mu, choli=mdl.choli, vscale=mdl.make_vscale())
    return mdl, numbers, pos, cell, pbc


def test_full_size_step_is_bit_reproducible():
    """4096 / 512: energy, forces, stress AND covloss repeat bit for bit (the covloss row sums go through per-tile
    partials in a fixed order, the forces through the gather-form reverse pass: no floating-point atomics)."""
    mdl, numbers, pos, cell, pbc = _lips_model(16, 512)
    ref = mdl.predict(numbers, pos, cell, pbc)
    assert np.isfinite(ref["beta"]).all() and ref["beta"].max() > 0
    for _ in range(4):
        out = mdl.predict(numbers, pos, cell, pbc)
        for k in ("energy", "forces", "stress", "beta"):
            np.testing.assert_array_equal(np.asarray(out[k]), np.asarray(ref[k]))
    mdl.close()


def test_full_size_lips4096_against_oracle():
    """BASELINE configs[2] sizes: 4096 atoms, 3 species, 512 inducing.  north_star tolerance:
    forces within 1e-6 relative of the CPU path; held to 1e-8 here."""
    from oracle import oracle as orc
    mdl, numbers, pos, cell, pbc = _lips_model(16, 512)
    out = mdl.predict(numbers, pos, cell, pbc, cov=True)
    X = mdl.X
    species = np.array(mdl.species, np.int32)
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(3, 3, 6.0, species, ind_z, ind_ptr, np.concatenate([x._b for x in X]),
                                       np.concatenate([x._r for x in X]))
    M = orc.kernel_matrix(ind_z, nnm, Pm, ind_z, nnm, Pm, 4.0)
    np.testing.assert_allclose(mdl.M, M, rtol=1e-10, atol=1e-13)
    # the oracle's own (linked-cell) neighbour list, pinned to the brute-force one by the CPU tests; the
    # device list must hold exactly the same pairs at this size too (7 bins per cell edge here)
    nl = orc.neighbors_cells(pos, cell, pbc, 6.0)
    assert pair_set(*mdl.neighbors(len(numbers))) == pair_set(*nl)
    ref = orc.frame(3, 3, 6.0, 4.0, species, numbers, pos, cell, nl, ind_z, nnm, Pm, mdl.mu, choli=mdl.choli)
    np.testing.assert_allclose(out["cov"], ref["cov"], rtol=1e-9, atol=1e-12)
    assert abs(out["energy"] - ref["energy"]) <= 1e-9 * abs(ref["energy"])
    assert np.abs(out["forces"] - ref["forces"]).max() <= 1e-8 * np.abs(ref["forces"]).max()
    assert np.abs(out["stress"] - ref["stress"]).max() <= 1e-8 * np.abs(ref["stress"]).max()
    vs = np.sqrt([mdl._vscale[int(z)] for z in numbers])
    np.testing.assert_allclose(out["beta"], ref["beta"] * vs, rtol=0, atol=2e-6 * vs.max())
    mdl.close()


def test_sharded_lips4096_against_oracle():
    """BASELINE configs[3] sizes on one device: the 4096-atom frame sharded over world = 2, 4, 8 ranks
    (every rank's share evaluated in turn: list build, descriptors and GEMMs of its atoms only, reverse
    pass in scatter form), partial sums added as the all-reduce does (calculator/active.py:562,600-602,
    770-777), against the ORACLE on its own neighbour list — not against the unsharded device result."""
    from oracle import oracle as orc
    mdl, numbers, pos, cell, pbc = _lips_model(16, 512)
    N = len(numbers)
    X = mdl.X
    species = np.array(mdl.species, np.int32)
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(3, 3, 6.0, species, ind_z, ind_ptr, np.concatenate([x._b for x in X]),
                                       np.concatenate([x._r for x in X]))
    nl = orc.neighbors_cells(pos, cell, pbc, 6.0)
    ref = orc.frame(3, 3, 6.0, 4.0, species, numbers, pos, cell, nl, ind_z, nnm, Pm, mdl.mu, choli=mdl.choli)
    vs = np.sqrt([mdl._vscale[int(z)] for z in numbers])
    for world in (2, 4, 8):
        acc = None
        for r in range(world):
            part = mdl.predict(numbers, pos, cell, pbc, rank=r, world=world, cov=True)
            if r == 0:
                assert np.count_nonzero(np.abs(part["cov"]).sum(1)) <= N // world + 3  # only this rank's rows
            acc = {k: np.array(v, dtype=float) for k, v in part.items()} if acc is None else \
                {k: acc[k] + part[k] for k in acc}
        np.testing.assert_allclose(acc["cov"], ref["cov"], rtol=1e-9, atol=1e-12)
        assert abs(acc["energy"] - ref["energy"]) <= 1e-9 * abs(ref["energy"])
        assert np.abs(acc["forces"] - ref["forces"]).max() <= 1e-8 * np.abs(ref["forces"]).max()
        assert np.abs(acc["stress"] - ref["stress"]).max() <= 1e-8 * np.abs(ref["stress"]).max()
        np.testing.assert_allclose(acc["beta"], ref["beta"] * vs, rtol=0, atol=2e-6 * vs.max())
    mdl.close()


def test_sharded_partial_sums_repeat_bit_for_bit():
    """The scatter form of the reverse pass (what a rank of a sharded run executes) hands forces to neighbours through
    atomics: as 64-bit fixed-point integer sums they do not depend on the order the atomics land in, so every rank's
    partial forces repeat bit for bit from run to run — as the single-rank path's do
    (test_full_size_step_is_bit_reproducible) — and the sharded total equals the unsharded one to a few units of the
    fixed-point grid (1.4e-14 eV/A per addition)."""
    mdl, numbers, pos, cell, pbc = _lips_model(16, 512)
    whole = mdl.predict(numbers, pos, cell, pbc)
    for world in (2, 4, 8):
        runs = []
        for _ in range(3):
            parts = [mdl.predict(numbers, pos + 0.0, cell, pbc, rank=r, world=world) for r in range(world)]
            runs.append(parts)
        for r in range(world):
            for k in ("forces", "energy", "stress", "beta"):
                assert np.array_equal(np.asarray(runs[0][r][k]), np.asarray(runs[1][r][k])), (world, r, k)
                assert np.array_equal(np.asarray(runs[0][r][k]), np.asarray(runs[2][r][k])), (world, r, k)
        tot = sum(p["forces"] for p in runs[0])
        assert np.abs(tot - whole["forces"]).max() <= 2e-11 * max(1.0, np.abs(whole["forces"]).max())
    mdl.close()


def test_native_communicator_single_rank():
    """The library's own RCCL path on the one GPU of this box: a communicator of one rank, the step's
    all-reduce enqueued on the step's stream (sgpr_comm_init / sgpr_step_dev), the free-standing
    all-reduce, and the overflow word of the packed buffer."""
    import torch
    from autoforce_amd import _lib
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    lib = _lib.load()
    N = len(g["numbers"])
    ref = mdl.predict(g["numbers"], g["positions"], g["cell"], g["pbc"])
    mdl.comm_init(mdl.comm_unique_id(), 0, 1)
    out = mdl.predict(g["numbers"], g["positions"], g["cell"], g["pbc"])
    for k in ("energy", "forces", "stress", "beta"):
        np.testing.assert_array_equal(np.asarray(out[k]), np.asarray(ref[k]))
    dev = torch.device("cuda", 0)
    buf = torch.arange(5, dtype=torch.float64, device=dev)
    _lib.check(lib.sgpr_comm_allreduce(mdl.handle, buf.data_ptr(), 5, 0, None))
    _lib.check(lib.sgpr_comm_allreduce(mdl.handle, buf.data_ptr(), 5, 1, None))
    _lib.check(lib.sgpr_sync_check(mdl.handle, None))
    np.testing.assert_array_equal(buf.cpu().numpy(), np.arange(5.0))
    pos = torch.from_numpy(g["positions"].copy()).to(dev)
    cell = torch.from_numpy(g["cell"].copy()).to(dev)
    packed = torch.zeros(int(lib.sgpr_packed_len(N)), dtype=torch.float64, device=dev)
    for _ in range(2):
        _lib.check(lib.sgpr_step_dev(mdl.handle, pos.data_ptr(), cell.data_ptr(), packed.data_ptr(), None))
    _lib.check(lib.sgpr_sync_check(mdl.handle, None))
    got = packed.cpu().numpy()
    np.testing.assert_array_equal(got[:3 * N].reshape(N, 3), ref["forces"])
    assert got[4 * N] == ref["energy"] and got[4 * N + 10] == 0.0
    mdl.comm_destroy()
    mdl.close()


def test_full_size_properties():
    """Size-independent properties at 4096 atoms: Newton's third law, translation invariance,
    invariance under a permutation of the caller's atom order, virial = -sum r_i (x) F_i for a
    non-periodic copy (no image terms), and neighbour-list symmetry."""
    mdl, numbers, pos, cell, pbc = _lips_model(16, 512)
    N = len(numbers)
    a = mdl.predict(numbers, pos, cell, pbc)
    fmax = np.abs(a["forces"]).max()
    assert np.abs(a["forces"].sum(0)).max() <= 1e-9 * fmax
    ptr, j, off = mdl.neighbors(N)
    i = np.repeat(np.arange(N), np.diff(ptr))
    fwd = set(zip(i.tolist(), j.tolist(), map(tuple, off.tolist())))
    assert all((jj, ii, (-o[0], -o[1], -o[2])) in fwd for ii, jj, o in list(fwd)[:20000])
    b = mdl.predict(numbers, pos + np.array([1.234, -5.1, 77.7]), cell, pbc)  # atoms far outside the cell
    assert abs(b["energy"] - a["energy"]) <= 1e-9 * abs(a["energy"])
    assert np.abs(b["forces"] - a["forces"]).max() <= 1e-8 * fmax
    perm = np.random.default_rng(3).permutation(N)
    c = mdl.predict(numbers[perm], pos[perm], cell, pbc)
    assert abs(c["energy"] - a["energy"]) <= 1e-10 * abs(a["energy"])
    assert np.abs(c["forces"] - a["forces"][perm]).max() <= 1e-9 * fmax
    np.testing.assert_allclose(c["beta"], a["beta"][perm], rtol=0, atol=1e-7)
    # open boundaries: stress * V == -sum_i r_i (x) F_i exactly (calculator/active.py:604-610 with dcell = 0)
    d = mdl.predict(numbers[:512], pos[:512], cell, [False] * 3)
    vir = -(pos[:512][:, :, None] * d["forces"][:, None, :]).sum(0)
    want = vir.flat[[0, 4, 8, 5, 2, 1]] / abs(np.linalg.det(cell))
    assert np.abs(d["stress"] - want).max() <= 1e-8 * np.abs(want).max()
    mdl.close()


def _workload_model(kind, m, seed=1):
    from autoforce_amd import SGPRModel
    from autoforce_amd import workloads as wl
    make = {"li": wl.li_bcc, "oxide": wl.oxide, "si": wl.si_diamond}[kind]
    numbers, pos, cell, pbc = make(seed=0)
    mdl = SGPRModel(3, 3, 4, 6.0, species=sorted(set(numbers.tolist())))
    n2, p2, c2, b2 = make(seed=seed)
    if m <= len(n2):
        X = wl.inducing_from_frame(mdl, n2, p2, c2, b2, m, seed=seed)
    else:  # more inducing LCEs than atoms in one frame: draw from several rattled copies
        X = []
        for k in range(-(-m // len(n2))):
            nk, pk, ck, bk = make(seed=seed + k)
            X += wl.inducing_from_frame(mdl, nk, pk, ck, bk, min(len(n2), m - len(X)), seed=seed + k)
    mdl.set_inducing(X)
    rng = np.random.default_rng(2)
    mdl.solve(rng.normal(size=(64, m)), rng.normal(size=64))
    mdl.set_weights(rng.normal(size=m), choli=mdl.choli, vscale=mdl.make_vscale())
    return mdl, numbers, pos, cell, pbc


def _oracle_frame(mdl, numbers, pos, cell, nl):
    from oracle import oracle as orc
    X = mdl.X
    species = np.array(mdl.species, np.int32)
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(3, 3, 6.0, species, ind_z, ind_ptr, np.concatenate([x._b for x in X]),
                                       np.concatenate([x._r for x in X]))
    return orc.frame(3, 3, 6.0, 4.0, species, numbers, pos, cell, nl, ind_z, nnm, Pm, mdl.mu, choli=mdl.choli)


def test_baseline_config1_si32_m64():
    """BASELINE configs[0] shape: 32-atom diamond Si, 8-atom cubic x (2,2,1) — the z edge is shorter than
    the cutoff, so every atom meets its own periodic images — with 64 inducing LCEs drawn from
    rattled copies; energy / forces / stress / covloss against the oracle."""
    from oracle import oracle as orc
    mdl, numbers, pos, cell, pbc = _workload_model("si", 64)
    assert len(numbers) == 32 and mdl.m == 64
    out = mdl.predict(numbers, pos, cell, pbc, cov=True)
    nl = orc.neighbors(pos, cell, pbc, 6.0)
    i = np.repeat(np.arange(32), np.diff(nl[0]))
    assert (nl[1] == i).any()  # self images are in the list
    ref = _oracle_frame(mdl, numbers, pos, cell, nl)
    np.testing.assert_allclose(out["cov"], ref["cov"], rtol=1e-9, atol=1e-12)
    assert abs(out["energy"] - ref["energy"]) <= 1e-9 * abs(ref["energy"])
    assert np.abs(out["forces"] - ref["forces"]).max() <= 1e-8 * np.abs(ref["forces"]).max()
    assert np.abs(out["stress"] - ref["stress"]).max() <= 1e-8 * np.abs(ref["stress"]).max()
    np.testing.assert_allclose(out["beta"], ref["beta"] * np.sqrt(mdl._vscale[14]), rtol=0, atol=2e-6)
    mdl.close()


def test_baseline_config2_li256_m128():
    """BASELINE configs[1]: 256-atom bcc Li, 128 inducing points: forces within 1e-6 eV/A (relative) of
    the CPU path; held to 1e-8.  One species: packed row 40 of 64."""
    from oracle import oracle as orc
    mdl, numbers, pos, cell, pbc = _workload_model("li", 128)
    out = mdl.predict(numbers, pos, cell, pbc, cov=True)
    nl = orc.neighbors(pos, cell, pbc, 6.0)
    p, j, off = mdl.neighbors(len(numbers))
    assert p[-1] == nl[0][-1]
    ref = _oracle_frame(mdl, numbers, pos, cell, nl)
    np.testing.assert_allclose(out["cov"], ref["cov"], rtol=1e-9, atol=1e-12)
    assert abs(out["energy"] - ref["energy"]) <= 1e-9 * abs(ref["energy"])
    assert np.abs(out["forces"] - ref["forces"]).max() <= 1e-8 * np.abs(ref["forces"]).max()
    assert np.abs(out["stress"] - ref["stress"]).max() <= 1e-8 * np.abs(ref["stress"]).max()
    mdl.close()


def test_baseline_config5_oxide16384_m1024():
    """BASELINE configs[4] sizes: 16384 atoms, 4 species, 1024 inducing points (single GPU, one
    frame): against the oracle (device neighbour list) and Newton's third law."""
    mdl, numbers, pos, cell, pbc = _workload_model("oxide", 1024)
    N = len(numbers)
    out = mdl.predict(numbers, pos, cell, pbc, cov=False)
    fmax = np.abs(out["forces"]).max()
    assert np.abs(out["forces"].sum(0)).max() <= 1e-8 * fmax
    from oracle import oracle as orc
    nl = orc.neighbors_cells(pos, cell, pbc, 6.0)  # independent of the device list (see the 4096-atom test)
    assert pair_set(*mdl.neighbors(N)) == pair_set(*nl)
    ref = _oracle_frame(mdl, numbers, pos, cell, nl)
    assert abs(out["energy"] - ref["energy"]) <= 1e-9 * abs(ref["energy"])
    assert np.abs(out["forces"] - ref["forces"]).max() <= 1e-8 * np.abs(ref["forces"]).max()
    assert np.abs(out["stress"] - ref["stress"]).max() <= 1e-8 * np.abs(ref["stress"]).max()
    vs = np.sqrt([mdl._vscale[int(z)] for z in numbers])
    np.testing.assert_allclose(out["beta"], ref["beta"] * vs, rtol=0, atol=5e-6 * vs.max())
    mdl.close()


def test_large_cell_beyond_the_bin_grid_cap():
    """110 592 atoms in a 130.6 A cube: the neighbour grid wants 20 bins per edge and gets its cap of 16 (wider bins,
    more candidates per sweep, same pairs).  Pair set and E / F / stress / covloss against the oracle's own
    linked-cell list and reverse pass."""
    from oracle import oracle as orc
    mdl, numbers, pos, cell, pbc = _lips_model(48, 64)
    N = len(numbers)
    assert N == 110592
    out = mdl.predict(numbers, pos, cell, pbc, cov=False)
    nl = orc.neighbors_cells(pos, cell, pbc, 6.0)
    ptr, j, off = mdl.neighbors(N)
    np.testing.assert_array_equal(ptr, nl[0])
    # same pairs: both lists sorted per atom by (j, image)
    def canon(p, jj, oo):
        key = np.lexsort((oo[:, 2], oo[:, 1], oo[:, 0], jj, np.repeat(np.arange(len(p) - 1), np.diff(p))))
        return jj[key], oo[key]
    ja, oa = canon(ptr, np.asarray(j), np.asarray(off))
    jb, ob = canon(nl[0], np.asarray(nl[1]), np.asarray(nl[2]))
    np.testing.assert_array_equal(ja, jb)
    np.testing.assert_array_equal(oa, ob)
    ref = _oracle_frame(mdl, numbers, pos, cell, nl)
    assert abs(out["energy"] - ref["energy"]) <= 1e-9 * abs(ref["energy"])
    assert np.abs(out["forces"] - ref["forces"]).max() <= 1e-8 * np.abs(ref["forces"]).max()
    assert np.abs(out["stress"] - ref["stress"]).max() <= 1e-8 * np.abs(ref["stress"]).max()
    vs = np.sqrt([mdl._vscale[int(z)] for z in numbers])
    np.testing.assert_allclose(out["beta"], ref["beta"] * vs, rtol=0, atol=5e-6 * vs.max())
    mdl.close()


def test_single_atom_lce_from_device():
    """sgpr_get_local == the LCE assembled on the host from the full neighbour list
    (descriptor/atoms.py:365-382), for periodic images and an empty environment."""
    for name in ("g5_si32", "g5_tric24", "g5_cluster16"):
        g = load(name)
        mdl = model_from_fixture(g)
        N = len(g["numbers"])
        mdl.predict(g["numbers"], g["positions"], g["cell"], g["pbc"], beta=False)
        ptr, j, off = mdl.neighbors(N)
        for k in (0, N // 2, N - 1):
            z, r = mdl.local(k)
            a, b = int(ptr[k]), int(ptr[k + 1])
            want_r = g["positions"][j[a:b]] - g["positions"][k] + off[a:b].astype(float) @ g["cell"]
            np.testing.assert_array_equal(z, g["numbers"][j[a:b]])
            np.testing.assert_allclose(r, want_r, rtol=0, atol=1e-13)
        with pytest.raises(Exception):
            mdl.local(N)
        mdl.close()


def test_changing_cell_between_steps_and_wait_modes():
    """One handle stepping through a sequence of cells (constant, strained, back: a barostat) gives, at every step, the bits
    a fresh handle gives for that frame alone: the neighbour grid kept from the previous step is reused only for the
    cell it was made for, and a changed cell rebuilds the candidate lists.  The same sequence with the blocking wait
    (option "spin_wait" = 0) instead of the polled stream gives the same bits again."""
    from autoforce_amd import _lib
    mdl, numbers, pos, cell, pbc = _lips_model(8, 128)
    rng = np.random.default_rng(4)
    strain = np.eye(3) + 0.01 * rng.normal(size=(3, 3))
    frames = []
    p = pos.copy()
    for step in range(7):
        c = cell if step in (0, 1, 2, 5, 6) else cell @ strain
        scaled = np.linalg.solve(cell.T, p.T).T          # fractional coordinates in the reference cell
        frames.append((scaled @ c, c.copy()))
        p = p + 0.01 * rng.normal(size=p.shape)
    seq = [mdl.predict(numbers, x, c, pbc) for x, c in frames]
    mdl.close()
    for k, (x, c) in enumerate(frames):
        fresh, *_ = _lips_model(8, 128)
        ref = fresh.predict(numbers, x, c, pbc)
        fresh.close()
        for key in ("energy", "forces", "stress", "beta"):
            np.testing.assert_array_equal(np.asarray(seq[k][key]), np.asarray(ref[key]), err_msg=f"step {k}: {key}")
    blocking, *_ = _lips_model(8, 128)
    _lib.check(_lib.load().sgpr_set_option(blocking._h, b"spin_wait", 0))
    for k, (x, c) in enumerate(frames):
        out = blocking.predict(numbers, x, c, pbc)
        for key in ("energy", "forces", "stress", "beta"):
            np.testing.assert_array_equal(np.asarray(out[key]), np.asarray(seq[k][key]), err_msg=f"blocking wait, step {k}: {key}")
    blocking.close()
