"""Pins the CPU oracle (oracle/sgpr_oracle.c) to the reference: every function is compared
with vectors captured from the imported reference (tests/golden/gen/make_golden.py) and
with the reference's own numeric KAT (theforce/descriptor/soap.py:488-525)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import oracle as orc

FRAMES = ["g5_si32", "g5_mixed64", "g5_tric24", "g5_cluster16", "g5_slab18_nearz", "g5_si32_l2n2", "g5_big40",
          "g5_bigtric36"]


def load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


# ---------------------------------------------------------------- G1: Ylm (ylm.py:113-225)
@pytest.mark.parametrize("lmax", [2, 3, 4])
@pytest.mark.parametrize("tag", ["plain", "nearz"])
def test_ylm(lmax, tag):
    g = load("g1_ylm")
    xyz = g[f"l{lmax}_{tag}_xyz"]
    Y, dY = orc.ylm(lmax, xyz)
    np.testing.assert_allclose(Y, g[f"l{lmax}_{tag}_Y"], rtol=1e-12, atol=1e-13)
    # the reference's analytic dY goes through an fp32-rounded coefficient table
    # (ylm.py:103-111): agreement is limited to ~1e-7 relative by the reference itself.
    # Near the z axis that error is amplified by 1/sin(theta) ~ 100 (measured 1.8e-6).
    ref = g[f"l{lmax}_{tag}_dY"]
    assert np.abs(dY - ref).max() <= (3e-7 if tag == "plain" else 1e-5) * np.abs(ref).max()
    # independent check of the oracle's gradient: central differences of its own values
    # (the batch keeps its shear state because the near-z members stay in the batch)
    h = 1e-6
    for a in range(3):
        e = np.zeros(3)
        e[a] = h
        fd = (orc.ylm(lmax, xyz + e, grad=False) - orc.ylm(lmax, xyz - e, grad=False)) / (2 * h)
        assert np.abs(fd - dY[..., a]).max() <= 2e-8 * max(1.0, np.abs(dY).max())


# ---------------------------------------------------------------- KAT (soap.py:488-525)
def test_kat_absseries():
    g = load("kat_absseries")
    xyz = g["xyz"]
    nn = len(xyz)
    # AbsSeriesSoap(2,2,PolyCut(3.0)): unit = rc/3 = 1, no gaussian, no nnl, no normalisation
    p = orc.descriptor(2, 2, 3.0, xyz, np.zeros(nn, np.int32), np.ones(nn), 1, flags=0)
    p = p.reshape(3, 3, 3).transpose(2, 0, 1)  # [n,n',l] -> [l,n,n']
    assert np.allclose(p, g["target_lnn"], rtol=1e-5, atol=1e-8)  # the reference's own criterion
    np.testing.assert_allclose(p, g["p_lnn"], rtol=1e-12, atol=1e-14)


# ---------------------------------------------------------------- G2: SeSoap (sesoap.py:161-260)
def _g2_case(g, name):
    lmax, nmax = int(g[name + "_lmax"]), int(g[name + "_nmax"])
    r, z, ab = g[name + "_r"], g[name + "_z"], g[name + "_ab"]
    species = sorted(set(z.tolist()))
    S = len(species)
    slots = np.array([species.index(v) for v in z], np.int32)
    units = orc.default_radii(species)[slots]
    # reference block k sits at COO index (ab[0][k], ab[1][k]); oracle layout p[slot(ab0)][slot(ab1)]
    idx = [(species.index(ab[0, k]), species.index(ab[1, k])) for k in range(ab.shape[1])]
    return lmax, nmax, r, slots, units, S, idx


def test_sesoap_values_and_vjp():
    g = load("g2_sesoap")
    for name in g["names"]:
        lmax, nmax, r, slots, units, S, idx = _g2_case(g, name)
        p = orc.descriptor(lmax, nmax, 6.0, r, slots, units, S)
        praw = orc.descriptor(lmax, nmax, 6.0, r, slots, units, S, flags=3)
        ref = np.zeros_like(p)
        refraw = np.zeros_like(p)
        for k, (i, j) in enumerate(idx):
            ref[i, j] = g[name + "_p"][k]
            refraw[i, j] = g[name + "_p_raw"][k]
        np.testing.assert_allclose(p, ref, rtol=1e-11, atol=1e-14, err_msg=name)
        np.testing.assert_allclose(praw, refraw, rtol=1e-11, atol=1e-300, err_msg=name)
        # reverse pass vs torch.autograd through the reference forward (the force path,
        # calculator/active.py:587-599), same random G
        G = np.zeros_like(p)
        for k, (i, j) in enumerate(idx):
            G[i, j] = g[name + "_G"][k]
        _, dr = orc.descriptor(lmax, nmax, 6.0, r, slots, units, S, G=G)
        want = g[name + "_vjp"]
        assert np.abs(dr - want).max() <= 1e-10 * max(np.abs(want).max(), 1e-300), name
        # ... and vs the reference's ANALYTIC Jacobian dp (sesoap.py:204-246), which carries
        # the fp32-coef taint (SURVEY §7 "two gradient paths") -> 1e-6.  For a single
        # neighbour the analytic path loses the exact angular cancellation (measured 1.6 %
        # off its own autograd), so that case is autograd-only.
        if name != "single":
            dp = g[name + "_dp"]  # [S^2, D, nn, 3]
            want = np.zeros_like(dr)
            for k, (i, j) in enumerate(idx):
                want += np.einsum("d,dna->na", G[i, j], dp[k])
            assert np.abs(dr - want).max() <= 1e-6 * max(np.abs(want).max(), 1e-300), name


def test_subsesoap_fixed_species_table():
    """G3: SubSeSoap (descriptor/sesoap.py:263-391), the dense fixed-table variant behind the
    `species=[...]` kernels (calculator/active.py:31-38), is the same function as the
    species-table layout used here: p[b][a][n][n'][l] over the table, zero blocks for absent
    species.  A neighbour whose species is outside the table contributes nothing in the
    reference; the library's table must list every species it meets (SGPR_E_SPECIES otherwise),
    so that case is reproduced by dropping those neighbours."""
    g2, g3 = load("g2_sesoap"), load("g3_subsesoap")
    for key in g3["names"]:
        name = str(g3[key + "_case"])
        table = g3[key + "_table"].tolist()
        lmax, nmax = int(g2[name + "_lmax"]), int(g2[name + "_nmax"])
        r, z = g2[name + "_r"], g2[name + "_z"]
        keep = np.isin(z, table)
        slots = np.array([table.index(v) for v in z[keep]], np.int32)
        units = orc.default_radii(table)[slots]
        S = len(table)
        p = orc.descriptor(lmax, nmax, 6.0, r[keep], slots, units, S).reshape(S, S, nmax + 1, nmax + 1, lmax + 1)
        # SubSeSoap indexes its blocks [alpha][beta] (sesoap.py:343-352), SeSoap's COO rows are
        # (beta, alpha) (sesoap.py:195-203): the same numbers with the two species axes swapped
        np.testing.assert_allclose(p.transpose(1, 0, 2, 3, 4), g3[key + "_p"], rtol=1e-11, atol=1e-14, err_msg=str(key))


def test_sesoap_symmetry():
    """p[b,a,n,n',l] == p[a,b,n',n,l] (what the packed layout of the HIP path relies on)."""
    g = load("g2_sesoap")
    lmax, nmax, r, slots, units, S, _ = _g2_case(g, "s3")
    p = orc.descriptor(lmax, nmax, 6.0, r, slots, units, S).reshape(S, S, nmax + 1, nmax + 1, lmax + 1)
    np.testing.assert_allclose(p, p.transpose(1, 0, 3, 2, 4), rtol=0, atol=1e-18)


# ---------------------------------------------------------------- neighbour list
@pytest.mark.parametrize("name", FRAMES)
def test_neighbors(name):
    g = load(name)
    ptr, j, off = orc.neighbors(g["positions"], g["cell"], g["pbc"], float(g["rc"]))
    np.testing.assert_array_equal(ptr, g["nl_ptr"])
    np.testing.assert_array_equal(j, g["nl_j"])
    np.testing.assert_array_equal(off, g["nl_off"])


@pytest.mark.parametrize("name", FRAMES)
def test_neighbors_linked_cells(name):
    """The linked-cell builder (used for the 4096- and 16384-atom frames) gives the golden list too."""
    g = load(name)
    ptr, j, off = orc.neighbors_cells(g["positions"], g["cell"], g["pbc"], float(g["rc"]))
    np.testing.assert_array_equal(ptr, g["nl_ptr"])
    np.testing.assert_array_equal(j, g["nl_j"])
    np.testing.assert_array_equal(off, g["nl_off"])


@pytest.mark.parametrize("seed", range(12))
def test_neighbors_linked_cells_random(seed):
    """... and equals the brute-force list on random triclinic cells: every periodicity pattern, cells
    thinner than the cutoff (self images), atoms outside the cell, >= 3 cells per edge."""
    rng = np.random.default_rng(100 + seed)
    n = int(rng.integers(20, 160))
    L = rng.uniform(2.0, 9.0, 3) if seed % 3 else rng.uniform(6.5, 9.5, 3)
    cell = np.diag(L) + rng.uniform(-0.25, 0.25, (3, 3)) * L.min()
    pbc = [bool(seed & 1), bool(seed & 2), bool(seed & 4)] if seed < 8 else [True] * 3
    pos = rng.uniform(-0.7, 1.7, (n, 3)) @ cell
    rc = float(rng.uniform(1.5, 3.2))
    a = orc.neighbors(pos, cell, pbc, rc)
    b = orc.neighbors_cells(pos, cell, pbc, rc)
    for x, y in zip(a, b):
        np.testing.assert_array_equal(x, y)


def test_neighbors_slab_with_zero_open_vector():
    """cell = [a, b, 0], pbc = TTF (valid in ASE, whose neighbour list completes the cell): the periodic
    directions keep their images; equal to the same slab with an explicit out-of-plane vector."""
    rng = np.random.default_rng(7)
    cell0 = np.array([[4.0, 0.3, 0.0], [-0.5, 3.6, 0.0], [0.0, 0.0, 0.0]])
    cell1 = cell0.copy()
    cell1[2] = [0.0, 0.0, 30.0]
    pos = rng.uniform(0, 1, (20, 3)) @ np.array([[4.0, 0.3, 0.0], [-0.5, 3.6, 0.0], [0.0, 0.0, 2.5]])
    pbc = [True, True, False]
    for fn in (orc.neighbors, orc.neighbors_cells):
        a, b = fn(pos, cell0, pbc, 3.0), fn(pos, cell1, pbc, 3.0)
        assert a[0][-1] > 20 * 5  # periodic images are there
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)


# ---------------------------------------------------------------- G4/G5/G6: frames
@pytest.mark.parametrize("name", FRAMES)
def test_frames(name):
    g = load(name)
    lmax, nmax, eta, rc = int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"])
    species = g["species"]
    Pm, nnm = orc.inducing_descriptors(lmax, nmax, rc, species, g["ind_z"], g["ind_ptr"], g["ind_nbr_z"], g["ind_nbr_r"])
    np.testing.assert_allclose(Pm, g["p_ind"], rtol=1e-11, atol=1e-14)
    M = orc.kernel_matrix(g["ind_z"], nnm, Pm, g["ind_z"], nnm, Pm, eta)
    np.testing.assert_allclose(M, g["M"], rtol=1e-11, atol=1e-14)
    L, ridge = orc.jitcholesky(M)
    assert ridge == float(g["ridge"])
    # K_mm of near-identical environments is ill-conditioned (cond ~ 1e8+): hold the
    # factor to backward-stable criteria, and to the reference factor at cond*eps.
    np.testing.assert_allclose(L @ L.T, M, rtol=0, atol=1e-14)
    np.testing.assert_allclose(L, g["L"], rtol=0, atol=1e-9)
    choli = orc.tril_inverse(L)
    np.testing.assert_allclose(choli @ L, np.eye(len(L)), rtol=0, atol=1e-9)
    np.testing.assert_allclose(choli, g["choli"], rtol=0, atol=1e-6 * np.abs(g["choli"]).max())
    out = orc.frame(lmax, nmax, rc, eta, species, g["numbers"], g["positions"], g["cell"],
                    (g["nl_ptr"], g["nl_j"], g["nl_off"]), g["ind_z"], nnm, Pm, g["mu"], choli=g["choli"])
    np.testing.assert_allclose(out["p"], g["p"], rtol=1e-11, atol=1e-14)
    np.testing.assert_allclose(out["cov"], g["cov"], rtol=1e-11, atol=1e-14)
    assert abs(out["energy"] - float(g["energy"])) <= 1e-12 * max(1.0, abs(float(g["energy"])))
    fmax = np.abs(g["forces"]).max()
    # BASELINE north_star: forces within 1e-6 relative; the oracle is held to 1e-9
    assert np.abs(out["forces"] - g["forces"]).max() <= 1e-9 * fmax
    assert np.abs(out["dcell"] - g["dcell"]).max() <= 1e-9 * max(np.abs(g["dcell"]).max(), 1e-12)
    assert np.abs(out["stress"] - g["stress"]).max() <= 1e-9 * max(np.abs(g["stress"]).max(), 1e-12)
    # beta = sqrt(1 - |L^-1 k|^2) amplifies rounding near 0: absolute tolerance
    np.testing.assert_allclose(out["beta"], g["beta"], rtol=0, atol=2e-7)
    vs = orc.vscale(g["M"], g["mu"], g["ind_z"], g["vscale_z"])
    np.testing.assert_allclose(vs, g["vscale"], rtol=1e-12)


# ---------------------------------------------------------------- G7: regression
def test_jitcholesky_ladder():
    g = load("g7_regression")
    L, ridge = orc.jitcholesky(g["chol_pd_M"])
    assert ridge == 0.0
    np.testing.assert_allclose(L, g["chol_pd_L"], rtol=1e-11, atol=1e-13)
    L, ridge = orc.jitcholesky(g["chol_sd_M"])
    # same rung of the ladder (1e-6 * mean(diag) * 2^0); the mean's summation order is torch's
    assert abs(ridge - float(g["chol_sd_ridge"])) <= 1e-14 * ridge
    M = g["chol_sd_M"] + ridge * np.eye(12)
    np.testing.assert_allclose(L @ L.T, M, rtol=0, atol=1e-12 * np.abs(M).max())
    # algebra.py:218-224: all-ones matrix
    L, ridge = orc.jitcholesky(np.ones((30, 30)))
    assert ridge == float(g["chol_ones_ridge"])
    np.testing.assert_allclose(L, g["chol_ones_L"], rtol=1e-6, atol=1e-9)


def test_jitcholesky_failure():
    with pytest.raises(RuntimeError, match="cholesky was not successful"):
        orc.jitcholesky(-np.eye(4))


def test_regression():
    g = load("g7_regression")
    K = np.concatenate([g["reg_Ke"], g["reg_Kf"], g["reg_Kv"]])
    Y = np.concatenate([g["reg_energies"] - g["reg_mean"], g["reg_forces"], g["reg_virial"]])
    out = orc.regression(g["reg_M"], K, Y, noise0=float(g["reg_noise0"]))
    assert out["ridge"] == float(g["reg_ridge"])
    assert abs(out["sigma"] - float(g["reg_sigma"])) <= 1e-15
    np.testing.assert_allclose(out["choli"], g["reg_choli"], rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(out["mu"], g["reg_mu"], rtol=1e-9, atol=1e-11)


# ---------------------------------------------------------------- G9: Distributer
@pytest.mark.parametrize("ws", [1, 2, 4, 8])
def test_distributer(ws):
    g = load("g9_distributer")
    ranks, loads, total = orc.distribute(g["numbers"], ws)
    np.testing.assert_array_equal(ranks, g[f"ranks_{ws}"])
    ranks2, _, _ = orc.distribute(g["numbers"][::-1], ws, loads, total)
    np.testing.assert_array_equal(ranks2, g[f"ranks2_{ws}"])


# ---------------------------------------------------------------- G6: training rows K_e, K_f, K_v
@pytest.mark.parametrize("name", ["g5_big40", "g5_bigtric36", "g5_cluster16"])
def test_kernel_rows(name):
    """The reference builds these rows with its ANALYTIC gradient code (fp32-rounded coefficient
    table, SURVEY §7 'two gradient paths'): agreement is limited to ~1e-6 by the reference.
    Frames are at least 2 rc wide: the reference scatters with `g[j] += f`
    (similarity/universal.py:148), which silently drops terms when an atom appears twice in a
    neighbour list (periodic self images) — a reference bug that is not reproduced."""
    g = load(name)
    rows = load(name.replace("g5_", "g6_rows_"))
    lmax, nmax, eta, rc = int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"])
    Pm, nnm = orc.inducing_descriptors(lmax, nmax, rc, g["species"], g["ind_z"], g["ind_ptr"], g["ind_nbr_z"], g["ind_nbr_r"])
    Ke, Kf, Kv = orc.kernel_rows(lmax, nmax, rc, eta, g["species"], g["numbers"], g["positions"], g["cell"],
                                 (g["nl_ptr"], g["nl_j"], g["nl_off"]), g["ind_z"], nnm, Pm)
    np.testing.assert_allclose(Ke, rows["Ke"], rtol=1e-11, atol=1e-13)
    assert np.abs(Kf - rows["Kf"]).max() <= 2e-6 * np.abs(rows["Kf"]).max()
    assert np.abs(Kv - rows["Kv"]).max() <= 2e-6 * np.abs(rows["Kv"]).max()


def test_g13_fixed_species_kernel_values():
    """The oracle's kernel restatement against the reference's `species=[...]` kernel LIST (one SubSeSoapKernel per
    species, summed: calculator/active.py:31-38, regression/gppotential.py:63-84): neighbours outside the table are
    dropped before the descriptor (descriptor/sesoap.py:343-346) and the lone-atom term counts once per kernel object
    (similarity/similarity.py:94-103)."""
    g = load("g13_subsesoap_kernel")
    table = g["table"].astype(np.int32)
    ptr = g["ptr"]
    keep = np.isin(g["nbr_z"], table)
    counts = np.array([keep[ptr[k]:ptr[k + 1]].sum() for k in range(len(g["zc"]))])
    ptr2 = np.concatenate([[0], np.cumsum(counts)])
    Pm, nnm = orc.inducing_descriptors(int(g["lmax"]), int(g["nmax"]), float(g["rc"]), table, g["zc"], ptr2, g["nbr_z"][keep],
                                       g["nbr_r"][keep])
    K = orc.kernel_matrix(g["zc"], nnm, Pm, g["zc"], nnm, Pm, float(g["eta"]))
    lone = nnm == 0
    K[np.ix_(lone, lone)] *= len(table)
    np.testing.assert_allclose(K, g["K"], rtol=1e-11, atol=1e-13)


def test_decisions_of_the_fixtures_sit_far_from_their_thresholds():
    """What the device's tolerances can and cannot flip.  The sampling rules compare numbers with thresholds:
    add_1inducing accepts when the energy change de reaches ediff (gppotential.py:955-982), update_lce compares the
    covloss beta with ediff / ediff_lb / ediff_ub (active.py:806-839).  The device holds de to ~1e-9 relative and beta
    to 5e-6 absolute (beta = sqrt(1 - |L^-1 k|^2) amplifies rounding near zero; beta^2 is held to 1e-9), so a decision
    can only differ from the reference's when the quantity lies within that distance of its threshold.  In the
    fixtures none does:
      * g11: every finite de / ediff of the reference's own decisions is at least 5 % away from 1;
      * the covloss of every atom of every golden frame is at least 1e-4 eV (20 x the device's absolute tolerance on
        beta) away from the default ediff = 2 kcal/mol = 0.0867 eV (active.py:79,118)."""
    g = load("g11_acceptance")
    ratios = [de / t1 for kind, idx, added, de, df, m, nd, ridge, t1, t2 in g["events"] if kind == 0 and np.isfinite(de)]
    assert len(ratios) >= 12
    assert min(abs(r - 1.0) for r in ratios) > 0.05, ratios
    for kind, idx, added, de, df, m, nd, ridge, t1, t2 in g["events"]:
        if kind == 0 and np.isfinite(de):
            assert bool(added) == (de >= t1)      # the rule itself, on the reference's numbers
    ediff = 2 * 0.04336410390059322
    near = []
    for name in ("g5_big40", "g5_bigtric36", "g5_cluster16", "g5_mixed64", "g5_si32", "g5_si32_l2n2", "g5_slab18_nearz", "g5_tric24"):
        f = load(name)
        near.append(float(np.abs(f["covloss"] - ediff).min()))
    assert min(near) > 1e-4, near
