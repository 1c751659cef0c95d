"""GPU: the resident training set (sgpr_data_*): the design matrix [K_e; K_f; K_v] of the stored frames
kept in device memory and edited in step with the data (add_data / pop_1data / popfirst_1data,
regression/gppotential.py:730-743, :793-813) and with the inducing set (add_inducing :745-772, pop /
select :782-842, :1037-1046).  After ANY sequence of edits it must equal what sgpr_kernel_rows returns
for every stored frame (itself pinned to the reference's rows and the oracle's in test_hip_rows.py), and
its solve must be the solve of the same matrix handed over from the host."""
import numpy as np
import pytest

import active_common as ac
from test_hip_parity import load, model_from_fixture

pytestmark = pytest.mark.gpu


def systems():
    """Three H/O/Zr frames of different sizes: two golden ones and a rattled, strained copy."""
    a, b = load("g5_mixed64"), load("g5_bigtric36")
    rng = np.random.default_rng(11)
    c = (b["numbers"], b["positions"] @ (np.eye(3) + 0.02 * rng.normal(size=(3, 3))) + 0.05 * rng.normal(size=b["positions"].shape),
         b["cell"], b["pbc"])
    c = (c[0], c[1], c[2] @ (np.eye(3) + 0.02 * rng.normal(size=(3, 3))), c[3])
    return [(a["numbers"], a["positions"], a["cell"], a["pbc"]), (b["numbers"], b["positions"], b["cell"], b["pbc"]), c]


def expected(mdl, frames, nvs):
    """[rows, m] in the store's order (frame-major: e, 3N f, nv v) from sgpr_kernel_rows."""
    out = []
    for fr, nv in zip(frames, nvs):
        ke, kf, kv = mdl.kernel_rows(*fr)
        out += [ke[None], kf, kv[:nv]]
    return np.concatenate(out) if out else np.zeros((0, mdl.m))


def test_store_follows_every_edit():
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    X = list(mdl.X)
    frames = systems()
    nvs = [6, 0, 6]
    mdl.set_inducing(X[:6])
    for fr, nv in zip(frames[:2], nvs[:2]):
        mdl.data_push(*fr, nv)
    assert mdl.data_info() == (2, sum(1 + 3 * len(f[0]) + nv for f, nv in zip(frames[:2], nvs[:2])))
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames[:2], nvs[:2]))
    # one new column per stored frame, computed by the library
    for x in X[6:9]:
        mdl.add_inducing(x)
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames[:2], nvs[:2]))
    # a frame pushed later gets all columns
    mdl.data_push(*frames[2], nvs[2])
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames, nvs))
    # pops of the inducing set: last (free), first (re-index), arbitrary selection
    mdl.remove_inducing(-1)
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames, nvs))
    mdl.remove_inducing(0)
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames, nvs))
    mdl.select_inducing([4, 0, 2, 5])
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames, nvs))
    mdl.add_inducing(X[9])
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames, nvs))
    # pops of the data: first (rows move up), last
    mdl.data_pop(0)
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames[1:], nvs[1:]))
    mdl.data_pop(-1)
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames[1:2], nvs[1:2]))
    # a whole new inducing set: everything is recomputed
    mdl.set_inducing(X[2:12])
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, frames[1:2], nvs[1:2]))
    # products
    v = np.random.default_rng(0).normal(size=mdl.m)
    K = mdl.data_get()
    np.testing.assert_allclose(mdl.data_matvec(v), K @ v, rtol=0, atol=1e-12 * np.abs(K).max() * np.abs(v).sum())
    # errors: bad index, a species outside the table; the store is untouched by a failed push
    from autoforce_amd import SgprError
    with pytest.raises(SgprError):
        mdl.data_pop(3)
    bad = (np.full(4, 99), np.random.default_rng(1).uniform(0, 4, (4, 3)), np.eye(3) * 8.0, [True] * 3)
    with pytest.raises(SgprError):
        mdl.data_push(*bad, 6)
    assert mdl.data_info()[0] == 1
    np.testing.assert_array_equal(mdl.data_get(), K)
    mdl.data_clear()
    assert mdl.data_info() == (0, 0)
    mdl.close()


def test_frames_stored_before_the_first_inducing_lce():
    """add_1atoms_fast with an empty X stores the frame without rows; the first add_inducing gives them
    (gppotential.py:757-764, the numel() == 0 branch)."""
    g = load("g5_big40")
    full = model_from_fixture(g)
    X = list(full.X)
    fr = (g["numbers"], g["positions"], g["cell"], g["pbc"])
    mdl = full.scratch()
    mdl.set_inducing([])
    mdl.data_push(*fr, 6)
    assert mdl.data_info() == (1, 1 + 3 * len(fr[0]) + 6)
    mdl.add_inducing(X[0])
    mdl.add_inducing(X[1])
    np.testing.assert_array_equal(mdl.data_get(), expected(mdl, [fr], [6]))
    mdl.close(); full.close()


@pytest.mark.parametrize("with_energies", [True, False])
def test_resident_solve_is_the_host_solve(with_energies):
    mdl = model_from_fixture(load("g5_mixed64"))
    frames = systems()[:2]
    nvs = [6, 6]
    for fr, nv in zip(frames, nvs):
        mdl.data_push(*fr, nv)
    K = mdl.data_get()
    rng = np.random.default_rng(3)
    Y = K @ rng.normal(size=mdl.m) + 1e-3 * rng.normal(size=len(K))
    is_e = np.zeros(len(K), bool)
    a = 0
    for fr, nv in zip(frames, nvs):
        is_e[a] = True
        a += 1 + 3 * len(fr[0]) + nv
    ref = mdl.scratch()
    ref.set_inducing(mdl.X)
    keep = np.ones(len(K), bool) if with_energies else ~is_e
    mu_h = ref.solve(K[keep], Y[keep], noise=0.02)
    mu_d = mdl.data_solve(Y, with_energies=with_energies, noise=0.02)
    assert mdl.ridge == ref.ridge and abs(mdl.sigma - ref.sigma) <= 1e-15 * ref.sigma
    np.testing.assert_allclose(mdl.choli, ref.choli, rtol=0, atol=1e-12 * np.abs(ref.choli).max())
    pred_h, pred_d = K @ mu_h, K @ mu_d
    assert np.abs(pred_h - pred_d).max() <= 1e-9 * np.abs(pred_h).max()
    # the cached first stage serves other noise values
    np.testing.assert_allclose(K @ mdl.resolve(noise=0.05), K @ ref.resolve(noise=0.05), rtol=0, atol=1e-9 * np.abs(pred_h).max())
    mdl.close(); ref.close()


@pytest.mark.parametrize("resident", [True, False])
def test_reference_edit_sequence_in_both_modes(resident):
    """g8: the reference's own fitted states after a sequence of data / inducing edits, with the design matrix on
    the device (default) and as host arrays (the mode every non-HIP engine uses)."""
    from autoforce_amd import SGPRModel
    g = load("g5_big40")
    eng = SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]), species=g["species"].tolist())
    eng.resident_data = resident
    ac.check_g8_edit_sequence(eng)


def test_acceptance_rules_host_mode():
    from autoforce_amd import SGPRModel
    g = load("g5_big40")
    eng = SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]), species=g["species"].tolist())
    eng.resident_data = False
    ac.check_g11_acceptance(eng)


def test_refit_after_a_rejected_trial():
    """add_1inducing / add_1atoms_fast (gppotential.py:898-982) try an edit, refit, and on rejection pop it and refit
    again: the second refit must reproduce the fit before the trial exactly (the library may serve its first
    stage from the factors it kept), and different targets must not."""
    mdl = model_from_fixture(load("g5_mixed64"))
    X = list(mdl.X)
    frames = systems()
    mdl.set_inducing(X[:20])
    for fr in frames[:2]:
        mdl.data_push(*fr, 6)
    rng = np.random.default_rng(9)
    Y = rng.normal(size=mdl.data_info()[1])
    mu0 = mdl.data_solve(Y, noise=0.02).copy()
    # inducing trial
    mdl.add_inducing(X[20])
    mu1 = mdl.data_solve(Y, noise=0.02).copy()
    assert len(mu1) == 21
    mdl.remove_inducing(-1)
    np.testing.assert_array_equal(mdl.data_solve(Y, noise=0.02), mu0)
    # data trial
    mdl.data_push(*frames[2], 6)
    Y3 = np.concatenate([Y, rng.normal(size=mdl.data_info()[1] - len(Y))])
    mu3 = mdl.data_solve(Y3, noise=0.02).copy()
    mdl.data_pop(-1)
    np.testing.assert_array_equal(mdl.data_solve(Y, noise=0.02), mu0)
    # other targets, other noise, force-only fit: each its own answer, equal to a model that never cached anything
    ref = mdl.scratch()
    ref.set_inducing(mdl.X)
    K = mdl.data_get()
    Y2 = Y.copy(); Y2[5] += 1.0
    for args, kw in (((Y2,), dict(noise=0.02)), ((Y,), dict(noise=0.05)), ((Y,), dict(noise=0.02, with_energies=False))):
        got = mdl.data_solve(*args, **kw)
        keep = np.ones(len(K), bool)
        if not kw.get("with_energies", True):
            a = 0
            for fr in frames[:2]:
                keep[a] = False
                a += 1 + 3 * len(fr[0]) + 6
        want = ref.solve(K[keep], args[0][keep], noise=kw["noise"])
        assert np.abs(K @ got - K @ want).max() <= 1e-9 * np.abs(K @ want).max()
    assert np.abs(mu3).max() > 0 and np.abs(mu1).max() > 0
    mdl.close(); ref.close()


def test_columns_through_the_kept_reflectors():
    """A long sequence of appended and popped inducing LCEs: every refit goes through the kept first-stage
    factorisation (new column = Q^T k + its own one-column panel, popped column dropped) and must agree with a model
    that factors the same matrix from scratch; targets change in between (Q^T Y is rebuilt)."""
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    X = list(mdl.X)
    frames = systems()
    mdl.set_inducing(X[:6])
    for fr in frames:
        mdl.data_push(*fr, 6)
    rows = mdl.data_info()[1]
    rng = np.random.default_rng(4)
    Y = rng.normal(size=rows)

    def check(Yv, noise=0.02, with_energies=True):
        got = mdl.data_solve(Yv, noise=noise, with_energies=with_energies).copy()
        ref = mdl.scratch()
        ref.set_inducing(mdl.X)
        K = mdl.data_get()
        keep = np.ones(len(K), bool)
        if not with_energies:
            a = 0
            for fr in frames[:mdl.data_info()[0]]:
                keep[a] = False
                a += 1 + 3 * len(fr[0]) + 6
        want = ref.solve(K[keep], Yv[keep], noise=noise)
        assert ref.ridge == mdl.ridge
        scale = np.abs(K @ want).max()
        assert np.abs(K @ got - K @ want).max() <= 1e-9 * scale, (len(mdl.X), np.abs(K @ got - K @ want).max() / scale)
        ref.close()
        return got

    check(Y)                                   # full factorisation
    for x in X[6:12]:                          # six appended columns
        mdl.add_inducing(x)
        check(Y)
    check(Y, with_energies=False)              # the force-only fit keeps its own factorisation ...
    mdl.remove_inducing(-1); check(Y)          # popped: an appended column
    check(Y, with_energies=False)              # ... and follows the columns too
    Y2 = Y + 0.1 * rng.normal(size=rows)
    check(Y2)                                  # new targets through all kept panels
    mdl.add_inducing(X[12]); check(Y2)
    for _ in range(8):                         # pops reaching into the columns of the full factorisation
        mdl.remove_inducing(-1)
    mu_a = check(Y2)
    mdl.add_inducing(X[13]); check(Y2)         # append below the original column count
    mdl.remove_inducing(-1)
    np.testing.assert_array_equal(check(Y2), mu_a)
    # a data trial: the other slot, and back
    mdl.data_pop(-1)
    n3 = 1 + 3 * len(frames[2][0]) + 6
    check(Y2[:-n3])
    mdl.add_inducing(X[14]); check(Y2[:-n3])
    mdl.data_push(*frames[2], 6)
    check(Y2)                                  # rows of the pushed frame appended to the kept factor
    check(Y2, with_energies=False)
    Y3 = Y2.copy(); Y3[0] += 0.5; Y3[199] -= 0.25   # the energy targets of the old frames move with the mean
    mdl.data_pop(-1); check(Y3[:-n3])
    mdl.data_push(*frames[2], 6); check(Y3)
    # more appended columns than the library keeps as one-column panels: it refactors on its own
    mdl.set_inducing(X[:4])
    check(Y)
    for k in range(40):
        mdl.add_inducing(X[4 + k % 20].__class__(X[4 + k % 20].number, X[4 + k % 20]._b, X[4 + k % 20]._r + 1e-3 * (k + 1)))
        if k % 7 == 0 or k > 36:
            check(Y)
    mdl.close()


def test_weights_stay_on_the_device_after_a_solve():
    """make_munu's tail (gppotential.py:548-605): after a device solve, commit_weights (mean only) must leave the
    evaluator in the state set_weights(mu, choli) builds from the downloaded arrays."""
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    fr = (g["numbers"], g["positions"], g["cell"], g["pbc"])
    mdl.data_push(*fr, 6)
    rng = np.random.default_rng(2)
    Y = rng.normal(size=mdl.data_info()[1])
    mu = mdl.data_solve(Y, noise=0.03)
    mean = {int(z): float(w) for z, w in zip(mdl.species, rng.normal(size=len(mdl.species)))}
    mdl.commit_weights(mean=mean)
    a = mdl.predict(*fr)
    choli = mdl.choli                       # downloaded on demand
    assert choli.shape == (mdl.m, mdl.m) and np.allclose(choli @ mdl.M @ choli.T, np.eye(mdl.m), atol=1e-6)
    ref = mdl.scratch()
    ref.set_inducing(mdl.X)
    ref.set_weights(mu, mean=mean, vscale=mdl._vscale, choli=choli)
    b = ref.predict(*fr)
    for k in ("energy", "forces", "stress", "beta"):
        np.testing.assert_array_equal(np.asarray(a[k]), np.asarray(b[k]))
    mdl.close(); ref.close()


def test_kept_and_from_scratch_factorisations_agree():
    """Option "qr_keep": the same edit sequence with the kept factorisation (default) and with a factorisation from
    scratch at every refit (what the reference does, gppotential.py:745-791)."""
    from autoforce_amd import _lib
    g = load("g5_mixed64")
    X = list(model_from_fixture(g).X)
    frames = systems()
    rng = np.random.default_rng(8)
    Y = None
    res = {}
    for mode in (1, 0):
        mdl = model_from_fixture(g)
        _lib.check(_lib.load().sgpr_set_option(mdl.handle, b"qr_keep", mode))
        mdl.set_inducing(X[:8])
        for fr in frames[:2]:
            mdl.data_push(*fr, 6)
        if Y is None:
            Y = rng.normal(size=mdl.data_info()[1] + 1 + 3 * len(frames[2][0]) + 6)
        out = []
        n2 = mdl.data_info()[1]
        out.append(mdl.data_solve(Y[:n2], noise=0.02).copy())
        for x in X[8:12]:
            mdl.add_inducing(x)
            out.append(mdl.data_solve(Y[:n2], noise=0.02).copy())
        mdl.remove_inducing(-1)
        out.append(mdl.data_solve(Y[:n2], noise=0.02).copy())
        out.append(mdl.data_solve(Y[:n2], noise=0.02, with_energies=False).copy())
        mdl.data_push(*frames[2], 6)
        out.append(mdl.data_solve(Y, noise=0.02).copy())
        K = mdl.data_get()
        res[mode] = (out, K)
        mdl.close()
    K = res[1][1]
    np.testing.assert_array_equal(K, res[0][1])
    for a, b in zip(res[1][0], res[0][0]):
        Ka = K[:, :len(a)] if len(a) == K.shape[1] else None
        assert np.abs(a - b).max() <= 1e-7 * np.abs(b).max()


def test_batched_resolve_equals_resolve():
    """sgpr_resolve_batch: a dozen noise values in one set of launches give the weights of a dozen sgpr_resolve calls
    (same arithmetic per problem: bit for bit), and leave the installed weights alone."""
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    for fr in systems()[:2]:
        mdl.data_push(*fr, 6)
    rng = np.random.default_rng(5)
    Y = rng.normal(size=mdl.data_info()[1])
    mu0 = mdl.data_solve(Y, noise=0.02).copy()
    before = mdl.predict(*systems()[0])
    noises = [1e-4, 3e-3, 0.02, 0.1, 0.3, 0.7, 0.02]
    many = mdl.resolve_many(noises)
    after = mdl.predict(*systems()[0])
    np.testing.assert_array_equal(before["forces"], after["forces"])
    np.testing.assert_array_equal(many[2], mu0)
    np.testing.assert_array_equal(many[6], mu0)
    for k, nz in enumerate(noises):
        np.testing.assert_array_equal(many[k], mdl.resolve(noise=nz))
    mdl.close()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_random_edit_sequences(seed):
    """Randomised differential test of the resident training set and the kept factorisation: a random walk over the
    edits the learning loop makes (append / pop / select inducing LCEs, push / pop frames, new targets, force-only
    fits, re-solves), the fit checked after every step against a model that is set up and factored from scratch."""
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    pool = list(mdl.X)
    frames_pool = systems()
    rng = np.random.default_rng(100 + seed)
    X = [pool[i] for i in rng.permutation(len(pool))[:5]]
    mdl.set_inducing(X)
    frames, targets = [], []

    def push():
        fr = frames_pool[int(rng.integers(len(frames_pool)))]
        # a rattled copy: every stored frame different
        fr = (fr[0], fr[1] + 0.03 * rng.normal(size=fr[1].shape), fr[2], fr[3])
        nv = int(rng.choice([0, 6]))
        mdl.data_push(*fr, nv)
        frames.append((fr, nv))
        targets.append(rng.normal(size=1 + 3 * len(fr[0]) + nv))

    push(); push()
    fresh_id = [0]

    def new_lce():
        base = pool[int(rng.integers(len(pool)))]
        fresh_id[0] += 1
        return base.__class__(base.number, base._b, base._r + 0.02 * rng.normal(size=base._r.shape))

    def check(with_energies=True, noise=0.02):
        Y = np.concatenate(targets)
        got = mdl.data_solve(Y, noise=noise, with_energies=with_energies).copy()
        ref = mdl.scratch()
        ref.set_inducing(mdl.X)
        K = np.concatenate([np.concatenate([ke[None], kf, kv[:nv]]) for (fr, nv) in frames
                            for ke, kf, kv in [ref.kernel_rows(*fr)]])
        np.testing.assert_array_equal(mdl.data_get(), K)
        keep = np.ones(len(K), bool)
        if not with_energies:
            a = 0
            for fr, nv in frames:
                keep[a] = False
                a += 1 + 3 * len(fr[0]) + nv
        want = ref.solve(K[keep], Y[keep], noise=noise)
        scale = max(np.abs(K @ want).max(), 1e-300)
        assert np.abs(K @ got - K @ want).max() <= 1e-8 * scale
        ref.close()

    check()
    for step in range(36):
        op = rng.choice(["add", "add", "add", "pop", "pop", "popfirst", "select", "push", "popdata", "popfirstdata",
                         "targets", "force_only", "resolve"])
        if op == "add":
            mdl.add_inducing(new_lce())
        elif op == "pop" and mdl.m > 3:
            mdl.remove_inducing(-1)
        elif op == "popfirst" and mdl.m > 3:
            mdl.remove_inducing(0)
        elif op == "select" and mdl.m > 4:
            idx = rng.permutation(mdl.m)[:mdl.m - 1].tolist()
            mdl.select_inducing(idx)
        elif op == "push" and len(frames) < 4:
            push()
        elif op == "popdata" and len(frames) > 1:
            mdl.data_pop(-1); frames.pop(); targets.pop()
        elif op == "popfirstdata" and len(frames) > 1:
            mdl.data_pop(0); frames.pop(0); targets.pop(0)
        elif op == "targets":
            k = int(rng.integers(len(targets)))
            targets[k] = targets[k] + 0.1 * rng.normal(size=len(targets[k]))
        elif op == "force_only":
            check(with_energies=False)
        elif op == "resolve":
            check()
            Y = np.concatenate(targets)
            many = mdl.resolve_many([0.01, 0.05])
            np.testing.assert_array_equal(many[0], mdl.resolve(noise=0.01))
            np.testing.assert_array_equal(many[1], mdl.resolve(noise=0.05))
            continue
        check()
    mdl.close()


def test_fit_statistics_on_the_device():
    """sgpr_data_fit_stats against numpy over the downloaded matrix (make_stats, gppotential.py:610-649)."""
    mdl = model_from_fixture(load("g5_mixed64"))
    frames = systems()
    nvs = [6, 0, 6]
    for fr, nv in zip(frames, nvs):
        mdl.data_push(*fr, nv)
    rng = np.random.default_rng(12)
    K = mdl.data_get()
    Y = rng.normal(size=len(K))
    v = rng.normal(size=mdl.m)
    e_pred, st = mdl.data_fit_stats(v, Y)
    pred = K @ v
    is_e = np.zeros(len(K), bool)
    a = 0
    for fr, nv in zip(frames, nvs):
        is_e[a] = True
        a += 1 + 3 * len(fr[0]) + nv
    np.testing.assert_allclose(e_pred, pred[is_e], rtol=0, atol=1e-12 * np.abs(pred).max())
    d, y = (pred - Y)[~is_e], Y[~is_e]
    want = [d.sum(), np.abs(d).sum(), (d * d).sum(), y.sum(), (y * y).sum(), np.abs(y).max(), len(d)]
    np.testing.assert_allclose(st, want, rtol=1e-12, atol=1e-10)
    np.testing.assert_array_equal(mdl.M_diag, np.diag(mdl.M))
    np.testing.assert_array_equal(mdl.M_rowsum, mdl.M.sum(axis=1))   # (numpy's summation order, on the device)
    # the objective of the noise search: force rows only (energy and virial rows out), a batch of weight vectors
    is_f = np.zeros(len(K), bool)
    a = 0
    for fr, nv in zip(frames, nvs):
        is_f[a + 1:a + 1 + 3 * len(fr[0])] = True
        a += 1 + 3 * len(fr[0]) + nv
    V = rng.normal(size=(19, mdl.m))  # more than one pass of sixteen
    want = [np.abs((K @ w - Y)[is_f]).mean() for w in V]
    np.testing.assert_allclose(mdl.data_force_mae(V, Y), want, rtol=1e-12)
    np.testing.assert_allclose(mdl.data_force_mae(V[:1], Y), want[:1], rtol=1e-12)
    mdl.close()


def test_downsize_follows_incrementally():
    """downsize(lii=True) / popfirst / removal at an index (gppotential.py:815-842, :1037-1046; the reference refits
    from scratch after each).  The library re-indexes what survives: K_mm bit for bit, the K_mm factor per species
    block (only the blocks whose order changed are factored again), the kept first-stage QR through the reflectors
    of R1[:, idx] — and the fit after every edit equals the fit of a model set up and factored from scratch."""
    g = load("g5_mixed64")
    mdl = model_from_fixture(g)
    pool = list(mdl.X)
    rng = np.random.default_rng(31)
    # a larger pool: rattled copies, so that every species block has a dozen LCEs
    X = [x.__class__(x.number, x._b, x._r + 0.03 * rng.normal(size=x._r.shape)) for x in pool for _ in range(2)]
    X = X[:44]
    frames = systems()
    mdl.set_inducing(X)
    for fr in frames:
        mdl.data_push(*fr, 6)
    K_rows = mdl.data_info()[1]
    Y = rng.normal(size=K_rows)

    def check(route=None, blocks=None):
        got = mdl.data_solve(Y, noise=0.02).copy()
        info = mdl.solve_info()
        if route is not None:
            assert route in info, info
        if blocks is not None:
            assert f"kmm_blocks={blocks}/" in info, info
        ref = mdl.scratch()
        ref.set_inducing(mdl.X)
        np.testing.assert_array_equal(mdl.M, ref.M)
        K = np.concatenate([np.concatenate([ke[None], kf, kv]) for fr in frames for ke, kf, kv in [ref.kernel_rows(*fr)]])
        np.testing.assert_array_equal(mdl.data_get(), K)
        want = ref.solve(K, Y, noise=0.02)
        assert ref.ridge == mdl.ridge == 0.0
        np.testing.assert_allclose(mdl.choli, ref.choli, rtol=0, atol=1e-10 * np.abs(ref.choli).max())
        scale = np.abs(K @ want).max()
        assert np.abs(K @ got - K @ want).max() <= 1e-8 * scale, np.abs(K @ got - K @ want).max() / scale
        ref.close()
        return got

    check(route="full factorisation")
    nblocks = len(set(x.number for x in mdl.X))
    # 1. an LCE in the middle of the list goes: one block is factored again, the QR follows through the selection
    victim = 7
    mdl.remove_inducing(victim)
    assert len(mdl.X) == 43
    check(route="columns selected through the kept reflectors", blocks=1)
    # 2. the first of the list (popfirst_1inducing)
    mdl.remove_inducing(0)
    check(route="columns selected through the kept reflectors", blocks=1)
    # 3. downsize(lii=True): the 36 LCEs with the smallest K_mm row sums, IN ARGSORT ORDER (a permutation)
    order = np.argsort(mdl.M.sum(axis=1), kind="stable").tolist()[:36]
    mdl.select_inducing(order)
    check(route="columns selected through the kept reflectors")
    # 4. trials on top of the selection: append, refit, pop, refit (the pop restores the pre-trial fit bit for bit)
    mu0 = check()
    mdl.add_inducing(X[3].__class__(X[3].number, X[3]._b, X[3]._r + 0.01))
    check(route="columns appended / popped through the kept reflectors", blocks=0)
    mdl.remove_inducing(-1)
    np.testing.assert_array_equal(check(route="columns appended / popped through the kept reflectors", blocks=0), mu0)
    # 5. a trailing LCE of a block goes: that block keeps its factor (a leading part of L is the factor of the
    #    leading part)
    last_of_block = max(i for i, x in enumerate(mdl.X) if x.number == mdl.X[-1].number)
    mdl.remove_inducing(last_of_block)
    check(blocks=0)
    # 6. several selections in a row, then appended columns again, against the from-scratch path of the library itself
    for _ in range(3):
        keep = sorted(rng.permutation(len(mdl.X))[:len(mdl.X) - 2].tolist())
        mdl.select_inducing(keep)
        check(route="columns selected through the kept reflectors")
        mdl.add_inducing(X[40].__class__(X[40].number, X[40]._b, X[40]._r + 0.02 * rng.normal(size=X[40]._r.shape)))
        check(route="columns appended / popped through the kept reflectors")
    assert nblocks >= 2
    mdl.close()


_LOOKAHEAD_SCRIPT = r"""
import hashlib, sys
import numpy as np
from autoforce_amd import SGPRModel
from autoforce_amd.workloads import inducing_from_frame, lips
mdl = SGPRModel(3, 3, 4, 6.0, species=[3, 15, 16])
mdl.set_inducing(inducing_from_frame(mdl, *lips(16, seed=0), 1024, seed=1))
rng = np.random.default_rng(0)
# a tall dense problem of the size of a 16384-atom frame's rows (no reflectors kept: the two-stream schedule); 1024
# columns: the 256-row chunk count of the panels drops four times on the way
K = rng.normal(size=(50183, 1024)); Y = rng.normal(size=50183)
out = []
for rep in range(4):
    mu = mdl.solve(K, Y)
    out.append(hashlib.sha256(np.ascontiguousarray(mu).tobytes()).hexdigest())
assert len(set(out)) == 1, out
print("MU", out[0])
"""


@pytest.mark.gpu
def test_tall_factorisation_with_lookahead_is_the_one_stream_one():
    """A tall factorisation that keeps no reflectors — the rows of a pushed frame appended to a kept factor — runs the far
    trailing update on a second stream beside the next panel's leaves (tsqr.hip): the weights are those of the one-stream
    schedule bit for bit, call after call.  (The scratch the two streams share is laid out once per call: a layout that
    followed the panels let a stacked R land on reflectors still being read, and an on-the-fly run took another
    trajectory now and then.)"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    got = {}
    for la in ("1", "0", "1"):
        env = dict(os.environ, SGPR_QR_LOOKAHEAD=la, PYTHONPATH=root)
        r = subprocess.run([sys.executable, "-c", _LOOKAHEAD_SCRIPT], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got.setdefault(la, []).append([ln for ln in r.stdout.splitlines() if ln.startswith("MU")][-1])
    assert len(set(got["1"])) == 1 and got["1"][0] == got["0"][0], got
