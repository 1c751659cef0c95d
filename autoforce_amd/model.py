"""Host-side mirror of the reference's model objects for the SGPR predict/solve hot path.

  Local      <- theforce/descriptor/atoms.py:36-55   (one local chemical environment, LCE)
  SGPRModel  <- theforce/regression/gppotential.py:453-1175 PosteriorPotential state
               (X, M, mu, choli, ridge, _vscale, mean) + the default kernel of
               theforce/calculator/active.py:28-38 (SeSoapKernel(lmax,nmax,exponent,cutoff,
               radii=DefaultRadii())).
All numerics run in libsgpr_hip.so (HIP, gfx950) through the C ABI; this file only marshals.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import check, f64, i32, i64, ptr


class Local:
    """An LCE: central atomic number, neighbour numbers `b` and displacement vectors `r`
    (= x_j - x_i + off.cell), as theforce/descriptor/atoms.py:36-55 (`number`, `_b`, `_r`)."""

    def __init__(self, number, b, r):
        self.number = int(number)
        self._b = i32(b).reshape(-1)
        self._r = f64(r).reshape(-1, 3)
        if len(self._b) != len(self._r):
            raise ValueError("Local: len(b) != len(r)")

    def __repr__(self):
        return f"Local(Z={self.number}, nn={len(self._b)})"


def default_radii(species):
    """DefaultRadii (theforce/descriptor/sesoap.py:84-99): 0.5 for H, else 1.0."""
    return np.array([0.5 if int(z) == 1 else 1.0 for z in species])


class SGPRModel:
    def __init__(self, lmax=3, nmax=3, exponent=4, cutoff=6.0, species=None, radii=None, device=0,
                 unknown_species="error", lone_weight=1):
        """unknown_species: "error" (default) or "ignore" — atoms and LCE neighbours whose atomic number is
        not in `species` are invisible, as in the reference's fixed-species kernels
        (descriptor/sesoap.py:343-346, similarity/heterosoap.py:37-71).
        lone_weight: k(x, x') of two lone atoms (no neighbour inside the cutoff) of one species.  The reference adds
        that term once per kernel OBJECT (similarity/similarity.py:38-40, :94-103) and sums the kernels
        (regression/gppotential.py:63-84): 1 for the wildcard SeSoapKernel, len(species) for the list of fixed-species
        kernels that `kernel_kw={'species': [...]}` builds (calculator/active.py:31-38)."""
        if species is None or len(species) == 0:
            raise ValueError("SGPRModel needs the species table (atomic numbers the model may meet)")
        self.lmax, self.nmax, self.exponent, self.cutoff = int(lmax), int(nmax), float(exponent), float(cutoff)
        self.species = [int(z) for z in species]
        self.radii = default_radii(self.species) if radii is None else f64(radii)
        self.device = int(device)
        self._h = C.c_void_p()
        lib = _lib.load()
        check(lib.sgpr_create(self.lmax, self.nmax, self.exponent, self.cutoff, len(self.species),
                              ptr(i32(self.species)), ptr(f64(self.radii)), self.device, C.byref(self._h)))
        self.unknown_species = unknown_species
        if unknown_species == "ignore":
            check(lib.sgpr_set_option(self._h, b"ignore_unknown_species", 1))
        elif unknown_species != "error":
            raise ValueError("unknown_species must be 'error' or 'ignore'")
        self.lone_weight = int(lone_weight)
        if self.lone_weight != 1:
            check(lib.sgpr_set_option(self._h, b"lone_atom_weight", self.lone_weight))
        self.X = []
        self.mu = None
        self._choli, self._choli_on_device = None, False
        self.ridge = 0.0
        self.sigma = None
        self.mean = {z: 0.0 for z in self.species}  # AutoMean weights (gppotential.py:200-231)
        self._vscale = {}
        self.generation = 0   # counts the frames the device evaluated (predict, training rows): whoever caches
                              # something about "the last frame" (calc.cov) can tell when it moved on
        self._pv = None       # predict_view's per-system cache
        self.comm_world = 1   # > 1 once an RCCL communicator is attached (comm_init)
        self.peer_world = 1   # > 1 once the library's own exchange is attached (peer_attach)

    # ------------------------------------------------------------------ lifetime
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _lib.load().sgpr_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def solve_info(self):
        """Route of the last data_solve / data_factor (sgpr_solve_info): 'stage1=...; kmm_blocks=a/b'."""
        buf = C.create_string_buffer(256)
        check(_lib.load().sgpr_solve_info(self._h, buf, 256))
        return buf.value.decode()

    def list_rebuilds(self):
        """How many steps of this handle rebuilt the Verlet candidate lists so far (the others filtered the kept
        candidates: descriptor.hip)."""
        n = C.c_int64(0)
        check(_lib.load().sgpr_get_list_rebuilds(self._h, C.addressof(n)))
        return int(n.value)

    # ------------------------------------------------------------------ multi-GPU (one process per GPU)
    @staticmethod
    def comm_unique_id():
        """ncclUniqueId bytes (rank 0 creates them; the host passes them to the other ranks)."""
        buf = C.create_string_buffer(128)
        check(_lib.load().sgpr_comm_unique_id(C.addressof(buf)))
        return buf.raw

    def comm_init(self, uid, rank, world):
        """Attach an RCCL communicator (collective over all ranks): from now on a sharded predict() /
        sgpr_step_dev ends with ONE all-reduce of the packed buffer on the step's stream and returns
        totals (the reference's four MPI collectives, calculator/active.py:562,601,602,777)."""
        buf = C.create_string_buffer(bytes(uid), 128)
        check(_lib.load().sgpr_comm_init(self._h, C.addressof(buf), int(rank), int(world)))
        self.comm_world = int(world)

    def comm_destroy(self):
        check(_lib.load().sgpr_comm_destroy(self._h))
        self.comm_world = 1

    PEER_HANDLE_BYTES = 128

    def peer_export(self, rank, world, capacity):
        """This rank's receive buffers of the library's own exchange (hipIpc all-gather + local sum in rank order,
        include/sgpr_hip.h): `capacity` doubles per source rank (>= 7 N + 11 for frames of N atoms).  Returns the bytes
        every peer needs for peer_attach."""
        buf = C.create_string_buffer(self.PEER_HANDLE_BYTES)
        check(_lib.load().sgpr_peer_export(self._h, int(rank), int(world), int(capacity), C.addressof(buf)))
        return buf.raw

    def peer_attach(self, handles):
        """handles: the peer_export bytes of ALL ranks in rank order.  From now on sharded predict() / step / md_run
        combine the ranks' partial sums through this exchange: the same bits on every rank, independent of the number of
        ranks (the reference's collectives: calculator/active.py:562,600-602,770-777)."""
        blob = b"".join(bytes(x) for x in handles)
        buf = C.create_string_buffer(blob, len(blob))
        check(_lib.load().sgpr_peer_attach(self._h, C.addressof(buf)))
        self.peer_world = self.comm_world = len(handles)   # (comm_world: "the library combines the ranks itself")

    def peer_selftest(self, rank, world):
        """One small exchange with a known answer (every rank contributes rank + 1 in eight doubles): True when the sum came
        back on this rank.  The hosts call it right after peer_attach and agree on the outcome before they rely on the
        exchange (a rank whose stores are not seen by a peer would otherwise surface as a time-out in the first step)."""
        import torch
        buf = torch.full((8,), float(rank + 1), dtype=torch.float64, device=f"cuda:{self.device}")
        lib = _lib.load()
        try:
            check(lib.sgpr_comm_allreduce(self._h, buf.data_ptr(), 8, 0, None))
            check(lib.sgpr_sync_check(self._h, None))
        except _lib.SgprError:
            return False
        return bool((buf.cpu().numpy() == world * (world + 1) / 2).all())

    def peer_destroy(self):
        check(_lib.load().sgpr_peer_destroy(self._h))
        self.peer_world = self.comm_world = 1

    def scratch(self):
        """A second, empty model with the same kernel on the same device (used for one-off
        K(atoms, atoms) evaluations that must not disturb this model's inducing set)."""
        return SGPRModel(self.lmax, self.nmax, self.exponent, self.cutoff, species=self.species, radii=self.radii,
                         device=self.device, unknown_species=self.unknown_species, lone_weight=self.lone_weight)

    def with_species(self, species):
        """The same model over another species table (same kernel, inducing LCEs, weights): how the
        wildcard kernel of the reference (calculator/active.py:28-38, a 120-wide sparse table) is
        served by a dense table — it is re-laid-out when a new species turns up.  Descriptor blocks
        of absent species are zero, so every kernel value, mu and choli stay what they were."""
        # species already in the table keep their length unit (a model built with custom radii must not fall back
        # to the defaults on the first re-layout: the kernel would change under mu and choli); new ones get the
        # default (descriptor/sesoap.py:84-99)
        have = {int(z): float(r) for z, r in zip(self.species, self.radii)}
        radii = [have.get(int(z), float(default_radii([z])[0])) for z in species]
        new = SGPRModel(self.lmax, self.nmax, self.exponent, self.cutoff, species=species, radii=radii,
                        device=self.device, unknown_species=self.unknown_species, lone_weight=self.lone_weight)
        new.mean.update(self.mean)
        new._vscale = dict(self._vscale)
        if self.X:
            new.set_inducing(self.X)
            if self.mu is not None:
                new.set_weights(self.mu, mean=self.mean, vscale=self._vscale or None, choli=self.choli)
        new.ridge, new.sigma = self.ridge, self.sigma
        return new

    # ------------------------------------------------------------------ inducing set
    def set_inducing(self, X):
        """model.X = inducing LCEs; builds descriptors and K_mm on the device
        (gppotential.py:484-509 set_data: self.M = kern(X, X))."""
        X = list(X)
        m = len(X)
        zc = i32([x.number for x in X])
        nptr = i64(np.concatenate([[0], np.cumsum([len(x._b) for x in X])]))
        nz = i32(np.concatenate([x._b for x in X] + [np.zeros(0, np.int32)]))
        nr = f64(np.concatenate([x._r for x in X] + [np.zeros((0, 3))]))
        self.generation += 1
        check(_lib.load().sgpr_set_inducing(self._h, m, ptr(zc), ptr(nptr), ptr(nz), ptr(nr)))
        self.X = X  # (after the device accepted it: on an error host and device lists still agree)
        self.mu = None
        self.choli = None

    @property
    def choli(self):
        """L^-1 of the K_mm factor, [m, m].  A device solve leaves it on the device; it is downloaded when somebody
        looks (leakage, model files), not at every refit."""
        if self._choli is None and self._choli_on_device and self.m:
            out = np.zeros((self.m, self.m))
            check(_lib.load().sgpr_get_choli(self._h, ptr(out)))
            self._choli = out
        return self._choli

    @choli.setter
    def choli(self, value):
        self._choli, self._choli_on_device = value, False

    def _weights_dropped(self):
        self.mu = None
        self.choli = None

    def add_inducing(self, loc):
        """Append one LCE (PosteriorPotential.add_inducing, gppotential.py:745-772, without the
        data columns — those are `kernel_columns`)."""
        self.generation += 1  # (stored data frames are re-bound for their new column)
        check(_lib.load().sgpr_add_inducing(self._h, loc.number, len(loc._b), ptr(loc._b), ptr(loc._r)))
        self.X.append(loc)
        self._weights_dropped()

    def remove_inducing(self, index=-1):
        """pop_1inducing / popfirst_1inducing (gppotential.py:782-813)."""
        check(_lib.load().sgpr_remove_inducing(self._h, int(index)))
        del self.X[index]
        self._weights_dropped()

    def select_inducing(self, indices):
        """Keep `indices` in the given order (gppotential.py:1037-1046)."""
        idx = i32(indices)
        check(_lib.load().sgpr_select_inducing(self._h, len(idx), ptr(idx)))
        self.X = [self.X[int(j)] for j in idx]
        self._weights_dropped()

    def kernel_local(self, loc):
        """(k(loc, X)[m], k(loc, loc)) for an LCE outside the inducing set (active.py:806-818)."""
        k = np.zeros(self.m)
        kxx = C.c_double(0)
        check(_lib.load().sgpr_kernel_local(self._h, loc.number, len(loc._b), ptr(loc._b), ptr(loc._r), ptr(k),
                                            C.addressof(kxx)))
        return k, kxx.value

    @property
    def m(self):
        return len(self.X)

    @property
    def M(self):
        out = np.zeros((self.m, self.m))
        check(_lib.load().sgpr_get_kmm(self._h, ptr(out)))
        return out

    @property
    def M_diag(self):
        out = np.zeros(self.m)
        check(_lib.load().sgpr_get_kmm_diag(self._h, ptr(out)))
        return out

    @property
    def M_rowsum(self):
        """K_mm.sum(axis=1), caller order, summed on the device in numpy's own order (bit for bit `self.M.sum(axis=1)`)."""
        out = np.zeros(self.m)
        check(_lib.load().sgpr_get_kmm_rowsum(self._h, ptr(out)))
        return out

    @property
    def dims(self):
        out = np.zeros(8, np.int32)
        check(_lib.load().sgpr_get_dims(self._h, ptr(out)))
        return dict(m=int(out[0]), S=int(out[1]), D=int(out[2]), Dc=int(out[3]), maxnn=int(out[4]), N=int(out[5]),
                    nn_max=int(out[6]), Dpad=int(out[7]))

    def inducing_descriptors(self):
        S, D = len(self.species), (self.nmax + 1) ** 2 * (self.lmax + 1)
        out = np.zeros((self.m, S, S, D))
        check(_lib.load().sgpr_get_inducing_descriptors(self._h, ptr(out)))
        return out

    # ------------------------------------------------------------------ weights
    def _table(self, d, default):
        return f64([d.get(z, default) if d is not None else default for z in self.species])

    def set_weights(self, mu, mean=None, vscale=None, choli=None):
        """Install mu / AutoMean weights / _vscale / choli (gppotential.py:548-605,644-649)."""
        self.mu = f64(mu).copy()
        if mean is not None:
            self.mean.update({int(z): float(w) for z, w in mean.items()})
        if vscale is not None:
            self._vscale = {int(z): float(v) for z, v in vscale.items()}
        self.choli = None if choli is None else f64(choli).copy()
        vs = f64([self._vscale.get(z, np.inf) for z in self.species]) if self._vscale else None
        check(_lib.load().sgpr_set_weights(self._h, ptr(self.mu), ptr(self._table(self.mean, 0.0)), ptr(vs),
                                           ptr(self._choli)))

    def commit_weights(self, mean=None):
        """The tail of make_munu after a device solve: mu and choli are already installed on the device, only the
        AutoMean weights follow (_vscale is what make_vscale just computed)."""
        if mean is not None:
            self.mean.update({int(z): float(w) for z, w in mean.items()})
        check(_lib.load().sgpr_set_mean(self._h, ptr(self._table(self.mean, 0.0)), None))

    def snapshot_weights(self):
        """What a rejected trial has to put back (see restore_weights)."""
        return dict(mu=None if self.mu is None else self.mu.copy(), ridge=self.ridge, sigma=self.sigma,
                    vscale=dict(self._vscale), mean=dict(self.mean))

    def restore_weights(self, snap):
        """After the pop that ends a rejected trial: the weights saved before it, instead of a refit that would
        reproduce them (gppotential.py:898-982)."""
        check(_lib.load().sgpr_restore_weights(self._h, ptr(f64(snap["mu"]))))
        self.mu = snap["mu"].copy()
        self.ridge, self.sigma = snap["ridge"], snap["sigma"]
        self._vscale = dict(snap["vscale"])
        self.mean.update(snap["mean"])
        self._choli, self._choli_on_device = None, True
        vs = f64([self._vscale.get(z, np.inf) for z in self.species])
        check(_lib.load().sgpr_set_mean(self._h, ptr(self._table(self.mean, 0.0)), ptr(vs)))

    def solve(self, K, Y, noise=0.01):
        """make_munu (gppotential.py:548-605 -> _regression :1204-1339, optimize=False) on the
        device: jitcholesky(M), choli = L^-1, mu = lstsq([K; sigma L^T], [Y; 0])."""
        K = f64(K).reshape(-1, self.m)
        Y = f64(Y).reshape(-1)
        mu = np.zeros(self.m)
        ridge, sigma = C.c_double(0), C.c_double(0)
        code = _lib.load().sgpr_solve(self._h, len(K), ptr(K), ptr(Y), float(noise), ptr(mu), None,
                                      C.addressof(ridge), C.addressof(sigma))
        if code == _lib.E_NOT_PD:
            raise RuntimeError("cholesky was not successful!")  # theforce/regression/algebra.py:45-46
        check(code)
        self.mu, self.ridge, self.sigma = mu, ridge.value, sigma.value
        self._choli, self._choli_on_device = None, True
        self.make_vscale()
        return mu

    def jitcholesky(self, A):
        """regression/algebra.py:29-47 on the device for any symmetric matrix: (L, ridge)."""
        A = f64(A)
        n = len(A)
        L = np.zeros((n, n))
        ridge = C.c_double(0)
        code = _lib.load().sgpr_jitcholesky(self._h, n, ptr(A), ptr(L), C.addressof(ridge))
        if code == _lib.E_NOT_PD:
            raise RuntimeError("cholesky was not successful!")
        check(code)
        return L, ridge.value

    def resolve(self, noise=0.01):
        """The same regression for another noise, re-using the factored [K | Y] of the last `solve`
        (the evaluations of _regression(optimize=True), gppotential.py:1265-1300)."""
        mu = np.zeros(self.m)
        ridge, sigma = C.c_double(0), C.c_double(0)
        # (choli = L^-1 and the ridge do not depend on the noise: they stay what the last solve returned)
        check(_lib.load().sgpr_resolve(self._h, float(noise), ptr(mu), None, C.addressof(ridge), C.addressof(sigma)))
        self.mu, self.ridge, self.sigma = mu, ridge.value, sigma.value
        return mu

    def resolve_many(self, noises):
        """mu for each of `noises` from the factored [K | Y] of the last solve, evaluated together on the device
        (the grid scan of the noise search); the installed weights do not change.  Returns [len(noises), m]."""
        noises = f64(noises).reshape(-1)
        out = np.zeros((len(noises), self.m))
        for a in range(0, len(noises), 64):
            part = np.zeros((min(64, len(noises) - a), self.m))
            check(_lib.load().sgpr_resolve_batch(self._h, len(part), ptr(f64(noises[a:a + 64])), ptr(part)))
            out[a:a + len(part)] = part
        return out

    def kernel_rows(self, numbers, positions, cell, pbc):
        """(Ke[m], Kf[3N,m], Kv[6,m]) of one data frame (gppotential.py:63-84, :495-497)."""
        numbers = i32(numbers)
        N = len(numbers)
        positions = f64(positions).reshape(N, 3)
        cell = f64(np.asarray(cell, float).reshape(3, 3))
        pbc = i32(np.asarray(pbc, bool).astype(np.int32))
        Ke, Kf, Kv = np.zeros(self.m), np.zeros((3 * N, self.m)), np.zeros((6, self.m))
        self.generation += 1
        check(_lib.load().sgpr_kernel_rows(self._h, N, ptr(numbers), ptr(positions), ptr(cell), ptr(pbc), ptr(Ke),
                                           ptr(Kf), ptr(Kv)))
        return Ke, Kf, Kv

    def kernel_columns(self, numbers, positions, cell, pbc, q_first, q_count):
        """The same rows restricted to inducing columns [q_first, q_first+q_count): the bordering
        step of add_inducing (gppotential.py:745-763)."""
        numbers = i32(numbers)
        N = len(numbers)
        positions = f64(positions).reshape(N, 3)
        cell = f64(np.asarray(cell, float).reshape(3, 3))
        pbc = i32(np.asarray(pbc, bool).astype(np.int32))
        Ke, Kf, Kv = np.zeros(q_count), np.zeros((3 * N, q_count)), np.zeros((6, q_count))
        self.generation += 1
        check(_lib.load().sgpr_kernel_columns(self._h, N, ptr(numbers), ptr(positions), ptr(cell), ptr(pbc),
                                              int(q_first), int(q_count), ptr(Ke), ptr(Kf), ptr(Kv)))
        return Ke, Kf, Kv

    # ------------------------------------------------------------------ resident training set
    def data_push(self, numbers, positions, cell, pbc, nv=6):
        """Append a data frame to the device-resident design matrix (PosteriorPotential.add_data,
        gppotential.py:730-743): its K_e / K_f / K_v rows are computed on the device and stay there."""
        numbers = i32(numbers)
        N = len(numbers)
        positions = f64(positions).reshape(N, 3)
        cell = f64(np.asarray(cell, float).reshape(3, 3))
        pbc = i32(np.asarray(pbc, bool).astype(np.int32))
        self.generation += 1
        check(_lib.load().sgpr_data_push(self._h, N, ptr(numbers), ptr(positions), ptr(cell), ptr(pbc), int(nv)))

    def data_pop(self, index=-1):
        """pop_1data (index -1) / popfirst_1data (index 0), gppotential.py:793-813."""
        check(_lib.load().sgpr_data_pop(self._h, int(index)))

    def data_clear(self):
        check(_lib.load().sgpr_data_clear(self._h))

    def data_info(self):
        n, rows = C.c_int32(0), C.c_int64(0)
        check(_lib.load().sgpr_data_info(self._h, C.addressof(n), C.addressof(rows)))
        return n.value, rows.value

    def data_matvec(self, v):
        """K v over all stored rows (frame-major: e, 3N f, nv v per frame)."""
        v = f64(v).reshape(self.m)
        out = np.zeros(self.data_info()[1])
        check(_lib.load().sgpr_data_matvec(self._h, ptr(v), ptr(out)))
        return out

    def data_fit_stats(self, v, Y):
        """(e_pred[frames], stats[7]) of K v against the targets Y, reduced on the device: the energy rows of K v and
        {sum d, sum |d|, sum d^2, sum y, sum y^2, max |y|, count} of d = K v - Y over the force / virial rows."""
        v = f64(v).reshape(self.m)
        Y = f64(Y).reshape(-1)
        n, rows = self.data_info()
        if len(Y) != rows:
            raise ValueError(f"data_fit_stats: {len(Y)} targets for {rows} stored rows")
        e, st = np.zeros(n), np.zeros(8)
        check(_lib.load().sgpr_data_fit_stats(self._h, ptr(v), ptr(Y), ptr(e), ptr(st)))
        return e, st[:7]

    def data_force_mae(self, V, Y):
        """mean |K_f v - Y_f| over the force rows for every row v of V [count, m] (sgpr_data_force_mae): the objective of
        the noise search, reduced on the device."""
        V = f64(V).reshape(-1, self.m)
        Y = f64(Y).reshape(-1)
        if len(Y) != self.data_info()[1]:
            raise ValueError(f"data_force_mae: {len(Y)} targets for {self.data_info()[1]} stored rows")
        out = np.zeros(len(V))
        check(_lib.load().sgpr_data_force_mae(self._h, len(V), ptr(V), ptr(Y), ptr(out)))
        return out

    def data_get(self):
        """The resident design matrix [rows, m] (diagnostics / tests)."""
        out = np.zeros((self.data_info()[1], self.m))
        if out.size:
            check(_lib.load().sgpr_data_get(self._h, ptr(out)))
        return out

    def data_factor(self, Y, with_energies=True):
        """The first stage of data_solve alone: [R1, z] of the resident [K | Y] for resolve / resolve_many."""
        Y = f64(Y).reshape(-1)
        if len(Y) != self.data_info()[1]:
            raise ValueError(f"data_factor: {len(Y)} targets for {self.data_info()[1]} stored rows")
        self.generation += 1
        code = _lib.load().sgpr_data_factor(self._h, ptr(Y), int(bool(with_energies)))
        if code == _lib.E_NOT_PD:
            raise RuntimeError("cholesky was not successful!")
        check(code)

    def data_solve(self, Y, with_energies=True, noise=0.01):
        """`solve` on the resident matrix (Y in its row order)."""
        Y = f64(Y).reshape(-1)
        if len(Y) != self.data_info()[1]:
            raise ValueError(f"data_solve: {len(Y)} targets for {self.data_info()[1]} stored rows")
        mu = np.zeros(self.m)
        ridge, sigma = C.c_double(0), C.c_double(0)
        self.generation += 1
        code = _lib.load().sgpr_data_solve(self._h, ptr(Y), int(bool(with_energies)), float(noise), ptr(mu), None,
                                           C.addressof(ridge), C.addressof(sigma))
        if code == _lib.E_NOT_PD:
            raise RuntimeError("cholesky was not successful!")  # theforce/regression/algebra.py:45-46
        check(code)
        self.mu, self.ridge, self.sigma = mu, ridge.value, sigma.value
        self._choli, self._choli_on_device = None, True
        self.make_vscale()
        return mu

    def fit(self, frames, noise=0.01):
        """set_data + make_munu (gppotential.py:484-509, :548-605) for a list of labelled frames
        dict(numbers, positions, cell, pbc, energy, forces[, stress]): builds K = [Ke; Kf; Kv] and
        Y = [E - mean; F; V * stress] and solves for mu on the device.  Frames without `stress`
        contribute no virial rows."""
        Ke, Kf, Kv, Ye, Yf, Yv = [], [], [], [], [], []
        for fr in frames:
            ke, kf, kv = self.kernel_rows(fr["numbers"], fr["positions"], fr["cell"], fr["pbc"])
            Ke.append(ke[None]); Kf.append(kf)
            mean = sum(self.mean.get(int(z), 0.0) for z in fr["numbers"])
            Ye.append([fr["energy"] - mean]); Yf.append(np.asarray(fr["forces"], float).reshape(-1))
            if fr.get("stress") is not None:
                vol = abs(np.linalg.det(np.asarray(fr["cell"], float).reshape(3, 3)))
                Kv.append(kv); Yv.append(np.asarray(fr["stress"], float) * vol)
        K = np.concatenate(Ke + Kf + Kv)
        Y = np.concatenate([np.concatenate(Ye)] + Yf + Yv)
        return self.solve(K, Y, noise=noise)

    def make_vscale(self):
        out = np.zeros(len(self.species))
        check(_lib.load().sgpr_make_vscale(self._h, ptr(out)))
        self._vscale = {z: float(v) for z, v in zip(self.species, out) if np.isfinite(v)}
        return self._vscale

    # ------------------------------------------------------------------ prediction
    def predict(self, numbers, positions, cell, pbc, rank=0, world=1, cov=False, beta=True):
        """One pass of the hot path (calculator/active.py:425-502): returns a dict with energy,
        forces [N,3], stress [6], and optionally beta [N] (covloss) and cov [N,m]."""
        numbers = i32(numbers)
        N = len(numbers)
        positions = f64(positions).reshape(N, 3)
        cell = f64(np.asarray(cell, float).reshape(3, 3))
        pbc = i32(np.asarray(pbc, bool).astype(np.int32))
        E = C.c_double(0)
        F = np.empty((N, 3))  # every output is written by the call (a failure raises)
        stress = np.zeros(6)
        b = np.empty(N) if beta else None
        K = np.zeros((N, self.m)) if cov else None
        self.generation += 1
        check(_lib.load().sgpr_compute(self._h, N, ptr(numbers), ptr(positions), ptr(cell), ptr(pbc), rank, world,
                                       C.addressof(E), ptr(F), ptr(stress), ptr(b), ptr(K)))
        return dict(energy=E.value, forces=F, stress=stress, beta=b, cov=K)

    def predict_view(self, numbers, positions, cell, pbc, rank=0, world=1):
        """predict() for callers in a loop (ActiveCalculator.calculate): the same pass, but forces / beta / stress come
        back as numpy VIEWS of the page-locked buffer the device wrote them to (include/sgpr_hip.h: sgpr_compute_view) — valid
        until the call after next —, and what does not change between calls (the int32 numbers, pbc, the views themselves)
        is kept instead of being rebuilt: ~20 us less per call at 4096 atoms."""
        N = len(numbers)
        if N == 0:   # an empty frame: zeros, as predict() gives (sgpr_compute_view has no buffer to hand out)
            return self.predict(numbers, positions, cell, pbc, rank=rank, world=world)
        c = self._pv
        if c is None or c["N"] != N or c["src"] is not numbers:
            n32 = i32(numbers)
            if c is not None and c["N"] == N and np.array_equal(c["n32"], n32):
                c["src"] = numbers
            else:
                c = self._pv = dict(N=N, src=numbers, n32=n32, n32p=ptr(n32), views={}, out=C.c_void_p(0))
                c["outp"] = C.addressof(c["out"])
        if positions.dtype != np.float64 or not positions.flags.c_contiguous:
            positions = f64(positions)
        if cell.dtype != np.float64 or not cell.flags.c_contiguous:
            cell = f64(cell)
        pb = (bool(pbc[0]), bool(pbc[1]), bool(pbc[2]))
        if c.get("pb") != pb:
            c["pb"], c["pbc32"] = pb, i32(np.asarray(pb, np.int32))
            c["pbcp"] = ptr(c["pbc32"])
        self.generation += 1
        code = _lib.load().sgpr_compute_view(self._h, N, c["n32p"], positions.ctypes.data, cell.ctypes.data, c["pbcp"], rank, world,
                                             c["outp"])
        if code:
            check(code)
        addr = c["out"].value
        v = c["views"].get(addr)
        if v is None:
            buf = np.frombuffer((C.c_double * (4 * N + 17)).from_address(addr), dtype=np.float64)
            v = c["views"][addr] = (buf[:3 * N].reshape(N, 3), buf[3 * N:4 * N], buf[4 * N:4 * N + 1].reshape(()), buf[4 * N + 11:4 * N + 17])
        return dict(energy=v[2], forces=v[0], stress=v[3], beta=v[1], cov=None)

    # ------------------------------------------------------------------ device-resident molecular dynamics
    MD_SCALARS = 16  # per evaluation: E, virial[9], overflow word, largest covloss, sum m v^2, 3 spare

    def md_begin(self, numbers, positions, cell, pbc, masses, velocities=None, dt=1.0, friction=0.0, kT=0.0, seed=0, ttime=None):
        """State of an MD run into device memory (cl/md.py:117-128 drives ase.md.langevin around calculate();
        here the integrator is part of the step's last kernel).  dt, friction and kT in the caller's units
        (workloads.FS / ase_shim.kB for fs / K)."""
        numbers = i32(numbers)
        N = len(numbers)
        self._md = dict(N=N, numbers=numbers, cell=f64(np.asarray(cell, float).reshape(3, 3)), masses=f64(masses), hdt=0.5 * dt)
        v = None if velocities is None else f64(velocities).reshape(N, 3)
        self.generation += 1
        check(_lib.load().sgpr_md_begin(self._h, N, ptr(numbers), ptr(f64(positions).reshape(N, 3)), ptr(self._md["cell"]),
                                        ptr(i32(np.asarray(pbc, bool).astype(np.int32))), ptr(self._md["masses"]), ptr(v),
                                        float(dt), float(friction), float(kT)))
        # seed != 0: md_run(noise=None) draws the Langevin deviates on the device (counter-based, md_deviates returns them)
        check(_lib.load().sgpr_md_seed(self._h, int(seed) & 0xFFFFFFFFFFFFFFFF))
        # ttime: Nose-Hoover NVT with that time constant instead of the Langevin / velocity-Verlet step (the reference's
        # default dynamics, cl/md.py:131-166: ase.md.npt.NPT with pfactor = None)
        if ttime is not None:
            check(_lib.load().sgpr_md_thermostat(self._h, 1, float(ttime), float(kT)))
            self._md["nh"] = True
        self._md["t"] = 0

    def md_deviates(self, t_first, count):
        out = np.empty((int(count), self._md["N"], 3))
        check(_lib.load().sgpr_md_deviates(self._h, int(t_first), int(count), ptr(out)))
        return out

    def md_run(self, nevals, noise=None, ediff=0.0, final=False):
        """Evaluate `nevals` configurations starting with the current one, integrating between them on the device
        (noise: [nevals, N, 3] standard normal deviates or None).  Returns (scalars [done, 16], halt code): code 1 =
        the last row's largest covloss reached ediff and the state is that configuration (calculator/active.py:492-499),
        2 = a neighbour capacity overflowed at evaluation `done` (repeat the call)."""
        N = self._md["N"]
        if noise is not None:
            noise = f64(noise).reshape(-1, N, 3)
            assert len(noise) >= nevals, "one row of noise per evaluation"
        sc = np.zeros((nevals, self.MD_SCALARS))
        done, code = C.c_int(0), C.c_int(0)
        self.generation += 1
        check(_lib.load().sgpr_md_run(self._h, int(nevals), ptr(noise), float(ediff), int(bool(final)), ptr(sc),
                                      C.addressof(done), C.addressof(code)))
        return sc[:done.value], code.value

    def md_state(self, which=0, results=False):
        """Positions, velocities of the current configuration (which = -1: the one before it); with results=True also
        the forces / covloss / energy / stress of its last evaluation, and the velocities include the closing half kick
        of that evaluation (what an observer of the trajectory sees, workloads.langevin_nvt)."""
        N = self._md["N"]
        x, v = np.empty((N, 3)), np.empty((N, 3))
        pend = C.c_int(0)
        packed = np.empty(4 * N + 11) if results else None
        check(_lib.load().sgpr_md_state(self._h, ptr(x), ptr(v), C.addressof(pend), ptr(packed), int(which)))
        out = dict(positions=x, velocities_pre=v, pending=bool(pend.value))
        if results:
            F = packed[:3 * N].reshape(N, 3).copy()
            stress = np.zeros(6)
            check(_lib.load().sgpr_stress_from_virial(ptr(f64(packed[4 * N + 1:4 * N + 10])), ptr(self._md["cell"]), ptr(stress)))
            out.update(forces=F, beta=packed[3 * N:4 * N].copy(), energy=float(packed[4 * N]), stress=stress)
            if self._md.get("nh"):   # Nose-Hoover: the centred velocity of this configuration (v is the one before it)
                vn = np.empty((N, 3))
                check(_lib.load().sgpr_md_velocities(self._h, ptr(vn)))
                out["velocities"] = vn
            else:
                out["velocities"] = v + self._md["hdt"] * F / self._md["masses"][:, None] if pend.value else v.copy()
        return out

    def md_end(self):
        check(_lib.load().sgpr_md_end(self._h))

    def descriptors(self, N):
        S, D = len(self.species), (self.nmax + 1) ** 2 * (self.lmax + 1)
        out = np.zeros((N, S, S, D))
        check(_lib.load().sgpr_get_descriptors(self._h, ptr(out)))
        return out

    def neighbors(self, N):
        p = np.zeros(N + 1, np.int64)
        check(_lib.load().sgpr_get_neighbors(self._h, ptr(p), None, None))
        j = np.zeros(int(p[-1]), np.int32)
        off = np.zeros((int(p[-1]), 3), np.int32)
        check(_lib.load().sgpr_get_neighbors(self._h, ptr(p), ptr(j), ptr(off)))
        return p, j, off

    def last_cov(self, N):
        """K_nm [N, m] of the last evaluated frame, downloaded on demand."""
        out = np.zeros((N, self.m))
        if self.m:
            check(_lib.load().sgpr_get_cov(self._h, int(N), int(self.m), ptr(out)))
        return out

    def local(self, atom):
        """The LCE of one atom of the last evaluated frame (TorchAtoms.local, descriptor/atoms.py:365-382)
        straight from the device neighbour list."""
        nn = C.c_int32(0)
        lib = _lib.load()
        check(lib.sgpr_get_local(self._h, int(atom), C.addressof(nn), None, None, 0))
        z, r = np.zeros(nn.value, np.int32), np.zeros((nn.value, 3))
        check(lib.sgpr_get_local(self._h, int(atom), C.addressof(nn), ptr(z), ptr(r), nn.value))
        return z, r

    def profile(self, on=True):
        check(_lib.load().sgpr_profile(self._h, int(bool(on))))

    def stage_times(self):
        ms = np.zeros(32)
        names = C.create_string_buffer(1024)
        n = _lib.load().sgpr_get_stage_times(self._h, ptr(ms), 32, C.addressof(names), 1024)
        return dict(zip(names.value.decode().split(";"), ms[:n])) if n > 0 else {}
