"""Synthetic workloads of BASELINE.json (recipes: SURVEY.md §8d).  Host-side numpy only.

The reference ships no structure files for these systems, so frames are generated from fixed
seeds: simple-cubic sites, shuffled species, Gaussian rattle with a minimum-distance rejection.
"""
import numpy as np

from .model import Local


def _rattled_lattice(shape, spacing, sigma, rng, dmin=1.6):
    g = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).reshape(-1, 3) * spacing
    pos = g.astype(float)
    cell = np.diag([n * spacing for n in shape]).astype(float)
    # rattle with rejection: a displacement is redrawn until no lattice neighbour (pre-rattle
    # distance = spacing) can come closer than dmin; cheap sufficient test per atom
    out = np.empty_like(pos)
    lim = (spacing - dmin) / 2.0
    for i in range(len(pos)):
        while True:
            d = sigma * rng.normal(size=3)
            if np.linalg.norm(d) <= lim:
                break
        out[i] = pos[i] + d
    return out, cell


def lips(n_side=16, seed=0, sigma=0.15):
    """C3/C4: "LiPS" 16^3 = 4096 simple-cubic sites, 2.72 A, 1536 Li / 512 P / 2048 S."""
    rng = np.random.default_rng(seed)
    dims = tuple(n_side) if np.ndim(n_side) else (n_side,) * 3  # (32, 32, 16): the 16384 atoms of config 5
    N = int(np.prod(dims))
    nLi, nP = 3 * N // 8, N // 8
    numbers = rng.permutation(np.array([3] * nLi + [15] * nP + [16] * (N - nLi - nP))).astype(np.int32)
    pos, cell = _rattled_lattice(dims, 2.72, sigma, rng)
    return numbers, pos, cell, np.array([True, True, True])


def si_diamond(reps=(2, 2, 1), seed=0, sigma=0.05):
    """C1: diamond Si a=5.431, 8-atom cubic x (2,2,1) = 32 atoms; L_z = 5.431 < rc: atoms meet their own images."""
    rng = np.random.default_rng(seed)
    a = 5.431
    base = np.array([[0, 0, 0], [0, .5, .5], [.5, 0, .5], [.5, .5, 0], [.25, .25, .25], [.25, .75, .75],
                     [.75, .25, .75], [.75, .75, .25]]) * a
    cells = np.stack(np.meshgrid(*[np.arange(n) for n in reps], indexing="ij"), -1).reshape(-1, 3) * a
    pos = (cells[:, None, :] + base[None]).reshape(-1, 3) + sigma * rng.normal(size=(len(cells) * 8, 3))
    cell = np.diag([n * a for n in reps]).astype(float)
    return np.full(len(pos), 14, np.int32), pos, cell, np.array([True, True, True])


def li_bcc(reps=(8, 4, 4), seed=0, sigma=0.10):
    """C2: bcc Li a=3.49, 2-atom cubic x reps = 256 atoms."""
    rng = np.random.default_rng(seed)
    a = 3.49
    base = np.array([[0, 0, 0], [0.5, 0.5, 0.5]]) * a
    cells = np.stack(np.meshgrid(*[np.arange(n) for n in reps], indexing="ij"), -1).reshape(-1, 3) * a
    pos = (cells[:, None, :] + base[None]).reshape(-1, 3) + sigma * rng.normal(size=(len(cells) * 2, 3))
    cell = np.diag([n * a for n in reps]).astype(float)
    return np.full(len(pos), 3, np.int32), pos, cell, np.array([True, True, True])


def oxide(shape=(32, 32, 16), seed=0, sigma=0.15):
    """C5: 16384 sites, 4 species 2:1:1:4 (Li, Zr, La, O)."""
    rng = np.random.default_rng(seed)
    N = int(np.prod(shape))
    counts = [2 * N // 8, N // 8, N // 8]
    z = [3] * counts[0] + [40] * counts[1] + [57] * counts[2]
    z += [8] * (N - len(z))
    numbers = rng.permutation(np.array(z)).astype(np.int32)
    pos, cell = _rattled_lattice(shape, 2.72, sigma, rng)
    return numbers, pos, cell, np.array([True, True, True])


def oxide_ordered(shape=(32, 32, 16), seed=0, sigma=0.05):
    """C5 for on-the-fly MD: the same 4 species and 2:1:1:4 stoichiometry (Li, Zr, La, O) on the same 2.72 A sites, but
    ORDERED — a rocksalt-type arrangement, O on the odd-parity sublattice, Zr / La / Li / Li on the four sites of the
    even one inside every 2x2x2 cube.  Its environments repeat, so a learner can cover them with ~10^3 inducing LCEs:
    in the randomly shuffled `oxide` every atom is its own environment (active.py:631-654 keeps every LCE whose
    similarity to the kept ones is below 0.95) and the reference's sampling loop would never stop adding."""
    rng = np.random.default_rng(seed)
    if any(n % 2 for n in shape):
        raise ValueError("even numbers of sites per direction")
    g = np.stack(np.meshgrid(*[np.arange(n) for n in shape], indexing="ij"), -1).reshape(-1, 3)
    par = g % 2
    even = par.sum(1) % 2 == 0
    key = par[:, 0] * 4 + par[:, 1] * 2 + par[:, 2]      # even sites: 0 (000), 3 (011), 5 (101), 6 (110)
    numbers = np.full(len(g), 8, np.int32)
    numbers[even & (key == 0)] = 40
    numbers[even & (key == 6)] = 57
    numbers[even & ((key == 3) | (key == 5))] = 3
    pos = g * 2.72 + sigma * rng.normal(size=(len(g), 3))
    return numbers, pos, np.diag([n * 2.72 for n in shape]).astype(float), np.array([True, True, True])


class PairTeacher:
    """Stand-in for the ab initio teacher of an on-the-fly run (a real run passes any ASE calculator: VASP, GPAW, ...):
    phi(r) = eps [(1 - e^{-a (r - r0)})^2 - 1] (1 - (r/rc)^2)^2 summed over the pairs of the DEVICE neighbour list,
    analytic forces and stress, ASE-calculator protocol (get_property(name, atoms))."""
    implemented_properties = ["energy", "forces", "stress", "free_energy"]

    def __init__(self, species, rc=5.0, eps=0.25, a=1.4, r0=2.8, device=0):
        from .model import SGPRModel
        self.rc, self.eps, self.a, self.r0 = rc, eps, a, r0
        self.nl = SGPRModel(3, 3, 4, rc, species=species, device=device)  # used for its neighbour list only
        self.calls, self.seconds, self.results, self._key = 0, 0.0, {}, None

    def calculate(self, atoms):
        import time
        t0 = time.time()
        self.calls += 1
        pos = np.asarray(atoms.positions, float)
        cell = np.asarray(getattr(atoms.cell, "array", atoms.cell), float)
        N = len(pos)
        self.nl.predict(atoms.numbers, pos, cell, atoms.pbc, beta=False)
        ptr, j, off = self.nl.neighbors(N)
        i = np.repeat(np.arange(N), np.diff(ptr))
        d = pos[j] - pos[i] + off @ cell
        r = np.linalg.norm(d, axis=1)
        x = np.exp(-self.a * (r - self.r0))
        m_, dm = self.eps * ((1 - x) ** 2 - 1), self.eps * 2 * (1 - x) * self.a * x
        s = 1 - (r / self.rc) ** 2
        phi, dphi = m_ * s * s, dm * s * s - m_ * 4 * s * r / self.rc ** 2
        g = (dphi / r)[:, None] * d
        F = np.stack([np.bincount(i, weights=g[:, k], minlength=N) for k in range(3)], axis=1)
        vir = 0.5 * np.einsum("pa,pb->ab", d, g)
        stress = (vir / abs(np.linalg.det(cell)))[[0, 1, 2, 1, 0, 0], [0, 1, 2, 2, 2, 1]]
        self.results = dict(energy=0.5 * phi.sum(), forces=F, stress=stress, free_energy=0.5 * phi.sum())
        self.seconds += time.time() - t0

    def get_property(self, name, atoms=None):
        key = None if atoms is None else atoms.positions.tobytes()
        if atoms is not None and key != self._key:
            self.calculate(atoms)
            self._key = key
        return self.results[name]

    def close(self):
        self.nl.close()


FS = 0.09822694788464063  # ase.units.fs: 1 fs in A sqrt(amu/eV)
MASS = {1: 1.008, 3: 6.94, 8: 15.999, 9: 18.998, 11: 22.99, 12: 24.305, 14: 28.085, 15: 30.974, 16: 32.06, 17: 35.45,
        40: 91.224, 57: 138.905}


def langevin_nvt(calc, numbers, pos, cell, pbc, steps, temperature=600.0, dt_fs=1.0, friction=1e-3, seed=1, vel=None, rng=None):
    """BAOAB Langevin dynamics in numpy around any calculator with the ASE surface; parameters as the reference's
    driver (cl/md.py:31,70-74: dt = 1 fs, friction 1e-3 per ASE time unit, T = 600 K; Maxwell-Boltzmann start as
    util/aseutil.py:11-20, or the velocities handed over).  Generator: yields (step, energy, temperature, wall seconds,
    positions, velocities) after every step.  With ASE installed, ase.md.langevin.Langevin drives the same calculator."""
    import time
    from .ase_shim import Atoms, kB
    rng = np.random.default_rng(seed) if rng is None else rng
    N = len(numbers)
    mass = np.array([MASS[int(z)] for z in numbers])[:, None]
    kT = kB * temperature
    if vel is None:
        vel = rng.normal(size=(N, 3)) * np.sqrt(kT / mass)
        vel -= (mass * vel).sum(0) / mass.sum()
    vel = np.array(vel, float)
    dt = dt_fs * FS
    c1 = np.exp(-friction * dt)
    c2 = np.sqrt(1 - c1 * c1)
    pos = np.array(pos, float)

    def forces(p, v):
        at = Atoms(numbers, p, cell, pbc, velocities=v, masses=mass[:, 0])
        at.calc = calc
        return at.get_forces(), at.get_potential_energy()

    t0 = time.time()
    F, E = forces(pos, vel)
    yield 0, E, float((mass * vel ** 2).sum() / (3 * N * kB)), time.time() - t0, pos, vel
    for step in range(1, steps + 1):
        t0 = time.time()
        vel += 0.5 * dt * F / mass
        pos = pos + 0.5 * dt * vel
        vel = c1 * vel + c2 * np.sqrt(kT / mass) * rng.normal(size=(N, 3))
        pos = pos + 0.5 * dt * vel
        F, E = forces(pos, vel)
        vel += 0.5 * dt * F / mass
        yield step, E, float((mass * vel ** 2).sum() / (3 * N * kB)), time.time() - t0, pos, vel


def _device_order_sum(x):
    """Sum of x in the order of md_nh_kernel (api.hip): 256 strided partial sums (thread t adds the elements t, t + 256, ...
    one after the other), then a pairwise tree in natural order."""
    x = np.asarray(x, float)
    pad = (-len(x)) % 256
    rows = np.concatenate([x, np.zeros(pad)]).reshape(-1, 256)
    p = np.zeros(256)
    for r in rows:
        p = p + r
    while len(p) > 1:
        p = p[0::2] + p[1::2]
    return float(p[0])


def nose_hoover_nvt(calc, numbers, pos, cell, pbc, steps, temperature=600.0, dt_fs=1.0, tdamp_fs=25.0, vel=None, seed=1, species=None):
    """Nose-Hoover NVT in numpy around any calculator with the ASE surface: the reference's DEFAULT dynamics —
    md(dynamics="NPT", bulk_modulus=None) = ase.md.npt.NPT(pfactor=None, ttime=tdamp fs), cl/md.py:17, :131-166 — restated
    from ASE's published algorithm (Melchionna, Ciccotti, Holian 1993; ASE is absent here):
        x_(n+1) = (2 x_n - x_(n-1) (1 - b) + dt^2 F_n / m) / (1 + b),  b = dt zeta_n / 2,  v_n = (x_(n+1) - x_(n-1)) / 2 dt
        zeta_(n+1) = zeta_(n-1) + 2 dt tfact (KE_n - 1.5 (N - 1) kT),  tfact = 2 / (3 N kT ttime^2)
    started with x_(-1) = x_0 - dt v_0 + dt^2 F_0 / 2m, zeta_0 = 0, zeta_(-1) = -dt tfact (KE_0 - ...).  The host twin of the
    device loop (sgpr_md_thermostat): same operations in the same order, bit for bit (the kinetic energy is summed over the
    atoms in the library's species-sorted order: `species` = the model's table, default the sorted atomic numbers).  Yields
    (step, energy, temperature, wall seconds, positions, velocities, zeta, integral of zeta) per evaluated configuration."""
    import time
    from .ase_shim import Atoms, kB
    N = len(numbers)
    mass = np.array([MASS[int(z)] for z in numbers])[:, None]
    kT = kB * temperature
    if vel is None:
        rng = np.random.default_rng(seed)
        vel = rng.normal(size=(N, 3)) * np.sqrt(kT / mass)
        vel -= (mass * vel).sum(0) / mass.sum()
    v0 = np.array(vel, float)
    dt = dt_fs * FS
    hdt = 0.5 * dt
    dt = 2.0 * hdt
    ttime = tdamp_fs * FS
    tfact = 2.0 / (float(3 * N) * kT * ttime * ttime)
    c1, c2, K0 = dt * tfact, 2.0 * dt * tfact, 1.5 * float(N - 1) * kT
    x = np.array(pos, float)
    xp = None
    zeta, zint = {0: 0.0}, {0: 0.0}
    table = sorted(set(int(z) for z in numbers)) if species is None else [int(z) for z in species]
    order = np.argsort([table.index(int(z)) if int(z) in table else len(table) for z in numbers], kind="stable")

    def forces(p, v):
        at = Atoms(numbers, p, cell, pbc, velocities=v, masses=mass[:, 0])
        at.calc = calc
        return at.get_forces(), at.get_potential_energy()

    for n in range(steps + 1):
        t0 = time.time()
        F, E = forces(x, v0 if n == 0 else v)   # (the velocities the integrator holds when it asks for forces: v_(n-1))
        a = ((dt * dt) * F) / mass
        if n == 0:
            xp = (x - dt * v0) + 0.5 * a
        b = hdt * zeta[n]
        xn = (((2.0 * x) - xp * (1.0 - b)) + a) / (1.0 + b)
        v = v0 if n == 0 else (xn - xp) / (2.0 * dt)
        ke3 = mass * (v * v)
        ke_atom = (ke3[:, 0] + ke3[:, 1]) + ke3[:, 2]
        KE = 0.5 * _device_order_sum(ke_atom[order])
        d = KE - K0
        zprev = -(c1 * d) if n == 0 else zeta[n - 1]
        zeta[n + 1] = zprev + c2 * d
        zint[n + 1] = zint[n] + dt * zeta[n + 1]
        yield n, E, float(2.0 * KE / (3 * N * kB)), time.time() - t0, x, v, zeta[n], zint[n]
        xp, x = x, xn


def langevin_nvt_device(model, numbers, pos, cell, pbc, steps, temperature=600.0, dt_fs=1.0, friction=1e-3, seed=1, vel=None,
                        ediff=0.0, chunk=256, on_halt=None, device_rng=False):
    """langevin_nvt with the state in device memory (SGPRModel.md_begin / md_run): same scheme, same random stream
    (one rng.normal(size=(N, 3)) per step, drawn here and uploaded a chunk at a time), so positions and velocities equal
    the host loop's bit for bit.  Yields (step, energy, temperature, largest covloss) per evaluation.  With ediff > 0 an
    evaluation whose largest covloss reaches it stops the run ON THE DEVICE; on_halt(model, state) — the model update of
    calculator/active.py:477-484 — is called with that configuration and its results, and the evaluation is repeated
    with whatever model on_halt left behind (an on_halt that leaves the covloss above ediff must raise ediff itself:
    it receives and may return the threshold)."""
    from .ase_shim import kB
    rng = np.random.default_rng(seed)
    N = len(numbers)
    mass = np.array([MASS[int(z)] for z in numbers])
    kT = kB * temperature
    if vel is None:
        vel = rng.normal(size=(N, 3)) * np.sqrt(kT / mass[:, None])
        vel -= (mass[:, None] * vel).sum(0) / mass.sum()
    # device_rng: the deviates are drawn on the device (counter-based on `seed`; SGPRModel.md_deviates returns them for a
    # host twin) instead of from numpy here: nothing is generated or uploaded on the step's path
    model.md_begin(numbers, pos, cell, pbc, mass, vel, dt=dt_fs * FS, friction=friction, kT=kT,
                   seed=(int(seed) or 1) if device_rng else 0)
    done = 0                      # evaluations accepted so far (evaluation k = the configuration after k steps)
    rows = np.empty((0, N, 3))    # deviates drawn and not yet consumed: rows[0] moves the current configuration on
    skip_gate = False
    while done <= steps:
        n = 1 if skip_gate else min(chunk, steps + 1 - done)
        final = done + n == steps + 1
        need = 0 if device_rng else (n - 1 if final else n)
        if len(rows) < need:  # (numpy fills an (r, N, 3) request like r requests of (N, 3): the host loop's stream)
            rows = np.concatenate([rows, rng.normal(size=(need - len(rows), N, 3))])
        noise = None if device_rng else (rows[:n] if len(rows) >= n else np.concatenate([rows, np.zeros((n - len(rows), N, 3))]))
        sc, code = model.md_run(n, noise, ediff=0.0 if skip_gate else ediff, final=final)
        accepted = len(sc) - 1 if code == 1 else len(sc)
        if accepted:
            skip_gate = False   # (code 2 with nothing accepted: a capacity was outgrown, the same call again re-sizes it)
        for r in sc[:accepted]:
            yield done, float(r[0]), float(r[12] / (3 * N * kB)), float(r[11])
            done += 1
        rows = rows[accepted:]
        if code == 1:
            if on_halt is None:
                raise RuntimeError("the covloss gate fired and no on_halt handler is installed")
            on_halt(model, model.md_state(results=True))
            skip_gate = True   # the repeated evaluation stands whatever its covloss (active.py:477-484 updates once per step)
    # (the state stays readable: SGPRModel.md_state; md_begin starts the next run)


def fit_to_teacher(model, numbers, pos, cell, pbc, noise=0.05, device=0):
    """Weights of `model` fitted to the PairTeacher's energy and forces on one frame (its inducing set as it is): a
    model whose forces hold the atoms together, for MD loops that have to run for hundreds of steps."""
    teacher = PairTeacher(sorted(set(int(z) for z in numbers)), device=device)
    at = type("A", (), dict(numbers=np.asarray(numbers), positions=np.asarray(pos, float), cell=np.asarray(cell, float), pbc=pbc))()
    teacher.calculate(at)
    mu = model.fit([dict(numbers=numbers, positions=pos, cell=cell, pbc=pbc, energy=teacher.results["energy"],
                         forces=teacher.results["forces"])], noise=noise)
    teacher.close()
    return mu


def inducing_from_frame(model, numbers, pos, cell, pbc, m, seed, noise=0.05):
    """m LCEs drawn species-proportionally from a frame (+ noise), using the MODEL's own device
    neighbour list (the product path; no oracle involved)."""
    rng = np.random.default_rng(seed)
    N = len(numbers)
    model.predict(numbers, pos, cell, pbc, beta=False)  # builds the neighbour list on the device
    ptr, j, off = model.neighbors(N)
    numbers = np.asarray(numbers)
    picks = []
    zs, cnt = np.unique(numbers, return_counts=True)
    quota = np.floor(cnt / N * m).astype(int)
    quota[np.argmax(cnt)] += m - quota.sum()
    for z, q in zip(zs, quota):
        picks.extend(rng.choice(np.nonzero(numbers == z)[0], size=q, replace=False).tolist())
    X = []
    rc = model.cutoff
    for a in picks:
        s = slice(int(ptr[a]), int(ptr[a + 1]))
        r = pos[j[s]] - pos[a] + off[s].astype(float) @ cell
        r = r + noise * rng.normal(size=r.shape)
        keep = np.linalg.norm(r, axis=1) < rc - 1e-3
        X.append(Local(int(numbers[a]), numbers[j[s]][keep], r[keep]))
    return X


def config5_preseeded(shape=(32, 32, 16), m_seed=1000, max_inducing=1024, n_exceed=8, seed=0, device=0, temperature=600.0,
                      friction=0.1, n_equil=250, **calc_kw):
    """BASELINE config 5 with the model ALREADY near its size limit, in a STATIONARY state: the ordered 4-species oxide
    (16384 atoms for the default shape) is first equilibrated at `temperature` by `n_equil` steps of Langevin MD under
    the teacher alone; an SGPR model is pre-seeded with `m_seed` inducing LCEs drawn species-proportionally from two
    frames of that trajectory and fitted to a third; a fourth, held out, sets the sampling threshold: ediff = the
    `n_exceed`-th largest covloss of that frame — so that on-the-fly MD continued from there offers a handful of
    environments per step (not none, not all: the default 2 kcal/mol is calibrated for DFT energies), reaches
    `max_inducing` within a few update steps, and from then on every update ends in downsize(lii=True)
    (calculator/active.py:963-969 -> regression/gppotential.py:829-832).
    Returns (calc, teacher, (numbers, positions, cell, pbc), velocities)."""
    from .calculator import ActiveCalculator
    from .model import SGPRModel
    from .posterior import Frame, PosteriorPotential
    numbers, pos, cell, pbc = oxide_ordered(shape, seed=seed, sigma=0.05)
    species = sorted(set(int(z) for z in numbers))
    teacher = PairTeacher(species, device=device)
    marks = sorted({max(1, int(n_equil * f)) for f in (0.6, 0.75, 0.9)} | {n_equil})
    snaps = {}
    vel = None
    for step, E, T, wall, p, v in langevin_nvt(teacher, numbers, pos, cell, pbc, n_equil, temperature, 1.0, friction, seed=seed + 5):
        if step in marks:
            snaps[step] = (p.copy(), teacher.results["energy"], teacher.results["forces"].copy(), teacher.results["stress"].copy())
        pos, vel = p, v
    model = SGPRModel(3, 3, 4, 6.0, species=species, device=device)
    X = []
    for k, st in enumerate(marks[:2]):
        X += inducing_from_frame(model, numbers, snaps[st][0], cell, pbc, m_seed // 2 + (k == 0) * (m_seed % 2), seed=seed + 21 + k, noise=0.0)
    p3, e3, f3, s3 = snaps[marks[2]]
    post = PosteriorPotential(model)
    post.set_data([Frame(numbers, p3, cell, pbc, e3, f3, s3)], X)
    post.make_munu(algo=3, noise_f=calc_kw.get("noise_f", 0.043))
    beta = np.sort(np.asarray(model.predict(numbers, snaps[marks[3]][0], cell, pbc)["beta"]))
    ediff = float(beta[-max(1, int(n_exceed))])
    kw = dict(logfile=None, tape=None, pckl=None, ediff=ediff, fdiff=3 * ediff, ediff_tot=2 * ediff, max_inducing=max_inducing)
    kw.update(calc_kw)
    calc = ActiveCalculator(covariance=post, calculator=teacher, **kw)
    return calc, teacher, (numbers, pos, cell, pbc), vel
