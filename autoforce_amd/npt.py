"""Combined Nose-Hoover / Parrinello-Rahman dynamics with a MOVING cell: what the reference's command line runs for
`dynamics = 'NPT'` with a bulk modulus — `ase.md.npt.NPT(atoms, dt, temperature_K, externalstress, ttime, pfactor, mask)`
around the calculator (theforce/cl/md.py:131-166; `iso`: `set_fraction_traceless(0)`, :163-164; the cell made upper
triangular first, :169-172, util/aseutil.py:61-71).

ASE is a third-party dependency that is absent from the build image, so its integrator is restated here from its
published algorithm (Melchionna, Ciccotti, Holian, Mol. Phys. 78, 533 (1993); Melchionna, Phys. Rev. E 61, 6165 (2000);
Holian, De Groot, Hoover, Hoover, Phys. Rev. A 41, 4552 (1990) for the centred-difference form ASE integrates):

    scaled coordinates   q = r h^-1 - 1/2            (h: rows = cell vectors, upper triangular)
    h_(n+1)    = h_(n-1) + 2 dt h_n eta_n
    eta_(n+1)  = eta_(n-1) + mask * U(-2 dt pfact det(h_n) (sigma_n - sigma_ext))      U: Voigt 6-vector -> upper triangle
    zeta_(n+1) = zeta_(n-1) + 2 dt tfact (KE_n - 3/2 (N - 1) kT)
    q_(n+2)    = (2 q_(n+1) + q_n (B - 1) + dt^2 (F_(n+1) / m) h_(n+1)^-1) (B + 1)^-1,
                 B = dt h_(n+1) (eta_(n+1) + zeta_(n+1)/2 1) h_(n+1)^-1
    p_(n+1)    = m (q_(n+2) - q_n) h_(n+1) / 2 dt
    tfact = 2 / (3 N kT ttime^2),  pfact = 1 / (pfactor det(h_0)),  pfactor = ptime^2 * bulk modulus,
    sigma_n = the calculator's stress at step n MINUS the ideal-gas part sum_i p_i (x) p_i / (m_i V)  (Voigt),
    sigma_ext = (-P, -P, -P, 0, 0, 0) for a scalar external pressure P.

started by one backward step (eta_(-1), zeta_(-1), h_(-1) by half the increments; q_(-1) = q_0 - dt (p/m) h^-1, twice
corrected so that the centred momentum equals the given one).  With pfactor = None (no barostat: eta = 0, h constant) the
recurrence is the one `workloads.nose_hoover_nvt` and the device loop (`sgpr_md_thermostat`) integrate.

Units: eV, Angstrom, amu; time in Angstrom sqrt(amu / eV) (`workloads.FS` per femtosecond); GPa = eV / A^3 / 160.21766208.

The integrator drives calculate() once per step like ASE's does — a moving cell changes the candidate lists' cell every
step; the library keeps them valid under strain (`test_npt_walk_reuses_candidates_under_strain`), so a step is still the
warm path of `sgpr_compute`."""
import time

import numpy as np

GPA = 1.0 / 160.21766208       # ase.units.GPa in eV / A^3
VOIGT = ((0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1))


def make_cell_upper_triangular(positions, cell):
    """util/aseutil.py:61-71 (the reference's `configure_cell`, cl/md.py:169-172): the rigid rotation that puts the third
    cell vector along +z and the second into the yz plane with a positive y component — ASE's two `rotate(..., rotate_cell
    =True)` calls end in that unique frame —, after which h[1,0] = h[2,0] = h[2,1] = 0 (what ase.md.npt.NPT demands of the
    cell).  Returns rotated positions, cell and the rotation R as it acts on row vectors (x_new = x R: velocities, forces)."""
    cell = np.asarray(cell, float)
    ez = cell[2] / np.linalg.norm(cell[2])
    by = cell[1] - (cell[1] @ ez) * ez
    nb = np.linalg.norm(by)
    if nb < 1e-12 * max(np.linalg.norm(cell[1]), 1.0):
        raise ValueError("the second and third cell vectors are parallel")
    ey = by / nb
    ex = np.cross(ey, ez)
    R = np.stack([ex, ey, ez])          # new components = R @ old (a proper rotation: ex x ey = ez)
    new_cell = cell @ R.T
    new_cell[1, 0] = new_cell[2, 0] = new_cell[2, 1] = 0.0
    return np.asarray(positions, float) @ R.T, new_cell, R.T


def _upper(six):
    return np.array(((six[0], six[5], six[4]), (0.0, six[1], six[3]), (0.0, 0.0, six[2])))


def _separate_trace(mat):
    tr = (mat[0, 0] + mat[1, 1] + mat[2, 2]) / 3.0
    trace_part = tr * np.identity(3)
    return trace_part, mat - trace_part


class FilterDeltas:
    """calculator/active.py:46-73 (`ml_filter` of cl/md.py:76-79): wraps the atoms so that the jump a model update puts into
    forces and stress decays smoothly — the accumulated jumps, shrunk by `shrink` at every call and (forces) clamped to
    1 eV/A, are subtracted from what the calculator returns.  Everything else is the wrapped atoms'."""

    def __init__(self, atoms, shrink=0.95):
        object.__setattr__(self, "atoms", atoms)
        object.__setattr__(self, "shrink", shrink)
        object.__setattr__(self, "f", 0)
        object.__setattr__(self, "s", 0)

    def get_forces(self, *args, **kwargs):
        f = self.atoms.get_forces(*args, **kwargs)
        deltas = self.atoms.calc.deltas
        f_acc = self.f
        if deltas:
            f_acc = f_acc + deltas["forces"]
        f_acc = f_acc * self.shrink
        object.__setattr__(self, "f", f_acc)
        return f - np.clip(f_acc, -1.0, 1.0)

    def get_stress(self, *args, **kwargs):
        s = self.atoms.get_stress(*args, **kwargs)
        deltas = self.atoms.calc.deltas
        s_acc = self.s
        if deltas:
            s_acc = s_acc + deltas["stress"]
        s_acc = s_acc * self.shrink
        object.__setattr__(self, "s", s_acc)
        return s - s_acc

    def __getattr__(self, attr):
        return getattr(self.atoms, attr)

    def __setattr__(self, attr, value):
        setattr(self.atoms, attr, value)

    def __len__(self):
        return len(self.atoms)


class NPT:
    """ase.md.npt.NPT restated (module docstring).  `atoms`: anything with the ASE Atoms surface the integrator uses —
    positions / cell / get_masses / get_velocities / set_velocities / get_forces / get_stress / get_potential_energy
    (`ase.Atoms`, or `ase_shim.Atoms`) — with a calculator attached.  temperature in K, externalstress in eV/A^3 (a scalar
    pressure, 6 Voigt components or a 3 x 3 matrix), ttime and pfactor in the units above (None: no thermostat / no barostat),
    mask: 3 or 3 x 3 zeros and ones (which cell components may move)."""

    def __init__(self, atoms, timestep, temperature_K, externalstress=0.0, ttime=None, pfactor=None, mask=None):
        from .ase_shim import kB
        self.atoms = atoms
        self.dt = float(timestep)
        m = np.asarray(atoms.get_masses(), float)
        self.masses = m[:, None]
        v = atoms.get_velocities()
        v = np.zeros((len(m), 3)) if v is None else np.asarray(v, float)
        p = v * self.masses
        p = p - p.sum(0) / len(m)           # NPT.zero_center_of_mass_momentum (ASE subtracts the MEAN momentum per atom)
        self._set_momenta(p)
        self.temperature = kB * float(temperature_K)
        self.set_stress(externalstress)
        self.set_mask(mask)
        self.eta = np.zeros((3, 3))
        self.zeta = 0.0
        self.zeta_integrated = 0.0
        self.initialized = False
        self.ttime = ttime
        self.pfactor_given = pfactor
        self.frac_traceless = 1
        self.timeelapsed = 0.0
        self.nsteps = 0
        self._constants()

    # ------------------------------------------------------------------ settings (ASE's names)
    def set_stress(self, stress):
        if np.isscalar(stress):
            stress = np.array([-stress, -stress, -stress, 0.0, 0.0, 0.0])
        else:
            stress = np.array(stress, float)
            if stress.shape == (3, 3):
                if not np.allclose(stress, stress.T):
                    raise ValueError("The external stress must be a symmetric tensor.")
                stress = np.array([stress[a, b] for a, b in VOIGT])
            elif stress.shape != (6,):
                raise ValueError("The external stress has the wrong shape.")
        self.externalstress = stress

    def set_mask(self, mask):
        mask = np.ones(3) if mask is None else np.array(mask)
        if mask.shape not in ((3,), (3, 3)):
            raise RuntimeError("The mask has the wrong shape (must be a 3-vector or 3x3 matrix)")
        mask = np.not_equal(mask, 0)
        self.mask = np.outer(mask, mask) if mask.shape == (3,) else mask

    def set_fraction_traceless(self, frac):
        self.frac_traceless = frac

    def _constants(self):
        n = len(self.masses)
        self.tfact = 0.0 if self.ttime is None else 2.0 / (3 * n * self.temperature * self.ttime * self.ttime)
        self.pfact = 0.0 if self.pfactor_given is None else 1.0 / (self.pfactor_given * np.linalg.det(self._box()))
        self.desiredEkin = 1.5 * (n - 1) * self.temperature

    # ------------------------------------------------------------------ the atoms
    def _box(self):
        return np.array(getattr(self.atoms.cell, "array", self.atoms.cell), float)

    def _set_momenta(self, p):
        self.p = np.array(p, float)
        self.atoms.set_velocities(self.p / self.masses)

    def kinetic_energy(self):
        return 0.5 * float(np.vdot(self.p, self.p / self.masses))

    def _forces(self):
        return np.array(self.atoms.get_forces(), float)

    def _stress(self):
        """atoms.get_stress(include_ideal_gas=True): the calculator's stress minus sum_i p_i p_i / m_i / V."""
        s = np.array(self.atoms.get_stress(), float)
        invvol = 1.0 / abs(np.linalg.det(self._box()))
        invm = 1.0 / self.masses[:, 0]
        for k, (a, b) in enumerate(VOIGT):
            s[k] -= (self.p[:, a] * self.p[:, b] * invm).sum() * invvol
        return s

    def _set_box_and_positions(self, h, q):
        cell = self.atoms.cell
        if hasattr(cell, "array"):
            self.atoms.set_cell(h)
        else:
            self.atoms.cell = np.array(h, float)
        self.atoms.positions = np.dot(q + 0.5, h)

    # ------------------------------------------------------------------ the integrator
    def _deta(self, factor):
        """factor * dt * pfact * det(h) * (stress - external), as the strain-rate increment with mask / iso applied."""
        if self.pfactor_given is None:
            de = np.zeros(6)
        else:
            de = -factor * self.dt * (self.pfact * np.linalg.det(self.h) * (self._stress() - self.externalstress))
        if self.frac_traceless == 1:
            return self.mask * _upper(de)
        trace_part, traceless_part = _separate_trace(_upper(de))      # (ASE applies no mask on this branch)
        return trace_part + self.frac_traceless * traceless_part

    def _q_future(self, force):
        dt, id3 = self.dt, np.identity(3)
        alpha = (dt * dt) * np.dot(force / self.masses, self.inv_h)
        beta = dt * np.dot(self.h, np.dot(self.eta + 0.5 * self.zeta * id3, self.inv_h))
        inv_b = np.linalg.inv(beta + id3)
        self.q_future = np.dot(2 * self.q + np.dot(self.q_past, beta - id3) + alpha, inv_b)

    def initialize(self):
        dt = self.dt
        self.h = self._box()
        if not (self.h[1, 0] == self.h[2, 0] == self.h[2, 1] == 0.0):
            raise NotImplementedError("Can (so far) only operate on lists of atoms where the computational box is an upper "
                                      "triangular matrix.")
        self.inv_h = np.linalg.inv(self.h)
        self.q = np.dot(np.asarray(self.atoms.positions, float), self.inv_h) - 0.5
        self.h_past = self.h - dt * np.dot(self.h, self.eta)
        self.eta_past = self.eta - self._deta(1.0)
        self.zeta_past = self.zeta - dt * self.tfact * (self.kinetic_energy() - self.desiredEkin)
        # q_past and q_future: a backward step, twice corrected so that the centred momentum is the given one
        p0 = self.p.copy()
        p = p0.copy()
        m = self.masses
        for _ in range(2):
            self.q_past = self.q - dt * np.dot(p / m, self.inv_h)
            self._q_future(self._forces())
            p = np.dot(self.q_future - self.q_past, self.h / (2 * dt)) * m
            if 0.5 * np.sum(np.sum(p * p, -1) / m[:, 0]) / len(m) < 1e-5:
                break
            p = (p0 - p) + p0
        self.initialized = True

    def step(self):
        """One time step: assumes forces and stress of the current configuration are the calculator's current results."""
        if not self.initialized:
            self.initialize()
        dt = self.dt
        h_future = self.h_past + 2 * dt * np.dot(self.h, self.eta)
        eta_future = self.eta_past + self._deta(2.0)
        zeta_future = self.zeta_past + 2 * dt * self.tfact * (self.kinetic_energy() - self.desiredEkin)
        self.timeelapsed += dt
        self.h_past, self.h = self.h, h_future
        self.inv_h = np.linalg.inv(self.h)
        self.q_past, self.q = self.q, self.q_future
        self._set_box_and_positions(self.h, self.q)
        self.eta_past, self.eta = self.eta, eta_future
        self.zeta_past, self.zeta = self.zeta, zeta_future
        self.zeta_integrated += dt * self.zeta
        force = self._forces()
        self._q_future(force)
        self._set_momenta(np.dot(self.q_future - self.q_past, self.h / (2 * dt)) * self.masses)
        self.nsteps += 1

    # ------------------------------------------------------------------ diagnostics
    def get_gibbs_free_energy(self):
        """ase.md.npt.NPT.get_gibbs_free_energy: the conserved quantity of the extended system."""
        n = len(self.masses)
        contractedeta = np.sum((self.eta * self.eta).ravel())
        gibbs = self.atoms.get_potential_energy() + self.kinetic_energy() - np.sum(self.externalstress[0:3]) * np.linalg.det(self.h) / 3.0
        if self.ttime is not None:
            gibbs += 1.5 * n * self.temperature * (self.ttime * self.zeta) ** 2 + 3 * self.temperature * (n - 1) * self.zeta_integrated
        if self.pfactor_given is not None:
            gibbs += 0.5 / self.pfact * contractedeta
        return float(gibbs)

    def run(self, steps):
        """Generator: (step, energy, temperature, wall seconds) per evaluated configuration, 0 … steps (ASE's `run(steps)` calls
        step() `steps` times; configuration 0 is the one the run starts from)."""
        from .ase_shim import kB
        n = len(self.masses)
        t0 = time.time()
        if not self.initialized:
            self.initialize()
        yield 0, float(self.atoms.get_potential_energy()), 2.0 * self.kinetic_energy() / (3 * n * kB), time.time() - t0
        for k in range(1, steps + 1):
            t0 = time.time()
            self.step()
            yield k, float(self.atoms.get_potential_energy()), 2.0 * self.kinetic_energy() / (3 * n * kB), time.time() - t0
