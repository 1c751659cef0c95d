"""Minimal stand-ins for the few ASE names the calculator surface touches, used ONLY when ASE is
not installed (it is absent from the build image; SURVEY.md §8c lists the API the reference's hot
path uses: Atoms.positions/numbers/cell/pbc/get_volume/get_temperature/copy/.calc and
Calculator.__init__/calculate/results/get_property).  With ASE present, autoforce_amd.calculator
subclasses ase.calculators.calculator.Calculator and these classes are not used.
"""
import numpy as np

all_changes = ["positions", "numbers", "cell", "pbc", "initial_charges", "initial_magmoms"]
kB = 8.617330337217213e-05  # eV/K (ase.units.kB)
kcal_mol = 0.04336410390059322  # eV (ase.units.kcal / ase.units.mol)


class Atoms:
    def __init__(self, numbers=None, positions=None, cell=None, pbc=False, velocities=None, masses=None,
                 calculator=None):
        self.numbers = np.asarray(numbers, dtype=int).copy()
        self.positions = np.asarray(positions, dtype=float).reshape(-1, 3).copy()
        self.cell = np.zeros((3, 3)) if cell is None else np.asarray(cell, dtype=float).reshape(3, 3).copy()
        self.pbc = np.broadcast_to(np.asarray(pbc, dtype=bool), (3,)).copy()
        self._velocities = None if velocities is None else np.asarray(velocities, float).copy()
        self._masses = None if masses is None else np.asarray(masses, float).copy()
        self.calc = calculator

    def __len__(self):
        return len(self.numbers)

    def get_global_number_of_atoms(self):
        return len(self)

    def get_atomic_numbers(self):
        return self.numbers.copy()

    def get_positions(self):
        return self.positions.copy()

    def set_positions(self, p):
        self.positions = np.asarray(p, float).reshape(-1, 3).copy()

    def get_cell(self):
        return self.cell.copy()

    def get_pbc(self):
        return self.pbc.copy()

    def get_volume(self):
        v = abs(np.linalg.det(self.cell))
        if v == 0.0:
            raise ValueError("You have atoms with no cell; volume not defined")
        return v

    def get_velocities(self):
        return None if self._velocities is None else self._velocities.copy()

    def set_velocities(self, v):
        self._velocities = np.asarray(v, float).reshape(-1, 3).copy()

    def get_masses(self):
        return np.ones(len(self)) if self._masses is None else self._masses.copy()

    def get_kinetic_energy(self):
        if self._velocities is None:
            return 0.0
        return 0.5 * float((self.get_masses()[:, None] * self._velocities**2).sum())

    def get_temperature(self):
        n = len(self)
        return 0.0 if n == 0 else 2.0 * self.get_kinetic_energy() / (3.0 * n * kB)

    def copy(self):
        return Atoms(self.numbers, self.positions, self.cell, self.pbc, self._velocities, self._masses)

    # ASE protocol: atoms.get_*() -> calc.get_property()
    def _get(self, name):
        if self.calc is None:
            raise RuntimeError("Atoms object has no calculator.")
        return self.calc.get_property(name, self)

    def get_potential_energy(self):
        return float(self._get("energy"))

    def get_forces(self):
        return np.array(self._get("forces"))

    def get_stress(self):
        return np.array(self._get("stress"))


class Calculator:
    implemented_properties = []

    def __init__(self, **kw):
        self.atoms = None
        self.results = {}

    def _changed(self, atoms):
        a = self.atoms
        # (positions first: in a loop they are what has changed — and of them the first coordinate, one scalar comparison,
        # before 3N of them)
        if a is None or len(a) != len(atoms) or (len(a) and a.positions[0, 0] != atoms.positions[0, 0]):
            return True
        return (not np.array_equal(a.positions, atoms.positions)
                or not np.array_equal(a.numbers, atoms.numbers) or not np.array_equal(a.cell, atoms.cell)
                or not np.array_equal(a.pbc, atoms.pbc))

    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        """ase.calculators.calculator.Calculator.calculate: the calculator keeps a COPY of the atoms it was asked about.  The
        copy of the previous call is re-used when the frame is the same system (same numbers): its arrays are overwritten in
        place instead of six fresh allocations per step."""
        if atoms is None:
            return
        a = self.atoms
        if (a is not None and a is not atoms and len(a) == len(atoms) and np.array_equal(a.numbers, atoms.numbers)
                and (a._velocities is None) == (atoms._velocities is None) and (a._masses is None) == (atoms._masses is None)):
            np.copyto(a.positions, atoms.positions)
            np.copyto(a.cell, atoms.cell)
            np.copyto(a.pbc, atoms.pbc)
            if atoms._velocities is not None:
                np.copyto(a._velocities, atoms._velocities)
            if atoms._masses is not None:
                np.copyto(a._masses, atoms._masses)
        else:
            self.atoms = atoms.copy()

    def get_property(self, name, atoms=None):
        if name not in self.implemented_properties:
            raise NotImplementedError(name)
        if atoms is not None and (self._changed(atoms) or name not in self.results):
            self.results = {}
            self.calculate(atoms, [name], all_changes)
        return self.results[name]


class SinglePointCalculator(Calculator):
    """ase.calculators.singlepoint.SinglePointCalculator: stored results for one configuration."""
    implemented_properties = ["energy", "forces", "stress", "free_energy"]

    def __init__(self, atoms, **results):
        Calculator.__init__(self)
        self.atoms = atoms.copy()
        self.results = {k: (np.array(v, float) if k != "energy" else float(v)) for k, v in results.items()
                        if v is not None}

    def get_property(self, name, atoms=None):
        if name not in self.results:
            raise NotImplementedError(f"SinglePointCalculator holds no {name}")
        if atoms is not None and self._changed(atoms):
            raise RuntimeError("SinglePointCalculator: the atoms have changed since the stored calculation")
        return self.results[name]
