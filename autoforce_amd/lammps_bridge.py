"""LAMMPS `fix external` bridge — drives any calculator with the ASE surface (ActiveCalculator on the
MI355X engine) from a LAMMPS run, the way the reference's command-line driver does
(theforce/cl/lmp.py:8-71).

The reference keeps the LAMMPS handle, the atoms and the calculator in module globals and leans on
`ase.calculators.lammps.convert`; here the state lives in one object and the unit factors needed by
the callback (distance, energy, force, pressure for `metal` and `real`; other unit styles raise) are
tabulated.  `lammps` itself is only touched through the four calls the reference makes:
extract_box, gather_atoms, fix_external_set_energy_global, fix_external_set_virial_global.

    units, numbers_of_type, fix_id, fix_index, commands = read_lammps_file("in.lammps")
    bridge = FixExternalBridge(lmp, calc, units, numbers_of_type, fix_id)
    lmp.set_fix_external_callback(fix_id, bridge)
"""
import numpy as np

try:  # pragma: no cover - ASE is optional
    from ase import Atoms
except ImportError:
    from .ase_shim import Atoms

# 1 [LAMMPS unit] = factor [ASE unit]  (ASE: Angstrom, eV, eV/Angstrom, eV/Angstrom^3), from the SI values
# ase.calculators.lammps.convert is built on (unitconvert_constants.py: LAMMPS' own kim_units numbers, CODATA 2014) —
# the reference converts with that function (cl/lmp.py:3,47-67); tests/golden/lammps_units.json holds the tables
_EV_SI, _KCAL_SI, _AVOGADRO = 1.6021766208e-19, 4184.0, 6.022140857e23
_EV_PER_KCALMOL = _KCAL_SI / _AVOGADRO / _EV_SI
_EV_A3_PER_PA = 1e-30 / _EV_SI
TO_ASE = {
    "metal": dict(distance=1.0, energy=1.0, force=1.0, pressure=1e5 * _EV_A3_PER_PA),                       # bar
    "real": dict(distance=1.0, energy=_EV_PER_KCALMOL, force=_EV_PER_KCALMOL, pressure=101325.0 * _EV_A3_PER_PA),  # atm
}
# cl/lmp.py:74-83 (LAMMPS' own nktv2p table)
NKTV2P = {"lj": 1.0, "real": 68568.415, "metal": 1.6021765e6, "si": 1.0, "cgs": 1.0, "electron": 2.94210108e13,
          "micro": 1.0, "nano": 1.0}


def convert(value, quantity, src, dst):
    """ase.calculators.lammps.convert for the quantities and unit styles the bridge uses."""
    def factor(style):
        if style == "ASE":
            return 1.0
        if style not in TO_ASE:
            raise NotImplementedError(f"LAMMPS units '{style}': only {sorted(TO_ASE)} are tabulated")
        return TO_ASE[style][quantity]
    return np.asarray(value, dtype=float) * (factor(src) / factor(dst))


def read_lammps_file(path):
    """cl/lmp.py:8-35: the commands of a LAMMPS input, its `units`, the `fix <id> ... external` line named
    `autoforce`, and the `#autoforce atomic_numbers = {type: Z, ...}` directive."""
    commands, units, fix_id, fix_index, scope = [], None, None, None, {}
    with open(path) as fh:
        for raw in fh:
            if raw.lower().startswith("#autoforce"):
                exec(raw[10:].strip(), scope)  # the reference evaluates these directives the same way
                continue
            line = " ".join(raw.split("#", 1)[0].split())
            if not line:
                continue
            words = line.split()
            if words[0] == "units":
                units = words[1]
            if line.lower().startswith("fix autoforce"):
                fix_id, fix_index = words[1], len(commands)
            commands.append(line)
    if fix_id is None:
        raise RuntimeError("no fix autoforce!")
    return units, scope["atomic_numbers"], fix_id, fix_index, commands


class FixExternalBridge:
    """The `fix external` callback (cl/lmp.py:44-71): LAMMPS coordinates in, forces / energy / virial out."""

    def __init__(self, lmp, calc, units, atomic_numbers, fix_id):
        self.lmp, self.calc, self.units, self.map_numbers, self.fix_id = lmp, calc, units, dict(atomic_numbers), fix_id
        self.atoms = None

    def get_cell(self):
        boxlo, (xhi, yhi, zhi), xy, yz, xz, pbc, box_change = self.lmp.extract_box()
        return np.array([[xhi, xy, xz], [0.0, yhi, yz], [0.0, 0.0, zhi]]), pbc  # cl/lmp.py:38-42

    def __call__(self, caller, ntimestep, nlocal, tag, pos, fext):
        cell, pbc = self.get_cell()
        cell = convert(cell, "distance", self.units, "ASE")
        xyz = np.array(self.lmp.gather_atoms("x", 1, 3)).reshape(-1, 3)
        positions = convert(xyz, "distance", self.units, "ASE")
        if self.atoms is None:
            types = np.array(self.lmp.gather_atoms("type", 0, 1))
            numbers = [self.map_numbers[int(t)] for t in types]
            self.atoms = Atoms(numbers=numbers, positions=positions, pbc=pbc, cell=cell)
            self.atoms.calc = self.calc
        else:
            self.atoms.cell = cell
            self.atoms.positions = positions
        f = self.atoms.get_forces()[np.asarray(tag) - 1]
        e = self.atoms.get_potential_energy()
        fext[:] = convert(f, "force", "ASE", self.units)
        self.lmp.fix_external_set_energy_global(self.fix_id, float(convert(e, "energy", "ASE", self.units)))
        if "stress" in self.calc.implemented_properties:
            v = convert(self.atoms.get_stress(), "pressure", "ASE", self.units)
            v = -v / (NKTV2P[self.units] / self.atoms.get_volume())
            v[3:] = v[3:][::-1]  # Voigt (yz, xz, xy) -> LAMMPS (xy, xz, yz)
            self.lmp.fix_external_set_virial_global(self.fix_id, v)
