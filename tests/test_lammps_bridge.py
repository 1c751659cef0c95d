"""The LAMMPS `fix external` bridge (theforce/cl/lmp.py:8-71) against a stand-in for the four `lammps`
calls it makes: unit conversion, tag ordering, energy and the virial in LAMMPS' component order."""
import numpy as np
import pytest

from autoforce_amd.calculator import ActiveCalculator
from autoforce_amd.lammps_bridge import NKTV2P, FixExternalBridge, convert, read_lammps_file
from helpers import OracleEngine, load


class FakeLammps:
    def __init__(self, positions, types, cell, units_factor):
        self.x, self.types, self.cell, self.k = positions / units_factor, types, cell / units_factor, units_factor
        self.energy, self.virial = None, None

    def extract_box(self):
        c = self.cell
        return [0, 0, 0], [c[0, 0], c[1, 1], c[2, 2]], c[0, 1], c[1, 2], c[0, 2], [1, 1, 1], 0

    def gather_atoms(self, name, kind, count):
        return (self.x.reshape(-1) if name == "x" else self.types).tolist()

    def fix_external_set_energy_global(self, fix_id, e):
        self.energy = (fix_id, e)

    def fix_external_set_virial_global(self, fix_id, v):
        self.virial = (fix_id, np.array(v))


def test_read_lammps_file(tmp_path):
    p = tmp_path / "in.lammps"
    p.write_text("# a comment\nunits metal   # style\n#autoforce atomic_numbers = {1: 3, 2: 15, 3: 16}\n"
                 "read_data  data.lips\nfix autoforce all external pf/callback 1 1\nrun 10\n")
    units, numbers, fix_id, fix_index, commands = read_lammps_file(str(p))
    assert (units, numbers, fix_id, fix_index) == ("metal", {1: 3, 2: 15, 3: 16}, "autoforce", 2)
    assert commands == ["units metal", "read_data data.lips", "fix autoforce all external pf/callback 1 1", "run 10"]
    (tmp_path / "bad").write_text("units metal\n#autoforce atomic_numbers = {1: 3}\n")
    with pytest.raises(RuntimeError, match="no fix autoforce"):
        read_lammps_file(str(tmp_path / "bad"))


@pytest.mark.parametrize("units", ["metal", "real"])
def test_callback(units):
    g = load("g5_tric24")
    cell = g["cell"]
    # LAMMPS boxes are upper-triangular in this convention: rotate nothing, use a triangular test cell
    cell = np.triu(cell) + np.diag([0.0, 0.0, 0.0])
    calc = ActiveCalculator(engine=OracleEngine(g), logfile=None)
    zs = sorted(set(g["numbers"].tolist()))
    types = np.array([zs.index(z) + 1 for z in g["numbers"]])
    lmp = FakeLammps(g["positions"], types, cell, 1.0)
    bridge = FixExternalBridge(lmp, calc, units, {k + 1: z for k, z in enumerate(zs)}, "autoforce")
    N = len(types)
    tag = np.random.default_rng(0).permutation(N) + 1
    fext = np.zeros((N, 3))
    bridge(None, 0, N, tag, None, fext)
    ref = calc.engine.predict(g["numbers"], g["positions"], cell, [True] * 3)
    np.testing.assert_allclose(fext, convert(ref["forces"][tag - 1], "force", "ASE", units), rtol=1e-12, atol=1e-14)
    assert lmp.energy[0] == "autoforce"
    assert abs(lmp.energy[1] - float(convert(ref["energy"], "energy", "ASE", units))) < 1e-12
    vol = abs(np.linalg.det(cell))
    want = -convert(ref["stress"], "pressure", "ASE", units) / (NKTV2P[units] / vol)
    np.testing.assert_allclose(lmp.virial[1], want[[0, 1, 2, 5, 4, 3]], rtol=1e-12, atol=1e-16)
    # second call: same atoms object, new positions
    lmp.x = lmp.x + 0.01
    bridge(None, 1, N, tag, None, fext)
    ref2 = calc.engine.predict(g["numbers"], g["positions"] + 0.01, cell, [True] * 3)
    np.testing.assert_allclose(fext, convert(ref2["forces"][tag - 1], "force", "ASE", units), rtol=1e-12, atol=1e-14)
    if units == "metal":
        assert abs(convert(1.0, "pressure", "metal", "ASE") * 1.6021766208e6 - 1.0) < 1e-12  # 1 bar in eV/A^3 (ASE's e)


def _fixture():
    import json
    import os
    from conftest import GOLDEN
    return json.load(open(os.path.join(GOLDEN, "lammps_units.json")))


@pytest.mark.parametrize("units", ["metal", "real"])
def test_units_against_the_lammps_tables(units):
    """tests/golden/lammps_units.json (LAMMPS' documented unit styles and Update::set_units constants, the SI values
    ase.calculators.lammps.convert uses): every factor of the bridge re-derived from the fixture, the nktv2p column the
    reference copies, and the physical identity that ties them — the virial handed to LAMMPS is -stress x volume in
    LAMMPS energy units."""
    fx = _fixture()
    st, si = fx["styles"][units], fx["si"]
    assert NKTV2P[units] == st["nktv2p"]
    e_si = {"ev": si["ev"], "kcal/mol": si["kcal"] / si["avogadro"]}[st["energy"]]
    p_si = si[st["pressure"]]
    assert st["distance"] == "angstrom"
    assert abs(float(convert(1.0, "distance", units, "ASE")) - 1.0) == 0.0
    assert abs(float(convert(1.0, "energy", units, "ASE")) - e_si / si["ev"]) <= 1e-15
    assert abs(float(convert(1.0, "force", units, "ASE")) - e_si / si["ev"]) <= 1e-15
    assert abs(float(convert(1.0, "pressure", units, "ASE")) / (p_si * si["angstrom"] ** 3 / si["ev"]) - 1.0) <= 1e-15
    # nktv2p converts [energy / volume] of the style into its pressure unit (LAMMPS doc: "nktv2p"): consistent with
    # the SI table to the vintage of the constants (LAMMPS' 1.6021765e-19 C against 1.6021766208e-19)
    assert abs(st["nktv2p"] / (e_si / si["angstrom"] ** 3 / p_si) - 1.0) < 2e-7
    # the callback's virial: -sigma V, energy units of the style
    sigma = np.array([0.011, -0.007, 0.004, 0.0021, -0.0013, 0.0008])   # eV / A^3, Voigt
    vol = 812.5
    v = -convert(sigma, "pressure", "ASE", units) / (NKTV2P[units] / vol)
    np.testing.assert_allclose(convert(v, "energy", units, "ASE"), -sigma * vol, rtol=2e-7, atol=0)


def test_virial_component_order():
    """LAMMPS: xx, yy, zz, xy, xz, yz; ASE Voigt: xx, yy, zz, yz, xz, xy — the bridge reverses the last three
    (cl/lmp.py:70), checked through the callback with a calculator whose stress names its components."""
    fx = _fixture()
    lam, ase = fx["virial_order_lammps"], fx["voigt_order_ase"]
    perm = [ase.index(c) for c in lam]
    assert perm == [0, 1, 2, 5, 4, 3]

    class Calc:
        implemented_properties = ["energy", "forces", "stress"]

        def get_property(self, name, atoms=None):
            return {"energy": 0.0, "forces": np.zeros((2, 3)), "stress": np.array([1.0, 2.0, 3.0, 4.0, 5.0, 6.0])}[name]

    cell = np.diag([5.0, 6.0, 7.0])
    lmp = FakeLammps(np.zeros((2, 3)), np.array([1, 1]), cell, 1.0)
    bridge = FixExternalBridge(lmp, Calc(), "metal", {1: 3}, "autoforce")
    bridge(None, 0, 2, np.array([1, 2]), None, np.zeros((2, 3)))
    got = lmp.virial[1] / lmp.virial[1][0]      # (xx carries 1.0)
    np.testing.assert_allclose(got, np.array([1.0, 2.0, 3.0, 6.0, 5.0, 4.0]), rtol=1e-14)
