"""Test the ML potential on stored frames — theforce/cl/test.py:

    python -m autoforce_amd.cl.test -i frames.xyz [-r ::10] [-o test.xyz]      # ARGS must say calculator = None

Every frame goes through `ActiveCalculator.calculate()` (no teacher: evaluation only, cl/test.py:10-11) and is written
back with the model's energy and forces, extended XYZ."""
import argparse

from . import gen_active_calc, read_args
from ..ase_shim import Atoms
from ..sgprio import Frame, format_extxyz
from .md import read_frames


def test(*args, r="::", o="test.xyz", calc=None):
    if calc is None:
        if read_args().get("calculator") is not None:
            raise RuntimeError("set calculator = None in ARGS!")
        calc = gen_active_calc()
    out = open(o, "w") if (o and calc.rank == 0) else None
    results = []
    for arg in args:
        for fr in read_frames(arg, r):
            atoms = Atoms(fr.numbers, fr.positions, fr.cell, fr.pbc)
            atoms.calc = calc
            forces = atoms.get_forces()
            energy = atoms.get_potential_energy()
            results.append((energy, forces.copy()))
            if out is not None:
                out.writelines(format_extxyz(Frame(fr.numbers, fr.positions, fr.cell, fr.pbc, energy, forces, None)))
    if out is not None:
        out.close()
    return results


def single_point(i, o, calc=None):
    """theforce/cl/singlepoint.py: the last frame of `i` with the model's energy and forces written to `o`."""
    return test(i, r="-1", o=o, calc=calc)[0]


def main(argv=None):
    ap = argparse.ArgumentParser(description="Test the ML potential on input data")
    ap.add_argument("-i", "--input", nargs="*", type=str, help="extended XYZ files")
    ap.add_argument("-r", "--read", type=str, default="::", help="index or [start]:[stop]:[step] e.g. 0 or -1 or ::10")
    ap.add_argument("-o", "--output", type=str, default="test.xyz")
    a = ap.parse_args(argv)
    test(*a.input, r=a.read, o=a.output)


if __name__ == "__main__":
    main()
