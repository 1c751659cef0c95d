"""Train the ML potential on stored configurations, with exact calculations as needed — theforce/cl/offline.py:

    python -m autoforce_amd.cl.offline -i frames.xyz [-r ::10] [-o offline.xyz]      # ARGS must name a calculator

Every frame goes through the ACTIVE calculator: where its uncertainty asks for it the teacher of `ARGS` is called and the
model updated (calculate(), as inside MD); the frames are written back with the model's energies and forces."""
import argparse

from . import gen_active_calc, read_args
from .test import test


def offline(*args, r=None, o="offline.xyz", calc=None):
    if calc is None:
        if read_args().get("calculator") is None:
            raise RuntimeError("set a calculator in ARGS!")
        calc = gen_active_calc()
    return test(*args, r=r, o=o, calc=calc)


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train the ML potential on input configurations. Ab initio calculations will be "
                                             "performed as needed.")
    ap.add_argument("-i", "--input", nargs="*", type=str, help="extended XYZ files")
    ap.add_argument("-r", "--read", type=str, default="::", help="index or [start]:[stop]:[step] e.g. 0 or -1 or ::10")
    ap.add_argument("-o", "--output", type=str, default="offline.xyz")
    a = ap.parse_args(argv)
    offline(*a.input, r=a.read, o=a.output)


if __name__ == "__main__":
    main()
