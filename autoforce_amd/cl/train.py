"""Train the ML potential from stored data — theforce/cl/train.py:

    python -m autoforce_amd.cl.train -i run1.sgpr frames.xyz [-r ::10]      # keywords from ./ARGS

`.sgpr` tapes of other runs are replayed through `ActiveCalculator.include_tape` (their LCEs through the sampling rule,
their frames through the data-acceptance test, calculator/active.py:1006-1053; `-r N`: the first N frames), labelled
frames (extended XYZ with energy / forces) through `include_data` (active.py:989-1004; `-r start:stop:step`)."""
import argparse

from . import gen_active_calc
from .md import read_frames


def train(*args, r=None, calc=None):
    calc = gen_active_calc() if calc is None else calc
    for arg in args:
        if arg.endswith(".sgpr"):
            if r is not None and r != "::":
                try:
                    ndata = int(r)
                except ValueError:
                    raise RuntimeError("For .sgpr files use -r with an integer (e.g. -r 100)")
            else:
                ndata = None
            calc.include_tape(arg, ndata=ndata)
        else:
            calc.include_data(read_frames(arg, r))
    return calc


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train ML potential using data")
    ap.add_argument("-i", "--input", nargs="*", type=str, help=".xyz (extended, labelled) or .sgpr")
    ap.add_argument("-r", "--read", type=str, default="::", help="index or [start]:[stop]:[step] e.g. 0 or -1 or ::10")
    a = ap.parse_args(argv)
    train(*a.input, r=a.read)


if __name__ == "__main__":
    main()
