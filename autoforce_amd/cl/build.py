"""Build the model of ./ARGS from its tape in one shot — theforce/cl/build.py: `python -m autoforce_amd.cl.build`
(`ActiveCalculator.build`, calculator/active.py:1065-1113: every frame and every LCE of the `.sgpr` tape, one fit)."""
from . import gen_active_calc


def main(argv=None):
    calc = gen_active_calc()
    calc.build()
    return calc


if __name__ == "__main__":
    main()
