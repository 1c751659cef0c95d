"""Command-line drivers in the reference's convention (theforce/cl/__init__.py): the run is configured by a file `ARGS`
in the working directory — one `keyword = value` per line, `#` comments — whose keywords are those of
`ActiveCalculator.__init__` plus those of the driver function (`md`), exactly as `theforce.cl.gen_active_calc` /
`update_args` split them.

    # ARGS
    covariance = 'model.npz'      # or 'pckl' semantics of the reference: a folder / file the model is kept in
    calculator = None             # None: evaluate only; 'PAIR': the built-in pair-potential teacher; 'teacher.py': a script
    ediff = 0.05                  #   that defines `calc` (the reference starts its teachers from such scripts,
    dynamics = 'Langevin'         #   theforce/cl/__init__.py:29-55, through a socket: one process per GPU here)
    tem = 600.
    picos = 0.5

The values are evaluated as Python literals with the reference's few names in scope (inf, kcal_mol, arange, linspace)."""
import inspect
import os

import numpy as np

from ..calculator import ActiveCalculator, inf, kcal_mol


def strip(line):
    return line[: line.index("#")].strip() if "#" in line else line.strip()


def get_default_args(func):
    """theforce/util/util.py::get_default_args: {keyword: default} of a function's signature."""
    return {k: v.default for k, v in inspect.signature(func).parameters.items() if v.default is not inspect.Parameter.empty}


def read_args(path="ARGS"):
    """theforce/cl/__init__.py:103-111."""
    if not os.path.isfile(path):
        return {}
    lines = [strip(ln) for ln in open(path).readlines()]
    text = ",".join(ln for ln in lines if ln)
    scope = {"__builtins__": {}, "inf": inf, "kcal_mol": kcal_mol, "arange": np.arange, "linspace": np.linspace,
             "dict": dict, "True": True, "False": False, "None": None}
    return dict(eval(f"dict({text})", scope))  # the reference evaluates the same expression (with full builtins)


def update_args(kwargs, source):
    for kw in kwargs:
        if kw in source:
            kwargs[kw] = source[kw]
    return kwargs


def teacher_from(name, species=None, device=0):
    """The `calculator` keyword: None, 'PAIR' (workloads.PairTeacher, the stand-in teacher of the examples) or a Python
    script that defines `calc` — the reference's teacher scripts define the same name (theforce/calculator/vasp.py, …)."""
    if name is None or not isinstance(name, str):
        return name
    if name.upper() == "PAIR":
        from ..workloads import PairTeacher
        return PairTeacher(species, device=device)
    if name.endswith(".py"):
        scope = {}
        exec(compile(open(name).read(), name, "exec"), scope)
        if "calc" not in scope:
            raise RuntimeError(f"{name} does not define `calc`")
        return scope["calc"]
    raise RuntimeError(f"calculator {name.upper()} is not implemented")


def gen_active_calc(args=None, species=None, **over):
    """theforce/cl/__init__.py:69-73: an ActiveCalculator from the ARGS keywords it knows."""
    args = read_args() if args is None else args
    kwargs = get_default_args(ActiveCalculator.__init__)
    kwargs.pop("engine", None)
    update_args(kwargs, args)
    update_args(kwargs, over)
    if isinstance(kwargs.get("covariance"), str) and os.path.isfile(kwargs["covariance"]):
        from ..modelio import load_model
        kwargs.setdefault("pckl", None)
        if kwargs["pckl"] is None:
            kwargs["pckl"] = kwargs["covariance"]      # ("covariance = 'pckl'": the model is kept where it was read)
        kwargs["covariance"] = load_model(kwargs["covariance"])
    kwargs["calculator"] = teacher_from(kwargs.get("calculator"), species)
    return ActiveCalculator(**kwargs)
