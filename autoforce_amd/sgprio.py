"""`.sgpr` tape — the reference's append-only text log of accepted data and inducing LCEs
(theforce/io/sgprio.py:16-143), readable and writable without the reference or ASE.

Blocks:
    start: local          Z of the central atom, then one line per neighbour "Z x y z"
    end: local            (sgprio.py:16-39; written with {:4d} / {:16.8f})
    start: atoms          one extended-XYZ frame as ASE writes it: Lattice, Properties, energy,
    end: atoms            stress (3x3 row-major), pbc; per-atom columns species, pos[, forces]
    start: params         "key value" lines
    end: params
    include: <path>       splice another tape (relative to this file); recursion is cut
"""
import os
import re
from collections import Counter

import numpy as np

from .model import Local
from .posterior import Frame

SYMBOLS = ("X H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr "
           "Rb Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu "
           "Hf Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg "
           "Bh Hs Mt Ds Rg Cn Nh Fl Mc Lv Ts Og").split()
NUMBER = {s: z for z, s in enumerate(SYMBOLS)}
VOIGT = ((0, 0), (1, 1), (2, 2), (1, 2), (0, 2), (0, 1))


# ---------------------------------------------------------------------------- local blocks
def format_lce(loc):
    lines = [f"{loc.number:4d}\n"]
    for z, r in zip(loc._b, loc._r):
        lines.append("{:4d} {:16.8f} {:16.8f} {:16.8f}\n".format(int(z), *[float(c) for c in r]))
    return lines


def parse_lce(blk):
    z = int(blk[0].strip())
    rows = [ln.split() for ln in blk[1:] if ln.strip()]
    b = np.array([int(r[0]) for r in rows], np.int32)
    r = np.array([[float(c) for c in r[1:4]] for r in rows], float).reshape(-1, 3)
    return Local(z, b, r)


# ---------------------------------------------------------------------------- extended XYZ frames
_KV = re.compile(r'(\w+)=("([^"]*)"|\S+)')


def _bools(txt):
    return [t.upper().startswith("T") for t in txt.split()]


def parse_extxyz(blk):
    lines = [ln for ln in blk if ln.strip()]
    n = int(lines[0].split()[0])
    info = {m.group(1): (m.group(3) if m.group(3) is not None else m.group(2)) for m in _KV.finditer(lines[1])}
    props = info.get("Properties", "species:S:1:pos:R:3").split(":")
    cols, c = {}, 0
    for name, kind, width in zip(props[0::3], props[1::3], props[2::3]):
        cols[name] = (c, c + int(width), kind)
        c += int(width)
    rows = [ln.split() for ln in lines[2:2 + n]]
    if len(rows) != n:
        raise ValueError(f"extxyz block: {n} atoms announced, {len(rows)} lines found")

    def column(name):
        a, b, _ = cols[name]
        return [r[a:b] for r in rows]

    if "Z" in cols:
        numbers = [int(v[0]) for v in column("Z")]
    else:
        numbers = [NUMBER[v[0]] for v in column("species")]
    pos = np.array(column("pos"), float)
    forces = None
    for key in ("forces", "force"):
        if key in cols:
            forces = np.array(column(key), float)
    cell = np.array(info["Lattice"].split(), float).reshape(3, 3) if "Lattice" in info else np.zeros((3, 3))
    pbc = _bools(info["pbc"]) if "pbc" in info else [("Lattice" in info)] * 3
    energy = float(info["energy"]) if "energy" in info else (float(info["free_energy"]) if "free_energy" in info else None)
    stress = None
    if "stress" in info:
        s = np.array(info["stress"].split(), float)
        stress = s if s.size == 6 else np.array([s.reshape(3, 3)[i, j] for i, j in VOIGT])
    return Frame(numbers, pos, cell, pbc, energy, forces, stress)


def format_extxyz(fr):
    s = np.zeros((3, 3))
    head = ['Lattice="{}"'.format(" ".join(repr(float(v)) for v in fr.cell.reshape(-1)))]
    props = "species:S:1:pos:R:3" + (":forces:R:3" if fr.forces is not None else "")
    head.append(f"Properties={props}")
    if fr.energy is not None:
        head.append(f"energy={fr.energy!r}")
    if fr.stress is not None:
        for v, (i, j) in zip(fr.stress, VOIGT):
            s[i, j] = s[j, i] = v
        head.append('stress="{}"'.format(" ".join(repr(float(v)) for v in s.reshape(-1))))
    head.append('pbc="{}"'.format(" ".join("T" if b else "F" for b in fr.pbc)))
    lines = [f"{fr.natoms}\n", " ".join(head) + "\n"]
    # one %-format per row over plain Python floats (the same text as f"{v:22.15e}" per value, a quarter of the time:
    # a 16384-atom frame goes to the tape after every accepted teacher call)
    cols = np.asarray(fr.positions, float).reshape(fr.natoms, 3)
    if fr.forces is not None:
        cols = np.hstack([cols, np.asarray(fr.forces, float).reshape(fr.natoms, 3)])
    fmt = "%-2s " + " ".join(["%22.15e"] * cols.shape[1]) + "\n"
    syms = [SYMBOLS[z] for z in np.asarray(fr.numbers, int).tolist()]
    lines.extend(fmt % (sym, *row) for sym, row in zip(syms, cols.tolist()))
    return lines


def convert_block(typ, blk):
    if typ == "atoms":
        return parse_extxyz(blk)
    if typ == "local":
        return parse_lce(blk)
    if typ == "params":
        out = {}
        for ln in blk:
            if ln.strip():
                key, val = ln.split(None, 1)
                try:
                    out[key] = float(val) if re.fullmatch(r"[-+0-9.eE]+|inf|nan", val.strip()) else val.strip()
                except ValueError:
                    out[key] = val.strip()
        return out  # the reference eval()s the value (sgprio.py:51-54); a tape is data, not code
    raise RuntimeError(f"type {typ} is unknown")


class SgprIO:
    def __init__(self, path, rank=0):
        self.path = os.path.abspath(os.path.expanduser(path))
        self.rank = rank

    def _append(self, typ, lines):
        if self.rank == 0:
            with open(self.path, "a") as f:
                f.write(f"\nstart: {typ}\n")
                f.writelines(lines)
                f.write(f"end: {typ}\n")

    def write(self, obj):
        if isinstance(obj, Local):
            self._append("local", format_lce(obj))
        elif isinstance(obj, Frame):
            self._append("atoms", format_extxyz(obj))
        else:
            raise RuntimeError(f"no recipe for {type(obj)}")

    def write_params(self, **kw):
        self._append("params", [f"{a} {b}\n" for a, b in kw.items()])

    def read(self, exclude=None, verbose=False):
        """List of (type, object) in tape order, includes spliced in place (sgprio.py:92-143)."""
        if not os.path.isfile(self.path):
            return []
        if exclude is None:
            exclude = []
        elif isinstance(exclude, str):
            exclude = [os.path.abspath(exclude)]
        elif isinstance(exclude, SgprIO):
            exclude = [exclude.path]
        if self.path in exclude:
            return []
        exclude.append(self.path)
        with open(self.path) as f:
            lines = f.readlines()
        data, on, typ, blk, c = [], False, None, [], Counter()
        for line in lines:
            if not on:
                if line.startswith("start:"):
                    on, typ, blk = True, line.split()[-1], []
                elif line.startswith("include:"):
                    inc = os.path.expanduser(os.path.expandvars(line.split()[-1]))
                    if not os.path.isabs(inc):
                        inc = os.path.join(os.path.dirname(self.path), inc)
                    data.extend(SgprIO(inc).read(exclude=exclude))
            elif line.startswith("end:"):
                if line.split()[-1] != typ:
                    raise ValueError(f"{self.path}: 'end: {line.split()[-1]}' closes 'start: {typ}'")
                on = False
                data.append((typ, convert_block(typ, blk)))
                c[typ] += 1
            else:
                blk.append(line)
        if verbose and self.rank == 0:
            print(f"included {self.path} {dict(c)}")
        return data
