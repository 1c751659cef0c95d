"""CPU-side checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/sgpr_hip.h declares, and fails loudly (no CPU fallback) when there is no GPU."""
import os
import re

import numpy as np
import pytest

from conftest import ROOT


def header_symbols():
    text = open(os.path.join(ROOT, "include", "sgpr_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sgpr_[a-z_0-9]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from autoforce_amd import _lib
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 20
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/sgpr_hip.h but not exported"
        assert n in _lib.SIGNATURES, f"{n} has no ctypes signature in autoforce_amd/_lib.py"
    assert set(_lib.SIGNATURES) == set(names)


def test_version_and_packed_len():
    from autoforce_amd import _lib
    lib = _lib.load()
    assert lib.sgpr_version() >= 1000
    assert lib.sgpr_packed_len(4096) == 4 * 4096 + 11


def test_stress_from_virial_host_helper():
    """calculator/active.py:574,604-610: Voigt picks [0,4,8,5,2,1] of virial/volume; volume -2
    for a rank-deficient cell."""
    from autoforce_amd import _lib
    lib = _lib.load()
    v = np.arange(1.0, 10.0)
    cell = np.diag([2.0, 3.0, 4.0])
    s = np.zeros(6)
    assert lib.sgpr_stress_from_virial(_lib.ptr(v), _lib.ptr(cell), _lib.ptr(s)) == 0
    np.testing.assert_allclose(s, v[[0, 4, 8, 5, 2, 1]] / 24.0)
    assert lib.sgpr_stress_from_virial(_lib.ptr(v), _lib.ptr(np.zeros((3, 3))), _lib.ptr(s)) == 0
    np.testing.assert_allclose(s, v[[0, 4, 8, 5, 2, 1]] / -2.0)


def test_no_cpu_fallback():
    """Without a GPU the product path must fail loudly, not compute on the host."""
    from autoforce_amd import SGPRModel, SgprError, device_count
    if device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(SgprError) as e:
        SGPRModel(species=[14])
    assert e.value.code == -2


def test_product_never_imports_oracle():
    """The oracle is test infrastructure: nothing under autoforce_amd/ may import, link or load it."""
    pkg = os.path.join(ROOT, "autoforce_amd")
    bad = re.compile(r"^\s*(from|import)\s+oracle\b|liborc|sgpr_oracle|orc_[a-z_]+\s*\(", re.M)
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".inc", ".cpp", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert not bad.search(src), f
