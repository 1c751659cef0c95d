"""GPU: BASELINE configs[4] as worded — on-the-fly inducing-set updates AT the size limit inside NVT MD on the
16384-atom 4-species oxide with max_inducing = 1024 (examples/md_nvt_config5.py).  The model starts pre-seeded at
m = 1012 on an equilibrated trajectory; the run must reach the limit and downsize (gppotential.py:829-832, lii) at least three times; afterwards

  * the edited model — K_mm, its per-block factor, the kept QR that followed every append / pop / selection — is
    compared with the same model set up and factored from scratch on the device;
  * energy, forces and covloss of the last frame are compared with the CPU oracle on the model's final state;
  * the thermostat must hold: mean temperature of the last 100 steps within 10 % of 600 K.
"""
import importlib.util
import os
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _example():
    spec = importlib.util.spec_from_file_location("md_nvt_config5", os.path.join(ROOT, "examples", "md_nvt_config5.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_config5_downsizes_inside_nvt_md():
    from oracle import oracle as orc
    ex = _example()
    t0 = time.time()
    res = ex.run(steps=220, m_seed=1012, stop_after_downsizes=3, min_steps=160)
    rows, stats = res["rows"], res["stats"]
    calc = res["calc"]
    assert stats["downsizes"] >= 3, (stats, calc.size)
    assert calc.size[1] <= 1024
    assert max(r["size"][1] for r in rows) >= 1024
    # the downsizes were followed incrementally, not by a rebuild
    assert all("selected through the kept reflectors" in r for r in stats["routes"]), stats["routes"]
    chk = ex.verify(res)
    assert chk["ridge"][0] == chk["ridge"][1]
    # thermostat
    T = np.array([r["T"] for r in rows])
    assert len(T) > 120
    assert abs(T[-100:].mean() - 600.0) <= 60.0, (T[-100:].mean(), T[-100:].min(), T[-100:].max())
    # the last frame against the oracle on the final model
    numbers, pos, cell, pbc = res["system"]
    eng = calc.model.engine
    out = eng.predict(numbers, pos, cell, pbc)
    X = eng.X
    species = np.array(eng.species, np.int32)
    ind_z = np.array([x.number for x in X], np.int32)
    ind_ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])])
    Pm, nnm = orc.inducing_descriptors(3, 3, 6.0, species, ind_z, ind_ptr, np.concatenate([x._b for x in X]),
                                       np.concatenate([x._r for x in X]))
    nl = orc.neighbors_cells(pos, cell, pbc, 6.0)
    ref = orc.frame(3, 3, 6.0, 4.0, species, numbers, pos, cell, nl, ind_z, nnm, Pm, eng.mu, choli=eng.choli, want_p=False)
    e_mean = sum(calc.model.mean.weights.get(int(z), 0.0) for z in numbers)
    fmax = np.abs(ref["forces"]).max()
    assert abs(out["energy"] - e_mean - ref["energy"]) <= 1e-8 * max(1.0, abs(ref["energy"]))
    assert np.abs(out["forces"] - ref["forces"]).max() <= 1e-7 * fmax
    vs = np.array([eng._vscale[int(z)] for z in numbers])
    assert np.abs(out["beta"] ** 2 - ref["beta"] ** 2 * vs).max() <= 1e-7 * max(1.0, (ref["beta"] ** 2 * vs).max())
    upd = [1e3 * r["wall"] - r["teacher_ms"] for r in rows[1:] if r["updated"] and r["size"][1] >= 1024]
    print(f"config 5: {len(rows) - 1} steps, {stats['downsizes']} downsizes (median {np.median(stats['downsize_ms']):.1f} ms), "
          f"update steps at the limit: median {np.median(upd):.1f} ms; total {time.time() - t0:.1f} s; {chk}")
    res["teacher"].close()
