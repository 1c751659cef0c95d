#!/usr/bin/env python3
"""cProfile of examples/md_nvt_config5.py's run (update steps at the size limit): where an update step's wall time goes."""
import cProfile
import importlib.util
import os
import pstats
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("c5", os.path.join(ROOT, "examples", "md_nvt_config5.py"))
c5 = importlib.util.module_from_spec(spec)
spec.loader.exec_module(c5)
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
pr = cProfile.Profile()
res = {}


def go():
    res.update(c5.run(steps=steps, m_seed=1020))


pr.runcall(go)
rows = res["rows"]
upd = [1e3 * r["wall"] - r["teacher_ms"] for r in rows[1:] if r["updated"] and r["size"][1] >= 1024]
import numpy as np
print(f"update steps at the limit: {len(upd)}, median {np.median(upd):.1f} ms; downsizes {res['stats']['downsizes']}")
st = pstats.Stats(pr)
st.sort_stats("cumulative")
st.print_stats(r"(posterior|model|calculator|workloads)\.py", 45)
