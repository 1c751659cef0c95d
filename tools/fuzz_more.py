"""tests/test_hip_fuzz.py::test_random_system_matches_oracle over more seeds than the suite runs (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import test_hip_fuzz as t
lo, hi = (int(a) for a in (sys.argv[1:3] or (16, 80)))
bad = 0
for seed in range(lo, hi):
    try:
        t.test_random_system_matches_oracle(seed)
    except Exception as e:  # report and keep going
        bad += 1
        print("seed", seed, "FAILED:", type(e).__name__, str(e)[:200])
print(f"{hi - lo - bad} of {hi - lo} seeds passed")
