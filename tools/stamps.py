import sys, os
os.environ["SGPR_STAMPS"] = "1"
sys.path.insert(0, "/root/repo")
import numpy as np, bench
from autoforce_amd.workloads import lips
numbers, pos, cell, pbc = lips(16, seed=0)
mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
for _ in range(5): mdl.predict(numbers, pos, cell, pbc)
mdl.close()
