"""Phase stamps (s_memtime per wave) of the two descriptor kernels for a rank's share of the frame: world 1 and world 8.
Needs the -DSGPR_PHASE_STAMPS build: SGPR_HIP_LIB=autoforce_amd/libsgpr_hip_stamps.so SGPR_STAMPS=1 python tools/stamps_share.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from autoforce_amd import _lib
from autoforce_amd.workloads import lips
numbers, pos, cell, pbc = lips(16, seed=0)
N = len(numbers)
for world in (1, 8):
    print(f"== world {world}: rank 0 holds {(N + world - 1) // world} atoms", file=sys.stderr, flush=True)
    mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
    for _ in range(6):
        mdl.predict(numbers, pos + 0.0, cell, pbc, rank=0, world=world)
    mdl.close()
