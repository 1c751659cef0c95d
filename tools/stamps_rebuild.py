"""Phase stamps of nl_fwd on steps that REBUILD the candidate lists (skin = 0: every step does): sweep | sort + lists out |
filter | c | spectrum.  SGPR_HIP_LIB=autoforce_amd/libsgpr_hip_stamps.so SGPR_STAMPS=1 python tools/stamps_rebuild.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from autoforce_amd import _lib
from autoforce_amd.workloads import lips
numbers, pos, cell, pbc = lips(16, seed=0)
mdl = bench.build_model(0, numbers, pos, cell, pbc, 512)
_lib.check(_lib.load().sgpr_set_option(mdl.handle, b"skin_milliangstrom", 0))
for _ in range(6):
    mdl.predict(numbers, pos + 0.0, cell, pbc)
mdl.close()
