import sys, pathlib, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import active_common as ac
from helpers import OracleModel
from autoforce_amd import SGPRModel
for tag, eng in (("hip", SGPRModel(3, 3, 4, 4.5, species=ac.SPECIES)), ("cpu", OracleModel(3, 3, 4, 4.5, species=ac.SPECIES))):
    d = pathlib.Path("/tmp/twin_" + tag); d.mkdir(exist_ok=True)
    c, t, tr = ac.run(eng, d, steps=5)
    print(tag, [x[0] for x in tr])
import subprocess
a = [l[20:] for l in open("/tmp/twin_hip/active.log")]
b = [l[20:] for l in open("/tmp/twin_cpu/active.log")]
for k, (x, y) in enumerate(zip(a, b)):
    if x != y:
        print(k, "HIP:", x.strip()[:150]); print(k, "CPU:", y.strip()[:150])
