"""Free device memory before / after handles that went through more and more of the API (a DevBuf has no destructor:
sgpr_destroy releases by name).  python tools/leak_probe.py on a GPU box."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from autoforce_amd import SGPRModel
from autoforce_amd.workloads import inducing_from_frame, lips, FS, MASS
from autoforce_amd.ase_shim import kB
numbers, pos, cell, pbc = lips(16, seed=0)
species = sorted(set(int(z) for z in numbers))
n2, p2, c2, b2 = lips(16, seed=1)
NAMES = ["create", "predict without a model", "neighbors()", "set_inducing", "solve + weights", "predict", "md"]
def run(stage):
    mdl = SGPRModel(3, 3, 4, 6.0, species=species)
    if stage >= 1:
        mdl.predict(n2, p2, c2, b2, beta=False)
    if stage >= 2:
        mdl.neighbors(len(n2))
    if stage >= 3:
        mdl.set_inducing(inducing_from_frame(mdl, n2, p2, c2, b2, 48, seed=1))
    if stage >= 4:
        rng = np.random.default_rng(2)
        mdl.solve(rng.normal(size=(64, 48)), rng.normal(size=64))
        mdl.set_weights(0.02 * rng.normal(size=48), choli=mdl.choli, vscale=mdl.make_vscale())
    if stage >= 5:
        mdl.predict(numbers, pos, cell, pbc)
    if stage >= 6:
        mass = np.array([MASS[int(z)] for z in numbers])
        mdl.md_begin(numbers, pos, cell, pbc, mass, np.zeros_like(pos), dt=FS, friction=1e-3, kT=kB * 300.0, seed=3)
        mdl.md_run(12, None)
        mdl.md_end()
    mdl.close()
for stage in range(len(NAMES)):
    run(stage); torch.cuda.synchronize()
    f0 = torch.cuda.mem_get_info()[0]
    for _ in range(4): run(stage)
    torch.cuda.synchronize()
    print(f"up to {NAMES[stage]:26s}: {(f0 - torch.cuda.mem_get_info()[0]) / 4 / 2**20:6.2f} MB per handle not given back", flush=True)
