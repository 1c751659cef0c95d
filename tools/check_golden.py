#!/usr/bin/env python3
"""Every golden frame through the HIP path, errors against the reference values printed per frame
(each frame in its own process: a device fault in one does not hide the others).
usage: python3 tools/check_golden.py [frame ...]"""
import faulthandler, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def one(name):
    import numpy as np
    from autoforce_amd import Local, SGPRModel
    faulthandler.enable()
    g = np.load(os.path.join(ROOT, "tests", "golden", name + ".npz"))
    mdl = SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["eta"]), float(g["rc"]), species=g["species"].tolist())
    ptr = g["ind_ptr"]
    X = [Local(int(z), g["ind_nbr_z"][ptr[q]:ptr[q + 1]], g["ind_nbr_r"][ptr[q]:ptr[q + 1]]) for q, z in enumerate(g["ind_z"])]
    mdl.set_inducing(X)
    vs = dict(zip(g["vscale_z"].tolist(), g["vscale"].tolist())) if "vscale_z" in g else None
    mdl.set_weights(g["mu"], vscale=vs, choli=g["choli"])
    out = mdl.predict(g["numbers"], g["positions"], g["cell"], g["pbc"])
    fmax = np.abs(g["forces"]).max()
    msg = f"{name:22s} N={len(g['numbers']):4d} dE={abs(out['energy'] - g['energy']):.2e} dF/Fmax={np.abs(out['forces'] - g['forces']).max() / fmax:.2e}"
    if "stress" in g:
        msg += f" dS/Smax={np.abs(out['stress'] - g['stress']).max() / np.abs(g['stress']).max():.2e}"
    if "covloss" in g and out.get("beta") is not None:
        msg += f" dbeta={np.abs(out['beta'] - g['covloss']).max():.2e}"
    print(msg, flush=True)
    mdl.close()


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "--one":
        one(sys.argv[2])
        sys.exit(0)
    names = sys.argv[1:] or sorted(os.path.basename(f)[:-4] for f in glob.glob(os.path.join(ROOT, "tests", "golden", "g5_*.npz")))
    for n in names:
        r = subprocess.run([sys.executable, __file__, "--one", n], capture_output=True, text=True, timeout=300)
        sys.stdout.write(r.stdout)
        if r.returncode != 0:
            print(f"{n}: rc={r.returncode}\n" + "\n".join((r.stderr or "").splitlines()[:12]), flush=True)
