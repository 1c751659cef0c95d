"""Model persistence for the MI355X-native path: one .npz per model (kernel hyper-parameters,
species table, inducing LCEs as a ragged CSR, mu / choli / mean / vscale).  Plays the role of the
reference's `model.pckl/` folder (regression/gppotential.py:1060-1119) — a torch pickle of Python
objects cannot be loaded without the reference, so the format is our own."""
import numpy as np

from .model import Local, SGPRModel


def save_model(path, model):
    X = model.X
    ptr = np.concatenate([[0], np.cumsum([len(x._b) for x in X])]).astype(np.int64)
    np.savez_compressed(
        path,
        format="autoforce_amd.sgpr.v1",
        lmax=model.lmax, nmax=model.nmax, exponent=model.exponent, cutoff=model.cutoff,
        species=np.array(model.species, np.int32), radii=np.asarray(model.radii, float),
        ind_z=np.array([x.number for x in X], np.int32), ind_ptr=ptr,
        ind_nbr_z=np.concatenate([x._b for x in X] + [np.zeros(0, np.int32)]),
        ind_nbr_r=np.concatenate([x._r for x in X] + [np.zeros((0, 3))]),
        mu=np.zeros(0) if model.mu is None else model.mu,
        choli=np.zeros((0, 0)) if model.choli is None else model.choli,
        mean_z=np.array(sorted(model.mean), np.int32),
        mean_w=np.array([model.mean[z] for z in sorted(model.mean)], float),
        vscale_z=np.array(sorted(model._vscale), np.int32),
        vscale=np.array([model._vscale[z] for z in sorted(model._vscale)], float),
        ridge=model.ridge,
    )


def load_model(path, device=0):
    g = np.load(path, allow_pickle=False)
    if str(g["format"]) != "autoforce_amd.sgpr.v1":
        raise ValueError(f"{path}: not an autoforce_amd model file")
    mdl = SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["exponent"]), float(g["cutoff"]),
                    species=g["species"].tolist(), radii=g["radii"], device=device)
    ptr = g["ind_ptr"]
    X = [Local(int(z), g["ind_nbr_z"][ptr[q]:ptr[q + 1]], g["ind_nbr_r"][ptr[q]:ptr[q + 1]])
         for q, z in enumerate(g["ind_z"])]
    if X:
        mdl.set_inducing(X)
        if g["mu"].size:
            mdl.set_weights(g["mu"], mean=dict(zip(g["mean_z"].tolist(), g["mean_w"].tolist())),
                            vscale=dict(zip(g["vscale_z"].tolist(), g["vscale"].tolist())),
                            choli=g["choli"] if g["choli"].size else None)
    mdl.ridge = float(g["ridge"])
    return mdl
