"""Model persistence for the MI355X-native path: one .npz per model (kernel hyper-parameters,
species table, inducing LCEs as a ragged CSR, mu / choli / mean / vscale, and — for a learning
model — the labelled data frames).  Plays the role of the reference's `model.pckl/` folder
(regression/gppotential.py:1060-1119) — a torch pickle of Python objects cannot be loaded without
the reference, so the format is our own.  The design matrix K is not stored (3 ΣN x m doubles):
it is rebuilt on the device at load time (`PosteriorPotentialFromFolder(update_data=True)`,
gppotential.py:1342-1380)."""
import numpy as np

from .model import Local, SGPRModel
from .posterior import Frame, PosteriorPotential


def _ragged(arrays, width=None, dtype=float):
    ptr = np.concatenate([[0], np.cumsum([len(a) for a in arrays])]).astype(np.int64)
    shape = (0,) if width is None else (0, width)
    flat = np.concatenate([np.asarray(a, dtype).reshape((-1,) + shape[1:]) for a in arrays] + [np.zeros(shape, dtype)])
    return ptr, flat


def save_model(path, model):
    post = model if isinstance(model, PosteriorPotential) else None
    eng = model.engine if post is not None else model
    X = eng.X
    ptr, nbr_z = _ragged([x._b for x in X], dtype=np.int32)
    _, nbr_r = _ragged([x._r for x in X], width=3)
    mean = post.mean.weights if post is not None else eng.mean
    extra = {}
    if post is not None:
        data = post.data
        dptr, dz = _ragged([fr.numbers for fr in data], dtype=np.int32)
        extra = dict(
            data_ptr=dptr, data_z=dz,
            data_pos=_ragged([fr.positions for fr in data], width=3)[1],
            data_forces=_ragged([fr.forces for fr in data], width=3)[1],
            data_cell=np.array([fr.cell for fr in data]).reshape(-1, 3, 3),
            data_pbc=np.array([fr.pbc for fr in data], bool).reshape(-1, 3),
            data_energy=np.array([fr.energy for fr in data], float),
            # frames labelled without stress are stored as NaN rows and come back as stress=None
            data_stress=np.array([fr.stress if fr.stress is not None else np.full(6, np.nan) for fr in data],
                                 float).reshape(-1, 6),
            noise_logit=post._noise["all"],
        )
    with open(path, "wb") as f:  # np.savez would append ".npz" to a bare path
        np.savez(  # uncompressed: a 16384-atom model with data frames is rewritten after every update
            f,
            format="autoforce_amd.sgpr.v2",
            lmax=eng.lmax, nmax=eng.nmax, exponent=eng.exponent, cutoff=eng.cutoff,
            species=np.array(eng.species, np.int32), radii=np.asarray(eng.radii, float),
            ind_z=np.array([x.number for x in X], np.int32), ind_ptr=ptr, ind_nbr_z=nbr_z, ind_nbr_r=nbr_r,
            mu=np.zeros(0) if eng.mu is None else eng.mu,
            choli=np.zeros((0, 0)) if eng.choli is None else eng.choli,
            mean_z=np.array(sorted(mean), np.int32),
            mean_w=np.array([mean[z] for z in sorted(mean)], float),
            vscale_z=np.array(sorted(eng._vscale), np.int32),
            vscale=np.array([eng._vscale[z] for z in sorted(eng._vscale)], float),
            ridge=eng.ridge, lone_weight=int(getattr(eng, "lone_weight", 1)), **extra,
        )


def load_model(path, device=0, engine=None):
    """Returns a PosteriorPotential (its `.engine` is the SGPRModel).  `engine`: test hook — an
    empty engine object to fill instead of a new SGPRModel."""
    g = np.load(path, allow_pickle=False)
    if str(g["format"]) not in ("autoforce_amd.sgpr.v1", "autoforce_amd.sgpr.v2"):
        raise ValueError(f"{path}: not an autoforce_amd model file")
    mdl = engine or SGPRModel(int(g["lmax"]), int(g["nmax"]), float(g["exponent"]), float(g["cutoff"]),
                              species=g["species"].tolist(), radii=g["radii"], device=device,
                              lone_weight=int(g["lone_weight"]) if "lone_weight" in g else 1)
    ptr = g["ind_ptr"]
    X = [Local(int(z), g["ind_nbr_z"][ptr[q]:ptr[q + 1]], g["ind_nbr_r"][ptr[q]:ptr[q + 1]])
         for q, z in enumerate(g["ind_z"])]
    mean = dict(zip(g["mean_z"].tolist(), g["mean_w"].tolist()))
    if X:
        mdl.set_inducing(X)
        if g["mu"].size:
            mdl.set_weights(g["mu"], mean=mean, vscale=dict(zip(g["vscale_z"].tolist(), g["vscale"].tolist())),
                            choli=g["choli"] if g["choli"].size else None)
    mdl.ridge = float(g["ridge"])
    post = PosteriorPotential(mdl)
    post.mean.weights.update({int(z): float(w) for z, w in mean.items()})
    if "data_ptr" in g:
        dp = g["data_ptr"]
        for k in range(len(dp) - 1):
            a, b = int(dp[k]), int(dp[k + 1])
            post.data.append(Frame(g["data_z"][a:b], g["data_pos"][a:b], g["data_cell"][k], g["data_pbc"][k],
                                   g["data_energy"][k], g["data_forces"][a:b],
                                   None if np.isnan(g["data_stress"][k]).any() else g["data_stress"][k]))
        post._noise["all"] = float(g["noise_logit"])
        if X and post.data:
            post.restore_rows()
            if g["mu"].size:
                post.make_stats()
        elif post.data and post.resident:
            post.restore_rows()
    return post
