"""BCMActiveCalculator — Bayesian committee of SGPR models (theforce/calculator/active_bcm.py).

A committee is a set of frozen sub-models plus one live model that keeps learning.  Every member
is evaluated on the device; the members are combined with the reference's weights
(active_bcm.py:589-633):

    covmax_k = max_i covloss_k(i),  beta_k = -ln(covmax_k) if covmax_k < 1 else 0,
    scale_k  = beta_k / covmax_k,   E = sum_k scale_k E_k / sum_k scale_k      (same for F, stress)

and the uncertainty that drives sampling is the member-wise minimum of the covlosses
(get_covloss_total, :885-894).  `initiate_bcm()` (:340-369) freezes the live model as a new member
and starts an empty one with its own model file and tape.
"""
import os

import numpy as np

from .calculator import ActiveCalculator, inf
from .posterior import PosteriorPotential


class BCMActiveCalculator(ActiveCalculator):
    def __init__(self, covariance=None, kernel_model_dict=None, pckl=None, tape=None, member_engine=None, **kw):
        """kernel_model_dict: {key: SGPRModel | PosteriorPotential | path of a saved model}: the frozen
        members.  pckl / tape: *heads*; the live model uses `<head>_<id>.npz` / `<head>_<id>.sgpr`
        (active_bcm.py:263-301).  member_engine: test hook — callable returning an empty engine for members
        loaded from disk."""
        self.model_dict = {}
        fresh = (lambda: member_engine()) if member_engine else (lambda: None)
        for key, mdl in (kernel_model_dict or {}).items():
            if isinstance(mdl, str):
                from .modelio import load_model
                mdl = load_model(mdl, engine=fresh())
            self.model_dict[key] = mdl if isinstance(mdl, PosteriorPotential) else PosteriorPotential(mdl)
        self.pckl_head = None if pckl is None else (pckl[:-4] if pckl.endswith(".npz") else pckl)
        self.tape_head = None if tape is None else (tape[:-5] if tape.endswith(".sgpr") else tape)
        self.pckl_id = 1
        while self.pckl_head and os.path.isfile(self._pckl_name(self.pckl_id + 1)):
            self.pckl_id += 1  # restart: earlier members are on disk (active_bcm.py:269-291)
        if self.pckl_head:
            from .modelio import load_model
            for k in range(1, self.pckl_id):
                self.model_dict[self._pckl_name(k)[:-4]] = load_model(self._pckl_name(k), engine=fresh())
        self._member_beta = {}
        self._ctor_kw = dict(kw)
        super().__init__(covariance=covariance, pckl=self._pckl_name(self.pckl_id) if self.pckl_head else None,
                         tape=self._tape_name(self.pckl_id) if self.tape_head else None, **kw)

    def _pckl_name(self, k):
        return f"{self.pckl_head}_{k}.npz"

    def _tape_name(self, k):
        return f"{self.tape_head}_{k}.sgpr"

    # ------------------------------------------------------------------ committee bookkeeping
    def initiate_bcm(self):
        """active_bcm.py:340-369: the live model becomes a frozen member; a new empty one takes over."""
        from .sgprio import SgprIO
        key = self.pckl[:-4] if self.pckl else f"member_{len(self.model_dict) + 1}"
        self.save_model()
        self.model_dict[key] = self.model
        self.pckl_id += 1
        if self.pckl_head:
            self.pckl = self._pckl_name(self.pckl_id)
        if self.tape_head:
            self.tape = SgprIO(self._tape_name(self.pckl_id), rank=self.rank)
        fresh = self.engine.scratch()
        self.get_model(fresh, {})
        self.log_settings()
        self.log("model size: {} {}".format(*self.size))

    def initiate_model(self):
        super().initiate_model()
        self.save_model()  # active_bcm.py:683

    def _needs_seed(self):
        return self.active and self.model.ndata == 0  # active_bcm.py:505-508: any step, e.g. after initiate_bcm

    # ------------------------------------------------------------------ combined prediction
    @staticmethod
    def _scale(beta):
        covmax = float(np.max(beta)) if len(beta) else inf
        b = -np.log(covmax) if covmax < 1.0 else 0.0
        return (b / covmax if covmax > 0.0 else inf), covmax

    def update_results(self, retain_graph=False, covloss_only=False):
        live = self._evaluate_engine(self.engine)
        self._cov = None
        self._nl = None
        self._beta = live["beta"]
        outs = {key: self._evaluate_engine(post.engine) for key, post in self.model_dict.items()}
        self._member_beta = {key: o["beta"] for key, o in outs.items()}
        if covloss_only:
            return
        members = list(outs.values()) + [live]
        scales = [self._scale(o["beta"])[0] for o in members]
        ready = [o["ready"] for o in members]
        w = np.array([s if r else 0.0 for s, r in zip(scales, ready)], float)
        if np.isinf(w).any():  # a member with zero covloss everywhere is certain: it alone decides
            w = np.isinf(w).astype(float)
        if w.sum() <= 0.0:  # every member is out of its depth: the live model answers (or zeros)
            w = np.zeros(len(members))
            w[-1] = 1.0
        w = w / w.sum()
        self.bcm_weights = dict(zip(list(outs.keys()) + ["live"], w.tolist()))
        self._set_results(dict(energy=sum(a * o["energy"] for a, o in zip(w, members)),
                               forces=sum(a * o["forces"] for a, o in zip(w, members)),
                               stress=sum(a * o["stress"] for a, o in zip(w, members))))

    def get_covloss_total(self):
        """active_bcm.py:885-894."""
        b = self._beta
        for mb in self._member_beta.values():
            b = np.minimum(b, mb)
        return b

    def get_covloss(self):
        # the sampling loop of the base class asks get_covloss(): in a committee that is the total
        return self.get_covloss_total() if self._member_beta else self._beta
