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
    def __init__(self, covariance=None, kernel_model_dict=None, pckl=None, tape=None, member_engine=None,
                 members_over_ranks=False, **kw):
        """kernel_model_dict: {key: SGPRModel | PosteriorPotential | path of a saved model}: the frozen
        members.  pckl / tape: *heads*; the live model uses `<head>_<id>.npz` / `<head>_<id>.sgpr`
        (active_bcm.py:263-301).  member_engine: test hook — callable returning an empty engine for members
        loaded from disk.
        members_over_ranks (with a process_group): ONE MEMBER PER RANK instead of every member sharded over all
        ranks — member k (the live model last) is evaluated, unsharded, by rank k mod world on its own GPU; the
        ranks exchange the members' weights (one tiny all-reduce), their weighted sums (one packed all-reduce of
        [F | E | stress], as the sharded path's) and the member-wise minimum covloss (one MIN all-reduce).  The
        members are independent models (active_bcm.py:589-633 loops over them): a committee of G members on G GPUs
        costs one member's step, where sharding each 4096-atom member eight ways gains 1.4x."""
        self.members_over_ranks = bool(members_over_ranks)
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

    def _whole(self, engine):
        """One UNSHARDED pass of `engine` over the current atoms, whatever the process group says."""
        numbers, positions, cell, pbc = self._system(self.atoms)
        N = len(numbers)
        if not (engine.m > 0 and engine.mu is not None):
            return dict(energy=0.0, forces=np.zeros((N, 3)), stress=np.zeros(6), beta=np.full(N, inf), ready=False)
        out = engine.predict(numbers, positions, cell, pbc, rank=0, world=1, cov=False, beta=True)
        out["ready"] = True
        return out

    def _update_results_over_ranks(self, covloss_only):
        """members_over_ranks: this rank evaluates the members it owns; the committee is combined across ranks."""
        import torch.distributed as dist
        rank, world = self._dist()
        keys = list(self.model_dict.keys())
        engines = [self.model_dict[k].engine for k in keys] + [self.engine]
        K, N = len(engines), len(self.atoms)
        mine = [k for k in range(K) if k % world == rank]
        outs = {k: self._whole(engines[k]) for k in mine}
        # 1. every member's weight (and whether it has a model at all): each rank fills in its own
        sc = np.zeros(2 * K)
        for k, o in outs.items():
            s_k = self._scale(o["beta"])[0]
            sc[k] = 1e300 if np.isinf(s_k) else s_k       # (inf does not survive a SUM of zeros and itself on every backend)
            sc[K + k] = 1.0 if o["ready"] else 0.0
        t = self._tensor(sc)
        dist.all_reduce(t, group=self.process_group)
        sc = t.cpu().numpy()
        scales = np.where(sc[:K] >= 1e300, inf, sc[:K])
        w = np.where(sc[K:] > 0.5, scales, 0.0)
        if np.isinf(w).any():
            w = np.isinf(w).astype(float)
        if w.sum() <= 0.0:
            w = np.zeros(K)
            w[-1] = 1.0
        w = w / w.sum()
        self.bcm_weights = dict(zip(keys + ["live"], w.tolist()))
        # 2. the member-wise minimum of the covlosses (active_bcm.py:885-894)
        b = np.full(N, inf)
        for o in outs.values():
            b = np.minimum(b, o["beta"])
        tb = self._tensor(np.where(np.isinf(b), 1e300, b))
        dist.all_reduce(tb, op=dist.ReduceOp.MIN, group=self.process_group)
        b = tb.cpu().numpy()
        self._beta = np.where(b >= 1e300, inf, b)
        self._member_beta = {}     # (get_covloss() returns _beta: already the total)
        self._cov = None
        self._nl = None
        if covloss_only:
            return
        # 3. the weighted sums, one packed buffer (the same collective as a sharded step's)
        v = np.zeros(3 * N + 7)
        for k, o in outs.items():
            if w[k] != 0.0:
                v[:3 * N] += w[k] * np.asarray(o["forces"], float).reshape(-1)
                v[3 * N] += w[k] * o["energy"]
                v[3 * N + 1:] += w[k] * np.asarray(o["stress"], float)
        tv = self._tensor(v)
        dist.all_reduce(tv, group=self.process_group)
        v = tv.cpu().numpy()
        self._set_results(dict(energy=float(v[3 * N]), forces=v[:3 * N].reshape(N, 3).copy(), stress=v[3 * N + 1:].copy()))

    def local(self, k):
        if self.members_over_ranks and self._dist()[1] > 1:
            # (no rank's sharded list to ask: the frame's list is built here, unsharded, on the live engine)
            return self._locals_of(*self._system(self.atoms), indices=[k])[0]
        return super().local(k)

    def update_results(self, retain_graph=False, covloss_only=False):
        if self.members_over_ranks and self._dist()[1] > 1:
            return self._update_results_over_ranks(covloss_only)
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
