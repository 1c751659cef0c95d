"""ActiveCalculator — the reference's ASE-calculator surface for the SGPR predict hot path,
backed by libsgpr_hip (MI355X).  Mirrors theforce/calculator/active.py:104-135 (constructor
keywords), :425-535 (calculate / post_calculate), :548-611 (results), :770-804 (gather, covloss).

Scope: prediction (calculator=None in the reference's terms): energy, forces,
stress, covloss, per-step log line, sharding over a torch.distributed process group.  The
on-the-fly training loop (teacher calls, data/inducing acceptance tests) is a "next" row and
raises NotImplementedError when a teacher `calculator` is passed.

Works with real ASE when installed (subclasses ase.calculators.calculator.Calculator); otherwise
with the minimal shim in autoforce_amd.ase_shim.
"""
import datetime
import time

import numpy as np

try:  # pragma: no cover - ASE is absent from the build image
    from ase.calculators.calculator import Calculator, all_changes
    from ase.units import kcal, mol
    kcal_mol = kcal / mol
    HAVE_ASE = True
except ImportError:
    from .ase_shim import Calculator, all_changes, kcal_mol
    HAVE_ASE = False

from .model import SGPRModel
from .sharding import pack_partial, unpack_total

inf = float("inf")


def default_kernel(lmax=3, nmax=3, exponent=4, cutoff=6.0, species=None, device=0):
    """theforce/calculator/active.py:28-38: SeSoapKernel(lmax,nmax,exponent,cutoff,
    radii=DefaultRadii()).  `species` is the table of atomic numbers the model may meet (the
    reference's wildcard kernel indexes a fixed 120-wide table; the device layout is dense)."""
    return SGPRModel(lmax, nmax, exponent, cutoff, species=species, device=device)


class ActiveCalculator(Calculator):
    implemented_properties = ["energy", "forces", "stress", "free_energy"]

    def __init__(self, covariance=None, calculator=None, process_group=None, meta=None, logfile="active.log",
                 pckl=None, tape=None, test=None, stdout=False, ediff=2 * kcal_mol, ediff_lb=None, ediff_ub=None,
                 ediff_tot=4 * kcal_mol, fdiff=3 * kcal_mol, noise_f=kcal_mol, ioptim=1, max_data=inf,
                 max_inducing=inf, kernel_kw=None, veto=None, include_params=None, eps_dr=0.1, ignore=None,
                 report_timings=False, step0_forced_fp=False, nbeads=1, engine=None):
        """
        covariance:    SGPRModel | path to a model .npz (autoforce_amd.modelio) | None (+ kernel_kw
                       with a `species` entry -> empty default kernel)
        calculator:    must be None (prediction only; the active-learning loop is not built yet)
        process_group: None | torch.distributed process group over which atoms are sharded
        engine:        test hook — any object with predict(numbers, positions, cell, pbc, rank, world,
                       cov, beta) and attributes m, species; defaults to the HIP-backed model
        The remaining keywords are accepted for signature compatibility with the reference.
        """
        Calculator.__init__(self)
        if calculator is not None:
            raise NotImplementedError(
                "on-the-fly learning with a teacher calculator is outside this build's scope (SURVEY.md §8f); "
                "pass calculator=None and a trained model")
        self._calc = None
        self.process_group = process_group
        if engine is not None:
            self.model = engine
        elif isinstance(covariance, SGPRModel):
            self.model = covariance
        elif isinstance(covariance, str):
            from .modelio import load_model
            self.model = load_model(covariance)
        else:
            kw = dict(kernel_kw or {})
            if not kw.get("species"):
                raise ValueError("kernel_kw={'species': [...]} is required to build an empty model")
            self.model = default_kernel(**kw)
        self.ediff = ediff
        self.ediff_lb = ediff_lb or ediff
        self.ediff_ub = ediff_ub or ediff
        self.ediff_tot, self.fdiff, self.noise_f = ediff_tot, fdiff, noise_f
        self.logfile, self.stdout = logfile, stdout
        self.report_timings = report_timings
        self.step = 0
        self.deltas = None
        self.updated = False
        self.covlog = ""
        self.cov = None
        self.maximum_force = inf
        self.meta = meta
        self.log("active calculator says Hello!", mode="w")
        self.log("model size: {} {}".format(*self.size))

    # ------------------------------------------------------------------ properties of the surface
    @property
    def active(self):
        return self._calc is not None

    @property
    def size(self):
        return 0, self.model.m  # (n_data, n_inducing), calculator/active.py:375-376

    @property
    def rank(self):
        return self._dist()[0]

    @property
    def world_size(self):
        return self._dist()[1]

    def _dist(self):
        if self.process_group is None:
            return 0, 1
        import torch.distributed as dist
        return dist.get_rank(self.process_group), dist.get_world_size(self.process_group)

    # ------------------------------------------------------------------ the hot path
    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        timings = [time.time()]
        if self.size[1] == 0 and not self.active:
            raise RuntimeError("you forgot to assign a DFT calculator!")  # calculator/active.py:429-430
        Calculator.calculate(self, atoms, properties, system_changes)
        a = self.atoms
        numbers = np.asarray(a.numbers, dtype=np.int32)
        positions = np.asarray(a.positions, dtype=float)
        cell = np.asarray(getattr(a.cell, "array", a.cell), dtype=float).reshape(3, 3)
        pbc = np.asarray(a.pbc, dtype=bool)
        N = len(numbers)
        rank, world = self._dist()
        out = self.model.predict(numbers, positions, cell, pbc, rank=rank, world=world, cov=True, beta=True)
        timings.append(time.time())
        self.cov = out["cov"]
        if world > 1:
            import torch
            import torch.distributed as dist
            v = torch.from_numpy(pack_partial(out, N))
            backend = dist.get_backend(self.process_group)
            if backend == "nccl":
                v = v.cuda()
            dist.all_reduce(v, group=self.process_group)  # active.py:562,601,602,777 in one collective
            out = unpack_total(v.cpu().numpy(), N)
        self.results["energy"] = np.asarray(out["energy"])
        self.results["forces"] = np.asarray(out["forces"])
        self.results["stress"] = np.asarray(out["stress"])
        self.results["free_energy"] = self.results["energy"]  # calculator/active.py:527
        self.maximum_force = float(np.abs(self.results["forces"]).max()) if N else 0.0
        self._beta = out["beta"]
        timings.append(time.time())
        # inactive branch of calculator/active.py:492-499
        covloss_max = float(np.max(self._beta)) if N else 0.0
        self.covlog = f"{covloss_max}"
        self.deltas = None
        timings.append(time.time())
        try:
            temperature = a.get_temperature()
        except Exception:
            temperature = 0.0
        self.log("{} {} {} {}".format(float(self.results["energy"]), temperature, self.covlog, ""))
        self.step += 1
        if self.report_timings:
            d = np.diff(timings)
            self.log(("timings:" + len(d) * " {:0.2g}").format(*d) + f" total: {d.sum():0.2g}")

    def get_covloss(self):
        """calculator/active.py:781-804 for the last calculated frame."""
        return self._beta

    # ------------------------------------------------------------------ logging (active.py:1129-1134)
    def log(self, mssge, mode="a"):
        if self.logfile and self.rank == 0:
            with open(self.logfile, mode) as f:
                f.write("{} {} {}\n".format(datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"), self.step, mssge))
        if self.stdout and self.rank == 0:
            print(mssge)
