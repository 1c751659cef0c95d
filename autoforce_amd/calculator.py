"""ActiveCalculator — the reference's ASE-calculator surface for the SGPR path, backed by
libsgpr_hip (MI355X).  Mirrors theforce/calculator/active.py:

  :104-135   constructor keywords               :425-535   calculate / post_calculate
  :548-611   results (E, F, stress)             :612-640   initiate_model / get_unique_lces
  :642-676   sample_rand_lces                   :706-768   _exact / snapshot / head
  :781-804   covloss                            :806-839   update_lce
  :841-885   update_inducing                    :887-933   update_data
  :942-984   update                             :989-1053  include_data / include_tape
  :1055-1098 build

All numerics (neighbour list, descriptors, K_nm, forces, covloss, training rows, K_mm edits,
Cholesky/QR) run on the device through `SGPRModel`; this file is the control flow and the
acceptance thresholds.  With `calculator=None` it is a pure evaluator; with a teacher
(`calculator=` any ASE-style calculator giving energy/forces/stress) it learns on the fly.

Works with real ASE when installed; otherwise with the minimal shim in autoforce_amd.ase_shim.
"""
import datetime
import os
import time

import numpy as np

try:  # pragma: no cover - ASE is absent from the build image
    from ase.calculators.calculator import Calculator, all_changes
    from ase.calculators.singlepoint import SinglePointCalculator
    from ase.units import GPa, kcal, mol
    kcal_mol = kcal / mol
    HAVE_ASE = True
except ImportError:
    from .ase_shim import Calculator, SinglePointCalculator, all_changes, kcal_mol
    GPa = 1.0 / 160.21766208
    HAVE_ASE = False

from .model import Local, SGPRModel
from .posterior import EPS, Frame, PosteriorPotential
from .sgprio import SgprIO
from .sharding import pack_partial, rank_of_atoms, unpack_total

inf = float("inf")


def default_kernel(lmax=3, nmax=3, exponent=4, cutoff=6.0, species=None, device=0, wildcard=False):
    """theforce/calculator/active.py:28-38.  Without `species`: the wildcard SeSoapKernel(lmax, nmax, exponent, cutoff,
    radii=DefaultRadii()) — the reference indexes a fixed 120-wide table, the device layout is dense over the species
    met so far (`wildcard=True` with the current table).  With `species`: the reference builds ONE SubSeSoapKernel per
    listed species and sums them; the only place where that differs from one kernel over the same table is the
    lone-atom term, added once per kernel object (SGPRModel's `lone_weight`)."""
    return SGPRModel(lmax, nmax, exponent, cutoff, species=species, device=device,
                     lone_weight=1 if wildcard or not species else len(species))


class Switch:
    """active.py:82-101: a threshold that depends on the largest force of the frame:
    Switch([v0, s1, v1, s2, v2]) is v0 below s1, v1 between s1 and s2, v2 above."""

    def __init__(self, value):
        self._value = value
        seq = list(value) if hasattr(value, "__iter__") else [value]
        self.switches = (-inf, *seq[1::2], inf)
        self.values = seq[0::2]
        if any(a > b for a, b in zip(self.switches, self.switches[1:])):
            raise RuntimeError("Switch is not ordered!")

    def __repr__(self):
        return f"{self._value}"

    def __call__(self, x):
        k = 0
        for k, (lo, hi) in enumerate(zip(self.switches, self.switches[1:])):
            if lo < x < hi:
                break
        return self.values[k]


def _switched(name):
    def get(self):
        return getattr(self, "_" + name)(self.maximum_force)

    def put(self, value):
        setattr(self, "_" + name, value if isinstance(value, Switch) else Switch(value))

    return property(get, put)


class ActiveCalculator(Calculator):
    implemented_properties = ["energy", "forces", "stress", "free_energy"]
    ediff, ediff_lb, ediff_ub, fdiff = (_switched(n) for n in ("ediff", "ediff_lb", "ediff_ub", "fdiff"))

    def __init__(self, covariance=None, calculator=None, process_group=None, meta=None, logfile="active.log",
                 pckl=None, tape=None, test=None, stdout=False, ediff=2 * kcal_mol, ediff_lb=None, ediff_ub=None,
                 ediff_tot=4 * kcal_mol, fdiff=3 * kcal_mol, noise_f=kcal_mol, ioptim=1, max_data=inf,
                 max_inducing=inf, kernel_kw=None, veto=None, include_params=None, eps_dr=0.1, ignore=None,
                 report_timings=False, step0_forced_fp=False, nbeads=1, engine=None):
        """
        covariance:    SGPRModel | PosteriorPotential | path to a saved model (.npz,
                       autoforce_amd.modelio) | None (+ kernel_kw with a `species` entry -> empty model)
        calculator:    None (evaluate only) | teacher: any ASE-style calculator (energy/forces/stress)
        process_group: None | torch.distributed process group over which atoms are sharded
        pckl:          None | path of the .npz the model is saved to after every update
        tape:          None | path of the .sgpr tape that accepted data / LCEs are appended to
        engine:        test hook — an object with SGPRModel's methods
        The remaining keywords have the reference's meaning (active.py:137-287).
        """
        Calculator.__init__(self)
        self._calc = calculator
        self.process_group = process_group
        self.pckl = pckl
        self.maximum_force = inf
        self.logfile, self.stdout, self._logpref = logfile, stdout, ""
        self.step = 0
        self._wildcard = False
        self._comm_note, self._hello = "", False
        self._peer_atoms = 0
        self._sys_cache = None
        self.get_model(engine if engine is not None else covariance, kernel_kw or {})
        self.ediff = ediff
        self.ediff_lb = ediff_lb or ediff
        self.ediff_ub = ediff_ub or ediff
        self.ediff_tot, self.fdiff, self.noise_f = ediff_tot, fdiff, noise_f
        self.ioptim, self._ioptim = ioptim, 0
        self.max_data, self.max_inducing = max_data, max_inducing
        self.meta = meta
        self.log("active calculator says Hello!", mode="w")
        self._hello = True
        if self._comm_note:
            self.log(self._comm_note)
        self.log_settings()
        self.log("model size: {} {}".format(*self.size))
        self.tape = None if tape is None else SgprIO(tape, rank=self.rank)
        self.test, self._last_test, self._ktest = test, 0, 0
        self.updated = False
        self._update_args = {}
        self._veto = {} if veto is None else dict(veto)
        self.include_params = {"fmax": inf}
        self.include_params.update(include_params or {})
        self.tune_for_md = True
        self.eps_dr = eps_dr
        self.ignore = [] if ignore is None else list(ignore)
        self.report_timings = report_timings
        self.step0_forced_fp = step0_forced_fp
        self.nbeads = nbeads
        self.deltas, self.covlog, self._cov = None, "", None
        self._cov_gen = None
        self.blind = False
        self._saved_for_tape = None
        self._beta = None
        self._nl = None
        if self.nbeads > 1:
            self.log(f"You are going quantum (PIMD)! Number of beads: {self.nbeads}")

    # ------------------------------------------------------------------ model plumbing
    def get_model(self, model, kernel_kw):
        """active.py:338-362."""
        if isinstance(model, str):
            from .modelio import load_model
            model = load_model(model)
        elif model is None:
            if self.pckl and os.path.isfile(self.pckl):
                from .modelio import load_model
                model = load_model(self.pckl)
            else:
                kw = dict(kernel_kw)
                if not kw.get("species"):
                    # the reference's default kernel is the species-wildcard one (active.py:28-38): the table
                    # starts with the first frame's species and grows when a new one turns up (_ensure_species)
                    self._wildcard = True
                    kw["species"] = [0]  # placeholder, replaced before the first evaluation
                    kw["wildcard"] = True
                model = default_kernel(**kw)
        if not isinstance(model, PosteriorPotential):
            model = PosteriorPotential(model)
        model._sync = self._broadcast if self.process_group is not None else None
        self.model = model
        self._attach_native_comm()

    def _attach_native_comm(self):
        """Sharded frames on the HIP engine: the ranks' partial sums are combined inside the library
        by ONE RCCL all-reduce on the step's stream (sgpr_comm_init); the process group only carries
        the 128-byte id during set-up.  Engines without that entry point (the CPU oracle engine of the
        tests) keep the host-side all-reduce of `_evaluate_engine`."""
        eng = self.model.engine
        if self.process_group is None or not hasattr(eng, "comm_init"):
            return
        import torch.distributed as dist
        rank, world = self._dist()
        if world < 2:
            return
        # SGPR_COLLECTIVE = auto (default): the library's own exchange through hipIpc-mapped buffers (deterministic sums, runs
        # with several ranks on one device, the device MD loop can run sharded on it), else RCCL, else the host-side
        # all-reduce; ipc / rccl / host: only that one (and the host-side all-reduce behind it)
        mode = os.environ.get("SGPR_COLLECTIVE", "auto")
        if mode in ("auto", "ipc") and hasattr(eng, "peer_export") and self._attach_peer(max(self._peer_atoms, 4096)):
            return
        if mode in ("ipc", "host"):
            return
        # every rank takes the same branch: the outcome is agreed on (MIN over ranks) before anybody evaluates.  A rank
        # that cannot build the communicator (RCCL missing; two ranks on one device — RCCL refuses duplicate GPUs)
        # must not leave the others inside ncclCommInitRank: the id travels first, the attempt is made by all, and on
        # any failure all fall back to the host-side all-reduce of `_evaluate_engine`.
        import torch
        ok = 1
        try:
            box = [eng.comm_unique_id() if rank == 0 else None]
        except Exception as exc:  # noqa: BLE001
            box, ok = [None], 0
            self._comm_note = f"native communicator unavailable on rank 0: {exc}"
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.process_group, 0), group=self.process_group)
        if box[0] is None:
            ok = 0
        else:
            try:
                from .watchdog import Watchdog
                with Watchdog("sgpr_comm_init (ncclCommInitRank)", rank=rank):
                    eng.comm_init(box[0], rank, world)
            except Exception as exc:  # noqa: BLE001
                ok = 0
                self._comm_note = f"native communicator not built ({exc}): host-side all-reduce instead"
        if self._min_over_ranks(ok) == 0:
            if getattr(eng, "comm_world", 1) > 1:
                eng.comm_destroy()
            if not self._comm_note:
                self._comm_note = "native communicator not built on another rank: host-side all-reduce instead"
        if self._comm_note and self._hello:
            self.log(self._comm_note)

    def _attach_peer(self, atoms_cap):
        """The library's own exchange (SGPRModel.peer_export / peer_attach) for frames of up to `atoms_cap` atoms: every rank
        exports its receive buffers, the handles travel over the process group, everybody maps everybody's.  The outcome is
        agreed on (MIN over the ranks): all attached, or nobody."""
        import torch
        import torch.distributed as dist
        eng = self.model.engine
        rank, world = self._dist()
        ok, blob, why = 1, None, ""
        try:
            blob = eng.peer_export(rank, world, 7 * int(atoms_cap) + 11)
        except Exception as exc:  # noqa: BLE001
            ok, why = 0, str(exc)
        blobs = [None] * world
        dist.all_gather_object(blobs, blob, group=self.process_group)
        if ok and all(b is not None for b in blobs):
            try:
                eng.peer_attach(blobs)
            except Exception as exc:  # noqa: BLE001
                ok, why = 0, str(exc)
        else:
            ok = 0
        # every rank tries the exchange once before anybody relies on it (collective: all ranks that attached take part)
        if self._min_over_ranks(ok) == 1 and not eng.peer_selftest(rank, world):
            ok, why = 0, "the self-test exchange did not return the expected sum"
        if self._min_over_ranks(ok) == 0:
            if blob is not None:
                eng.peer_destroy()
            self._comm_note = f"the library's own exchange was not built ({why or 'another rank failed'})"
            if self._hello:
                self.log(self._comm_note)
            return False
        self._peer_atoms = int(atoms_cap)
        return True

    def _min_over_ranks(self, value):
        """MIN of an integer over the process group (an NCCL / RCCL group reduces device tensors only)."""
        import torch
        import torch.distributed as dist
        t = torch.tensor([int(value)])
        if dist.get_backend(self.process_group) == "nccl":
            t = t.cuda()
        dist.all_reduce(t, op=dist.ReduceOp.MIN, group=self.process_group)
        return int(t.item())

    def _peer_fit(self, engine, N):
        """A frame larger than the buffers of the library's own exchange hold: export and attach again (collective: every
        rank sees the same N and takes the same branch)."""
        world = self._dist()[1]
        if world > 1 and getattr(engine, "peer_world", 1) == world and N > self._peer_atoms and engine is self.engine:
            self._attach_peer(2 * N)

    @property
    def maximum_force(self):
        """max |F| of the frame in `results` (active.py:573-576; what the Switch thresholds depend on), computed when
        somebody asks: a prediction-only step never does."""
        if self._maxf is None:
            f = self.results.get("forces")
            self._maxf = inf if f is None else (float(np.abs(f).max()) if len(f) else 0.0)
        return self._maxf

    @maximum_force.setter
    def maximum_force(self, value):
        self._maxf = value

    @property
    def engine(self):
        return self.model.engine

    @property
    def active(self):
        return self._calc is not None

    @property
    def size(self):
        return self.model.ndata, len(self.model.X)  # active.py:375-376

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

    def _tensor(self, a):
        import torch
        import torch.distributed as dist
        t = torch.from_numpy(np.ascontiguousarray(a))
        return t.cuda() if dist.get_backend(self.process_group) == "nccl" else t

    def _broadcast(self, arrays):
        """rank 0's mu / choli / ridge / mean weights become everyone's (gppotential.py:592-596)."""
        import torch.distributed as dist
        for a in arrays:
            t = self._tensor(a)
            dist.broadcast(t, 0, group=self.process_group)
            a[...] = t.cpu().numpy()

    # ------------------------------------------------------------------ the hot path
    def _system(self, atoms):
        cell = np.asarray(getattr(atoms.cell, "array", atoms.cell), dtype=float).reshape(3, 3)
        # (the int32 copy of the atomic numbers is kept while the frame's numbers array is the same object with the same
        # content: the conversion of 4096 numbers is 2-3 us of a 130-us step)
        src = atoms.numbers
        c = self._sys_cache
        if c is None or c[0] is not src or len(c[1]) != len(src):   # (the calculator's copy of the atoms keeps ONE numbers array per system)
            c = self._sys_cache = (src, np.asarray(src, dtype=np.int32))
        return (c[1], np.asarray(atoms.positions, dtype=float), cell, np.asarray(atoms.pbc, dtype=bool))

    def _evaluate_engine(self, engine):
        """One device pass of `engine` over the current atoms (sharded + all-reduced when a process
        group is attached): dict(energy, forces, stress, beta, cov)."""
        numbers, positions, cell, pbc = self._system(self.atoms)
        N = len(numbers)
        rank, world = self._dist()
        if not (engine.m > 0 and engine.mu is not None):
            # an empty model predicts its mean (zeros) and knows nothing: covloss = inf
            return dict(energy=0.0, forces=np.zeros((N, 3)), stress=np.zeros(6), beta=np.full(N, inf), ready=False)
        self._peer_fit(engine, N)
        fast = getattr(engine, "predict_view", None)
        if fast is not None and (world == 1 or getattr(engine, "comm_world", 1) == world):
            # results as views of the buffer the device wrote (valid until the call after next): copied where they are kept
            v = fast(numbers, positions, cell, pbc, rank=rank, world=world)
            out = dict(energy=float(v["energy"]), forces=np.array(v["forces"]), stress=np.array(v["stress"]), beta=np.array(v["beta"]))
        else:
            out = engine.predict(numbers, positions, cell, pbc, rank=rank, world=world, cov=False, beta=True)
        if world > 1 and getattr(engine, "comm_world", 1) == world:
            pass  # the library's own RCCL all-reduce already combined the ranks (totals on every rank)
        elif world > 1:
            import torch.distributed as dist
            v = self._tensor(pack_partial(out, N))
            dist.all_reduce(v, group=self.process_group)  # active.py:562,601,602,777 in one collective
            out = unpack_total(v.cpu().numpy(), N)
        out["ready"] = True
        return out

    @property
    def cov(self):
        """K_nm of the last evaluated frame, [N, m] (this rank's rows) — active.py:464 `self.cov`.
        It stays on the device until somebody looks: N x m doubles per step over PCIe would cost
        more than the step itself."""
        if self._cov is None and self.atoms is not None and self.engine.m > 0:
            eng = self.engine
            gen = getattr(eng, "generation", None)
            if gen is not None and gen != self._cov_gen:
                # the device moved on to other frames since this one was evaluated (training rows, trial
                # models, rattled copies): evaluate the current atoms again rather than hand out the
                # K_nm of whatever frame came last
                numbers, positions, cell, pbc = self._system(self.atoms)
                rank, world = self._dist()
                eng.predict(numbers, positions, cell, pbc, rank=rank, world=world, cov=False, beta=False)
                self._cov_gen = eng.generation
            fetch = getattr(eng, "last_cov", None)
            self._cov = fetch(len(self.atoms)) if fetch else None
        return self._cov

    @cov.setter
    def cov(self, value):
        self._cov = value

    def update_results(self, retain_graph=False, covloss_only=False):
        """active.py:548-611 + :781-804 in one device pass: E, F, stress, covloss, cov.
        covloss_only: refresh `cov` and the covloss after the model changed but leave `results`
        alone — inside update_inducing the reference extends cov by a column and keeps the
        pre-update predictions, which update_data then offers as 'fake' labels."""
        out = self._evaluate_engine(self.engine)
        self._cov = None  # fetched lazily
        self._cov_gen = getattr(self.engine, "generation", None)  # the frame `cov` would be fetched from
        self._nl = None
        self._beta = out["beta"]
        if covloss_only:
            return
        self._set_results(out)

    def _set_results(self, out):
        self.results["energy"] = np.asarray(out["energy"])
        self.results["forces"] = np.asarray(out["forces"])
        self.results["stress"] = np.asarray(out["stress"])
        self._maxf = None   # (maximum_force: computed when somebody asks, from these forces)

    def _ensure_species(self, numbers):
        """Wildcard mode (no `species` in kernel_kw): extend the model's table to the species met."""
        if not self._wildcard:
            return
        have = [z for z in self.engine.species if z != 0]
        new = sorted(set(int(z) for z in numbers) - set(have))
        if new or 0 in self.engine.species:
            table = sorted(set(have) | set(new))
            if len(table) > 16:
                raise RuntimeError(f"{len(table)} species: the device kernels hold at most 16 species slots")
            self.model.retable(table)
            self._attach_native_comm()
            self.log(f"species table -> {table}")

    def calculate(self, atoms=None, properties=("energy",), system_changes=all_changes):
        timings = [time.time()]
        if self.size[1] == 0 and not self.active:
            raise RuntimeError("you forgot to assign a DFT calculator!")  # active.py:429-430
        Calculator.calculate(self, atoms, properties, system_changes)
        self._ensure_species(self.atoms.numbers)
        self.maximum_force = inf
        timings.append(time.time())
        if self._needs_seed():
            self.initiate_model()
            self._update_args = dict(data=False)
        timings.append(time.time())
        self.update_results(self.active or (self.meta is not None))
        timings.append(time.time())
        self.deltas = None
        self.covlog = ""
        if self.active and not self.veto():
            if (self.step + 1) % self.nbeads == 1 or self.nbeads == 1:  # PIMD: only the first bead samples
                pre = dict(self.results)
                m, n = self.update(**self._update_args)
                if n > 0 or m > 0:
                    self.update_results(self.meta is not None)
                    if self.step > 0:
                        self.deltas = {q: self.results[q] - pre[q] for q in ("energy", "forces", "stress")}
        else:
            covloss_max = float(np.max(self.get_covloss())) if len(self.atoms) else 0.0
            self.covlog = f"{covloss_max}"
            if self.rank == 0 and covloss_max > self.ediff:   # (ediff last: a Switch asks for max |F|)
                self._side_file("active_uncertain", self.atoms, None, "a")
        timings.append(time.time())
        self.post_calculate(timings)

    def _needs_seed(self):
        return self.step == 0 and self.active and self.model.ndata == 0  # active.py:458-461

    def post_calculate(self, timings):
        """active.py:504-535."""
        energy = self.results["energy"]
        if self.active and self.test and self.step - self._last_test > self.test:
            self._test()
        meta = ""
        if self.meta is not None:
            energies, kwargs = self.meta(self)
            if energies is not None:
                meta = f"meta: {float(np.sum(energies))}"
        if (self.logfile or self.stdout) and self.rank == 0:  # (nobody reads the line otherwise: skip the kinetic energy)
            try:
                temperature = self.atoms.get_temperature()
            except Exception:
                temperature = 0.0
            self.log("{} {} {} {}".format(float(energy), temperature, self.covlog, meta))
        self.step += 1
        self.results["free_energy"] = self.results["energy"]  # active.py:527
        timings.append(time.time())
        if self.report_timings:
            d = np.diff(timings)
            self.log(("timings:" + len(d) * " {:0.2g}").format(*d) + f" total: {d.sum():0.2g}")

    # ------------------------------------------------------------------ MD with the state in device memory
    def md_on_device_ok(self):
        """The device loop replaces calculate() only where calculate() does nothing the device cannot see: a
        single process, no periodic test, no meta-dynamics hook, no force veto, one bead."""
        eng = self.engine
        world = self._dist()[1]
        # (several ranks: only over the library's own exchange — every rank then integrates all atoms from the summed
        # forces and halts at the same step, sgpr_md_run)
        return (hasattr(eng, "md_run") and (world == 1 or getattr(eng, "peer_world", 1) == world) and not self.test
                and self.meta is None and "forces" not in self._veto and self.nbeads == 1 and eng.m > 0 and eng.mu is not None)

    def _md_gate(self, numbers):
        """The smallest covloss at which calculate() would do more than log (update_lce, active.py:806-839): below
        ediff_lb nothing happens unless a species of the frame has fewer than two inducing LCEs."""
        if not self.active:
            return 0.0  # evaluate only: never halt
        if any(self.model.indu_counts[int(z)] < 2 for z in set(int(z) for z in numbers)):
            return EPS
        return float(min(self._ediff_lb.values))

    def run_md(self, atoms, steps, temperature_K, dt_fs=1.0, friction=1e-3, rng=None, chunk=256, seed=1, sync_every=None,
               tdamp_fs=None):
        """`steps` steps of Langevin NVT (friction = 0: NVE) from atoms.positions / velocities, as cl/md.py:117-128 sets
        it up around this calculator — but the state stays in device memory between model updates: the integrator runs
        inside the step's last kernel (SGPRModel.md_run), the host reads 16 scalars per step and writes the same log
        line calculate() would ("energy temperature covloss", active.py:518-523), and a step whose largest covloss
        reaches the sampling threshold stops the device, is handed to calculate() — which updates the model exactly as
        it does inside an ASE loop — and the run goes on from there with the new model.  Yields (step, energy,
        temperature, updated, wall seconds) per step (device steps share the wall time of their batch evenly).  rng: a
        numpy Generator whose normal deviates move the atoms (the stream of workloads.langevin_nvt: the two loops then
        agree bit for bit), or None — the integrator draws its own on the device (counter-based on `seed`: no host
        generator and no upload on the step's path, same trajectory however the run is batched).  sync_every: steps
        k = 0 mod sync_every end a batch and atoms.positions / velocities are theirs when they are yielded (a trajectory
        writer's loginterval, cl/md.py:24); atoms.positions / velocities are current at every yield that follows an update
        and at the end.  Falls back to the host loop (workloads.langevin_nvt) where md_on_device_ok() says no.
        tdamp_fs: Nose-Hoover NVT with that damping time instead of Langevin — the reference's DEFAULT dynamics,
        md(dynamics="NPT", bulk_modulus=None) = ase.md.npt.NPT(pfactor=None, ttime=tdamp fs) (cl/md.py:17, :131-166); no
        deviates, `friction` / `rng` / `seed` unused; host loop: workloads.nose_hoover_nvt."""
        from .ase_shim import kB
        from .workloads import FS, MASS, langevin_nvt, nose_hoover_nvt
        nh = tdamp_fs is not None
        if len(getattr(atoms, "constraints", None) or ()):
            # (neither integrator of this method knows ASE's constraints: an ASE dynamics object around calculate() does)
            raise NotImplementedError("run_md integrates unconstrained atoms; with atoms.constraints set, drive calculate() "
                                      "from an ase.md dynamics object as theforce/cl/md.py does")
        numbers, pos, cell, pbc = self._system(atoms)
        N = len(numbers)
        on_device_rng = rng is None and friction > 0.0 and not nh
        rng = np.random.default_rng(seed) if rng is None else rng
        if getattr(atoms, "_masses", "ase") is None:  # (the stand-in Atoms without masses; ase.Atoms knows its own)
            masses = np.array([MASS[int(z)] for z in numbers])
            atoms._masses = masses.copy()
        else:
            masses = np.asarray(atoms.get_masses(), float)
        vel = atoms.get_velocities()
        vel = np.zeros((N, 3)) if vel is None else np.asarray(vel, float)
        first_on_host = self._needs_seed() or not self.md_on_device_ok()
        if first_on_host:
            # (an empty model is seeded by its first calculate(); then the device loop can take over)
            atoms.calc = self
            atoms.get_forces()
            if not self.md_on_device_ok():
                loop = (nose_hoover_nvt(self, numbers, pos, cell, pbc, steps, temperature_K, dt_fs, tdamp_fs, vel=vel) if nh else
                        langevin_nvt(self, numbers, pos, cell, pbc, steps, temperature_K, dt_fs, friction, vel=vel, rng=rng))
                for st, E, T, _, p, v, *rest in loop:
                    atoms.positions = p
                    atoms.set_velocities(v)
                    yield st, E, T, bool(self.updated), _
                return
        eng = self.engine
        kT = kB * temperature_K
        self._peer_fit(eng, N)
        eng.md_begin(numbers, pos, cell, pbc, masses, vel, dt=dt_fs * FS, friction=0.0 if nh else friction, kT=kT,
                     seed=(int(seed) or 1) if on_device_rng else 0, ttime=tdamp_fs * FS if nh else None)
        # (skip_gate: the configuration has been through calculate() — logged, counted, the model updated if need be —
        # and is evaluated once more on the device, whatever its covloss, to move on from it)
        done, rows, skip_gate, t_host = 0, np.empty((0, N, 3)), first_on_host, 0.0
        batch = min(8, chunk)   # evaluations per md_run call: grows while nothing halts the device, shrinks back after a halt
        while done <= steps:    # (every call uploads its rows of deviates; a halt throws the unused ones' upload away)
            n = 1 if skip_gate else min(batch, steps + 1 - done)
            if sync_every and not skip_gate:
                n = min(n, sync_every - done % sync_every if done % sync_every else 1)   # (… a batch ends on a multiple)
            final = done + n == steps + 1
            need = 0 if (on_device_rng or nh) else (n - 1 if final else n)
            if len(rows) < need:
                rows = np.concatenate([rows, rng.normal(size=(need - len(rows), N, 3))])
            noise = None if (on_device_rng or nh) else (rows[:n] if len(rows) >= n else np.concatenate([rows, np.zeros((n - len(rows), N, 3))]))
            gate = 0.0 if skip_gate else self._md_gate(numbers)
            t_run = time.time()
            sc, code = eng.md_run(n, noise, ediff=gate, final=final)
            accepted = len(sc) - 1 if code == 1 else len(sc)
            share = (time.time() - t_run) / max(accepted, 1)
            lines, out = [], []
            for r in sc[:accepted]:
                upd, wall = False, share
                if skip_gate:      # (the configuration calculate() has just dealt with, evaluated again with the new model:
                    skip_gate, upd, wall = False, bool(self.updated), share + t_host   # its line is written, its step counted)
                else:
                    lines.append((self.step, "{} {} {} {}".format(float(r[0]), float(r[13] / (3 * N * kB)), float(r[11]), "")))
                    self.step += 1
                out.append((done, float(r[0]), float(r[12] / (3 * N * kB)), upd, wall))
                done += 1
            self._log_lines(lines)
            if sync_every and out and out[-1][0] % sync_every == 0 and code != 1:
                # the configuration of the batch's last row: the device has moved on to the next one unless the run is over
                st = eng.md_state(which=0 if final else -1)
                atoms.positions = st["positions"]
                atoms.set_velocities(st["velocities_pre"])
            yield from out
            rows = rows[accepted:]
            batch = min(8, chunk) if code else min(2 * batch, chunk)
            if code == 1:
                t_host = time.time()
                st = eng.md_state(results=True)
                atoms.positions = st["positions"]
                atoms.set_velocities(st["velocities_pre"])   # what the integrator holds when it asks for forces
                atoms.calc = self
                self.results = {}
                self.calculate(atoms)        # update_results + update + the log line, as inside an ASE loop
                skip_gate = True
                t_host = time.time() - t_host
        st = eng.md_state(results=True)
        atoms.positions = st["positions"]
        atoms.set_velocities(st["velocities"])

    def _log_lines(self, lines):
        """A batch of per-step lines in one open (a device loop produces them by the hundred)."""
        if not lines:
            return
        if self.logfile and self.rank == 0:
            stamp = datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S")
            with open(self.logfile, "a") as f:
                f.write("".join("{}{} {} {}\n".format(self._logpref, stamp, st, m) for st, m in lines))
        if self.stdout and self.rank == 0:
            for _, m in lines:
                print(m)

    def veto(self):
        """active.py:537-546."""
        if self.size[0] < 2 or "forces" not in self._veto:
            return False
        if np.abs(self.results["forces"]).max() >= self._veto["forces"]:
            self.log("an update is vetoed!")
            return True
        return False

    def get_covloss(self):
        """active.py:781-804 for the last evaluated frame (computed with it on the device)."""
        return self._beta

    # ------------------------------------------------------------------ LCEs of the current frame
    def _neighbors(self):
        if self._nl is None:
            self._nl = self.engine.neighbors(len(self.atoms))
        return self._nl

    def _local_here(self, k, system=None):
        if system is None and hasattr(self.engine, "local"):
            z, r = self.engine.local(k)  # one atom's list from the device, not the whole neighbour list
            return Local(int(self.atoms.numbers[k]), z, r)
        numbers, positions, cell, _ = system or self._system(self.atoms)
        ptr, j, off = self._neighbors()
        a, b = int(ptr[k]), int(ptr[k + 1])
        r = positions[j[a:b]] - positions[k] + off[a:b].astype(float) @ cell  # descriptor/atoms.py:367-368
        return Local(int(numbers[k]), numbers[j[a:b]], r)

    def local(self, k):
        """atoms.local(k, detach=True) (descriptor/atoms.py:365-382); in a sharded run the owner of
        atom k hands the LCE to everyone."""
        rank, world = self._dist()
        if world == 1:
            return self._local_here(k)
        import torch.distributed as dist
        owner = int(rank_of_atoms(self.atoms.numbers, self.engine.species, world)[k])
        box = [self._local_here(k) if rank == owner else None]
        dist.broadcast_object_list(box, src=dist.get_global_rank(self.process_group, owner), group=self.process_group)
        return box[0]

    def _locals_of(self, numbers, positions, cell, pbc, indices=None):
        """LCEs of an arbitrary configuration (device neighbour list, unsharded)."""
        N = len(numbers)
        self.engine.predict(numbers, positions, cell, pbc, cov=False, beta=False)
        self._nl = self.engine.neighbors(N)
        sysm = (numbers, positions, cell, pbc)
        out = [self._local_here(k, sysm) for k in (range(N) if indices is None else indices)]
        self._nl = None
        return out

    # ------------------------------------------------------------------ teacher calls
    def _exact(self, atoms):
        """active.py:706-738: one teacher (ab initio) calculation."""
        tmp = atoms.copy()
        tmp.calc = self._calc
        energy = tmp.get_potential_energy()
        forces = tmp.get_forces()
        stress = tmp.get_stress()
        self.log("exact energy: {}".format(energy))
        self.log("exact stress[GPa]: {}  {}  {}".format(*(stress[:3] / GPa)))
        if self.model.ndata > 0 and "energy" in self.results:
            dE = self.results["energy"] - energy
            df = np.abs(self.results["forces"] - forces)
            self.log("predicted stress[GPa]: {}  {}  {}".format(*(self.results["stress"][:3] / GPa)))
            self.log("errors (pre):  del-E: {:.2g}  max|del-F|: {:.2g}  mean|del-F|: {:.2g}".format(
                float(dE), df.max(), df.mean()))
        self._last_test = self.step
        return energy, forces, stress

    def snapshot(self, fake=False, atoms=None):
        """active.py:740-751: the current frame with exact (teacher) or fake (own prediction) labels."""
        atoms = self.atoms if atoms is None else atoms
        if fake:
            e, f, s = self.results["energy"], self.results["forces"], self.results["stress"]
        else:
            e, f, s = self._exact(atoms)
        fr = Frame.from_atoms(atoms, e, f, s)
        if not fake and self.tape:
            self._saved_for_tape = fr
        return fr

    def head(self):
        """active.py:753-762: replace the fake labels of the newest datum by exact ones."""
        e, f, s = self._exact(self._as_atoms(self.model.data[-1]))
        self.model.refresh_targets(-1, e, f, s)
        if self.tape:
            self._saved_for_tape = self.model.data[-1]

    def _side_file(self, stem, atoms, results, mode):
        """The trajectory side files of the reference (active.py:495-499 `active_uncertain.traj`, :684-696 `active_FP.traj` /
        `active_ML.traj`): ase.io.Trajectory when ASE is installed; without it the same frames as extended XYZ (`<stem>.xyz`,
        ASE's own text format: `ase.io.read` takes it), results as energy / forces / stress of the frame.  Written beside the
        log file — the working directory for the reference's default `logfile="active.log"`, where the reference writes them —
        by calculators that keep a log: one created with `logfile=None` (a quiet evaluator inside somebody else's loop, the
        bench's timed calculate()) writes no files at all."""
        if not self.logfile:
            return
        stem = os.path.join(os.path.dirname(os.path.abspath(self.logfile)), stem)
        if HAVE_ASE:  # pragma: no cover
            import ase.io
            tmp = atoms.copy()
            tmp.calc = None if results is None else SinglePointCalculator(tmp, **results)
            ase.io.Trajectory(stem + ".traj", mode).write(tmp)
            return
        from .sgprio import Frame as XyzFrame, format_extxyz
        r = results or {}
        e = r.get("energy")
        fr = XyzFrame(np.asarray(atoms.numbers), np.asarray(atoms.positions, float), np.asarray(getattr(atoms.cell, "array", atoms.cell), float),
                      np.asarray(atoms.pbc, bool), None if e is None else float(e), r.get("forces"), r.get("stress"))
        with open(stem + ".xyz", mode) as f:
            f.writelines(format_extxyz(fr))

    def _test(self):
        """active.py:678-704."""
        energy, forces, stress = self._exact(self.atoms)
        self._ktest += 1
        if self.rank == 0:
            mode = "a" if self._ktest > 1 else "w"
            self._side_file("active_FP", self.atoms, dict(energy=energy, forces=forces, stress=stress), mode)
            self._side_file("active_ML", self.atoms, {q: self.results[q] for q in ("energy", "forces", "stress")}, mode)
        self.log("testing energy: {}".format(energy))
        dE = self.results["energy"] - energy
        df = np.abs(self.results["forces"] - forces)
        ds = np.abs(self.results["stress"] - stress)
        self.log("errors (test):  del-E: {:.2g}  max|del-F|: {:.2g}  mean|del-F|: {:.2g} mean|del-P|: {:.2g}".format(
            float(dE), df.max(), df.mean(), np.mean(ds[:3])))
        self._last_test = self.step
        return energy, forces

    # ------------------------------------------------------------------ seeding
    def initiate_model(self):
        """active.py:612-629."""
        data = [self.snapshot()]
        idx = self.get_unique_lces()
        system = self._system(self.atoms)
        inducing = self._locals_of(*system, indices=idx)
        self.model.set_data(data, inducing)
        if self.tape:
            if self._saved_for_tape is not None:
                self.tape.write(self._saved_for_tape)
                self._saved_for_tape = None
            for loc in inducing:
                self.tape.write(loc)
        details = [(int(j), int(self.atoms.numbers[j])) for j in idx]
        self.log("seed size: {} {} details: {}".format(*self.size, details))
        if self.tune_for_md:
            self.sample_rand_lces(indices=idx, repeat=1)
        self.optimize()

    def get_unique_lces(self, thresh=0.95):
        """active.py:631-654: greedy cover — atom i is kept unless k(i, j) >= thresh for a kept j.
        k(atoms, atoms) is K_mm of the frame's own LCEs, one device GEMM."""
        system = self._system(self.atoms)
        locs = self._locals_of(*system)
        scratch = self.engine.scratch()
        scratch.set_inducing(locs)
        k = scratch.M
        scratch.close()
        unique = []
        for i in range(len(locs)):
            if all(k[i, j] < thresh for j in unique):
                unique.append(i)
        return unique

    def sample_rand_lces(self, indices=None, repeat=1):
        """active.py:656-676: LCEs of a slightly rattled copy, offered to update_lce."""
        added = 0
        numbers, positions, cell, pbc = self._system(self.atoms)
        for _ in range(repeat):
            rattled = positions + np.random.uniform(-0.05, 0.05, size=positions.shape)
            order = np.random.permutation(len(numbers)) if indices is None else indices
            for loc in self._locals_of(numbers, rattled, cell, pbc, indices=order):
                added += abs(self.update_lce(loc))
        self.log(f"added {added} randomly displaced LCEs")

    # ------------------------------------------------------------------ the sampling rules
    # The sampling rules of active.py:806-984 as DATA: what to do with an environment is a function of where its covloss
    # sits relative to the two thresholds and of how many inducing LCEs its species already has.
    #   band     covloss >= ediff_ub          ediff_lb <= covloss < ediff_ub        covloss < ediff_lb
    #   m >= 2   take it (+1)                 trial against ediff                   leave it
    #   m < 2    take it, flag blind (-1)     trial against EPS (any change)        take it if covloss > EPS (-1)
    # "take" = add_inducing; "trial" = add_1inducing (add, refit, keep only if the predictions moved by the threshold).
    _TAKE, _TRIAL, _LEAVE = "take", "trial", "leave"

    def _lce_rule(self, beta, m):
        """(action, code reported when the LCE is kept, trial threshold) for a covloss and an inducing count."""
        scarce = m < 2
        if beta >= self.ediff_ub:
            return self._TAKE, (-1 if scarce else 1), None
        if beta >= self.ediff_lb:
            return self._TRIAL, None, (EPS if scarce else self.ediff)
        if scarce and beta > EPS:
            return self._TAKE, -1, None
        return self._LEAVE, 0, None

    def _covloss_of(self, loc):
        """Covloss of an environment that is not an atom of the current frame (active.py:809-818)."""
        if not (self.engine.m > 0 and self.model.choli is not None):
            return inf
        k, _ = self.engine.kernel_local(loc)
        proj = self.model.choli @ k
        with np.errstate(invalid="ignore"):
            return np.sqrt(np.maximum((1.0 - proj @ proj) * self.model._vscale.get(loc.number, inf), 0.0))

    def update_lce(self, loc, beta=None):
        """active.py:806-839: returns +1 / -1 (kept; -1: its species was short of inducing LCEs — "blind") or 0."""
        if loc.number not in self.engine.species:
            return 0
        beta = self._covloss_of(loc) if beta is None else beta
        action, code, threshold = self._lce_rule(beta, self.model.indu_counts[loc.number])
        if action == self._LEAVE:
            return 0
        if action == self._TAKE:
            self.model.add_inducing(loc)
        else:
            code, _ = self.model.add_1inducing(loc, threshold)
            if code == 0:
                return 0
        if self.model.ridge > 0.0:          # the new LCE made K_mm need a jitter: not worth having (active.py:829-831)
            self.model.pop_1inducing()
            return 0
        if self.tape:
            self.tape.write(loc)
        if self.ioptim == 0:
            self.optimize()
        return code

    def _largest_covloss(self, beta, chosen):
        """The first atom in descending order of covloss that is neither chosen nor ignored (active.py:851-856 walks
        a full argsort; only its head is ever used: one argmax over the eligible atoms, ties to the lowest index as a
        stable sort would — 0.8 ms of every step of an active run at 16384 atoms)."""
        skip = chosen + self.ignore
        if len(skip) < len(beta) and not np.isnan(beta).any():
            if not skip:
                return int(np.argmax(beta))
            masked = np.array(beta, dtype=float)
            idx = [int(i) for i in skip if 0 <= int(i) < len(masked)]   # (an `ignore` entry outside the frame is inert, as
            masked[np.asarray(idx, dtype=int)] = -inf                    # in the reference's `k not in self.ignore`)
            if np.isfinite(masked).any() or (masked == inf).any():
                return int(np.argmax(masked))
        order = np.argsort(-beta, kind="stable")
        return next((int(i) for i in order if int(i) not in chosen and int(i) not in self.ignore), int(order[-1]))

    def update_inducing(self):
        """active.py:841-885: greedy — offer the atom with the largest covloss until one is refused."""
        added_beta = added_diff = 0
        chosen = []
        added_covloss = None
        N = len(self.atoms)
        while len(chosen) < N:
            beta = self.get_covloss()
            k = self._largest_covloss(beta, chosen)
            if np.isclose(beta[k], 1.0):
                self.blind = True
            loc = self.local(k)
            added = self.update_lce(loc, beta=beta[k])
            if added == 0:
                break
            if added == -1:
                self.blind = True
                added_beta += 1
            else:
                added_diff += 1
            chosen.append(k)
            added_covloss = beta[k]
            self.update_results(covloss_only=True)  # cov gains a column, choli changed: every covloss moves
        added = added_beta + added_diff
        if added > 0:
            self.log("added indu: {} ({},{}) -> size: {} {} details: {:.2g} {}".format(
                added, added_beta, added_diff, *self.size, float(added_covloss), ""))
            if self.blind:
                self.log("model may be blind -> go robust")
        self.covlog = f"{float(np.max(self.get_covloss()))}"
        return added

    def _same_as_last_datum(self):
        """tune_for_md (active.py:888-897): the frame has barely moved since the newest datum."""
        if not (self.tune_for_md and len(self.model.data) > 2):
            return False
        last = self.model.data[-1]
        return (last.natoms == len(self.atoms) and bool((last.numbers == self.atoms.numbers).all())
                and bool((np.abs(last.positions - self.atoms.positions) < self.eps_dr).all()))

    def _optimize_after_data(self):
        """ioptim: 0 / 2 refit the hyper-parameters after every datum, k > 2 after every (k - 1)-th (active.py:919-926)."""
        if self.ioptim in (0, 2):
            self.optimize()
        elif self.ioptim > 2:
            self._ioptim = (self._ioptim + 1) % (self.ioptim - 1)
            if self._ioptim == 0:
                self.optimize()

    def update_data(self, try_fake=True, internal=False, save_model=True):
        """active.py:887-933: offer the current frame as a datum — with the model's own predictions as labels first
        (try_fake), exact ones once it is accepted.  Returns the number of data added (0 or 1)."""
        if self._same_as_last_datum():
            return 0
        before = self.model.ndata
        _, de, df = self.model.add_1atoms_fast(self.snapshot(fake=try_fake), self.ediff_tot, self.fdiff)
        added = self.model.ndata - before
        self.log(f"DF: {df}  accept: {added}")
        if added <= 0:
            return added
        if try_fake:
            self.head()
        if self.tape and self._saved_for_tape is not None:
            self.tape.write(self._saved_for_tape)
            self._saved_for_tape = None
        self.log("added data: {} -> size: {} {}".format(added, *self.size))
        self._optimize_after_data()
        if save_model:
            self.save_model()
        return added

    def optimize(self):
        self.model.make_munu(algo=3, noise_f=self.noise_f)  # active.py:939-940

    def _wants_data(self, n_lces, inducing, data):
        """Is the frame offered as a datum?  After new LCEs (if data sampling is on); and when only data are sampled
        (include_tape), whenever some covloss is still above ediff (active.py:947-951)."""
        if inducing:
            return n_lces > 0 and data
        return bool(np.max(self.get_covloss()) > self.ediff)

    def _after_model_change(self):
        """Size limits, the refit that goes with them, the three log lines, the model file (active.py:961-981)."""
        # (ioptim == 1: optimize() below refits the downsized model from the same matrix — hyper-parameter search
        # and weights — so the refit downsize would end with is the one result nobody reads)
        if any(self.model.downsize(self.max_data, self.max_inducing, first=True, lii=True, remake=self.ioptim != 1)):
            self.log("downsized -> size: {} {}".format(*self.size))
        if self.ioptim == 1:
            self.optimize()
        self.log("fit error (mean,mae): E: {:.2g} {:.2g}   F: {:.2g} {:.2g}   R2: {:.4g}".format(
            *(float(v) for v in self.model._stats)))
        self.log(f"noise: {self.model.scaled_noise}")
        self.log(f"mean: {self.model.mean}")
        self.save_model()
        self.updated = True

    def update(self, inducing=True, data=True):
        """active.py:942-984: sample LCEs, then (maybe) the frame; returns (LCEs added, data added)."""
        self.updated = self.blind = False
        n_lces = self.update_inducing() if inducing else 0
        n_data = 0
        if self._wants_data(n_lces, inducing, data):
            exact_labels = self.blind or isinstance(self._calc, SinglePointCalculator)
            n_data = self.update_data(try_fake=not exact_labels, internal=True, save_model=False)
        if n_data == 0 and data and self.step == 0 and self.step0_forced_fp:
            self.log("forced data addition")
            self.model.add_data([self.snapshot()])
            self.log("added data: {} -> size: {} {}".format(1, *self.size))
            n_data = 1
        if n_lces > 0 or n_data > 0:
            self._after_model_change()
        self._update_args = {}
        return n_lces, n_data

    def save_model(self):
        if self.pckl and self.rank == 0:
            from .modelio import save_model
            save_model(self.pckl, self.model)

    # ------------------------------------------------------------------ training from stored data
    def _as_atoms(self, fr):
        if HAVE_ASE:  # pragma: no cover
            from ase import Atoms
            return Atoms(numbers=fr.numbers, positions=fr.positions, cell=fr.cell, pbc=fr.pbc)
        from .ase_shim import Atoms
        return Atoms(fr.numbers, fr.positions, fr.cell, fr.pbc)

    def _learn_from(self, atoms, calc):
        saved = self._calc
        self._calc = calc
        try:
            atoms.calc = self
            atoms.get_potential_energy()
        finally:
            atoms.calc = calc
            self._calc = saved

    def include_data(self, data):
        """active.py:989-1004: `data` = labelled frames (Frame objects, or atoms whose .calc holds
        energy/forces/stress); each one is offered to the learner as if met during MD."""
        for item in data:
            if isinstance(item, Frame):
                atoms = self._as_atoms(item)
                calc = SinglePointCalculator(atoms, energy=item.energy, forces=item.forces, stress=item.stress)
            else:
                atoms, calc = item, item.calc
            atoms.calc = calc
            if np.abs(atoms.get_forces()).max() > self.include_params["fmax"]:
                continue
            self._learn_from(atoms, calc)

    def include_tape(self, tape, ndata=None):
        """active.py:1006-1053: replay another run's tape — its LCEs through update_lce, its frames
        through the data-acceptance test (inducing=False)."""
        if isinstance(tape, str):
            if self.tape is not None and os.path.abspath(tape) == self.tape.path:
                raise RuntimeError("ActiveCalculator can not include it own .sgpr tape!")
            tape = SgprIO(tape, rank=self.rank)
        tune_for_md, self.tune_for_md = self.tune_for_md, False
        added_lce = [0, 0]

        def _save():
            if added_lce[0] > 0:
                if self.ioptim == 1:
                    self.optimize()
                self.save_model()
                self.log("added lone indus: {}/{} -> size: {} {}".format(*added_lce, *self.size))
                self.log("fit error (mean,mae): E: {:.2g} {:.2g}   F: {:.2g} {:.2g}   R2: {:.4g}".format(
                    *(float(v) for v in self.model._stats)))

        cdata = 0
        try:
            for cls, obj in tape.read(exclude=self.tape):
                if cls == "atoms":
                    obj.check_labels()  # a frame without energy / forces cannot be learned from (stress is optional)
                    if np.abs(obj.forces).max() > self.include_params["fmax"] and len(self.model.data) > 0:
                        continue
                    _save()
                    self._update_args = dict(inducing=False)
                    atoms = self._as_atoms(obj)
                    self._learn_from(atoms, SinglePointCalculator(atoms, energy=obj.energy, forces=obj.forces,
                                                                  stress=obj.stress))
                    cdata += 1
                    if ndata and cdata >= ndata:
                        break
                    added_lce = [0, 0]
                elif cls == "local":
                    added_lce[0] += abs(self.update_lce(obj))
                    added_lce[1] += 1
            _save()
        finally:
            self.tune_for_md = tune_for_md

    def build(self):
        """active.py:1055-1098: rebuild the model from this calculator's own tape in one shot."""
        if self.pckl and os.path.exists(self.pckl):
            raise RuntimeError(f"{self.pckl} already exists and can not be overwritten by build!"
                               " remove this file and try again.")
        blocks = self.tape.read()
        data = [obj for cls, obj in blocks if cls == "atoms"]
        lce = [obj for cls, obj in blocks if cls == "local"]
        self.model.set_data(data, lce)
        self.optimize()
        self.log("built from tape {} {} -> size: {} {}".format(len(data), len(lce), *self.size))
        self.log("fit error (mean,mae): E: {:.2g} {:.2g}   F: {:.2g} {:.2g}   R2: {:.4g}".format(
            *(float(v) for v in self.model._stats)))
        self.save_model()

    # ------------------------------------------------------------------ logging (active.py:1117-1134)
    def log_settings(self):
        self.log(f"kernel: lmax={self.engine.lmax} nmax={self.engine.nmax} exponent={self.engine.exponent} "
                 f"cutoff={self.engine.cutoff} species={list(self.engine.species)}")
        self.log(f"ediff: {self._ediff}  ediff_lb: {self._ediff_lb}  ediff_ub: {self._ediff_ub}  "
                 f"ediff_tot: {self.ediff_tot}  fdiff: {self._fdiff}  noise_f: {self.noise_f}  ioptim: {self.ioptim}")

    def log(self, mssge, mode="a"):
        if self.logfile and self.rank == 0:
            with open(self.logfile, mode) as f:
                f.write("{}{} {} {}\n".format(self._logpref, datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"),
                                              self.step, mssge))
        if self.stdout and self.rank == 0:
            print(mssge)
