/*
 * sgpr_hip.h — C ABI of libsgpr_hip.so: MI355X-native (gfx950) SGPR force-field evaluator.
 *
 * Drop-in boundary for AutoForce's predict hot path.  The reference has no
 * FFI for this path (it is pure Python/torch); each entry point below names the reference
 * interface it replaces (file:line under /root/reference/theforce).  A Python `Calculator`
 * binds these with ctypes (autoforce_amd/_lib.py; INTEGRATION.md shows the reference-side
 * stub).  Conventions:
 *   - every function returns an int status: 0 = ok, <0 = error (SGPR_E_*);
 *     sgpr_last_error() returns a static message for the calling thread's last failure;
 *   - all arrays are caller-allocated, row-major, fp64 / int32 unless stated;
 *   - `*_dev` variants take DEVICE pointers and a HIP stream and never synchronise;
 *     the plain variants take HOST pointers, copy, and synchronise before returning;
 *   - a handle is thread-compatible (one thread at a time), not re-entrant;
 *   - units: Angstrom, eV (as ASE), stress in eV/A^3, Voigt order xx,yy,zz,yz,xz,xy.
 * There is no CPU fallback: every entry point fails with SGPR_E_NODEVICE if no gfx950
 * device is usable.
 */
#ifndef SGPR_HIP_H
#define SGPR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sgpr_model sgpr_model;

enum {
    SGPR_OK = 0,
    SGPR_E_INVALID = -1,   /* bad argument */
    SGPR_E_NODEVICE = -2,  /* no usable HIP device / HIP runtime error */
    SGPR_E_NOMODEL = -3,   /* inducing set / weights not set ("you forgot to assign a DFT calculator!", calculator/active.py:429-430) */
    SGPR_E_SPECIES = -4,   /* an atomic number is missing from the model's species table */
    SGPR_E_NOT_PD = -5,    /* "cholesky was not successful!" (regression/algebra.py:45-46) */
    SGPR_E_UNSUPPORTED = -6, /* (lmax,nmax,S) not compiled in: lmax, nmax in 2..4, S <= 8 (<= 16 for lmax = nmax = 3); least squares beyond m = 8192 */
    SGPR_E_OVERFLOW = -7   /* internal capacity exceeded after retry */
};

/* Static description of the last error on this thread. */
const char *sgpr_last_error(void);

/* Library/ABI version (major*1000+minor). */
int sgpr_version(void);

/* Number of usable gfx950 devices (0 if none; never initialises a context). */
int sgpr_device_count(void);

/*
 * Create a model = kernel hyper-parameters + species table.
 * Replaces: calculator/active.py:28-38 default_kernel(lmax,nmax,exponent,cutoff) ->
 *   similarity/sesoap.py:10-24 SeSoapKernel(lmax,nmax,exponent,cutoff,radii=DefaultRadii())
 *   and descriptor/sesoap.py:102-135 SeSoap.__init__ (nnl table, radii).
 * species_z[S]: atomic numbers the model knows (the reference's 120-wide wildcard table,
 *   sesoap.py:134, restricted to those that occur); radii[S]: length unit per species
 *   (sesoap.py:84-99; DefaultRadii = 0.5 for H, 1.0 otherwise).
 */
int sgpr_create(int lmax, int nmax, double eta, double rc, int S, const int32_t *species_z,
                const double *radii, int device, sgpr_model **out);
void sgpr_destroy(sgpr_model *h);

/*
 * Set the inducing set X (m local chemical environments) and build their descriptors on
 * the device, plus K_mm.
 * Replaces: descriptor/atoms.py:36-59 Local(...).stage -> similarity/universal.py:100-107
 *   precalculate for every x in model.X, and regression/gppotential.py:506 / :764-768
 *   M = kern(X, X).
 * zc[m] central atomic numbers; nbr_ptr[m+1] CSR offsets into nbr_z[] / nbr_r[][3]
 * (neighbour atomic numbers and displacement vectors r_j - r_i, as Local._b / Local._r).
 * The library sorts X by species internally; all m-sized inputs/outputs of this API stay in
 * the CALLER's order.
 */
int sgpr_set_inducing(sgpr_model *h, int m, const int32_t *zc, const int64_t *nbr_ptr,
                      const int32_t *nbr_z, const double *nbr_r);

/* K_mm in caller order, [m][m] (regression/gppotential.py:506 self.M). */
int sgpr_get_kmm(sgpr_model *h, double *M);

/* Dense p-hat of the inducing LCEs in the reference's layout [m][S][S][D],
 * D=(nmax+1)^2(lmax+1), block [sb][sa] flattened [n][n'][l] (descriptor/sesoap.py:195-203,
 * :254-258 COO block (species[b], species[a])). Test/inspection export. */
int sgpr_get_inducing_descriptors(sgpr_model *h, double *P);
/* diag(K_mm)[m], caller order. */
int sgpr_get_kmm_diag(sgpr_model *h, double *diag);
/* Row sums of K_mm [m], caller order — `self.M.sum(dim=1)` of the downsizing rule (regression/gppotential.py:815-842,
 * lii=True) without the m x m matrix crossing the bus; summed in numpy's pairwise order over the caller's column order,
 * so the values are the bits numpy gives for sgpr_get_kmm's matrix. */
int sgpr_get_kmm_rowsum(sgpr_model *h, double *sums);

/*
 * Set the regression state used by prediction.
 * Replaces: regression/gppotential.py:548-605 make_munu outputs (mu, choli),
 *   :219-227 AutoMean weights, :644-649 _vscale.
 * mu[m]; mean_w[S] per-species constant energy (may be NULL = 0); vscale[S] (may be NULL =
 * 1; use +inf for species with no inducing point, calculator/active.py:795-800);
 * choli[m][m] = L^-1 lower-triangular, caller order (may be NULL: beta is then not computed).
 */
int sgpr_set_weights(sgpr_model *h, const double *mu, const double *mean_w, const double *vscale,
                     const double *choli);

/*
 * After a device solve (sgpr_solve / sgpr_data_solve / sgpr_resolve) mu and choli are already where prediction
 * reads them: sgpr_set_mean installs the remaining outputs of make_munu — the AutoMean weights
 * (regression/gppotential.py:219-227) and, optionally, _vscale (NULL: keep the one sgpr_make_vscale computed) —
 * without moving the m x m choli over PCIe and back.  sgpr_get_choli downloads choli[m][m] (caller's order) for the
 * callers that still want it on the host (leakage, gppotential.py:706-713; model files).
 */
int sgpr_set_mean(sgpr_model *h, const double *mean_w, const double *vscale);
int sgpr_get_choli(sgpr_model *h, double *choli);
/* The end of a rejected trial (add_1inducing / add_1atoms_fast, gppotential.py:898-982: edit, refit, measure, pop,
 * refit): after the pop the model is the one the trial started from, so the caller hands back the mu it saved instead
 * of asking for the same fit again.  The K_mm factor (hence choli) of the restored inducing set is the cached one or
 * is recomputed here; follow with sgpr_set_mean for the mean / _vscale. */
int sgpr_restore_weights(sgpr_model *h, const double *mu);

/*
 * Solve side on the device (regression/gppotential.py:1204-1339 _regression with
 * optimize=False; regression/algebra.py:29-47 jitcholesky):
 *   L, ridge = jitcholesky(K_mm); choli = L^-1; sigma = noise0*0.99*mean(diag K_mm);
 *   mu = argmin |[K; sigma L^T] mu - [Y; 0]|, K = [Ke;Kf;Kv] (rows x m, caller order).
 * On success installs mu/choli in the handle (as sgpr_set_weights) and returns them.
 * Returns SGPR_E_NOT_PD when the jitter ladder is exhausted.
 */
int sgpr_solve(sgpr_model *h, int rows, const double *K, const double *Y, double noise0,
               double *mu_out, double *choli_out, double *ridge_out, double *sigma_out);

/*
 * jitcholesky of a matrix the caller holds (regression/algebra.py:29-47): L L^T = A + ridge I with the reference's
 * ladder — ridge 0 first, then 1e-6 * mean(diag A) doubled until the factorisation succeeds; SGPR_E_NOT_PD
 * ("cholesky was not successful!", :45-46) once the ridge exceeds mean(diag A).  A: host, row-major n x n
 * (symmetric; the lower triangle is read).  L_out (may be NULL): row-major n x n, upper triangle zero.
 */
int sgpr_jitcholesky(sgpr_model *h, int n, const double *A, double *L_out, double *ridge_out);

/* The same regression for another noise value, with the design matrix of the last sgpr_solve:
 * the QR of [K | Y] is kept on the device (R, Q^T Y), the noise only enters a 2m x m second stage.
 * This is what _regression(optimize=True) evaluates repeatedly (gppotential.py:1265-1300). */
int sgpr_resolve(sgpr_model *h, double noise0, double *mu, double *choli, double *ridge, double *sigma);
/* The same for `count` (<= 64) noise values in one go, mu_out[count][m]: the grid scan of the noise search
 * (gppotential.py:1283-1296 describes it) as ONE batch of independent second-stage problems sharing every kernel
 * launch.  Evaluation only: the weights installed for prediction stay those of the last sgpr_solve / sgpr_resolve. */
int sgpr_resolve_batch(sgpr_model *h, int count, const double *noise, double *mu_out);

/*
 * Training rows of one data frame against the inducing set, on the device
 * (regression/gppotential.py:63-84 energy_energy / forces_energy / virial_energy through
 * similarity/universal.py:109-183 get_func / get_leftgrad / get_virial; used by set_data :495-497
 * and add_data :728-737):
 *   Ke[m]     = sum_i k(i,q)
 *   Kf[3N][m] = -d(sum_i k(i,q))/dx   (row 3*atom+component, caller atom order)
 *   Kv[6][m]  = sum_pairs r (x) dk/dr, Voigt xx,yy,zz,yz,xz,xy (pairs with stress * volume)
 * computed as m reverse passes with mu = e_q (exact derivative; the reference's analytic path
 * carries an fp32-rounded table and agrees to ~1e-6).  Any output may be NULL.
 */
int sgpr_kernel_rows(sgpr_model *h, int N, const int32_t *numbers, const double *positions,
                     const double *cell, const int32_t *pbc, double *Ke, double *Kf, double *Kv);

/* The same for the inducing columns [q_first, q_first + q_count) only: outputs are Ke[q_count],
 * Kf[3N][q_count], Kv[6][q_count].  This is the bordering step of add_inducing
 * (gppotential.py:745-772: one new column of K_e, K_f, K_v per data frame). */
int sgpr_kernel_columns(sgpr_model *h, int N, const int32_t *numbers, const double *positions,
                        const double *cell, const int32_t *pbc, int q_first, int q_count, double *Ke,
                        double *Kf, double *Kv);

/*
 * Resident training set: the regression's design matrix K = [K_e; K_f; K_v] of the stored data frames, kept in
 * device memory for the life of the model (the reference keeps it as torch tensors next to the model:
 * PosteriorPotential.set_data / add_data / pop_1data / popfirst_1data, regression/gppotential.py:484-509,
 * :730-743, :793-813).  Rows are frame-major: frame f owns 1 + 3N_f + nv_f consecutive rows (its energy row, its
 * force rows in the caller's atom order, then nv = 0 or 6 virial rows in Voigt order); columns are the inducing LCEs
 * in the caller's order.  The inducing-set entry points below keep the columns in step (sgpr_add_inducing computes
 * ONE new column per stored frame on the device, gppotential.py:745-772; remove / select re-index; sgpr_set_inducing
 * recomputes), so after any sequence of edits the matrix equals what sgpr_kernel_rows would return for every frame.
 *
 *   sgpr_data_push    append a frame (its rows are computed on the device and stay there)
 *   sgpr_data_pop     drop frame `index` (-1 = last: pop_1data; 0: popfirst_1data)
 *   sgpr_data_clear   drop all frames
 *   sgpr_data_info    number of frames / rows
 *   sgpr_data_matvec  out[rows] = K v   (v[m] in the caller's column order): fit residuals, k·mu of a stored frame
 *   sgpr_data_fit_stats  make_stats (gppotential.py:610-649) without the residual vector: e_pred[frames] = the
 *                     energy rows of K v, stats[7] = {sum d, sum |d|, sum d^2, sum y, sum y^2, max |y|, count} of
 *                     d = K v - Y over the force and virial rows
 *   sgpr_data_get     K[rows][m] row-major (diagnostics and tests; the product path never needs it)
 *   sgpr_data_solve   sgpr_solve on the resident matrix: Y[rows] in the same row order; with_energies = 0 drops
 *                     the energy rows (the force-only fit of _regression(optimize=True), gppotential.py:1265-1300);
 *                     sgpr_resolve re-solves it for another noise as after sgpr_solve.
 *   sgpr_data_factor  the first stage of sgpr_data_solve alone (K_mm factor, R1 and z kept on the device for
 *                     sgpr_resolve / sgpr_resolve_batch); no weights are produced or installed.
 */
int sgpr_data_push(sgpr_model *h, int N, const int32_t *numbers, const double *positions, const double *cell,
                   const int32_t *pbc, int nv);
int sgpr_data_pop(sgpr_model *h, int index);
int sgpr_data_clear(sgpr_model *h);
int sgpr_data_info(sgpr_model *h, int32_t *frames, int64_t *rows);
int sgpr_data_matvec(sgpr_model *h, const double *v, double *out);
int sgpr_data_fit_stats(sgpr_model *h, const double *v, const double *Y, double *e_pred, double *stats);
/* mae_out[b] = mean |K_f V[b] - Y_f| over the force rows of the resident matrix, for `count` weight vectors V[count][m]
 * against the targets Y[rows] (store order): the objective of the noise search of _regression(optimize=True)
 * (regression/gppotential.py:1265-1300), reduced on the device in a fixed order, sixteen vectors per pass over the matrix. */
int sgpr_data_force_mae(sgpr_model *h, int count, const double *V, const double *Y, double *mae_out);
int sgpr_data_get(sgpr_model *h, double *K);
int sgpr_data_solve(sgpr_model *h, const double *Y, int with_energies, double noise, double *mu_out,
                    double *choli_out, double *ridge_out, double *sigma_out);
int sgpr_data_factor(sgpr_model *h, const double *Y, int with_energies);
/* Diagnostics of the last sgpr_data_solve / sgpr_data_factor: which route the first stage took ("full factorisation",
 * "columns appended / popped through the kept reflectors", "kept", "rows of the new frame appended to the kept
 * factor", "found in the cache", "from scratch") and how many species blocks of K_mm were factored, as text
 * ("stage1=...; stage2=...; kmm_blocks=a/b; rows=...": stage2 = "factorisation" or, for the model a kept second stage was
 * made for asked again / with one appended column, "kept reflectors, ...").  The reference always refits from scratch (gppotential.py:548-605); tests use
 * this to assert that an edit (append, pop, select / downsize) was followed incrementally.  "rows=" names the kernel
 * form of the last rows / columns call: "sixteen columns per workgroup pass" (lmax and nmax <= 3, <= 4 species, lists of
 * <= 64 neighbours) or "one column per wave". */
int sgpr_solve_info(sgpr_model *h, char *buf, int cap);

/*
 * Inducing-set edits (PosteriorPotential.add_inducing / pop_1inducing / popfirst_1inducing /
 * select_inducing, gppotential.py:745-842, :1037-1046).  The caller's order is kept: a new LCE is
 * appended as index m; `remove` deletes one index (-1 = last); `select` keeps `indices` in the
 * given order (downsize(lii=True), gppotential.py:829-832, selects the max_inducing LCEs with the smallest K_mm
 * row sums in argsort order).  All three are incremental on the device: descriptors, K_mm and the resident design
 * matrix are re-indexed (one new row / column for an appended LCE), the cached K_mm factor is bordered (append),
 * trimmed (pop) or re-factored per species block (select), and the kept first-stage QR of [K | Y] follows
 * through its reflectors; weights are invalidated (call sgpr_solve / sgpr_data_solve / sgpr_set_weights next, as
 * the reference calls make_munu).
 */
int sgpr_add_inducing(sgpr_model *h, int32_t zc, int nn, const int32_t *nbr_z, const double *nbr_r);
int sgpr_remove_inducing(sgpr_model *h, int index);
int sgpr_select_inducing(sgpr_model *h, int count, const int32_t *indices);

/* k(loc, X)[m] and k(loc, loc) for one LCE that is not (yet) in the inducing set
 * (gp.kern(loc, X) in ActiveCalculator.update_lce, active.py:806-818, and
 * PosteriorPotential.leakage, gppotential.py:706-713).  Either output may be NULL. */
int sgpr_kernel_local(sgpr_model *h, int32_t zc, int nn, const int32_t *nbr_z, const double *nbr_r,
                      double *k_out, double *kxx_out);

/* vscale[S] = mean_{q: Z_q = z} mu_q (K_mm mu)_q (regression/gppotential.py:644-649);
 * +inf where a species has no inducing point. Installs it in the handle as well. */
int sgpr_make_vscale(sgpr_model *h, double *vscale_out);

/*
 * One prediction = one MD step of the hot path.
 * Replaces: calculator/active.py:425-502 ActiveCalculator.calculate ->
 *   descriptor/atoms.py:384-413 TorchAtoms.update (neighbour list :348-363, Local :365-382,
 *   descriptors sesoap.py:161-260), regression/gppotential.py:63-84 K_nm,
 *   calculator/active.py:548-611 energy / autograd forces / stress, :781-804 covloss.
 * Inputs (host): numbers[N], positions[N][3], cell[3][3] (rows = lattice vectors), pbc[3].
 * shard: atoms are dealt to `world` ranks; this call evaluates rank `rank`'s share only and
 *   returns PARTIAL sums (energy without the mean term on rank != 0, forces on all N atoms
 *   from this rank's LCEs, beta zero outside the share) to be summed over ranks
 *   (calculator/active.py:562,600-602,770-777).  world = 1 for a single process.
 * Outputs (host, any may be NULL): energy[1]; forces[N][3]; stress[6]; beta[N] =
 *   covloss (already scaled by sqrt(vscale)); cov[N][m] = K_nm rows in caller order (zero
 *   rows outside the share).
 */
int sgpr_compute(sgpr_model *h, int N, const int32_t *numbers, const double *positions,
                 const double *cell, const int32_t *pbc, int rank, int world, double *energy,
                 double *forces, double *stress, double *beta, double *cov);

/* sgpr_compute without the copies out: *packed_out points at this call's results where the device wrote them — page-locked
 * host memory owned by the handle, [F 3N | beta N | E | virial 9 (row-major) | overflow word | stress 6 (Voigt)], caller atom
 * order — valid until the call AFTER THE NEXT on this handle (two buffers alternate: the previous call's results stay
 * intact while this one runs; a buffer that a larger frame replaces is kept for one more generation).  N >= 1 (an empty
 * frame has nothing to point at: sgpr_compute returns its zeros).  What ActiveCalculator.results hands out as views (the reference's results are views of
 * tensors the calculator owns, calculator/active.py:572-574; ASE copies what it passes on). */
int sgpr_compute_view(sgpr_model *h, int N, const int32_t *numbers, const double *positions, const double *cell,
                      const int32_t *pbc, int rank, int world, const double **packed_out);

/*
 * Device-resident form for MD loops and benchmarks: no host copies, no synchronisation.
 * sgpr_bind_system fixes N, numbers, pbc and the sharding (host arrays; may be called again
 * when they change); sgpr_step_dev consumes DEVICE positions[N][3] + cell[9] and fills the
 * DEVICE buffer `packed` (length sgpr_packed_len(N) doubles):
 *   packed[0..3N)      forces (partial over ranks)
 *   packed[3N..4N)     beta   (this rank's share, 0 elsewhere)
 *   packed[4N]         energy (partial; mean term added on rank 0 only)
 *   packed[4N+1..+9)   virial sum_pairs r (x) dE/dr (partial), row-major 3x3
 *   packed[4N+10]      1 if this rank's step overflowed a capacity (results invalid), else 0
 * (Sharded steps accumulate the forces neighbours receive as 64-bit fixed-point integers, 2^-46 eV/A per unit, so that a
 * rank's partial sums do not depend on the order the atomics land in: a single pair force beyond 1024 eV/A — two atoms
 * unphysically close — is outside that format and fails the step with SGPR_E_OVERFLOW; the unsharded path has no such limit.)
 * i.e. exactly what one all-reduce(SUM) over ranks must combine (the reference's four
 * collectives calculator/active.py:562,601,602,777 fused into one buffer).  With a communicator
 * attached (sgpr_comm_init) the step ends with that all-reduce, enqueued on the same stream: the
 * buffer then holds the TOTALS on every rank, and packed[4N+10] > 0 tells every rank alike that some
 * rank must repeat the step (sgpr_sync_check on each rank, then the step again).
 * `stream` is a hipStream_t passed as void*.  With option "graph" the step is captured into a HIP graph on first
 * use and replayed afterwards.
 */
int sgpr_bind_system(sgpr_model *h, int N, const int32_t *numbers, const int32_t *pbc, int rank,
                     int world);
int64_t sgpr_packed_len(int N);
int sgpr_step_dev(sgpr_model *h, const double *positions_dev, const double *cell_dev,
                  double *packed_dev, void *stream);
/*
 * sgpr_step_dev with the DEVICE positions of the step that follows named (same cell buffer): the last kernel of this
 * step bins them and takes the rebuild decision of the next step's neighbour candidates, so the next call — which must
 * pass exactly `positions_next_dev` — starts with the list filter instead of a binning launch (6 -> 5 launches per
 * step).  A call that passes anything else is served as usual.  positions_next_dev = NULL: sgpr_step_dev.
 * The next step is binned with the cell AS IT IS NOW: a driver that changes the contents of `cell_dev` between two steps
 * (NPT) passes NULL for the step in front of the change — every step of such a run bins for itself.  (A chain that runs in
 * another cell than the one its candidate lists were built in notices — the last kernel compares the two — and rebuilds
 * them at its next step.)
 * (The reference asks ASE for a fresh list inside every calculate(), descriptor/atoms.py:348-363, :402.)
 */
int sgpr_step_dev_next(sgpr_model *h, const double *positions_dev, const double *cell_dev,
                       double *packed_dev, const double *positions_next_dev, void *stream);

/*
 * Device-resident molecular dynamics.  The reference integrates in ASE — ase.md.langevin.Langevin around the
 * calculator (cl/md.py:117-128), velocities from util/aseutil.py:11-20 — and crosses into calculate() once per step,
 * whose covloss gate (calculator/active.py:492-499, :842-885: update when the largest covloss reaches ediff) decides
 * whether the model is edited.  Here positions and velocities live in device memory, the integrator (BAOAB Langevin;
 * friction = 0: velocity Verlet) is part of the step's last kernel, and the gate halts the run on the device.
 *
 * sgpr_md_begin   binds the system (single process), uploads positions[N][3], velocities[N][3] (NULL: zero) and
 *                 masses[N] (caller atom order), fixes dt, friction (per unit time) and kT (energy units):
 *                 c1 = exp(-friction dt), sigma_i = sqrt(1 - c1^2) sqrt(kT / m_i).
 * sgpr_md_run     evaluates `nevals` configurations starting with the current one.  After every evaluation (with
 *                 final_eval != 0: every one but the last) the state moves on:
 *                     v += (dt/2) F/m;  x += (dt/2) v;  v = c1 v + sigma xi;  x += (dt/2) v;  [evaluate]  v += (dt/2) F/m
 *                 with xi the next row of noise[nevals][N][3] (standard normal deviates, caller atom order; NULL: none).
 *                 scalars[nevals][16] (may be NULL) receives per evaluation: E, virial[9] (row-major), the capacity
 *                 overflow word, the largest covloss, sum_i m_i v_i^2 with the velocities AFTER the closing half kick and
 *                 BEFORE it (what a calculator inside the integrator's step is handed, cl/md.py:117-128), 0, 0.
 *                 The run stops at the first evaluation whose largest covloss is >= ediff (ediff <= 0: never):
 *                 *evals_done then counts that evaluation as the last one, *halt_code = 1, and the state IS that
 *                 configuration — the next sgpr_md_run evaluates it again, with whatever model the caller has installed
 *                 meanwhile, exactly as the reference recomputes the results after an update
 *                 (calculator/active.py:477-484).  *halt_code = 2: a neighbour capacity overflowed at evaluation
 *                 *evals_done (not counted); the next call re-sizes and repeats it.
 * sgpr_md_state   positions[N][3], velocities_pre[N][3] (before the closing half kick; *pending says whether one is
 *                 due) of the current configuration (which = 0) or of the one before it (which = -1), and the packed
 *                 results (sgpr_step_dev layout, length sgpr_packed_len(N)) of that configuration's last evaluation;
 *                 any pointer may be NULL.
 * Same-seed runs are bit-reproducible; with the same noise rows the trajectory equals the host-side loop around
 * sgpr_compute bit for bit (tests/test_hip_md.py).
 */
int sgpr_md_begin(sgpr_model *h, int N, const int32_t *numbers, const double *positions, const double *cell,
                  const int32_t *pbc, const double *masses, const double *velocities, double dt, double friction,
                  double kT);
int sgpr_md_run(sgpr_model *h, int nevals, const double *noise, double ediff, int final_eval, double *scalars,
                int *evals_done, int *halt_code);
int sgpr_md_state(sgpr_model *h, double *positions, double *velocities_pre, int *pending, double *packed, int which);
/* The velocities an observer sees at the current configuration (evaluated by the last sgpr_md_run: a halt, or final_eval):
 * the closing half kick applied (Langevin / velocity Verlet), or the centred velocity (Nose-Hoover, sgpr_md_thermostat — there
 * sgpr_md_state's velocities_pre of configuration n is v_(n-1): what the integrator holds when it asks for F_n). */
int sgpr_md_velocities(sgpr_model *h, double *velocities);
int sgpr_md_end(sgpr_model *h);
/* Deviates drawn on the device: with a seed != 0, sgpr_md_run called with noise = NULL draws xi itself — deviate
 * (configuration index, atom, component) of a counter-based generator (Philox4x32-10, Box-Muller), so a run does not
 * depend on how it is cut into calls and an evaluation repeated after a halt draws the same numbers; no host generator
 * and no upload on the step's path (the reference draws from numpy inside ase.md.langevin, cl/md.py:117-128).
 * sgpr_md_deviates returns the rows of configurations [t_first, t_first + count): out[count][N][3], caller atom order. */
int sgpr_md_seed(sgpr_model *h, uint64_t seed);
/* Nose-Hoover NVT instead of the Langevin / velocity-Verlet step (kind = 1; 0 = back): the reference's DEFAULT dynamics,
 * md(dynamics="NPT", bulk_modulus=None) = ase.md.npt.NPT with pfactor = None, ttime = tdamp fs (cl/md.py:17, :131-166;
 * Melchionna, Ciccotti, Holian 1993 as ASE integrates it):
 *     x_(n+1) = (2 x_n - x_(n-1) (1 - b) + dt^2 F_n / m) / (1 + b),  b = dt zeta_n / 2,  v_n = (x_(n+1) - x_(n-1)) / 2 dt,
 *     zeta_(n+1) = zeta_(n-1) + 2 dt tfact (KE_n - 1.5 (N - 1) kT),  tfact = 2 / (3 N kT ttime^2),
 * started with x_(-1) = x_0 - dt v_0 + dt^2 F_0 / 2m, zeta_0 = 0, zeta_(-1) = -dt tfact (KE_0 - ...).  The integrator stays
 * in the step's last kernel; zeta_(n+1) needs the kinetic energy of ALL atoms at step n, reduced by one small launch
 * behind each evaluation.  Call between sgpr_md_begin and the first sgpr_md_run.  scalars[.][12] = [13] = sum m v_n^2 with
 * the centred velocities, [14] = zeta_n, [15] = its time integral (the conserved quantity is
 * E + KE + 1.5 N kT (ttime zeta)^2 + 3 (N - 1) kT int zeta dt, ASE's get_gibbs_free_energy).  sgpr_md_state returns the
 * centred velocities (pending = 0). */
int sgpr_md_thermostat(sgpr_model *h, int kind, double ttime, double kT);
int sgpr_md_deviates(sgpr_model *h, int64_t t_first, int count, double *out);
/*
 * Multi-GPU (one process per GPU, atoms sharded as in sgpr_bind_system): the reference combines the
 * ranks' partial sums with four MPI all-reduces per step (calculator/active.py:562,601,602,777,
 * util/parallel.py); here ONE RCCL all-reduce of the packed buffer over xGMI, issued by the library
 * on the step's stream.  Rank 0 creates the id (SGPR_COMM_ID_BYTES bytes), the host passes it to the
 * other ranks by any means (MPI, a TCP store, a file), and every rank calls sgpr_comm_init
 * (collective).  Without a communicator the caller combines the partial sums itself.
 * sgpr_comm_allreduce: in-place SUM (op_max = 0) or MAX (1) of a device buffer of doubles over the
 * same communicator (barriers, max-over-ranks timings).
 */
#define SGPR_COMM_ID_BYTES 128
int sgpr_comm_unique_id(void *id_out);
int sgpr_comm_init(sgpr_model *h, const void *id, int rank, int world);
int sgpr_comm_destroy(sgpr_model *h);
int sgpr_comm_allreduce(sgpr_model *h, double *buf_dev, int64_t count, int op_max, void *stream);

/*
 * The library's own exchange between the ranks of ONE node — an alternative to RCCL that also runs when several ranks
 * share one device (RCCL refuses that) and that gives every rank the SAME bits: every rank owns a receive buffer with
 * one slice per source rank, exported through hipIpcGetMemHandle; a step's partial sums are pushed into the slice of every
 * peer (one hop on the fully connected xGMI mesh: the all-gather BASELINE.json words, the zero-padded all-reduces of
 * calculator/active.py:770-777), released by one flag store per peer, and summed locally in rank order.  The force sums of
 * the sharded reverse pass travel as fixed-point integers and are added as integers: the total force on an atom is the same
 * bits for every number of ranks.
 *   sgpr_peer_export   allocates this rank's buffers for `world` ranks and `capacity` doubles per slice (a sharded step of N
 *                      atoms needs 3 N + 4 ceil(N / world) + 11; 7 N + 11 serves every world size) and writes
 *                      SGPR_PEER_HANDLE_BYTES bytes for the peers
 *   sgpr_peer_attach   handles[world][SGPR_PEER_HANDLE_BYTES] of all ranks in rank order (the host passes them by any means);
 *                      from then on sgpr_compute / sgpr_step_dev / sgpr_step_dev_next combine a sharded step through this
 *                      exchange (preferred over a communicator of sgpr_comm_init), sgpr_comm_allreduce works over it, and
 *                      sgpr_md_begin / sgpr_md_run run SHARDED: every rank evaluates its share, and after the exchange
 *                      integrates all atoms from the summed forces (the deviates are counter-based or uploaded alike), so the
 *                      ranks' states stay identical; the covloss gate and capacity overflows halt every rank at the same step.
 * A peer that never arrives is a time-out (SGPR_PEER_TIMEOUT_MS, default 30000: a rank's first sharded step may grow
 * capacities and load code objects while its peers already wait), not a hang: the consumer of a timed-out exchange marks the
 * step's results (poison in the overflow word; NaN in a buffer of sgpr_comm_allreduce), so the call that reads them —
 * sgpr_compute / sgpr_compute_view on their warm path included, sgpr_sync_check, sgpr_md_run — fails with the time-out's
 * message.  A time-out is permanent: every later exchange of the handle fails at once until sgpr_peer_export + sgpr_peer_attach
 * have been called again on every rank.  Exchanges are collective: every rank issues the same sequence of them, one after
 * the other (an exchange issued on another stream than the one before is ordered behind it).
 */
#define SGPR_PEER_HANDLE_BYTES 128
int sgpr_peer_export(sgpr_model *h, int rank, int world, int64_t capacity, void *handle_out);
int sgpr_peer_attach(sgpr_model *h, const void *handles);
int sgpr_peer_destroy(sgpr_model *h);

/* Synchronise `stream` (NULL = the handle's own) and verify that no step since the last
 * check overflowed the neighbour-list capacity.  Returns SGPR_E_OVERFLOW if one did (those
 * steps' results are invalid; capacity has been grown, the next step re-sizes eagerly). */
int sgpr_sync_check(sgpr_model *h, void *stream);
/* Options:
 *  "graph" = 0/1            replay sgpr_step_dev from a captured HIP graph (default 0: eager launches pipeline
 *                           well while a step is ~100 us of kernels and measured faster than replay)
 *  "qr_keep" = 1/0/2        the first-stage factorisation of sgpr_data_solve is kept with its reflectors, so that
 *                           an appended / popped inducing LCE and a pushed frame update it instead of refactoring
 *                           (default 1; 0: every refit factors from scratch, as the reference does,
 *                           gppotential.py:745-791; 2: both, the difference of mu printed on stderr).  The
 *                           environment variable SGPR_QR_KEEP sets the initial value.
 *  "skin_milliangstrom"     Verlet skin of the neighbour candidates (default 500 = 0.5 A).  The reference asks ASE
 *                           for a list with skin 0 at every step (descriptor/atoms.py:349-355); here candidate
 *                           lists of |r| < rc + skin are kept and rebuilt ON THE DEVICE whenever an atom's
 *                           displacement since the last build — measured against the affine image of its build-time
 *                           position when the cell has changed (NPT: cl/md.py:147-150) — exceeds
 *                           (sigma_min(h0^-1 h) (rc + skin) - rc) / 2 (= skin/2 at constant cell), and every step
 *                           filters them to |r| < rc: the same pairs in the same order as a from-scratch list, bit
 *                           for bit.  0 = rebuild every step
 *  "ignore_unknown_species" = 0/1  atoms and LCE neighbours outside the species table are invisible
 *                           (descriptor/sesoap.py:343-346) instead of SGPR_E_SPECIES
 *  "overlap" = 0/1          covloss product on a side stream (measured slower; default 0)
 *  "cov_in_rev" = 0/1       covloss tiles in the reverse kernel's launch instead of grouped with the W product
 *                           (measured slower at 4096 atoms: default 0).  Environment: SGPR_COV_IN_REV
 *  "zero_copy_out" = 1/0    single-rank sgpr_compute: the step's last kernel writes its results into mapped
 *                           page-locked host memory instead of a device buffer that is copied back (default 1).
 *                           Environment: SGPR_ZERO_COPY
 *  "lone_atom_weight" = k   k(x, x') of two lone atoms (no neighbour inside the cutoff) of one species: the
 *                           reference adds that term once per kernel OBJECT (similarity/similarity.py:38-40, :94-103)
 *                           and sums its kernels (regression/gppotential.py:63-84) — 1 for the wildcard kernel
 *                           (default), S for the list of S fixed-species kernels of calculator/active.py:31-38.
 *                           Set before the inducing set.
 *  "fuse_next" = 1/0        the last kernel of a step may open the next one (sgpr_step_dev_next, sgpr_md_run;
 *                           default 1; 0: every step bins for itself — sgpr_md_run then returns SGPR_E_UNSUPPORTED).
 *                           Environment: SGPR_FUSE_NEXT
 *  "gemm_fused" = 1/0       K_nm, W and covloss of a step in ONE launch (consumer tiles wait on per-panel counters;
 *                           same bits, measured slower at 4096 atoms: default 0).  Environment: SGPR_GEMM_FUSED
 *  "reverse_scatter" = 0/1  the scatter form of the reverse pass (fixed-point force sums; what sharded frames use) on a
 *                           single rank too: the twin a sharded run is compared with bit for bit (default 0: gather form)
 *  "spin_wait" = 1/0        sgpr_compute polls its stream for the end of a step instead of a blocking wait
 *                           (default 1: one host thread spins for the ~0.1 ms of a step, 16 us less wall time per
 *                           call on a 4096-atom frame; 0: hipStreamSynchronize).  Environment: SGPR_SPIN_WAIT */
int sgpr_set_option(sgpr_model *h, const char *name, int value);
/* How many times the neighbour candidates were rebuilt since sgpr_create (see "skin_milliangstrom"). */
int sgpr_get_list_rebuilds(sgpr_model *h, int64_t *count);
/* stress[6] (Voigt, eV/A^3) from a (summed) packed buffer on the host:
 * calculator/active.py:604-610, volume = |det cell| or -2 for a rank-deficient cell. */
int sgpr_stress_from_virial(const double *virial9, const double *cell, double *stress6);

/* Inspection/test exports for the last sgpr_compute / sgpr_step_dev on this handle.
 * p: dense p-hat [N][S][S][D] in the reference layout (see sgpr_get_inducing_descriptors).
 * nl: neighbour list in CSR form; call with j == NULL to get only ptr[N+1]. */
int sgpr_get_descriptors(sgpr_model *h, double *P);
int sgpr_get_neighbors(sgpr_model *h, int64_t *ptr, int32_t *j, int32_t *off);

/* One atom's LCE (neighbour numbers and displacement vectors x_j - x_i + off.cell, descriptor/atoms.py:
 * 365-382 `TorchAtoms.local`) out of the neighbour list of the last evaluated frame, without moving
 * the whole list to the host (the cell is read from the device buffer that step used).
 * nbr_z / nbr_r may be NULL to ask for the count only. */
int sgpr_get_local(sgpr_model *h, int atom, int32_t *nn, int32_t *nbr_z, double *nbr_r, int capacity);

/* K_nm [N][m] of the last evaluated frame (caller order; rows of other ranks' atoms are zero): the
 * `cov` output of sgpr_compute, fetched only when somebody looks at it (calc.cov, active.py:464).
 * N and m are the dimensions the caller's buffer was sized for: SGPR_E_INVALID if the last frame the
 * device evaluated (training rows and trial models rebind it) has other dimensions. */
int sgpr_get_cov(sgpr_model *h, int N, int m, double *cov);

/* Model dimensions, out[8]: out[0]=m, out[1]=S, out[2]=D (dense per block), out[3]=Dc (packed
 * row length used on the device), out[4]=neighbour capacity per atom, out[5]=N bound,
 * out[6]=largest neighbour count seen at the last checked step, out[7]=padded row stride. */
int sgpr_get_dims(sgpr_model *h, int32_t *out);

/* Per-kernel device timings (ms) of the last profiled step; enables HIP-event timing
 * around each stage when `on` != 0 (adds synchronisation: never leave on in production).
 * names: semicolon-separated stage names; returns the number of stages. */
int sgpr_profile(sgpr_model *h, int on);
int sgpr_get_stage_times(sgpr_model *h, double *ms, int cap, char *names, int names_cap);

#ifdef __cplusplus
}
#endif
#endif /* SGPR_HIP_H */
