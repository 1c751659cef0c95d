// api.hip — C ABI (include/sgpr_hip.h) and step orchestration of the gfx950 SGPR evaluator.
//
// One handle = one model on one GPU, one HIP stream.  A step is the fixed launch sequence
//   nl_bin -> nl_build -> desc_fwd -> gemm<KERNEL> -> gemm<STORE> -> desc_dc -> desc_pair
//          -> gemm<ROWSQ> -> finalize
// launched eagerly (or, with option "graph", captured once into a HIP graph and replayed).
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <numeric>

#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>  // types only: librccl is opened on the first sgpr_comm_* call (single-GPU installs need none)

#include "../../include/sgpr_hip.h"
#include "sgpr_internal.h"
#include <cmath>
#include <limits>

static thread_local char g_err[512] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}
#define HIPCHK(x)                                                                                   \
    do {                                                                                            \
        hipError_t e_ = (x);                                                                        \
        if (e_ != hipSuccess) return fail(SGPR_E_NODEVICE, "%s: %s", #x, hipGetErrorString(e_));    \
    } while (0)

#define SGPR_PEER_POISON 1e9  // overflow word of a rank whose step failed outright (summed by the all-reduce)
static inline int rup(int x, int q) { return (x + q - 1) / q * q; }

// RCCL entry points, resolved on first use: a single-GPU process never loads the library
struct RcclApi {
    void *lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};
static RcclApi g_rccl;
static int rccl_load()
{
    if (g_rccl.lib) return SGPR_OK;
    // A copy that is ALREADY mapped wins (a process that imported torch has torch's bundled librccl.so.1: opening the
    // unversioned name beside it could map a second, different RCCL with global symbol interposition).  Only when
    // nothing is loaded is a library opened, and then with local binding.
    void *lib = nullptr;
    const char *loaded[] = {"librccl.so.1", "librccl.so"};
    for (const char *n : loaded)
        if ((lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!lib) {
        const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char *n : names)
            if ((lib = dlopen(n, RTLD_NOW | RTLD_LOCAL))) break;
    }
    if (!lib) return fail(SGPR_E_NODEVICE, "RCCL is not available (dlopen librccl.so: %s): multi-GPU runs need it", dlerror());
    RcclApi a;
    a.lib = lib;
    a.GetUniqueId = (decltype(a.GetUniqueId))dlsym(lib, "ncclGetUniqueId");
    a.CommInitRank = (decltype(a.CommInitRank))dlsym(lib, "ncclCommInitRank");
    a.CommDestroy = (decltype(a.CommDestroy))dlsym(lib, "ncclCommDestroy");
    a.AllReduce = (decltype(a.AllReduce))dlsym(lib, "ncclAllReduce");
    a.GetErrorString = (decltype(a.GetErrorString))dlsym(lib, "ncclGetErrorString");
    if (!a.GetUniqueId || !a.CommInitRank || !a.CommDestroy || !a.AllReduce || !a.GetErrorString) {
        dlclose(lib);
        return fail(SGPR_E_NODEVICE, "librccl.so lacks an expected entry point");
    }
    // the types compiled in (ncclUniqueId, the enums) are those of <rccl/rccl.h>: the library must be of that major version
    if (auto get_version = (ncclResult_t(*)(int *))dlsym(lib, "ncclGetVersion")) {
        int v = 0;
        if (get_version(&v) == ncclSuccess && v / 10000 != NCCL_VERSION_CODE / 10000) {
            dlclose(lib);
            return fail(SGPR_E_NODEVICE, "the RCCL in this process is version %d, the library was built against %d: rebuild "
                        "libsgpr_hip.so against the installed rccl.h", v, (int)NCCL_VERSION_CODE);
        }
    }
    g_rccl = a;
    return SGPR_OK;
}


template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    int alloc(size_t count, bool zero = true)
    {
        if (count > n || !p) {
            if (p) (void)hipFree(p);
            p = nullptr;
            n = 0;
            if (hipMalloc((void **)&p, std::max<size_t>(count, 1) * sizeof(T)) != hipSuccess) return -1;
            n = count;
        }
        if (zero && count) (void)hipMemset(p, 0, count * sizeof(T));
        return 0;
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
};
// a DevBuf that is a LOCAL of one entry point: given back on every way out of the scope (the members of the handle have
// no destructor on purpose: sgpr_destroy releases them in its own order)
template <typename T>
struct ScopedBuf : DevBuf<T> {
    ScopedBuf() = default;
    ScopedBuf(const ScopedBuf &) = delete;
    ScopedBuf &operator=(const ScopedBuf &) = delete;
    ~ScopedBuf() { this->release(); }
};

// The library's own exchange between the ranks of ONE node (peer.inc): every rank owns a receive buffer
// recv[2][world][cap] (two epochs' parities x one slice per source rank) and one flag per source rank, exported through
// hipIpcGetMemHandle and mapped by every peer; a step's partial sums are PUSHED into the slice `rank` of every peer (posted
// writes over xGMI: an all-gather in one hop on the fully connected mesh), released with one flag store per peer, and summed
// locally in rank order — the same bits on every rank, whatever the arrival order.
#define SGPR_PEER_MAX 16
struct PeerXchg {
    int world = 1, rank = 0;
    bool attached = false;
    size_t cap = 0;                        // doubles per slice
    size_t bytes = 0, flag_off = 0;
    char *base = nullptr;                  // own allocation: recv | flags
    char *peer_base[SGPR_PEER_MAX] = {};   // every rank's allocation as mapped here (own: base)
    bool opened[SGPR_PEER_MAX] = {};       // mapped through hipIpcOpenMemHandle (to be closed)
    unsigned epoch = 0;                    // exchanges so far (the same on every rank: exchanges are collective)
    DevBuf<int> ctl;                       // [SGPR_PEER_MAX + 2] lines of 32 ints: push counters per peer | error word | dead word
    long long timeout_ticks = 3000000000LL; // bounded spin of the wait kernel: 30 s of 100 MHz ticks (SGPR_PEER_TIMEOUT_MS) — a rank's first
                                           // sharded step may grow capacities and load code objects while its peers already wait
    hipStream_t last_st = nullptr;         // the stream of the last exchange; an exchange on ANOTHER stream is ordered behind it (ev):
    bool any_st = false;                   // the parity scheme and the push counters assume this handle's exchanges run one after the other
    hipEvent_t ev = nullptr;
    const double *slice(int parity, int r) const { return (const double *)base + ((size_t)parity * world + r) * cap; }
};

// device-resident molecular dynamics (sgpr_md_*): positions / velocities / results of the last three evaluations in
// rings (sorted atom order), see FinNext
struct MdState {
    bool active = false;
    int N = 0;
    // the system the run was begun on: another evaluation on the handle in between (a model update computes training rows
    // of stored frames, trial models rebind) leaves the handle bound to something else — sgpr_md_run binds it back
    std::vector<int32_t> numbers;
    std::vector<int> perm;          // sorted -> caller of THAT system (sgpr_md_state does not depend on the binding)
    int32_t pbc[3] = {1, 1, 1};
    int rank = 0, world = 1;
    long long t = 0;               // evaluations completed (= index of the configuration to evaluate next)
    double hdt = 0.0, c1 = 1.0, dt = 0.0;
    int ring = 3;                  // slots of the X / V / P / KE rings in use: 3 (Langevin / velocity Verlet), 4 (Nose-Hoover: the
                                   //   speculative step behind a halt must not overwrite x_(k-1), which the restart needs)
    // Nose-Hoover NVT (sgpr_md_thermostat): zeta_(n+1) = zeta_(n-1) + 2 dt tfact (KE_n - K0), rings of four in `zeta`
    bool nh = false;
    double nh_c1 = 0.0, nh_c2 = 0.0, nh_K0 = 0.0;   // dt tfact, 2 dt tfact, desired kinetic energy
    DevBuf<double> zeta;           // [4] zeta by evaluation index & 3 | [4] its time integral
    bool evaluated = false;        // the current configuration has been evaluated by the last sgpr_md_run (halted / final)
    // a run that is cut into several sgpr_md_run calls goes on where the last call stopped — candidate lists, the bins its last
    // kernel filled for the next configuration — when nothing else has touched the handle in between
    bool chain_ok = false;
    unsigned chain_step = 0, chain_bind = 0, chain_opt = 0;
    long long chain_t = 0;
    const double *chain_pos = nullptr;
    double *scal_pin = nullptr;    // [rows][SGPR_MD_SCAL] page-locked: the scalars of a call travel behind its last kernel, one wait
    unsigned long long seed = 0;   // != 0: the integrator draws its own deviates (sgpr_md_seed)
    DevBuf<double> X, V, P, KE, mass, sig, noise, noise_raw, cell;
    DevBuf<int> halt;
    int *halt_host = nullptr, *halt_host_dev = nullptr;
    DevBuf<double> scal_d;                    // [rows][SGPR_MD_SCAL] per-evaluation scalars, device memory (copied out once per run)
    int *mark = nullptr, *mark_dev = nullptr;  // [rows] mapped host memory: evaluation j has passed its last kernel (look-ahead throttle)
    size_t scal_rows = 0;
    std::vector<double> mass_sorted;
};

struct sgpr_model {
    int lmax, nmax, S, device;
    double eta, rc;
    std::vector<int> species;
    std::vector<double> radii;
    hipStream_t stream = nullptr, side = nullptr;
    // multi-GPU: one process per GPU, the packed partial sums of a step are combined by ONE RCCL
    // all-reduce enqueued on the step's stream (sgpr_comm_init)
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    PeerXchg peer;                 // the hipIpc all-gather exchange (sgpr_peer_*): preferred over RCCL when attached
    DevBuf<double> d_xpacked;      // a sharded step's partial sums in the exchange layout (peer.inc)
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    // packed layout
    int D, Dc, Dpad, CS;
    std::vector<PackEntry> h_pack;
    DevBuf<PackEntry> d_pack;
    DevBuf<double> d_radii;
    // inducing set (device arrays in species-sorted order)
    int m = 0, m_pad = 0, m_rows = 0;
    std::vector<int32_t> env_zc, env_z;  // the caller's LCE list (caller order), kept for add/remove/select
    std::vector<int64_t> env_ptr;
    std::vector<double> env_r;
    std::vector<int> ind_perm;  // sorted -> caller
    std::vector<int> ind_slot;  // sorted
    std::vector<int> qoff;      // [S+1]
    DevBuf<int> d_ind_slot, d_ind_nn, d_qoff;
    DevBuf<double> d_Pm, d_PmT, d_pm_norm, d_M, d_mu, d_choli;
    bool has_mu = false, has_choli = false, choli_lower = true;
    const double *rows_mu = nullptr;  // sgpr_kernel_rows: unit weights for the K_nm pass, no reverse pass
    std::vector<double> mean_w, vscale;
    DevBuf<double> d_vs_sqrt;  // sqrt(vscale) per slot
    // bound system
    int N = 0, N_rows = 0, rank = 0, world = 1, cnt = 0, cnt_rows = 0;
    int pbc[3] = {1, 1, 1};
    std::vector<int> numbers, perm /*sorted->caller*/, slot_sorted, aoff /*local [S+1]*/;
    double mean_energy = 0.0;
    DevBuf<int> d_perm, d_slot, d_aoff, d_lslot /*local rows*/, d_lnn;
    DevBuf<double> d_pos_in, d_cell_in, d_pos;
    const double *last_cell = nullptr;  // device cell of the last enqueued step (sgpr_get_local)
    // neighbour list
    int maxnn = 0, nn_max_seen = 0;
    bool warm = false;  // a synchronised, capacity-checked step has run since the last bind
    bool zero_copy_out = true;  // sgpr_compute's warm path: finalize writes the results into mapped host memory
    bool spin_wait = true;   // sgpr_compute's warm path polls the stream instead of a blocking wait (option "spin_wait"; -16 us per call)
    DevBuf<char> d_grid;
    DevBuf<int> d_bin_of, d_bin_count, d_nn, d_nbr_j, d_nbr_shift, d_stat;
    int bin_cap = 0;  // slots per bin of the binned copies
    DevBuf<BinRec> d_b_rec;
    DevBuf<BinAux> d_b_aux;
    DevBuf<int> d_nn_raw;
    // reverse pass in gather form (single-process frames): pair records from the forward pass, pair
    // gradients G, and the reverse index (neighbor.hip) that lets the last kernel sum them per atom
    DevBuf<int> d_kslot, d_aux;
    DevBuf<unsigned short> d_T;
    int t_stride = 0;
    // Verlet candidates (lists of |r| < rc + skin, rebuilt on the device when needed; descriptor.hip)
    double skin = 0.5;          // Angstrom; 0: rebuild every step (option "skin_milliangstrom")
    bool lists_valid = false;   // false: the next step is told to rebuild (new frame, capacity change)
    unsigned step_count = 0;
    unsigned bind_gen = 0, opt_gen = 0;   // sgpr_bind_system calls / option changes that void kept lists (MdState::chain_*)
    DevBuf<int> d_flag, d_ncand, d_cand_j, d_cand_code, d_cidx;
    DevBuf<double> d_pos0, d_cell0;
    DevBuf<unsigned long long> d_hm;
    int hmw = 1;
    const int *step_flag = nullptr;  // the rebuild flag of the step being enqueued
    // the last kernel of a step as the first of the next (finalize_next_kernel): what it pre-binned
    bool pre_valid = false;
    const double *pre_pos = nullptr, *pre_cell = nullptr;
    unsigned pre_step = 0;
    bool fuse_next = true;           // option "fuse_next" (SGPR_FUSE_NEXT=0 at creation): off = every step bins for itself
    bool bin_identity = false;       // the positions handed to the binning kernel are in sorted order already (MD state)
    bool reduce_done = false;        // the step just enqueued has combined the ranks' partial sums itself (shard_next_kernel)
    bool force_scatter = false;      // option "reverse_scatter": the scatter form of the reverse pass on a single rank too
    MdState md;
    bool gather_ok = true;   // false: bin capacity / list length beyond the reverse-index format -> scatter form
    DevBuf<double> d_prec, d_G;
    DevBuf<double> d_gpart;
    DevBuf<double> d_rows_ones, d_rows_out, d_rows_ke;  // sgpr_kernel_rows / _columns scratch
    DevBuf<double> d_rows_bG, d_rows_bF, d_rows_bV;     // (batch of columns: G, [Fnbr | Fself], virial partials)
    DevBuf<double> d_rows_kepart;                       // (K_e: per-chunk column sums)
    DevBuf<double> sc_dense_part;                       // partial sums of a dense Q^T application (qr_keep_apply_all)
    DevBuf<double> sc_mae_v;                            // sgpr_data_force_mae: the batch of weight vectors
    DevBuf<int64_t> sc_mae_rows;                        //                      energy row and end of the force rows per frame
    DevBuf<double> sc_ea_y;                             // energy_rows_append: the targets on the device
    DevBuf<int64_t> sc_ea_rows;                         //                     rows of the frames' energies
    DevBuf<int> d_rows_cols, d_rows_rowof, d_rows_qoff;
    DevBuf<double> d_rows_vpart;
    bool rows16 = true;                                 // SGPR_ROWS16=0: one column per wave for every rows call
    int rows16_mb = 16384;                              // hand-over buffer of the sixteen-column form (SGPR_ROWS16_MB)
    int rows16_min = 1;                                 // fewest columns of a call that take it (SGPR_ROWS16_MIN; 1: a column has the same bits whatever call computed it)
    const char *info_rows = "none yet";                 // form of the last rows call (sgpr_solve_info)
    std::vector<int> rows_cols;
    // resident training set (data.inc): the design matrix [K_e; K_f; K_v] of the stored frames, column-major in the
    // caller's column order, design[c * design_rcap + r]; rows in frame-major blocks (e, 3N f, nv v)
    struct DataFrame {
        int N = 0, nv = 0;
        int64_t row0 = 0, rows = 0;
        std::vector<int32_t> numbers;
        std::vector<double> pos;
        double cell[9];
        int32_t pbc[3];
    };
    std::vector<DataFrame> frames;
    DevBuf<double> d_design, d_qr_A, d_qr_work;
    // scratch of the update-path entry points, kept between calls (hipMalloc / hipFree per call cost more than the
    // kernels they served: a refit makes a dozen such calls)
    DevBuf<double> sc_s2A, sc_s2x, sc_s2work, sc_mv_v, sc_mv_o, sc_ra_y, sc_ra_t, sc_vs_t, sc_ai_er, sc_ai_p, sc_ai_norm,
        sc_ai_krow, sc_ai_kself, sc_y, sc_bA, sc_bx, sc_bwork;
    DevBuf<int> sc_s2so, sc_ai_eslot, sc_ai_oslot, sc_ai_onn, sc_ai_info, sc_sel_idx, sc_sel_map, sc_chol_info;
    DevBuf<double> sc_sel_A, sc_sel_work;
    DevBuf<double> d_design_alt;  // second buffer of the resident matrix: column selections gather into it, then swap
    DevBuf<int64_t> sc_ai_ptr, sc_erow;
    DevBuf<unsigned char> sc_ise;
    int64_t design_rcap = 0, design_ccap = 0, design_rows = 0;
    bool design_hold = false;  // an edit entry point is re-indexing the columns itself
    // identity of the matrix: one id per stored frame and per inducing column (a pop returns to the earlier lists),
    // and the first-stage factors of the last few (matrix, targets) pairs: a rejected trial (add, refit, pop, refit:
    // gppotential.py:898-982) finds its second refit here
    std::vector<int64_t> frame_ids, col_ids;
    int64_t next_id = 1;
    struct R1Entry { uint64_t key[2] = {0, 0}; int m = 0; DevBuf<double> r1; uint64_t age = 0; };
    R1Entry r1_cache[4];
    uint64_t r1_clock = 0;
    // the first-stage factorisation itself, kept with its reflectors (data.inc): a new inducing LCE is one more
    // column pushed through them, a popped one is dropped.  Two of them: a data trial (push a frame, refit, pop it)
    // comes back to the one it left; the force-only fit of the noise search keeps its own.
    struct QrKeep {
        bool valid = false;
        int with_energies = 1;             // 0: the force-only fit of the noise search (energy rows zeroed)
        DevBuf<int64_t> erows;             // the energy rows of the stored frames
        int n_erows = 0;
        int rows = 0, R = 0, ldr = 0;      // design rows; rows / leading dimension of the factored work array
        int ncols = 0, ccap = 0;           // current columns, capacity of Rc
        std::vector<int64_t> frame_ids, col_ids;
        uint64_t ykey[2] = {0, 0};
        DevBuf<double> store, Rc, yt, yraw, ysnap, vec;
        size_t store_used = 0;
        std::vector<TsqrPanel> panels;     // the panels of the full factorisation
        size_t store_base = 0;             // where the appended columns' reflectors start in `store`: slot a holds
                                           // v [ldr], sc, alpha
        // what came after the full factorisation, in order (each one an orthogonal map of the rows):
        //   kind 0  ONE flat reflector over rows [k0, R): the column appended at position k0
        //   kind 1  a column SELECTION (sgpr_select_inducing: downsize(lii) / popfirst / remove at an index,
        //           gppotential.py:815-842, :1037-1046): the QR of R1[:, idx] over the leading `rows` rows, kept as the
        //           explicit matrix E = Q2^T (`rows` x `rows`, E[c * ld + r]): the identity rode along behind the targets.
        //           Applying it is two launches; its ~64 panel levels, one launch each, made every later refit 0.8 ms
        //           longer per selection in the chain (config 5 selects once per model update)
        struct Op {
            int kind = 0, k0 = 0, flat_ix = 0, rows = 0, ld = 0;
            char snap = 0;                 // flat: ysnap[flat_ix] holds Q^T Y from before it
            DevBuf<double> pstore;         // selection: E
        };
        std::vector<Op> ops;
        bool selected = false;             // a selection since the last refit (diagnostics)
        int nflat = 0;                     // flats among the ops (slots of `store` / `ysnap` in use: a stack)
        void clear_ops() { for (auto &o : ops) o.pstore.release(); ops.clear(); nflat = 0; }
        uint64_t age = 0;
    };
    QrKeep qr_keep[4];
    // the banded second stage of the last resident-matrix solve, kept with its reflectors (solve.inc::solve_stage2): an
    // inducing trial's refit is that factorisation asked about one more column
    struct S2Keep {
        bool valid = false;
        int m = 0, with_energies = -1;
        double noise0 = 0.0, sigma = 0.0, ridge = 0.0;
        std::vector<int64_t> col_ids, frame_ids;
        DevBuf<double> VT, R1;   // the panels' reflectors; the first-stage factor it was made for (m x m, column-major)
    };
    S2Keep s2k;
    const char *info_stage2 = "none";
    DevBuf<double> sc_s2W;
    DevBuf<int> sc_s2flag;
    DevBuf<double> sc_potrf;   // [2][64][64]: where the Cholesky panels park their diagonal blocks (linalg.hip)
    int qr_keep_mode = 1;  // option "qr_keep" (environment SGPR_QR_KEEP at creation): 1 on, 0 off, 2 verify
    // sgpr_solve state kept for sgpr_resolve: L of K_mm (+ridge) and the R factor of the last [K | Y]
    DevBuf<double> d_L, d_R1;
    DevBuf<double> d_edit_tmp;  // scratch of the incremental inducing-set edits
    bool chol_valid = false, r1_valid = false;
    // K_mm is block diagonal by species, so is its factor: blocks are factored one by one and an edit only
    // invalidates the blocks it touches.  blk_ok[s]: block s of d_L / d_choli is the factor of block s of d_M at
    // chol_ridge (only meaningful while chol_shape: both arrays have the current m x m_pad shape)
    std::vector<char> blk_ok;
    bool chol_shape = false;
    std::string info_stage1 = "none";        // sgpr_solve_info
    int info_blocks_done = 0, info_blocks = 0;
    double chol_ridge = 0.0, chol_dmean = 0.0;
    // per-step work arrays (local rows)
    DevBuf<double> d_Pn, d_norm, d_C, d_K, d_Aw, d_W, d_F, d_virpart, d_Epart, d_csq, d_packed;
    int csq_slots = 1;
    double *pin_dev = nullptr; // the same buffer as the device addresses it (zero-copy results)
    double *pin = nullptr;     // page-locked staging of sgpr_compute: [3N + 9] in | [4N + 11] out
    size_t pin_doubles = 0;
    int pin_flip = 0;          // which of the two output halves of `pin` the last sgpr_compute wrote
    double *pin_old = nullptr; // the buffer `pin` replaced when a larger frame came: views of the call before stay readable until the next growth
    DevBuf<int> d_shear;
    int epart_len = 0, virpart_len = 0;
    DevBuf<long long> d_stamps;  // SGPR_STAMPS=1 diagnostic
    DevBuf<long long> d_stamps2; // the same for the grouped W + covloss launch
    DevBuf<long long> d_pstamps; // SGPR_STAMPS=1 + a -DSGPR_PHASE_STAMPS build: [2][N][8] phase stamps (forward | reverse)
    DevBuf<int4> t_knm, t_w, t_cov, t_kmm, t_wcov;  // working-tile tables of the GEMMs
    DevBuf<int4> t_covl;                            // covloss tiles alone, longest reductions first (ride in the reverse kernel)
    bool cov_in_rev = false; // option "cov_in_rev": covloss tiles in the reverse kernel's launch instead of grouped with W
                             // (measured at 4096 / 512: W alone 22.7 -> 18.5 us, reverse + covloss 17.0 -> 24.6: the reverse pass
                             // already fills every SIMD's four wave slots, the tiles only push a third of its workgroups
                             // into a second round)
    int gemm_bm_k = 64, gemm_bm_w = 64;  // rows per tile of t_knm  /  t_w, t_cov, t_wcov
    int gemm_kd_k = 16, gemm_kd_w = 16;  // stage depth of the 32-row form (SGPR_GEMM_KD="k,w" overrides)
    int gemm_waves_k = 8;                // waves per K_nm tile (SGPR_GEMM_WAVES=4: the four-wave form)
    bool gemm_k64 = false, gemm_w64 = false;  // 64 x 64 tiles on eight waves for K_nm / for W + covloss: by size
    int gemm_wgs64 = 3;
    bool gemm_half = true;                    // 16-row half tiles for launches of fewer 32-row tiles than CUs (build_tiles)                       // workgroups per CU of those tiles: 2 (six register stage sets) or 3 (three sets); SGPR_GEMM_WGS64
    bool gemm_64_forced = false;              //   (decide_tile_heights) unless SGPR_GEMM_64="k,w" says so
    int cus_per_xcd = 32;                // CUs behind one XCD's dispatcher (multiProcessorCount / 8)
    bool tile_balance = true;            // SGPR_TILE_BALANCE=0: plain longest-first tile tables
    int tile_chain = 1;                  // SGPR_TILE_CHAIN=0: every W + covloss tile its own workgroup, longest first (build_tiles)
    int wcov_grid = 0;                   // workgroups of the grouped W + covloss launch (<= t_wcov.n: the rest are chained)
    bool xcd_quads = true;               // SGPR_XCD_QUADS=0: workgroup b of the descriptor kernels works on atoms 4b .. 4b+3
    std::vector<int4> h_t_w, h_t_cov, h_t_knm, h_t_both;
    // the three products of a step as ONE launch (gemm.hip::launch_gemm_fused): K_nm tiles first, then their consumers
    DevBuf<int4> t_fused;
    DevBuf<int> d_panel_cnt;
    int fuse_epoch = 0;
    // OFF by default: bit-identical, but 42.9 us against 18.3 + 22.5 for the two launches at 4096 / 512 (62 before every
    // panel counter had a cache line of its own).  DESIGN.md §3: both phases are ONE wave of tiles that end together, so
    // there is nothing to overlap; the fused K_nm tiles run on four waves, not eight; and the hand-off without an acquire
    // fence is the guide's measured-not-guaranteed form, here at four workgroups per CU
    bool gemm_fused = false;   // option "gemm_fused" (SGPR_GEMM_FUSED=1 at creation)
    // graph
    hipGraphExec_t gexec = nullptr;
    const void *g_pos = nullptr, *g_cell = nullptr, *g_out = nullptr;
    hipStream_t g_stream = nullptr;
    bool use_graph = false;  // eager launches pipeline fine while a step is >100 us of kernels; graph replay
                             // measured 8 us/step slower (177 vs 169 us) — opt in with sgpr_set_option("graph",1)
    double lone_w = 1.0;          // option "lone_atom_weight": k of two lone atoms of one species (GemmParams::lone_m1)
    bool ignore_unknown = false;  // option "ignore_unknown_species": atoms and LCE neighbours whose species is not in
                                  // the table are invisible (the reference's fixed-species kernels drop them
                                  // silently, descriptor/sesoap.py:343-346); default: SGPR_E_SPECIES
    bool use_fork = false;   // measured neutral (169.0 vs 167.6 us): kept as an option only
    // profiling
    bool profile = false;
    std::vector<hipEvent_t> ev;
    std::vector<std::string> stage_names;
    std::vector<double> stage_ms;
};

// ---------------------------------------------------------------------------- small kernels
__global__ void transpose_kernel(int rows, int cols, const double *A, int lda, double *B, int ldb)
{
    __shared__ double t[32][33];
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        const int r = r0 + k, c = c0 + threadIdx.x;
        t[k][threadIdx.x] = (r < rows && c < cols) ? A[(size_t)r * lda + c] : 0.0;
    }
    __syncthreads();
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        const int c = c0 + k, r = r0 + threadIdx.x;
        if (c < cols && r < rows) B[(size_t)c * ldb + r] = t[threadIdx.x][k];
    }
}

// packed = [F(3N) | beta(N) | E | virial(9)] in CALLER atom order.
// The scalar reductions (energy partials of the K_nm tiles, virial partials of the reverse kernel,
// largest neighbour count) are done by eleven extra workgroups, one per scalar, in a fixed order:
// reproducible sums, no hand-shake with the packer workgroups.  (An earlier two-level form — every
// block reduces a slice, release fence, ticket, last arriver combines — spent most of its 8 us in that
// dependent chain.)
// The last kernel of step s can also be the first of step s + 1 (finalize_next_kernel): when the next positions are
// already on the device — the next frame of a resident batch (mode 1), or the outcome of the integrator, which is run
// HERE, one wave per atom, right behind the force sum (mode 2: BAOAB Langevin / velocity Verlet as
// workloads.langevin_nvt; the reference drives ase.md.langevin through cl/md.py:117-128) — the wave that finished
// atom i bins it for step s + 1 and takes the rebuild decision (neighbor.hip::nl_bin_kernel, same rules at constant
// cell).  One launch and one cold-start chain less per step; an MD loop never leaves the device.
//   Quantities that need ALL atoms of step s (largest covloss, kinetic energy) are reduced by the reducer workgroups of
// step s + 1's launch (the per-atom values of step s are in memory by then): the covloss gate of the reference's
// calculate() (calculator/active.py:492-499) therefore fires one launch late — step s + 1 has been computed
// speculatively by then and is discarded; positions, velocities and results of the last three steps are kept in
// rings, so the state handed back is exactly step s.  After a halt every later launch of the queue exits at once.
#define SGPR_MD_SCAL 16   // doubles per step in the host-visible scalar ring: E, virial[9], overflow, max covloss, 2 x kinetic energy
struct FinNext {
    int mode;                  // 0 none, 1 frames, 2 md, 3 md tail (only the lagged reductions of the last step)
    int S, cap, force, step;   // species slots, slots per bin, rebuild regardless, this step's counter
    int pbc[3];
    double thr2;               // (skin / 2)^2: an atom farther than that from where the candidates were built -> rebuild
    const double *pos_in;      // mode 1: positions of the next frame, caller order
    const int *iperm, *cslot;  // mode 1: caller atom -> sorted index, species slot by caller index
    double *pos;               // [N][3] sorted working positions (this step's, overwritten with the next step's)
    int *flags;                // [4] rebuild flags by step counter & 3
    int *bc_next, *bc_cur;     // bin populations: the next step's (filled here) and this step's (cleared here)
    int *bin_of, *kslot;
    BinRec *b_rec;
    BinAux *b_aux;
    double *csq_rw;            // covloss partials of this step: cleared behind the read
    // md
    const double *x_cur, *v_cur;   // [N][3] sorted: positions of this step, velocities BEFORE its closing half kick
    double *x_next, *v_next;
    const double *mass, *sig;      // [N] sorted: mass, c2 sqrt(kT / m)
    const double *noise;           // [N][3] SORTED order (sorted on upload): the normal deviates of the next step's O (null: none
                                   //   — or, with a seed, drawn here: md_deviate)
    unsigned long long seed;       // != 0 and noise == null: counter-based deviates (Philox4x32-10 + Box-Muller) of
    long long t_index;             //   (seed; configuration index t_index, caller atom, component)
    double hdt, c1;                // dt / 2, exp(-friction dt)
    int nh, nh_first;              // Nose-Hoover (Melchionna) step instead of BAOAB; the first evaluation of the trajectory
    const double *x_prev;          // nh: positions of the configuration before this one
    double *v_now;                 // nh: the centred velocity (x_next - x_prev) / 2 dt of THIS configuration, known once it is evaluated
    const double *nh_zeta;         // nh: zeta of this configuration (device: written by md_nh_kernel behind the previous evaluation)
    int pending;                   // the closing half kick of this step is due (0 only for the very first evaluation)
    double *ke_cur;                // [N][2] m v^2 of this step: after the closing half kick | before it
    const double *ke_prev, *packed_prev;   // the same / the packed results of step s - 1 (null: no such step in this run)
    double ediff;                  // halt when the largest covloss of a step reaches it
    int *halt;                     // device: the first step that halted the run (INT_MAX: running; atomicMin)
    int *halt_host;                // mapped host memory, polled between chunks of launches: [0] the step whose covloss
                                   //   reached ediff, [1] the step that overflowed a capacity (INT_MAX: none)
    double *scal_cur, *scal_prev;  // rows of the scalar ring (device memory)
    int *mark_cur;                 // mapped host memory: set to 1 by this evaluation's last kernel (ONE posted write per step)
};

struct FinArgs {
    int N, cnt, first, stride, maxnn, t_stride, has_beta, nE, nV, bin_cap, t_check, csq_slots;
    int nbins_clear;            // bin counters to clear (4096: all)
    const int *perm, *slot, *nn, *nbr_j, *aux, *nn_raw;
    const unsigned short *T;
    const double *G;            // gather form: [N][maxnn][4]
    const double *Fnbr, *Fself; // scatter form
    const double *csq, *vs_sqrt, *Epart, *virpart;
    double mean_energy;
    double *packed;
    int *stat, *bin_count;
    const int *row_cols, *row_colslot, *nbr_code;   // training rows: column of each batch entry, species slot of every
                                                    // column, the lists' code words (neighbour species in bits 24..31)
    size_t g_stride, f_stride, v_stride, p_stride;  // batch (blockIdx.y, training rows): doubles between entries of
                                                    // G, [Fnbr | Fself], virpart, packed
    int xp;                     // scatter form: `packed` has the EXCHANGE layout (peer.inc): [fixed-point sums 3N | own part 3C | beta C | scalars 11]
    int xcmax;                  //   C = ceil(N / world): atoms of the largest share
    size_t scal_off;            // where the eleven scalars start in `packed` (4N; exchange layout: 7N)
    const int *flag;            // this step's rebuild flag: set -> the candidates were rebuilt from `pos`
    int *rebuilds;              // running count of rebuilds
    const double *pos;          // [N][3] sorted order
    double *pos0;               // [N][3] positions the candidate lists were built at
    const double *cell;         // this step's cell and bin grid (its inverse): kept as cell0[0..8], cell0[9..17] on
    const NlGrid *grid;         //   rebuild steps — the reference frame of the affine rebuild rule (neighbor.hip)
    double *cell0;
    FinNext nx;                 // finalize_next_kernel: what this launch does for the NEXT step
};

// wave64 sum on the DPP network (quads, half rows, rows) and four scalar row sums: the result is wave-uniform and
// costs no LDS round trip (six ds_bpermute rounds per sum were a third of this kernel's dependent chain)
template <int CTRL>
__device__ __forceinline__ double fin_dpp(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double fin_lane(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double fin_wave_sum(double v)
{
    v += fin_dpp<0xB1>(v);
    v += fin_dpp<0x4E>(v);
    v += fin_dpp<0x141>(v);
    v += fin_dpp<0x140>(v);
    return (fin_lane(v, 0) + fin_lane(v, 16)) + (fin_lane(v, 32) + fin_lane(v, 48));
}

// reducer workgroup q: E (0), the nine virial components (1..9), the largest neighbour count (10)
__device__ __forceinline__ void finalize_reduce(const FinArgs &f, int q)
{
    __shared__ double wsum[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const size_t by = blockIdx.y;
    double s = 0.0;
    if (q == 10) {
        // eight loads in flight per thread: one workgroup with one load per thread and trip moves 2 KB per memory
        // round trip (32768 atoms: 60 us of reducers behind a 15 us gather)
        // (four 16-B loads, not eight: the reducers share the kernel's register allocation with the gather waves, and
        // 64 VGPRs keep eight of those per SIMD)
        int m8[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        const int4 *src4 = (const int4 *)f.nn_raw;  // the array is padded to whole rows of 64: 16-B loads stay inside it
        const int n4 = (f.cnt + 3) / 4;
        for (int k0 = tid; k0 < n4; k0 += 1024) {
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int k = k0 + 256 * u;
                const int4 v = src4[min(k, n4 - 1)];
                const int e = 4 * k;
                int m = e < f.cnt ? v.x : 0;
                m = max(m, e + 1 < f.cnt ? v.y : 0);
                m = max(m, e + 2 < f.cnt ? v.z : 0);
                m = max(m, e + 3 < f.cnt ? v.w : 0);
                m8[u] = max(m8[u], m);
            }
        }
        const int mx = max(max(max(m8[0], m8[1]), max(m8[2], m8[3])), max(max(m8[4], m8[5]), max(m8[6], m8[7])));
        s = (double)mx;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
    } else {
        const double *src = q == 0 ? f.Epart : f.virpart + by * f.v_stride + (size_t)(q - 1) * f.nV;
        const int n = q == 0 ? f.nE : f.nV;
        double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};  // fixed assignment of terms to partial sums: same bits every run
        for (int k0 = tid; k0 < n; k0 += 2048) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + 256 * u;
                const double v = src[min(k, n - 1)];
                a8[u] += k < n ? v : 0.0;
            }
        }
        s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        s = fin_wave_sum(s);
    }
    if (lane == 0) wsum[wave] = s;
    __syncthreads();
    if (tid == 0) {
        if (q == 10 && by == 0) {
            if (*f.flag) {
                atomicAdd(f.rebuilds, 1);
                if (f.cell0 && f.cell) {
                    // (all eighteen values in registers before the first store: store-by-store the compiler must assume
                    // aliasing and serialises eighteen memory round trips on this one lane)
                    double cv[18];
#pragma unroll
                    for (int k = 0; k < 9; k++) { cv[k] = f.cell[k]; cv[9 + k] = f.grid->inv[k]; }
#pragma unroll
                    for (int k = 0; k < 18; k++) f.cell0[k] = cv[k];
                }
            }
            const int mx = (int)fmax(fmax(wsum[0], wsum[1]), fmax(wsum[2], wsum[3]));
            f.stat[0] = max(f.stat[0], mx);  // sticky
            // packed[4N+10]: 1 when this rank's step overflowed a capacity (its results are invalid); summed
            // over ranks by the all-reduce, so every rank learns that the step must be repeated
            const bool ov = mx > f.maxnn || f.stat[1] > f.bin_cap || (f.t_check && f.stat[2] > f.t_stride) || f.stat[3] != 0;
            f.packed[f.scal_off + 10] = ov ? 1.0 : 0.0;
            if (f.nx.mode == 2) {
                f.nx.scal_cur[10] = ov ? 1.0 : 0.0;
                *f.nx.mark_cur = 1;
                if (ov) {  // the lists of this step were clamped: its results and the state integrated from them are void
                    atomicMin(&f.nx.halt[0], f.nx.step);
                    f.nx.halt_host[1] = f.nx.step;
                }
            }
        } else if (q != 10) {
            const double v = ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3])) + (q == 0 ? f.mean_energy : 0.0);
            f.packed[by * f.p_stride + f.scal_off + q] = v;
            if (f.nx.mode == 2) f.nx.scal_cur[q] = v;
        }
    }
}

// lagged reductions of an MD run (finalize_next_kernel): q = 0 the largest covloss of the PREVIOUS step (and the halt
// decision of the covloss gate, calculator/active.py:492-499), q = 1 its kinetic energy sum m v^2 (fixed order)
__device__ __forceinline__ void finalize_reduce_prev(const FinArgs &f, int q)
{
    __shared__ double wprev[4], wprev2[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int N = f.N;
    double s = 0.0, s2 = 0.0;
    if (q == 0) {
        const double *b = f.nx.packed_prev + 3 * (size_t)N;
        double m8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int k0 = tid; k0 < N; k0 += 2048) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + 256 * u;
                const double v = b[min(k, N - 1)];
                m8[u] = fmax(m8[u], k < N ? v : 0.0);
            }
        }
        s = fmax(fmax(fmax(m8[0], m8[1]), fmax(m8[2], m8[3])), fmax(fmax(m8[4], m8[5]), fmax(m8[6], m8[7])));
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s = fmax(s, __shfl_xor(s, o, 64));
    } else {
        const double2 *src = (const double2 *)f.nx.ke_prev;
        double a8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, b8[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
        for (int k0 = tid; k0 < N; k0 += 2048) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int k = k0 + 256 * u;
                const double2 v = src[min(k, N - 1)];
                a8[u] += k < N ? v.x : 0.0;
                b8[u] += k < N ? v.y : 0.0;
            }
        }
        s = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        s = fin_wave_sum(s);
        s2 = ((b8[0] + b8[1]) + (b8[2] + b8[3])) + ((b8[4] + b8[5]) + (b8[6] + b8[7]));
        s2 = fin_wave_sum(s2);
    }
    if (lane == 0) { wprev[wave] = s; wprev2[wave] = s2; }
    __syncthreads();
    if (tid == 0) {
        if (q == 0) {
            const double bmax = fmax(fmax(wprev[0], wprev[1]), fmax(wprev[2], wprev[3]));
            f.nx.scal_prev[11] = bmax;
            if (bmax >= f.nx.ediff) {
                atomicMin(&f.nx.halt[0], f.nx.step - 1);
                f.nx.halt_host[0] = f.nx.step - 1;
            }
        } else {
            f.nx.scal_prev[12] = (wprev[0] + wprev[1]) + (wprev[2] + wprev[3]);
            f.nx.scal_prev[13] = (wprev2[0] + wprev2[1]) + (wprev2[2] + wprev2[3]);
        }
    }
}

// A fused last kernel decides the NEXT step's rebuild by |x - pos0| <= skin / 2, the constant-cell rule.  If the cell of this
// step is not the one the candidates were built in (a caller strained it between two steps without a rebuild), that rule
// under-counts (the binning kernel's affine rule, neighbor.hip, is what applies): the next step then simply rebuilds.  One
// lane, 18 loads; on a step that rebuilt, cell0 is being set to this very cell by the reducer.
__device__ __forceinline__ void fin_cell_guard(const FinArgs &f, int rebuilt, int s1)
{
    if (rebuilt || !f.cell0 || !f.cell) return;
    double a[9], b[9];
#pragma unroll
    for (int k = 0; k < 9; k++) { a[k] = f.cell[k]; b[k] = f.cell0[k]; }
    bool differ = false;
#pragma unroll
    for (int k = 0; k < 9; k++) differ |= a[k] != b[k];
    if (differ) atomicMax(&f.nx.flags[s1 & 3], 1);
}

// scatter form (sharded frames): F = atomic part + own part, one thread per atom
__global__ __launch_bounds__(256) void finalize_kernel(FinArgs f)
{
    const int tid = threadIdx.x, b = blockIdx.x, nA = gridDim.x - 11;
    if (b >= nA) { finalize_reduce(f, b - nA); return; }
    const int i = b * blockDim.x + tid;
    const size_t by = blockIdx.y;
    for (int k = i; k < f.nbins_clear; k += nA * blockDim.x) f.bin_count[(size_t)k * SGPR_BIN_STRIDE] = 0;
    if (i < f.N) {
        const int c = f.perm[i];
        if (*f.flag) {
#pragma unroll
            for (int k = 0; k < 3; k++) f.pos0[3 * i + k] = f.pos[3 * i + k];
        }
        double *packed = f.packed + by * f.p_stride;
        const int il = (i - f.first) / f.stride;
        const bool mine = i >= f.first && (i - f.first) % f.stride == 0 && il < f.cnt;
        if (f.xp) {
            // exchange layout (peer.inc): the fixed-point sums travel as INTEGERS and are added as integers over the ranks —
            // the total force is then the same bits for every number of ranks —; the own part only for this rank's atoms
#pragma unroll
            for (int k = 0; k < 3; k++) ((long long *)packed)[3 * (size_t)c + k] = ((const long long *)f.Fnbr)[3 * (size_t)i + k];
            if (mine) {
#pragma unroll
                for (int k = 0; k < 3; k++) packed[3 * (size_t)f.N + 3 * (size_t)il + k] = f.Fself[3 * (size_t)i + k];
            }
        } else {
            // (the scattered part is a fixed-point integer sum: order-independent, sgpr_internal.h)
#pragma unroll
            for (int k = 0; k < 3; k++)
                packed[3 * c + k] = (double)((const long long *)f.Fnbr)[by * f.f_stride + 3 * i + k] * (1.0 / SGPR_FIX_SCALE) + f.Fself[by * f.f_stride + 3 * i + k];
        }
        double bt = 0.0;
        if (f.has_beta && mine) {
            double cs = 0.0;  // |choli k_i|^2: the tile partials in their fixed order
            for (int k = 0; k < f.csq_slots; k++) cs += f.csq[(size_t)il * f.csq_slots + k];
            const double v = 1.0 - cs;
            bt = sqrt(v > 0.0 ? v : 0.0) * f.vs_sqrt[f.slot[i]];
        }
        if (!f.xp) packed[3 * (size_t)f.N + c] = bt;
        else if (mine) packed[3 * (size_t)f.N + 3 * (size_t)f.xcmax + il] = bt;
    }
}

// scatter form with the next frame named (sharded steps over resident frames, FinNext mode 1): thread w packs sorted
// atom w as finalize_kernel does — and clears what the binning kernel would clear for the next step: its fixed-point
// force sums, its covloss partials — and bins CALLER atom w of the next frame (finalize_next_kernel<1>'s second job).
// Every rank bins all atoms, as every rank's stand-alone binning launch did: one launch less per rank and step.
__global__ __launch_bounds__(256) void finalize_scatter_next_kernel(FinArgs f)
{
    const int tid = threadIdx.x, b = blockIdx.x, nA = gridDim.x - 11;
    if (b >= nA) { finalize_reduce(f, b - nA); return; }
    const FinNext &x = f.nx;
    const int i = b * 256 + tid, s1 = x.step + 1;
    const bool act = i < f.N;
    const int ia = act ? i : 0;
    // requests (unconditional, see finalize_next_kernel)
    const int rebuilt = *f.flag;
    const NlGrid g = *f.grid;
    const int c = f.perm[ia], slot_i = f.slot[ia], ib = x.iperm[ia], slot_b = x.cslot[ia];
    long long fn[3];
    double fsv[3], xn[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        fn[k] = ((const long long *)f.Fnbr)[3 * (size_t)ia + k];
        fsv[k] = f.Fself[3 * (size_t)ia + k];
        xn[k] = x.pos_in[3 * (size_t)ia + k];
    }
    const int il = (ia - f.first) / f.stride;
    const bool mine = f.has_beta && act && ia >= f.first && (ia - f.first) % f.stride == 0 && il < f.cnt;
    double cs = 0.0;
    if (mine)
        for (int k = 0; k < f.csq_slots; k++) cs += f.csq[(size_t)il * f.csq_slots + k];
    double xc[3], p0[3];
#pragma unroll
    for (int k = 0; k < 3; k++) { xc[k] = x.pos[3 * (size_t)ib + k]; p0[k] = f.pos0[3 * (size_t)ib + k]; }
    const double vs = mine ? f.vs_sqrt[slot_i < x.S ? slot_i : 0] : 0.0;
    for (int k = i; k < f.nbins_clear; k += nA * 256) x.bc_cur[(size_t)k * SGPR_BIN_STRIDE] = 0;
    if (i == 0) {
        if (x.force) atomicMax(&x.flags[s1 & 3], 1);
        x.flags[(s1 + 2) & 3] = 0;
        fin_cell_guard(f, rebuilt, s1);
    }
    if (!act) return;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        f.packed[3 * (size_t)c + k] = (double)fn[k] * (1.0 / SGPR_FIX_SCALE) + fsv[k];
        ((long long *)f.Fnbr)[3 * (size_t)i + k] = 0;
    }
    if (mine)
        for (int k = 0; k < f.csq_slots; k++) x.csq_rw[(size_t)il * f.csq_slots + k] = 0.0;
    const double v = 1.0 - cs;
    f.packed[3 * (size_t)f.N + c] = mine ? sqrt(v > 0.0 ? v : 0.0) * vs : 0.0;
    // ---- the next frame's atom w
    int bidx[3], w[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double fr = xn[0] * g.inv[k] + xn[1] * g.inv[3 + k] + xn[2] * g.inv[6 + k];
        w[k] = 0;
        bidx[k] = 0;
        if (x.pbc[k] && (g.inv[k] != 0.0 || g.inv[3 + k] != 0.0 || g.inv[6 + k] != 0.0)) {
            const double fl = floor(fr);
            w[k] = (int)fl;
            fr -= fl;
            const int bb = (int)(fr * g.nb[k]);
            bidx[k] = bb >= g.nb[k] ? g.nb[k] - 1 : (bb < 0 ? 0 : bb);
        }
    }
    const int bin = (bidx[0] * g.nb[1] + bidx[1]) * g.nb[2] + bidx[2];
    int kb = -1;
    if (slot_b < x.S) kb = atomicAdd(&x.bc_next[(size_t)bin * SGPR_BIN_STRIDE], 1);
    double d2 = 0.0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double q0 = rebuilt ? xc[k] : p0[k];
        d2 += (xn[k] - q0) * (xn[k] - q0);
        x.pos[3 * (size_t)ib + k] = xn[k];
        if (rebuilt) f.pos0[3 * (size_t)ib + k] = xc[k];
    }
    if (!(d2 <= x.thr2)) atomicMax(&x.flags[s1 & 3], 1);
    x.bin_of[ib] = bin;
    x.kslot[ib] = kb;
    if (slot_b < x.S) {
        if (max(max(abs(w[0]), abs(w[1])), abs(w[2])) > 32767) atomicMax(&f.stat[3], 1);
        if (kb < x.cap) {
            const size_t e = (size_t)bin * x.cap + kb;
            BinRec r;
            r.x = xn[0]; r.y = xn[1]; r.z = xn[2]; r.idx = ib; r.pad = 0;
            x.b_rec[e] = r;
            BinAux ax;
            ax.w0 = (short)w[0]; ax.w1 = (short)w[1]; ax.w2 = (short)w[2]; ax.slot = (short)slot_b;
            x.b_aux[e] = ax;
        } else
            atomicMax(&f.stat[1], kb + 1);
    }
}

// gather form: one wave per atom i,  F_i = (sum_t g_it, kept by the reverse kernel) - sum_t' G[i][t']
// where G[i][t'] is the gradient of the pair (j_t' -> i), stored at i's own list position by the
// reverse kernel: one coalesced row per wave, fixed shuffle tree: reproducible.
__global__ __launch_bounds__(256) void finalize_gather_kernel(FinArgs f)
{
    const int tid = threadIdx.x, b = blockIdx.x, nA = gridDim.x - 11;
    if (b >= nA) { finalize_reduce(f, b - nA); return; }
    for (int k = b * 256 + tid; k < f.nbins_clear; k += nA * 256) f.bin_count[(size_t)k * SGPR_BIN_STRIDE] = 0;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = b * 4 + wave;
    if (i >= f.N) return;
    const size_t by = blockIdx.y;
    // the first 64 pair gradients of the row are requested before the neighbour count is known (the row has maxnn
    // slots; what lies beyond the count is masked after the load): one cold miss less on the chain
    // (frames of a few thousand atoms are bound by this chain; large ones by bytes, and the slots beyond the count are
    // a third of the row: there the count is read first)
    // (training rows: only the neighbours of the column's species carry a gradient — their slots are picked out after the
    // list's code words are known, not the whole row on speculation)
    const bool spec = f.N <= 16384 && !f.row_cols;
    double2 g0 = make_double2(0.0, 0.0), g1 = make_double2(0.0, 0.0);
    if (spec && lane < f.maxnn) {
        const double2 *row = (const double2 *)(f.G + by * f.g_stride + ((size_t)i * f.maxnn + lane) * 4);
        g0 = row[0]; g1 = row[1];
    }
    const int n = f.nn[i];
    if (!spec && lane < n && !f.row_cols) {
        const double2 *row = (const double2 *)(f.G + by * f.g_stride + ((size_t)i * f.maxnn + lane) * 4);
        g0 = row[0]; g1 = row[1];
    }
    double fs = lane < 3 ? f.Fself[by * f.f_stride + 3 * (size_t)i + lane] : 0.0;
    double *packed = f.packed + by * f.p_stride;
    const int c = f.perm[i];
    double cs = 1.0;
    if (f.has_beta) {  // |choli k_i|^2: the tile partials, summed by a fixed tree
        double x = 0.0;
        for (int k = lane; k < f.csq_slots; k += 64) x += f.csq[(size_t)i * f.csq_slots + k];
        cs = fin_wave_sum(x);
    }
    const double vs = f.has_beta ? f.vs_sqrt[f.slot[i]] : 0.0;
    double fx = 0.0, fy = 0.0, fz = 0.0;
    const int want = f.row_cols ? f.row_colslot[f.row_cols[by]] : -1;  // rows: only neighbours of the column's species wrote
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        if (t < n && (want < 0 || ((f.nbr_code[(size_t)i * f.maxnn + t] >> 24) & 0xff) == want)) {
            double2 b0 = g0, b1 = g1;
            if (t0 > 0 || (!spec && want >= 0)) {
                const double2 *row = (const double2 *)(f.G + by * f.g_stride + ((size_t)i * f.maxnn + t) * 4);
                b0 = row[0]; b1 = row[1];
            }
            fx += b0.x; fy += b0.y; fz += b1.x;
        }
    }
    fx = fin_wave_sum(fx); fy = fin_wave_sum(fy); fz = fin_wave_sum(fz);
    if (lane < 3) packed[3 * (size_t)c + lane] = fs - (lane == 0 ? fx : lane == 1 ? fy : fz);
    if (lane < 3 && *f.flag) f.pos0[3 * (size_t)i + lane] = f.pos[3 * (size_t)i + lane];
    if (lane == 3) {
        const double v = 1.0 - cs;
        packed[3 * (size_t)f.N + c] = f.has_beta ? sqrt(v > 0.0 ? v : 0.0) * vs : 0.0;
    }
}

// Standard normal deviate number (t, atom, component) of the stream `seed`: Philox4x32-10 (Salmon et al., SC'11) on the
// counter (t lo, t hi, atom, component), two 53-bit uniforms, Box-Muller.  Counter-based: a deviate depends on WHAT it
// is for, not on when it is drawn — a run reproduces whatever its batching, and an evaluation repeated after a halt
// draws the same numbers.  (The host loop draws from numpy instead: sgpr_md_run accepts its rows; sgpr_md_deviates
// returns these for a host twin.)
__device__ __forceinline__ double md_deviate(unsigned long long seed, long long t, int atom, int comp)
{
    unsigned c0 = (unsigned)t, c1 = (unsigned)((unsigned long long)t >> 32), c2 = (unsigned)atom, c3 = (unsigned)comp;
    unsigned k0 = (unsigned)seed, k1 = (unsigned)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; r++) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
        const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1, n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    const unsigned long long a = ((unsigned long long)c0 << 32) | c1, b = ((unsigned long long)c2 << 32) | c3;
    const double u1 = (double)((a >> 11) + 1ull) * (1.0 / 9007199254740992.0);   // (0, 1]
    const double u2 = (double)(b >> 11) * (1.0 / 9007199254740992.0);           // [0, 1)
    return sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2);
}

__global__ void md_deviates_kernel(int N, int rows, unsigned long long seed, long long t0, double *out)
{
    const size_t n3 = (size_t)3 * N, tot = n3 * rows;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / n3, w = e - r * n3;
        out[e] = md_deviate(seed, t0 + (long long)r, (int)(w / 3), (int)(w % 3));
    }
}

// One Nose-Hoover step of a coordinate (constant cell): the scheme of Melchionna, Ciccotti and Holian (1993) as ase.md.npt.NPT
// integrates it with pfactor = None — what the reference's default md(dynamics="NPT", bulk_modulus=None) runs, cl/md.py:17,
// :131-166 (ASE is a third-party dependency absent here: restated from its published algorithm) —
//     x_(n+1) = (2 x_n - x_(n-1) (1 - b) + dt^2 F_n / m) / (1 + b),   b = dt zeta_n / 2,
//     v_n = (x_(n+1) - x_(n-1)) / (2 dt)          (the momenta ASE sets at the end of its step: the kinetic energy of the log line)
// and, the first time, x_(-1) = x_0 - dt v_0 + dt^2 F_0 / (2 m) (its _calculate_q_past_and_future with zeta = 0).
// Operations and their order are those of workloads.nose_hoover_nvt: no contraction, true divisions.  Returns v_n.
__device__ __forceinline__ double md_nh_advance(const FinNext &x, double F, double ms, double xc, double v0, double xprev, double zeta, double &xn)
{
#pragma clang fp contract(off)
    const double dt = 2.0 * x.hdt;
    const double a = __ddiv_rn((dt * dt) * F, ms);
    double xp = xprev;
    if (x.nh_first) xp = (xc - dt * v0) + 0.5 * a;
    const double b = x.hdt * zeta;
    const double num = ((2.0 * xc) - xp * (1.0 - b)) + a;
    xn = __ddiv_rn(num, 1.0 + b);
    // (the first time v_0 is the caller's own: ASE's first thermostat step takes the kinetic energy of the initial momenta)
    return x.nh_first ? v0 : __ddiv_rn(xn - xp, 2.0 * dt);
}

// zeta_(n+1) = zeta_(n-1) + 2 dt tfact (KE_n - K0) behind evaluation n (ONE workgroup; its own launch: the integrating waves of
// evaluation n + 1 all need the sum over the atoms of evaluation n).  Fixed order: thread t sums the atoms t, t + 256, ...,
// then a pairwise tree in natural order — workloads.nose_hoover_nvt adds in the same order.
__global__ __launch_bounds__(256) void md_nh_kernel(int N, const double *ke, double *zeta, int n, double dt, double c1, double c2, double K0,
                                                    const int *halt, int step, double *scal_row)
{
    if (*halt < step) return;
    __shared__ double wsum[4];
    const int tid = threadIdx.x;
    double s = 0.0;
    for (int k = tid; k < N; k += 256) s += ke[2 * (size_t)k];
    s = fin_wave_sum(s);
    if ((tid & 63) == 0) wsum[tid >> 6] = s;
    __syncthreads();
    if (tid == 0) {
#pragma clang fp contract(off)
        const double KE = 0.5 * ((wsum[0] + wsum[1]) + (wsum[2] + wsum[3]));
        const int sc = n & 3, sn = (n + 1) & 3, sp = (n + 3) & 3;
        const double d = KE - K0;
        const double zprev = n == 0 ? -(c1 * d) : zeta[sp];   // (ASE's initialize(): zeta_past = -dt tfact (KE - K0), zeta = 0)
        const double zcur = n == 0 ? 0.0 : zeta[sc];
        const double zint = n == 0 ? 0.0 : zeta[4 + sc];
        const double znew = zprev + c2 * d;
        zeta[sn] = znew;
        zeta[4 + sn] = zint + dt * znew;                      // (ASE: zeta_integrated += dt * zeta, after the shift)
        scal_row[14] = zcur;
        scal_row[15] = zint;
    }
}

// The same gather, and with it the first kernel of the NEXT step (FinNext): a wave takes an atom to its next position —
// read from the next frame (MODE 1) or integrated (MODE 2) —, bins it there and takes part in the rebuild decision.
// Grid: ceil(N / 4) gather workgroups, 11 reducers of this step, 2 lagged reducers.
//   What bounds this kernel is its chain of dependent memory round trips, each of them a cold miss (2 - 3 us per link at
// 4096 atoms).  The gather alone is two links (row + counts -> sums -> stores), the binning kernel four (index -> position
// -> returning atomic -> record).  Fused naively they add up (12.4 us against 7.1 + 4.4 as two launches, measured).  So:
//   * MODE 1: the two jobs of a wave are INDEPENDENT — it sums the forces of sorted atom number w and bins CALLER atom
//     number w (position and species slot by caller index: no indirection in front of the atomic); the atomic is issued
//     before the sums, its record is stored after them;
//   * MODE 2: the position depends on the force, but on nothing else behind an indirection (the noise is sorted on upload);
//   * the halt word of an MD run is requested with everything else and looked at before the first store.
template <int MODE>
__global__ __launch_bounds__(256) void finalize_next_kernel(FinArgs f)
{
    const int tid = threadIdx.x, b = blockIdx.x, nA = gridDim.x - 13;
    const FinNext &x = f.nx;
    const int halt_w = MODE == 2 ? *x.halt : 0x7fffffff;
    if (b >= nA) {
        // a run that has halted (covloss gate / capacity overflow at an earlier step): nothing may be touched any more
        if (halt_w < x.step) return;
        if (b >= nA + 11) {
            if (MODE == 2 && x.packed_prev) finalize_reduce_prev(f, b - nA - 11);
        } else
            finalize_reduce(f, b - nA);
        return;
    }
    const int s1 = x.step + 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int i = b * 4 + wave;          // sorted atom whose forces this wave sums
    const bool act = i < f.N;
    const int ia = act ? i : 0;
    // ---- requests: ONE round trip.  Every load below is unconditional (clamped indices, the value masked afterwards): a
    // load inside `if (lane < 3)` or behind `act` becomes a branch with an s_waitcnt vmcnt(0) at its join, and the five
    // such joins of the first version of this kernel were five cold misses one behind the other (ISA; 12.4 us)
    const int l3 = lane < 3 ? lane : 2, lg = lane < f.maxnn ? lane : f.maxnn - 1, lc = lane < f.csq_slots ? lane : f.csq_slots - 1;
    const int rebuilt = *f.flag;
    const NlGrid g = *f.grid;
    const double2 *grow = (const double2 *)(f.G + ((size_t)ia * f.maxnn + lg) * 4);
    double2 g0 = grow[0], g1 = grow[1];
    const int n_ld = f.nn[ia];
    double fs = f.Fself[3 * (size_t)ia + l3];
    const int c = f.perm[ia];
    const int slot_i = f.slot[ia];
    // the atom this wave BINS: MODE 1: caller atom number i (sorted index ib = iperm[i]); MODE 2: sorted atom i itself
    const int ib = MODE == 1 ? x.iperm[ia] : ia;
    const int slot_b = MODE == 1 ? x.cslot[ia] : slot_i;
    double xc = 0.0, p0 = 0.0, vc = 0.0, ms = 1.0, sg = 0.0, nz = 0.0, xn = 0.0;
    if (MODE == 1) xn = x.pos_in[3 * (size_t)ia + l3];
    if (MODE == 2) {
        xc = x.x_cur[3 * (size_t)ia + l3];
        p0 = f.pos0[3 * (size_t)ia + l3];
        vc = x.v_cur[3 * (size_t)ia + l3];
        ms = x.mass[ia];
        sg = x.sig[ia];
        nz = x.noise ? x.noise[3 * (size_t)ia + l3] : 0.0;   // (wave-uniform condition)
    }
    double xpv = 0.0, zeta = 0.0;
    if (MODE == 2 && x.nh) { xpv = x.x_prev[3 * (size_t)ia + l3]; zeta = *x.nh_zeta; }   // (wave-uniform condition)
    double csv = f.has_beta ? f.csq[(size_t)ia * f.csq_slots + lc] : 0.0;   // (wave-uniform condition)
    if (lane >= f.csq_slots) csv = 0.0;
    if (f.has_beta)
        for (int k = lane + 64; k < f.csq_slots; k += 64) csv += f.csq[(size_t)ia * f.csq_slots + k];  // (more than 64 slots: rare)
    if (lane >= f.maxnn) { g0 = make_double2(0.0, 0.0); g1 = g0; }
    if (lane >= 3) fs = 0.0;
    const int n = act ? n_ld : 0;
    // second link (beside the atomic below): what hangs on an index that was itself loaded
    const double vs = f.has_beta ? f.vs_sqrt[slot_i < x.S ? slot_i : 0] : 0.0;
    if (MODE == 1) {
        xc = x.pos[3 * (size_t)ib + l3];
        p0 = f.pos0[3 * (size_t)ib + l3];
    }
    if (halt_w < x.step) return;  // (before the first store)
    for (int k = b * 256 + tid; k < f.nbins_clear; k += nA * 256) x.bc_cur[(size_t)k * SGPR_BIN_STRIDE] = 0;
    if (b == 0 && tid == 0) {
        // binning of step s + 1: its flag is flag[(s + 1) & 3] (cleared two binnings ago, read by nobody now); the flag of
        // step s + 3 is cleared for the binning after the next (neighbor.hip does the same with its cycle)
        if (x.force) atomicMax(&x.flags[s1 & 3], 1);
        x.flags[(s1 + 2) & 3] = 0;
        fin_cell_guard(f, rebuilt, s1);
    }
    if (!act) return;
    if (f.has_beta)
        for (int k = lane; k < f.csq_slots; k += 64) x.csq_rw[(size_t)i * f.csq_slots + k] = 0.0;  // (the binning kernel clears these)
    // bin of a position (lane 0 of the wave), and the atomic that hands out its slot
    int bin = 0, kb = -1, w0 = 0, w1 = 0, w2 = 0;
    auto place = [&](double X, double Y, double Z) {
        int bidx[3], w[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double fr = X * g.inv[k] + Y * g.inv[3 + k] + Z * g.inv[6 + k];
            w[k] = 0;
            bidx[k] = 0;
            if (x.pbc[k] && (g.inv[k] != 0.0 || g.inv[3 + k] != 0.0 || g.inv[6 + k] != 0.0)) {
                const double fl = floor(fr);
                w[k] = (int)fl;
                fr -= fl;
                const int bb = (int)(fr * g.nb[k]);
                bidx[k] = bb >= g.nb[k] ? g.nb[k] - 1 : (bb < 0 ? 0 : bb);
            }
        }
        bin = (bidx[0] * g.nb[1] + bidx[1]) * g.nb[2] + bidx[2];
        w0 = w[0]; w1 = w[1]; w2 = w[2];
        if (slot_b < x.S) kb = atomicAdd(&x.bc_next[(size_t)bin * SGPR_BIN_STRIDE], 1);
    };
    double X = 0.0, Y = 0.0, Z = 0.0;
    if (MODE == 1) {
        X = fin_lane(xn, 0); Y = fin_lane(xn, 1); Z = fin_lane(xn, 2);
        if (lane == 0) place(X, Y, Z);   // (the atomic's round trip runs beside the second link's loads)
    }
    // ---- the forces of atom i
    const double cs = f.has_beta ? fin_wave_sum(csv) : 1.0;
    double fx = 0.0, fy = 0.0, fz = 0.0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        if (t < n) {
            double2 b0 = g0, b1 = g1;
            if (t0 > 0) {
                const double2 *row = (const double2 *)(f.G + ((size_t)i * f.maxnn + t) * 4);
                b0 = row[0]; b1 = row[1];
            }
            fx += b0.x; fy += b0.y; fz += b1.x;
        }
    }
    fx = fin_wave_sum(fx); fy = fin_wave_sum(fy); fz = fin_wave_sum(fz);
    const double Fv = fs - (lane == 0 ? fx : lane == 1 ? fy : fz);
    if (lane < 3) f.packed[3 * (size_t)c + lane] = Fv;
    if (lane == 3) {
        const double v = 1.0 - cs;
        f.packed[3 * (size_t)f.N + c] = f.has_beta ? sqrt(v > 0.0 ? v : 0.0) * vs : 0.0;
    }
    // ---- the next step
    double ke = 0.0, kp = 0.0;
    if (MODE == 2 && x.nh) {
        if (lane < 3) {
            const double vnow = md_nh_advance(x, Fv, ms, xc, vc, xpv, zeta, xn);
            ke = ms * (vnow * vnow);
            kp = ke;
            x.x_next[3 * (size_t)i + lane] = xn;
            x.v_now[3 * (size_t)i + lane] = vnow;
        }
    } else if (MODE == 2 && lane < 3) {
        // BAOAB, exactly the operations (and their order) of workloads.langevin_nvt: no contraction into fused
        // multiply-adds, a true division
#pragma clang fp contract(off)
        const double kick = __ddiv_rn(x.hdt * Fv, ms);
        double v = vc;
        if (x.pending) v = v + kick;       // closes step s: the velocity an observer sees at step s
        ke = ms * (v * v);
        kp = ms * (vc * vc);               // ... and the one the calculator is handed with the positions (its log line)
        const double v2 = v + kick;        // B
        const double x1 = xc + x.hdt * v2; // A
        if (!x.noise && x.seed != 0ull && sg != 0.0) nz = md_deviate(x.seed, x.t_index, c, lane);
        const double v3 = x.c1 * v2 + sg * nz;  // O
        xn = x1 + x.hdt * v3;              // A
        x.x_next[3 * (size_t)i + lane] = xn;
        x.v_next[3 * (size_t)i + lane] = v3;
    }
    if (MODE == 2) {
        const double k3 = fin_lane(ke, 0) + fin_lane(ke, 1) + fin_lane(ke, 2);
        const double p3 = fin_lane(kp, 0) + fin_lane(kp, 1) + fin_lane(kp, 2);
        if (lane == 0) *(double2 *)(x.ke_cur + 2 * (size_t)i) = make_double2(k3, p3);
        X = fin_lane(xn, 0); Y = fin_lane(xn, 1); Z = fin_lane(xn, 2);
        if (lane == 0) place(X, Y, Z);
    }
    if (rebuilt) p0 = xc;  // this step rebuilt the candidates: built at this step's positions
    if (lane < 3) {
        x.pos[3 * (size_t)ib + lane] = xn;
        if (rebuilt) f.pos0[3 * (size_t)ib + lane] = xc;
    }
    const double dd = xn - p0;
    const double d2 = fin_lane(dd, 0) * fin_lane(dd, 0) + fin_lane(dd, 1) * fin_lane(dd, 1) + fin_lane(dd, 2) * fin_lane(dd, 2);
    if (lane == 0) {
        if (!(d2 <= x.thr2)) atomicMax(&x.flags[s1 & 3], 1);
        x.bin_of[ib] = bin;
        x.kslot[ib] = kb;
        if (slot_b < x.S) {
            if (max(max(abs(w0), abs(w1)), abs(w2)) > 32767) atomicMax(&f.stat[3], 1);
            if (kb < x.cap) {
                const size_t e = (size_t)bin * x.cap + kb;
                BinRec r;
                r.x = X; r.y = Y; r.z = Z; r.idx = ib; r.pad = 0;
                x.b_rec[e] = r;
                BinAux ax;
                ax.w0 = (short)w0; ax.w1 = (short)w1; ax.w2 = (short)w2; ax.slot = (short)slot_b;
                x.b_aux[e] = ax;
            } else
                atomicMax(&f.stat[1], kb + 1);
        }
    }
}

// normal deviates of an MD run, caller order -> sorted order (so that the integrator reads them without an indirection)
__global__ void md_sort_rows_kernel(int N, int rows, const int *perm, const double *in, double *out)
{
    const size_t n3 = (size_t)3 * N, tot = n3 * rows;
    for (size_t e = (size_t)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += (size_t)gridDim.x * blockDim.x) {
        const size_t r = e / n3, w = e - r * n3;
        const int i = (int)(w / 3), k = (int)(w - 3 * (size_t)i);
        out[e] = in[r * n3 + 3 * (size_t)perm[i] + k];
    }
}

// the lagged reductions of the LAST step of an MD run (there is no next launch to carry them)
__global__ __launch_bounds__(256) void finalize_tail_kernel(FinArgs f)
{
    if (*f.nx.halt < f.nx.step) return;
    finalize_reduce_prev(f, (int)blockIdx.x);
}

// ---------------------------------------------------------------------------- host tables
void host_build_harm_coef(HarmCoef *hc)
{
    memset(hc, 0, sizeof(*hc));
    hc->y00 = sqrt(1.0 / (4.0 * M_PI));  // descriptor/ylm.py:56
    for (int l = 2; l <= SGPR_MAX_L; l++)
        for (int m = 0; m < l - 1; m++) {
            hc->al[l][m] = sqrt((4.0 * l * l - 1.0) / (l * l - m * m));                                  // :57-65
            hc->bl[l][m] = -sqrt(((l - 1.0) * (l - 1.0) - m * m) / (4.0 * (l - 1.0) * (l - 1.0) - 1.0)); // :66-76
        }
    for (int l = 0; l <= SGPR_MAX_L; l++) hc->cl[l] = sqrt(2.0 * l + 1.0);              // :77
    for (int l = 1; l <= SGPR_MAX_L; l++) hc->dl[l] = -sqrt(1.0 + 1.0 / (2.0 * l));      // :78-80
}

static double factorial(int n)
{
    double f = 1.0;
    for (int i = 2; i <= n; i++) f *= i;
    return f;
}

static void build_pack(sgpr_model *h)
{
    const int N1 = h->nmax + 1, L1 = h->lmax + 1, U = h->S * N1;
    h->D = N1 * N1 * L1;
    h->Dc = U * (U + 1) / 2 * L1;
    h->Dpad = rup(h->Dc, 32);
    h->CS = h->S * N1 * L1 * L1;
    h->h_pack.assign(h->Dc, PackEntry());
    for (int u = 0; u < U; u++)
        for (int v = u; v < U; v++)
            for (int l = 0; l < L1; l++) {
                const int pair = u * U - (u * (u - 1)) / 2 + (v - u);
                PackEntry &e = h->h_pack[pair * L1 + l];
                const int n1 = u % N1, n2 = v % N1;
                // descriptor/sesoap.py:116-128
                const double a1 = 1.0 / ((2 * l + 1) * pow(2.0, 2 * n1 + l) * factorial(n1) * factorial(n1 + l));
                const double a2 = 1.0 / ((2 * l + 1) * pow(2.0, 2 * n2 + l) * factorial(n2) * factorial(n2 + l));
                e.u = (int16_t)u; e.v = (int16_t)v; e.l = (int16_t)l; e.pad = 0;
                e.coef = sqrt(a1 * a2) * (u == v ? 1.0 : 1.4142135623730951);
            }
}

static int slot_of(const sgpr_model *h, int z)
{
    for (int k = 0; k < h->S; k++)
        if (h->species[k] == z) return k;
    return -1;
}

// ---------------------------------------------------------------------------- ABI
extern "C" const char *sgpr_last_error(void) { return g_err; }
extern "C" int sgpr_version(void) { return 1000; }

extern "C" int sgpr_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int sgpr_create(int lmax, int nmax, double eta, double rc, int S, const int32_t *species_z,
                           const double *radii, int device, sgpr_model **out)
{
    if (!out || !species_z || S < 1 || S > SGPR_MAX_S) return fail(SGPR_E_INVALID, "sgpr_create: bad species table (S=%d)", S);
    if (!(rc > 0.0) || !(eta > 0.0)) return fail(SGPR_E_INVALID, "sgpr_create: rc and eta must be positive");
    const bool in234 = lmax >= 2 && lmax <= 4 && nmax >= 2 && nmax <= 4;
    // every (lmax, nmax) of {2,3,4}^2 with up to eight species slots; nine to sixteen (SGPR_MAX_S) for the reference's default
    // lmax = nmax = 3 (descriptor.hip::DISPATCH_LNS; the reverse kernel then runs one atom per workgroup: SGPR_REV_WPW)
    const bool ok = in234 && (S <= 8 || (lmax == 3 && nmax == 3));
    if (!ok) return fail(SGPR_E_UNSUPPORTED, "sgpr_create: (lmax,nmax,S)=(%d,%d,%d) is not compiled in", lmax, nmax, S);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(SGPR_E_NODEVICE, "no HIP device: libsgpr_hip has no CPU fallback");
    if (device < 0 || device >= ndev) return fail(SGPR_E_INVALID, "sgpr_create: device %d out of range", device);
    HIPCHK(hipSetDevice(device));
    sgpr_model *h = new sgpr_model();
    h->lmax = lmax; h->nmax = nmax; h->eta = eta; h->rc = rc; h->S = S; h->device = device;
    h->species.assign(species_z, species_z + S);
    h->radii.assign(S, 1.0);
    for (int k = 0; k < S; k++) h->radii[k] = radii ? radii[k] : (species_z[k] == 1 ? 0.5 : 1.0);  // sesoap.py:84-99
    h->mean_w.assign(S, 0.0);
    h->vscale.assign(S, 1.0);
    if (hipStreamCreate(&h->stream) != hipSuccess) { delete h; return fail(SGPR_E_NODEVICE, "hipStreamCreate failed"); }
    if (hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking) != hipSuccess) h->side = nullptr;
    (void)hipEventCreateWithFlags(&h->ev_fork, hipEventDisableTiming);
    (void)hipEventCreateWithFlags(&h->ev_join, hipEventDisableTiming);
    build_pack(h);
    h->d_pack.alloc(h->Dc, false);
    (void)hipMemcpy(h->d_pack.p, h->h_pack.data(), sizeof(PackEntry) * h->Dc, hipMemcpyHostToDevice);
    h->d_radii.alloc(S, false);
    (void)hipMemcpy(h->d_radii.p, h->radii.data(), sizeof(double) * S, hipMemcpyHostToDevice);
    h->d_vs_sqrt.alloc(S + 1, false);  // (+ a zero for ghost atoms, slot S)
    std::vector<double> one(S + 1, 1.0);
    one[S] = 0.0;
    (void)hipMemcpy(h->d_vs_sqrt.p, one.data(), sizeof(double) * (S + 1), hipMemcpyHostToDevice);
    HarmCoef hc;
    host_build_harm_coef(&hc);
    upload_harm_coef(hc);
    h->d_grid.alloc(1024);  // the step's grid (128 B) + two cached {grid, cell} records by step parity (256 B each from 256)
    h->d_stat.alloc(4);
    h->d_bin_count.alloc(2 * SGPR_BIN_INTS);  // by step parity (a fused last kernel fills the next step's while this step's is cleared)
    h->d_cell_in.alloc(9);
    h->d_flag.alloc(8);  // [0..3] rebuild flags by step counter & 3, [4] count of rebuilds, [5] always zero
    h->d_cell0.alloc(18);  // cell at the last rebuild + its inverse
    if (const char *e = getenv("SGPR_SPIN_WAIT")) h->spin_wait = atoi(e) != 0;
    if (const char *e = getenv("SGPR_FUSE_NEXT")) h->fuse_next = atoi(e) != 0;
    if (const char *e = getenv("SGPR_GEMM_FUSED")) h->gemm_fused = atoi(e) != 0;
    if (const char *e = getenv("SGPR_ZERO_COPY")) h->zero_copy_out = atoi(e) != 0;
    if (const char *e = getenv("SGPR_COV_IN_REV")) h->cov_in_rev = atoi(e) != 0;
    if (const char *e = getenv("SGPR_GEMM_WAVES")) h->gemm_waves_k = atoi(e) == 8 ? 8 : 4;
    if (const char *e = getenv("SGPR_GEMM_WGS64")) h->gemm_wgs64 = atoi(e) == 3 ? 3 : 2;
    if (const char *e = getenv("SGPR_GEMM_HALF")) h->gemm_half = atoi(e) != 0;
    if (const char *e = getenv("SGPR_GEMM_64")) {
        int k = 0, w = 0;
        if (sscanf(e, "%d,%d", &k, &w) == 2) { h->gemm_k64 = k != 0; h->gemm_w64 = w != 0; h->gemm_64_forced = true; }
    }
    if (const char *e = getenv("SGPR_TILE_BALANCE")) h->tile_balance = atoi(e) != 0;
    if (const char *e = getenv("SGPR_TILE_CHAIN")) h->tile_chain = atoi(e);
    if (const char *e = getenv("SGPR_XCD_QUADS")) h->xcd_quads = atoi(e) != 0;
    if (const char *e = getenv("SGPR_ROWS16")) h->rows16 = atoi(e) != 0;
    if (const char *e = getenv("SGPR_ROWS16_MB")) h->rows16_mb = std::max(1, atoi(e));
    if (const char *e = getenv("SGPR_ROWS16_MIN")) h->rows16_min = std::max(1, atoi(e));
    {
        int ncu = 0;
        if (hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, h->device) == hipSuccess && ncu >= 8) h->cus_per_xcd = ncu / 8;
    }
    if (getenv("SGPR_STAMPS")) { h->d_stamps.alloc(8 * 4096); h->d_stamps2.alloc(8 * 8192); }
    if (const char *e = getenv("SGPR_QR_KEEP")) h->qr_keep_mode = std::min(std::max(atoi(e), 0), 2);
    *out = h;
    return SGPR_OK;
}

static void drop_graph(sgpr_model *h)
{
    if (h->gexec) (void)hipGraphExecDestroy(h->gexec);
    h->gexec = nullptr;
}

extern "C" void sgpr_destroy(sgpr_model *h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    (void)hipDeviceSynchronize();
    if (h->d_stamps.p && h->t_knm.n) {
        std::vector<long long> st(8 * h->t_knm.n);
        (void)hipMemcpy(st.data(), h->d_stamps.p, sizeof(long long) * st.size(), hipMemcpyDeviceToHost);
        double pro = 0, loop = 0, epi = 0; int n = 0; long long t0 = 1LL << 62, t1 = 0;
        for (size_t b = 0; b < h->t_knm.n; b++) {
            if (st[8 * b + 3] <= 0) continue;
            loop += st[8 * b + 1] - st[8 * b]; epi += st[8 * b + 2] - st[8 * b + 1]; n++;
            t0 = std::min(t0, st[8 * b]); t1 = std::max(t1, st[8 * b + 2]);
        }
        fprintf(stderr, "[sgpr stamps] K_nm gemm: %d tiles, start->loop end %.0f cyc, epilogue %.0f cyc, first start->last end %lld cyc\n",
                n, loop / std::max(n, 1), epi / std::max(n, 1), t1 - t0);
        (void)pro;
        if (const char *path = getenv("SGPR_STAMPS_FILE")) {  // every record, for tools/stamps_wcov.py
            if (FILE *f = fopen(path, "w")) {
                for (size_t b = 0; b < h->t_knm.n; b++)
                    fprintf(f, "knm %zu %lld %lld %lld %lld %lld %lld\n", b, st[8 * b], st[8 * b + 1], st[8 * b + 2], st[8 * b + 3], st[8 * b + 4], st[8 * b + 5]);
                std::vector<long long> s2(8 * std::min<size_t>(h->t_wcov.n, 8192));
                if (h->d_stamps2.p && !s2.empty()) {
                    (void)hipMemcpy(s2.data(), h->d_stamps2.p, sizeof(long long) * s2.size(), hipMemcpyDeviceToHost);
                    for (size_t b = 0; b < s2.size() / 8; b++)
                        fprintf(f, "wcov %zu %lld %lld %lld %lld %lld %lld\n", b, s2[8 * b], s2[8 * b + 1], s2[8 * b + 2], s2[8 * b + 3], s2[8 * b + 4], s2[8 * b + 5]);
                }
                fclose(f);
            }
        }
    }
    if (h->d_pstamps.p && h->N > 0) {
        const int N = h->N;
        std::vector<long long> st((size_t)16 * N);
        (void)hipMemcpy(st.data(), h->d_pstamps.p, sizeof(long long) * st.size(), hipMemcpyDeviceToHost);
        for (int pass = 0; pass < 2; pass++) {
            double d[7] = {0, 0, 0, 0, 0, 0, 0};
            long long t0 = 1LL << 62, t1 = 0;
            int cntw = 0;
            const int np = pass == 0 ? 6 : 3;
            for (int i = 0; i < h->cnt; i++) {
                const long long *w = st.data() + ((size_t)pass * N + i) * 8;
                if (w[0] <= 0 || w[np - 1] <= 0) continue;
                for (int k = 0; k + 1 < np; k++) d[k] += (double)(w[k + 1] - w[k]);
                t0 = std::min(t0, w[0]); t1 = std::max(t1, w[np - 1]);
                cntw++;
            }
            fprintf(stderr, "[sgpr stamps] %s: %d waves, cycles per phase:", pass == 0 ? "list+forward (sweep, sort, list out, tiles, c+spectrum)" : "reverse (dE/dc, pairs)", cntw);
            for (int k = 0; k + 1 < np; k++) fprintf(stderr, " %.0f", d[k] / std::max(cntw, 1));
            fprintf(stderr, "; first start -> last end %lld\n", t1 - t0);
            if (pass == 0) {   // a step that reused its candidates has no stamps 1, 2: start -> list, list -> c, c -> spectrum out
                double e[3] = {0, 0, 0};
                for (int i = 0; i < h->cnt; i++) {
                    const long long *w = st.data() + (size_t)i * 8;
                    if (w[0] <= 0 || w[5] <= 0) continue;
                    e[0] += (double)(w[3] - w[0]); e[1] += (double)(w[4] - w[3]); e[2] += (double)(w[5] - w[4]);
                }
                fprintf(stderr, "[sgpr stamps]    as a reuse step: list filter %.0f | c (radial, harmonics, MFMA) %.0f | spectrum + rows out %.0f\n",
                        e[0] / std::max(cntw, 1), e[1] / std::max(cntw, 1), e[2] / std::max(cntw, 1));
            }
        }
    }
    h->d_pstamps.release();
    drop_graph(h);
    for (auto e : h->ev) (void)hipEventDestroy(e);
    DevBuf<int> *ib[] = {&h->d_ind_slot, &h->d_ind_nn, &h->d_qoff, &h->d_perm, &h->d_slot, &h->d_aoff, &h->d_lslot,
                         &h->d_lnn, &h->d_bin_of, &h->d_bin_count, &h->d_nn_raw, &h->d_nn,
                         &h->d_nbr_j, &h->d_nbr_shift, &h->d_stat, &h->d_shear, &h->d_kslot, &h->d_aux,
                         &h->d_flag, &h->d_ncand, &h->d_cand_j, &h->d_cand_code, &h->d_cidx};
    for (auto b : ib) b->release();
    DevBuf<double> *db[] = {&h->d_radii, &h->d_Pm, &h->d_PmT, &h->d_pm_norm, &h->d_M, &h->d_mu, &h->d_choli,
                            &h->d_vs_sqrt, &h->d_gpart, &h->d_pos_in, &h->d_cell_in, &h->d_pos, &h->d_Pn, &h->d_norm, &h->d_C, &h->d_prec, &h->d_G,
                            &h->d_K, &h->d_Aw, &h->d_W, &h->d_F, &h->d_virpart, &h->d_Epart, &h->d_csq, &h->d_packed,
                            &h->d_rows_ones, &h->d_rows_out, &h->d_rows_ke, &h->d_L, &h->d_R1, &h->d_edit_tmp,
                            &h->d_rows_bG, &h->d_rows_bF, &h->d_rows_bV, &h->d_rows_kepart, &h->d_design, &h->d_qr_A, &h->d_qr_work};
    for (auto b : db) b->release();
    h->d_rows_cols.release();
    h->d_rows_rowof.release(); h->d_rows_qoff.release(); h->d_rows_vpart.release();
    if (h->pin) (void)hipHostFree(h->pin);
    if (h->pin_old) (void)hipHostFree(h->pin_old);
    if (h->md.halt_host) (void)hipHostFree(h->md.halt_host);
    if (h->md.mark) (void)hipHostFree(h->md.mark);
    if (h->md.scal_pin) (void)hipHostFree(h->md.scal_pin);
    {   // (a DevBuf has no destructor — handles are copied around as plain structs —: every buffer is released by name)
        MdState &m = h->md;
        DevBuf<double> *mdb[] = {&m.X, &m.V, &m.P, &m.KE, &m.mass, &m.sig, &m.noise, &m.noise_raw, &m.cell, &m.scal_d};
        for (auto b : mdb) b->release();
        m.halt.release();
        m.zeta.release();
        DevBuf<int4> *tb[] = {&h->t_knm, &h->t_w, &h->t_cov, &h->t_kmm, &h->t_wcov, &h->t_fused};
        for (auto b : tb) b->release();
        h->d_panel_cnt.release();
        h->d_stamps.release(); h->d_stamps2.release();
    }
    {
        DevBuf<double> *sd[] = {&h->sc_s2A, &h->sc_s2x, &h->sc_s2work, &h->sc_mv_v, &h->sc_mv_o, &h->sc_ra_y, &h->sc_ra_t, &h->sc_vs_t,
                                &h->sc_ai_er, &h->sc_ai_p, &h->sc_ai_norm, &h->sc_ai_krow, &h->sc_ai_kself, &h->sc_y, &h->sc_bA, &h->sc_bx,
                                &h->sc_bwork};
        for (auto b : sd) b->release();
        DevBuf<int> *si[] = {&h->sc_s2so, &h->sc_ai_eslot, &h->sc_ai_oslot, &h->sc_ai_onn, &h->sc_ai_info, &h->sc_sel_idx,
                             &h->sc_sel_map, &h->sc_chol_info};
        h->sc_sel_A.release(); h->sc_sel_work.release(); h->d_design_alt.release();
        h->sc_dense_part.release(); h->sc_ea_y.release(); h->sc_ea_rows.release();
        h->sc_mae_v.release(); h->sc_mae_rows.release();
        for (auto b : si) b->release();
        h->sc_ai_ptr.release();
        h->sc_erow.release();
        h->sc_ise.release();
    }
    for (auto &e : h->r1_cache) e.r1.release();
    h->s2k.VT.release(); h->s2k.R1.release(); h->sc_s2W.release(); h->sc_s2flag.release(); h->sc_potrf.release();
    for (auto &k : h->qr_keep) { k.clear_ops(); k.erows.release(); k.store.release(); k.Rc.release(); k.yt.release(); k.yraw.release(); k.ysnap.release(); k.vec.release(); }
    h->d_pack.release();
    h->t_covl.release();
    h->d_T.release();
    h->d_hm.release();
    h->d_pos0.release();
    h->d_cell0.release();
    h->d_b_rec.release();
    h->d_b_aux.release();
    h->d_grid.release();
    if (h->comm) (void)g_rccl.CommDestroy(h->comm);
    (void)sgpr_peer_destroy(h);
    h->d_xpacked.release();
    if (h->stream) (void)hipStreamDestroy(h->stream);
    if (h->side) (void)hipStreamDestroy(h->side);
    if (h->ev_fork) (void)hipEventDestroy(h->ev_fork);
    if (h->ev_join) (void)hipEventDestroy(h->ev_join);
    delete h;
}

// Working-tile tables (64x64 tiles).  kind 0: K_nm (rows = this rank's atoms, cols = inducing, full
// k); 1: W = Aw.Pm (cols = packed row, k = inducing range of the row tile's species); 2: covloss
// (cols = inducing, k = inducing range, clipped at the column tile when choli is lower-triangular);
// 3: K_mm.  Entries are dealt to list positions so that position % 8 == row tile % 8 (XCD affinity);
// holes are padding entries with kend = 0.
// Order of one XCD's share of a tile table.  The dispatcher hands workgroup p to XCD p % 8 and, while every CU still has a
// free slot, the XCD's j-th workgroup to its CU j % 32 (read off the HW_ID stamps of the grouped launch: the tiles at
// table positions p, p + 256, p + 512, p + 768 share a CU).  A plain longest-first order therefore stacks the long
// reductions: CU c gets sorted[c], sorted[c + 32], ... — at 4096 / 512 between 20 and 28 stages per CU around a mean of
// 23.6.  Here every round of 32 goes to the CUs in the order of their load so far, the shortest tile of the round to the
// most loaded CU (22 - 24 stages per CU).  Only the rounds that fill the free slots of an empty chip are placed; the rest
// (and a last partial round) stay longest-first and run wherever a slot frees up.
// `extra` / `cov_first` (the grouped W + covloss table, build_tiles): an entry may carry a SUCCESSOR (index + 1 into `extra` in
// the high half of .y; its reduction counts with the entry's), and of two entries of one length the covloss one (bit 16 of
// .x: the longer epilogue) is dispatched first.
static void balance_xcd_share(std::vector<int4> &b, int ncu, int slots, const std::vector<int4> *extra = nullptr, bool cov_first = false)
{
    auto len = [&](const int4 &t) {
        int l = t.w - t.z;
        const unsigned sidx = (unsigned)t.y >> 16;
        if (extra && sidx) l += (*extra)[sidx - 1].w - (*extra)[sidx - 1].z;
        return l;
    };
    auto cov = [&](const int4 &t) { return cov_first && ((t.x >> 16) & 1); };
    std::stable_sort(b.begin(), b.end(), [&](const int4 &p, const int4 &q) { return len(p) != len(q) ? len(p) > len(q) : cov(p) > cov(q); });
    if (ncu <= 1) return;
    std::vector<long long> load(ncu, 0);
    std::vector<int> order(ncu);
    for (int r = 0; r < slots && (size_t)(r + 1) * ncu <= b.size(); r++) {
        for (int c = 0; c < ncu; c++) order[c] = c;
        std::stable_sort(order.begin(), order.end(), [&](int p, int q) { return load[p] > load[q]; });
        std::vector<int4> round(b.begin() + (size_t)r * ncu, b.begin() + (size_t)(r + 1) * ncu);
        std::stable_sort(round.begin(), round.end(), [&](const int4 &p, const int4 &q) { return len(p) > len(q); });  // descending
        for (int k = 0; k < ncu; k++) {
            const int4 t = round[ncu - 1 - k];  // ascending: the shortest to the most loaded
            b[(size_t)r * ncu + order[k]] = t;
            load[order[k]] += len(t);
        }
    }
}

// number of working tiles of product `kind` (0 K_nm, 1 W, 2 covloss) at `bm` rows per tile: the loop of build_tiles
static size_t count_tiles(const sgpr_model *h, int kind, int bm)
{
    const int nrows = h->cnt, ncols = kind == 1 ? h->Dpad : h->m;
    const int nrt = (nrows + bm - 1) / bm, nct = (ncols + 63) / 64;
    size_t n = 0;
    for (int rt = 0; rt < nrt; rt++) {
        const int r0 = rt * bm, r1 = std::min(nrows, r0 + bm) - 1;
        int sa = 0, sb = 0;
        while (sa + 1 < h->S && h->aoff[sa + 1] <= r0) sa++;
        while (sb + 1 < h->S && h->aoff[sb + 1] <= r1) sb++;
        const int qlo = h->qoff[sa], qhi = h->qoff[sb + 1];
        if (kind == 1) { n += qhi > qlo ? nct : 0; continue; }
        for (int ct = 0; ct < nct; ct++)
            if (ct * 64 < qhi && std::min(ncols, ct * 64 + 64) > qlo) n++;
    }
    return n;
}

// Tile heights of the three products of a step.  32 x 64 tiles fill the chip at a few thousand atoms (416 K_nm tiles on
// 256 CUs at 4096 / 512); once a CU holds several of them, 64 x 64 tiles on eight waves move two thirds of the bytes
// for the same flops and win: K_nm -5 % at 8000 / 512, -7 % at 15625 / 512, -10 % at 32768 / 1024 (+1..4 % at
// 4096 / 512: fewer tiles than CUs).  Round 6: those tiles on THREE register stage sets, three workgroups per CU
// (gemm_tile_body8r64<EPI, 3>: a tile's 9k cycles without an MFMA — entry, first arrival, epilogue — are covered by two
// other tiles instead of one): K_nm -4 % and W + covloss -4.5 % at 16384 / 1024, -8 % at 32768 / 512; and with three
// workgroups per CU the 64-row form wins for W + covloss from ~1400 32-row tiles on (it took 6144): -10 % at 4096 / 1024,
// -12 % at 8000 / 512 and 10648 / 512, -7 % at 10648 / 1024, -11 % at 15625 / 1024, -3 % at 5832 / 512; 4096 / 512
// (1056 tiles) stays on the 32-row form (21.3 against 21.7 us).  Measured and not kept: 128 x 64 tiles (four blocks per
// wave, 72 KB of LDS, two workgroups per CU): within +-3 % of the three-workgroup 64-row form at every size (tools/ab_sizes.sh).
// SGPR_GEMM_64="k,w" (0 / 1 each) and SGPR_GEMM_WGS64=2 override.
static void decide_tile_heights(sgpr_model *h)
{
    if (h->gemm_64_forced) return;
    const size_t ncu = (size_t)h->cus_per_xcd * 8;
    h->gemm_k64 = count_tiles(h, 0, 32) >= 3 * ncu;
    h->gemm_w64 = 2 * (count_tiles(h, 1, 32) + count_tiles(h, 2, 32)) >= 11 * ncu;
}

// The fused table: the K_nm tiles in their own order (so that their energy partials keep their slots), tagged kind 2,
// then the grouped W + covloss table with the number of K_nm tiles of each entry's row panel in bits 20..27.  Only for the
// 32-row / 16-deep forms of all three products (the row panels of producer and consumers must coincide).
static int build_fused_tiles(sgpr_model *h)
{
    h->t_fused.release();
    h->fuse_epoch = 0;
    if (h->gemm_bm_k != 32 || h->gemm_bm_w != 32 || h->gemm_kd_k != 16 || h->gemm_kd_w != 16 || h->gemm_k64 || h->gemm_w64) return 0;
    if (h->h_t_knm.empty() || h->h_t_both.empty()) return 0;
    const int nrt = (h->cnt + 31) / 32;
    std::vector<int> need(std::max(nrt, 1), 0);
    std::vector<int4> fused;
    fused.reserve(h->h_t_knm.size() + h->h_t_both.size());
    for (const int4 &t : h->h_t_knm) {
        if (t.w > t.z) need[t.x] += 1;
        fused.push_back(make_int4(t.x | (2 << 16), t.y, t.z, t.w));
    }
    for (const int4 &t : h->h_t_both) {
        if (t.w <= t.z) { fused.push_back(t); continue; }
        const int rt = t.x & 0xffff;
        if (need[rt] <= 0 || need[rt] > 255) { h->t_fused.release(); return 0; }   // (no producer / beyond the field: stay unfused)
        fused.push_back(make_int4(t.x | (need[rt] << 20), t.y, t.z, t.w));
    }
    if (h->t_fused.alloc(fused.size(), false) || h->d_panel_cnt.alloc((size_t)32 * std::max(nrt, 1))) return -1;
    if (hipMemcpy(h->t_fused.p, fused.data(), sizeof(int4) * fused.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
    if (hipMemset(h->d_panel_cnt.p, 0, h->d_panel_cnt.n * sizeof(int)) != hipSuccess) return -1;
    return 0;
}

static int build_tiles(sgpr_model *h, int kind)
{
    const int KTc = 32;
    const std::vector<int> &roff = kind == 3 ? h->qoff : h->aoff;
    const int nrows = kind == 3 ? h->m : h->cnt;
    const int ncols = kind == 1 ? h->Dpad : h->m;
    // 32-row tiles with 16-deep stages (four workgroups per CU) for the three products of a step at every size
    // measured (4096 / 512: K_nm 22.2 -> 18-20 us, W + covloss 24.9 -> 22.6 us; 32768 / 1024: 198 -> 184 and
    // 313 -> 287 us against the 64-row tiles); K_mm keeps the 64-row form
    int bm = kind == 3 ? 64 : 32;
    if (const char *e = getenv("SGPR_GEMM_BM")) {  // experiment: "k,w" tile heights
        int bk = 0, bw = 0;
        if (sscanf(e, "%d,%d", &bk, &bw) == 2 && kind != 3) bm = kind == 0 ? bk : bw;
    }
    if (const char *e = getenv("SGPR_GEMM_KD")) {
        int kk = 0, kw = 0;
        if (sscanf(e, "%d,%d", &kk, &kw) == 2) { h->gemm_kd_k = kk == 16 ? 16 : 32; h->gemm_kd_w = kw == 16 ? 16 : 32; }
    }
    if (kind == 0 && h->gemm_k64) bm = 64;
    if ((kind == 1 || kind == 2) && h->gemm_w64) bm = 64;
    // Fewer 32-row tiles than HALF the CUs (a rank's share of a sharded frame, frames of a few hundred atoms): a tile alone on its CU
    // runs at the pace of its own chain of dependent MFMAs, and half tiles — 16 x 64 on four waves, one per SIMD, twice as many
    // CUs — halve that chain (gemm_tile_body8<EPI, true>; same sums, same bits).  SGPR_GEMM_HALF=0 keeps the 32-row tiles.
    if (kind != 3 && bm == 32 && h->gemm_half && !getenv("SGPR_GEMM_BM")) {
        const size_t ncu = (size_t)h->cus_per_xcd * 8;
        const size_t n32 = kind == 0 ? count_tiles(h, 0, 32) : count_tiles(h, 1, 32) + count_tiles(h, 2, 32);
        // (while nearly every half tile finds a CU of its own — 132 W + covloss tiles of rank 0's share at world 8: 12.7 -> 10.7 us,
        // 104 K_nm tiles at world 4: 14.0 -> 12.1; at 208 tiles, world 2, two half tiles per CU are a whole tile again: 14.6 -> 15.8)
        // (K_nm: only while EVERY half tile is alone — at world 4 the 0.9 us its 104 tiles gain go back to the forward kernel,
        // whose workgroup-to-atom mapping follows the tile height)
        if (5 * n32 <= (kind == 0 ? 2 : 3) * ncu) bm = 16;
    }
    if (kind == 0) h->gemm_bm_k = bm;
    if (kind == 1 || kind == 2) h->gemm_bm_w = bm;
    const int nrt = (nrows + bm - 1) / bm, nct = (ncols + 63) / 64;
    std::vector<std::vector<int4>> bucket(8);
    auto species_of = [&](const std::vector<int> &off, int idx) {
        int s = 0;
        while (s + 1 < h->S && off[s + 1] <= idx) s++;
        return s;
    };
    for (int rt = 0; rt < nrt; rt++) {
        const int r0 = rt * bm, r1 = std::min(nrows, r0 + bm) - 1;
        const int sa = species_of(roff, r0), sb = species_of(roff, r1);
        const int qlo = h->qoff[sa], qhi = h->qoff[sb + 1];  // inducing range of these species
        for (int ct = 0; ct < nct; ct++) {
            const int c0 = ct * 64, c1 = std::min(ncols, c0 + 64);
            int kb = 0, ke = 0;
            if (kind == 0 || kind == 3) {
                if (c0 >= qhi || c1 <= qlo) continue;
                kb = 0; ke = h->Dpad;
            } else if (kind == 1) {
                kb = qlo; ke = qhi;
            } else {
                if (c0 >= qhi || c1 <= qlo) continue;
                kb = qlo; ke = h->choli_lower ? std::min(qhi, c1) : qhi;
            }
            kb = kb / KTc * KTc;
            ke = (ke + KTc - 1) / KTc * KTc;
            if (ke <= kb) continue;
            bucket[rt % 8].push_back(make_int4(rt, ct, kb, ke));
        }
    }
    size_t depth = 0;
    for (auto &b : bucket) depth = std::max(depth, b.size());
    std::vector<int4> list(depth * 8, make_int4(0, 0, 0, 0));
    for (int x = 0; x < 8; x++)
        for (size_t j = 0; j < bucket[x].size(); j++) list[j * 8 + x] = bucket[x][j];
    DevBuf<int4> &dst = kind == 0 ? h->t_knm : kind == 1 ? h->t_w : kind == 2 ? h->t_cov : h->t_kmm;
    if (kind == 0) h->h_t_knm = list;
    if (kind == 1) h->h_t_w = list;
    if (kind == 2) h->h_t_cov = list;
    if (kind == 1 || kind == 2) {
        // grouped table W + covloss: same XCD rule (position % 8 == row tile % 8), covloss entries tagged
        std::vector<std::vector<int4>> bk(8);
        for (const int4 &t : h->h_t_w)
            if (t.w > t.z) bk[t.x % 8].push_back(t);
        for (const int4 &t : h->h_t_cov)
            if (t.w > t.z) bk[t.x % 8].push_back(make_int4(t.x | (1 << 16), t.y, t.z, t.w));
        // longest reductions first (LPT): row tiles come in species order and the species with the most
        // inducing points — the deepest reductions — would otherwise form the tail of the launch.
        // What the per-tile stamps of the 32-row form showed (tools/stamps_wcov.py, round 5): the four tiles of a CU share
        // the matrix pipe OLDEST WAVE FIRST, so they finish in the order they were dispatched whatever their length (the
        // longest, first in the table, at 28k cycles; the shortest, starved until then, at 33-34k); a CU's finish time goes
        // with its stage count (1450 cycles per 32-deep stage) as long as nothing runs alone at the end; and a tile beyond
        // the slots x CUs that start with the launch (1056 tiles for 1024 slots at 4096 / 512) waits for a slot until 34k
        // and then runs its entry, first loads and epilogue alone: those 32 CUs ended at 42-43k, the others at 34-37k, and
        // the launch lasts as long as they do.  So no tile waits for a slot: each overflow tile (the shortest of the XCD's
        // share) is run by the OLDEST workgroup of one of the least loaded CUs AHEAD of that workgroup's own tile — at full
        // priority, out of the way after 5k cycles; the entry's .y carries the position of its successor + 1 in the high
        // half and the launch has `wcov_grid` workgroups.  (Behind the first tile to finish, the overflow tile ran from 30k
        // to 39k; paired with another short tile in a young workgroup, from 32k to 41k.)  SGPR_TILE_CHAIN=0: one workgroup
        // per tile.
        const bool chain = h->tile_chain && h->tile_balance && !h->gemm_w64;
        std::vector<std::vector<int4>> extra(8);
        for (int x = 0; x < 8; x++) {
            auto &b = bk[x];
            balance_xcd_share(b, h->tile_balance ? h->cus_per_xcd : 1, h->gemm_w64 ? h->gemm_wgs64 : 4, nullptr, chain);
            const size_t ncu_x = (size_t)h->cus_per_xcd, cap = 4 * ncu_x;
            if (!chain || b.size() <= cap || b.size() - cap > ncu_x) continue;
            std::vector<long long> load(ncu_x, 0);
            for (size_t r = 0; r < 4; r++)
                for (size_t c = 0; c < ncu_x; c++) load[c] += b[r * ncu_x + c].w - b[r * ncu_x + c].z;
            std::vector<bool> taken(ncu_x, false);
            for (size_t j = cap; j < b.size(); j++) {
                size_t c = ncu_x;
                for (size_t k = 0; k < ncu_x; k++)
                    if (!taken[k] && (c == ncu_x || load[k] < load[c])) c = k;
                taken[c] = true;
                extra[x].push_back(b[c]);                              // round 0 of CU c: its oldest (longest) tile, now the successor
                b[c] = b[j];
                b[c].y |= (int)(extra[x].size() << 16);
                load[c] += b[j].w - b[j].z;
            }
            b.resize(cap);
        }
        size_t dp = 0;
        for (auto &b : bk) dp = std::max(dp, b.size());
        std::vector<int4> both(dp * 8, make_int4(0, 0, 0, 0));
        for (int x = 0; x < 8; x++)
            for (size_t j = 0; j < bk[x].size(); j++) both[j * 8 + x] = bk[x][j];
        {   // the covloss tiles alone, same XCD rule, longest first
            std::vector<std::vector<int4>> bc(8);
            for (const int4 &t : h->h_t_cov)
                if (t.w > t.z) bc[t.x % 8].push_back(t);
            for (auto &b : bc) balance_xcd_share(b, h->tile_balance ? h->cus_per_xcd : 1, h->gemm_w64 ? h->gemm_wgs64 : 4);
            size_t dc = 0;
            for (auto &b : bc) dc = std::max(dc, b.size());
            std::vector<int4> only(dc * 8, make_int4(0, 0, 0, 0));
            for (int x = 0; x < 8; x++)
                for (size_t j = 0; j < bc[x].size(); j++) only[j * 8 + x] = bc[x][j];
            h->t_covl.release();
            if (!only.empty()) {
                if (h->t_covl.alloc(only.size(), false)) return -1;
                if (hipMemcpy(h->t_covl.p, only.data(), sizeof(int4) * only.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
            }
        }
        h->wcov_grid = (int)both.size();
        {
            // the successors behind the dispatched part; links become table positions
            size_t base[8];
            for (int x = 0; x < 8; x++) { base[x] = both.size(); both.insert(both.end(), extra[x].begin(), extra[x].end()); }
            // (successors exist only in tables of little more than slots x CUs entries: positions fit the half word by far;
            // a table without successors — every larger frame — carries no links and may have any size)
            for (size_t ppos = 0; ppos < (size_t)h->wcov_grid; ppos++) {
                const unsigned sidx = (unsigned)both[ppos].y >> 16;
                if (sidx) both[ppos].y = (both[ppos].y & 0xffff) | (int)((base[ppos % 8] + sidx) << 16);
            }
            h->h_t_both = both;
            for (auto &t : h->h_t_both) t.y &= 0xffff;   // (the fused table has one workgroup per entry)
        }
        h->t_wcov.release();
        if (!both.empty()) {
            if (h->t_wcov.alloc(both.size(), false)) return -1;
            if (hipMemcpy(h->t_wcov.p, both.data(), sizeof(int4) * both.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
        }
        if (build_fused_tiles(h)) return -1;
    }
    dst.release();
    if (list.empty()) return 0;
    if (dst.alloc(list.size(), false)) return -1;
    if (hipMemcpy(dst.p, list.data(), sizeof(int4) * list.size(), hipMemcpyHostToDevice) != hipSuccess) return -1;
    return 0;
}

static GemmParams knm_params(sgpr_model *h, const double *A, int M, const int *row_slot, const int *row_nn,
                             const DevBuf<int4> &tiles, double *Kout, double *Aw, const double *mu, double *Epart);

static void gemm_kernel_pm(sgpr_model *h, const double *A, int M, const int *row_slot, const int *row_nn,
                           const DevBuf<int4> &tiles, double *Kout, double *Aw, const double *mu, double *Epart,
                           hipStream_t st)
{
    launch_gemm_nt(knm_params(h, A, M, row_slot, row_nn, tiles, Kout, Aw, mu, Epart), EPI_KERNEL, st);
}

static GemmParams knm_params(sgpr_model *h, const double *A, int M, const int *row_slot, const int *row_nn,
                             const DevBuf<int4> &tiles, double *Kout, double *Aw, const double *mu, double *Epart)
{
    GemmParams g = {};
    g.M = M; g.N = h->m; g.K = h->Dpad;
    g.lda = h->Dpad; g.ldb = h->Dpad; g.ldc = h->m_pad;
    g.A = A; g.B = h->d_Pm.p; g.C = Kout;
    g.tiles = tiles.p; g.ntiles = (int)tiles.n;
    g.bm = (&tiles == &h->t_kmm) ? 64 : h->gemm_bm_k;
    g.kd = h->gemm_kd_k;
    g.waves = &tiles != &h->t_knm ? 4 : h->gemm_k64 ? 8 : (h->gemm_bm_k == 32 && h->gemm_kd_k == 16) ? h->gemm_waves_k : 4;
    g.wgs = h->gemm_wgs64;
    g.eta = h->eta; g.lone_m1 = h->lone_w - 1.0; g.mu = mu; g.row_nn = row_nn; g.col_nn = h->d_ind_nn.p; g.Aw = Aw; g.Esum = Epart;
    g.row_slot = row_slot; g.col_slot = h->d_ind_slot.p;
    g.stamps = h->d_stamps.p ? h->d_stamps.p : nullptr;
    return g;
}

static int alloc_work(sgpr_model *h);
static int design_after_set(sgpr_model *h);   // data.inc: the resident design matrix follows the inducing set
static int design_after_add(sgpr_model *h);
static void design_after_pop_last(sgpr_model *h);
static int design_select(sgpr_model *h, int count, const int32_t *idx);

extern "C" int sgpr_set_inducing(sgpr_model *h, int m, const int32_t *zc, const int64_t *nbr_ptr,
                                 const int32_t *nbr_z, const double *nbr_r)
{
    if (!h || m < 0 || (m > 0 && (!zc || !nbr_ptr))) return fail(SGPR_E_INVALID, "sgpr_set_inducing: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    // validate everything first: on an error the handle keeps its previous inducing set untouched
    std::vector<int> slot(m);
    for (int q = 0; q < m; q++) {
        slot[q] = slot_of(h, zc[q]);
        if (slot[q] < 0) return fail(SGPR_E_SPECIES, "inducing LCE %d: Z=%d is not in the species table", q, zc[q]);
        if (!h->ignore_unknown)
            for (int64_t e = nbr_ptr[q]; e < nbr_ptr[q + 1]; e++)
                if (slot_of(h, nbr_z[e]) < 0)
                    return fail(SGPR_E_SPECIES, "inducing LCE %d: neighbour Z=%d is not in the species table", q, nbr_z[e]);
    }
    // sizes and host tables of the NEW set in locals; the device arrays are allocated before anything in the handle
    // changes, so a failed allocation leaves the previous inducing set fully usable
    const int m_pad_n = rup(std::max(m, 1), 32), m_rows_n = rup(std::max(m, 1), 64);
    std::vector<int> ind_perm_n(m);
    std::iota(ind_perm_n.begin(), ind_perm_n.end(), 0);
    std::stable_sort(ind_perm_n.begin(), ind_perm_n.end(), [&](int a, int b) { return slot[a] < slot[b]; });
    std::vector<int> ind_slot_n(m), qoff_n(h->S + 1, 0);
    std::vector<int64_t> ptr(m + 1, 0);
    std::vector<int> eslot;
    std::vector<double> er;
    std::vector<int> nn(m_rows_n, 1), dslot(m_rows_n, -2);
    for (int k = 0; k < m; k++) {
        const int q = ind_perm_n[k];
        ind_slot_n[k] = slot[q];
        dslot[k] = slot[q];
        qoff_n[slot[q] + 1]++;
        const int64_t a = nbr_ptr[q], b = nbr_ptr[q + 1];
        const size_t before = eslot.size();
        for (int64_t e = a; e < b; e++) {
            const int s = slot_of(h, nbr_z[e]);
            if (s < 0) continue;  // descriptor/sesoap.py:343-346 (ignore_unknown; anything else was refused above)
            eslot.push_back(s);
            er.push_back(nbr_r[3 * e]); er.push_back(nbr_r[3 * e + 1]); er.push_back(nbr_r[3 * e + 2]);
        }
        ptr[k + 1] = (int64_t)eslot.size();
        nn[k] = (int)(eslot.size() - before);
    }
    for (int s = 0; s < h->S; s++) qoff_n[s + 1] += qoff_n[s];
    DevBuf<int> n_ind_slot, n_ind_nn, n_qoff;
    DevBuf<double> n_Pm, n_PmT, n_pm_norm, n_M, n_mu, n_choli;
    if (n_ind_slot.alloc(m_rows_n) || n_ind_nn.alloc(m_rows_n) || n_qoff.alloc(h->S + 1) ||
        n_Pm.alloc((size_t)m_rows_n * h->Dpad) || n_PmT.alloc((size_t)rup(h->Dpad, 64) * m_pad_n) ||
        n_pm_norm.alloc(m_rows_n) || n_M.alloc((size_t)m_rows_n * m_pad_n) || n_mu.alloc(std::max(m_pad_n, m_rows_n)) ||
        n_choli.alloc((size_t)m_rows_n * m_pad_n)) {
        n_ind_slot.release(); n_ind_nn.release(); n_qoff.release(); n_Pm.release(); n_PmT.release(); n_pm_norm.release();
        n_M.release(); n_mu.release(); n_choli.release();
        return fail(SGPR_E_NODEVICE, "hipMalloc failed (inducing set): the previous set is untouched");
    }
    // ---- commit
    drop_graph(h);
    h->has_mu = h->has_choli = false;
    h->chol_valid = h->r1_valid = false;
    h->chol_shape = false;
    h->m = m; h->m_pad = m_pad_n; h->m_rows = m_rows_n;
    h->ind_perm.swap(ind_perm_n); h->ind_slot.swap(ind_slot_n); h->qoff.swap(qoff_n);
    auto take = [](auto &dst, auto &src) { dst.release(); dst = src; src.p = nullptr; src.n = 0; };
    take(h->d_ind_slot, n_ind_slot); take(h->d_ind_nn, n_ind_nn); take(h->d_qoff, n_qoff); take(h->d_Pm, n_Pm);
    take(h->d_PmT, n_PmT); take(h->d_pm_norm, n_pm_norm); take(h->d_M, n_M); take(h->d_mu, n_mu); take(h->d_choli, n_choli);
    {   // keep the caller's list for the edit entry points (copy first: the inputs may alias h->env_*)
        const int64_t lo = m > 0 ? nbr_ptr[0] : 0, hi = m > 0 ? nbr_ptr[m] : 0;
        std::vector<int32_t> ezc(zc, zc + m), ez(nbr_z + lo, nbr_z + hi);
        std::vector<double> evr(nbr_r + 3 * lo, nbr_r + 3 * hi);
        std::vector<int64_t> eptr(m + 1, 0);
        for (int q = 0; q < m; q++) eptr[q + 1] = nbr_ptr[q + 1] - lo;
        h->env_zc.swap(ezc); h->env_z.swap(ez); h->env_ptr.swap(eptr); h->env_r.swap(evr);
    }
    HIPCHK(hipMemcpy(h->d_ind_slot.p, dslot.data(), sizeof(int) * h->m_rows, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_ind_nn.p, nn.data(), sizeof(int) * h->m_rows, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_qoff.p, h->qoff.data(), sizeof(int) * (h->S + 1), hipMemcpyHostToDevice));
    if (m == 0) return SGPR_OK;
    DevBuf<int64_t> d_ptr;
    DevBuf<int> d_eslot;
    DevBuf<double> d_er;
    if (d_ptr.alloc(m + 1, false) || d_eslot.alloc(eslot.size(), false) || d_er.alloc(er.size(), false))
        return fail(SGPR_E_NODEVICE, "hipMalloc failed (inducing environments)");
    HIPCHK(hipMemcpy(d_ptr.p, ptr.data(), sizeof(int64_t) * (m + 1), hipMemcpyHostToDevice));
    if (!eslot.empty()) {
        HIPCHK(hipMemcpy(d_eslot.p, eslot.data(), sizeof(int) * eslot.size(), hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_er.p, er.data(), sizeof(double) * er.size(), hipMemcpyHostToDevice));
    }
    DescParams dp = {};
    dp.lmax = h->lmax; dp.nmax = h->nmax; dp.S = h->S; dp.N = m; dp.Nall = m; dp.first = 0; dp.stride = 1;
    dp.maxnn = 0; dp.Dc = h->Dc; dp.Dpad = h->Dpad; dp.CS = h->CS; dp.rc = h->rc;
    for (int k = 0; k < SGPR_MAX_S; k++) dp.radii_v[k] = k < h->S ? h->radii[k] : 1.0;
    const int rcd = launch_descriptor_forward_env(dp, d_ptr.p, d_eslot.p, d_er.p, h->d_radii.p, h->d_pack.p,
                                                  h->d_Pm.p, h->d_pm_norm.p, h->stream);
    if (rcd) return fail(SGPR_E_UNSUPPORTED, "descriptor kernel for (lmax,nmax,S)=(%d,%d,%d) not compiled in", h->lmax, h->nmax, h->S);
    hipLaunchKernelGGL(transpose_kernel, dim3((h->Dpad + 31) / 32, (m + 31) / 32), dim3(32, 8), 0, h->stream, m,
                       h->Dpad, h->d_Pm.p, h->Dpad, h->d_PmT.p, h->m_pad);
    // K_mm (regression/gppotential.py:506): same kernel epilogue, no weights
    build_tiles(h, 3);
    gemm_kernel_pm(h, h->d_Pm.p, m, h->d_ind_slot.p, h->d_ind_nn.p, h->t_kmm, h->d_M.p, nullptr, nullptr, nullptr,
                   h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipGetLastError());
    d_ptr.release(); d_eslot.release(); d_er.release();
    if (h->N > 0) {  // a system is bound: K/Aw buffers and tile tables follow m
        const int rc_ = alloc_work(h);
        if (rc_) return rc_;
    }
    return design_after_set(h);
}

extern "C" int sgpr_get_kmm(sgpr_model *h, double *M)
{
    if (!h || !M) return fail(SGPR_E_INVALID, "sgpr_get_kmm: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const int m = h->m;
    std::vector<double> buf((size_t)h->m_rows * h->m_pad);
    HIPCHK(hipMemcpy(buf.data(), h->d_M.p, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    for (int a = 0; a < m; a++)
        for (int b = 0; b < m; b++) M[(size_t)h->ind_perm[a] * m + h->ind_perm[b]] = buf[(size_t)a * h->m_pad + b];
    return SGPR_OK;
}

// diag(K_mm) alone, caller order (make_stats' kern_diag_mean, gppotential.py:644-649: no need for the m x m matrix)
extern "C" int sgpr_get_kmm_diag(sgpr_model *h, double *diag)
{
    if (!h || !diag) return fail(SGPR_E_INVALID, "sgpr_get_kmm_diag: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const int m = h->m;
    if (m <= 0) return SGPR_OK;
    std::vector<double> d(m);
    HIPCHK(hipMemcpy2D(d.data(), sizeof(double), h->d_M.p, sizeof(double) * (h->m_pad + 1), sizeof(double), m, hipMemcpyDeviceToHost));
    for (int a = 0; a < m; a++) diag[h->ind_perm[a]] = d[a];
    return SGPR_OK;
}

// Row sums of K_mm in the CALLER's order of inducing LCEs, each summed the way numpy sums a contiguous row (its pairwise
// scheme: eight running sums over blocks of at most 128 elements, halves split at a multiple of eight), so that
// `argsort(M.sum(axis=1))` of the downsizing rule (gppotential.py:815-842, lii) picks the same LCEs whether the matrix is
// summed here or fetched and summed on the host.  One thread per row; inv[c] = sorted index of caller index c.
__device__ static double numpy_pairwise(const double *row, const int *inv, int lo, int n)
{
    if (n < 8) {
        double r = 0.0;
        for (int i = 0; i < n; i++) r += row[inv[lo + i]];
        return r;
    }
    if (n <= 128) {
        double r[8];
        for (int j = 0; j < 8; j++) r[j] = row[inv[lo + j]];
        int i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; j++) r[j] += row[inv[lo + i + j]];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; i++) res += row[inv[lo + i]];
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return numpy_pairwise(row, inv, lo, n2) + numpy_pairwise(row, inv, lo + n2, n - n2);
}

__global__ void kmm_rowsum_kernel(int m, const double *M, int ld, const int *inv, double *out)
{
    const int c = blockIdx.x * blockDim.x + threadIdx.x;   // caller index of the row
    if (c >= m) return;
    out[c] = numpy_pairwise(M + (size_t)inv[c] * ld, inv, 0, m);
}

extern "C" int sgpr_get_kmm_rowsum(sgpr_model *h, double *sums)
{
    if (!h || !sums) return fail(SGPR_E_INVALID, "sgpr_get_kmm_rowsum: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const int m = h->m;
    if (m <= 0) return SGPR_OK;
    std::vector<int> inv(m);
    for (int a = 0; a < m; a++) inv[h->ind_perm[a]] = a;
    ScopedBuf<int> d_inv;
    ScopedBuf<double> d_out;
    if (d_inv.alloc(m) || d_out.alloc(m)) return fail(SGPR_E_NODEVICE, "hipMalloc failed");
    HIPCHK(hipMemcpyAsync(d_inv.p, inv.data(), sizeof(int) * m, hipMemcpyHostToDevice, h->stream));
    hipLaunchKernelGGL(kmm_rowsum_kernel, dim3((m + 63) / 64), dim3(64), 0, h->stream, m, h->d_M.p, h->m_pad, d_inv.p, d_out.p);
    HIPCHK(hipMemcpyAsync(sums, d_out.p, sizeof(double) * m, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return SGPR_OK;
}

extern "C" int sgpr_get_inducing_descriptors(sgpr_model *h, double *P)
{
    if (!h || !P) return fail(SGPR_E_INVALID, "sgpr_get_inducing_descriptors: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const size_t row = (size_t)h->S * h->S * h->D;
    DevBuf<double> d;
    if (d.alloc(row * std::max(h->m, 1))) return fail(SGPR_E_NODEVICE, "hipMalloc failed");
    launch_unpack_descriptors(h->m, h->S, h->lmax, h->nmax, h->Dc, h->Dpad, h->d_pack.p, h->d_Pm.p, d.p, h->stream);
    std::vector<double> buf(row * h->m);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(buf.data(), d.p, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    for (int k = 0; k < h->m; k++) memcpy(P + row * h->ind_perm[k], buf.data() + row * k, sizeof(double) * row);
    d.release();
    return SGPR_OK;
}

static void upload_vscale(sgpr_model *h)
{
    std::vector<double> s(h->S + 1, 0.0);
    for (int k = 0; k < h->S; k++) s[k] = sqrt(h->vscale[k]);
    (void)hipMemcpy(h->d_vs_sqrt.p, s.data(), sizeof(double) * (h->S + 1), hipMemcpyHostToDevice);
}

extern "C" int sgpr_set_weights(sgpr_model *h, const double *mu, const double *mean_w, const double *vscale,
                                const double *choli)
{
    if (!h || !mu) return fail(SGPR_E_INVALID, "sgpr_set_weights: bad arguments");
    if (h->m <= 0) return fail(SGPR_E_NOMODEL, "sgpr_set_weights: no inducing set");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int m = h->m;
    std::vector<double> mus(h->m_pad, 0.0);
    for (int k = 0; k < m; k++) mus[k] = mu[h->ind_perm[k]];
    HIPCHK(hipMemcpy(h->d_mu.p, mus.data(), sizeof(double) * h->m_pad, hipMemcpyHostToDevice));
    h->has_mu = true;
    for (int k = 0; k < h->S; k++) {
        h->mean_w[k] = mean_w ? mean_w[k] : 0.0;
        h->vscale[k] = vscale ? vscale[k] : 1.0;
    }
    upload_vscale(h);
    if (choli) {
        std::vector<double> c((size_t)h->m_rows * h->m_pad, 0.0);
        for (int a = 0; a < m; a++)
            for (int b = 0; b < m; b++)
                c[(size_t)a * h->m_pad + b] = choli[(size_t)h->ind_perm[a] * m + h->ind_perm[b]];
        // K_mm is block-diagonal by species, so L and L^-1 have no cross-species entries and the
        // stable species sort keeps each block lower-triangular.  A caller may still hand over
        // a general matrix: detect, and only then give up the triangular trimming.
        bool lower = true;
        for (int a = 0; a < m && lower; a++)
            for (int b = a + 1; b < m; b++)
                if (c[(size_t)a * h->m_pad + b] != 0.0) { lower = false; break; }
        h->has_choli = true;
        if (h->choli_lower != lower) {
            h->choli_lower = lower;
            if (h->N > 0 && build_tiles(h, 2)) return fail(SGPR_E_NODEVICE, "hipMalloc failed (tile table)");
        }
        HIPCHK(hipMemcpy(h->d_choli.p, c.data(), sizeof(double) * c.size(), hipMemcpyHostToDevice));
    } else
        h->has_choli = false;
    // the mean term depends on the bound system (rank 0 carries it)
    h->mean_energy = 0.0;
    if (h->rank == 0)
    for (int z : h->numbers) {
        const int s = slot_of(h, z);
        if (s >= 0) h->mean_energy += h->mean_w[s];
    }
    drop_graph(h);
    return SGPR_OK;
}

// The mean offsets and the per-species variance scale alone: mu and choli stay what the last sgpr_solve /
// sgpr_data_solve / sgpr_resolve left on the device (make_munu ends by installing exactly those,
// gppotential.py:548-605 — there is no reason to carry an m x m matrix over PCIe and back for it).
extern "C" int sgpr_set_mean(sgpr_model *h, const double *mean_w, const double *vscale)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_set_mean: bad arguments");
    if (h->m <= 0 || !h->has_mu) return fail(SGPR_E_NOMODEL, "sgpr_set_mean: no solved weights on the device");
    HIPCHK(hipSetDevice(h->device));
    for (int k = 0; k < h->S; k++) {
        h->mean_w[k] = mean_w ? mean_w[k] : 0.0;
        if (vscale) h->vscale[k] = vscale[k];
    }
    upload_vscale(h);
    h->mean_energy = 0.0;
    if (h->rank == 0)
        for (int z : h->numbers) {
            const int s = slot_of(h, z);
            if (s >= 0) h->mean_energy += h->mean_w[s];
        }
    drop_graph(h);
    return SGPR_OK;
}

// choli = L^-1 of the cached K_mm factor, [m][m] in the caller's order
extern "C" int sgpr_get_choli(sgpr_model *h, double *choli)
{
    if (!h || !choli) return fail(SGPR_E_INVALID, "sgpr_get_choli: bad arguments");
    if (h->m <= 0 || !h->has_choli) return fail(SGPR_E_NOMODEL, "sgpr_get_choli: no factor on the device");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int m = h->m, ld = h->m_pad;
    std::vector<double> cs((size_t)h->m_rows * ld);
    HIPCHK(hipMemcpy(cs.data(), h->d_choli.p, sizeof(double) * cs.size(), hipMemcpyDeviceToHost));
    for (int a = 0; a < m; a++)
        for (int b = 0; b < m; b++) choli[(size_t)h->ind_perm[a] * m + h->ind_perm[b]] = cs[(size_t)a * ld + b];
    return SGPR_OK;
}

// ---------------------------------------------------------------------------- system binding
static int alloc_work(sgpr_model *h)
{
    if (h->N <= 0) return 0;
    const int cr = h->cnt_rows;
    int bad = 0;
    bad |= h->d_Pn.alloc((size_t)cr * h->Dpad);
    bad |= h->d_norm.alloc(cr);
    bad |= h->d_C.alloc((size_t)std::max(h->cnt, 1) * h->CS);
    bad |= h->d_shear.alloc(cr);
    bad |= h->d_W.alloc((size_t)cr * h->Dpad);
    h->csq_slots = 2 * ((std::max(h->m_pad, 1) + 63) / 64);  // covloss partials: one per (64-column tile, wave column)
    bad |= h->d_csq.alloc((size_t)cr * h->csq_slots);
    bad |= h->d_F.alloc((size_t)6 * h->N);
    h->virpart_len = (h->cnt + SGPR_REV_WPW(h->S) - 1) / SGPR_REV_WPW(h->S);  // one partial per workgroup of the reverse kernel
    bad |= h->d_virpart.alloc((size_t)std::max(h->virpart_len, 1) * 9);
    bad |= h->d_packed.alloc((size_t)4 * h->N + 11);
    if (h->m > 0) {
        bad |= h->d_K.alloc((size_t)cr * h->m_pad);   // zero-filled: off-species entries are never written
        bad |= h->d_Aw.alloc((size_t)cr * h->m_pad);
        decide_tile_heights(h);
        if (build_tiles(h, 0) || build_tiles(h, 1) || build_tiles(h, 2)) bad = 1;
        h->epart_len = 8 * (int)h->t_knm.n;  // one partial per wave of every K_nm tile (four- or eight-wave form: the
                                             // unused half of a four-wave tile's eight slots stays zero)
        bad |= h->d_Epart.alloc(std::max(h->epart_len, 1));
    } else
        h->epart_len = 0;
    return bad ? fail(SGPR_E_NODEVICE, "hipMalloc failed (work arrays)") : 0;
}

static int ensure_nl(sgpr_model *h, int maxnn)
{
    if (maxnn <= h->maxnn && h->d_nbr_j.p) return 0;
    h->maxnn = maxnn;
    drop_graph(h);
    int bad = 0;
    bad |= h->d_nbr_j.alloc((size_t)h->N * maxnn, false);
    bad |= h->d_nbr_shift.alloc((size_t)h->N * maxnn, false);
    bad |= h->d_aux.alloc((size_t)h->N * maxnn, false);
    bad |= h->d_prec.alloc((size_t)h->N * maxnn * 4, false);
    bad |= h->d_G.alloc((size_t)h->N * maxnn * 4, false);
    bad |= h->d_cand_j.alloc((size_t)h->N * maxnn, false);
    bad |= h->d_cand_code.alloc((size_t)h->N * maxnn, false);
    bad |= h->d_cidx.alloc((size_t)h->N * maxnn, false);
    h->hmw = (maxnn + 63) / 64;
    bad |= h->d_hm.alloc((size_t)h->N * h->hmw);
    h->lists_valid = false;
    return bad ? fail(SGPR_E_NODEVICE, "hipMalloc failed (neighbour list, maxnn=%d)", maxnn) : 0;
}

// reverse-index table: N rows of `stride` 2-byte entries (stride = neighbouring bins x bin capacity)
static int ensure_rev(sgpr_model *h, int stride)
{
    if (stride <= h->t_stride && h->d_T.p) return 0;
    h->t_stride = stride;
    h->lists_valid = false;
    drop_graph(h);
    if (h->d_T.alloc((size_t)std::max(h->N, 1) * stride, false))
        return fail(SGPR_E_NODEVICE, "hipMalloc failed (reverse index, %d entries per atom)", stride);
    return 0;
}

static int ensure_bins(sgpr_model *h, int cap)
{
    if (cap <= h->bin_cap && h->d_b_rec.p) return 0;
    h->bin_cap = cap;
    h->lists_valid = false;
    drop_graph(h);
    const size_t slots = (size_t)4096 * cap;
    int bad = 0;
    bad |= h->d_b_rec.alloc(slots);  // zero-filled: the sweep may read (and discard) slot 0 of an empty bin
    bad |= h->d_b_aux.alloc(slots);
    return bad ? fail(SGPR_E_NODEVICE, "hipMalloc failed (bins, cap=%d)", cap) : 0;
}

extern "C" int sgpr_bind_system(sgpr_model *h, int N, const int32_t *numbers, const int32_t *pbc, int rank, int world)
{
    if (!h || N < 0 || (N > 0 && !numbers) || world < 1 || rank < 0 || rank >= world)
        return fail(SGPR_E_INVALID, "sgpr_bind_system: bad arguments");
    if (N >= (1 << 24)) return fail(SGPR_E_UNSUPPORTED, "sgpr_bind_system: %d atoms (the neighbour keys hold 24-bit indices)", N);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    drop_graph(h);
    std::vector<int> slot(N);
    for (int i = 0; i < N; i++) {
        slot[i] = slot_of(h, numbers[i]);
        if (slot[i] < 0 && h->ignore_unknown) slot[i] = h->S;  // a ghost: sorted last, never binned, no neighbours
        if (slot[i] < 0) return fail(SGPR_E_SPECIES, "atom %d: Z=%d is not in the model's species table", i, numbers[i]);
    }
    h->N = N; h->rank = rank; h->world = world;
    h->N_rows = rup(std::max(N, 1), 64);
    for (int k = 0; k < 3; k++) h->pbc[k] = pbc ? (pbc[k] != 0) : 1;
    h->numbers.assign(numbers, numbers + N);
    h->perm.resize(N);
    std::iota(h->perm.begin(), h->perm.end(), 0);
    std::stable_sort(h->perm.begin(), h->perm.end(), [&](int a, int b) { return slot[a] < slot[b]; });
    h->slot_sorted.resize(N);
    for (int i = 0; i < N; i++) h->slot_sorted[i] = slot[h->perm[i]];
    // this rank's share: sorted atoms rank, rank+world, ... (per-species round robin, the
    // reference's Distributer policy, descriptor/atoms.py:235-246)
    h->cnt = N > rank ? (N - rank + world - 1) / world : 0;
    h->cnt_rows = rup(std::max(h->cnt, 1), 64);
    h->aoff.assign(h->S + 1, 0);
    std::vector<int> lslot(h->cnt_rows, -1), lnn(h->cnt_rows, 1);
    for (int il = 0; il < h->cnt; il++) {
        const int s = h->slot_sorted[rank + il * world];
        if (s >= h->S) { lnn[il] = 0; continue; }  // ghost row: outside every species block, slot -1
        lslot[il] = s;
        h->aoff[s + 1]++;
    }
    for (int s = 0; s < h->S; s++) h->aoff[s + 1] += h->aoff[s];
    h->mean_energy = 0.0;
    if (rank == 0)
        for (int i = 0; i < N; i++)
            if (slot[i] < h->S) h->mean_energy += h->mean_w[slot[i]];
    int bad = 0;
    bad |= h->d_perm.alloc(3 * (size_t)std::max(N, 1), false);  // sorted -> caller | caller -> sorted | species slot by caller index
    bad |= h->d_slot.alloc(h->N_rows, false);
    bad |= h->d_aoff.alloc(h->S + 1, false);
    bad |= h->d_lslot.alloc(h->cnt_rows, false);
    bad |= h->d_lnn.alloc(h->cnt_rows, false);
    bad |= h->d_pos_in.alloc((size_t)3 * std::max(N, 1) + 16);  // + the cell behind the positions (warm path: one copy)
    bad |= h->d_pos.alloc((size_t)3 * std::max(N, 1));
    bad |= h->d_bin_of.alloc(std::max(N, 1));
    bad |= h->d_kslot.alloc(std::max(N, 1));
    bad |= h->d_ncand.alloc(std::max(N, 1));
    bad |= h->d_pos0.alloc((size_t)3 * std::max(N, 1));
    h->lists_valid = false;
    bad |= h->d_nn_raw.alloc(h->cnt_rows);
    bad |= h->d_nn.alloc(std::max(N, 1));
    bad |= h->d_gpart.alloc((size_t)12 * ((std::max(N, 1) + 255) / 256));
    if (bad) return fail(SGPR_E_NODEVICE, "hipMalloc failed (system arrays)");
    if (N > 0) {
        std::vector<int> pp(3 * (size_t)N);
        for (int i = 0; i < N; i++) { pp[i] = h->perm[i]; pp[(size_t)N + h->perm[i]] = i; pp[2 * (size_t)N + i] = slot[i]; }
        HIPCHK(hipMemcpy(h->d_perm.p, pp.data(), sizeof(int) * pp.size(), hipMemcpyHostToDevice));
        std::vector<int> ss(h->N_rows, -1);
        std::copy(h->slot_sorted.begin(), h->slot_sorted.end(), ss.begin());
        HIPCHK(hipMemcpy(h->d_slot.p, ss.data(), sizeof(int) * h->N_rows, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemcpy(h->d_aoff.p, h->aoff.data(), sizeof(int) * (h->S + 1), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_lslot.p, lslot.data(), sizeof(int) * h->cnt_rows, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(h->d_lnn.p, lnn.data(), sizeof(int) * h->cnt_rows, hipMemcpyHostToDevice));
    h->maxnn = 0;
    h->d_nbr_j.release();
    h->d_nbr_shift.release();
    h->d_T.release();
    h->t_stride = 0;
    h->gather_ok = !h->force_scatter;
    h->warm = false;
    h->bind_gen++;
    return alloc_work(h);
}

extern "C" int64_t sgpr_packed_len(int N) { return 4 * (int64_t)N + 11; }

#include "peer.inc"

// ---------------------------------------------------------------------------- one step
static void stamp(sgpr_model *h, const char *name, hipStream_t st)
{
    if (!h->profile) return;
    const size_t k = h->stage_names.size();
    if (h->ev.size() <= k + 1) {
        while (h->ev.size() <= k + 1) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            h->ev.push_back(e);
        }
    }
    h->stage_names.push_back(name);
    (void)hipEventRecord(h->ev[k + 1], st);
}

// the step's last kernel.  gather: forces from the pair gradients G (reverse pass in gather form);
// otherwise from the two force buffers of the scatter form (zero when no reverse pass ran).
struct FinBatch { int batch; double *G, *F, *virpart; size_t g_stride, f_stride, v_stride, p_stride; const int *cols; };

// what the last kernel of a step does for the NEXT one (FinNext): the host side
struct StepNext {
    int mode = 0;                      // 1: the next frame's positions are on the device, 2: integrate (sgpr_md_run)
    const double *pos_next = nullptr;  // mode 1: caller order
    FinNext md;                        // mode 2: the integrator's fields (the binning fields are filled in launch_finalize)
};

static void launch_finalize(sgpr_model *h, bool gather, int nE, int nV, bool beta, double mean_energy,
                            double *packed_dev, hipStream_t st, const FinBatch *fb = nullptr, const StepNext *nx = nullptr,
                            unsigned step = 0, bool xp = false, const ShardSrc *consume = nullptr)
{
    const int N = h->N;
    FinArgs f = {};
    f.xp = xp ? 1 : 0; f.xcmax = (int)peer_cmax(N, h->world);
    f.scal_off = xp ? 3 * (size_t)N + 4 * (size_t)f.xcmax : 4 * (size_t)N;
    f.N = N; f.cnt = h->cnt; f.first = h->rank; f.stride = h->world; f.maxnn = h->maxnn; f.t_stride = h->t_stride;
    f.has_beta = beta ? 1 : 0; f.nE = nE; f.nV = nV; f.bin_cap = h->bin_cap; f.nbins_clear = 4096;
    f.t_check = (h->world == 1 && h->gather_ok) ? 1 : 0;
    f.perm = h->d_perm.p; f.slot = h->d_slot.p; f.nn = h->d_nn.p; f.nbr_j = h->d_nbr_j.p; f.aux = h->d_aux.p;
    f.nn_raw = h->d_nn_raw.p; f.T = h->d_T.p; f.G = h->d_G.p; f.Fnbr = h->d_F.p; f.Fself = h->d_F.p + 3 * (size_t)N;
    f.csq = h->d_csq.p; f.csq_slots = h->csq_slots; f.vs_sqrt = h->d_vs_sqrt.p; f.Epart = h->d_Epart.p; f.virpart = h->d_virpart.p;
    f.mean_energy = mean_energy; f.packed = packed_dev; f.stat = h->d_stat.p; f.bin_count = h->d_bin_count.p;
    f.flag = h->step_flag ? h->step_flag : h->d_flag.p; f.pos = h->d_pos.p; f.pos0 = h->d_pos0.p;
    f.rebuilds = h->d_flag.p + 4;
    f.cell = h->last_cell; f.grid = (const NlGrid *)h->d_grid.p; f.cell0 = h->d_cell0.p;
    int batch = 1;
    if (fb) step = h->step_count - 1;  // (the step whose lists the batch re-uses)
    if (fb) {
        // column batches of the training rows re-use the lists of the step that ran just before them: that step's
        // finalize has done the rebuild bookkeeping (pos0, cell0, the rebuild counter) — not again per batch
        f.flag = h->d_flag.p + 5;
        batch = fb->batch;
        f.G = fb->G; f.Fnbr = fb->F; f.Fself = fb->F + 3 * (size_t)N; f.virpart = fb->virpart;
        f.g_stride = fb->g_stride; f.f_stride = fb->f_stride; f.v_stride = fb->v_stride; f.p_stride = fb->p_stride;
        f.row_cols = fb->cols; f.row_colslot = h->d_ind_slot.p; f.nbr_code = h->d_nbr_shift.p;
    }
    f.bin_count = h->d_bin_count.p + SGPR_BIN_INTS * (step & 1u);
    if (nx && nx->mode) {
        FinNext &x = f.nx;
        if (nx->mode == 2) x = nx->md;
        x.mode = nx->mode; x.S = h->S; x.cap = h->bin_cap; x.force = h->skin <= 0.0 ? 1 : 0; x.step = (int)step;
        for (int k = 0; k < 3; k++) x.pbc[k] = h->pbc[k];
        x.thr2 = 0.25 * h->skin * h->skin;
        x.pos_in = nx->pos_next; x.pos = h->d_pos.p; x.flags = h->d_flag.p;
        x.iperm = h->d_perm.p + N; x.cslot = h->d_perm.p + 2 * (size_t)N;
        x.bc_cur = h->d_bin_count.p + SGPR_BIN_INTS * (step & 1u); x.bc_next = h->d_bin_count.p + SGPR_BIN_INTS * ((step + 1) & 1u);
        x.bin_of = h->d_bin_of.p; x.kslot = h->d_kslot.p; x.b_rec = h->d_b_rec.p; x.b_aux = h->d_b_aux.p;
        x.csq_rw = h->d_csq.p;
        const dim3 grid((std::max(N, 1) + 3) / 4 + 13);
        if (consume) {
            // the consumer of a sharded step's exchange (peer.inc): totals, the next positions, the next step's bins
            const dim3 gs((std::max(N, 1) + 63) / 64 + 3);
            if (nx->mode == 1) hipLaunchKernelGGL(shard_next_kernel<1>, gs, dim3(256), 0, st, f, *consume);
            else hipLaunchKernelGGL(shard_next_kernel<2>, gs, dim3(256), 0, st, f, *consume);
            return;
        }
        if (!gather) {
            x.csq_rw = h->d_csq.p;
            hipLaunchKernelGGL(finalize_scatter_next_kernel, dim3((std::max(N, 1) + 255) / 256 + 11), dim3(256), 0, st, f);
        } else if (nx->mode == 1) hipLaunchKernelGGL(finalize_next_kernel<1>, grid, dim3(256), 0, st, f);
        else hipLaunchKernelGGL(finalize_next_kernel<2>, grid, dim3(256), 0, st, f);
        return;
    }
    if (gather)
        hipLaunchKernelGGL(finalize_gather_kernel, dim3((std::max(N, 1) + 3) / 4 + 11, batch), dim3(256), 0, st, f);
    else
        hipLaunchKernelGGL(finalize_kernel, dim3((std::max(N, 1) + 255) / 256 + 11, batch), dim3(256), 0, st, f);
}

static int enqueue_step(sgpr_model *h, const double *pos_dev, const double *cell_dev, double *packed_dev,
                        hipStream_t st, const StepNext *nx = nullptr, bool no_exchange = false)
{
    const int N = h->N, cnt = h->cnt;
    h->last_cell = cell_dev;
    if (h->profile) {
        h->stage_names.clear();
        if (h->ev.empty()) {
            hipEvent_t e;
            (void)hipEventCreate(&e);
            h->ev.push_back(e);
        }
        (void)hipEventRecord(h->ev[0], st);
    }
    NlParams np = {};
    np.N = N; np.S = h->S;
    for (int k = 0; k < 3; k++) np.pbc[k] = h->pbc[k];
    // single-process frames: reverse pass in gather form (reverse index from the list build, pair
    // gradients summed by the last kernel); sharded frames scatter with atomics (remote atoms' lists
    // are not built here)
    const bool gather = h->world == 1 && h->gather_ok && h->d_T.p != nullptr;
    NlScratch sc = {(NlGrid *)h->d_grid.p, h->d_bin_of.p, h->d_kslot.p, h->d_bin_count.p, h->bin_cap, h->d_b_rec.p,
                    h->d_b_aux.p, h->d_slot.p, h->d_stat.p, h->d_nn_raw.p, h->d_aux.p, gather ? h->d_T.p : nullptr,
                    h->t_stride};
    // candidate lists: reused while nobody moved more than skin/2 (decided on the device, neighbor.hip); a
    // captured graph cannot alternate the flag parity, so it rebuilds every step
    const double skin = h->use_graph ? 0.0 : h->skin;
    const unsigned step = h->step_count++;
    sc.bin_count = h->d_bin_count.p + SGPR_BIN_INTS * (step & 1u);
    sc.flag = h->d_flag.p; sc.parity = (int)(step & 1u); sc.fslot = (int)(step & 3u); sc.fclear = (int)((step + 2) & 3u);
    sc.force = (!h->lists_valid || skin <= 0.0) ? 1 : 0;
    sc.skin = skin; sc.pos0 = h->d_pos0.p; sc.cell0 = h->d_cell0.p; sc.ncand = h->d_ncand.p; sc.cand_j = h->d_cand_j.p;
    sc.cand_code = h->d_cand_code.p; sc.cidx = h->d_cidx.p; sc.hm = h->d_hm.p; sc.hmw = h->hmw;
    h->step_flag = h->d_flag.p + sc.fslot;
    // the previous step's last kernel may have binned this step already (finalize_next_kernel): same positions, same cell,
    // consecutive step counters.  Pre-binned for something else: its bin populations are dropped.
    const bool pre = h->pre_valid && h->pre_pos == pos_dev && h->pre_cell == cell_dev && h->pre_step == step && !sc.force;
    if (h->pre_valid && !pre) (void)hipMemsetAsync(sc.bin_count, 0, SGPR_BIN_INTS * sizeof(int), st);
    h->pre_valid = false;
    if (!pre) {
        launch_neighbor_bin(np, h->bin_identity ? nullptr : h->d_perm.p, pos_dev, h->d_pos.p, cell_dev, h->rc + skin, sc, h->d_F.p, 3 * N,
                            h->d_csq.p, cnt * h->csq_slots, st);
        stamp(h, "neighbor_bin", st);
    }
    DescParams dp = {};
    dp.lmax = h->lmax; dp.nmax = h->nmax; dp.S = h->S; dp.N = cnt; dp.Nall = N; dp.first = h->rank;
    dp.stride = h->world; dp.maxnn = h->maxnn; dp.Dc = h->Dc; dp.Dpad = h->Dpad; dp.CS = h->CS; dp.rc = h->rc;
    for (int k = 0; k < SGPR_MAX_S; k++) dp.radii_v[k] = k < h->S ? h->radii[k] : 1.0;
    if (h->d_stamps.p && h->d_pstamps.n < (size_t)16 * N) h->d_pstamps.alloc((size_t)16 * N);
    dp.stamps = h->d_stamps.p ? h->d_pstamps.p : nullptr;
    dp.xq = h->xcd_quads ? h->gemm_bm_k / 4 : 0;  // the forward pass writes the rows of the K_nm tiles
    dp.stat = h->d_stat.p;
    int rcd = launch_list_forward(dp, sc, h->d_pos.p, cell_dev, h->d_pack.p, h->d_nn.p, h->d_lnn.p, h->d_nbr_j.p,
                                  h->d_nbr_shift.p, h->d_Pn.p, h->d_norm.p, h->d_C.p, h->d_shear.p, h->d_prec.p, st);
    if (rcd) return fail(SGPR_E_UNSUPPORTED, "descriptor kernel not compiled in");
    stamp(h, "list_forward", st);
    const bool predict = h->m > 0 && h->has_mu && cnt > 0 && !h->rows_mu;
    const bool beta = h->m > 0 && h->has_choli && cnt > 0 && !h->rows_mu;
    // one launch for the three products when their tile forms allow it: the K_nm tiles first, the W and covloss tiles of
    // a row panel start when that panel's K_nm tiles have signalled (gemm_tile.inc, EPI_FUSED)
    const bool fused3 = predict && beta && h->gemm_fused && h->t_fused.n > 0 && !h->use_graph &&
                        !(h->use_fork && h->side && !h->profile) &&
                        !(h->cov_in_rev && h->gemm_bm_w == 32 && h->gemm_kd_w == 16 && h->t_covl.n > 0);
    if (h->m > 0 && cnt > 0 && !fused3) {
        gemm_kernel_pm(h, h->d_Pn.p, cnt, h->d_lslot.p, h->d_lnn.p, h->t_knm, h->d_K.p, h->d_Aw.p,
                       h->rows_mu ? h->rows_mu : (h->has_mu ? h->d_mu.p : nullptr), h->d_Epart.p, st);
        stamp(h, "gemm_knm", st);
    }
    // W = Aw.Pm (reverse-pass seed) and the covloss product K.choli^T depend only on the K_nm
    // kernel and share the row dimension: they go out as ONE grouped launch (582 tiles instead of
    // 320 + 262, one kernel boundary less).  A side-stream fork of the two measured neutral.
    GemmParams gw = {}, gc = {};
    gw.M = cnt; gw.N = h->Dpad; gw.K = h->m_pad;
    gw.lda = h->m_pad; gw.ldb = h->m_pad; gw.ldc = h->Dpad;
    gw.A = h->d_Aw.p; gw.B = h->d_PmT.p; gw.C = h->d_W.p;
    gw.tiles = h->t_w.p; gw.ntiles = (int)h->t_w.n; gw.bm = h->gemm_bm_w; gw.kd = h->gemm_kd_w;
    gw.waves = gc.waves = (h->gemm_w64 && h->gemm_bm_w == 64) ? 8 : 4;
    gw.wgs = gc.wgs = h->gemm_wgs64;
    gc.M = cnt; gc.N = h->m; gc.K = h->m_pad;
    gc.lda = h->m_pad; gc.ldb = h->m_pad; gc.ldc = 0;
    gc.A = h->d_K.p; gc.B = h->d_choli.p; gc.C = nullptr;
    gc.tiles = h->t_cov.p; gc.ntiles = (int)h->t_cov.n; gc.bm = h->gemm_bm_w; gc.kd = h->gemm_kd_w;
    gc.rowsq = h->d_csq.p; gc.rowsq_ld = h->csq_slots;
    gw.stamps = (h->d_stamps2.p && h->t_wcov.n <= 8192) ? h->d_stamps2.p : nullptr;
    GemmParams gcl = gc;  // the covloss product on its own tile list (longest first)
    gcl.tiles = h->t_covl.p; gcl.ntiles = (int)h->t_covl.n;
    // "overlap" option: the covloss product (MFMA-bound) runs on a side stream next to the reverse pass
    // (VALU/latency-bound) instead of being grouped with the W product
    bool forked = false, cov_rides = false;
    if (fused3) {
        if (h->fuse_epoch >= (1 << 22)) {   // (counters reach epoch x 255 at most: far from the end of an int)
            (void)hipMemsetAsync(h->d_panel_cnt.p, 0, h->d_panel_cnt.n * sizeof(int), st);
            h->fuse_epoch = 0;
        }
        GemmParams gk = knm_params(h, h->d_Pn.p, cnt, h->d_lslot.p, h->d_lnn.p, h->t_knm, h->d_K.p, h->d_Aw.p, h->d_mu.p,
                                   h->d_Epart.p);
        launch_gemm_fused(gk, gw, gc, h->t_fused.p, (int)h->t_fused.n, h->d_Epart.p, h->d_panel_cnt.p, ++h->fuse_epoch,
                          h->d_stat.p + 3, st);
        stamp(h, "gemm_fused", st);
    } else if (predict && beta && h->use_fork && h->side && !h->profile) {
        (void)hipEventRecord(h->ev_fork, st);
        (void)hipStreamWaitEvent(h->side, h->ev_fork, 0);
        launch_gemm_nt(gc, EPI_ROWSQ, h->side);
        (void)hipEventRecord(h->ev_join, h->side);
        launch_gemm_nt(gw, EPI_STORE, st);
        forked = true;
    } else if (predict && beta && h->cov_in_rev && h->gemm_bm_w == 32 && h->gemm_kd_w == 16 && h->t_covl.n > 0) {
        // W alone; the covloss tiles ride in the reverse kernel's launch below
        launch_gemm_nt(gw, EPI_STORE, st);
        stamp(h, "gemm_w", st);
        cov_rides = true;
    } else if (predict && beta) {
        launch_gemm_wcov(gw, gc, h->t_wcov.p, (int)h->t_wcov.n, st, h->wcov_grid);
        stamp(h, "gemm_w_covloss", st);
    } else if (predict) {
        launch_gemm_nt(gw, EPI_STORE, st);
        stamp(h, "gemm_w", st);
    } else if (beta) {
        launch_gemm_nt(gc, EPI_ROWSQ, st);
        stamp(h, "gemm_covloss", st);
    }
    if (predict) {
        dp.xq = h->xcd_quads ? h->gemm_bm_w / 4 : 0;  // the reverse pass reads the rows of the W tiles
        rcd = launch_descriptor_backward(dp, h->d_pos.p, cell_dev, h->d_slot.p, h->d_radii.p, h->d_nn.p, h->d_nbr_j.p,
                                         h->d_nbr_shift.p, h->d_pack.p, h->d_Pn.p, h->d_norm.p, h->d_C.p, h->d_shear.p,
                                         h->d_W.p, h->d_prec.p, gather ? h->d_G.p : nullptr, h->d_aux.p,
                                         h->d_T.p, h->t_stride, h->d_cidx.p, h->d_hm.p, h->hmw, h->d_F.p, h->d_virpart.p,
                                         st, nullptr, cov_rides ? &gcl : nullptr);
        if (rcd) return fail(SGPR_E_UNSUPPORTED, "descriptor kernel not compiled in");
        stamp(h, cov_rides ? "descriptor_rev_covloss" : "descriptor_rev", st);
    }
    if (forked) (void)hipStreamWaitEvent(st, h->ev_join, 0);
    // the last kernel also opens the next step when the caller has said where the next positions are (same cell), the frame
    // is not sharded and nothing of the step ran on a side stream
    // (frames: also the scatter form of a sharded rank, whose all-reduce follows on the same stream; the integrator only in
    // the single-rank gather form)
    // With the library's own exchange attached (peer.inc) a sharded step leaves its partial sums in the exchange layout; and
    // when the next positions are known, the exchange and its consumer — totals, integrator or next frame, next bins — run
    // here (shard_next_kernel): the caller's reduce_packed finds nothing left to do.  A scatter-form step of a SINGLE rank
    // takes the same consumer with its own partial sums as the one slice (frames the gather form cannot serve; the twin a
    // sharded run is compared with bit for bit).
    const bool px = peer_on(h) && h->world > 1;
    const bool can_next = nx && nx->mode && h->fuse_next && predict && !forked && !h->use_graph && h->skin > 0.0;
    const bool shard_next = can_next && !gather && (px || (nx->mode == 2 && h->world == 1)) && !no_exchange;
    const bool fuse = can_next && !shard_next &&
                      ((gather && h->comm == nullptr) || (nx->mode == 1 && !gather && h->world > 1 && !px));
    const bool xp = !gather && (px || shard_next);
    const size_t xlen = peer_xlen(N, h->world);
    if (xp && h->d_xpacked.n < xlen + 1 && h->d_xpacked.alloc(xlen + 1)) return fail(SGPR_E_NODEVICE, "hipMalloc failed (exchange buffer)");
    launch_finalize(h, gather && predict, predict ? (fused3 ? 4 * (int)h->t_knm.n : h->epart_len) : 0, predict ? h->virpart_len : 0, beta, h->mean_energy,
                    xp ? h->d_xpacked.p : packed_dev, st, nullptr, fuse ? nx : nullptr, step, xp);
    h->reduce_done = false;
    if (shard_next) {
        stamp(h, "finalize", st);
        ShardSrc src;
        src.base = h->d_xpacked.p; src.n = 1; src.stride = 0; src.ctl = nullptr; src.cmax = (int)peer_cmax(N, h->world);
        if (px) {
            int parity = 0;
            if (h->peer.world != h->world || h->peer.rank != h->rank)
                return fail(SGPR_E_INVALID, "the bound sharding (rank %d of %d) differs from the exchange's (rank %d of %d)",
                            h->rank, h->world, h->peer.rank, h->peer.world);
            const int *halt = nx->mode == 2 ? nx->md.halt : nullptr;
            const int re = peer_exchange(h, h->d_xpacked.p, xlen, st, &parity, halt, (int)step);
            if (re) return re;
            src = peer_src(h->peer, parity);
            src.cmax = (int)peer_cmax(N, h->world);
            stamp(h, "exchange", st);
        }
        launch_finalize(h, false, 0, 0, beta, h->mean_energy, packed_dev, st, nullptr, nx, step, false, &src);
        h->reduce_done = true;
    }
    if (fuse || shard_next) {
        h->pre_valid = true; h->pre_pos = nx->pos_next; h->pre_cell = cell_dev; h->pre_step = step + 1;
    }
    stamp(h, shard_next ? "sum_bin_next" : fuse ? "finalize_bin_next" : "finalize", st);
    return SGPR_OK;
}

// run eagerly, synchronise, grow the neighbour capacity until nothing overflowed
static int run_checked(sgpr_model *h, const double *pos_dev, const double *cell_dev, double *packed_dev,
                       hipStream_t st)
{
    if (h->maxnn == 0) {
        const int rc_ = ensure_nl(h, 64);
        if (rc_) return rc_;
    }
    if (h->bin_cap == 0) {
        const int rc_ = ensure_bins(h, 64);
        if (rc_) return rc_;
    }
    if (h->world == 1 && h->gather_ok && h->t_stride == 0) {
        const int rc_ = ensure_rev(h, 27 * h->bin_cap);  // the usual 3x3x3 bins; grown below if the grid needs more
        if (rc_) return rc_;
    }
    for (int attempt = 0; attempt < 12; attempt++) {
        HIPCHK(hipMemsetAsync(h->d_stat.p, 0, 4 * sizeof(int), st));
        HIPCHK(hipMemsetAsync(h->d_bin_count.p, 0, 2 * SGPR_BIN_INTS * sizeof(int), st));
        h->pre_valid = false;
        const int rc_ = enqueue_step(h, pos_dev, cell_dev, packed_dev, st, nullptr, true);   // (local attempts: the ranks' ONE exchange follows)
        if (rc_) return rc_;
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipGetLastError());
        int stat[4] = {0, 0, 0, 0};
        HIPCHK(hipMemcpy(stat, h->d_stat.p, 4 * sizeof(int), hipMemcpyDeviceToHost));
        if (stat[3] == 2) return fail(SGPR_E_INVALID, "the cell vector of a periodic direction is zero");
        if (stat[3] == 4) {   // a consumer tile of the fused GEMM launch gave up waiting for its producers: never again
            h->gemm_fused = false;
            fprintf(stderr, "[sgpr] the fused GEMM launch timed out waiting for its K_nm tiles: continuing with separate launches\n");
            continue;
        }
        if (stat[3] == 3) return fail(SGPR_E_OVERFLOW, "a pair force beyond 1024 eV/A left the fixed-point range of the sharded reverse pass");
        if (stat[3])
            return fail(SGPR_E_OVERFLOW, "an atom lies more than 127 periodic images away from a neighbour (or > 32767 "
                        "cells from the origin): wrap the positions into the cell");
        if (stat[1] > h->bin_cap) {  // a bin overflowed: grow the bins, the lists of this attempt are incomplete
            int cap = h->bin_cap;
            while (cap < stat[1] + stat[1] / 4) cap *= 2;
            const int rc2 = ensure_bins(h, cap);
            if (rc2) return rc2;
            if (h->world == 1 && h->gather_ok) {  // the reverse-index rows scale with the bin capacity
                const int rc3 = ensure_rev(h, std::max(h->t_stride, 27 * cap));
                if (rc3) return rc3;
            }
            continue;
        }
        if (stat[2] > h->t_stride && h->world == 1 && h->gather_ok) {
            // the bin grid has more neighbouring bins than the reverse-index rows hold (small cells with
            // many images), or a capacity beyond its format: grow, or fall back to the scatter form
            if (stat[2] == 0x7fffffff || (double)stat[2] * std::max(h->N, 1) > 2e9) h->gather_ok = false;
            else {
                const int rc3 = ensure_rev(h, stat[2]);
                if (rc3) return rc3;
            }
            continue;
        }
        if (stat[0] <= h->maxnn) {
            h->nn_max_seen = stat[0];
            HIPCHK(hipMemset(h->d_stat.p, 0, 4 * sizeof(int)));
            h->lists_valid = true;  // candidates complete: later steps rebuild them only when atoms have moved
            return SGPR_OK;
        }
        h->lists_valid = false;
        if (stat[0] > 100000) return fail(SGPR_E_OVERFLOW, "neighbour count %d is unreasonable", stat[0]);
        const int rc2 = ensure_nl(h, rup(stat[0] + stat[0] / 8 + 4, 8));
        if (rc2) return rc2;
    }
    return fail(SGPR_E_OVERFLOW, "neighbour capacity kept overflowing");
}

// combine the ranks' partial sums: ONE all-reduce of the packed buffer on the step's stream
// (replaces the reference's four MPI collectives, calculator/active.py:562,601,602,777)
static int reduce_packed(sgpr_model *h, double *packed_dev, hipStream_t st)
{
    if (h->reduce_done) { h->reduce_done = false; return SGPR_OK; }   // (the step's own consumer kernel has: enqueue_step)
    if (peer_on(h)) {
        // the library's own exchange (peer.inc): the step's last kernel left the partial sums in the exchange layout
        if (h->world == 1) return SGPR_OK;   // a replicated evaluation (below)
        if (h->peer.world != h->world || h->peer.rank != h->rank)
            return fail(SGPR_E_INVALID, "the bound sharding (rank %d of %d) differs from the exchange's (rank %d of %d)",
                        h->rank, h->world, h->peer.rank, h->peer.world);
        int parity = 0;
        const int re = peer_exchange(h, h->d_xpacked.p, peer_xlen(h->N, h->world), st, &parity);
        if (re) return re;
        ShardSrc src = peer_src(h->peer, parity);
        src.cmax = (int)peer_cmax(h->N, h->world);
        hipLaunchKernelGGL(peer_sum_xp_kernel, dim3(std::min(256, (4 * std::max(h->N, 1) + 11 + 255) / 256)), dim3(256), 0, st,
                           src, h->N, packed_dev, (const int *)h->peer.ctl.p, (const int *)(h->d_perm.p + h->N));
        return SGPR_OK;
    }
    if (!h->comm) return SGPR_OK;  // no communicator attached: the caller combines the partial sums
    // an unsharded bind (rank 0 of 1) under a multi-rank communicator is a REPLICATED evaluation: every rank holds
    // the totals of the whole frame already (initiate_model / get_unique_lces / training rows evaluate whole frames
    // on every rank, calculator/active.py:612-676) — nothing to combine
    if (h->world == 1) return SGPR_OK;
    if (h->comm_world != h->world || h->comm_rank != h->rank)
        return fail(SGPR_E_INVALID, "the bound sharding (rank %d of %d) differs from the communicator's (rank %d of %d)",
                    h->rank, h->world, h->comm_rank, h->comm_world);
    const ncclResult_t r = g_rccl.AllReduce(packed_dev, packed_dev, (size_t)sgpr_packed_len(h->N), ncclDouble, ncclSum, h->comm, st);
    if (r != ncclSuccess) return fail(SGPR_E_NODEVICE, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
    return SGPR_OK;
}

extern "C" int sgpr_comm_unique_id(void *id_out)
{
    if (!id_out) return fail(SGPR_E_INVALID, "sgpr_comm_unique_id: bad arguments");
    if (const int rl = rccl_load()) return rl;
    ncclUniqueId id;
    const ncclResult_t r = g_rccl.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(SGPR_E_NODEVICE, "ncclGetUniqueId: %s", g_rccl.GetErrorString(r));
    static_assert(sizeof(id) == SGPR_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return SGPR_OK;
}

extern "C" int sgpr_comm_init(sgpr_model *h, const void *id_in, int rank, int world)
{
    if (!h || !id_in || world < 1 || rank < 0 || rank >= world) return fail(SGPR_E_INVALID, "sgpr_comm_init: bad arguments");
    if (const int rl = rccl_load()) return rl;
    HIPCHK(hipSetDevice(h->device));
    if (h->comm) { (void)g_rccl.CommDestroy(h->comm); h->comm = nullptr; }
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    const ncclResult_t r = g_rccl.CommInitRank(&h->comm, world, id, rank);
    if (r != ncclSuccess) { h->comm = nullptr; return fail(SGPR_E_NODEVICE, "ncclCommInitRank: %s", g_rccl.GetErrorString(r)); }
    h->comm_rank = rank; h->comm_world = world;
    drop_graph(h);
    return SGPR_OK;
}

extern "C" int sgpr_comm_destroy(sgpr_model *h)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_comm_destroy: bad arguments");
    if (h->comm) {
        HIPCHK(hipSetDevice(h->device));
        HIPCHK(hipStreamSynchronize(h->stream));
        (void)g_rccl.CommDestroy(h->comm);
        h->comm = nullptr;
        drop_graph(h);
    }
    h->comm_rank = 0; h->comm_world = 1;
    return SGPR_OK;
}

// device buffer of `count` doubles, all-reduced (SUM or MAX) in place over the communicator: what a host
// needs besides the step (barriers, the max-over-ranks of a timing) without a second communication library
extern "C" int sgpr_comm_allreduce(sgpr_model *h, double *buf_dev, int64_t count, int op_max, void *stream)
{
    if (!h || !buf_dev || count < 0) return fail(SGPR_E_INVALID, "sgpr_comm_allreduce: bad arguments");
    if (!h->comm && !peer_on(h)) return fail(SGPR_E_INVALID, "sgpr_comm_allreduce: no communicator (sgpr_comm_init / sgpr_peer_attach)");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (peer_on(h)) {
        if (count == 0) return SGPR_OK;
        int parity = 0;
        const int re = peer_exchange(h, buf_dev, (size_t)count, st, &parity);
        if (re) return re;
        hipLaunchKernelGGL(peer_sum_kernel, dim3((unsigned)std::min<int64_t>(256, (count + 255) / 256)), dim3(256), 0, st,
                           peer_src(h->peer, parity), (size_t)count, op_max, buf_dev, (const int *)h->peer.ctl.p);
        return SGPR_OK;
    }
    const ncclResult_t r = g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, op_max ? ncclMax : ncclSum, h->comm, st);
    if (r != ncclSuccess) return fail(SGPR_E_NODEVICE, "ncclAllReduce: %s", g_rccl.GetErrorString(r));
    return SGPR_OK;
}

extern "C" int sgpr_stress_from_virial(const double *v, const double *cell, double *stress6)
{
    if (!v || !cell || !stress6) return fail(SGPR_E_INVALID, "sgpr_stress_from_virial: bad arguments");
    const double *c = cell;
    double vol = fabs(c[0] * (c[4] * c[8] - c[5] * c[7]) - c[1] * (c[3] * c[8] - c[5] * c[6]) +
                      c[2] * (c[3] * c[7] - c[4] * c[6]));
    if (!(vol > 0.0)) vol = -2.0;  // calculator/active.py:606-609
    const int voigt[6] = {0, 4, 8, 5, 2, 1};  // calculator/active.py:574
    for (int k = 0; k < 6; k++) stress6[k] = v[voigt[k]] / vol;
    return SGPR_OK;
}

// A rank-local failure inside a sharded call must not leave the peers blocked in the step's all-reduce: this rank still
// issues its ONE collective of the call, with a poison value in the overflow word, so that every rank fails the call
// alike; and it forgets its warm state, like its peers will (the next call takes the checked path on every rank: the
// same number of collectives everywhere).
static void poison_peers(sgpr_model *h, int N)
{
    h->warm = false;
    h->lists_valid = false;
    if (!((h->comm || peer_on(h)) && h->world > 1)) return;
    char keep[sizeof(g_err)];
    memcpy(keep, g_err, sizeof(keep));
    static const double poison = SGPR_PEER_POISON;
    const size_t n_out = (size_t)4 * N + 11;
    h->reduce_done = false;
    // (the exchange layout has its overflow word at 7N + 10; the all-reduce combines the packed layout itself)
    double *buf = peer_on(h) ? h->d_xpacked.p : h->d_packed.p;
    const size_t n_buf = peer_on(h) ? peer_xlen(N, h->world) : n_out;
    if (peer_on(h) && h->d_xpacked.n < n_buf + 1 && h->d_xpacked.alloc(n_buf + 1)) { memcpy(g_err, keep, sizeof(keep)); return; }
    buf = peer_on(h) ? h->d_xpacked.p : h->d_packed.p;
    if (hipMemsetAsync(buf, 0, sizeof(double) * n_buf, h->stream) == hipSuccess &&
        hipMemcpyAsync(buf + n_buf - 1, &poison, sizeof(double), hipMemcpyHostToDevice, h->stream) == hipSuccess &&
        reduce_packed(h, h->d_packed.p, h->stream) == SGPR_OK)
        (void)hipStreamSynchronize(h->stream);
    memcpy(g_err, keep, sizeof(keep));
}

// One evaluation with host arrays in, results left in page-locked host memory: *po_out points at [F 3N | beta N | E | virial 9 |
// overflow | stress 6] of THIS call (caller atom order).  Two output buffers alternate, so the results of the previous call stay
// intact while this one runs.  The body of sgpr_compute (which copies them out) and of sgpr_compute_view (which hands them out).
static int compute_core(sgpr_model *h, int N, const int32_t *numbers, const double *positions, const double *cell, const int32_t *pbc,
                        int rank, int world, bool want_cov, double **po_out)
{
    HIPCHK(hipSetDevice(h->device));
    bool same = (N == h->N && rank == h->rank && world == h->world && (int)h->numbers.size() == N);
    if (same && memcmp(h->numbers.data(), numbers, sizeof(int32_t) * (size_t)N) != 0) same = false;
    if (same && pbc)
        for (int k = 0; k < 3; k++) same = same && (h->pbc[k] == (pbc[k] != 0));
    if (!same) {
        const int rc_ = sgpr_bind_system(h, N, numbers, pbc, rank, world);
        if (rc_) return rc_;
    }
    // warm path (capacities already sized for this system by a checked pass): one page-locked staging buffer each
    // way, ONE synchronisation; the step's own overflow word (packed[4N+10], finalize) says whether the capacities
    // held — if not, or on the first call, the checked path below re-sizes and repeats
    const size_t n_in = (size_t)3 * N + 9, n_out = (size_t)4 * N + 11 + 6;
    if (h->pin_doubles < n_in + 2 * n_out) {
        // (the stream is idle here: every call ends synchronised.  The outgoing buffer is kept for one more generation: a view
        // handed out by the previous call stays valid "until the call after next" also across a growth)
        if (h->pin_old) (void)hipHostFree(h->pin_old);
        h->pin_old = h->pin;
        h->pin = nullptr; h->pin_doubles = 0;
        if (hipHostMalloc((void **)&h->pin, sizeof(double) * (n_in + 2 * n_out + 64), hipHostMallocMapped) == hipSuccess) {
            h->pin_doubles = n_in + 2 * n_out + 64;
            h->pin_dev = nullptr;
            if (hipHostGetDevicePointer((void **)&h->pin_dev, h->pin, 0) != hipSuccess) h->pin_dev = nullptr;
        }
    }
    if (!h->pin) return fail(SGPR_E_NODEVICE, "sgpr_compute: no page-locked host memory");
    h->pin_flip ^= 1;
    const size_t off_out = n_in + (size_t)h->pin_flip * n_out;
    double *pi = h->pin, *po = h->pin + off_out;
    *po_out = po;
    auto finish = [&]() {   // stress behind the packed results (calculator/active.py:604-610)
        sgpr_stress_from_virial(po + 4 * (size_t)N + 1, cell, po + 4 * (size_t)N + 11);
    };
    if (h->warm && !want_cov) {
        static const bool tl = getenv("SGPR_COMPUTE_TIMELINE") != nullptr;  // diagnostic: host-side timeline of the warm path
        auto nowus = []() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec * 1e6 + t.tv_nsec * 1e-3; };
        double tls[6] = {0, 0, 0, 0, 0, 0};
        if (tl) tls[0] = nowus();
        memcpy(pi, positions, sizeof(double) * 3 * N);
        memcpy(pi + 3 * (size_t)N, cell, sizeof(double) * 9);
        // positions and cell travel as ONE copy (a second 72-byte copy is a whole DMA command of its own) — or, by default,
        // as none: the binning kernel reads them from the page-locked buffer itself (mapped into the device's address space;
        // SGPR_ZERO_COPY_IN=0: the copy command), which takes a DMA command and its enqueue out of every call
        if (tl) tls[1] = nowus();
        static const bool zin = !(getenv("SGPR_ZERO_COPY_IN") && atoi(getenv("SGPR_ZERO_COPY_IN")) == 0);
        const bool in_direct = zin && h->pin_dev;
        if (!in_direct) HIPCHK(hipMemcpyAsync(h->d_pos_in.p, pi, sizeof(double) * n_in, hipMemcpyHostToDevice, h->stream));
        if (tl) tls[2] = nowus();
        // single rank: the last kernel writes the packed results straight into the page-locked buffer (host memory
        // mapped into the device's address space: posted PCIe writes inside the kernel) — no device-to-host copy
        // command behind the step, one synchronisation point less on the way out (option "zero_copy_out")
        const bool direct = h->zero_copy_out && !h->comm && !peer_on(h) && h->world == 1 && h->pin_dev;
        const double *pos_src = in_direct ? h->pin_dev : h->d_pos_in.p;
        int rf = enqueue_step(h, pos_src, pos_src + 3 * (size_t)N, direct ? h->pin_dev + off_out : h->d_packed.p, h->stream);
        if (rf) { poison_peers(h, N); return rf; }   // (a local enqueue failure: the peers are about to enter the all-reduce)
        rf = reduce_packed(h, h->d_packed.p, h->stream);
        if (rf) { h->warm = false; h->lists_valid = false; return rf; }
        if (!direct) HIPCHK(hipMemcpyAsync(po, h->d_packed.p, sizeof(double) * ((size_t)4 * N + 11), hipMemcpyDeviceToHost, h->stream));
        if (tl) tls[3] = nowus();
        if (h->spin_wait) {  // option "spin_wait": poll the stream instead of a blocking wait
            hipError_t q;
            while ((q = hipStreamQuery(h->stream)) == hipErrorNotReady) {}
            HIPCHK(q);
        } else
            HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipGetLastError());
        if (tl) tls[4] = nowus();
        if (po[4 * (size_t)N + 10] == 0.0) {
            finish();
            if (tl) {
                tls[5] = nowus();
                static double acc[5] = {0, 0, 0, 0, 0};
                static int cnt = 0;
                for (int k = 0; k < 5; k++) acc[k] += tls[k + 1] - tls[k];
                if (++cnt % 200 == 0) {
                    fprintf(stderr, "[sgpr timeline] warm sgpr_compute, mean of 200 (us): copy in %.1f | H2D enqueue %.1f | kernels enqueue %.1f | wait %.1f | stress %.1f\n",
                            acc[0] / 200, acc[1] / 200, acc[2] / 200, acc[3] / 200, acc[4] / 200);
                    for (double &v : acc) v = 0.0;
                }
            }
            return SGPR_OK;
        }
        h->warm = false;  // a capacity overflowed (or the cell is degenerate): the checked path sorts it out
        h->lists_valid = false;
        // ... unless the word is the poison value: a wait of this rank's exchange gave up (the consumer kernel has seen the dead
        // word: peer.inc) or another rank failed the step outright — then the call fails here, like its peers', with the ONE
        // exchange every rank has issued for it
        if (po[4 * (size_t)N + 10] >= 0.5 * SGPR_PEER_POISON) {
            if (const int pc = peer_check(h)) return pc;
            return fail(SGPR_E_OVERFLOW, "sgpr_compute: another rank of the communicator failed this step (its own error "
                        "message says why); the call fails on every rank");
        }
    }
    HIPCHK(hipMemcpyAsync(h->d_pos_in.p, positions, sizeof(double) * 3 * N, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(h->d_cell_in.p, cell, sizeof(double) * 9, hipMemcpyHostToDevice, h->stream));
    const int rc_ = run_checked(h, h->d_pos_in.p, h->d_cell_in.p, h->d_packed.p, h->stream);
    if (rc_) { poison_peers(h, N); return rc_; }
    h->warm = true;
    {   // (after the capacity-checked local pass: every rank issues exactly one collective per call)
        const int rr = reduce_packed(h, h->d_packed.p, h->stream);
        if (rr) return rr;
        HIPCHK(hipStreamSynchronize(h->stream));
        if (const int pc = peer_check(h)) return pc;
    }
    HIPCHK(hipMemcpy(po, h->d_packed.p, sizeof(double) * ((size_t)4 * N + 11), hipMemcpyDeviceToHost));
    if (po[4 * (size_t)N + 10] >= 0.5 * SGPR_PEER_POISON) {
        h->warm = false;
        h->lists_valid = false;
        return fail(SGPR_E_OVERFLOW, "sgpr_compute: another rank of the communicator failed this step (its own error "
                    "message says why); the call fails on every rank");
    }
    finish();
    return SGPR_OK;
}

extern "C" int sgpr_compute(sgpr_model *h, int N, const int32_t *numbers, const double *positions,
                            const double *cell, const int32_t *pbc, int rank, int world, double *energy,
                            double *forces, double *stress, double *beta, double *cov)
{
    if (!h || N < 0 || !numbers || !positions || !cell) return fail(SGPR_E_INVALID, "sgpr_compute: bad arguments");
    if (N == 0) {
        HIPCHK(hipSetDevice(h->device));
        const int rc_ = sgpr_bind_system(h, 0, numbers, pbc, rank, world);
        if (rc_) return rc_;
        if (energy) *energy = 0.0;
        if (stress) memset(stress, 0, sizeof(double) * 6);
        return SGPR_OK;
    }
    double *po = nullptr;
    const int rc_ = compute_core(h, N, numbers, positions, cell, pbc, rank, world, cov != nullptr, &po);
    if (rc_) return rc_;
    if (forces) memcpy(forces, po, sizeof(double) * 3 * N);
    if (beta) memcpy(beta, po + 3 * (size_t)N, sizeof(double) * N);
    if (energy) *energy = po[4 * (size_t)N];
    if (stress) memcpy(stress, po + 4 * (size_t)N + 11, sizeof(double) * 6);
    if (cov && h->m > 0) return sgpr_get_cov(h, N, h->m, cov);
    return SGPR_OK;
}

// sgpr_compute without the copies out: *packed_out points at this call's [F 3N | beta N | E | virial 9 | overflow | stress 6] in
// page-locked host memory owned by the handle (caller atom order), valid until the call AFTER THE NEXT on this handle.
extern "C" int sgpr_compute_view(sgpr_model *h, int N, const int32_t *numbers, const double *positions, const double *cell,
                                 const int32_t *pbc, int rank, int world, const double **packed_out)
{
    if (!h || N <= 0 || !numbers || !positions || !cell || !packed_out) return fail(SGPR_E_INVALID, "sgpr_compute_view: bad arguments");
    double *po = nullptr;
    const int rc_ = compute_core(h, N, numbers, positions, cell, pbc, rank, world, false, &po);
    *packed_out = rc_ ? nullptr : po;
    return rc_;
}

extern "C" int sgpr_get_cov(sgpr_model *h, int N_expected, int m_expected, double *cov)
{
    if (!h || !cov) return fail(SGPR_E_INVALID, "sgpr_get_cov: bad arguments");
    if (h->N <= 0 || h->m <= 0) return fail(SGPR_E_NOMODEL, "sgpr_get_cov: no evaluated frame / inducing set");
    if (N_expected != h->N || m_expected != h->m)
        return fail(SGPR_E_INVALID, "sgpr_get_cov: the caller expects a %d x %d matrix, the last evaluated frame has %d x %d",
                    N_expected, m_expected, h->N, h->m);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int N = h->N;
    memset(cov, 0, sizeof(double) * (size_t)N * h->m);
    std::vector<double> kb((size_t)h->cnt_rows * h->m_pad);
    HIPCHK(hipMemcpy(kb.data(), h->d_K.p, sizeof(double) * kb.size(), hipMemcpyDeviceToHost));
    for (int il = 0; il < h->cnt; il++) {
        const int c = h->perm[h->rank + il * h->world];
        for (int q = 0; q < h->m; q++) cov[(size_t)c * h->m + h->ind_perm[q]] = kb[(size_t)il * h->m_pad + q];
    }
    return SGPR_OK;
}

extern "C" int sgpr_step_dev(sgpr_model *h, const double *positions_dev, const double *cell_dev, double *packed_dev,
                             void *stream)
{
    if (!h || !positions_dev || !cell_dev || !packed_dev) return fail(SGPR_E_INVALID, "sgpr_step_dev: bad arguments");
    if (h->N <= 0) return fail(SGPR_E_INVALID, "sgpr_step_dev: call sgpr_bind_system first");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    if (!h->warm) {
        // first step after (re)binding: eager, synchronised, sizes the neighbour capacity
        const int rc_ = run_checked(h, positions_dev, cell_dev, packed_dev, st);
        if (rc_) return rc_;
        h->warm = true;
        return reduce_packed(h, packed_dev, st);
    }
    if (!h->use_graph || h->profile) {
        int rc_ = enqueue_step(h, positions_dev, cell_dev, packed_dev, st);
        if (!rc_) rc_ = reduce_packed(h, packed_dev, st);
        if (!rc_ && h->comm && h->world > 1) stamp(h, "allreduce", st);  // (profiling: the collective as its own stage)
        return rc_;
    }
    if (!h->gexec || h->g_pos != positions_dev || h->g_cell != cell_dev || h->g_out != packed_dev || h->g_stream != st) {
        drop_graph(h);
        hipGraph_t graph = nullptr;
        HIPCHK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
        int rc_ = enqueue_step(h, positions_dev, cell_dev, packed_dev, st);
        if (!rc_) rc_ = reduce_packed(h, packed_dev, st);
        const hipError_t e = hipStreamEndCapture(st, &graph);
        if (rc_) { if (graph) (void)hipGraphDestroy(graph); return rc_; }
        if (e != hipSuccess) return fail(SGPR_E_NODEVICE, "hipStreamEndCapture: %s", hipGetErrorString(e));
        const hipError_t e2 = hipGraphInstantiate(&h->gexec, graph, nullptr, nullptr, 0);
        (void)hipGraphDestroy(graph);
        if (e2 != hipSuccess) { h->gexec = nullptr; return fail(SGPR_E_NODEVICE, "hipGraphInstantiate: %s", hipGetErrorString(e2)); }
        h->g_pos = positions_dev; h->g_cell = cell_dev; h->g_out = packed_dev; h->g_stream = st;
    }
    HIPCHK(hipGraphLaunch(h->gexec, st));
    return SGPR_OK;
}

// sgpr_step_dev with the positions of the NEXT step named: the step's last kernel bins them (one launch and one cold
// start less per step: DESIGN.md §3).  The next call must pass exactly `positions_next_dev` and the same cell pointer to
// profit; any other call is served as usual.
extern "C" int sgpr_step_dev_next(sgpr_model *h, const double *positions_dev, const double *cell_dev, double *packed_dev,
                                  const double *positions_next_dev, void *stream)
{
    if (!h || !positions_dev || !cell_dev || !packed_dev) return fail(SGPR_E_INVALID, "sgpr_step_dev_next: bad arguments");
    if (h->N <= 0) return fail(SGPR_E_INVALID, "sgpr_step_dev_next: call sgpr_bind_system first");
    if (!h->warm || h->use_graph || !positions_next_dev) return sgpr_step_dev(h, positions_dev, cell_dev, packed_dev, stream);
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    StepNext nx;
    nx.mode = 1; nx.pos_next = positions_next_dev;
    int rc_ = enqueue_step(h, positions_dev, cell_dev, packed_dev, st, &nx);
    if (!rc_) rc_ = reduce_packed(h, packed_dev, st);
    return rc_;
}

// ---------------------------------------------------------------------------- device-resident molecular dynamics
// The reference integrates in ASE (cl/md.py:117-128: ase.md.langevin.Langevin around ActiveCalculator; velocities
// from util/aseutil.py:11-20) and crosses into the calculator once per step.  Here the state (positions, velocities)
// lives in HBM, the integrator is part of the step's last kernel (finalize_next_kernel<2>), and the host reads a few
// scalars per step; the covloss gate of calculate() (calculator/active.py:492-499) halts the run ON THE DEVICE at the
// step whose largest covloss reaches `ediff`, with that step's state and results intact for the model update.
static int md_alloc(sgpr_model *h, int N)
{
    MdState &m = h->md;
    bool bad = false;
    bad |= m.X.alloc((size_t)12 * N); bad |= m.V.alloc((size_t)12 * N); bad |= m.P.alloc(4 * (size_t)sgpr_packed_len(N));
    bad |= m.KE.alloc((size_t)8 * N); bad |= m.mass.alloc(N); bad |= m.sig.alloc(N); bad |= m.cell.alloc(9);
    bad |= m.halt.alloc(4); bad |= m.zeta.alloc(8);
    if (bad) return fail(SGPR_E_NODEVICE, "sgpr_md_begin: device allocation failed");
    if (!m.halt_host) {
        if (hipHostMalloc((void **)&m.halt_host, 64, hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&m.halt_host_dev, m.halt_host, 0) != hipSuccess)
            return fail(SGPR_E_NODEVICE, "sgpr_md_begin: no mapped host memory");
    }
    return SGPR_OK;
}

extern "C" int sgpr_md_begin(sgpr_model *h, int N, const int32_t *numbers, const double *positions, const double *cell,
                             const int32_t *pbc, const double *masses, const double *velocities, double dt,
                             double friction, double kT)
{
    if (!h || N <= 0 || !numbers || !positions || !cell || !masses)
        return fail(SGPR_E_INVALID, "sgpr_md_begin: bad arguments");
    if (!(dt > 0.0) || friction < 0.0 || kT < 0.0) return fail(SGPR_E_INVALID, "sgpr_md_begin: dt > 0, friction >= 0, kT >= 0");
    HIPCHK(hipSetDevice(h->device));
    // with the library's own exchange attached the run is sharded over its ranks (every rank integrates all atoms from the
    // summed forces: shard_next_kernel); otherwise a single process
    const int tr = peer_on(h) ? h->peer.rank : 0, tw = peer_on(h) ? h->peer.world : 1;
    bool same = (N == h->N && h->rank == tr && h->world == tw && (int)h->numbers.size() == N);
    for (int i = 0; i < N && same; i++) same = h->numbers[i] == numbers[i];
    for (int k = 0; k < 3 && same; k++) same = h->pbc[k] == (pbc ? (pbc[k] != 0) : 1);
    int rc_ = same ? SGPR_OK : sgpr_bind_system(h, N, numbers, pbc, tr, tw);
    if (rc_) return rc_;
    rc_ = md_alloc(h, N);
    if (rc_) return rc_;
    MdState &m = h->md;
    m.numbers.assign(numbers, numbers + N);
    m.perm = h->perm;
    for (int k = 0; k < 3; k++) m.pbc[k] = pbc ? (pbc[k] != 0) : 1;
    m.rank = tr; m.world = tw;
    m.N = N; m.t = 0; m.dt = dt; m.hdt = 0.5 * dt; m.c1 = exp(-friction * dt);
    m.ring = 3; m.nh = false; m.evaluated = false;
    const double c2 = sqrt(1.0 - m.c1 * m.c1);
    std::vector<double> xs((size_t)3 * N), vs((size_t)3 * N, 0.0), ms(N), sg(N);
    for (int i = 0; i < N; i++) {
        const int c = h->perm[i];
        for (int k = 0; k < 3; k++) {
            xs[3 * (size_t)i + k] = positions[3 * (size_t)c + k];
            if (velocities) vs[3 * (size_t)i + k] = velocities[3 * (size_t)c + k];
        }
        ms[i] = masses[c];
        if (!(ms[i] > 0.0)) return fail(SGPR_E_INVALID, "sgpr_md_begin: mass of atom %d is not positive", c);
        sg[i] = friction > 0.0 ? c2 * sqrt(kT / ms[i]) : 0.0;   // (as workloads.langevin_nvt: c2 * np.sqrt(kT / mass))
    }
    m.mass_sorted = ms;
    HIPCHK(hipMemcpy(m.X.p, xs.data(), sizeof(double) * 3 * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.V.p, vs.data(), sizeof(double) * 3 * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.mass.p, ms.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.sig.p, sg.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(m.cell.p, cell, sizeof(double) * 9, hipMemcpyHostToDevice));
    m.active = true;
    h->pre_valid = false;
    return SGPR_OK;
}

// Nose-Hoover NVT for the run begun by sgpr_md_begin (kind = 1; 0 = back to the Langevin / velocity-Verlet step of
// sgpr_md_begin's friction): what the reference's default md(dynamics="NPT", bulk_modulus=None) is — ase.md.npt.NPT with
// pfactor = None and ttime = tdamp fs (cl/md.py:17, :131-166) — restated in md_nh_advance / md_nh_kernel.  kT as given to
// sgpr_md_begin; tfact = 2 / (3 N kT ttime^2), desired kinetic energy 1.5 (N - 1) kT (ASE's constants).  Before the first
// sgpr_md_run of the run.
extern "C" int sgpr_md_thermostat(sgpr_model *h, int kind, double ttime, double kT)
{
    if (!h || (kind != 0 && kind != 1)) return fail(SGPR_E_INVALID, "sgpr_md_thermostat: kind is 0 (Langevin / velocity Verlet) or 1 (Nose-Hoover)");
    MdState &m = h->md;
    if (!m.active) return fail(SGPR_E_INVALID, "sgpr_md_thermostat: call sgpr_md_begin first");
    if (m.t != 0) return fail(SGPR_E_INVALID, "sgpr_md_thermostat: the run has started");
    if (kind == 0) { m.nh = false; m.ring = 3; return SGPR_OK; }
    if (!(ttime > 0.0) || !(kT > 0.0)) return fail(SGPR_E_INVALID, "sgpr_md_thermostat: ttime > 0 and kT > 0");
    const double tfact = 2.0 / ((double)(3 * m.N) * kT * ttime * ttime);
    m.nh = true; m.ring = 4;
    m.nh_c1 = m.dt * tfact; m.nh_c2 = 2.0 * m.dt * tfact; m.nh_K0 = 1.5 * (double)(m.N - 1) * kT;
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipMemset(m.zeta.p, 0, 8 * sizeof(double)));
    return SGPR_OK;
}

// Evaluates `nevals` configurations starting with the current one; after each evaluation but (with `final`) the last
// the integrator moves on with the next row of `noise` ([nevals][N][3] standard normal deviates, caller atom order; null:
// velocity Verlet).  Stops at the first evaluation whose largest covloss reaches `ediff` (<= 0: never): *evals_done
// counts the evaluations whose results stand, the halting one included; the state then IS that configuration (its
// forces are evaluated again by the next call — after the caller has updated the model).
// scalars: [nevals][SGPR_MD_SCAL] = E, virial(9), overflow, largest covloss, sum m v^2 (closed | before the closing half
// kick), 0, 0 per evaluation.
extern "C" int sgpr_md_run(sgpr_model *h, int nevals, const double *noise, double ediff, int final_eval, double *scalars,
                           int *evals_done, int *halt_code)
{
    if (!h || nevals <= 0 || !evals_done) return fail(SGPR_E_INVALID, "sgpr_md_run: bad arguments");
    MdState &m = h->md;
    if (!m.active) return fail(SGPR_E_INVALID, "sgpr_md_run: call sgpr_md_begin first");
    if (!(h->m > 0 && h->has_mu)) return fail(SGPR_E_NOMODEL, "sgpr_md_run: the model has no weights");
    HIPCHK(hipSetDevice(h->device));
    {   // whatever ran on the handle since sgpr_md_begin (a model update computes the training rows of stored frames and
        // trial models evaluate them: each rebinds the handle) — the run's own system is bound again before it goes on
        bool same = h->N == m.N && h->rank == m.rank && h->world == m.world && (int)h->numbers.size() == m.N;
        for (int i = 0; i < m.N && same; i++) same = h->numbers[i] == m.numbers[i];
        for (int k = 0; k < 3 && same; k++) same = h->pbc[k] == (m.pbc[k] != 0);
        if (!same) {
            const int rb = sgpr_bind_system(h, m.N, m.numbers.data(), m.pbc, m.rank, m.world);   // (warm = false: the checked pass below)
            if (rb) return rb;
        }
        if (m.world > 1 && !(peer_on(h) && h->peer.world == m.world && h->peer.rank == m.rank))
            return fail(SGPR_E_UNSUPPORTED, "sgpr_md_run: the run was begun on %d ranks, the exchange between them is gone", m.world);
    }
    hipStream_t st = h->stream;
    const int N = m.N;
    const size_t plen = (size_t)sgpr_packed_len(N);
    *evals_done = 0;
    if (halt_code) *halt_code = 0;
    // scalar ring in mapped host memory
    if (m.scal_rows < (size_t)nevals + 1) {
        if (m.mark) (void)hipHostFree(m.mark);
        if (m.scal_pin) (void)hipHostFree(m.scal_pin);
        m.mark = nullptr; m.scal_pin = nullptr; m.scal_rows = 0;
        if (m.scal_d.alloc((size_t)SGPR_MD_SCAL * ((size_t)nevals + 1), false) ||
            hipHostMalloc((void **)&m.mark, sizeof(int) * ((size_t)nevals + 1), hipHostMallocMapped) != hipSuccess ||
            hipHostGetDevicePointer((void **)&m.mark_dev, m.mark, 0) != hipSuccess ||
            hipHostMalloc((void **)&m.scal_pin, sizeof(double) * SGPR_MD_SCAL * ((size_t)nevals + 1), hipHostMallocDefault) != hipSuccess)
            return fail(SGPR_E_NODEVICE, "sgpr_md_run: no memory for the scalar ring");
        m.scal_rows = (size_t)nevals + 1;
    }
    HIPCHK(hipMemsetAsync(m.scal_d.p, 0, sizeof(double) * SGPR_MD_SCAL * ((size_t)nevals + 1), st));
    memset(m.mark, 0, sizeof(int) * ((size_t)nevals + 1));
    if (noise) {
        if (m.noise.alloc((size_t)nevals * 3 * N) || m.noise_raw.alloc((size_t)nevals * 3 * N))
            return fail(SGPR_E_NODEVICE, "sgpr_md_run: device allocation failed");
        HIPCHK(hipMemcpyAsync(m.noise_raw.p, noise, sizeof(double) * (size_t)nevals * 3 * N, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(md_sort_rows_kernel, dim3(1024), dim3(256), 0, st, N, nevals, h->d_perm.p, m.noise_raw.p, m.noise.p);
    }
    const int halt_none = 0x7fffffff;
    {
        m.halt_host[0] = halt_none; m.halt_host[1] = halt_none;  // (page-locked: the source of the copy outlives the call)
        HIPCHK(hipMemcpyAsync(m.halt.p, m.halt_host, 2 * sizeof(int), hipMemcpyHostToDevice, st));
    }
    h->bin_identity = true;
    struct Restore { sgpr_model *h; ~Restore() { h->bin_identity = false; h->pre_valid = false; } } restore{h};
    // the first evaluation sizes the capacities for this configuration if nothing has yet (synchronised, results discarded)
    const int RG = m.ring;
    const int s0 = (int)(m.t % RG);
    if (!h->warm) {
        const int rc_ = run_checked(h, m.X.p + (size_t)3 * N * s0, m.cell.p, m.P.p + plen * s0, st);
        if (rc_) return rc_;
        h->warm = true;
    }
    // the continuation of the last call (nothing else ran on the handle since, no re-binding, no option touched): its candidate
    // lists stand and its last kernel has binned this call's first configuration.  Otherwise: whatever ran in between — a model
    // update evaluates other frames — may have left other lists and bin populations (a run that halted: those of a step that
    // never ran)
    const bool chain = m.chain_ok && h->warm && h->step_count == m.chain_step && h->bind_gen == m.chain_bind && h->opt_gen == m.chain_opt &&
                       m.t == m.chain_t && m.chain_pos == m.X.p + (size_t)3 * N * s0;
    m.chain_ok = false;
    if (chain) {
        h->lists_valid = true;
        h->pre_valid = true; h->pre_pos = m.chain_pos; h->pre_cell = m.cell.p; h->pre_step = h->step_count;
    } else {
        h->lists_valid = false;
        h->pre_valid = false;
        HIPCHK(hipMemsetAsync(h->d_bin_count.p, 0, 2 * SGPR_BIN_INTS * sizeof(int), st));
    }
    const unsigned step0 = h->step_count;
    const unsigned epoch0 = h->peer.epoch;
    const bool pend0 = m.t > 0;   // (the closing half kick of the first configuration: due unless it is the start of the trajectory)
    // The host runs AHEAD of the device by at most `LA` evaluations: before evaluation j is enqueued, evaluation j - LA must
    // have set its mark (one int per evaluation in mapped host memory, written by the reducer of the overflow word: the ONLY
    // posted write of a step — the sixteen scalars stay in device memory and are copied out once per call), or the run must
    // have halted.  A halt therefore leaves at most LA + 1 evaluations in the queue (they exit at once or
    // recompute a discarded step), where a fixed chunk of 16 left up to 32 (14 ms per halt at 16384 atoms).  LA = 6 at 4096
    // atoms (the host needs ~25 us to enqueue a step of ~80 us), 2 at 16384 (~0.5 ms per step).
    const int LA = std::min(6, std::max(2, (int)lround(24576.0 / std::max(N, 1))));   // (fewer for large frames: their steps are long)
    int enq = 0;
    bool halted = false;
    int rc_ = SGPR_OK;
    for (int j = 0; j < nevals && !halted && !rc_; j++) {
        if (j >= LA) {
            const volatile int *mark = m.mark + (j - LA);
            const volatile int *hh = m.halt_host;
            unsigned spins = 0;
            while (*mark == 0 && hh[0] == halt_none && hh[1] == halt_none) {
                if ((++spins & 0x3fffu) == 0) {  // (a dead queue must not hang the host)
                    const hipError_t q = hipStreamQuery(st);
                    if (q == hipSuccess && *mark == 0) { rc_ = fail(SGPR_E_NODEVICE, "sgpr_md_run: the queue drained without evaluation %d reporting", j - LA); break; }
                    if (q != hipSuccess && q != hipErrorNotReady) { rc_ = fail(SGPR_E_NODEVICE, "sgpr_md_run: %s", hipGetErrorString(q)); break; }
                }
            }
            if (rc_) break;
            if (hh[0] != halt_none || hh[1] != halt_none) { halted = true; break; }
        }
        const int sl = (int)((m.t + j) % RG), sn = (sl + 1) % RG, sp = (sl + RG - 1) % RG;
        StepNext nx;
        const bool integrate = !(final_eval && j == nevals - 1);
        nx.mode = 2;
        nx.pos_next = m.X.p + (size_t)3 * N * sn;
        FinNext &x = nx.md;
        memset(&x, 0, sizeof(x));
        x.x_cur = m.X.p + (size_t)3 * N * sl; x.v_cur = m.V.p + (size_t)3 * N * sl;
        x.x_next = m.X.p + (size_t)3 * N * sn; x.v_next = m.V.p + (size_t)3 * N * sn;
        x.mass = m.mass.p; x.sig = m.sig.p; x.noise = noise ? m.noise.p + (size_t)j * 3 * N : nullptr;
        x.hdt = m.hdt; x.c1 = m.c1; x.pending = (j > 0 || pend0) ? 1 : 0;
        if (m.nh) {
            x.nh = 1; x.nh_first = (m.t + j) == 0 ? 1 : 0;
            x.x_prev = m.X.p + (size_t)3 * N * sp; x.v_now = m.V.p + (size_t)3 * N * sl;
            // (v_cur: what the integrator holds when it asks for the forces — ASE sets the momenta of step n after its force
            // call: v_(n-1), the caller's v_0 the first time; its kinetic energy is scalars[13], the calculator's log line)
            if (m.t + j > 0) x.v_cur = m.V.p + (size_t)3 * N * sp;
            x.nh_zeta = m.zeta.p + ((m.t + j) & 3);
        }
        x.seed = noise ? 0ull : m.seed; x.t_index = m.t + j;
        x.ke_cur = m.KE.p + (size_t)2 * N * sl; x.ke_prev = j > 0 ? m.KE.p + (size_t)2 * N * sp : nullptr;
        x.packed_prev = j > 0 ? m.P.p + plen * sp : nullptr;
        x.ediff = ediff > 0.0 ? ediff : 1e300;
        x.halt = m.halt.p; x.halt_host = m.halt_host_dev;
        x.scal_cur = m.scal_d.p + (size_t)SGPR_MD_SCAL * j; x.scal_prev = m.scal_d.p + (size_t)SGPR_MD_SCAL * (j > 0 ? j - 1 : 0);
        x.mark_cur = m.mark_dev + j;
        (void)integrate;  // (the last evaluation of a `final` run integrates speculatively too: its outcome is not adopted below)
        rc_ = enqueue_step(h, x.x_cur, m.cell.p, m.P.p + plen * sl, st, &nx);
        if (rc_) break;
        h->lists_valid = true;  // (the first evaluation rebuilt the candidates; an overflow halts the run: FinNext)
        if (!h->pre_valid) { rc_ = fail(SGPR_E_UNSUPPORTED, "sgpr_md_run: the fused last kernel is not available for this model / frame (sharded "
                                        "without the library's own exchange, graph capture or a zero skin)"); break; }
        if (m.nh)   // zeta of the next configuration from this one's kinetic energy (every integrating wave of the next launch needs it)
            hipLaunchKernelGGL(md_nh_kernel, dim3(1), dim3(256), 0, st, N, m.KE.p + (size_t)2 * N * sl, m.zeta.p, (int)((m.t + j) & 0x3fffffff),
                               m.dt, m.nh_c1, m.nh_c2, m.nh_K0, m.halt.p, (int)(step0 + j), m.scal_d.p + (size_t)SGPR_MD_SCAL * j);
        enq = j + 1;
    }
    if (rc_) { (void)hipStreamSynchronize(st); return rc_; }
    if (enq > 0) {  // the lagged reductions of the last evaluation enqueued
        FinArgs f = {};
        f.N = N;
        const int sl = (int)((m.t + enq - 1) % RG);
        f.nx.mode = 3; f.nx.step = (int)(step0 + enq);
        f.nx.ke_prev = m.KE.p + (size_t)2 * N * sl; f.nx.packed_prev = m.P.p + plen * sl;
        f.nx.ediff = ediff > 0.0 ? ediff : 1e300; f.nx.halt = m.halt.p; f.nx.halt_host = m.halt_host_dev;
        f.nx.scal_prev = m.scal_d.p + (size_t)SGPR_MD_SCAL * (enq - 1);
        hipLaunchKernelGGL(finalize_tail_kernel, dim3(2), dim3(256), 0, st, f);
    }
    // the scalars travel behind the last kernel: ONE wait for the whole call (the halt words are in mapped host memory)
    if (scalars && enq > 0)
        HIPCHK(hipMemcpyAsync(m.scal_pin, m.scal_d.p, sizeof(double) * SGPR_MD_SCAL * (size_t)enq, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    if (const int pc = peer_check(h)) return pc;
    // (covloss halts and capacity overflows each leave the evaluation in their own mapped word; the earlier one halted the run)
    int hv[4] = {std::min(m.halt_host[0], m.halt_host[1]), 0, 0, 0};
    // Where the run goes on: when every evaluation stands and the last one was integrated, the handle is left as the last kernel
    // left it — the next sgpr_md_run continues from there (chain, above).  Otherwise the last kernel enqueued has binned a step
    // that will not run — or, after a halt, the bins are those of a discarded speculative step: whoever uses the handle next
    // starts from clean bin populations.
    const bool keep_chain = hv[0] == halt_none && !final_eval && enq == nevals && h->pre_valid;
    const double *keep_pos = h->pre_pos;
    if (!keep_chain) HIPCHK(hipMemsetAsync(h->d_bin_count.p, 0, 2 * SGPR_BIN_INTS * sizeof(int), st));
    int done = enq, code = 0;
    if (hv[0] != halt_none) {
        const int k = hv[0] - (int)step0;   // evaluation (relative to this call) that halted the run
        if (k < 0 || k >= enq) return fail(SGPR_E_INVALID, "sgpr_md_run: inconsistent halt record (%d of %d)", k, enq);
        code = (m.halt_host[1] != halt_none && m.halt_host[1] == hv[0]) ? 2 : 1;
        done = k + 1;
        // exchanges that took place: the ranks have enqueued different numbers of evaluations behind the halt, all of them
        // skipped on the device (peer_push_kernel): a covloss halt at evaluation k is seen by evaluation k + 1 (or by the
        // tail kernel when k is the last), an overflow by evaluation k itself — the same count on every rank
        if (peer_on(h)) {
            h->peer.epoch = epoch0 + (unsigned)(code == 2 ? k + 1 : std::min(nevals, k + 2));
            if (getenv("SGPR_PEER_TRACE"))
                fprintf(stderr, "[sgpr peer] rank %d md_run halt: code %d k %d enq %d nevals %d -> epoch %u\n", h->peer.rank, code, k, enq, nevals, h->peer.epoch);
        }
        m.t += k;                           // the state is configuration k, not evaluated (as far as the NEXT call goes)
        if (code == 2) { h->warm = false; }  // a capacity overflowed: the next call's checked pass grows it
    } else {
        // all evaluations stand.  final: the state stays at the last configuration evaluated; else it is the next one
        const int adv = final_eval ? enq - 1 : enq;
        m.t += adv;
    }
    if (code == 2) done -= 1;  // (the overflowing evaluation's own results are void)
    if (keep_chain) {
        m.chain_ok = true; m.chain_step = h->step_count; m.chain_bind = h->bind_gen; m.chain_opt = h->opt_gen; m.chain_t = m.t;
        m.chain_pos = keep_pos;
    }
    if (scalars && done > 0) {
        memcpy(scalars, m.scal_pin, sizeof(double) * SGPR_MD_SCAL * (size_t)done);
        if (!m.nh)   // (Nose-Hoover: zeta and its time integral of the evaluation's configuration; else spare)
            for (int r = 0; r < done; r++) scalars[(size_t)SGPR_MD_SCAL * r + 14] = scalars[(size_t)SGPR_MD_SCAL * r + 15] = 0.0;
    }
    m.evaluated = code == 1 || (code == 0 && final_eval != 0);
    *evals_done = done;
    if (halt_code) *halt_code = code;
    h->lists_valid = false;
    return SGPR_OK;
}

// State of the run in caller atom order: positions of the current configuration, its velocities BEFORE the closing half
// kick (`pending` says whether one is due: v = v_pre + (dt/2) F / m once F is known), and — when the configuration has
// been evaluated by the last sgpr_md_run (a halted or `final` run) — its packed results [F | beta | E | virial | overflow].
extern "C" int sgpr_md_state(sgpr_model *h, double *positions, double *velocities_pre, int *pending, double *packed,
                             int which /*0: the current configuration; -1: the one evaluated before it*/)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_md_state: bad arguments");
    MdState &m = h->md;
    if (!m.active) return fail(SGPR_E_INVALID, "sgpr_md_state: call sgpr_md_begin first");
    if (which != 0 && which != -1) return fail(SGPR_E_INVALID, "sgpr_md_state: which = 0 or -1");
    if (which == -1 && m.t == 0) return fail(SGPR_E_INVALID, "sgpr_md_state: no earlier configuration");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int N = m.N;
    const int sl = (int)((m.t + which + m.ring) % m.ring);
    std::vector<double> buf((size_t)3 * N);
    // (the run's OWN permutation: the handle may be bound to another frame by now)
    if (positions) {
        HIPCHK(hipMemcpy(buf.data(), m.X.p + (size_t)3 * N * sl, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; i++)
            for (int k = 0; k < 3; k++) positions[3 * (size_t)m.perm[i] + k] = buf[3 * (size_t)i + k];
    }
    if (velocities_pre) {
        HIPCHK(hipMemcpy(buf.data(), m.V.p + (size_t)3 * N * sl, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; i++)
            for (int k = 0; k < 3; k++) velocities_pre[3 * (size_t)m.perm[i] + k] = buf[3 * (size_t)i + k];
    }
    if (pending) *pending = (!m.nh && (m.t + which) > 0) ? 1 : 0;   // (every configuration but the start of the trajectory)
    if (m.nh && velocities_pre && (m.t + which) > 0) {
        // Nose-Hoover: what the integrator holds when it asks for the forces of configuration n is the centred velocity of
        // configuration n - 1 (ASE sets the momenta of a step after its force call); sgpr_md_velocities has v_n itself
        const int sp = (sl + m.ring - 1) % m.ring;
        HIPCHK(hipMemcpy(buf.data(), m.V.p + (size_t)3 * N * sp, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
        for (int i = 0; i < N; i++)
            for (int k = 0; k < 3; k++) velocities_pre[3 * (size_t)m.perm[i] + k] = buf[3 * (size_t)i + k];
    }
    if (packed) HIPCHK(hipMemcpy(packed, m.P.p + (size_t)sgpr_packed_len(N) * sl, sizeof(double) * sgpr_packed_len(N), hipMemcpyDeviceToHost));
    return SGPR_OK;
}

// The velocities an observer of the trajectory sees at the current configuration, which the last sgpr_md_run must have
// evaluated (it halted there, or ran with final_eval): Langevin / velocity Verlet: the closing half kick applied; Nose-Hoover:
// the centred velocity (x_(n+1) - x_(n-1)) / 2 dt.  Caller atom order.
extern "C" int sgpr_md_velocities(sgpr_model *h, double *velocities)
{
    if (!h || !velocities) return fail(SGPR_E_INVALID, "sgpr_md_velocities: bad arguments");
    MdState &m = h->md;
    if (!m.active) return fail(SGPR_E_INVALID, "sgpr_md_velocities: call sgpr_md_begin first");
    if (!m.evaluated) return fail(SGPR_E_INVALID, "sgpr_md_velocities: the current configuration has not been evaluated (run with final_eval, or after a halt)");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int N = m.N, sl = (int)(m.t % m.ring);
    std::vector<double> v((size_t)3 * N), F;
    HIPCHK(hipMemcpy(v.data(), m.V.p + (size_t)3 * N * sl, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
    const bool kick = !m.nh && m.t > 0;
    if (kick) {
        F.resize((size_t)3 * N);   // (packed forces are in caller order)
        HIPCHK(hipMemcpy(F.data(), m.P.p + (size_t)sgpr_packed_len(N) * sl, sizeof(double) * 3 * N, hipMemcpyDeviceToHost));
    }
    for (int i = 0; i < N; i++) {
        const int c = m.perm[i];
        for (int k = 0; k < 3; k++)
            velocities[3 * (size_t)c + k] = kick ? v[3 * (size_t)i + k] + m.hdt * F[3 * (size_t)c + k] / m.mass_sorted[i] : v[3 * (size_t)i + k];
    }
    return SGPR_OK;
}

// Deviates of the integrator on the device: seed != 0 makes sgpr_md_run (called with noise = NULL) draw the standard
// normal deviate of (configuration index, atom, component) from a counter-based generator; 0 switches that off.
extern "C" int sgpr_md_seed(sgpr_model *h, uint64_t seed)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_md_seed: bad arguments");
    h->md.seed = seed;
    return SGPR_OK;
}

// The deviates sgpr_md_run uses to move configurations [t_first, t_first + count) on, out[count][N][3] in caller atom
// order (for a host-side twin of a seeded run).
extern "C" int sgpr_md_deviates(sgpr_model *h, int64_t t_first, int count, double *out)
{
    if (!h || count <= 0 || !out) return fail(SGPR_E_INVALID, "sgpr_md_deviates: bad arguments");
    MdState &m = h->md;
    if (!m.active || m.seed == 0) return fail(SGPR_E_INVALID, "sgpr_md_deviates: call sgpr_md_begin and sgpr_md_seed first");
    HIPCHK(hipSetDevice(h->device));
    ScopedBuf<double> d;
    if (d.alloc((size_t)count * 3 * m.N, false)) return fail(SGPR_E_NODEVICE, "sgpr_md_deviates: device allocation failed");
    hipLaunchKernelGGL(md_deviates_kernel, dim3(1024), dim3(256), 0, h->stream, m.N, count, m.seed, (long long)t_first, d.p);
    HIPCHK(hipMemcpy(out, d.p, sizeof(double) * (size_t)count * 3 * m.N, hipMemcpyDeviceToHost));
    return SGPR_OK;
}

extern "C" int sgpr_md_end(sgpr_model *h)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_md_end: bad arguments");
    h->md.active = false;
    return SGPR_OK;
}

extern "C" int sgpr_sync_check(sgpr_model *h, void *stream)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_sync_check: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    hipStream_t st = stream ? (hipStream_t)stream : h->stream;
    HIPCHK(hipStreamSynchronize(st));
    HIPCHK(hipGetLastError());
    int stat[4] = {0, 0, 0, 0};
    HIPCHK(hipMemcpy(stat, h->d_stat.p, 4 * sizeof(int), hipMemcpyDeviceToHost));
    HIPCHK(hipMemset(h->d_stat.p, 0, 4 * sizeof(int)));
    if (stat[3] == 2) { h->warm = false; h->lists_valid = false; return fail(SGPR_E_INVALID, "the cell vector of a periodic direction is zero"); }
    if (stat[3] == 4) {
        h->warm = false; h->gemm_fused = false;
        return fail(SGPR_E_OVERFLOW, "the fused GEMM launch timed out waiting for its K_nm tiles (separate launches from now on); "
                    "results since the last check are invalid");
    }
    if (stat[3] == 3) {
        h->warm = false;
        h->lists_valid = false;
        return fail(SGPR_E_OVERFLOW, "a pair force beyond 1024 eV/A left the fixed-point range of the sharded (scatter-form) reverse pass: "
                    "two atoms are unphysically close; results since the last check are invalid (the unsharded gather form has no such limit)");
    }
    if (stat[3]) {
        h->warm = false;
        h->lists_valid = false;
        return fail(SGPR_E_OVERFLOW, "an atom lies more than 127 periodic images away from a neighbour (or > 32767 cells "
                    "from the origin): wrap the positions into the cell; results since the last check are invalid");
    }
    if (stat[0] > h->maxnn || stat[1] > h->bin_cap || (h->world == 1 && h->gather_ok && stat[2] > h->t_stride)) {
        h->warm = false;  // next step re-sizes eagerly
        h->lists_valid = false;  // ... and rebuilds the candidates: a clamped list must not be reused as if complete
        return fail(SGPR_E_OVERFLOW, "neighbour-list capacity exceeded (neighbours %d/%d, bin %d/%d, reverse index %d/%d); "
                    "results of the steps since the last check are invalid", stat[0], h->maxnn, stat[1], h->bin_cap,
                    stat[2], h->t_stride);
    }
    h->nn_max_seen = stat[0];
    return peer_check(h);
}

extern "C" int sgpr_set_option(sgpr_model *h, const char *name, int value)
{
    if (!h || !name) return fail(SGPR_E_INVALID, "sgpr_set_option: bad arguments");
    h->opt_gen++;
    if (!strcmp(name, "graph")) {
        // the effective list cutoff switches between rc (graph: rebuild every step) and rc + skin: candidates and
        // the cached bin grid of the other mode must not be reused
        h->use_graph = value != 0; drop_graph(h); h->lists_valid = false; return SGPR_OK;
    }
    if (!strcmp(name, "overlap")) { h->use_fork = value != 0; drop_graph(h); return SGPR_OK; }
    if (!strcmp(name, "cov_in_rev")) { h->cov_in_rev = value != 0; drop_graph(h); return SGPR_OK; }
    if (!strcmp(name, "spin_wait")) { h->spin_wait = value != 0; return SGPR_OK; }
    if (!strcmp(name, "gemm_fused")) { h->gemm_fused = value != 0; return SGPR_OK; }
    if (!strcmp(name, "fuse_next")) { h->fuse_next = value != 0; h->pre_valid = false; h->lists_valid = false; return SGPR_OK; }
    if (!strcmp(name, "zero_copy_out")) { h->zero_copy_out = value != 0; return SGPR_OK; }
    if (!strcmp(name, "reverse_scatter")) {
        h->force_scatter = value != 0; h->gather_ok = !h->force_scatter; h->warm = false; h->lists_valid = false; h->pre_valid = false;
        drop_graph(h);
        return SGPR_OK;
    }
    if (!strcmp(name, "ignore_unknown_species")) { h->ignore_unknown = value != 0; return SGPR_OK; }
    if (!strcmp(name, "lone_atom_weight")) {
        if (value < 1) return fail(SGPR_E_INVALID, "sgpr_set_option: lone_atom_weight >= 1");
        if (h->m > 0 && (double)value != h->lone_w) return fail(SGPR_E_INVALID, "sgpr_set_option: lone_atom_weight is set before the inducing set");
        h->lone_w = (double)value;
        return SGPR_OK;
    }
    if (!strcmp(name, "qr_keep")) {
        if (value < 0 || value > 2) return fail(SGPR_E_INVALID, "sgpr_set_option: qr_keep is 0, 1 or 2");
        h->qr_keep_mode = value;
        for (auto &k : h->qr_keep) k.valid = false;
        return SGPR_OK;
    }
    if (!strcmp(name, "skin_milliangstrom")) {
        if (value < 0) return fail(SGPR_E_INVALID, "sgpr_set_option: negative skin");
        h->skin = 1e-3 * value;
        h->lists_valid = false;
        return SGPR_OK;
    }
    return fail(SGPR_E_INVALID, "sgpr_set_option: unknown option %s", name);
}

// ---------------------------------------------------------------------------- inspection
// one atom's LCE out of the device neighbour list: out_r[t] = x_j - x_i + shift.cell, out_slot[t]
__global__ void local_extract_kernel(int g, int maxnn, const int *nn, const int *nbr_j, const int *nbr_shift,
                                     const double *pos, const double *cell, double *out_r, int *out_slot)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= nn[g]) return;
    const size_t e = (size_t)g * maxnn + t;
    const int j = nbr_j[e], code = nbr_shift[e];
    const double s0 = (double)(int)(int8_t)(code & 0xff), s1 = (double)(int)(int8_t)((code >> 8) & 0xff),
                 s2 = (double)(int)(int8_t)((code >> 16) & 0xff);
    for (int k = 0; k < 3; k++)
        out_r[3 * t + k] = pos[3 * (size_t)j + k] - pos[3 * (size_t)g + k] + (s0 * cell[k] + s1 * cell[3 + k] + s2 * cell[6 + k]);
    out_slot[t] = (code >> 24) & 0xff;
}

extern "C" int sgpr_get_local(sgpr_model *h, int atom, int32_t *nn_out, int32_t *nbr_z, double *nbr_r, int capacity)
{
    if (!h || !nn_out) return fail(SGPR_E_INVALID, "sgpr_get_local: bad arguments");
    if (h->N <= 0 || atom < 0 || atom >= h->N) return fail(SGPR_E_INVALID, "sgpr_get_local: atom %d outside the bound system", atom);
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    int g = -1;
    for (int k = 0; k < h->N; k++)
        if (h->perm[k] == atom) { g = k; break; }
    if (g < h->rank || (g - h->rank) % h->world != 0)
        return fail(SGPR_E_INVALID, "sgpr_get_local: atom %d belongs to another rank's share", atom);
    int nn = 0;
    HIPCHK(hipMemcpy(&nn, h->d_nn.p + g, sizeof(int), hipMemcpyDeviceToHost));
    *nn_out = nn;
    if (!nbr_z || !nbr_r) return SGPR_OK;
    if (nn > capacity) return fail(SGPR_E_OVERFLOW, "sgpr_get_local: %d neighbours, room for %d", nn, capacity);
    if (nn == 0) return SGPR_OK;
    DevBuf<double> d_r;
    DevBuf<int> d_s;
    if (d_r.alloc(3 * (size_t)nn, false) || d_s.alloc(nn, false)) return fail(SGPR_E_NODEVICE, "hipMalloc failed");
    hipLaunchKernelGGL(local_extract_kernel, dim3((nn + 255) / 256), dim3(256), 0, h->stream, g, h->maxnn, h->d_nn.p,
                       h->d_nbr_j.p, h->d_nbr_shift.p, h->d_pos.p, h->last_cell ? h->last_cell : h->d_cell_in.p, d_r.p, d_s.p);
    std::vector<int> slots(nn);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(nbr_r, d_r.p, sizeof(double) * 3 * nn, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(slots.data(), d_s.p, sizeof(int) * nn, hipMemcpyDeviceToHost));
    for (int t = 0; t < nn; t++) nbr_z[t] = h->species[slots[t]];
    d_r.release(); d_s.release();
    return SGPR_OK;
}

extern "C" int sgpr_get_descriptors(sgpr_model *h, double *P)
{
    if (!h || !P) return fail(SGPR_E_INVALID, "sgpr_get_descriptors: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const size_t row = (size_t)h->S * h->S * h->D;
    DevBuf<double> d;
    if (d.alloc(row * std::max(h->cnt, 1))) return fail(SGPR_E_NODEVICE, "hipMalloc failed");
    launch_unpack_descriptors(h->cnt, h->S, h->lmax, h->nmax, h->Dc, h->Dpad, h->d_pack.p, h->d_Pn.p, d.p, h->stream);
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<double> buf(row * std::max(h->cnt, 1));
    HIPCHK(hipMemcpy(buf.data(), d.p, sizeof(double) * buf.size(), hipMemcpyDeviceToHost));
    memset(P, 0, sizeof(double) * row * h->N);
    for (int il = 0; il < h->cnt; il++)
        memcpy(P + row * h->perm[h->rank + il * h->world], buf.data() + row * il, sizeof(double) * row);
    d.release();
    return SGPR_OK;
}

extern "C" int sgpr_get_neighbors(sgpr_model *h, int64_t *ptr, int32_t *j, int32_t *off)
{
    if (!h || !ptr) return fail(SGPR_E_INVALID, "sgpr_get_neighbors: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    const int N = h->N;
    std::vector<int> nn(N), nj((size_t)N * h->maxnn), ns((size_t)N * h->maxnn);
    HIPCHK(hipMemcpy(nn.data(), h->d_nn.p, sizeof(int) * N, hipMemcpyDeviceToHost));
    std::vector<int> inv(N);
    for (int g = 0; g < N; g++) inv[h->perm[g]] = g;
    ptr[0] = 0;
    for (int c = 0; c < N; c++) {
        const int g = inv[c];
        const bool mine = g >= h->rank && (g - h->rank) % h->world == 0;
        ptr[c + 1] = ptr[c] + (mine ? nn[g] : 0);
    }
    if (!j || !off) return SGPR_OK;
    if (h->maxnn > 0) {
        HIPCHK(hipMemcpy(nj.data(), h->d_nbr_j.p, sizeof(int) * nj.size(), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(ns.data(), h->d_nbr_shift.p, sizeof(int) * ns.size(), hipMemcpyDeviceToHost));
    }
    for (int c = 0; c < N; c++) {
        const int g = inv[c];
        for (int64_t t = 0; t < ptr[c + 1] - ptr[c]; t++) {
            const size_t e = (size_t)g * h->maxnn + t;
            j[ptr[c] + t] = h->perm[nj[e]];
            const int code = ns[e];
            off[3 * (ptr[c] + t)] = (int)(int8_t)(code & 0xff);
            off[3 * (ptr[c] + t) + 1] = (int)(int8_t)((code >> 8) & 0xff);
            off[3 * (ptr[c] + t) + 2] = (int)(int8_t)((code >> 16) & 0xff);
        }
    }
    return SGPR_OK;
}

extern "C" int sgpr_get_dims(sgpr_model *h, int32_t *out)
{
    if (!h || !out) return fail(SGPR_E_INVALID, "sgpr_get_dims: bad arguments");
    out[0] = h->m; out[1] = h->S; out[2] = h->D; out[3] = h->Dc; out[4] = h->maxnn; out[5] = h->N;
    out[6] = h->nn_max_seen; out[7] = h->Dpad;
    return SGPR_OK;
}

extern "C" int sgpr_get_list_rebuilds(sgpr_model *h, int64_t *count)
{
    if (!h || !count) return fail(SGPR_E_INVALID, "sgpr_get_list_rebuilds: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    HIPCHK(hipStreamSynchronize(h->stream));
    int c = 0;
    HIPCHK(hipMemcpy(&c, h->d_flag.p + 4, sizeof(int), hipMemcpyDeviceToHost));
    *count = c;
    return SGPR_OK;
}

extern "C" int sgpr_profile(sgpr_model *h, int on)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_profile: bad arguments");
    h->profile = on != 0;
    return SGPR_OK;
}

extern "C" int sgpr_get_stage_times(sgpr_model *h, double *ms, int cap, char *names, int names_cap)
{
    if (!h) return fail(SGPR_E_INVALID, "sgpr_get_stage_times: bad arguments");
    HIPCHK(hipSetDevice(h->device));
    const int n = (int)h->stage_names.size();
    std::string all;
    for (int k = 0; k < n; k++) {
        float t = 0.f;
        if (hipEventSynchronize(h->ev[k + 1]) == hipSuccess) (void)hipEventElapsedTime(&t, h->ev[k], h->ev[k + 1]);
        if (ms && k < cap) ms[k] = t;
        all += h->stage_names[k];
        if (k + 1 < n) all += ";";
    }
    if (names && names_cap > 0) {
        strncpy(names, all.c_str(), names_cap - 1);
        names[names_cap - 1] = 0;
    }
    return n;
}

#include "solve.inc"
#include "data.inc"
