// sgpr_internal.h — shared declarations of the gfx950 SGPR evaluator (not part of the ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <string>
#include <vector>

#define SGPR_MAX_S 16       // species slots supported by the compiled kernels (nine to sixteen: lmax = nmax = 3 only)
#define SGPR_REV_WPW(S) ((S) > 8 ? 1 : 4)   // atoms (waves) per workgroup of the reverse kernel: its LDS region is 83 KB per wave at sixteen slots
#define SGPR_MAX_L 4        // lmax supported
#define SGPR_MAX_N 4        // nmax supported
#define SGPR_EPS 2.220446049250313e-16  // torch.finfo(float64).eps (descriptor/sesoap.py:250)
#define SGPR_TINY_ANGLE 1e-2            // descriptor/ylm.py:10
// Scatter form of the reverse pass (sharded frames): the force a neighbour receives is accumulated as a 64-bit FIXED-POINT
// integer, SGPR_FIX_SCALE = 2^46 units per eV/A.  Integer addition is associative: the sum no longer depends on the order
// in which the atomics land, so a sharded step repeats bit for bit (fp64 atomics did not: 1e-16-level noise that an
// ill-conditioned refit amplifies, and the hardest runs to debug are the sharded ones).  One unit is 1.4e-14 eV/A — a few
// ulp of a typical force; a single contribution is limited to 2^10 eV/A (flagged beyond: stat[3] = 3), a sum to 2^17.
#define SGPR_FIX_SCALE 70368744177664.0
#define SGPR_FIX_LIMIT 1024.0

// Coefficients of the solid-harmonic recurrences (descriptor/ylm.py:57-77), fp64, host-built.
struct HarmCoef {
    double y00;
    double al[SGPR_MAX_L + 1][SGPR_MAX_L + 1];
    double bl[SGPR_MAX_L + 1][SGPR_MAX_L + 1];
    double cl[SGPR_MAX_L + 1];
    double dl[SGPR_MAX_L + 1];
};

// Packed power-spectrum layout.  u = s*(nmax+1)+n indexes (species, radial) channels,
// U = S*(nmax+1).  p[u][v][l] is symmetric in (u,v) (descriptor/sesoap.py:195-203), so the
// device keeps only u<=v, scaled by sqrt(2) off the diagonal: dot products and norms of
// packed rows equal those of the reference's dense S*S*(nmax+1)^2*(lmax+1) rows.
// Entry e = pair(u,v)*(lmax+1)+l ; pair(u,v) = u*U - u*(u-1)/2 + (v-u).
struct PackEntry {
    int16_t u, v;   // channel indices, u <= v
    int16_t l;      // angular index
    int16_t pad;
    double coef;    // nnl[n_u][n_v][l] * (u==v ? 1 : sqrt(2))
};

struct DevModel;  // defined in api.hip

// ---- launchers (each enqueues on `st`, never synchronises) -------------------------------
struct NlParams {
    int N;            // all atoms (binned)
    int S;            // species slots of the model; atoms with slot >= S are ghosts (never binned)
    int pbc[3];
};

// bin grid of a step, built on the device from the device-resident cell (neighbor.hip)
struct NlGrid {
    double inv[9];   // inverse cell (columns = reciprocal vectors): frac = pos . inv
    int nb[3];       // bins per cell vector
    int rng[3];      // neighbouring bins searched on either side
    int nbins;
    int ortho;       // 1: the three bin-plane normals are mutually orthogonal (point-to-bin distances add in squares)
    double w[3];     // bin width along each cell vector's normal (perpendicular height / nb)
};

// Bin populations sit SGPR_BIN_STRIDE ints apart: one counter per 128-byte line.  Packed, the ~350 counters of a 4096-atom
// frame are eleven lines, and the returning atomics of the binning (one per atom, device scope: they execute at the
// memory side, one after the other per line) were most of the binning kernel's 7 us.
#define SGPR_BIN_STRIDE 32
#define SGPR_BIN_INTS (4096 * SGPR_BIN_STRIDE)

// binned copy of an atom: one 32-B record + one 8-B record per candidate of the list sweep
struct BinRec { double x, y, z; int idx; int pad; };   // position, sorted atom index
struct BinAux { short w0, w1, w2, slot; };             // wrap (floor of the fractional coordinates), species slot

struct NlScratch {
    NlGrid *grid;
    int *bin_of;       // [N]
    int *kslot;        // [N] slot of the atom in its bin
    int *bin_count;    // [4096 * SGPR_BIN_STRIDE] atoms per bin, one counter per 128-byte line; zero on entry (finalize re-zeroes it)
    int cap;           // slots per bin of the binned copies below
    BinRec *b_rec;     // [4096][cap]
    BinAux *b_aux;     // [4096][cap]
    const int *slot;   // [N] species slot by sorted index (input)
    int *stat;         // [4]: [0] max neighbour count, [1] max bin population beyond cap, [2] reverse-index
                       //      row stride needed beyond t_stride, [3] image shift / wrap beyond the packed formats
                       //      (all sticky)
    int *nn_raw;       // [count] unclamped neighbour counts (overflow check)
    // reverse index (descriptor.hip, list build): null T = not built (sharded frames use the scatter form)
    int *aux;          // [N][maxnn] bin-sweep id q*cap + k of each candidate
    unsigned short *T; // [N][t_stride] candidate position of the pair as seen from the other end
    int t_stride;
    // Verlet candidates: lists of |r| < rc + skin, rebuilt on the device when an atom moved > skin/2 since the
    // last build or the cell changed (flag[parity] of this step, set by the binning kernel; the other entry is
    // cleared for the next step); every step filters them down to |r| < rc
    int *flag;         // [4] by step counter & 3 (a fused last kernel writes the next step's flag while this step's is
                       //     still being read: finalize_next_kernel, api.hip)
    int parity;        // step counter & 1 (the two cached grid records of the binning kernel)
    int fslot, fclear; // this step's flag; the flag cleared by this step's binning (the one two steps ahead)
    int force;         // 1: rebuild regardless (new frame, capacity change, skin = 0)
    double skin;
    double *pos0;      // [N][3] positions at the last rebuild (sorted order)
    double *cell0;     // [9] cell at the last rebuild
    int *ncand, *cand_j, *cand_code, *cidx;  // [N], [N][maxnn] x 3
    unsigned long long *hm;  // [N][hmw]
    int hmw;
};

// Bins ALL N atoms (also: gathers pos_in[perm] -> pos in species-sorted order and clears the
// step's accumulators zero_a/zero_b).  The lists are built by the forward kernel (descriptor.hip).
void launch_neighbor_bin(const NlParams &p, const int *perm, const double *pos_in, double *pos, const double *cell,
                         double rc, NlScratch s, double *zero_a, int n_zero_a, double *zero_b, int n_zero_b,
                         hipStream_t st);

struct DescParams {
    int lmax, nmax, S;
    int N;            // atoms in this launch
    int Nall;         // all atoms of the frame
    int first, stride; // local atom ia <-> sorted global index first + ia*stride (rank, world)
    int maxnn;
    int Dc, Dpad;     // packed row length and padded stride
    int CS;           // c stride per atom = S*(nmax+1)*(lmax+1)^2
    double radii_v[SGPR_MAX_S];  // length unit per species slot
    double rc;
    long long *stamps;  // diagnostic build only (SGPR_STAMPS=1 + -DSGPR_PHASE_STAMPS): [2][Nall][8]
    int xq;             // quads of atoms per row tile of the GEMM that follows / precedes (XCD-aware workgroup -> atoms map), 0: off
    int *stat;          // sticky status words (NlScratch::stat), may be null
};

// Neighbour lists + forward descriptors of this rank's atoms, one launch (one wave per atom): sweep of
// the neighbouring bins, sorted list (+ reverse index) written for the reverse pass, p^, c, pair records.
int launch_list_forward(const DescParams &p, const NlScratch &nl, const double *pos, const double *cell,
                        const PackEntry *pack, int *nn /*[Nall] by sorted index*/, int *nn_local /*[N]*/,
                        int *nbr_j, int *nbr_shift, double *Pn /*[N][Dpad]*/, double *norm /*[N]*/,
                        double *C /*[N][CS]*/, int *shear /*[N]*/,
                        double *prec /*[Nall][maxnn][4] pair records (r, exp(-d^2/2))*/, hipStream_t st);

// explicit-environment form for the inducing set: CSR of neighbour vectors instead of a NL
int launch_descriptor_forward_env(const DescParams &p, const int64_t *env_ptr, const int *env_slot,
                                  const double *env_r, const double *radii, const PackEntry *pack,
                                  double *Pn, double *norm, hipStream_t st);

// Training rows: a batch of inducing columns per launch (blockIdx.y); the seed of column q is Aw[i][q] * Pm[q][:]
struct RowsBatch {
    const double *aw;   // [N][ld]
    const double *pm;   // [m][Dpad]
    const int *cols;    // [batch] species-sorted inducing index (device)
    const int *col_slot; // [m] species slot of every (sorted) inducing LCE: an atom of another species has
                        //     k(i,q) = 0 identically and sits the column out
    int ld, batch;
    size_t g_stride, f_stride, v_stride;  // doubles between batch entries of G, F, virial partials
};

// Reverse pass (one launch).  Own sums go to F[3*Nall:6*Nall].  G != null: gather form, the gradient of
// pair (i -> j) is stored to G[j][rev] (rev = T[i][aux[i][t]], the reverse index of the neighbour list)
// and the finalize kernel subtracts each atom's row.  G == null: scatter form (sharded frames), fp64
// atomics into F[0:3*Nall].
int launch_descriptor_backward(const DescParams &p, const double *pos, const double *cell,
                               const int *slot, const double *radii, const int *nn, const int *nbr_j,
                               const int *nbr_shift, const PackEntry *pack, const double *Pn,
                               const double *norm, const double *C, const int *shear,
                               const double *W /*[N][Dpad] dE/dp-hat*/, const double *prec /*from the forward pass*/,
                               double *G /*[Nall][maxnn][4] or null*/, const int *aux, const unsigned short *T,
                               int t_stride, const int *cidx, const unsigned long long *hm, int hmw,
                               double *F /*[2][Nall][3]: atomic part | own part*/,
                               double *virial /*[9][workgroups]*/, hipStream_t st, const RowsBatch *rows = nullptr,
                               const struct GemmParams *cov = nullptr /*covloss tiles (EPI_ROWSQ, 32-row, 16-deep) to run in the same launch*/);

// Training rows, sixteen columns per workgroup pass (rows16.inc): chunk c of every species block of the sorted inducing
// set per launch, `nch` chunks; pair gradients to G[ch][Nall][gnn][3][16], own sums to Fself[ch][Nall][3][16],
// virial sums to vir[ch][Nall][9][16].  Returns -6 when the instantiation is not compiled (N1 * LL > 64 or more than
// four species slots): the caller keeps the one-column-per-wave form.  Lists of at most 64 neighbours.
struct Rows16Params {
    const double *aw;   // [N][ld]  d k / d dot (K_nm pass with unit weights)
    const double *k;    // [N][ld]  k
    double eta;
    int ld;
    const double *pm;   // [m][Dpad]
    int nch;                         // chunks in this launch (<= 32)
    int gnn;                         // list slots per atom in G (>= the longest list of the frame)
    short chunk[32];                 // chunk numbers
    unsigned short cmask[32][4];     // columns asked for, per chunk and species block
    int qoff[SGPR_MAX_S + 1];
    double *G, *Fself, *vir;
    size_t g_stride, f_stride, v_stride;
};
int launch_rows16(const DescParams &p, const int *slot, const int *nn, const int *nbr_j, const int *nbr_shift,
                  const PackEntry *pack, const double *Pn, const double *norm, const double *C, const int *shear,
                  const double *prec, const int *aux, const unsigned short *T, int t_stride, const int *cidx,
                  const unsigned long long *hm, int hmw, const Rows16Params &rp, hipStream_t st);

// Unpack packed rows [n][Dpad] -> dense reference layout [n][S][S][D]
void launch_unpack_descriptors(int n, int S, int lmax, int nmax, int Dc, int Dpad, const PackEntry *pack,
                               const double *Pp, double *Pdense, hipStream_t st);

// ---- fp64 MFMA GEMM family (C = A * B^T, both operands row-major with K contiguous) --------
enum GemmEpilogue { EPI_STORE = 0, EPI_KERNEL = 1, EPI_ROWSQ = 2, EPI_SUBLOWER = 3, EPI_WCOV = 4, EPI_FUSED = 5 };

struct GemmParams {
    int M, N, K;          // C is M x N, reduction K
    int lda, ldb, ldc;
    const double *A, *B;
    double *C;
    // species structure (sorted operands): optional host-built list of working tiles
    // {row tile, col tile, kbeg, kend} (kend <= kbeg marks a padding entry); null = dense.
    const int4 *tiles;
    int ntiles;
    int bm;               // rows per tile of the table: 64 (default when 0) or 32
    int kd;               // 32-row tiles: depth of an LDS stage, 32 (default when 0) or 16 (four workgroups per CU)
    int wgs;              // 64-row eight-wave tiles: 3 = the three-register-set form, three workgroups per CU (default two)
    int waves;            // 32-row tiles with 16-deep stages: 8 = the eight-wave form (EPI_KERNEL / EPI_STORE: one 16 x 16
                          //   block per wave, two waves per SIMD per tile; energy partials: 8 per tile), else four waves
    int tri;              // unused by the kernel (the tile table carries the trimmed k range)
    // EPI_KERNEL extras
    double eta;
    double lone_m1;       // value of k(lone atom, lone atom of the same species) minus one: the reference adds its lone-atom
                          //   term (similarity/similarity.py:94-103) once PER KERNEL of the list, so the fixed-species
                          //   kernel list of calculator/active.py:31-38 gives S there, the wildcard kernel 1
    const double *mu;     // [N]
    const int *row_nn;    // [M] neighbour counts (lone-atom term), may be null => all > 0
    const int *col_nn;    // [N]
    double *Aw;           // [M][ldc]  mu_q * eta * dot^(eta-1) (masked)
    double *Esum;         // [grid blocks] per-block energy partials (plain stores), may be null
    const int *row_slot;  // [M] species slot per row (EPI_KERNEL)
    const int *col_slot;  // [N] species slot per column
    // EPI_ROWSQ extras
    double *rowsq;        // [M][rowsq_ld] partial sums, slot 2 * column tile + wave column (plain stores: a fixed
    int rowsq_ld;         //   summation order downstream)
    long long *stamps;    // diagnostic only
};
void launch_gemm_nt(const GemmParams &p, GemmEpilogue epi, hipStream_t st);
// one launch for two products sharing the row dimension: pw with EPI_STORE, pc with EPI_ROWSQ;
// tiles[].x carries the row tile in its low 16 bits and the problem (0 = pw, 1 = pc) in bit 16
void launch_gemm_wcov(const GemmParams &pw, const GemmParams &pc, const int4 *tiles, int ntiles, hipStream_t st,
                      int grid = 0 /*workgroups, when entries beyond them are chained to earlier ones (32-row four-wave form); 0: ntiles*/);
// ONE launch for the three products of a step: K_nm (pk, EPI_KERNEL epilogue) and, behind per-row-panel counters, its
// consumers W (pw) and covloss (pc).  tiles[].x: row tile | kind << 16 (0 W, 1 covloss, 2 K_nm) | K tiles of the row
// panel << 20; K_nm entries first.  panel_cnt: [row tiles] ints, zero when `epoch` starts at 1; err: one int, zero.
void launch_gemm_fused(const GemmParams &pk, const GemmParams &pw, const GemmParams &pc, const int4 *tiles, int ntiles,
                       double *Epart, int *panel_cnt, int epoch, int *err, hipStream_t st);

// ---- dense solve side ---------------------------------------------------------------------
// All on one stream, m x m row-major with leading dimension ld.
int launch_cholesky_lower(int m, double *A /*in: M+ridge I, out: L (lower, upper zeroed)*/, int ld,
                          int *info /*device: 0 ok else failing pivot+1*/, hipStream_t st,
                          double *dsave /*device scratch [2][64][64]: where diagonal blocks wait while others still read A*/);
void launch_tril_inverse(int m, const double *L, int ld, double *Li, hipStream_t st);
void launch_add_diag(int m, const double *A, int ld, double ridge, double *out, hipStream_t st);
// Blocked Householder QR least squares on the TRANSPOSED matrix At[cols + 1][ldr] (column c of
// [A | y] is the contiguous row c of At; rows padded with zeros to ldr, a multiple of 64):
// panels of 32 reflectors, compact-WY trailing updates on the MFMA GEMM.  At is overwritten.
size_t lstsq_qr_blocked_work_doubles(int rows, int cols);
// a kept panel of the factorisation (tsqr.hip): reflectors V and compact-WY T of every level of its tree
struct TsqrLevel { double *V, *T; int n, chunks, stride; };
struct TsqrPanel { int k0, nb, row_end, nlev; TsqrLevel lv[8]; };
size_t tsqr_panel_doubles(int n);                       // V + T doubles of one panel over n rows
size_t tsqr_keep_doubles(int rows, int cols, int band); // ... of every panel of a rows x cols factorisation
int launch_lstsq_qr_blocked(int rows, int cols, double *At, int ldr, double *x /*[cols] or NULL: factor only*/,
                            double *work, hipStream_t st, int band = 0, double *keep = nullptr,
                            std::vector<TsqrPanel> *panels = nullptr,
                            int extra = 0 /*columns behind the targets that are carried along: Q^T applied to them*/,
                            int band_off = 0 /*band > 0: column c is zero below row band * (c + 1) + band_off*/);
int launch_lstsq_qr_batched(int rows, int cols, double *At, int ldr, size_t bs_mat, double *x, double *work, int batch,
                            hipStream_t st, int band);
void tsqr_apply_panels(const TsqrPanel *panels, int count, double *vec, hipStream_t st);
// the banded 2m x m second stage on flat register panels (bandqr.inc) with its reflectors KEPT, and the same factorisation
// asked about new targets / one appended column (an inducing trial's refit: O(m^2))
size_t band_qr_keep_doubles(int cols);
int launch_band2_keep(int rows, int cols, double *At, int ldr, double *x, double *work, double *keepVT, hipStream_t st);
int launch_band2_append(int m_old, int nnew, double *W, int ldw, const double *keepVT, const double *At_kept, int ldr_kept, double *x,
                        double *tmp, hipStream_t st);
// one-column Householder reflectors over n elements (appended columns of a kept factorisation)
size_t flat_part_doubles(int n);
void flat_reflector_make(const double *x, int n, double *v, double *sc, double *alpha, double *part, hipStream_t st);
void flat_reflector_apply(const double *v, const double *sc, int n, double *a, double *part, hipStream_t st);
void launch_qr_gather_r(int cols, const double *At, int ldr, double *Rc, int ldc, double *z, hipStream_t st);

void host_build_harm_coef(HarmCoef *hc);
void upload_harm_coef(const HarmCoef &hc);
