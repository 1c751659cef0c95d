// tsqr.hip — Householder QR least squares for the regression of the SGPR model (regression/gppotential.py:
// 1245-1263: mu = lstsq([K; sigma L^T], [Y; 0]) through torch.linalg.qr), communication-avoiding form for gfx950.
//
// The design matrix of the first stage is tall and skinny (10^5 rows of forces against m <= 2048 inducing columns).
// A column-by-column panel factorisation of such a matrix is a chain of global reductions: two launches per column
// over all rows (the round-1/2 kernels spent 64 ms there at 98k x 1024 for 5 ms worth of flops).  Here a 32-column
// panel is factored by a TREE of independent workgroups instead (TSQR, Demmel et al. 2012):
//
//   level 0   every 256-row chunk of the panel is Householder-factored in LDS by one workgroup (no global
//             synchronisation at all): chunk = Q_i [R_i; 0]
//   level l   the R_i (32 rows each) are stacked, 8 to a chunk, and factored the same way ... until one R is left.
//
// The orthogonal factor is never formed.  Each chunk keeps its reflectors V_i and the triangular T_i of the compact
// WY form, and the trailing matrix (the other columns and the targets) is updated chunk by chunk, level by level,
//             A_i <- (I - V_i T_i^T V_i^T) A_i,
// by ONE kernel per level that holds V_i (64 KB) and a 256 x 32 tile of A (64 KB) in LDS and runs both products on
// v_mfma_f64_16x16x4_f64: the tile is read once and written once, nothing is reduced across workgroups, no atomics —
// the result is bit-reproducible.  A panel costs 2 launches per level (4 levels at 10^5 rows) instead of 64.
//
// Storage: At[c][r] = A[r][c] (column c is contiguous over the rows, leading dimension ldr), as the callers build it.
// On return the upper triangle (At[c][r], r <= c) holds R and At[cols][0:cols] holds (Q^T y)[0:cols].
//
// band > 0: column c is known to be zero below row band (c + 1) (the stacked triangular system of the second solve
// stage with its rows interleaved: band = 2); a panel then only touches the rows above band (k0 + nb).
#include <algorithm>
#include <vector>

#include "sgpr_internal.h"

#define TNB 32    // panel width
#define TCH 256   // rows per chunk (= threads per workgroup)
#define TLD 260   // LDS leading dimension of a chunk column (doubles): 2-way bank spread for the MFMA operand reads
#define TSB (TCH / TNB)  // R blocks per chunk of the upper levels

typedef double v4d __attribute__((ext_vector_type(4)));

struct TsqrLeaf {
    const double *src;  // element (c, i) of chunk b: src[b * chunk_stride + c * ld + i]
    size_t chunk_stride;
    int ld;
    int n, nb;          // logical rows of this level, panel columns
    double *V;          // [chunk][TNB][TCH]   reflectors (unnormalised: Q_j = I - scal_j v_j v_j^T), zero above the diagonal
    double *T;          // [chunk][TNB][TNB]   compact WY: Q = I - V T V^T
    double *Rnext;      // next level's chunks [chunk'][TNB][TCH] (R blocks stacked), or null at the top
    double *Rfinal;     // top: At + k0 * ldr + k0 (column-major, ldr)
    int ldr;
    // a batch of independent problems of the same shape (blockIdx.y): doubles between their matrices / work arrays
    size_t bs_mat, bs_work;
    int src_in_work;    // level >= 1: src is the stacked R of the level below (work), else the matrix
    long long *stamps;  // diagnostic (SGPR_TSQR_STAMPS=1): [chunk][8] s_memtime at the phase boundaries of chunk 0..
};
#define LEAF_STAMP(K) if (q.stamps && threadIdx.x == 0 && blockIdx.y == 0) q.stamps[blockIdx.x * 8 + (K)] = (long long)__builtin_amdgcn_s_memtime();

// wave64 sum on the DPP network (no LDS round trips): quads, half rows, rows, then the four row sums through
// scalar registers; the result is wave-uniform
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_f64(double v, int l)
{
    return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}
__device__ __forceinline__ double wave_sum64(double v)
{
    v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
    v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
    v += dpp_f64<0x141>(v);  // row_half_mirror
    v += dpp_f64<0x140>(v);  // row_mirror
    return (lane_f64(v, 0) + lane_f64(v, 16)) + (lane_f64(v, 32) + lane_f64(v, 48));
}

// One chunk (<= 256 rows x <= 32 columns) factored by one workgroup of 512 threads with the chunk IN REGISTERS:
// thread (c, g) = (tid & 31, tid >> 5) owns the 16 rows g, g + 16, g + 32, ... of column c for the whole
// factorisation.  The interleaved ownership puts the rows that can hold a pivot (0..31) at register slots 0 and 1 of
// EVERY thread: two cheap masks per thread instead of a wave that masks all of its rows while seven others wait.
// A column step:
//   the owner of column j has published it (v_j, zero above row j) and its norm in LDS          [barrier]
//   every thread: 16 products of its rows with v_j -> partial sums in LDS                          [barrier]
//   every thread: g_c = 16 partials, rank-1 update of its 16 rows; the owner of column j + 1 publishes it.
// No cross-lane reductions.  Two waves per SIMD hide each other's latencies (the step is a chain of dependent short
// operations).  V^T V (for T) is one MFMA product at the end; T is built in 16 x 16 blocks.
// the value held by the lane 32 away (v_permlane32_swap: upper half of one operand <-> lower half of the other)
__device__ __forceinline__ double half_swap(double v)
{
    const int lo = __double2loint(v), hi = __double2hiint(v);
    const auto rl = __builtin_amdgcn_permlane32_swap(lo, lo, false, false);
    const auto rh = __builtin_amdgcn_permlane32_swap(hi, hi, false, false);
    const bool low = (threadIdx.x & 32) == 0;
    return __hiloint2double(low ? rh[1] : rh[0], low ? rl[1] : rl[0]);
}

#define TSQR_TINY2 1e-280  // squared column norms below this are zero (see tsqr_leaf_kernel)
#define TLT 512  // threads of the leaf kernel
#define TRG 16   // rows per thread (= row groups)
__global__ __launch_bounds__(TLT) void tsqr_leaf_kernel(TsqrLeaf q)
{
    extern __shared__ double lds[];
    double *sm = lds;                     // [TNB][TLD]  the chunk on its way in, V (masked) on its way out
    double *G = lds + TNB * TLD;          // [TNB][TNB + 1]:  V^T V
    double *Ts = G + TNB * (TNB + 1);     // [TNB][TNB + 1]
    __shared__ __attribute__((aligned(16))) double vbuf[2][TCH], pn[2][8];   // vbuf[.][16 g + k] = row g + 16 k
    __shared__ double part[16][TNB + 1], rowj[TNB], s_alpha[TNB], s_scal[TNB], s_vjj[TNB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = tid & 31, rg = tid >> 5, r0 = rg * TRG;
    const int chunk = blockIdx.x;
    LEAF_STAMP(0)
    const int nr = min(TCH, q.n - chunk * TCH);
    const size_t off_m = blockIdx.y * q.bs_mat, off_w = blockIdx.y * q.bs_work;
    q.src += q.src_in_work ? off_w : off_m;
    q.V += off_w; q.T += off_w;
    if (q.Rnext) q.Rnext += off_w;
    q.Rfinal += off_m;
    {
        // coalesced in (thread: 16 consecutive rows of its column), interleaved out.  Branch-free: a dead pair reads
        // the chunk's first element instead; a pair cut by the end of the matrix reads one element of padding — ldr
        // is a multiple of 64 — and drops it.
        const double *col = q.src + chunk * q.chunk_stride + (size_t)(c < q.nb ? c : 0) * q.ld;
#pragma unroll
        for (int k = 0; k < TRG; k += 2) {
            const bool ok0 = c < q.nb && r0 + k < nr, ok1 = c < q.nb && r0 + k + 1 < nr;
            const double2 v = *(const double2 *)(col + (ok0 ? r0 + k : 0));
            *(double2 *)&sm[c * TLD + r0 + k] = make_double2(ok0 ? v.x : 0.0, ok1 ? v.y : 0.0);
        }
    }
    for (int e = tid; e < TNB * (TNB + 1); e += TLT) Ts[e] = 0.0;
    if (tid < TNB) { s_alpha[tid] = 0.0; s_scal[tid] = 0.0; s_vjj[tid] = 0.0; }
    __syncthreads();
    double a[TRG];
#pragma unroll
    for (int k = 0; k < TRG; k++) a[k] = sm[c * TLD + rg + TRG * k];
    if (c == 0) {  // column 0 and its norm (per wave: both row groups)
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < TRG; k++) { vbuf[0][r0 + k] = a[k]; s += a[k] * a[k]; }
        const double sp = s + half_swap(s);
        if (lane < 32) pn[0][wave] = sp;
    }
    __syncthreads();
    LEAF_STAMP(1)
    for (int j = 0; j < q.nb; j++) {
        const int jb = j & 1;
        // the scalar chain (norm -> alpha -> 2 / v.v) does not feed the products below: they run on the RAW column x
        // (v_j = x - alpha e_j), and g_c = x . a_c - alpha a_c[j] is put together after the barrier — the chain's
        // square root and division overlap with the LDS traffic
        double s2 = 0.0;
#pragma unroll
        for (int g = 0; g < 8; g += 2) {
            const double2 x = *(const double2 *)&pn[jb][g];
            s2 += x.x + x.y;
        }
        const double akk = vbuf[jb][(j & 15) * TRG + (j >> 4)];
        // |x| and 2 / (v.v) = 1 / (|x| (|x| + |x_j|)) from the hardware seeds + one Newton step each (every wave
        // repeats this chain at every step: the IEEE sqrt / division sequences were a third of the step's instructions)
        // (a column whose squared norm is below TSQR_TINY2 counts as zero: its reflector is the identity.  The chunks
        // of a panel can be EXACTLY rank deficient — columns that are combinations of one or two reflectors of the
        // previous panel, as in the column selections of a kept factor — and then every further pivot is rounding
        // noise of rounding noise, a factor 1e-16 smaller each step, until s2 is subnormal and the reciprocal
        // square root seed overflows: NaN)
        double nrm = 0.0, sc = 0.0;
        if (s2 > TSQR_TINY2) {
            double rs = __builtin_amdgcn_rsq(s2);
            rs = rs * (1.5 - 0.5 * s2 * rs * rs);
            nrm = s2 * rs;
            nrm = nrm + 0.5 * rs * (s2 - nrm * nrm);
            const double d = nrm * (nrm + fabs(akk));
            double rd = __builtin_amdgcn_rcp(d);
            rd = rd * (2.0 - d * rd);
            sc = rd * (2.0 - d * rd);
        }
        const double alpha = akk > 0.0 ? -nrm : nrm;
        const double vjj = akk - alpha;
        if (tid == 0) { s_alpha[j] = alpha; s_scal[j] = sc; s_vjj[j] = vjj; }
        double xj[TRG], t4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < TRG; k += 2) {
            const double2 x = *(const double2 *)&vbuf[jb][r0 + k];
            xj[k] = x.x; xj[k + 1] = x.y;
        }
#pragma unroll
        for (int k = 0; k < TRG; k++) t4[k & 3] += xj[k] * a[k];
        {
            // the two row groups of a wave are summed before they reach LDS (one v_permlane32_swap per half)
            const double t = (t4[0] + t4[1]) + (t4[2] + t4[3]);
            const double tp = t + half_swap(t);
            if (lane < 32) part[wave][c] = tp;
        }
        const bool pivot_group = rg == (j & 15);
        if (pivot_group) rowj[c] = j < 16 ? a[0] : a[1];  // row j of every column
        __syncthreads();
        double g4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 8; u++) g4[u & 3] += part[u][c];
        const double g = ((g4[0] + g4[1]) + (g4[2] + g4[3])) - alpha * rowj[c];
        const double f = c > j ? sc * g : 0.0;
#pragma unroll
        for (int k = 0; k < TRG; k++) a[k] -= f * xj[k];
        if (pivot_group) {  // the pivot entry of v_j is x_j[j] - alpha
            if (j < 16) a[0] += f * alpha;
            else a[1] += f * alpha;
        }
        if (c == j + 1) {  // publish the next pivot column (zero above its diagonal: rows <= j) and its norm
            double x[TRG], s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int k = 0; k < TRG; k++) x[k] = a[k];
            if (rg <= j) x[0] = 0.0;
            if (rg + 16 <= j) x[1] = 0.0;
#pragma unroll
            for (int k = 0; k < TRG; k += 2) {
                *(double2 *)&vbuf[jb ^ 1][r0 + k] = make_double2(x[k], x[k + 1]);
                s4[k & 3] += x[k] * x[k];
                s4[(k + 1) & 3] += x[k + 1] * x[k + 1];
            }
            const double sn = (s4[0] + s4[1]) + (s4[2] + s4[3]);
            const double sp = sn + half_swap(sn);
            if (lane < 32) pn[jb ^ 1][wave] = sp;
        }
        __syncthreads();
    }
    LEAF_STAMP(2)
    // V (zero above the diagonal, v_jj on it) -> LDS: for the Gram product, and for the coalesced way out
    {
        const double d = s_vjj[c];
#pragma unroll
        for (int k = 0; k < TRG; k++) {
            const int r = rg + TRG * k;
            sm[c * TLD + r] = (c < q.nb && r >= c) ? (r == c ? d : a[k]) : 0.0;
        }
    }
    __syncthreads();
    {
        double *V = q.V + (size_t)chunk * TNB * TCH + c * TCH + r0;
#pragma unroll
        for (int k = 0; k < TRG; k += 2) *(double2 *)(V + k) = *(const double2 *)&sm[c * TLD + r0 + k];
    }
    if (wave < 4) {
        const int l15 = lane & 15, l4 = lane >> 4;
        const double *pa = sm + ((wave >> 1) * 16 + l15) * TLD + l4, *pb = sm + ((wave & 1) * 16 + l15) * TLD + l4;
        v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 4
        for (int k = 0; k < TCH; k += 16) {
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k + 4 * u], pb[k + 4 * u], acc[u], 0, 0, 0);
        }
        const v4d w = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) G[((wave >> 1) * 16 + l4 + 4 * reg) * (TNB + 1) + (wave & 1) * 16 + l15] = w[reg];
    }
    __syncthreads();
    LEAF_STAMP(3)
    // T = [T11 T12; 0 T22] in 16 x 16 blocks.  Diagonal blocks: T[t][t] = scal_t, T[t][j] = -scal_j sum_{l=t}^{j-1}
    // T[t][l] (v_l . v_j) — row t only depends on itself: lane t of each half keeps it in registers (120 multiply-adds,
    // unrolled).  Then T12 = -T11 (V1^T V2) T22 by 256 threads.
    if (tid < TNB) {
        const int t = tid & 15, base = tid & 16;
        double Tr[16];
#pragma unroll
        for (int l = 0; l < 16; l++) Tr[l] = l == t ? s_scal[base + t] : 0.0;
#pragma unroll
        for (int j = 1; j < 16; j++) {
            double s2[2] = {0.0, 0.0};
#pragma unroll
            for (int l = 0; l < j; l++) s2[l & 1] += Tr[l] * G[(base + j) * (TNB + 1) + base + l];
            const double v = -s_scal[base + j] * (s2[0] + s2[1]);
            if (j > t) Tr[j] = v;
        }
#pragma unroll
        for (int l = 0; l < 16; l++) Ts[(base + t) * (TNB + 1) + base + l] = Tr[l];
    }
    __syncthreads();
    {
        double *Xs = &part[0][0];  // [16][16]: (V1^T V2) T22
        const int ia = (tid >> 4) & 15, ib = tid & 15;
        if (tid < 256) {
            double x = 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++) x += G[ia * (TNB + 1) + 16 + l] * Ts[(16 + l) * (TNB + 1) + 16 + ib];
            Xs[ia * 16 + ib] = x;
        }
        __syncthreads();
        if (tid < 256) {
            double y = 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++) y += Ts[ia * (TNB + 1) + l] * Xs[l * 16 + ib];
            Ts[ia * (TNB + 1) + 16 + ib] = -y;
        }
    }
    __syncthreads();
    LEAF_STAMP(4)
    double *T = q.T + (size_t)chunk * TNB * TNB;
    for (int e = tid; e < TNB * TNB; e += TLT) T[e] = Ts[(e / TNB) * (TNB + 1) + e % TNB];
    // R_i (upper triangular, alpha on the diagonal): row g is slot 0 of group g, row g + 16 its slot 1
#pragma unroll
    for (int k = 0; k < 2; k++) {
        const int r = rg + TRG * k;
        const double val = c < q.nb ? (r < c ? a[k] : (r == c ? s_alpha[c] : 0.0)) : 0.0;
        if (q.Rnext) {
            const int rho = chunk * TNB + r;
            q.Rnext[(size_t)(rho / TCH) * TNB * TCH + c * TCH + rho % TCH] = val;
        } else if (c < q.nb && r <= c) {
            q.Rfinal[(size_t)c * q.ldr + r] = val;
        }
    }
    LEAF_STAMP(5)
}

__global__ __launch_bounds__(TLT) void tsqr_leaf_wave_kernel(TsqrLeaf q)
{
    extern __shared__ double lds[];
    double *sm = lds;                     // [TNB][TLD]  the chunk on its way in, V (masked) on its way out
    double *G = lds + TNB * TLD;          // [TNB][TNB + 1]:  V^T V
    double *Ts = G + TNB * (TNB + 1);     // [TNB][TNB + 1]
    __shared__ __attribute__((aligned(16))) double vbuf[2][TCH], pn[2][8];   // vbuf[.][16 g + k] = row g + 16 k
    __shared__ double part[16][TNB + 1], rowj[TNB], s_alpha[TNB], s_scal[TNB], s_vjj[TNB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int c = tid & 31, rg = tid >> 5, r0 = rg * TRG;  // (the way in: as the other form)
    const int chunk = blockIdx.x;
    LEAF_STAMP(0)
    const int nr = min(TCH, q.n - chunk * TCH);
    const size_t off_m = blockIdx.y * q.bs_mat, off_w = blockIdx.y * q.bs_work;
    q.src += q.src_in_work ? off_w : off_m;
    q.V += off_w; q.T += off_w;
    if (q.Rnext) q.Rnext += off_w;
    q.Rfinal += off_m;
    {
        // coalesced in (thread: 16 consecutive rows of its column), interleaved out.  Branch-free: a dead pair reads
        // the chunk's first element instead; a pair cut by the end of the matrix reads one element of padding — ldr
        // is a multiple of 64 — and drops it.
        const double *col = q.src + chunk * q.chunk_stride + (size_t)(c < q.nb ? c : 0) * q.ld;
#pragma unroll
        for (int k = 0; k < TRG; k += 2) {
            const bool ok0 = c < q.nb && r0 + k < nr, ok1 = c < q.nb && r0 + k + 1 < nr;
            const double2 v = *(const double2 *)(col + (ok0 ? r0 + k : 0));
            *(double2 *)&sm[c * TLD + r0 + k] = make_double2(ok0 ? v.x : 0.0, ok1 ? v.y : 0.0);
        }
    }
    for (int e = tid; e < TNB * (TNB + 1); e += TLT) Ts[e] = 0.0;
    if (tid < TNB) { s_alpha[tid] = 0.0; s_scal[tid] = 0.0; s_vjj[tid] = 0.0; }
    __syncthreads();
    // ---- register layout of THIS form: wave w owns the columns w, w + 8, w + 16, w + 24 (cyclic: the waves stay equally
    // loaded while the active columns shrink), lane l the rows l, l + 64, l + 128, l + 192 of each.  A column step:
    //   every wave reads v_j and 2 / v.v from LDS; the owner wave of column j + 1 updates THAT column first, takes its norm by
    //   a wave reduction, runs the scalar chain once (not in every wave) and publishes v_(j+1) at once; then every wave:
    //   g_c = v_j . a_c by a wave reduction per own column c > j + 1, a_c -= (2 / v.v) g_c v_j                  [barrier]
    // One barrier per step (v_j alternates between two buffers), no partial sums through LDS, the reductions on the DPP
    // network, the pivot's chain hidden behind the other waves' updates (look-ahead of one column).  (The other form keeps sixteen row groups per column across the waves: two barriers and ~240 instructions
    // per wave and step, 2770 cycles; stamps in tools/README.md.)
    const int wv = tid >> 6, ln = tid & 63;
    double a[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int k = 0; k < 4; k++) a[i][k] = sm[(wv + 8 * i) * TLD + ln + 64 * k];
    __shared__ double sscal[2];
    // publish column j of the owner wave (its values x, rows l + 64 k of lane l): v_j, 2 / v.v, and the scalars kept for T and R
    auto publish = [&](int j, const double (&col)[4]) {
        const int jb = j & 1;
        double x[4];
#pragma unroll
        for (int k = 0; k < 4; k++) x[k] = ln + 64 * k >= j ? col[k] : 0.0;  // zero above the diagonal
        const double s2 = wave_sum64((x[0] * x[0] + x[1] * x[1]) + (x[2] * x[2] + x[3] * x[3]));
        const double akk = lane_f64(x[0], j);  // (j < 32: row j is slot 0 of lane j)
        double nrm = 0.0, sc = 0.0;
        if (s2 > TSQR_TINY2) {  // (see the other form: rank-deficient chunks)
            double rs = __builtin_amdgcn_rsq(s2);
            rs = rs * (1.5 - 0.5 * s2 * rs * rs);
            nrm = s2 * rs;
            nrm = nrm + 0.5 * rs * (s2 - nrm * nrm);
            const double d = nrm * (nrm + fabs(akk));
            double rd = __builtin_amdgcn_rcp(d);
            rd = rd * (2.0 - d * rd);
            sc = rd * (2.0 - d * rd);
        }
        const double alpha = akk > 0.0 ? -nrm : nrm;
        const double vjj = akk - alpha;
        if (ln == j) x[0] = vjj;  // the pivot entry of v_j
#pragma unroll
        for (int k = 0; k < 4; k++) vbuf[jb][ln + 64 * k] = x[k];
        if (ln == 0) { sscal[jb] = sc; s_alpha[j] = alpha; s_scal[j] = sc; s_vjj[j] = vjj; }
    };
    if (wv == 0) publish(0, a[0]);
    __syncthreads();
    LEAF_STAMP(1)
    for (int j = 0; j < q.nb; j++) {
        const int jb = j & 1;
        const double sc = sscal[jb];
        double v[4];
#pragma unroll
        for (int k = 0; k < 4; k++) v[k] = vbuf[jb][ln + 64 * k];
        // the NEXT pivot column first: its owner updates it and publishes v_(j+1) at once, while the other waves (and this one,
        // afterwards) are still busy with step j — the owner's chain (update, norm, scalars) is off the step's critical path
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (wv + 8 * i == j + 1 && j + 1 < q.nb) {  // (wave-uniform)
                const double g = wave_sum64((v[0] * a[i][0] + v[1] * a[i][1]) + (v[2] * a[i][2] + v[3] * a[i][3]));
                const double f = sc * g;
#pragma unroll
                for (int k = 0; k < 4; k++) a[i][k] -= f * v[k];
                publish(j + 1, a[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; i++) {
            if (wv + 8 * i > j + 1) {  // (wave-uniform)
                const double g = wave_sum64((v[0] * a[i][0] + v[1] * a[i][1]) + (v[2] * a[i][2] + v[3] * a[i][3]));
                const double f = sc * g;
#pragma unroll
                for (int k = 0; k < 4; k++) a[i][k] -= f * v[k];
            }
        }
        __syncthreads();
    }
    LEAF_STAMP(2)
    // V (zero above the diagonal, v_jj on it) -> LDS: for the Gram product, and for the coalesced way out
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const int cc = wv + 8 * i;
        const double d = s_vjj[cc];
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int r = ln + 64 * k;
            sm[cc * TLD + r] = (cc < q.nb && r >= cc) ? (r == cc ? d : a[i][k]) : 0.0;
        }
    }
    __syncthreads();
    {
        double *V = q.V + (size_t)chunk * TNB * TCH + c * TCH + r0;
#pragma unroll
        for (int k = 0; k < TRG; k += 2) *(double2 *)(V + k) = *(const double2 *)&sm[c * TLD + r0 + k];
    }
    if (wave < 4) {
        const int l15 = lane & 15, l4 = lane >> 4;
        const double *pa = sm + ((wave >> 1) * 16 + l15) * TLD + l4, *pb = sm + ((wave & 1) * 16 + l15) * TLD + l4;
        v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll 4
        for (int k = 0; k < TCH; k += 16) {
#pragma unroll
            for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(pa[k + 4 * u], pb[k + 4 * u], acc[u], 0, 0, 0);
        }
        const v4d w = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) G[((wave >> 1) * 16 + l4 + 4 * reg) * (TNB + 1) + (wave & 1) * 16 + l15] = w[reg];
    }
    __syncthreads();
    LEAF_STAMP(3)
    // T = [T11 T12; 0 T22] in 16 x 16 blocks.  Diagonal blocks: T[t][t] = scal_t, T[t][j] = -scal_j sum_{l=t}^{j-1}
    // T[t][l] (v_l . v_j) — row t only depends on itself: lane t of each half keeps it in registers (120 multiply-adds,
    // unrolled).  Then T12 = -T11 (V1^T V2) T22 by 256 threads.
    if (tid < TNB) {
        const int t = tid & 15, base = tid & 16;
        double Tr[16];
#pragma unroll
        for (int l = 0; l < 16; l++) Tr[l] = l == t ? s_scal[base + t] : 0.0;
#pragma unroll
        for (int j = 1; j < 16; j++) {
            double s2[2] = {0.0, 0.0};
#pragma unroll
            for (int l = 0; l < j; l++) s2[l & 1] += Tr[l] * G[(base + j) * (TNB + 1) + base + l];
            const double v = -s_scal[base + j] * (s2[0] + s2[1]);
            if (j > t) Tr[j] = v;
        }
#pragma unroll
        for (int l = 0; l < 16; l++) Ts[(base + t) * (TNB + 1) + base + l] = Tr[l];
    }
    __syncthreads();
    {
        double *Xs = &part[0][0];  // [16][16]: (V1^T V2) T22
        const int ia = (tid >> 4) & 15, ib = tid & 15;
        if (tid < 256) {
            double x = 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++) x += G[ia * (TNB + 1) + 16 + l] * Ts[(16 + l) * (TNB + 1) + 16 + ib];
            Xs[ia * 16 + ib] = x;
        }
        __syncthreads();
        if (tid < 256) {
            double y = 0.0;
#pragma unroll
            for (int l = 0; l < 16; l++) y += Ts[ia * (TNB + 1) + l] * Xs[l * 16 + ib];
            Ts[ia * (TNB + 1) + 16 + ib] = -y;
        }
    }
    __syncthreads();
    LEAF_STAMP(4)
    double *T = q.T + (size_t)chunk * TNB * TNB;
    for (int e = tid; e < TNB * TNB; e += TLT) T[e] = Ts[(e / TNB) * (TNB + 1) + e % TNB];
    // R_i (upper triangular, alpha on the diagonal): rows 0..31 are slot 0 of lanes 0..31
    if (ln < TNB) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int cc = wv + 8 * i, r = ln;
            const double val = cc < q.nb ? (r < cc ? a[i][0] : (r == cc ? s_alpha[cc] : 0.0)) : 0.0;
            if (q.Rnext) {
                const int rho = chunk * TNB + r;
                q.Rnext[(size_t)(rho / TCH) * TNB * TCH + cc * TCH + rho % TCH] = val;
            } else if (cc < q.nb && r <= cc) {
                q.Rfinal[(size_t)cc * q.ldr + r] = val;
            }
        }
    }
    LEAF_STAMP(5)
}

struct TsqrApply {
    double *A;        // trailing columns: A[c * ldr + physical row]
    int ldr, ntrail;
    int row0, row_end;  // physical rows of the panel: [row0, row_end)
    int n;            // logical rows of this level
    int stride;       // physical rows between consecutive 32-row blocks of the logical numbering (32: contiguous)
    int tpw;          // column tiles per workgroup
    size_t bs_mat, bs_work;  // batch of independent problems (blockIdx.z)
    const double *V;  // [chunk][TNB][TCH]
    const double *T;  // [chunk][TNB][TNB]
};

// A_i <- A_i - V_i (T_i^T (V_i^T A_i)) for one chunk and a run of `tpw` tiles of 32 trailing columns: V_i and T_i are
// loaded once, the next tile of A is in flight (registers) while the current one is worked on in LDS.
// W = V^T A: four 16 x 16 blocks, one per wave, the 256-deep contraction on four interleaved accumulators;
// Z = T^T W: one 16 x 16 block per wave; A -= V Z: 16 row blocks x 2 column blocks, four row blocks per wave.
__global__ __launch_bounds__(TCH) void tsqr_apply_kernel(TsqrApply q)
{
    extern __shared__ double lds[];
    double *Vs = lds;                    // [TNB][TLD]   Vs[i][r] = V[r][i]
    double *As = lds + TNB * TLD;        // [TNB][TLD]   As[c][r] = A[r][c]
    double *Ts = As + TNB * TLD;         // [TNB][TNB + 1]
    double *Ws = Ts + TNB * (TNB + 1);   // [TNB][TNB + 1]
    double *Zs = Ws + TNB * (TNB + 1);   // [TNB][TNB + 1]  (-Z)
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int chunk = blockIdx.y;
    q.A += blockIdx.z * q.bs_mat;
    q.V += blockIdx.z * q.bs_work; q.T += blockIdx.z * q.bs_work;
    const int tile0 = blockIdx.x * q.tpw, tile1 = min(tile0 + q.tpw, (q.ntrail + TNB - 1) / TNB);
    // rows (2p, 2p + 1) of the chunk x 16 columns per thread; logical row -> physical row of the matrix
    const int p = tid & 127, h = tid >> 7;
    const int rho = chunk * TCH + 2 * p;
    const int phys = q.row0 + (rho >> 5) * q.stride + (rho & 31);
    const bool live0 = rho < q.n && phys < q.row_end, live1 = rho + 1 < q.n && phys + 1 < q.row_end;
    const double *V = q.V + (size_t)chunk * TNB * TCH;
    double2 aa[16];
    auto fetch = [&](int tile) {
        const int c0 = tile * TNB, nc = min(TNB, q.ntrail - c0);
        const double *A = q.A + (size_t)c0 * q.ldr + phys;
#pragma unroll
        for (int u = 0; u < 16; u++) {
            const int c = 16 * h + u;
            const double *a = A + (size_t)c * q.ldr;
            aa[u] = make_double2(0.0, 0.0);
            if (c < nc) {
                if (live1) aa[u] = *(const double2 *)a;
                else if (live0) aa[u].x = a[0];
            }
        }
    };
    fetch(tile0);
    {
        double2 vv[16];
#pragma unroll
        for (int u = 0; u < 16; u++) vv[u] = *(const double2 *)&V[(16 * h + u) * TCH + 2 * p];
#pragma unroll
        for (int u = 0; u < 16; u++) *(double2 *)&Vs[(16 * h + u) * TLD + 2 * p] = vv[u];
    }
    for (int e = tid; e < TNB * TNB; e += TCH) Ts[(e / TNB) * (TNB + 1) + e % TNB] = q.T[(size_t)chunk * TNB * TNB + e];
    const int l15 = lane & 15, l4 = lane >> 4;
    const int i0 = (wave >> 1) * 16, cc0 = (wave & 1) * 16;
    for (int tile = tile0; tile < tile1; tile++) {
#pragma unroll
        for (int u = 0; u < 16; u++) *(double2 *)&As[(16 * h + u) * TLD + 2 * p] = aa[u];
        __syncthreads();
        if (tile + 1 < tile1) fetch(tile + 1);
        {
            // W[i][c] = sum_r V[r][i] A[r][c]:  A operand (row i, k = r), B operand (k = r, col c)
            v4d acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
            const double *va = Vs + (i0 + l15) * TLD + l4;
            const double *ab = As + (cc0 + l15) * TLD + l4;
#pragma unroll 4
            for (int k = 0; k < TCH; k += 16) {
#pragma unroll
                for (int u = 0; u < 4; u++) acc[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(va[k + 4 * u], ab[k + 4 * u], acc[u], 0, 0, 0);
            }
            const v4d w = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Ws[(i0 + l4 + 4 * reg) * (TNB + 1) + cc0 + l15] = w[reg];
        }
        __syncthreads();
        {
            // -Z[i][c] = -sum_l T[l][i] W[l][c]:  A operand (row i, k = l) = T[l][i], B operand (k = l, col c) = W[l][c]
            v4d acc[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
#pragma unroll
            for (int k = 0; k < TNB; k += 8) {
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ts[(k + l4) * (TNB + 1) + i0 + l15], Ws[(k + l4) * (TNB + 1) + cc0 + l15], acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(Ts[(k + 4 + l4) * (TNB + 1) + i0 + l15], Ws[(k + 4 + l4) * (TNB + 1) + cc0 + l15], acc[1], 0, 0, 0);
            }
            const v4d z = acc[0] + acc[1];
#pragma unroll
            for (int reg = 0; reg < 4; reg++) Zs[(i0 + l4 + 4 * reg) * (TNB + 1) + cc0 + l15] = -z[reg];
        }
        __syncthreads();
        {
            // A[r][c] += sum_i V[r][i] (-Z)[i][c]:  A operand (row r, k = i), B operand (k = i, col c); C/D: col = lane & 15,
            // row = (lane >> 4) + 4 reg
            const int r0 = wave * 64;
#pragma unroll
            for (int cb = 0; cb < 2; cb++) {
                v4d acc[4];
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) acc[rb][reg] = As[(16 * cb + l15) * TLD + r0 + 16 * rb + l4 + 4 * reg];
#pragma unroll
                for (int k = 0; k < TNB; k += 4) {
                    const double zb = Zs[(k + l4) * (TNB + 1) + 16 * cb + l15];
#pragma unroll
                    for (int rb = 0; rb < 4; rb++)
                        acc[rb] = __builtin_amdgcn_mfma_f64_16x16x4f64(Vs[(k + l4) * TLD + r0 + 16 * rb + l15], zb, acc[rb], 0, 0, 0);
                }
#pragma unroll
                for (int rb = 0; rb < 4; rb++)
#pragma unroll
                    for (int reg = 0; reg < 4; reg++) As[(16 * cb + l15) * TLD + r0 + 16 * rb + l4 + 4 * reg] = acc[rb][reg];
            }
        }
        __syncthreads();
        {
            const int c0 = tile * TNB, nc = min(TNB, q.ntrail - c0);
            double *A = q.A + (size_t)c0 * q.ldr + phys;
#pragma unroll
            for (int u = 0; u < 16; u++) {
                const int c = 16 * h + u;
                if (c < nc) {
                    const double2 v = *(const double2 *)&As[c * TLD + 2 * p];
                    double *a = A + (size_t)c * q.ldr;
                    if (live1) *(double2 *)a = v;
                    else if (live0) a[0] = v.x;
                }
            }
        }
        __syncthreads();  // As is overwritten by the next tile
    }
}

// R (upper triangle of the factored work array), column-major Rc[c * ldc + i] (i <= c) or row-major Rc[i * ldc + c],
// and z
template <bool ROWMAJOR>
__global__ void qr_gather_r_kernel(int cols, const double *At, int ldr, double *Rc, int ldc, double *z, size_t bs_mat = 0,
                                   size_t bs_work = 0)
{
    At += blockIdx.z * bs_mat; Rc += blockIdx.z * bs_work; z += blockIdx.z * bs_work;  // (a batch of problems over blockIdx.z)
    if (ROWMAJOR) {
        const int i = blockIdx.y, c = blockIdx.x * blockDim.x + threadIdx.x;
        if (i < cols && c < cols) Rc[(size_t)i * ldc + c] = c >= i ? At[(size_t)c * ldr + i] : 0.0;
        if (i < cols && c == 0) z[i] = At[(size_t)cols * ldr + i];
    } else {
        const int c = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
        if (c < cols && i < cols) Rc[(size_t)c * ldc + i] = i <= c ? At[(size_t)c * ldr + i] : 0.0;
        if (c == 0 && i < cols) z[i] = At[(size_t)cols * ldr + i];
    }
}

// back substitution R x = z, R row-major [cols][cols], in blocks of 32 columns from the end: wave 0 solves the
// 32 x 32 diagonal block (lane = row, its row of the block in registers, the pivots passed through readlane: no
// barriers inside), then all 1024 threads take the block's contribution out of the rows above (one 32-term dot per
// row).  Two workgroup barriers per 32 columns instead of two per column.
__global__ __launch_bounds__(1024) void qr_backsolve_kernel(int cols, const double *A, int ld, const double *y, double *x,
                                                           size_t bs_mat = 0)
{
    // R COLUMN-major: element (i, c) at A[c * ld + i] — the factored matrix itself (its transposed storage), no gathered copy:
    // a thread's row i of the 32-column block is 32 loads that are contiguous ACROSS the threads (the row-major copy made
    // every one of them a cache line of its own: 17 us per block of 32 at m = 1024, 0.54 ms per solve)
    extern __shared__ double zs[];   // [cols] (dynamic: cols <= QR_MAX_COLS)
    __shared__ double xb[32];
    A += blockIdx.x * bs_mat; y += blockIdx.x * bs_mat; x += (size_t)blockIdx.x * cols;  // (a batch: one workgroup per problem)
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < cols; i += 1024) zs[i] = y[i];
    __syncthreads();
    for (int b1 = cols; b1 > 0; b1 -= 32) {
        const int b0 = max(b1 - 32, 0), nbk = b1 - b0;
        if (tid < 64) {
            const bool on = lane < nbk;
            const int i = b0 + (on ? lane : 0);
            double rr[32];
#pragma unroll
            for (int k = 0; k < 32; k++) rr[k] = (on && k < nbk) ? A[(size_t)(b0 + k) * ld + i] : (k == lane ? 1.0 : 0.0);
            double zi = on ? zs[i] : 0.0;
#pragma unroll
            for (int k = 31; k >= 0; k--) {
                if (k < nbk) {  // (uniform)
                    const double xk = lane_f64(zi, k) / lane_f64(rr[k], k);
                    if (lane < k) zi -= rr[k] * xk;
                    if (lane == k) xb[k] = xk;
                }
            }
        }
        __syncthreads();
        for (int i = tid; i < b0; i += 1024) {
            const double *r = A + (size_t)b0 * ld + i;
            double s4[4] = {0.0, 0.0, 0.0, 0.0};
            for (int k = 0; k < nbk; k++) s4[k & 3] += r[(size_t)k * ld] * xb[k];
            zs[i] -= (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
        if (tid < nbk) x[b0 + tid] = xb[tid];
        __syncthreads();
    }
}

// The only thing in the least squares that is sized by the number of columns is the right-hand side the back substitution
// keeps in LDS: 8192 columns are 64 KB (the reference has no limit on the inducing set; m <= 2048 was this array's size).
#define QR_MAX_COLS 8192
static void backsolve_attr()
{
    static bool done = false;
    if (!done) {
        (void)hipFuncSetAttribute((const void *)qr_backsolve_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(sizeof(double) * QR_MAX_COLS));
        done = true;
    }
}

// chunks of every level of the tree over n rows (level 0 first)
static void tsqr_levels(int n, std::vector<int> &chunks)
{
    chunks.clear();
    for (;;) {
        const int c = (n + TCH - 1) / TCH;
        chunks.push_back(c);
        if (c == 1) break;
        n = c * TNB;
    }
}

size_t lstsq_qr_blocked_work_doubles(int rows, int cols)
{
    std::vector<int> ch;
    tsqr_levels(std::max(rows, 1), ch);
    size_t total = 0;
    for (int c : ch) total += (size_t)c * (2 * TNB * TCH + TNB * TNB);  // V, stacked R of the level above, T
    for (int c : ch) total += (size_t)c * (TNB * TCH + TNB * TNB);      // a second V | T set (look-ahead: tsqr_panel)
    total = std::max(total, (size_t)2 * (32 * 1088 + 32 * 32));          // (the flat-panel form of banded problems: bandqr.inc)
    return total + (size_t)cols * cols + cols + 64;
}

// doubles of V and T a kept panel over n rows needs (all levels)
size_t tsqr_panel_doubles(int n)
{
    std::vector<int> ch;
    tsqr_levels(std::max(n, 1), ch);
    size_t total = 0;
    for (int c : ch) total += (size_t)c * (TNB * TCH + TNB * TNB);
    return total;
}

size_t tsqr_keep_doubles(int rows, int cols, int band)
{
    size_t total = 0;
    for (int k0 = 0; k0 < cols; k0 += TNB) {
        const int nb = std::min(TNB, cols - k0);
        const int row_end = band > 0 ? std::min(rows, band * (k0 + nb)) : rows;
        total += tsqr_panel_doubles(row_end - k0);
    }
    return total;
}

static size_t lds_leaf_bytes() { return sizeof(double) * (TNB * TLD + 2 * TNB * (TNB + 1)); }
static size_t lds_apply_bytes() { return sizeof(double) * (2 * TNB * TLD + 3 * TNB * (TNB + 1)); }
static void tsqr_attrs()
{
    static bool done = false;
    if (done) return;
    (void)hipFuncSetAttribute((const void *)tsqr_leaf_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_leaf_bytes());
    (void)hipFuncSetAttribute((const void *)tsqr_leaf_wave_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_leaf_bytes());
    (void)hipFuncSetAttribute((const void *)tsqr_apply_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_apply_bytes());
    done = true;
}

// One panel: factor columns [k0, k0 + nb) of the column-major matrix `panel_cols` (element (c, r) at
// panel_cols[c * ld + r], r a physical row) over rows [k0, row_end), level by level, and update `ntrail` trailing
// columns (trail[c * ld + r]) after every level.  V / T of every level go to `keep` (consecutive) when given — the
// panel can then be applied again later (tsqr_apply_panels) — else to the per-level scratch in `work`.
struct TsqrLookahead {
    // the trailing update in two parts (launch_lstsq_qr_blocked): `near` columns right behind the panel on the panel's own
    // stream, the rest on `far_stream` once `leaf_done` is reached; null far_stream: everything on the panel's stream
    int near = 0;
    hipStream_t far_stream = nullptr;
    hipEvent_t leaf_done = nullptr, far_wait = nullptr, far_done = nullptr;
    double *set2 = nullptr;   // V | T scratch of the odd panels (the far update of the panel before still reads the even set); null: even
    int layout_rows = 0;      // the scratch layout is that of a panel over this many rows, the SAME for every panel of the call: a
                              // layout that followed each panel's own row count would shift when a 256-row chunk drops out, and
                              // the next panel's stacked R would land on the V | T the far update is still reading
};

static void tsqr_panel(double *panel_cols, int ld, int k0, int nb, int row_end, double *trail, int ntrail, double *work,
                       double *keep, TsqrPanel *rec, hipStream_t st, int batch = 1, size_t bs_mat = 0, size_t bs_work = 0,
                       const TsqrLookahead *la = nullptr)
{
    std::vector<int> ch;
    tsqr_levels(std::max(la && la->layout_rows > 0 && !keep ? la->layout_rows : row_end - k0, 1), ch);
    // scratch layout of this panel: per level V | S (input of the level) | T; a second V | T set behind (look-ahead)
    std::vector<double *> Vl(ch.size()), Tl(ch.size()), Sl(ch.size());
    double *w = work;
    for (size_t l = 0; l < ch.size(); l++) {
        Vl[l] = w; w += (size_t)ch[l] * TNB * TCH;
        Sl[l] = w; w += (size_t)ch[l] * TNB * TCH;
        Tl[l] = w; w += (size_t)ch[l] * TNB * TNB;
    }
    if (la && la->set2 && !keep) {
        w = la->set2;
        for (size_t l = 0; l < ch.size(); l++) {
            Vl[l] = w; w += (size_t)ch[l] * TNB * TCH;
            Tl[l] = w; w += (size_t)ch[l] * TNB * TNB;
        }
    }
    if (keep) {
        double *kp = keep;
        for (size_t l = 0; l < ch.size(); l++) {
            Vl[l] = kp; kp += (size_t)ch[l] * TNB * TCH;
            Tl[l] = kp; kp += (size_t)ch[l] * TNB * TNB;
        }
    }
    if (rec) { rec->k0 = k0; rec->nb = nb; rec->row_end = row_end; rec->nlev = 0; }
    struct Lev { int n, chunks, stride; };
    std::vector<Lev> levs;
    // the update of a run of trailing columns by every level in turn (a level's leaf does not read the trailing columns: the
    // leaves of all levels may run before any update, and the updates of different column ranges beside one another)
    auto apply = [&](double *cols0, int ncols, hipStream_t s_, size_t l_first, size_t l_end) {
        if (ncols <= 0) return;
        for (size_t l = l_first; l < l_end; l++) {
            TsqrApply ap = {};
            ap.A = cols0;
            ap.ldr = ld; ap.ntrail = ncols;
            ap.row0 = k0; ap.row_end = row_end;
            ap.n = levs[l].n; ap.stride = levs[l].stride;
            ap.V = Vl[l]; ap.T = Tl[l];
            // enough workgroups to fill the chip several times over, else as many tiles per workgroup as possible
            const int ntiles = (ncols + TNB - 1) / TNB;
            ap.tpw = std::max(1, std::min(8, (int)((size_t)ntiles * levs[l].chunks * batch / 1024)));
            ap.bs_mat = bs_mat; ap.bs_work = bs_work;
            hipLaunchKernelGGL(tsqr_apply_kernel, dim3((ntiles + ap.tpw - 1) / ap.tpw, levs[l].chunks, batch), dim3(TCH), lds_apply_bytes(), s_, ap);
        }
    };
    const bool split = la && la->far_stream;
    int n = row_end - k0, stride = TNB;
    for (int l = 0;; l++) {
        const int chunks = (n + TCH - 1) / TCH;
        const bool top = chunks == 1;
        TsqrLeaf lf = {};
        if (l == 0) { lf.src = panel_cols + k0; lf.chunk_stride = TCH; lf.ld = ld; }
        else { lf.src = Sl[l]; lf.chunk_stride = (size_t)TNB * TCH; lf.ld = TCH; }
        lf.n = n; lf.nb = nb;
        lf.V = Vl[l]; lf.T = Tl[l];
        lf.Rnext = top ? nullptr : Sl[l + 1];
        lf.Rfinal = panel_cols + k0;
        lf.ldr = ld;
        lf.bs_mat = bs_mat; lf.bs_work = bs_work; lf.src_in_work = l > 0;
        {
            static long long *d_st = nullptr;
            static const bool on = getenv("SGPR_TSQR_STAMPS") != nullptr;
            if (on && !d_st) (void)hipMalloc((void **)&d_st, sizeof(long long) * 8 * 4096);
            lf.stamps = on && chunks <= 4096 ? d_st : nullptr;
            if (on && d_st && l == 0 && k0 >= 512 && k0 < 544) {  // print the previous launch's stamps now and then
                long long hs[8 * 8];
                (void)hipStreamSynchronize(st);
                (void)hipMemcpy(hs, d_st, sizeof(hs), hipMemcpyDeviceToHost);
                fprintf(stderr, "[tsqr stamps] chunk 0 of the previous leaf: load %lld | steps %lld | V out + Gram %lld | T %lld | T out %lld..end %lld cycles\n",
                        hs[1] - hs[0], hs[2] - hs[1], hs[3] - hs[2], hs[4] - hs[3], hs[5] - hs[4], hs[5] - hs[0]);
            }
        }
        {
            static const bool rows_form = getenv("SGPR_TSQR_LEAF") && atoi(getenv("SGPR_TSQR_LEAF")) == 1;  // the row-group form
            if (rows_form) hipLaunchKernelGGL(tsqr_leaf_kernel, dim3(chunks, batch), dim3(TLT), lds_leaf_bytes(), st, lf);
            else hipLaunchKernelGGL(tsqr_leaf_wave_kernel, dim3(chunks, batch), dim3(TLT), lds_leaf_bytes(), st, lf);
        }
        if (rec && rec->nlev < 8) {
            TsqrLevel &lv = rec->lv[rec->nlev++];
            lv.V = Vl[l]; lv.T = Tl[l]; lv.n = n; lv.chunks = chunks; lv.stride = stride;
        }
        levs.push_back({n, chunks, stride});
        if (!split) apply(trail, ntrail, st, (size_t)l, (size_t)l + 1);   // (one stream: level by level, as the tree goes up)
        if (top) break;
        n = chunks * TNB;
        stride = l == 0 ? TCH : stride * TSB;
    }
    if (split) {
        const int near = std::min(la->near, ntrail);
        (void)hipEventRecord(la->leaf_done, st);
        // the far columns: beside whatever this stream does next (the near columns, then the next panel's leaves)
        if (ntrail - near > 0) {
            (void)hipStreamWaitEvent(la->far_stream, la->leaf_done, 0);
            apply(trail + (size_t)near * ld, ntrail - near, la->far_stream, 0, levs.size());
        }
        (void)hipEventRecord(la->far_done, la->far_stream);
        // the near columns were last written by the far update of the panel before
        if (la->far_wait) (void)hipStreamWaitEvent(st, la->far_wait, 0);
        apply(trail, near, st, 0, levs.size());
    }
}

// the second stream and the events of the look-ahead: one set per host thread and device
struct LookaheadRes {
    hipStream_t side = nullptr;
    hipEvent_t leaf[2] = {nullptr, nullptr}, far[2] = {nullptr, nullptr};
};
static LookaheadRes *lookahead_res()
{
    thread_local LookaheadRes res[16];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
    LookaheadRes &r = res[dev];
    if (!r.side) {
        if (hipStreamCreateWithFlags(&r.side, hipStreamNonBlocking) != hipSuccess) { r.side = nullptr; return nullptr; }
        for (int i = 0; i < 2; i++)
            if (hipEventCreateWithFlags(&r.leaf[i], hipEventDisableTiming) != hipSuccess ||
                hipEventCreateWithFlags(&r.far[i], hipEventDisableTiming) != hipSuccess)
                return nullptr;
    }
    return &r;
}

int g_bandqr_force = -1;   // tools/bandqr_test.hip: 0 / 1 overrides the environment
static bool band_qr_on()
{
    static const bool on = !(getenv("SGPR_BANDQR") && atoi(getenv("SGPR_BANDQR")) == 0);   // SGPR_BANDQR=0: the tree form everywhere
    return g_bandqr_force >= 0 ? g_bandqr_force != 0 : on;
}
#include "bandqr.inc"

int launch_lstsq_qr_blocked(int rows, int cols, double *At, int ldr, double *x, double *work, hipStream_t st, int band,
                            double *keep, std::vector<TsqrPanel> *panels, int extra, int band_off)
{
    if (cols > QR_MAX_COLS || rows < cols || ldr < rows) return -1;
    // banded problems whose panels fit one workgroup's registers: one launch per panel (bandqr.inc)
    if (band > 0 && !keep && !panels && band_qr_on() &&
        launch_band_qr(rows, cols, At, ldr, x, work, st, band, band_off, extra, 1, 0, 0) == 0)
        return 0;
    tsqr_attrs();
    size_t scratch = 0, set1 = 0;
    {
        std::vector<int> ch;
        tsqr_levels(rows, ch);
        for (int c : ch) set1 += (size_t)c * (2 * TNB * TCH + TNB * TNB);
        for (int c : ch) scratch += (size_t)c * (3 * TNB * TCH + 2 * TNB * TNB);
    }
    double *Rc = work + scratch, *z = Rc + (size_t)cols * cols;
    if (panels) panels->clear();
    // Look-ahead (tall dense problems): a panel's leaves occupy one wave per 256-row chunk — a fifth of the chip at 50 000
    // rows — and its trailing update is bound by HBM, so the update of the columns BEHIND the next panel runs on a second
    // stream beside the next panel's leaves; only the next panel's own 32 columns are updated in line.  Same kernels, same
    // operands, same order for every column: the factorisation is bit for bit the one-stream one (SGPR_QR_LOOKAHEAD=0).
    // Measured at 16384-atom frames, m = 1024: the rows of a frame appended to a kept factor (50 183 x 1024) 14.7 -> 13.3 ms
    // — the chain leaves -> next panel's own columns -> leaves (330 us per panel, eight small launches) is what remains;
    // a factorisation that KEEPS its reflectors (98 318 rows and more) gained nothing and stays on one stream.
    static const bool la_on = !(getenv("SGPR_QR_LOOKAHEAD") && atoi(getenv("SGPR_QR_LOOKAHEAD")) == 0);
    LookaheadRes *res = (la_on && band <= 0 && !keep && rows >= 8192 && cols > TNB) ? lookahead_res() : nullptr;
    int npanel = 0;
    for (int k0 = 0; k0 < cols; k0 += TNB, npanel++) {
        const int nb = std::min(TNB, cols - k0);
        const int row_end = band > 0 ? std::min(rows, band * (k0 + nb) + band_off) : rows;
        const int ntrail = cols + 1 + extra - (k0 + nb);  // the other columns, the targets, `extra` columns that follow Q^T
        TsqrPanel rec;
        TsqrLookahead la;
        if (res) {
            la.near = std::min(TNB, cols - (k0 + nb));   // the next panel's columns (none behind the last panel)
            la.far_stream = res->side;
            la.leaf_done = res->leaf[npanel & 1];
            la.far_done = res->far[npanel & 1];
            la.far_wait = npanel > 0 ? res->far[(npanel - 1) & 1] : nullptr;
            la.set2 = (npanel & 1) ? work + set1 : nullptr;
            la.layout_rows = rows;
        }
        tsqr_panel(At + (size_t)k0 * ldr, ldr, k0, nb, row_end, At + (size_t)(k0 + nb) * ldr, ntrail, work, keep,
                   panels ? &rec : nullptr, st, 1, 0, 0, res ? &la : nullptr);
        if (keep) keep += tsqr_panel_doubles(row_end - k0);
        if (panels) panels->push_back(rec);
    }
    if (res && npanel > 0) (void)hipStreamWaitEvent(st, res->far[(npanel - 1) & 1], 0);   // the caller's stream sees the whole result
    if (x) {
        (void)Rc; (void)z;
        backsolve_attr();
        hipLaunchKernelGGL(qr_backsolve_kernel, dim3(1), dim3(1024), sizeof(double) * cols, st, cols, At, ldr, At + (size_t)cols * ldr, x);
    }
    return 0;
}

// `batch` independent least-squares problems of one shape in the same launches (grid dimension = problem): matrix b at
// At + b * bs_mat, its solution at x + b * cols, its scratch at work + b * lstsq_qr_blocked_work_doubles(rows, cols).
// The second stage of the regression for a dozen noise values at once: every launch of a single 2m x m band problem
// occupies at most five of the 256 CUs.
int launch_lstsq_qr_batched(int rows, int cols, double *At, int ldr, size_t bs_mat, double *x, double *work, int batch,
                            hipStream_t st, int band)
{
    if (cols > QR_MAX_COLS || rows < cols || ldr < rows || batch < 1) return -1;
    tsqr_attrs();
    const size_t bs_work = lstsq_qr_blocked_work_doubles(rows, cols);
    if (band > 0 && band_qr_on() && launch_band_qr(rows, cols, At, ldr, x, work, st, band, 0, 0, batch, bs_mat, bs_work) == 0)
        return 0;
    size_t scratch = 0;
    {
        std::vector<int> ch;
        tsqr_levels(rows, ch);
        for (int c : ch) scratch += (size_t)c * (2 * TNB * TCH + TNB * TNB);
    }
    for (int k0 = 0; k0 < cols; k0 += TNB) {
        const int nb = std::min(TNB, cols - k0);
        const int row_end = band > 0 ? std::min(rows, band * (k0 + nb)) : rows;
        const int ntrail = cols + 1 - (k0 + nb);
        tsqr_panel(At + (size_t)k0 * ldr, ldr, k0, nb, row_end, At + (size_t)(k0 + nb) * ldr, ntrail, work, nullptr, nullptr, st,
                   batch, bs_mat, bs_work);
    }
    {
        // the back substitutions side by side, one workgroup per problem, straight from the factored matrices (one after
        // the other they were two thirds of a sixteen-problem scan: 16 x 0.4 ms behind 3 ms of shared factorisation launches)
        (void)scratch;
        backsolve_attr();
        hipLaunchKernelGGL(qr_backsolve_kernel, dim3(batch), dim3(1024), sizeof(double) * cols, st, cols, At, ldr, At + (size_t)cols * ldr, x, bs_mat);
    }
    return 0;
}

// ---- ONE column through kept reflectors.  tsqr_apply_kernel is built for tiles of 32 trailing columns (137 KB of LDS, three
// MFMA products, five barriers): for the single column a kept factorisation is asked about — an appended inducing LCE, new
// targets — that machinery was 23 us per launch and 128 launches per column (32 panels x 4 levels at 49k rows: 3 ms for
// 0.4 GB of reflectors).  Here a chunk is a 256-thread workgroup doing what one column needs: w = V^T a with the 32 dot
// products dealt to the four waves (coalesced rows, eight independent wave reductions each), z = T^T w by 32 threads,
// a -= V z; and the small upper levels of a panel's tree (four chunks or fewer) run one after the other in ONE workgroup.
struct VecLevel { const double *V, *T; int n, chunks, stride; };
struct TsqrVec {
    double *a;            // the column over the physical rows
    int row0, row_end;
    int nlev;             // levels this launch works through (> 1: one workgroup, level after level)
    VecLevel lv[4];
};

__device__ __forceinline__ void apply_vec_chunk(double *a, int row0, int row_end, const VecLevel &L, int chunk, double *ws /*[32]*/, double *zs /*[32]*/)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const double *V = L.V + (size_t)chunk * TNB * TCH, *T = L.T + (size_t)chunk * TNB * TNB;
    // this thread's row of the chunk for the update; the wave's rows lane, lane + 64, ... for the products
    double av[4];
    int ph[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int rho = chunk * TCH + lane + 64 * j;
        const int phys = row0 + (rho >> 5) * L.stride + (rho & 31);
        const bool live = rho < L.n && phys < row_end;
        ph[j] = live ? phys : -1;
        av[j] = live ? a[phys] : 0.0;
    }
    double w[8];
#pragma unroll
    for (int u = 0; u < 8; u++) {
        const double *vi = V + (size_t)(8 * wave + u) * TCH + lane;
        w[u] = (vi[0] * av[0] + vi[64] * av[1]) + (vi[128] * av[2] + vi[192] * av[3]);
    }
#pragma unroll
    for (int u = 0; u < 8; u++) w[u] = wave_sum64(w[u]);
    if (lane == 0) {
#pragma unroll
        for (int u = 0; u < 8; u++) ws[8 * wave + u] = w[u];
    }
    __syncthreads();
    if (tid < TNB) {   // z_i = sum_l T[l][i] w_l
        double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int l = 0; l < TNB; l++) s4[l & 3] += T[l * TNB + tid] * ws[l];
        zs[tid] = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    }
    __syncthreads();
    {   // a_r -= sum_i V[r][i] z_i: thread = row (wave-major, so that the four rows of the products' layout are covered)
        const int rho = chunk * TCH + tid;
        const int phys = row0 + (rho >> 5) * L.stride + (rho & 31);
        if (rho < L.n && phys < row_end) {
            double s4[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int i = 0; i < TNB; i++) s4[i & 3] += V[(size_t)i * TCH + tid] * zs[i];
            a[phys] -= (s4[0] + s4[1]) + (s4[2] + s4[3]);
        }
    }
    (void)ph;
}

__global__ __launch_bounds__(TCH) void tsqr_apply_vec_kernel(TsqrVec q)
{
    __shared__ double ws[TNB], zs[TNB];
    if (q.nlev == 1) {
        apply_vec_chunk(q.a, q.row0, q.row_end, q.lv[0], blockIdx.x, ws, zs);
        return;
    }
    for (int l = 0; l < q.nlev; l++)
        for (int c = 0; c < q.lv[l].chunks; c++) {
            apply_vec_chunk(q.a, q.row0, q.row_end, q.lv[l], c, ws, zs);
            __threadfence();      // (the next chunk / level reads rows this one wrote, through other threads)
            __syncthreads();
        }
}

// vec <- Q_p^T vec for the kept panels p = first .. first + count - 1, in order (vec: one column over the physical rows)
void tsqr_apply_panels(const TsqrPanel *panels, int count, double *vec, hipStream_t st)
{
    static const bool tiles = getenv("SGPR_APPLY_VEC") && atoi(getenv("SGPR_APPLY_VEC")) == 0;   // SGPR_APPLY_VEC=0: the tile kernel
    if (tiles) tsqr_attrs();
    for (int p = 0; p < count; p++) {
        const TsqrPanel &pn = panels[p];
        if (tiles) {
            for (int l = 0; l < pn.nlev; l++) {
                TsqrApply ap = {};
                ap.A = vec;
                ap.ldr = 0; ap.ntrail = 1;
                ap.row0 = pn.k0; ap.row_end = pn.row_end;
                ap.n = pn.lv[l].n; ap.stride = pn.lv[l].stride;
                ap.V = pn.lv[l].V; ap.T = pn.lv[l].T;
                ap.tpw = 1;
                hipLaunchKernelGGL(tsqr_apply_kernel, dim3(1, pn.lv[l].chunks), dim3(TCH), lds_apply_bytes(), st, ap);
            }
            continue;
        }
        int l = 0;
        while (l < pn.nlev) {
            TsqrVec q = {};
            q.a = vec; q.row0 = pn.k0; q.row_end = pn.row_end;
            // the levels from here on in one workgroup when they are small (<= 4 chunks in all, <= 4 levels)
            int tail_chunks = 0;
            for (int k = l; k < pn.nlev; k++) tail_chunks += pn.lv[k].chunks;
            if (tail_chunks <= 4 && pn.nlev - l <= 4 && pn.nlev - l > 1) {
                q.nlev = pn.nlev - l;
                for (int k = 0; k < q.nlev; k++) q.lv[k] = {pn.lv[l + k].V, pn.lv[l + k].T, pn.lv[l + k].n, pn.lv[l + k].chunks, pn.lv[l + k].stride};
                hipLaunchKernelGGL(tsqr_apply_vec_kernel, dim3(1), dim3(TCH), 0, st, q);
                break;
            }
            q.nlev = 1;
            q.lv[0] = {pn.lv[l].V, pn.lv[l].T, pn.lv[l].n, pn.lv[l].chunks, pn.lv[l].stride};
            hipLaunchKernelGGL(tsqr_apply_vec_kernel, dim3(pn.lv[l].chunks), dim3(TCH), 0, st, q);
            l++;
        }
    }
}

// ---------------------------------------------------------------- one-column reflectors (appended columns)
// A column appended to a kept factorisation gets a single Householder reflector over all remaining rows,
// H = I - sc v v^T — no tree: the norm and the products with it are two-stage sums in a fixed order (per-chunk
// partials, then every workgroup adds the partials itself), two launches per application.
#define FCH 2048  // elements per workgroup

__device__ __forceinline__ double block_sum256(double s, double *red)
{
    s = wave_sum64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    const double t = (red[0] + red[1]) + (red[2] + red[3]);
    __syncthreads();
    return t;
}

__global__ __launch_bounds__(256) void flat_dot_kernel(int n, const double *x, const double *y, double *part)
{
    __shared__ double red[4];
    const int i0 = blockIdx.x * FCH, i1 = min(n, i0 + FCH);
    double s = 0.0;
    for (int i = i0 + threadIdx.x; i < i1; i += 256) s += x[i] * y[i];
    s = block_sum256(s, red);
    if (threadIdx.x == 0) part[blockIdx.x] = s;
}

__device__ __forceinline__ double sum_parts(const double *part, int np, double *red)
{
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += part[i];
    return block_sum256(s, red);
}

// a <- a - sc (v . a) v, the dot product given as partials
__global__ __launch_bounds__(256) void flat_apply_kernel(int n, int np, const double *part, const double *sc, const double *v,
                                                         double *a)
{
    __shared__ double red[4];
    const double f = sc[0] * sum_parts(part, np, red);
    const int i0 = blockIdx.x * FCH, i1 = min(n, i0 + FCH);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) a[i] -= f * v[i];
}

// v = x - alpha e_0, sc = 2 / v.v, alpha = -sign(x_0) |x| (|x|^2 given as partials); x is left as it is
__global__ __launch_bounds__(256) void flat_make_kernel(int n, int np, const double *part, const double *x, double *v,
                                                        double *sc, double *alpha_out)
{
    __shared__ double red[4];
    const double s2 = sum_parts(part, np, red);
    const double akk = x[0];
    const double nrm = sqrt(s2);
    const double alpha = akk > 0.0 ? -nrm : nrm;
    const double v0 = akk - alpha;
    const double vv = s2 - akk * akk + v0 * v0;
    const int i0 = blockIdx.x * FCH, i1 = min(n, i0 + FCH);
    for (int i = i0 + threadIdx.x; i < i1; i += 256) v[i] = i == 0 ? v0 : x[i];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        sc[0] = vv > TSQR_TINY2 ? 2.0 / vv : 0.0;
        alpha_out[0] = alpha;
    }
}

size_t flat_part_doubles(int n) { return (size_t)(n + FCH - 1) / FCH + 8; }

void flat_reflector_make(const double *x, int n, double *v, double *sc, double *alpha, double *part, hipStream_t st)
{
    const int np = (n + FCH - 1) / FCH;
    hipLaunchKernelGGL(flat_dot_kernel, dim3(np), dim3(256), 0, st, n, x, x, part);
    hipLaunchKernelGGL(flat_make_kernel, dim3(np), dim3(256), 0, st, n, np, part, x, v, sc, alpha);
}

void flat_reflector_apply(const double *v, const double *sc, int n, double *a, double *part, hipStream_t st)
{
    const int np = (n + FCH - 1) / FCH;
    hipLaunchKernelGGL(flat_dot_kernel, dim3(np), dim3(256), 0, st, n, v, a, part);
    hipLaunchKernelGGL(flat_apply_kernel, dim3(np), dim3(256), 0, st, n, np, part, sc, v, a);
}

void launch_qr_gather_r(int cols, const double *At, int ldr, double *Rc, int ldc, double *z, hipStream_t st)
{
    hipLaunchKernelGGL(qr_gather_r_kernel<false>, dim3((cols + 255) / 256, cols), dim3(256), 0, st, cols, At, ldr, Rc, ldc, z);
}
