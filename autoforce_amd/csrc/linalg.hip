// linalg.hip — dense solve side of the SGPR model on the device (gfx950).
//
//   jitcholesky  regression/algebra.py:29-47     (ladder driven from api.hip::sgpr_solve)
//   choli = L^-1 regression/gppotential.py:1234
//   mu           regression/gppotential.py:1255-1263: Householder QR least squares of
//                [K; sigma L^T] mu = [Y; 0]
// Blocked right-looking Cholesky, NB = 64: panel kernel (diagonal block factorised in LDS by
// every workgroup, one 64-row panel block solved per workgroup) + trailing update on the
// fp64 MFMA GEMM (gemm.hip, EPI_SUBLOWER).
#include <algorithm>

#include "sgpr_internal.h"
#include <algorithm>
#include <vector>

#define NB 64

__global__ void add_diag_kernel(int m, const double *A, int ld, double ridge, double *out)
{
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m && j < m) out[(size_t)i * ld + j] = A[(size_t)i * ld + j] + (i == j ? ridge : 0.0);
}

void launch_add_diag(int m, const double *A, int ld, double ridge, double *out, hipStream_t st)
{
    if (m <= 0) return;
    hipLaunchKernelGGL(add_diag_kernel, dim3((m + 255) / 256, m), dim3(256), 0, st, m, A, ld, ridge, out);
}

// One panel step at column k0.  Workgroup 0 factorises and stores the diagonal block;
// workgroup b >= 1 computes rows [k0+nb+(b-1)*64, +64) of L21 = A21 L11^-T.
__global__ __launch_bounds__(256) void potrf_panel_kernel(int m, double *A, int ld, int k0, int *info)
{
    __shared__ double D[NB][NB + 1];
    __shared__ double P[NB][NB + 1];
    __shared__ int failed;
    const int tid = threadIdx.x;
    const int nb = min(NB, m - k0);
    if (*info != 0) return;
    if (tid == 0) failed = 0;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        D[i][j] = (i < nb && j <= i) ? A[(size_t)(k0 + i) * ld + k0 + j] : 0.0;
    }
    __syncthreads();
    for (int j = 0; j < nb; j++) {
        if (tid == 0) {
            const double d = D[j][j];
            if (!(d > 0.0)) failed = j + 1;  // LAPACK potrf: leading minor not positive definite
            D[j][j] = sqrt(d);
        }
        __syncthreads();
        if (failed) break;
        const double djj = D[j][j];
        for (int i = j + 1 + tid; i < nb; i += 256) D[i][j] /= djj;
        __syncthreads();
        // trailing update of the lower triangle: D[i][k] -= D[i][j] D[k][j], j < k <= i
        const int t = nb - j - 1;
        for (int e = tid; e < t * t; e += 256) {
            const int i = j + 1 + e / t, k = j + 1 + e % t;
            if (k <= i) D[i][k] -= D[i][j] * D[k][j];
        }
        __syncthreads();
    }
    if (failed) {
        if (tid == 0 && blockIdx.x == 0) atomicCAS(info, 0, k0 + failed);
        return;
    }
    if (blockIdx.x == 0) {
        for (int e = tid; e < nb * nb; e += 256) {
            const int i = e / nb, j = e % nb;
            A[(size_t)(k0 + i) * ld + k0 + j] = j <= i ? D[i][j] : 0.0;
        }
        // zero the strictly-upper part right of the diagonal block (rows k0..k0+nb)
        for (int i = 0; i < nb; i++)
            for (int j = k0 + nb + tid; j < m; j += 256) A[(size_t)(k0 + i) * ld + j] = 0.0;
        return;
    }
    const int r0 = k0 + nb + (blockIdx.x - 1) * NB;
    const int nr = min(NB, m - r0);
    if (nr <= 0) return;
    for (int e = tid; e < NB * NB; e += 256) {
        const int i = e / NB, j = e % NB;
        P[i][j] = (i < nr && j < nb) ? A[(size_t)(r0 + i) * ld + k0 + j] : 0.0;
    }
    __syncthreads();
    if (tid < nr) {
        for (int j = 0; j < nb; j++) {
            double v = P[tid][j];
            for (int k = 0; k < j; k++) v -= P[tid][k] * D[j][k];
            P[tid][j] = v / D[j][j];
        }
    }
    __syncthreads();
    for (int e = tid; e < nr * nb; e += 256) {
        const int i = e / nb, j = e % nb;
        A[(size_t)(r0 + i) * ld + k0 + j] = P[i][j];
    }
}

int launch_cholesky_lower(int m, double *A, int ld, int *info, hipStream_t st)
{
    (void)hipMemsetAsync(info, 0, sizeof(int), st);
    for (int k0 = 0; k0 < m; k0 += NB) {
        const int nb = std::min(NB, m - k0);
        const int below = m - k0 - nb;
        const int nblk = 1 + (below + NB - 1) / NB;
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(nblk), dim3(256), 0, st, m, A, ld, k0, info);
        if (below > 0) {
            // A22 -= L21 L21^T on the MFMA GEMM (lower tiles only)
            GemmParams g = {};
            g.M = below; g.N = below; g.K = (nb + 31) / 32 * 32;
            g.lda = ld; g.ldb = ld; g.ldc = ld;
            g.A = A + (size_t)(k0 + nb) * ld + k0;
            g.B = g.A;
            g.C = A + (size_t)(k0 + nb) * ld + (k0 + nb);
            launch_gemm_nt(g, EPI_SUBLOWER, st);
        }
    }
    return 0;
}

// Li = L^-1, one workgroup per 64-column block: block forward substitution
//   X_i = L_ii^-1 (I_ic - sum_{c<=k<i} L_ik X_k)
__global__ __launch_bounds__(256) void tril_inverse_kernel(int m, const double *L, int ld, double *Li)
{
    __shared__ double Ls[NB][NB + 1];
    __shared__ double Xs[NB][NB + 1];
    __shared__ double Ts[NB][NB + 1];
    const int tid = threadIdx.x;
    const int cb = blockIdx.x;
    const int c0 = cb * NB;
    const int nc = min(NB, m - c0);
    const int nblk = (m + NB - 1) / NB;
    const int ty = tid / 16, tx = tid % 16;  // 16x16 threads, 4x4 outputs each
    // zero the block rows above the diagonal
    for (int i = 0; i < c0; i++)
        for (int j = tid; j < nc; j += 256) Li[(size_t)i * ld + c0 + j] = 0.0;
    for (int ib = cb; ib < nblk; ib++) {
        const int r0 = ib * NB, nr = min(NB, m - r0);
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) acc[a][b] = 0.0;
        for (int kb = cb; kb < ib; kb++) {
            const int k0 = kb * NB;
            __syncthreads();
            for (int e = tid; e < NB * NB; e += 256) {
                const int i = e / NB, j = e % NB;
                Ls[i][j] = (i < nr) ? L[(size_t)(r0 + i) * ld + k0 + j] : 0.0;
                Xs[i][j] = (j < nc) ? Li[(size_t)(k0 + i) * ld + c0 + j] : 0.0;
            }
            __syncthreads();
            for (int k = 0; k < NB; k++) {
                double a4[4], b4[4];
#pragma unroll
                for (int a = 0; a < 4; a++) a4[a] = Ls[ty * 4 + a][k];
#pragma unroll
                for (int b = 0; b < 4; b++) b4[b] = Xs[k][tx * 4 + b];
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) acc[a][b] += a4[a] * b4[b];
            }
        }
        __syncthreads();
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int i = ty * 4 + a, j = tx * 4 + b;
                Ts[i][j] = ((ib == cb && i == j) ? 1.0 : 0.0) - acc[a][b];
            }
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e % NB;
            Ls[i][j] = (i < nr && j < nr) ? L[(size_t)(r0 + i) * ld + r0 + j] : (i == j ? 1.0 : 0.0);
        }
        __syncthreads();
        if (tid < NB) {
            for (int r = 0; r < nr; r++) {
                double v = Ts[r][tid];
                for (int k = 0; k < r; k++) v -= Ls[r][k] * Xs[k][tid];
                Xs[r][tid] = v / Ls[r][r];
            }
        }
        __syncthreads();
        for (int e = tid; e < nr * nc; e += 256) {
            const int i = e / nc, j = e % nc;
            Li[(size_t)(r0 + i) * ld + c0 + j] = Xs[i][j];
        }
        __threadfence_block();
    }
}

void launch_tril_inverse(int m, const double *L, int ld, double *Li, hipStream_t st)
{
    if (m <= 0) return;
    hipLaunchKernelGGL(tril_inverse_kernel, dim3((m + NB - 1) / NB), dim3(256), 0, st, m, L, ld, Li);
}

// ------------------------------------------------------------------ back substitution of the QR least squares
__global__ __launch_bounds__(1024) void qr_backsolve_kernel(int cols, const double *A, const double *y, double *x)
{
    __shared__ double red[1024];
    __shared__ double xs[2048];
    const int tid = threadIdx.x;
    for (int k = cols - 1; k >= 0; k--) {
        double s = 0.0;
        for (int j = k + 1 + tid; j < cols; j += 1024) s += A[(size_t)k * cols + j] * xs[j];
        red[tid] = s;
        __syncthreads();
        for (int o = 512; o > 0; o >>= 1) {
            if (tid < o) red[tid] += red[tid + o];
            __syncthreads();
        }
        if (tid == 0) {
            xs[k] = (y[k] - red[0]) / A[(size_t)k * cols + k];
            x[k] = xs[k];
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ blocked Householder QR (compact WY)
// Storage: At[c][r] = A[r][c] (each column of the least-squares matrix is a contiguous row of
// length ldr), so every column operation streams whole cache lines, and both trailing products
// are NT GEMMs for gemm_nt_kernel:
//     W  = V^T A_trail      (32 x ntrail, contraction over the rows: split-K + fp64 atomics)
//     A_trail -= V (T^T W)  (contraction over the 32 reflectors)
// Panel factorisation (32 columns): two launches per column —
//   dots:   v_j from the column and its norm; g[c] = v_j . (earlier v_c | later column c)
//   update: later columns -= scal_j g[c] v_j, and the norm^2 of the next column on the way.
// Reflectors are kept unnormalised: Q_j = I - scal_j v_j v_j^T, scal_j = 2 / v_j.v_j.
#define QNB 32
#define QCH 256  // rows per workgroup in the panel kernels

struct QrPanel {
    double *At;    // [cols+1 ...][ldr]
    double *Vt;    // [QNB][ldr]   reflectors, zero above their diagonal
    double *Vrm;   // [ldr][QNB]   the same, row-major
    double *G;     // [QNB][QNB]   g of every column step
    double *scal;  // [QNB]
    double *nrm2;  // [QNB + 1]    squared norms of the panel columns at their own step
    int ldr, rows, k0, nb;
};

__global__ __launch_bounds__(256) void qr_colnorm_kernel(QrPanel q, int j)
{
    const int k = q.k0 + j;
    const int r = k + blockIdx.x * QCH + threadIdx.x;
    double s = 0.0;
    if (r < q.rows) {
        const double a = q.At[(size_t)k * q.ldr + r];
        s = a * a;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if ((threadIdx.x & 63) == 0 && s != 0.0) unsafeAtomicAdd(&q.nrm2[j], s);
}

__global__ __launch_bounds__(256) void qr_panel_dots_kernel(QrPanel q, int j)
{
    __shared__ double red[4][QNB];
    const int k = q.k0 + j, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int r = k + blockIdx.x * QCH + tid;  // rows >= k only
    const double s2 = q.nrm2[j];
    const double akk = q.At[(size_t)k * q.ldr + k];
    const double nrm = sqrt(s2);
    const double alpha = akk > 0.0 ? -nrm : nrm;
    double v = 0.0;
    if (r < q.rows) v = q.At[(size_t)k * q.ldr + r] - (r == k ? alpha : 0.0);
    if (r < q.ldr) {
        q.Vt[(size_t)j * q.ldr + r] = v;
        q.Vrm[(size_t)r * QNB + j] = v;
    }
    double acc[QNB];
#pragma unroll
    for (int c = 0; c < QNB; c++) {
        double o = 0.0;
        if (r < q.rows && c != j && c < q.nb)
            o = c < j ? q.Vt[(size_t)c * q.ldr + r] : q.At[(size_t)(q.k0 + c) * q.ldr + r];
        acc[c] = v * o;
    }
#pragma unroll
    for (int c = 0; c < QNB; c++) {
        double t = acc[c];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
        if (lane == 0) red[wave][c] = t;
    }
    __syncthreads();
    if (tid < QNB && tid != j) {
        const double t = red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid];
        if (t != 0.0) unsafeAtomicAdd(&q.G[j * QNB + tid], t);
    }
    if (blockIdx.x == 0 && tid == 0) {
        // v.v = |a|^2 - akk^2 + (akk - alpha)^2
        const double vv = s2 - akk * akk + (akk - alpha) * (akk - alpha);
        q.scal[j] = vv > 0.0 ? 2.0 / vv : 0.0;
    }
}

__global__ __launch_bounds__(256) void qr_panel_update_kernel(QrPanel q, int j)
{
    const int k = q.k0 + j, tid = threadIdx.x;
    const int r = k + blockIdx.x * QCH + tid;
    const double sc = q.scal[j];
    double nxt = 0.0;
    if (r < q.rows) {
        const double v = q.Vt[(size_t)j * q.ldr + r];
        for (int c = j + 1; c < q.nb; c++) {
            double *a = q.At + (size_t)(q.k0 + c) * q.ldr + r;
            const double an = *a - sc * q.G[j * QNB + c] * v;
            *a = an;
            if (c == j + 1 && r > k) nxt = an * an;
        }
        if (r == k) {  // R_kk, and zeros are never read below it
            const double akk = q.At[(size_t)k * q.ldr + k];
            const double nrm = sqrt(q.nrm2[j]);
            q.At[(size_t)k * q.ldr + k] = akk > 0.0 ? -nrm : nrm;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) nxt += __shfl_xor(nxt, o, 64);
    if ((tid & 63) == 0 && nxt != 0.0) unsafeAtomicAdd(&q.nrm2[j + 1], nxt);
}

// Panel factorisation of a SHORT matrix (the 2m x m second stage of the solve) by ONE workgroup with
// the whole panel in LDS: no launches and no global reductions between the column steps.
// 16 waves; wave w owns panel column w for the dot product with v_j and its own rank-1 update, so a
// column step needs two workgroup barriers (norm of the pivot column, then v_j visible to all).
#define QNBL 16
__global__ __launch_bounds__(1024) void qr_panel_lds_kernel(QrPanel q)
{
    extern __shared__ double sm[];  // [nb <= QNBL][rl]
    __shared__ double s_red[16], s_alpha[QNBL], s_scal[QNBL];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int k0 = q.k0, nb = q.nb, rl = q.rows - k0;
    for (int c = wave; c < nb; c += 16)
        for (int i = lane; i < rl; i += 64) sm[c * rl + i] = q.At[(size_t)(k0 + c) * q.ldr + k0 + i];
    __syncthreads();
    for (int j = 0; j < nb; j++) {
        // |a_j|^2 over rows >= j, by all threads
        double s = 0.0;
        for (int i = j + tid; i < rl; i += 1024) {
            const double a = sm[j * rl + i];
            s += a * a;
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
        if (lane == 0) s_red[wave] = s;
        __syncthreads();
        double s2 = 0.0;
#pragma unroll
        for (int w = 0; w < 16; w++) s2 += s_red[w];
        const double akk = sm[j * rl + j];
        const double nrm = sqrt(s2);
        const double alpha = akk > 0.0 ? -nrm : nrm;
        const double vv = s2 - akk * akk + (akk - alpha) * (akk - alpha);
        const double sc = vv > 0.0 ? 2.0 / vv : 0.0;
        __syncthreads();  // everyone has read sm[j][j] and s_red
        if (tid == 0) {
            sm[j * rl + j] = akk - alpha;  // v_j[j]
            s_alpha[j] = alpha;
            s_scal[j] = sc;
        }
        __syncthreads();
        // wave c: g = v_j . (column c for c > j | v_c for c < j); the later columns are updated in place
        for (int c = wave; c < nb; c += 16) {
            if (c == j) continue;
            double g = 0.0;
            for (int i = j + lane; i < rl; i += 64) g += sm[j * rl + i] * sm[c * rl + i];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) g += __shfl_xor(g, o, 64);
            if (lane == 0) q.G[j * QNB + c] = g;
            if (c > j) {
                const double f = sc * g;
                for (int i = j + lane; i < rl; i += 64) sm[c * rl + i] -= f * sm[j * rl + i];
            }
        }
        __syncthreads();
    }
    // write back: R entries above the diagonals and alpha on them, reflectors into Vt / Vrm
    for (int c = wave; c < nb; c += 16) {
        for (int i = lane; i < rl; i += 64) {
            const double v = sm[c * rl + i];
            if (i < c) q.At[(size_t)(k0 + c) * q.ldr + k0 + i] = v;
            else {
                q.Vt[(size_t)c * q.ldr + k0 + i] = v;
                q.Vrm[(size_t)(k0 + i) * QNB + c] = v;
            }
        }
        if (lane == 0) {
            q.At[(size_t)(k0 + c) * q.ldr + k0 + c] = s_alpha[c];
            q.scal[c] = s_scal[c];
        }
    }
}

// T (upper triangular, Q = I - V T V^T) from scal and the v.v products in G; then clears G.
__global__ __launch_bounds__(64) void qr_panel_T_kernel(QrPanel q, double *T)
{
    __shared__ double Ts[QNB][QNB + 1];
    const int t = threadIdx.x;
    for (int e = t; e < QNB * QNB; e += 64) Ts[e / QNB][e % QNB] = 0.0;
    __syncthreads();
    for (int j = 0; j < q.nb; j++) {
        // T[0:j][j] = -scal_j T[0:j][0:j] (V[:,0:j]^T v_j);  G[j][c] (c < j) = v_c . v_j
        if (t < j) {
            double s = 0.0;
            for (int l = t; l < j; l++) s += Ts[t][l] * q.G[j * QNB + l];
            Ts[t][j] = -q.scal[j] * s;
        }
        if (t == j) Ts[j][j] = q.scal[j];
        __syncthreads();
    }
    for (int e = t; e < QNB * QNB; e += 64) T[e] = Ts[e / QNB][e % QNB];
}

// Zt[c][i] = sum_l T[l][i] W[l][c]   (Z = T^T W)
__global__ __launch_bounds__(256) void qr_z_kernel(int ncol, const double *T, const double *W, int ldw, double *Zt)
{
    __shared__ double Ts[QNB * QNB];
    for (int e = threadIdx.x; e < QNB * QNB; e += 256) Ts[e] = T[e];
    __syncthreads();
    const int c = blockIdx.x * 8 + (threadIdx.x >> 5), i = threadIdx.x & 31;
    if (c >= ncol) return;
    double s = 0.0;
    for (int l = 0; l <= i; l++) s += Ts[l * QNB + i] * W[(size_t)l * ldw + c];
    Zt[(size_t)c * QNB + i] = s;
}

__global__ void qr_gather_r_kernel(int cols, const double *At, int ldr, double *Rm /*[cols][cols]*/, double *z)
{
    const int i = blockIdx.y, j = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < cols && j < cols) Rm[(size_t)i * cols + j] = j >= i ? At[(size_t)j * ldr + i] : 0.0;
    if (i < cols && j == 0) z[i] = At[(size_t)cols * ldr + i];
}

size_t lstsq_qr_blocked_work_doubles(int rows, int cols)
{
    const size_t ldr = (size_t)(rows + 63) / 64 * 64, cpad = (size_t)(cols + 1 + 63) / 64 * 64 + 64;
    return 2 * QNB * ldr + 3 * QNB * QNB + 2 * QNB + 8 + QNB * cpad + cpad * QNB + (size_t)cols * cols + cols + 64;
}

// band > 0: column c is known to be zero below row band * (c + 1) (the stacked triangular system of the second
// solve stage, rows interleaved: band = 2).  A panel of columns [k0, k0 + nb) then lives in rows < band (k0 + nb):
// its reflectors, the products with them and the trailing update are restricted to those rows.
int launch_lstsq_qr_blocked(int rows, int cols, double *At, int ldr, double *x, double *work, hipStream_t st, int band)
{
    if (cols > 2048 || rows < cols || ldr % 64 || ldr < rows) return -1;
    const int cpad = (cols + 1 + 63) / 64 * 64 + 64;
    QrPanel q = {};
    q.At = At; q.ldr = ldr; q.rows = rows;
    double *w = work;
    q.Vt = w; w += (size_t)QNB * ldr;
    q.Vrm = w; w += (size_t)QNB * ldr;
    q.G = w; w += QNB * QNB;
    double *T = w; w += QNB * QNB;
    q.scal = w; w += QNB;
    q.nrm2 = w; w += QNB + 8;
    double *W = w; w += (size_t)QNB * cpad;
    double *Zt = w; w += (size_t)cpad * QNB;
    double *Rm = w; w += (size_t)cols * cols;
    double *z = w; w += cols;
    // tile tables of all panels, uploaded once
    std::vector<int4> tiles;
    struct Span { size_t w0, wn, u0, un; };
    std::vector<Span> span;
    const int KSPLIT = 1024;
    // short matrices: 16-column panels factored by one workgroup in LDS (the panel must fit 144 KB)
    int PNB = QNB;
    for (int w = QNBL; w >= 8; w >>= 1)
        if ((size_t)rows * w * sizeof(double) <= 144 * 1024) { PNB = w; break; }
    const bool lds_panels = PNB != QNB;
    if (lds_panels)
        (void)hipFuncSetAttribute((const void *)qr_panel_lds_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  rows * PNB * (int)sizeof(double));
    auto rows_of_panel = [&](int k0) {  // rows the panel at k0 can touch
        const int nbp = std::min(PNB, cols - k0);
        return band > 0 ? std::min(rows, band * (k0 + nbp)) : rows;
    };
    for (int k0 = 0; k0 < cols; k0 += PNB) {
        const int ntrail = cols + 1 - (k0 + std::min(PNB, cols - k0));
        const int rlim = band > 0 ? std::min(ldr, (rows_of_panel(k0) + 63) / 64 * 64) : ldr;
        Span sp = {tiles.size(), 0, 0, 0};
        if (ntrail > 0) {
            const int ctl = (ntrail + 63) / 64;
            const int kb0 = k0 / 32 * 32;
            for (int kb = kb0; kb < rlim; kb += KSPLIT)
                for (int ct = 0; ct < ctl; ct++) tiles.push_back(make_int4(0, ct, kb, std::min(rlim, kb + KSPLIT)));
            sp.wn = tiles.size() - sp.w0;
            sp.u0 = tiles.size();
            for (int ct = k0 / 64; ct < rlim / 64; ct++)
                for (int rt = 0; rt < ctl; rt++) tiles.push_back(make_int4(rt, ct, 0, QNB));
            sp.un = tiles.size() - sp.u0;
        }
        span.push_back(sp);
    }
    int4 *d_tiles = nullptr;
    if (!tiles.empty()) {
        if (hipMalloc(&d_tiles, sizeof(int4) * tiles.size()) != hipSuccess) return -2;
        (void)hipMemcpyAsync(d_tiles, tiles.data(), sizeof(int4) * tiles.size(), hipMemcpyHostToDevice, st);
    }
    int pi = 0;
    for (int k0 = 0; k0 < cols; k0 += PNB, pi++) {
        q.k0 = k0;
        q.nb = std::min(PNB, cols - k0);
        q.rows = rows_of_panel(k0);
        (void)hipMemsetAsync(q.Vt, 0, sizeof(double) * 2 * (size_t)QNB * ldr, st);  // Vt and Vrm
        (void)hipMemsetAsync(q.G, 0, sizeof(double) * (2 * QNB * QNB + 2 * QNB + 8), st);  // G, T, scal, nrm2
        if (lds_panels) {
            hipLaunchKernelGGL(qr_panel_lds_kernel, dim3(1), dim3(1024), (size_t)(q.rows - k0) * PNB * sizeof(double), st, q);
        } else {
            const int nwg0 = (q.rows - k0 + QCH - 1) / QCH;
            hipLaunchKernelGGL(qr_colnorm_kernel, dim3(nwg0), dim3(256), 0, st, q, 0);
            for (int j = 0; j < q.nb; j++) {
                const int nwg = (ldr - (k0 + j) + QCH - 1) / QCH;
                hipLaunchKernelGGL(qr_panel_dots_kernel, dim3(nwg), dim3(256), 0, st, q, j);
                hipLaunchKernelGGL(qr_panel_update_kernel, dim3(nwg), dim3(256), 0, st, q, j);
            }
        }
        const int ntrail = cols + 1 - (k0 + q.nb);
        if (ntrail <= 0) continue;
        hipLaunchKernelGGL(qr_panel_T_kernel, dim3(1), dim3(64), 0, st, q, T);
        (void)hipMemsetAsync(W, 0, sizeof(double) * (size_t)QNB * cpad, st);
        double *Atr = At + (size_t)(k0 + q.nb) * ldr;
        GemmParams gw = {};
        gw.M = QNB; gw.N = ntrail; gw.K = ldr; gw.lda = ldr; gw.ldb = ldr; gw.ldc = cpad;
        gw.A = q.Vt; gw.B = Atr; gw.C = W; gw.bm = 32;
        gw.tiles = d_tiles + span[pi].w0; gw.ntiles = (int)span[pi].wn;
        launch_gemm_nt(gw, EPI_ATOMIC, st);
        hipLaunchKernelGGL(qr_z_kernel, dim3((ntrail + 7) / 8), dim3(256), 0, st, ntrail, T, W, cpad, Zt);
        GemmParams gu = {};
        gu.M = ntrail; gu.N = ldr; gu.K = QNB; gu.lda = QNB; gu.ldb = QNB; gu.ldc = ldr;
        gu.A = Zt; gu.B = q.Vrm; gu.C = Atr; gu.bm = 64;
        gu.tiles = d_tiles + span[pi].u0; gu.ntiles = (int)span[pi].un;
        launch_gemm_nt(gu, EPI_SUB, st);
    }
    if (x) {
        hipLaunchKernelGGL(qr_gather_r_kernel, dim3((cols + 255) / 256, cols), dim3(256), 0, st, cols, At, ldr, Rm, z);
        hipLaunchKernelGGL(qr_backsolve_kernel, dim3(1), dim3(1024), 0, st, cols, Rm, z, x);
    }
    if (d_tiles) {
        (void)hipStreamSynchronize(st);
        (void)hipFree(d_tiles);
    }
    return 0;
}

void launch_qr_gather_r(int cols, const double *At, int ldr, double *Rm, double *z, hipStream_t st)
{
    hipLaunchKernelGGL(qr_gather_r_kernel, dim3((cols + 255) / 256, cols), dim3(256), 0, st, cols, At, ldr, Rm, z);
}
