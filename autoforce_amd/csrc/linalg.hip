// linalg.hip — dense solve side of the SGPR model on the device (gfx950).
//
//   jitcholesky  regression/algebra.py:29-47     (ladder driven from api.hip::sgpr_solve)
//   choli = L^-1 regression/gppotential.py:1234
//   (mu: the Householder QR least squares of [K; sigma L^T] mu = [Y; 0], gppotential.py:1255-1263, is tsqr.hip)
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
// EVERY workgroup factorises the diagonal block for itself from A.  Workgroup 0 therefore must not store L11 over A11
// while another workgroup of the launch may still have to read A11: a workgroup that starts late — another process on
// the same device, more workgroups than fit — would factorise the FACTOR.  (Round 5: two ranks sharing one GPU produced a
// wrong L in one run of ten.)  With more than one workgroup L11 goes to `dsave` [2][64][64] (by panel parity) and is
// copied into A by workgroup 0 of the NEXT launch, which nobody reads it before; the last panel has one workgroup.
__global__ __launch_bounds__(256) void potrf_panel_kernel(int m, double *A, int ld, int k0, int *info, double *dsave)
{
    __shared__ double D[NB][NB + 1];
    __shared__ double P[NB][NB + 1];
    __shared__ int failed;
    const int tid = threadIdx.x;
    const int nb = min(NB, m - k0);
    if (blockIdx.x == 0 && k0 > 0) {   // the previous panel's diagonal block, parked in dsave, goes home (also after a failure)
        const double *src = dsave + (size_t)(((k0 / NB) - 1) & 1) * NB * NB;
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e % NB;
            A[(size_t)(k0 - NB + i) * ld + k0 - NB + j] = src[e];
        }
    }
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
        if (gridDim.x == 1) {   // nobody else reads A11
            for (int e = tid; e < nb * nb; e += 256) {
                const int i = e / nb, j = e % nb;
                A[(size_t)(k0 + i) * ld + k0 + j] = j <= i ? D[i][j] : 0.0;
            }
        } else {                // (a full 64 x 64 block: panels with workgroups below them are never the ragged last one)
            double *dst = dsave + (size_t)((k0 / NB) & 1) * NB * NB;
            for (int e = tid; e < NB * NB; e += 256) {
                const int i = e / NB, j = e % NB;
                dst[e] = (i < nb && j <= i) ? D[i][j] : 0.0;
            }
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

int launch_cholesky_lower(int m, double *A, int ld, int *info, hipStream_t st, double *dsave /*[2][64][64] scratch of the caller*/)
{
    (void)hipMemsetAsync(info, 0, sizeof(int), st);
    for (int k0 = 0; k0 < m; k0 += NB) {
        const int nb = std::min(NB, m - k0);
        const int below = m - k0 - nb;
        const int nblk = 1 + (below + NB - 1) / NB;
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(nblk), dim3(256), 0, st, m, A, ld, k0, info, dsave);
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
