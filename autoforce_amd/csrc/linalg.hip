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
#ifdef POTRF_STAMPS   // tools/potrf_test.hip: where a panel launch spends its time (workgroup 1, thread 0)
__device__ long long potrf_stamps[8];
#define PSTAMP(i) if (blockIdx.x == 1 && threadIdx.x == 0) potrf_stamps[i] = wall_clock64();
#else
#define PSTAMP(i)
#endif

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
    PSTAMP(0)
    if (blockIdx.x == 0 && k0 > 0) {   // the previous panel's diagonal block, parked in dsave, goes home (also after a failure)
        const double *src = dsave + (size_t)(((k0 / NB) - 1) & 1) * NB * NB;
        for (int e = tid; e < NB * NB; e += 256) {
            const int i = e / NB, j = e % NB;
            A[(size_t)(k0 - NB + i) * ld + k0 - NB + j] = src[e];
        }
    }
    if (*info != 0) return;
    if (tid == 0) failed = 0;
    // The 64 x 64 diagonal block in REGISTERS: thread (ty, tx) of a 16 x 16 grid holds the 4 x 4 tile of rows 4 ty .., columns
    // 4 tx ..; a column step publishes the scaled column j through LDS and every thread takes it out of its tile — two
    // barriers and 16 multiply-adds per step (the LDS form walked (nb - j)^2 elements with a division each: 100 us a block).
    // Same operations on every element in the same order as before: the factor is the same bits.
    const int ty = tid >> 4, tx = tid & 15;
    double t[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = 4 * ty + a, j = 4 * tx + b;
            t[a][b] = (i < nb && j <= i) ? A[(size_t)(k0 + i) * ld + k0 + j] : 0.0;
        }
    __shared__ double colj[NB];
    __shared__ double s_d;
    __syncthreads();
    PSTAMP(1)
    // (the column loop in groups of four with the position inside the tile a compile-time constant: no selects)
    for (int jb = 0; jb < (nb + 3) / 4 && !failed; jb++) {
#pragma unroll
        for (int jr = 0; jr < 4; jr++) {
            const int j = 4 * jb + jr;
            if (j >= nb) break;
            if (ty == jb && tx == jb) {          // the owner of (j, j)
                const double d = t[jr][jr];
                if (!(d > 0.0)) failed = j + 1;  // LAPACK potrf: leading minor not positive definite
                s_d = sqrt(d);
            }
            __syncthreads();
            if (failed) break;
            const double djj = s_d;
            if (tx == jb) {                      // the owners of column j: scale and publish
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    const int i = 4 * ty + a;
                    if (i == j) t[a][jr] = djj;
                    else if (i > j) t[a][jr] /= djj;
                    colj[i] = t[a][jr];
                }
            }
            __syncthreads();
            // trailing update of the lower triangle: D[i][k] -= D[i][j] D[k][j], j < k <= i
            if (ty >= tx && tx >= jb) {
                double ci[4], ck[4];
#pragma unroll
                for (int a = 0; a < 4; a++) { ci[a] = colj[4 * ty + a]; ck[a] = colj[4 * tx + a]; }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++) {
                        const int i = 4 * ty + a, k = 4 * tx + b;
                        if (k > j && k <= i && i < nb) t[a][b] -= ci[a] * ck[b];
                    }
            }
        }
    }
    __syncthreads();
    if (!failed) {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) D[4 * ty + a][4 * tx + b] = (4 * tx + b <= 4 * ty + a) ? t[a][b] : 0.0;
    }
    __syncthreads();
    PSTAMP(2)
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
    // L21 = A21 L11^-T, the 64 x 64 row block in register tiles as above, column by column: the owners of column k divide it
    // by D[k][k] and publish it, everybody takes P[:, k] D[j][k] out of the columns j > k — per element the terms k = 0 .. j - 1
    // in order, then the division: the bits of the row-by-row substitution it replaces (one thread per row, 2016 dependent
    // multiply-adds through LDS: half of the launch).
    double p[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = 4 * ty + a, j = 4 * tx + b;
            p[a][b] = (i < nr && j < nb) ? A[(size_t)(r0 + i) * ld + k0 + j] : 0.0;
        }
    PSTAMP(3)
    double (*pc)[NB] = (double (*)[NB])&P[0][0];   // [2][64]: column k of P, by parity of k (P itself is not needed any more)
    for (int kb = 0; kb < (nb + 3) / 4; kb++) {
#pragma unroll
        for (int kr = 0; kr < 4; kr++) {
            const int k = 4 * kb + kr;
            if (k >= nb) break;
            if (tx == kb) {
                const double dkk = D[k][k];
#pragma unroll
                for (int a = 0; a < 4; a++) {
                    p[a][kr] /= dkk;
                    pc[k & 1][4 * ty + a] = p[a][kr];
                }
            }
            __syncthreads();
            if (tx >= kb) {
                double ci[4], dj[4];
#pragma unroll
                for (int a = 0; a < 4; a++) { ci[a] = pc[k & 1][4 * ty + a]; dj[a] = D[4 * tx + a][k]; }
#pragma unroll
                for (int a = 0; a < 4; a++)
#pragma unroll
                    for (int b = 0; b < 4; b++)
                        if (4 * tx + b > k) p[a][b] -= ci[a] * dj[b];
            }
        }
    }
    PSTAMP(4)
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int i = 4 * ty + a, j = 4 * tx + b;
            if (i < nr && j < nb) A[(size_t)(r0 + i) * ld + k0 + j] = p[a][b];
        }
    PSTAMP(5)
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
        {
            // forward substitution with the diagonal block, all 256 threads: thread = (column c, row group g), rows g, g + 4, ...
            // in registers; step k: the owner of row k divides and publishes x_k, everybody takes L[r][k] x_k out of its rows
            // below.  Every element sees the same operations in the same order as one thread per column running
            // v -= L[r][k] x[k] for k = 0 .. r - 1 (which this replaces: a chain of 2000 dependent LDS round trips per block,
            // 85 us of the kernel's 0.75 ms at m = 1024) — the same bits.  Rows beyond nr are identity padding.
            const int c = tid & 63, gq = tid >> 6;
            double v[16];
#pragma unroll
            for (int i = 0; i < 16; i++) v[i] = Ts[gq + 4 * i][c];
#pragma unroll
            for (int k = 0; k < NB; k++) {
                if (gq == (k & 3)) Xs[k][c] = v[k >> 2] / Ls[k][k];
                __syncthreads();
                const double xk = Xs[k][c];
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    if (4 * i + 3 <= k) continue;                  // (rows gq + 4 i <= k for every gq)
                    const int r = gq + 4 * i;
                    if (r > k) v[i] = fma(-Ls[r][k], xk, v[i]);
                }
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
