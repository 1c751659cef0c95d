// gemm.hip — fp64 MFMA GEMM family for the dense contractions of the SGPR predict path.
//
//   C[M][N] = A[M][K] . B[N][K]^T      (both operands row-major, K contiguous)
// on v_mfma_f64_16x16x4_f64, with three fused epilogues:
//   EPI_KERNEL  K_nm: k = [Z_i==Z_q] dot^eta (+ lone-atom term), writes K, Aw = mu_q eta dot^(eta-1)
//               and per-block energy partials   (similarity/universal.py:109-122,
//               similarity/similarity.py:94-103, calculator/active.py:549-560)
//   EPI_STORE   W = Aw . Pm  (dE/dp-hat, the reverse-pass seed of calculator/active.py:587-599)
//   EPI_ROWSQ   c_i = |choli . k_i|^2 without materialising choli.K^T (calculator/active.py:782-783)
//
// Operands are species-sorted, so K_nm, K_mm, choli are block-diagonal: each 64x64 tile works
// only on the species blocks it touches (tile skip + trimmed reduction range).
// All leading dimensions are multiples of 16 and all row counts are padded to 64 with zeros
// by the allocator (api.hip), so the main loop carries no bounds checks.
//
// Tile: 64x64 per 256-thread workgroup, 4 waves as 2x2, each wave 32x32 = 2x2 MFMA tiles;
// K-step 16 through LDS (row stride 17 doubles: conflict-free ds_read_b64 fragment reads).
#include "sgpr_internal.h"

typedef double v4d __attribute__((ext_vector_type(4)));

#define BM 64
#define BN 64
#define KT 16
#define LDS_LD 17

struct GemmArgs {
    GemmParams p;
    int ieta;        // integer exponent or -1
    const int *row_slot, *col_slot;
    double *Epart;
};

__device__ __forceinline__ double ipow_d(double x, int n)
{
    double y = 1.0;
    for (int k = 0; k < n; k++) y *= x;
    return y;
}

template <int EPI>
__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmArgs g)
{
    const GemmParams &p = g.p;
    __shared__ double As[2][BM * LDS_LD];
    __shared__ double Bs[2][BN * LDS_LD];
    __shared__ double red[4];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int row0 = blockIdx.y * BM, col0 = blockIdx.x * BN;

    // species range of this tile's rows / cols -> skip test and reduction range
    int kbeg = 0, kend = p.K;
    bool skip = false;
    if (p.row_off) {
        const int rlast = min(row0 + BM, p.M) - 1;
        int sa = 0, sb = 0;
        for (int s = 0; s < p.S; s++) {
            if (p.row_off[s + 1] <= row0) sa = s + 1;
            if (p.row_off[s + 1] <= rlast) sb = s + 1;
        }
        sa = min(sa, p.S - 1); sb = min(sb, p.S - 1);
        if (p.col_off) {
            const int clo = p.col_off[sa], chi = p.col_off[sb + 1];
            if (col0 >= chi || col0 + BN <= clo) skip = true;
        }
        if (p.k_off) {
            kbeg = p.k_off[sa];
            kend = p.k_off[sb + 1];
            if (EPI == EPI_ROWSQ && p.tri) kend = min(kend, col0 + BN);
            if (EPI == EPI_ROWSQ && p.col_off == nullptr) {
                // columns index the same (inducing) dimension as k
                if (col0 >= p.k_off[sb + 1] || col0 + BN <= p.k_off[sa]) skip = true;
            }
        }
        kbeg = (kbeg / KT) * KT;
        kend = ((kend + KT - 1) / KT) * KT;
        if (kend <= kbeg) skip = true;
    }
    if (row0 >= p.M) skip = true;
    if (EPI == EPI_SUBLOWER && col0 > row0) skip = true;  // symmetric update: lower tiles only

    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    if (!skip) {
        // global -> LDS staging: thread t moves 4 consecutive doubles of row t/4
        const int lr = tid >> 2, lk = (tid & 3) * 4;
        const double *Ag = p.A + (size_t)(row0 + lr) * p.lda + lk;
        const double *Bg = p.B + (size_t)(col0 + lr) * p.ldb + lk;
        double ra[4], rb[4];
        auto gload = [&](int k0) {
            const double2 a0 = *(const double2 *)(Ag + k0), a1 = *(const double2 *)(Ag + k0 + 2);
            const double2 b0 = *(const double2 *)(Bg + k0), b1 = *(const double2 *)(Bg + k0 + 2);
            ra[0] = a0.x; ra[1] = a0.y; ra[2] = a1.x; ra[3] = a1.y;
            rb[0] = b0.x; rb[1] = b0.y; rb[2] = b1.x; rb[3] = b1.y;
        };
        auto lstore = [&](int buf) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                As[buf][lr * LDS_LD + lk + q] = ra[q];
                Bs[buf][lr * LDS_LD + lk + q] = rb[q];
            }
        };
        gload(kbeg);
        lstore(0);
        __syncthreads();
        int buf = 0;
        const int fa = (wr * 32 + (lane & 15)) * LDS_LD + (lane >> 4);
        const int fb = (wc * 32 + (lane & 15)) * LDS_LD + (lane >> 4);
        for (int k0 = kbeg; k0 < kend; k0 += KT) {
            const bool more = k0 + KT < kend;
            if (more) gload(k0 + KT);
#pragma unroll
            for (int kk = 0; kk < KT; kk += 4) {
                const double a0 = As[buf][fa + kk], a1 = As[buf][fa + 16 * LDS_LD + kk];
                const double b0 = Bs[buf][fb + kk], b1 = Bs[buf][fb + 16 * LDS_LD + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
            if (more) {
                lstore(buf ^ 1);
                __syncthreads();
                buf ^= 1;
            }
        }
    }

    // ------------------------------------------------------------------ epilogues
    // C/D map of v_mfma_f64_16x16x4_f64: col = lane&15, row = (lane>>4) + 4*reg
    double esum = 0.0;
    if (!skip) {
#pragma unroll
        for (int tm = 0; tm < 2; tm++) {
            double rsq[4] = {0, 0, 0, 0};
#pragma unroll
            for (int tn = 0; tn < 2; tn++) {
                const int col = col0 + wc * 32 + tn * 16 + (lane & 15);
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const int row = row0 + wr * 32 + tm * 16 + (lane >> 4) + 4 * r;
                    const double v = acc[tm][tn][r];
                    if (EPI == EPI_STORE) {
                        if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] = v;
                    } else if (EPI == EPI_SUBLOWER) {
                        if (row < p.M && col < p.N) p.C[(size_t)row * p.ldc + col] -= v;
                    } else if (EPI == EPI_KERNEL) {
                        if (row < p.M && col < p.N && g.row_slot[row] == g.col_slot[col]) {
                            const bool rl = p.row_nn ? p.row_nn[row] == 0 : false;
                            const bool cl = p.col_nn ? p.col_nn[col] == 0 : false;
                            double k = 0.0, kp = 0.0;
                            if (!rl && !cl) {
                                if (g.ieta >= 1) {
                                    const double pm1 = ipow_d(v, g.ieta - 1);
                                    k = pm1 * v;
                                    kp = g.ieta * pm1;
                                } else {
                                    k = pow(v, p.eta);
                                    kp = p.eta * pow(v, p.eta - 1.0);
                                }
                            } else if (rl && cl)
                                k = 1.0;  // similarity/similarity.py:94-103
                            const double mu = p.mu ? p.mu[col] : 0.0;
                            p.C[(size_t)row * p.ldc + col] = k;
                            if (p.Aw) p.Aw[(size_t)row * p.ldc + col] = mu * kp;
                            esum += k * mu;
                        }
                    } else {  // EPI_ROWSQ
                        if (col < p.N) rsq[r] += v * v;
                    }
                }
            }
            if (EPI == EPI_ROWSQ) {
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    double s = rsq[r];
                    s += __shfl_xor(s, 1, 64);
                    s += __shfl_xor(s, 2, 64);
                    s += __shfl_xor(s, 4, 64);
                    s += __shfl_xor(s, 8, 64);
                    const int row = row0 + wr * 32 + tm * 16 + (lane >> 4) + 4 * r;
                    if ((lane & 15) == 0 && row < p.M && s != 0.0) unsafeAtomicAdd(&p.rowsq[row], s);
                }
            }
        }
    }
    if (EPI == EPI_KERNEL && g.Epart) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) esum += __shfl_xor(esum, o, 64);
        if (lane == 0) red[wave] = esum;
        __syncthreads();
        if (tid == 0) g.Epart[blockIdx.y * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
    }
}

void launch_gemm_nt(const GemmParams &p, GemmEpilogue epi, hipStream_t st)
{
    if (p.M <= 0 || p.N <= 0) return;
    GemmArgs g;
    g.p = p;
    g.ieta = (p.eta == (double)(int)p.eta && p.eta >= 1.0 && p.eta <= 64.0) ? (int)p.eta : -1;
    g.row_slot = p.row_slot;
    g.col_slot = p.col_slot;
    g.Epart = p.Esum;
    dim3 grid((p.N + BN - 1) / BN, (p.M + BM - 1) / BM), block(256);
    if (epi == EPI_STORE) hipLaunchKernelGGL(gemm_nt_kernel<EPI_STORE>, grid, block, 0, st, g);
    else if (epi == EPI_KERNEL) hipLaunchKernelGGL(gemm_nt_kernel<EPI_KERNEL>, grid, block, 0, st, g);
    else if (epi == EPI_SUBLOWER) hipLaunchKernelGGL(gemm_nt_kernel<EPI_SUBLOWER>, grid, block, 0, st, g);
    else hipLaunchKernelGGL(gemm_nt_kernel<EPI_ROWSQ>, grid, block, 0, st, g);
}
