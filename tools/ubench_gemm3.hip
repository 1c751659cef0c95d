// 8-wave (two waves per SIMD, K split inside each stage) variant of the LDS-staged fp64 MFMA GEMM.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define BM 64
#define KS 32
#define LD 34
template <int NW>  // NW = 4 or 8 waves
__global__ __launch_bounds__(NW * 64) void k(const double *A, const double *B, double *C, int K, int lda, int ldb, int row_tiles)
{
    __shared__ double As[2][BM * LD];
    __shared__ double Bs[2][BM * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int grp = wave >> 2, wr = (wave >> 1) & 1, wc = wave & 1;
    const int rt = blockIdx.x % row_tiles, ct = blockIdx.x / row_tiles;
    const int row0 = rt * 64, col0 = ct * 64, kbeg = 0, kend = K;
    v4d acc[2][2] = {};
    constexpr int PER = 64 * KS / (NW * 64);  // doubles per thread per operand per stage (8 or 4)
    constexpr int TPR = KS / PER;             // threads per row (4 or 8)
    const int lr = tid / TPR, lk = (tid % TPR) * PER;
    const double *Ag = A + (size_t)(row0 + lr) * lda + lk;
    const double *Bg = B + (size_t)(col0 + lr) * ldb + lk;
    double2 pa[PER / 2], pb[PER / 2], qa[PER / 2], qb[PER / 2];
#define GLOAD(S, K0) if ((K0) < kend) { _Pragma("unroll") for (int q = 0; q < PER / 2; q++) { S##a[q] = *(const double2 *)(Ag + (K0) + 2 * q); S##b[q] = *(const double2 *)(Bg + (K0) + 2 * q); } }
#define LSTORE(S, BUF) { _Pragma("unroll") for (int q = 0; q < PER / 2; q++) { *(double2 *)&As[BUF][lr * LD + lk + 2 * q] = S##a[q]; *(double2 *)&Bs[BUF][lr * LD + lk + 2 * q] = S##b[q]; } }
    const int fa = (wr * 32 + (lane & 15)) * LD + (lane >> 4);
    const int fb = (wc * 32 + (lane & 15)) * LD + (lane >> 4);
    constexpr int KK0 = 0, KKN = KS / (NW / 4);
    auto compute = [&](int buf) {
        const int kofs = grp * KKN;
#pragma unroll
        for (int kk = KK0; kk < KKN; kk += 4) {
            const double a0 = As[buf][fa + kofs + kk], a1 = As[buf][fa + 16 * LD + kofs + kk];
            const double b0 = Bs[buf][fb + kofs + kk], b1 = Bs[buf][fb + 16 * LD + kofs + kk];
            acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
            acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
            acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
            acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
        }
    };
    for (int q = 0; q < PER / 2; q++) { pa[q] = pb[q] = qa[q] = qb[q] = make_double2(0, 0); }
    GLOAD(p, kbeg); GLOAD(q, kbeg + KS); LSTORE(p, 0); GLOAD(p, kbeg + 2 * KS);
    __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 2 * KS) {
        if (k0 + KS < kend) LSTORE(q, 1);
        GLOAD(q, k0 + 3 * KS);
        compute(0);
        __syncthreads();
        if (k0 + KS >= kend) break;
        if (k0 + 2 * KS < kend) LSTORE(p, 0);
        GLOAD(p, k0 + 4 * KS);
        compute(1);
        __syncthreads();
    }
    double s = 0;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    C[(size_t)blockIdx.x * NW * 64 + tid] = s;
}
template <int NW> float run(const double *A, const double *B, double *C, int K, int rt, int ct)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
        hipEventRecord(e0);
        k<NW><<<rt * ct, NW * 64>>>(A, B, C, K, K, K, rt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3;
}
int main()
{
    const int M = 4096, N = 512, K = 320;
    double *A, *B, *C;
    hipMalloc(&A, sizeof(double) * M * K); hipMalloc(&B, sizeof(double) * N * K); hipMalloc(&C, sizeof(double) * 1024 * 512);
    hipMemset(A, 0, sizeof(double) * M * K); hipMemset(B, 0, sizeof(double) * N * K);
    const int rt = M / 64;
    for (int ct : {3, 8}) printf("tiles=%d: 4 waves %.1f us   8 waves %.1f us (event pair overhead ~6 us included)\n", rt * ct, run<4>(A, B, C, K, rt, ct), run<8>(A, B, C, K, rt, ct));
    return 0;
}
