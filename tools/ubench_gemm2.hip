// LDS-staged fp64 MFMA GEMM main loop in isolation; variants by MODE.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define BM 64
#define KS 32
#define LD 34
// MODE bit0: do global loads, bit1: do LDS stores+barriers, bit2: do ds_read+MFMA
template <int MODE>
__global__ __launch_bounds__(256) void k(const double *A, const double *B, double *C, int K, int lda, int ldb, int row_tiles)
{
    __shared__ double As[2][BM * LD];
    __shared__ double Bs[2][BM * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int rt = blockIdx.x % row_tiles, ct = blockIdx.x / row_tiles;
    const int row0 = rt * 64, col0 = ct * 64, kbeg = 0, kend = K;
    v4d acc[2][2] = {};
    const int lr = tid >> 2, lk = (tid & 3) * 8;
    const double *Ag = A + (size_t)(row0 + lr) * lda + lk;
    const double *Bg = B + (size_t)(col0 + lr) * ldb + lk;
    double2 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
    pa0 = pa1 = pa2 = pa3 = pb0 = pb1 = pb2 = pb3 = make_double2(1.0, 2.0);
    qa0 = qa1 = qa2 = qa3 = qb0 = qb1 = qb2 = qb3 = make_double2(1.0, 2.0);
#define GLOAD(S, K0)                                                                             \
    if ((MODE & 1) && (K0) < kend) {                                                             \
        S##a0 = *(const double2 *)(Ag + (K0)); S##a1 = *(const double2 *)(Ag + (K0) + 2);        \
        S##a2 = *(const double2 *)(Ag + (K0) + 4); S##a3 = *(const double2 *)(Ag + (K0) + 6);    \
        S##b0 = *(const double2 *)(Bg + (K0)); S##b1 = *(const double2 *)(Bg + (K0) + 2);        \
        S##b2 = *(const double2 *)(Bg + (K0) + 4); S##b3 = *(const double2 *)(Bg + (K0) + 6);    \
    }
#define LSTORE(S, BUF)                                                                           \
    if (MODE & 2) {                                                                              \
        double *da = &As[BUF][lr * LD + lk], *db = &Bs[BUF][lr * LD + lk];                       \
        *(double2 *)(da) = S##a0; *(double2 *)(da + 2) = S##a1;                                  \
        *(double2 *)(da + 4) = S##a2; *(double2 *)(da + 6) = S##a3;                              \
        *(double2 *)(db) = S##b0; *(double2 *)(db + 2) = S##b1;                                  \
        *(double2 *)(db + 4) = S##b2; *(double2 *)(db + 6) = S##b3;                              \
    }
    const int fa = (wr * 32 + (lane & 15)) * LD + (lane >> 4);
    const int fb = (wc * 32 + (lane & 15)) * LD + (lane >> 4);
    auto compute = [&](int buf) {
        if (MODE & 4) {
#pragma unroll
            for (int kk = 0; kk < KS; kk += 4) {
                const double a0 = As[buf][fa + kk], a1 = As[buf][fa + 16 * LD + kk];
                const double b0 = Bs[buf][fb + kk], b1 = Bs[buf][fb + 16 * LD + kk];
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b0, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a1, b1, acc[1][1], 0, 0, 0);
            }
        }
    };
    GLOAD(p, kbeg); GLOAD(q, kbeg + KS); LSTORE(p, 0); GLOAD(p, kbeg + 2 * KS);
    if (MODE & 2) __syncthreads();
    for (int k0 = kbeg; k0 < kend; k0 += 2 * KS) {
        if (k0 + KS < kend) LSTORE(q, 1);
        GLOAD(q, k0 + 3 * KS);
        compute(0);
        if (MODE & 2) __syncthreads();
        if (k0 + KS >= kend) break;
        if (k0 + 2 * KS < kend) LSTORE(p, 0);
        GLOAD(p, k0 + 4 * KS);
        compute(1);
        if (MODE & 2) __syncthreads();
    }
    double s = pa0.x + qa0.y + pb3.x + qb3.y;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    C[(size_t)blockIdx.x * 256 + tid] = s;
}
template <int MODE> float run(const double *A, const double *B, double *C, int K, int rt, int ct)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
        hipEventRecord(e0);
        k<MODE><<<rt * ct, 256>>>(A, B, C, K, K, K, rt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3;
}
__global__ void empty(double *C) { if (threadIdx.x == 0 && blockIdx.x == 0) C[0] = 1; }
int main()
{
    const int M = 4096, N = 512, K = 320;
    double *A, *B, *C;
    hipMalloc(&A, sizeof(double) * M * K); hipMalloc(&B, sizeof(double) * N * K); hipMalloc(&C, sizeof(double) * 1024 * 256);
    hipMemset(A, 0, sizeof(double) * M * K); hipMemset(B, 0, sizeof(double) * N * K);
    const int rt = M / 64;
    {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1); float best = 1e9;
        for (int rep = 0; rep < 20; rep++) { hipEventRecord(e0); empty<<<192, 256>>>(C); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("empty kernel (event pair): %.1f us\n", best * 1e3);
    }
    for (int ct : {3, 8}) {
        printf("tiles=%d: all=%.1f  loads=%.1f  loads+lds=%.1f  lds+mfma=%.1f  mfma(ds_read)=%.1f us\n", rt * ct,
               run<7>(A, B, C, K, rt, ct), run<1>(A, B, C, K, rt, ct), run<3>(A, B, C, K, rt, ct), run<6>(A, B, C, K, rt, ct),
               run<4>(A, B, C, K, rt, ct));
    }
    return 0;
}
