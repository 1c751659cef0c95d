// A 16-row panel against 64*NT columns per workgroup (4 waves x NT MFMA column tiles, one workgroup per CU): the
// main loop a fused K_nm -> W -> covloss kernel per row panel would run.  Question: with 256 workgroups each
// streaming the whole B operand (64*NT rows x 16 doubles = 32-40 KB per 1024-1280 MFMA cycles) out of L2, does the
// loop still run near the MFMA rate?   Same three-stage LDS scheme as gemm.hip (16-deep stages, XOR swizzle).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define KS 16

template <int NT>
__global__ __launch_bounds__(256, 1) void k(const double *A, const double *B, double *C, int K, int lda, int ldb, int ncolgroups)
{
    constexpr int BR = 64 * NT, ASZ = 16 * KS, BSZ = BR * KS, NBL = BR / 32;  // B loads per thread per stage
    __shared__ double As[3 * ASZ];
    __shared__ double Bs[3 * BSZ];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int row0 = blockIdx.x * 16, col0 = (blockIdx.x % ncolgroups) * BR;
    v4d acc[NT] = {};
    // global side: 8 threads per row (16 B each): a wave covers 8 full 128-B lines per instruction
    const int lr = tid >> 3, lc = tid & 7;
    const double *Ag = A + (size_t)(row0 + (lr & 15)) * lda + lc * 2;
    const double *Bg = B + (size_t)(col0 + lr) * ldb + lc * 2;
    const int la = (lr & 15) * KS + ((lc ^ (((lr & 15) >> 1) & 7)) << 1);
#define B_ALL(OP, S, ...) OP(S, 0, __VA_ARGS__) OP(S, 1, __VA_ARGS__) OP(S, 2, __VA_ARGS__) OP(S, 3, __VA_ARGS__) OP(S, 4, __VA_ARGS__) \
    OP(S, 5, __VA_ARGS__) OP(S, 6, __VA_ARGS__) OP(S, 7, __VA_ARGS__) OP(S, 8, __VA_ARGS__) OP(S, 9, __VA_ARGS__)
    // named scalars, not arrays: the register stages of ten 16-B loads went to scratch as arrays
#define DEF_LB(S, I, X) const int lb##I = (lr + 32 * I) * KS + ((lc ^ (((lr + 32 * I) >> 1) & 7)) << 1);
    B_ALL(DEF_LB, x, 0)
    const int fr_a = lane & 15;
    int oa[4], ob[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int c = 2 * j + (lane >> 5), lo = (lane >> 4) & 1;
        oa[j] = fr_a * KS + ((c ^ ((fr_a >> 1) & 7)) << 1) + lo;
        ob[j] = (wave * 16 * NT + fr_a) * KS + ((c ^ ((fr_a >> 1) & 7)) << 1) + lo;  // + 16 rows per column tile: same key
    }
    double2 pa, qa, pb0, pb1, pb2, pb3, pb4, pb5, pb6, pb7, pb8, pb9, qb0, qb1, qb2, qb3, qb4, qb5, qb6, qb7, qb8, qb9;
    double fra[2], frb[2][NT];
    const int nst = K / KS;
#define GLB_(S, I, ST) if (I < NBL) S##b##I = *(const double2 *)(Bg + (size_t)(32 * I) * ldb + (ST) * KS);
#define GL(S, ST) { if (tid < 128) S##a = *(const double2 *)(Ag + (ST) * KS); B_ALL(GLB_, S, ST) }
#define LSB_(S, I, BUF, I0, I1) if (I >= (I0) && I < (I1)) *(double2 *)(Bs + (BUF) * BSZ + lb##I) = S##b##I;
#define LS(S, BUF, I0, I1) { B_ALL(LSB_, S, BUF, I0, I1) }
#define LSA_(S, BUF) { if (tid < 128) *(double2 *)(As + (BUF) * ASZ + la) = S##a; }
#define RDF(SL, BUF, J) { fra[SL] = As[(BUF) * ASZ + oa[J]]; _Pragma("unroll") for (int t = 0; t < NT; t++) frb[SL][t] = Bs[(BUF) * BSZ + ob[J] + t * 16 * KS]; }
#define STAGE(R, DO_ST, DO_LD, SLD)                                                              \
    {                                                                                            \
        __syncthreads();                                                                         \
        _Pragma("unroll") for (int j = 0; j < 4; j++) {                                          \
            const int sl = j & 1;                                                                \
            if (j < 3) RDF(sl ^ 1, cur, j + 1) else RDF(sl ^ 1, nxt, 0)                          \
            _Pragma("unroll") for (int t = 0; t < NT; t++)                                       \
                acc[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(fra[sl], frb[sl][t], acc[t], 0, 0, 0); \
            if (DO_ST) { if (j == 0) { LSA_(R, stb) LS(R, stb, 0, NBL / 2) } if (j == 1) LS(R, stb, NBL / 2, NBL) } \
            if (DO_LD) { if (j == 2) GL(R, SLD) }                                                \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
        const int t_ = cur; cur = nxt; nxt = stb; stb = t_;                                      \
    }
    int cur = 0, nxt = 1, stb = 2, s = 0;
    GL(p, 0) GL(q, 1)
    LSA_(p, 0) LS(p, 0, 0, NBL) GL(p, 2)
    LSA_(q, 1) LS(q, 1, 0, NBL) GL(q, 3)
    __syncthreads();
    RDF(0, 0, 0)
    for (; s + 5 < nst; s += 2) {
        STAGE(p, 1, 1, s + 4)
        STAGE(q, 1, 1, s + 5)
    }
    for (; s < nst; s += 2) {
        if (s + 4 < nst) STAGE(p, 1, 1, s + 4)
        else if (s + 2 < nst) STAGE(p, 1, 0, 0)
        else STAGE(p, 0, 0, 0)
        if (s + 1 >= nst) break;
        if (s + 3 < nst) STAGE(q, 1, 0, 0)
        else STAGE(q, 0, 0, 0)
    }
    // D layout: col = lane & 15, row = (lane >> 4) + 4 reg
#pragma unroll
    for (int t = 0; t < NT; t++)
#pragma unroll
        for (int r = 0; r < 4; r++)
            C[(size_t)(row0 + (lane >> 4) + 4 * r) * (64 * NT * ncolgroups) + col0 + wave * 16 * NT + t * 16 + (lane & 15)] = acc[t][r];
}

template <int NT> float run(const double *A, const double *B, double *C, int K, int nwg, int ncg)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
        hipEventRecord(e0);
        k<NT><<<nwg, 256>>>(A, B, C, K, K, K, ncg);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3;
}

int main()
{
    const int M = 4096, N = 640, K = 320;
    double *A, *B, *C;
    double *ha = (double *)malloc(sizeof(double) * M * K), *hb = (double *)malloc(sizeof(double) * N * K);
    for (int i = 0; i < M * K; i++) ha[i] = (double)((i * 2654435761u) >> 20) / 4096.0 - 0.5;
    for (int i = 0; i < N * K; i++) hb[i] = (double)(((i + 77) * 2246822519u) >> 20) / 4096.0 - 0.5;
    hipMalloc(&A, sizeof(double) * M * K); hipMalloc(&B, sizeof(double) * N * K); hipMalloc(&C, sizeof(double) * M * N);
    hipMemcpy(A, ha, sizeof(double) * M * K, hipMemcpyHostToDevice); hipMemcpy(B, hb, sizeof(double) * N * K, hipMemcpyHostToDevice);
    // check NT = 4 (two column groups of 256) against the host
    k<4><<<256, 256>>>(A, B, C, K, K, K, 2);
    double *hc = (double *)malloc(sizeof(double) * M * 512);
    hipMemcpy(hc, C, sizeof(double) * M * 512, hipMemcpyDeviceToHost);
    double err = 0;
    for (int wg = 0; wg < 256; wg += 37)
        for (int r = 0; r < 16; r++)
            for (int c = 0; c < 256; c += 5) {
                const int row = wg * 16 + r, col = (wg % 2) * 256 + c;
                double s = 0; for (int kk = 0; kk < K; kk++) s += ha[row * K + kk] * hb[col * K + kk];
                err = fmax(err, fabs(s - hc[(size_t)row * 512 + col]));
            }
    printf("max error NT=4: %g\n", err);
    printf("256 panels x 16 rows x 256 cols x K=320 (MFMA bound 8.5 us + ~6 us event pair): %.1f us\n", run<4>(A, B, C, K, 256, 2));
    printf("256 panels x 16 rows x 320 cols x K=256 (W phase, bound 8.5 us):                 %.1f us\n", run<5>(A, B, C, 256, 256, 2));
    printf("128 panels (half the chip) NT=4:                                                  %.1f us\n", run<4>(A, B, C, K, 128, 2));
    return 0;
}
