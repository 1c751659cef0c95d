// A 16-row panel per 1024-thread workgroup (16 waves = four per SIMD, one workgroup per CU): the A panel sits in LDS, every
// wave streams the B fragments of ITS 16 columns straight from global memory (k-major B: a fragment load is 4 rows x 128
// contiguous bytes) with PF loads in flight, one accumulator chain per column block, k ascending.  Question (DESIGN §9.1,
// round 4): ubench_gemm5's 4-wave form had one wave per SIMD and nobody to hide behind — does the same panel reach the MFMA
// rate with four waves per SIMD and no LDS staging of B?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int KDIM, int PF, bool X4>
__global__ __launch_bounds__(1024) void kp(const double *A, const double *Bt, double *C, int lda, int ldb, int ldc, int ncb)
{
    constexpr int AS = KDIM + 2, KST = KDIM / 4;
    __shared__ double As[16 * AS];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int row0 = blockIdx.x * 16;
    for (int idx = tid; idx < 16 * KDIM; idx += 1024) As[(idx / KDIM) * AS + idx % KDIM] = A[(size_t)(row0 + idx / KDIM) * lda + idx % KDIM];
    __syncthreads();
    const double *ap = As + (lane & 15) * AS + (lane >> 4);
    if constexpr (!X4) {
        for (int cb = wave; cb < ncb; cb += 16) {
            const double *bp = Bt + (size_t)(lane >> 4) * ldb + cb * 16 + (lane & 15);
            v4d acc = {0, 0, 0, 0};
            double b[PF];
#pragma unroll
            for (int p = 0; p < PF; p++) b[p] = bp[(size_t)(4 * p) * ldb];
#pragma unroll
            for (int ks = 0; ks < KST; ks++) {
                const double a = ap[4 * ks];
                acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[ks % PF], acc, 0, 0, 0);
                const int nx = ks + PF < KST ? ks + PF : KST - 1;
                b[ks % PF] = bp[(size_t)(4 * nx) * ldb];
            }
#pragma unroll
            for (int r = 0; r < 4; r++) C[(size_t)(row0 + (lane >> 4) + 4 * r) * ldc + cb * 16 + (lane & 15)] = acc[r];
        }
    } else {
        // 16-byte loads: the lane holds columns 2n, 2n+1 of a 32-column unit: two accumulators, the A fragment shared
        for (int cb = wave; cb < ncb / 2; cb += 16) {
            const double *bp = Bt + (size_t)(lane >> 4) * ldb + cb * 32 + 2 * (lane & 15);
            v4d acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
            double2 b[PF];
#pragma unroll
            for (int p = 0; p < PF; p++) b[p] = *(const double2 *)(bp + (size_t)(4 * p) * ldb);
#pragma unroll
            for (int ks = 0; ks < KST; ks++) {
                const double a = ap[4 * ks];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[ks % PF].x, acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b[ks % PF].y, acc1, 0, 0, 0);
                const int nx = ks + PF < KST ? ks + PF : KST - 1;
                b[ks % PF] = *(const double2 *)(bp + (size_t)(4 * nx) * ldb);
            }
#pragma unroll
            for (int r = 0; r < 4; r++) {
                C[(size_t)(row0 + (lane >> 4) + 4 * r) * ldc + cb * 32 + 2 * (lane & 15)] = acc0[r];
                C[(size_t)(row0 + (lane >> 4) + 4 * r) * ldc + cb * 32 + 2 * (lane & 15) + 1] = acc1[r];
            }
        }
    }
}

template <int KDIM, int PF, bool X4>
float run(const double *A, const double *Bt, double *C, int lda, int ldb, int ldc, int ncb, int nwg)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 10; rep++) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; i++) kp<KDIM, PF, X4><<<nwg, 1024>>>(A, Bt, C, lda, ldb, ldc, ncb);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3 / 20;
}

int main()
{
    const int M = 4096, K = 320, N = 256;
    double *A, *Bt, *C;
    double *ha = (double *)malloc(sizeof(double) * M * K), *hb = (double *)malloc(sizeof(double) * K * K);
    for (int i = 0; i < M * K; i++) ha[i] = (double)((i * 2654435761u) >> 20) / 4096.0 - 0.5;
    for (int i = 0; i < K * K; i++) hb[i] = (double)(((i + 77) * 2246822519u) >> 20) / 4096.0 - 0.5;
    hipMalloc(&A, sizeof(double) * M * K); hipMalloc(&Bt, sizeof(double) * K * K); hipMalloc(&C, sizeof(double) * M * K);
    hipMemcpy(A, ha, sizeof(double) * M * K, hipMemcpyHostToDevice); hipMemcpy(Bt, hb, sizeof(double) * K * K, hipMemcpyHostToDevice);
    // K phase: C[4096 x 256] = A[4096 x 320] . Bt[320 x 256]   (ldb = 256)
    kp<320, 8, false><<<256, 1024>>>(A, Bt, C, K, N, N, 16);
    double *hc = (double *)malloc(sizeof(double) * M * N);
    hipMemcpy(hc, C, sizeof(double) * M * N, hipMemcpyDeviceToHost);
    double err = 0;
    for (int row = 0; row < M; row += 97)
        for (int c = 0; c < N; c += 5) {
            double s = 0; for (int kk = 0; kk < K; kk++) s += ha[row * K + kk] * hb[kk * N + c];
            err = fmax(err, fabs(s - hc[(size_t)row * N + c]));
        }
    printf("max error x2: %g\n", err);
    kp<320, 8, true><<<256, 1024>>>(A, Bt, C, K, N, N, 16);
    hipMemcpy(hc, C, sizeof(double) * M * N, hipMemcpyDeviceToHost);
    err = 0;
    for (int row = 0; row < M; row += 97)
        for (int c = 0; c < N; c += 5) {
            double s = 0; for (int kk = 0; kk < K; kk++) s += ha[row * K + kk] * hb[kk * N + c];
            err = fmax(err, fabs(s - hc[(size_t)row * N + c]));
        }
    printf("max error x4: %g\n", err);
    printf("MFMA bound for both phases at 2.4 GHz: 8.5 us (K phase: 16 units x 80; W phase: 20 units x 64 MFMAs per CU)\n");
    printf("K phase 256 WG x (16 x 256, K=320), 8-B loads, PF 4/8/16: %.1f %.1f %.1f us\n",
           run<320, 4, false>(A, Bt, C, K, N, N, 16, 256), run<320, 8, false>(A, Bt, C, K, N, N, 16, 256), run<320, 16, false>(A, Bt, C, K, N, N, 16, 256));
    printf("K phase, 16-B loads (8 waves busy), PF 4/8/16:             %.1f %.1f %.1f us\n",
           run<320, 4, true>(A, Bt, C, K, N, N, 16, 256), run<320, 8, true>(A, Bt, C, K, N, N, 16, 256), run<320, 16, true>(A, Bt, C, K, N, N, 16, 256));
    // W phase: C[4096 x 320] = A[4096 x 256] . Bt[256 x 320]
    printf("W phase 256 WG x (16 x 320, K=256), 8-B loads, PF 4/8/16: %.1f %.1f %.1f us\n",
           run<256, 4, false>(A, Bt, C, K, K, K, 20, 256), run<256, 8, false>(A, Bt, C, K, K, K, 20, 256), run<256, 16, false>(A, Bt, C, K, K, K, 20, 256));
    printf("K phase, 192 columns (12 units), PF 8: %.1f us; 64 columns: %.1f us\n", run<320, 8, false>(A, Bt, C, K, N, N, 12, 256),
           run<320, 8, false>(A, Bt, C, K, N, N, 4, 256));
    printf("empty-ish launch (ncb = 0): %.1f us\n", run<320, 8, false>(A, Bt, C, K, N, N, 0, 256));
    return 0;
}
