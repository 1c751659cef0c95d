// Isolates the fp64 MFMA GEMM main loop: loads only / MFMA only / both, direct-fragment form.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

template <int MODE>  // 0 both, 1 loads only, 2 mfma only
__global__ __launch_bounds__(256) void k(const double *A, const double *B, double *C, int M, int N, int K, int lda, int ldb,
                                         int row_tiles)
{
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int rt = blockIdx.x % row_tiles, ct = blockIdx.x / row_tiles;
    const int row0 = rt * 64, col0 = ct * 64;
    v4d acc[2][2] = {};
    const int fr = lane & 15, fk = 2 * (lane >> 4);
    const double *A0 = A + (size_t)(row0 + wr * 32 + fr) * lda + fk;
    const double *A1 = A0 + (size_t)16 * lda;
    const double *B0 = B + (size_t)(col0 + wc * 32 + fr) * ldb + fk;
    const double *B1 = B0 + (size_t)16 * ldb;
    double2 a0[2], a1[2], b0[2], b1[2];
    double sink = 0;
    auto gload = [&](int k0) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            a0[h] = *(const double2 *)(A0 + k0 + 8 * h);
            a1[h] = *(const double2 *)(A1 + k0 + 8 * h);
            b0[h] = *(const double2 *)(B0 + k0 + 8 * h);
            b1[h] = *(const double2 *)(B1 + k0 + 8 * h);
        }
    };
    if (MODE != 2) gload(0);
    else { for (int h = 0; h < 2; h++) { a0[h] = {1.0 * lane, 2.0}; a1[h] = a0[h]; b0[h] = a0[h]; b1[h] = a0[h]; } }
    for (int k0 = 0; k0 < K; k0 += 16) {
        double2 ca0[2], ca1[2], cb0[2], cb1[2];
        for (int h = 0; h < 2; h++) { ca0[h] = a0[h]; ca1[h] = a1[h]; cb0[h] = b0[h]; cb1[h] = b1[h]; }
        if (MODE != 2 && k0 + 16 < K) gload(k0 + 16);
        if (MODE == 1) {
            for (int h = 0; h < 2; h++) sink += ca0[h].x + ca1[h].y + cb0[h].x + cb1[h].y;
        } else {
#pragma unroll
            for (int h = 0; h < 2; h++) {
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca0[h].x, cb0[h].x, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca0[h].x, cb1[h].x, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca1[h].x, cb0[h].x, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca1[h].x, cb1[h].x, acc[1][1], 0, 0, 0);
                acc[0][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca0[h].y, cb0[h].y, acc[0][0], 0, 0, 0);
                acc[0][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca0[h].y, cb1[h].y, acc[0][1], 0, 0, 0);
                acc[1][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca1[h].y, cb0[h].y, acc[1][0], 0, 0, 0);
                acc[1][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(ca1[h].y, cb1[h].y, acc[1][1], 0, 0, 0);
            }
        }
    }
    double s = sink;
    for (int i = 0; i < 2; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 4; r++) s += acc[i][j][r];
    C[(size_t)blockIdx.x * 256 + tid] = s;
}

int main()
{
    const int M = 4096, N = 512, K = 320;
    double *A, *B, *C;
    hipMalloc(&A, sizeof(double) * M * K); hipMalloc(&B, sizeof(double) * N * K); hipMalloc(&C, sizeof(double) * 1024 * 256);
    hipMemset(A, 0, sizeof(double) * M * K); hipMemset(B, 0, sizeof(double) * N * K);
    const int rt = M / 64, ctl = N / 64;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; mode++) for (int ct : {8, 3}) {
        float best = 1e9;
        for (int rep = 0; rep < 20; rep++) {
            hipEventRecord(e0);
            if (mode == 0) k<0><<<rt * ct, 256>>>(A, B, C, M, N, K, K, K, rt);
            if (mode == 1) k<1><<<rt * ct, 256>>>(A, B, C, M, N, K, K, K, rt);
            if (mode == 2) k<2><<<rt * ct, 256>>>(A, B, C, M, N, K, K, K, rt);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        printf("mode=%d (0 both,1 loads,2 mfma) tiles=%d: %.1f us  (%.1f TF/s equiv)\n", mode, rt * ct, best * 1e3,
               2.0 * 64 * 64 * K * rt * ct / best / 1e9);
    }
    return 0;
}
