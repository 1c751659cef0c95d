// potrf_test.hip — one panel launch of the blocked Cholesky (csrc/linalg.hip::potrf_panel_kernel) on a random SPD matrix:
// time per launch (hip events) and where workgroup 1 spends it (wall_clock64 stamps: 100 MHz ticks).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DPOTRF_STAMPS -o tools/potrf_test tools/potrf_test.hip
#define POTRF_STAMPS 1
#include "../autoforce_amd/csrc/linalg.hip"
#include <stdio.h>
#include <random>
void launch_gemm_nt(const GemmParams &, GemmEpilogue, hipStream_t) {}
int main(int argc, char **argv)
{
    const int m = argc > 1 ? atoi(argv[1]) : 1024, ld = m;
    std::vector<double> A((size_t)m * m, 0.0);
    std::mt19937_64 g(1);
    std::normal_distribution<double> nd;
    for (int i = 0; i < m; i++)
        for (int j = 0; j <= i; j++) A[(size_t)i * ld + j] = A[(size_t)j * ld + i] = (i == j ? m : 0.0) + 0.3 * nd(g);
    double *dA, *dsave; int *info;
    hipMalloc((void **)&dA, sizeof(double) * A.size()); hipMalloc((void **)&dsave, sizeof(double) * 2 * 64 * 64); hipMalloc((void **)&info, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 5; rep++) {
        hipMemcpy(dA, A.data(), sizeof(double) * A.size(), hipMemcpyHostToDevice);
        hipMemset(info, 0, 4);
        const int nblk = 1 + (m - 64 + 63) / 64;
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(potrf_panel_kernel, dim3(nblk), dim3(256), 0, 0, m, dA, ld, 0, info, dsave);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long st[8]; hipMemcpyFromSymbol(st, HIP_SYMBOL(potrf_stamps), sizeof(st));
        printf("launch %.1f us | ticks(10 ns): load %lld diag %lld gap %lld loadP %lld solve %lld store %lld\n", ms * 1e3, st[1] - st[0], st[2] - st[1],
               st[3] - st[2], 0LL, st[4] - st[3], st[5] - st[4]);
    }
    return 0;
}
