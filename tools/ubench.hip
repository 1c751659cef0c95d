// Micro-benchmarks used to calibrate the kernel design (fp64 MFMA issue rate, load latency).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef double v4d __attribute__((ext_vector_type(4)));

__global__ void mfma_rate(int iters, int nacc, double *out, long long *cyc)
{
    v4d a0 = {0,0,0,0}, a1 = a0, a2 = a0, a3 = a0;
    double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-4;
    long long t0 = __builtin_amdgcn_s_memtime();
    if (nacc == 1) {
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; i++) {
            a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void fma_rate(int iters, double *out, long long *cyc)
{
    double a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, x = 1.0000001, y = 1e-9;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) {
        a0 = fma(a0, x, y); a1 = fma(a1, x, y); a2 = fma(a2, x, y); a3 = fma(a3, x, y);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

__global__ void chase(const int *next, int steps, int *out, long long *cyc)
{
    int p = 0;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < steps; i++) p = next[p];
    long long t1 = __builtin_amdgcn_s_memtime();
    out[0] = p;
    cyc[0] = t1 - t0;
}

int main()
{
    double *out; long long *cyc;
    hipMalloc(&out, 8 * 1024 * 1024); hipMalloc(&cyc, 8 * 4096);
    std::vector<long long> h(4096);
    for (int blocks : {1, 256, 1024}) for (int nacc : {1, 4}) {
        const int iters = 2000;
        mfma_rate<<<blocks, 256>>>(iters, nacc, out, cyc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); mfma_rate<<<blocks, 256>>>(iters, nacc, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
        double flops = (double)blocks * 4 * iters * 4 * 2048;
        printf("mfma f64 16x16x4: blocks=%4d nacc=%d  memtime ticks/MFMA=%.1f  wall=%.3f ms  %.1f TF/s\n", blocks, nacc,
               (double)h[0] / (iters * 4), ms, flops / ms / 1e9);
    }
    for (int blocks : {256, 2048}) {
        const int iters = 20000;
        fma_rate<<<blocks, 256>>>(iters, out, cyc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); fma_rate<<<blocks, 256>>>(iters, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, 8 * blocks, hipMemcpyDeviceToHost);
        printf("v_fma_f64: blocks=%4d ticks/FMA=%.2f wall=%.3f ms %.1f TF/s\n", blocks, (double)h[0] / (iters * 4), ms,
               (double)blocks * 256 * iters * 4 * 2 / ms / 1e9);
    }
    for (size_t bytes : {(size_t)256 << 10, (size_t)16 << 20, (size_t)512 << 20}) {
        const size_t n = bytes / 4, stride = 4099;  // pseudo-random walk, one int per 64B+ line mostly
        std::vector<int> nx(n);
        for (size_t i = 0; i < n; i++) nx[i] = (int)((i * 1 + stride * 16) % n);
        int *dn, *dout; hipMalloc(&dn, bytes); hipMalloc(&dout, 64);
        hipMemcpy(dn, nx.data(), bytes, hipMemcpyHostToDevice);
        chase<<<1, 1>>>(dn, 2000, dout, cyc);
        chase<<<1, 1>>>(dn, 2000, dout, cyc);
        hipDeviceSynchronize();
        hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
        printf("pointer chase %6zu KB: %.0f memtime ticks per dependent load\n", bytes >> 10, (double)h[0] / 2000);
        hipFree(dn); hipFree(dout);
    }
    // s_memtime tick rate: compare against wall clock
    {
        mfma_rate<<<1, 64>>>(200000, 4, out, cyc);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0); mfma_rate<<<1, 64>>>(200000, 4, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(h.data(), cyc, 8, hipMemcpyDeviceToHost);
        printf("s_memtime: %lld ticks in %.3f ms -> %.1f MHz\n", h[0], ms, h[0] / ms / 1e3);
    }
    return 0;
}
