// Does spreading a step's working set over many small hipMalloc allocations cost latency (TLB)?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
struct Ptrs { double *p[32]; };
__global__ void touch(Ptrs P, int nbuf, int n, double *out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    double s = 0;
    int idx = i % n;
    for (int b = 0; b < nbuf; b++) {  // dependent chain through the buffers
        const double v = P.p[b][idx];
        s += v;
        idx = (idx + (int)v + 1) % n;
    }
    out[i] = s;
}
int main()
{
    const int nbuf = 24, n = 16384;  // 128 KB each
    Ptrs sep, arena;
    for (int b = 0; b < nbuf; b++) { hipMalloc(&sep.p[b], sizeof(double) * n); hipMemset(sep.p[b], 0, sizeof(double) * n); }
    double *big; hipMalloc(&big, (size_t)64 << 20); hipMemset(big, 0, (size_t)64 << 20);
    for (int b = 0; b < nbuf; b++) arena.p[b] = big + (size_t)b * n;
    double *out; hipMalloc(&out, sizeof(double) * 4096);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int pass = 0; pass < 2; pass++)
        for (int which = 0; which < 2; which++) {
            float best = 1e9, sum = 0;
            for (int rep = 0; rep < 50; rep++) {
                hipEventRecord(e0);
                touch<<<16, 256>>>(which ? arena : sep, nbuf, n, out);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best; sum += ms;
            }
            printf("%s: %d dependent loads: best %.1f us avg %.1f us (event pair ~6 us)\n", which ? "one arena      " : "separate allocs", nbuf, best * 1e3, sum / 50 * 1e3);
        }
    return 0;
}
