// ipc_probe.hip — can P processes on ONE GPU exchange packed partial sums through hipIpc-mapped buffers, with kernels of
// different processes handing off to each other through flags?  (probe for the library's all-gather exchange)
//   ipc_probe <rank> <world> <dir> [len] [iters] [finegrained 0/1]
// Every rank: recv[2][world][len] doubles + flag[world] (one 128-byte line each) exported through hipIpcGetMemHandle;
// per iteration: fill partial -> push kernel (partial into recv[parity][rank] of EVERY rank, release, flag = epoch) ->
// wait kernel (one wave, bounded spin) -> sum kernel (rank order) -> check on the host at the end.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include <unistd.h>

#include <string>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "[rank %d] %s: %s\n", g_rank, #x, hipGetErrorString(e_)); exit(2); } } while (0)
static int g_rank = 0;

struct Peers { double *recv[8]; int *flag[8]; };

__global__ void fill_kernel(double *part, int len, int rank, int it)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < len; k += gridDim.x * blockDim.x)
        part[k] = (double)(rank + 1) * 1e-3 * (double)(k % 977) + (double)it;
}

// one workgroup per (peer, chunk): plain stores into the peer's slot, then ONE release + flag store per peer by the last
// workgroup of that peer (counter in local memory)
__global__ __launch_bounds__(256) void push_kernel(Peers p, const double *part, int len, int rank, int world, int parity, int epoch, int *done_cnt)
{
    const int peer = blockIdx.y, nb = gridDim.x;
    double *dst = p.recv[peer] + ((size_t)parity * world + rank) * len;
    for (int k = blockIdx.x * 256 + threadIdx.x; k < len; k += nb * 256) dst[k] = part[k];
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        const int prev = atomicAdd(&done_cnt[peer * 32], 1);
        if (prev == nb - 1) {
            done_cnt[peer * 32] = 0;
            __threadfence_system();
            __hip_atomic_store(p.flag[peer] + rank * 32, epoch, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

__global__ void wait_kernel(const int *flag, int world, int epoch, int *err, long long max_ticks)
{
    const int lane = threadIdx.x;
    if (lane >= world) return;
    const long long t0 = wall_clock64();
    while (__hip_atomic_load(flag + lane * 32, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < epoch) {
        if (wall_clock64() - t0 > max_ticks) { atomicMax(err, epoch); break; }
        __builtin_amdgcn_s_sleep(8);
    }
}

__global__ void sum_kernel(const double *recv, int len, int world, int parity, double *total)
{
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < len; k += gridDim.x * blockDim.x) {
        double s = 0.0;
        for (int r = 0; r < world; r++) s += __builtin_nontemporal_load(recv + ((size_t)parity * world + r) * len + k);
        total[k] = s;
    }
}

static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }

int main(int argc, char **argv)
{
    if (argc < 4) { fprintf(stderr, "usage: ipc_probe rank world dir [len] [iters] [fine]\n"); return 1; }
    const int rank = atoi(argv[1]), world = atoi(argv[2]);
    const std::string dir = argv[3];
    const int len = argc > 4 ? atoi(argv[4]) : 16395, iters = argc > 5 ? atoi(argv[5]) : 200, fine = argc > 6 ? atoi(argv[6]) : 1;
    g_rank = rank;
    CHK(hipSetDevice(0));
    const size_t recv_bytes = sizeof(double) * 2 * world * len, flag_bytes = sizeof(int) * 32 * 8;
    char *base = nullptr;
    const size_t tot = ((recv_bytes + 255) & ~(size_t)255) + flag_bytes;
    if (fine) CHK(hipExtMallocWithFlags((void **)&base, tot, hipDeviceMallocFinegrained));
    else CHK(hipMalloc((void **)&base, tot));
    CHK(hipMemset(base, 0, tot));
    CHK(hipDeviceSynchronize());
    hipIpcMemHandle_t hnd;
    CHK(hipIpcGetMemHandle(&hnd, base));
    {
        const std::string tmp = dir + "/h" + std::to_string(rank) + ".tmp", fin = dir + "/h" + std::to_string(rank);
        FILE *f = fopen(tmp.c_str(), "wb");
        fwrite(&hnd, sizeof(hnd), 1, f);
        fclose(f);
        rename(tmp.c_str(), fin.c_str());
    }
    Peers p = {};
    std::vector<char *> bases(world, nullptr);
    for (int r = 0; r < world; r++) {
        if (r == rank) { bases[r] = base; }
        else {
            const std::string fin = dir + "/h" + std::to_string(r);
            FILE *f = nullptr;
            const double t0 = now();
            while (!(f = fopen(fin.c_str(), "rb"))) { if (now() - t0 > 60) { fprintf(stderr, "[rank %d] no handle of rank %d\n", rank, r); return 3; } usleep(1000); }
            hipIpcMemHandle_t h2;
            if (fread(&h2, sizeof(h2), 1, f) != 1) return 3;
            fclose(f);
            CHK(hipIpcOpenMemHandle((void **)&bases[r], h2, hipIpcMemLazyEnablePeerAccess));
        }
        p.recv[r] = (double *)bases[r];
        p.flag[r] = (int *)(bases[r] + ((recv_bytes + 255) & ~(size_t)255));
    }
    double *part, *total;
    int *cnt, *err;
    CHK(hipMalloc((void **)&part, sizeof(double) * len));
    CHK(hipMalloc((void **)&total, sizeof(double) * len));
    CHK(hipMalloc((void **)&cnt, sizeof(int) * 32 * 8));
    CHK(hipMalloc((void **)&err, sizeof(int)));
    CHK(hipMemset(cnt, 0, sizeof(int) * 32 * 8));
    CHK(hipMemset(err, 0, sizeof(int)));
    hipStream_t st;
    CHK(hipStreamCreate(&st));
    // barrier through files: everybody has opened everybody
    {
        const std::string fin = dir + "/ready" + std::to_string(rank);
        FILE *f = fopen(fin.c_str(), "wb"); fclose(f);
        for (int r = 0; r < world; r++) {
            const std::string o = dir + "/ready" + std::to_string(r);
            const double t0 = now();
            while (access(o.c_str(), F_OK) != 0) { if (now() - t0 > 60) return 3; usleep(1000); }
        }
    }
    const long long max_ticks = 100000000LL / 2;  // 0.5 s at 100 MHz
    std::vector<double> out(len);
    int bad = 0;
    double t_start = 0.0;
    const int nb = 8;
    for (int it = 0; it < iters; it++) {
        if (it == iters / 4) { CHK(hipStreamSynchronize(st)); t_start = now(); }
        const int epoch = it + 1, parity = it & 1;
        hipLaunchKernelGGL(fill_kernel, dim3(64), dim3(256), 0, st, part, len, rank, it);
        hipLaunchKernelGGL(push_kernel, dim3(nb, world), dim3(256), 0, st, p, part, len, rank, world, parity, epoch, cnt);
        hipLaunchKernelGGL(wait_kernel, dim3(1), dim3(64), 0, st, p.flag[rank], world, epoch, err, max_ticks);
        hipLaunchKernelGGL(sum_kernel, dim3(64), dim3(256), 0, st, p.recv[rank], len, world, parity, total);
        if (it % 37 == 0 || it == iters - 1) {
            CHK(hipMemcpyAsync(out.data(), total, sizeof(double) * len, hipMemcpyDeviceToHost, st));
            CHK(hipStreamSynchronize(st));
            for (int k = 0; k < len; k++) {
                double s = 0.0;
                for (int r = 0; r < world; r++) s += (double)(r + 1) * 1e-3 * (double)(k % 977) + (double)it;
                if (out[k] != s) { if (bad < 5) fprintf(stderr, "[rank %d] it %d k %d: %.17g != %.17g\n", rank, it, k, out[k], s); bad++; }
            }
        }
    }
    CHK(hipStreamSynchronize(st));
    const double dt = now() - t_start;
    int herr = 0;
    CHK(hipMemcpy(&herr, err, sizeof(int), hipMemcpyDeviceToHost));
    printf("[rank %d/%d] len %d fine %d: %d iterations, %.1f us per iteration (4 launches), mismatches %d, wait timeouts at epoch %d\n", rank, world, len,
           fine, iters - iters / 4, 1e6 * dt / (iters - iters / 4), bad, herr);
    for (int r = 0; r < world; r++)
        if (r != rank) (void)hipIpcCloseMemHandle(bases[r]);
    return (bad || herr) ? 4 : 0;
}
