// neighbor.hip — periodic neighbour list on the device (gfx950).
//
// Replaces ase.neighborlist.NeighborList as driven by descriptor/atoms.py:348-363,:402
// (radii rc/2, skin 0, bothways, self_interaction False): pair (i -> j, off) is kept iff
// |x_j - x_i + off.cell| < rc and (j,off) != (i,0); periodic self-images are kept.
// General triclinic cells, any pbc combination, atoms may sit outside the cell.
//
// Two launches:
//   nl_bin_kernel   ONE workgroup (1024 threads): derives the bin grid from the cell
//                   (device-resident, so NPT cells need no host round trip), bins all atoms
//                   with LDS counters, scans, fills and index-sorts each bin (deterministic
//                   neighbour order).
//   nl_build_kernel one wave64 per atom: sweeps the (2R+1)^3 neighbouring bins, 64 candidates
//                   at a time, ballot/popcount-compacts the hits into nbr_j/nbr_shift[i][:].
#include "sgpr_internal.h"

#define NL_MAX_BINS 8192

struct NlGrid {
    double inv[9];   // inverse cell (columns = reciprocal vectors): frac = pos . inv
    int nb[3];
    int rng[3];
    int nbins;
    int pad;
};

__device__ __forceinline__ double det3d(const double *h)
{
    return h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) + h[2] * (h[3] * h[7] - h[4] * h[6]);
}

__global__ __launch_bounds__(1024) void nl_bin_kernel(int N, const double *pos, const double *cell, int pbc0,
                                                      int pbc1, int pbc2, double rc, NlGrid *grid, int *bin_of,
                                                      int *bin_start /*[NL_MAX_BINS+1]*/, int *bin_atoms /*[N]*/,
                                                      int *wrap /*[N][3]*/, int *stat /*[4]*/)
{
    __shared__ int cnt[NL_MAX_BINS];
    __shared__ int start[NL_MAX_BINS + 1];
    __shared__ int part[1024];
    __shared__ NlGrid g;
    const int tid = threadIdx.x;
    if (tid == 0) {
        const int pbc[3] = {pbc0, pbc1, pbc2};
        double h[9];
        for (int k = 0; k < 9; k++) h[k] = cell[k];
        const double dt = det3d(h);
        if (fabs(dt) > 1e-12) {
            const double *a = h, *b = h + 3, *c = h + 6;
            const double bc[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
            const double ca[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
            const double ab[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            for (int k = 0; k < 3; k++) {
                g.inv[3 * k + 0] = bc[k] / dt;
                g.inv[3 * k + 1] = ca[k] / dt;
                g.inv[3 * k + 2] = ab[k] / dt;
            }
            const double V = fabs(dt);
            const double hgt[3] = {V / sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]),
                                   V / sqrt(ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]),
                                   V / sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2])};
            for (int k = 0; k < 3; k++) {
                if (pbc[k]) {
                    int nb = (int)floor(hgt[k] / rc);
                    nb = nb < 1 ? 1 : (nb > 16 ? 16 : nb);
                    g.nb[k] = nb;
                    g.rng[k] = (int)ceil(rc * nb / hgt[k]);
                } else {
                    g.nb[k] = 1;  // open direction: one slab, no images
                    g.rng[k] = 0;
                }
            }
        } else {
            // no usable cell (cluster): everything in one bin, no images
            for (int k = 0; k < 9; k++) g.inv[k] = 0.0;
            for (int k = 0; k < 3; k++) { g.nb[k] = 1; g.rng[k] = 0; }
        }
        g.nbins = g.nb[0] * g.nb[1] * g.nb[2];
        *grid = g;
        stat[0] = 0;  // max neighbour count seen by the build kernel
    }
    __syncthreads();
    const int nbins = g.nbins;
    for (int b = tid; b < nbins; b += 1024) cnt[b] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) {
        const double x = pos[3 * i], y = pos[3 * i + 1], z = pos[3 * i + 2];
        int bidx[3], w[3];
        const int pbc[3] = {pbc0, pbc1, pbc2};
#pragma unroll
        for (int k = 0; k < 3; k++) {
            double f = x * g.inv[k] + y * g.inv[3 + k] + z * g.inv[6 + k];
            w[k] = 0;
            bidx[k] = 0;
            if (pbc[k] && g.nb[k] >= 1 && (g.inv[k] != 0.0 || g.inv[3 + k] != 0.0 || g.inv[6 + k] != 0.0)) {
                const double fl = floor(f);
                w[k] = (int)fl;
                f -= fl;
                int b = (int)(f * g.nb[k]);
                bidx[k] = b >= g.nb[k] ? g.nb[k] - 1 : (b < 0 ? 0 : b);
            }
            wrap[3 * i + k] = w[k];
        }
        const int bin = (bidx[0] * g.nb[1] + bidx[1]) * g.nb[2] + bidx[2];
        bin_of[i] = bin;
        atomicAdd(&cnt[bin], 1);
    }
    __syncthreads();
    // exclusive scan of cnt[0..nbins) -> start
    const int per = (nbins + 1023) / 1024;
    int loc = 0;
    for (int k = 0; k < per; k++) {
        const int b = tid * per + k;
        if (b < nbins) loc += cnt[b];
    }
    part[tid] = loc;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? part[tid - o] : 0;
        __syncthreads();
        part[tid] += v;
        __syncthreads();
    }
    int run = part[tid] - loc;
    for (int k = 0; k < per; k++) {
        const int b = tid * per + k;
        if (b < nbins) {
            start[b] = run;
            run += cnt[b];
        }
    }
    if (tid == 0) start[nbins] = N;
    __syncthreads();
    for (int b = tid; b <= nbins; b += 1024) bin_start[b] = start[b];
    for (int b = tid; b < nbins; b += 1024) cnt[b] = 0;
    __syncthreads();
    for (int i = tid; i < N; i += 1024) {
        const int bin = bin_of[i];
        const int k = atomicAdd(&cnt[bin], 1);
        bin_atoms[start[bin] + k] = i;
    }
    __threadfence_block();
    __syncthreads();
    // index-sort each bin (insertion sort; bins hold ~rc^3 * density atoms)
    for (int b = tid; b < nbins; b += 1024) {
        const int s = start[b], e = start[b + 1];
        for (int p = s + 1; p < e; p++) {
            const int v = bin_atoms[p];
            int q = p - 1;
            while (q >= s && bin_atoms[q] > v) {
                bin_atoms[q + 1] = bin_atoms[q];
                q--;
            }
            bin_atoms[q + 1] = v;
        }
    }
}

__global__ __launch_bounds__(256) void nl_build_kernel(int N, int first, int stride, int count, const double *pos,
                                                       const double *cell, double rc, const NlGrid *grid,
                                                       const int *bin_of, const int *bin_start,
                                                       const int *bin_atoms, const int *wrap, int maxnn, int *nn,
                                                       int *nn_local, int *nbr_j, int *nbr_shift, int *stat)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int il = blockIdx.x * 4 + wave;
    if (il >= count) return;
    const int i = first + il * stride;
    const NlGrid g = *grid;
    double h[9];
#pragma unroll
    for (int k = 0; k < 9; k++) h[k] = cell[k];
    const double xi = pos[3 * i], yi = pos[3 * i + 1], zi = pos[3 * i + 2];
    const int wi0 = wrap[3 * i], wi1 = wrap[3 * i + 1], wi2 = wrap[3 * i + 2];
    const int bi = bin_of[i];
    const int b2 = bi % g.nb[2], b1 = (bi / g.nb[2]) % g.nb[1], b0 = bi / (g.nb[2] * g.nb[1]);
    int base = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int o0 = -g.rng[0]; o0 <= g.rng[0]; o0++) {
        const int t0 = b0 + o0;
        const int c0 = (int)floor((double)t0 / g.nb[0]);
        const int n0 = t0 - c0 * g.nb[0];
        for (int o1 = -g.rng[1]; o1 <= g.rng[1]; o1++) {
            const int t1 = b1 + o1;
            const int c1 = (int)floor((double)t1 / g.nb[1]);
            const int n1 = t1 - c1 * g.nb[1];
            for (int o2 = -g.rng[2]; o2 <= g.rng[2]; o2++) {
                const int t2 = b2 + o2;
                const int c2 = (int)floor((double)t2 / g.nb[2]);
                const int n2 = t2 - c2 * g.nb[2];
                const int nbin = (n0 * g.nb[1] + n1) * g.nb[2] + n2;
                const int s = bin_start[nbin], e = bin_start[nbin + 1];
                for (int p0 = s; p0 < e; p0 += 64) {
                    const int p = p0 + lane;
                    bool hit = false;
                    int j = 0, f0 = 0, f1 = 0, f2 = 0;
                    if (p < e) {
                        j = bin_atoms[p];
                        f0 = c0 - wrap[3 * j] + wi0;
                        f1 = c1 - wrap[3 * j + 1] + wi1;
                        f2 = c2 - wrap[3 * j + 2] + wi2;
                        const double dx = pos[3 * j] - xi + (f0 * h[0] + f1 * h[3] + f2 * h[6]);
                        const double dy = pos[3 * j + 1] - yi + (f0 * h[1] + f1 * h[4] + f2 * h[7]);
                        const double dz = pos[3 * j + 2] - zi + (f0 * h[2] + f1 * h[5] + f2 * h[8]);
                        const double rr = sqrt(dx * dx + dy * dy + dz * dz);
                        hit = rr < rc && !(j == i && f0 == 0 && f1 == 0 && f2 == 0);
                    }
                    const unsigned long long m = __ballot(hit);
                    if (hit) {
                        const int slot = base + __popcll(m & lt);
                        if (slot < maxnn) {
                            nbr_j[(size_t)i * maxnn + slot] = j;
                            nbr_shift[(size_t)i * maxnn + slot] = (f0 & 0xff) | ((f1 & 0xff) << 8) | ((f2 & 0xff) << 16);
                        }
                    }
                    base += __popcll(m);
                }
            }
        }
    }
    if (lane == 0) {
        nn[i] = base < maxnn ? base : maxnn;
        nn_local[il] = base < maxnn ? base : maxnn;
        atomicMax(&stat[0], base);
    }
}

void launch_neighbor_list(const NlParams &p, const double *pos, const double *cell, double rc, void *grid,
                          int *bin_of, int *bin_start, int *bin_atoms, int *wrap, int *nn, int *nn_local,
                          int *nbr_j, int *nbr_shift, int *stat, hipStream_t st)
{
    if (p.N <= 0) return;
    hipLaunchKernelGGL(nl_bin_kernel, dim3(1), dim3(1024), 0, st, p.N, pos, cell, p.pbc[0], p.pbc[1], p.pbc[2], rc,
                       (NlGrid *)grid, bin_of, bin_start, bin_atoms, wrap, stat);
    if (p.count > 0)
        hipLaunchKernelGGL(nl_build_kernel, dim3((p.count + 3) / 4), dim3(256), 0, st, p.N, p.first, p.stride > 0 ? p.stride : 1, p.count, pos,
                           cell, rc, (const NlGrid *)grid, bin_of, bin_start, bin_atoms, wrap, p.maxnn, nn, nn_local,
                           nbr_j, nbr_shift, stat);
}
