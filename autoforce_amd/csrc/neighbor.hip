// neighbor.hip — periodic neighbour list on the device (gfx950).
//
// Replaces ase.neighborlist.NeighborList as driven by descriptor/atoms.py:348-363,:402
// (radii rc/2, skin 0, bothways, self_interaction False): pair (i -> j, off) is kept iff
// |x_j - x_i + off.cell| < rc and (j,off) != (i,0); periodic self-images are kept.
// General triclinic cells, any pbc combination, atoms may sit outside the cell.
//
// Two launches (every kernel boundary costs ~3-5 us here: dispatch + write-back of what the
// kernel dirtied; an earlier count -> scan -> scatter binning was three launches, 21 us):
//   nl_bin_kernel   256 atoms per workgroup: species-sort gather of the caller's positions, bin
//                   grid from the (device-resident) cell, and direct placement into fixed-capacity
//                   bins through one returning atomic per atom (binned copies: index, position,
//                   wrap, species slot).  Also clears the step's accumulators.
//   nl_build_kernel one wave64 per atom: the (2R+1)^3 neighbouring bins are flattened into one
//                   candidate range (lane-parallel prefix over bins) and swept 64 candidates at a
//                   time from the binned copies; hits are ballot/popcount-compacted into LDS,
//                   SORTED by (j, image) with an in-wave bitonic network — the atomic placement
//                   order inside a bin is not reproducible, the sorted list is — and written out.
// The bin counters are re-zeroed by the step's last kernel (finalize).
//
// Reverse index.  The reverse pass stores one gradient per ordered pair, G[i][t], and atom j needs the
// entry of every pair (i -> j): the position of j in i's SORTED list.  Both ends can name the pair
// without knowing each other's list: i found j as candidate (q, k) = (bin offset index, slot in that
// bin) of its sweep, and j finds i under the mirrored offset nbox-1-q at i's own slot.  So wave i
// writes its list position t into T[j][(nbox-1-q)*cap + k_i] and remembers aux[i][t] = q*cap + k_j;
// later atom j reads rev = T[j][aux[j][t']] from its OWN row: one scattered 2-byte store per pair
// here, one local 2-byte load per pair there, no search.
#include "sgpr_internal.h"

#define NL_MAX_BINS 4096
#define NL_SORT_MAX 256  // lists up to this length are sorted in LDS (longer ones keep sweep order)

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

struct BinArgs {
    int N, cap;
    const int *perm;        // sorted -> caller (may be null: identity)
    const double *pos_in;   // caller order
    const double *cell;
    int pbc[3];
    double rc;
    NlGrid *grid;
    double *pos;            // [N][3] sorted order (out)
    int *bin_count;         // [NL_MAX_BINS] atoms per bin (zero on entry)
    int *b_idx;             // [nbins][cap] atom (sorted index)
    double *b_pos;          // [nbins][cap][3] its position
    int *b_wrap;            // [nbins][cap][3] its wrap (floor of the fractional coordinate)
    int *b_slot;            // [nbins][cap] its species slot
    const int *slot;        // [N] species slot by sorted index
    int *bin_of;            // [N]
    int *kslot;             // [N] slot of the atom inside its bin
    int *wrap;              // [N][3]
    int *stat;              // [4]: [1] = largest bin population seen beyond cap (overflow)
    double *zero_a; int n_zero_a;   // accumulators to clear for this step
    double *zero_b; int n_zero_b;
};

__device__ void nl_make_grid(const double *cell, const int *pbc, double rc, NlGrid &g)
{
    double h[9];
    for (int k = 0; k < 9; k++) h[k] = cell[k];
    const double dt = det3d(h);
    if (fabs(dt) > 1e-12) {
        const double *p = h, *q = h + 3, *r = h + 6;
        const double bc[3] = {q[1] * r[2] - q[2] * r[1], q[2] * r[0] - q[0] * r[2], q[0] * r[1] - q[1] * r[0]};
        const double ca[3] = {r[1] * p[2] - r[2] * p[1], r[2] * p[0] - r[0] * p[2], r[0] * p[1] - r[1] * p[0]};
        const double ab[3] = {p[1] * q[2] - p[2] * q[1], p[2] * q[0] - p[0] * q[2], p[0] * q[1] - p[1] * q[0]};
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
}

__global__ __launch_bounds__(256) void nl_bin_kernel(BinArgs a)
{
    __shared__ NlGrid g;
    const int tid = threadIdx.x, wg = blockIdx.x;
    if (tid == 0) {
        nl_make_grid(a.cell, a.pbc, a.rc, g);
        if (wg == 0) *a.grid = g;
    }
    const int gsz = gridDim.x * 256, gid = wg * 256 + tid;
    // this atom's position is requested before the barrier: the gather (perm -> pos_in) and the grid
    // set-up by thread 0 are independent latency chains
    const int i = gid;
    double x = 0.0, y = 0.0, z = 0.0;
    int slot_i = 0;
    if (i < a.N) {
        const int c = a.perm ? a.perm[i] : i;
        x = a.pos_in[3 * c]; y = a.pos_in[3 * c + 1]; z = a.pos_in[3 * c + 2];
        slot_i = a.slot[i];
    }
    for (int k = gid; k < a.n_zero_a; k += gsz) a.zero_a[k] = 0.0;
    for (int k = gid; k < a.n_zero_b; k += gsz) a.zero_b[k] = 0.0;
    __syncthreads();
    if (i >= a.N) return;
    a.pos[3 * i] = x; a.pos[3 * i + 1] = y; a.pos[3 * i + 2] = z;
    int bidx[3], w[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        double f = x * g.inv[k] + y * g.inv[3 + k] + z * g.inv[6 + k];
        w[k] = 0;
        bidx[k] = 0;
        if (a.pbc[k] && (g.inv[k] != 0.0 || g.inv[3 + k] != 0.0 || g.inv[6 + k] != 0.0)) {
            const double fl = floor(f);
            w[k] = (int)fl;
            f -= fl;
            const int b = (int)(f * g.nb[k]);
            bidx[k] = b >= g.nb[k] ? g.nb[k] - 1 : (b < 0 ? 0 : b);
        }
        a.wrap[3 * i + k] = w[k];
    }
    const int bin = (bidx[0] * g.nb[1] + bidx[1]) * g.nb[2] + bidx[2];
    a.bin_of[i] = bin;
    const int k = atomicAdd(&a.bin_count[bin], 1);
    a.kslot[i] = k;
    if (k < a.cap) {
        const size_t e = (size_t)bin * a.cap + k;
        a.b_idx[e] = i;
        a.b_slot[e] = slot_i;
        a.b_pos[3 * e] = x; a.b_pos[3 * e + 1] = y; a.b_pos[3 * e + 2] = z;
        a.b_wrap[3 * e] = w[0]; a.b_wrap[3 * e + 1] = w[1]; a.b_wrap[3 * e + 2] = w[2];
    } else
        atomicMax(&a.stat[1], k + 1);  // rare: capacity exceeded, the host grows it and reruns
}

__global__ __launch_bounds__(256) void nl_build_kernel(int N, int first, int stride, int count, const double *pos,
                                                       const double *cell, double rc, const NlGrid *grid,
                                                       const int *bin_of, const int *bin_count, int cap,
                                                       const int *b_idx, const double *b_pos, const int *b_wrap,
                                                       const int *b_slot, const int *wrap, int maxnn, int *nn,
                                                       int *nn_local, int *nbr_j, int *nbr_shift, int *nn_raw,
                                                       const int *kslot, int *aux, unsigned short *T, int t_stride,
                                                       int *stat)
{
    __shared__ int s_start[4][64], s_pref[4][65], s_code[4][64];
    __shared__ unsigned long long s_key[4][NL_SORT_MAX];
    __shared__ int s_hq[4][NL_SORT_MAX];  // candidate id (q << 12 | k) of the hit with sweep ordinal o
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
    // index arithmetic without integer division (a ~30-instruction sequence each on this ISA): small
    // non-negative operands, so floor((q + 1/2) * (1/w)) in fp32 is exact (the argument is never within
    // 0.5/33 of an integer), and floor(t / nb) for the image count goes through an fp64 reciprocal
    auto fdiv = [](int q, int w, float inv) { (void)w; return (int)(((float)q + 0.5f) * inv); };
    const int bi = bin_of[i];
    const float i_n2 = 1.0f / (float)g.nb[2], i_n1 = 1.0f / (float)g.nb[1];
    const int bq = fdiv(bi, g.nb[2], i_n2);
    const int b2 = bi - bq * g.nb[2];
    const int b0 = fdiv(bq, g.nb[1], i_n1);
    const int b1 = bq - b0 * g.nb[1];
    const int w0 = 2 * g.rng[0] + 1, w1 = 2 * g.rng[1] + 1, w2 = 2 * g.rng[2] + 1;
    const float i_w2 = 1.0f / (float)w2, i_w1 = 1.0f / (float)w1;
    const double r_n0 = 1.0 / g.nb[0], r_n1 = 1.0 / g.nb[1], r_n2 = 1.0 / g.nb[2];
    const int nbox = w0 * w1 * w2;
    // reverse-index table: row stride nbox*cap entries; a smaller allocation is reported (sticky) and
    // the host grows it and reruns, like the other capacities
    const int ki = kslot[i];
    const bool t_ok = T != nullptr && (long long)nbox * cap <= (long long)t_stride && cap <= 4096 && maxnn <= 65535 &&
                      ki < cap;  // (ki >= cap: the bin overflowed, the host grows it and reruns the step)
    if (T != nullptr && !t_ok && ki < cap && lane == 0 && il == 0) atomicMax(&stat[2], cap <= 4096 ? nbox * cap : 0x7fffffff);
    int base = 0;
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    unsigned long long *keys = s_key[wave];
    for (int q0 = 0; q0 < nbox; q0 += 64) {
        // lane -> one neighbouring bin (image-aware)
        const int q = q0 + lane;
        int cntb = 0, sb = 0, code = 0;
        if (q < nbox) {
            const int qa = fdiv(q, w2, i_w2), qb = fdiv(qa, w1, i_w1);
            const int o2 = q - qa * w2 - g.rng[2], o1 = qa - qb * w1 - g.rng[1], o0 = qb - g.rng[0];
            const int t0 = b0 + o0, t1 = b1 + o1, t2 = b2 + o2;
            const int c0 = (int)floor((double)t0 * r_n0 + 1e-9), c1 = (int)floor((double)t1 * r_n1 + 1e-9),
                      c2 = (int)floor((double)t2 * r_n2 + 1e-9);
            const int nbin = ((t0 - c0 * g.nb[0]) * g.nb[1] + (t1 - c1 * g.nb[1])) * g.nb[2] + (t2 - c2 * g.nb[2]);
            sb = nbin * cap;
            cntb = min(bin_count[nbin], cap);
            code = (c0 & 0xff) | ((c1 & 0xff) << 8) | ((c2 & 0xff) << 16);
        }
        int incl = cntb;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int v = __shfl_up(incl, o, 64);
            if (lane >= o) incl += v;
        }
        const int total = __shfl(incl, 63, 64);
        s_start[wave][lane] = sb;
        s_pref[wave][lane] = incl - cntb;
        s_code[wave][lane] = code;
        if (lane == 0) s_pref[wave][64] = total;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int c0 = 0; c0 < total; c0 += 64) {
            const int c = c0 + lane;
            bool hit = false;
            int j = 0, f0 = 0, f1 = 0, f2 = 0, sj = 0, lo_hit = 0, k_hit = 0;
            if (c < total) {
                // largest b with pref[b] <= c  (empty bins share a prefix value: take the last)
                int lo = 0, hi = 63;
                while (lo < hi) {
                    const int mid = (lo + hi + 1) >> 1;
                    if (s_pref[wave][mid] <= c) lo = mid; else hi = mid - 1;
                }
                const int k = s_start[wave][lo] + (c - s_pref[wave][lo]);
                const int cd = s_code[wave][lo];
                lo_hit = lo;
                k_hit = c - s_pref[wave][lo];
                j = b_idx[k];
                sj = b_slot[k];
                f0 = (int)(int8_t)(cd & 0xff) - b_wrap[3 * k] + wi0;
                f1 = (int)(int8_t)((cd >> 8) & 0xff) - b_wrap[3 * k + 1] + wi1;
                f2 = (int)(int8_t)((cd >> 16) & 0xff) - b_wrap[3 * k + 2] + wi2;
                const double dx = b_pos[3 * k] - xi + (f0 * h[0] + f1 * h[3] + f2 * h[6]);
                const double dy = b_pos[3 * k + 1] - yi + (f0 * h[1] + f1 * h[4] + f2 * h[7]);
                const double dz = b_pos[3 * k + 2] - zi + (f0 * h[2] + f1 * h[5] + f2 * h[8]);
                const double rr = sqrt(dx * dx + dy * dy + dz * dz);
                hit = rr < rc && !(j == i && f0 == 0 && f1 == 0 && f2 == 0);
            }
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int slot = base + __popcll(m & lt);
                // key: neighbour index (24 bits), the image triple biased to sort as unsigned (24), species
                // slot (4), sweep ordinal (12: finds the candidate id again after the sort)
                const unsigned img = (unsigned)((f0 + 128) & 0xff) << 16 | (unsigned)((f1 + 128) & 0xff) << 8 |
                                     (unsigned)((f2 + 128) & 0xff);
                const unsigned long long key = ((unsigned long long)(unsigned)j << 40) | ((unsigned long long)img << 16) |
                                               ((unsigned long long)(unsigned)sj << 12) | (unsigned)(slot & 0xfff);
                const int qq = q0 + lo_hit, kk = k_hit;
                if (slot < NL_SORT_MAX) {
                    keys[slot] = key;
                    s_hq[wave][slot] = (qq << 12) | kk;
                } else if (slot < maxnn) {  // very long lists: keep sweep order beyond the sortable part
                    const size_t e = (size_t)i * maxnn + slot;
                    nbr_j[e] = j;
                    nbr_shift[e] = (f0 & 0xff) | ((f1 & 0xff) << 8) | ((f2 & 0xff) << 16) | (sj << 24);
                    if (t_ok) {
                        aux[e] = qq * cap + kk;
                        T[(size_t)j * t_stride + (size_t)(nbox - 1 - qq) * cap + ki] = (unsigned short)slot;
                    }
                }
            }
            base += __popcll(m);
        }
        __builtin_amdgcn_wave_barrier();
    }
    // bitonic sort of the first min(base, NL_SORT_MAX) keys, then coalesced write-out.  Up to 64 keys
    // (the usual case) sort in registers, one key per lane, partners by cross-lane shuffle: a third of
    // the instructions of the LDS network below and no barriers.
    const int ns = min(base, NL_SORT_MAX);
    if (ns <= 64) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        unsigned long long key = lane < ns ? keys[lane] : ~0ull;
#pragma unroll
        for (int k2 = 2; k2 <= 64; k2 <<= 1)
#pragma unroll
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                const unsigned lo = __shfl_xor((unsigned)key, j2, 64), hi = __shfl_xor((unsigned)(key >> 32), j2, 64);
                const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                const bool lower = (lane & j2) == 0, up = (lane & k2) == 0;
                // the lower lane of a pair keeps the smaller key in an ascending block
                const bool take_min = lower == up;
                key = take_min ? (other < key ? other : key) : (other > key ? other : key);
            }
        if (lane < ns) keys[lane] = key;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    } else {
        int np2 = 1;
        while (np2 < ns) np2 <<= 1;
        for (int t = ns + lane; t < np2; t += 64) keys[t] = ~0ull;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int k2 = 2; k2 <= np2; k2 <<= 1)
            for (int j2 = k2 >> 1; j2 > 0; j2 >>= 1) {
                for (int t = lane; t < np2; t += 64) {
                    const int p = t ^ j2;
                    if (p > t) {
                        const unsigned long long a0 = keys[t], a1 = keys[p];
                        const bool up = (t & k2) == 0;
                        if ((a0 > a1) == up) { keys[t] = a1; keys[p] = a0; }
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
    }
    for (int t = lane; t < ns && t < maxnn; t += 64) {
        const unsigned long long key = keys[t];
        const unsigned img = (unsigned)(key >> 16) & 0xffffffu;
        const int f0 = (int)((img >> 16) & 0xff) - 128, f1 = (int)((img >> 8) & 0xff) - 128, f2 = (int)(img & 0xff) - 128,
                  sj = (int)(key >> 12) & 0xf, j = (int)(key >> 40);
        const size_t e = (size_t)i * maxnn + t;
        nbr_j[e] = j;
        nbr_shift[e] = (f0 & 0xff) | ((f1 & 0xff) << 8) | ((f2 & 0xff) << 16) | (sj << 24);
        aux[e] = 0;
        if (t_ok) {
            const int hq = s_hq[wave][(int)key & 0xfff];
            const int qq = hq >> 12, kk = hq & 0xfff;
            aux[e] = qq * cap + kk;
            T[(size_t)j * t_stride + (size_t)(nbox - 1 - qq) * cap + ki] = (unsigned short)t;
        }
    }
    if (lane == 0) {
        nn[i] = base < maxnn ? base : maxnn;
        nn_local[il] = base < maxnn ? base : maxnn;
        nn_raw[il] = base;  // unclamped: finalize reduces the max for the overflow check
    }
}

void launch_neighbor_list(const NlParams &p, const int *perm, const double *pos_in, double *pos, const double *cell,
                          double rc, NlScratch s, int *nn, int *nn_local, int *nbr_j, int *nbr_shift,
                          double *zero_a, int n_zero_a, double *zero_b, int n_zero_b, int phase, hipStream_t st)
{
    if (p.N <= 0) return;
    BinArgs a = {};
    a.N = p.N; a.cap = s.cap; a.perm = perm; a.pos_in = pos_in; a.cell = cell; a.rc = rc;
    for (int k = 0; k < 3; k++) a.pbc[k] = p.pbc[k];
    a.grid = (NlGrid *)s.grid; a.pos = pos; a.bin_count = s.bin_count; a.b_idx = s.b_idx; a.b_pos = s.b_pos;
    a.b_wrap = s.b_wrap; a.b_slot = s.b_slot; a.slot = s.slot; a.bin_of = s.bin_of; a.kslot = s.kslot; a.wrap = s.wrap; a.stat = s.stat;
    a.zero_a = zero_a; a.n_zero_a = n_zero_a; a.zero_b = zero_b; a.n_zero_b = n_zero_b;
    if (phase != 2) hipLaunchKernelGGL(nl_bin_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, a);
    if (phase != 1 && p.count > 0)
        hipLaunchKernelGGL(nl_build_kernel, dim3((p.count + 3) / 4), dim3(256), 0, st, p.N, p.first,
                           p.stride > 0 ? p.stride : 1, p.count, pos, cell, rc, (const NlGrid *)s.grid, s.bin_of,
                           s.bin_count, s.cap, s.b_idx, s.b_pos, s.b_wrap, s.b_slot, s.wrap, p.maxnn, nn, nn_local, nbr_j,
                           nbr_shift, s.nn_raw, s.kslot, s.aux, s.T, s.t_stride, s.stat);
}
