// neighbor.hip — binning for the periodic neighbour list (gfx950).
//
// The list replaces ase.neighborlist.NeighborList as driven by descriptor/atoms.py:348-363,:402
// (radii rc/2, skin 0, bothways, self_interaction False): pair (i -> j, off) is kept iff
// |x_j - x_i + off.cell| < rc and (j,off) != (i,0); periodic self-images are kept.
// General triclinic cells, any pbc combination, atoms may sit outside the cell.
//
//   nl_bin_kernel (here)   256 atoms per workgroup: species-sort gather of the caller's positions, bin
//                   grid from the (device-resident) cell, and direct placement into fixed-capacity
//                   bins through one returning atomic per atom (binned copy: one 32-B record with
//                   position and index, one 8-B record with wrap and species slot).  Also clears the
//                   step's accumulators.  (An earlier count -> scan -> scatter binning was three
//                   launches, 21 us: every kernel boundary costs 2-5 us here.)
//   list build      in the forward kernel (descriptor.hip, nl_fwd_kernel): one wave64 per atom sweeps
//                   the (2R+1)^3 neighbouring bins, four bins x 16 slots per step, sorts the hits by
//                   (j, image) — the atomic placement order inside a bin is not reproducible, the
//                   sorted list is — and goes straight on to the descriptor.
// The bin counters are re-zeroed by the step's last kernel (finalize).
#include "sgpr_internal.h"

#define NL_MAX_BINS 4096
#define NL_SORT_MAX 256  // lists up to this length are sorted in LDS (longer ones keep sweep order)

__device__ __forceinline__ double det3d(const double *h)
{
    return h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) + h[2] * (h[3] * h[7] - h[4] * h[6]);
}

struct BinArgs {
    int N, cap, S;
    const int *perm;        // sorted -> caller (may be null: identity)
    const double *pos_in;   // caller order
    const double *cell;
    int pbc[3];
    double rc;
    NlGrid *grid;
    double *pos;            // [N][3] sorted order (out)
    int *bin_count;         // [NL_MAX_BINS] atoms per bin (zero on entry)
    BinRec *b_rec;          // [nbins][cap] position + sorted atom index
    BinAux *b_aux;          // [nbins][cap] wrap (floor of the fractional coordinate) + species slot
    const int *slot;        // [N] species slot by sorted index
    int *bin_of;            // [N]
    int *kslot;             // [N] slot of the atom inside its bin
    int *stat;              // [4]: [1] = largest bin population seen beyond cap, [3] = wrap beyond int16
    double *zero_a; int n_zero_a;   // accumulators to clear for this step
    double *zero_b; int n_zero_b;
    // Verlet candidates: rebuild decision of this step
    int *flag; int parity, force, fslot, fclear;
    double half_skin2;      // (skin / 2)^2
    double rc_list;         // rc + skin: the cutoff the candidates were built with
    double rc_phys;         // rc
    const double *pos0;     // [N][3] positions at the last rebuild (sorted order)
    const double *cell0;    // [9] cell at the last rebuild, [9..17] its inverse (both written by finalize on rebuild steps)
};

__device__ void nl_make_grid(const double *cell, const int *pbc, double rc, NlGrid &g, int *stat)
{
    double h[9];
    for (int k = 0; k < 9; k++) h[k] = cell[k];
    // Slabs and wires may come with a zero vector along an open direction (cell = [a, b, 0], pbc = TTF is
    // valid in ASE): complete such vectors orthogonally to the others before inverting (ase.geometry
    // complete_cell, which ASE's neighbour list applies), so the periodic directions keep their images.
    // A zero vector along a PERIODIC direction is an input error (stat[3] = 2).
    {
        int zero[3], nz = 0;
        for (int k = 0; k < 3; k++) {
            zero[k] = h[3 * k] * h[3 * k] + h[3 * k + 1] * h[3 * k + 1] + h[3 * k + 2] * h[3 * k + 2] < 1e-24;
            nz += zero[k];
            if (zero[k] && pbc[k] && stat) atomicMax(&stat[3], 2);
        }
        if (nz > 0 && nz < 3) {
            for (int k = 0; k < 3; k++) {
                if (!zero[k] || pbc[k]) continue;
                const double *p = h + 3 * ((k + 1) % 3), *q = h + 3 * ((k + 2) % 3);
                double v[3];
                if (!zero[(k + 1) % 3] && !zero[(k + 2) % 3]) {
                    v[0] = p[1] * q[2] - p[2] * q[1]; v[1] = p[2] * q[0] - p[0] * q[2]; v[2] = p[0] * q[1] - p[1] * q[0];
                } else {
                    // one vector only: any direction perpendicular to it (the other open axis follows next)
                    const double *w = zero[(k + 1) % 3] ? q : p;
                    const int a = fabs(w[0]) <= fabs(w[1]) && fabs(w[0]) <= fabs(w[2]) ? 0 : (fabs(w[1]) <= fabs(w[2]) ? 1 : 2);
                    double e[3] = {0.0, 0.0, 0.0};
                    e[a] = 1.0;
                    v[0] = w[1] * e[2] - w[2] * e[1]; v[1] = w[2] * e[0] - w[0] * e[2]; v[2] = w[0] * e[1] - w[1] * e[0];
                }
                const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                if (nv > 1e-12) {
                    h[3 * k] = v[0] / nv; h[3 * k + 1] = v[1] / nv; h[3 * k + 2] = v[2] / nv;
                    zero[k] = 0;
                }
            }
        }
    }
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
            g.w[k] = hgt[k] / g.nb[k];
        }
        // plane normals bc, ca, ab: orthogonal cells let the sweep bound the distance to a bin by the
        // Euclidean norm of the three plane gaps (otherwise only by the largest gap)
        const double d01 = bc[0] * ca[0] + bc[1] * ca[1] + bc[2] * ca[2], d02 = bc[0] * ab[0] + bc[1] * ab[1] + bc[2] * ab[2],
                     d12 = ca[0] * ab[0] + ca[1] * ab[1] + ca[2] * ab[2];
        const double n0 = bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2], n1 = ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2],
                     n2 = ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2];
        g.ortho = (d01 * d01 < 1e-20 * n0 * n1 && d02 * d02 < 1e-20 * n0 * n2 && d12 * d12 < 1e-20 * n1 * n2) ? 1 : 0;
    } else {
        // no usable cell (cluster): everything in one bin, no images
        for (int k = 0; k < 9; k++) g.inv[k] = 0.0;
        for (int k = 0; k < 3; k++) { g.nb[k] = 1; g.rng[k] = 0; g.w[k] = 0.0; }
        g.ortho = 0;
    }
    g.nbins = g.nb[0] * g.nb[1] * g.nb[2];
}

static_assert(sizeof(NlGrid) <= 128, "the cached grid records of nl_bin_kernel are 128 B + the cell");

__global__ __launch_bounds__(256) void nl_bin_kernel(BinArgs a)
{
    __shared__ NlGrid g;
    __shared__ double Aff[9];
    __shared__ double thr2;
    const int tid = threadIdx.x, wg = blockIdx.x;
    if (tid == 0) {
        // The grid of the previous step is kept with the cell it was made for (two records, by step parity: this
        // launch reads the one the previous launch wrote and writes the other — nobody writes what somebody reads).
        // An MD run at constant cell finds it there and skips the inversion, the three heights and their dozen
        // fp64 divisions and square roots on ONE lane while 255 wait at the barrier.
        char *cache = (char *)a.grid + 256;
        const NlGrid *g_prev = (const NlGrid *)(cache + 256 * (a.parity ^ 1));
        const double *c_prev = (const double *)(cache + 256 * (a.parity ^ 1) + 128);
        // EVERYTHING this lane compares is requested first and compared without short-circuits: `same && cell[k] ==
        // prev[k]` compiled to nine load -> wait -> branch round trips one behind the other (ISA), i.e. nine cold misses
        // in a row on the one lane the other 255 wait for at the barrier
        const NlGrid gp = *g_prev;
        double cc[9], cp[9], c0[18];
#pragma unroll
        for (int k = 0; k < 9; k++) { cc[k] = a.cell[k]; cp[k] = c_prev[k]; }
#pragma unroll
        for (int k = 0; k < 18; k++) c0[k] = a.cell0[k];
        int same_i = a.force == 0, cell_same_i = 1;
#pragma unroll
        for (int k = 0; k < 9; k++) { same_i &= (cc[k] == cp[k]) ? 1 : 0; cell_same_i &= (cc[k] == c0[k]) ? 1 : 0; }
        const bool same = same_i != 0, cell_same = cell_same_i != 0;
        if (same) g = gp;
        else nl_make_grid(a.cell, a.pbc, a.rc, g, wg == 0 ? a.stat : nullptr);
        // Candidates under a CHANGED cell (NPT: cl/md.py:147-150 strains the cell every step).  With A = h0^-1 h
        // (the affine map from the build-time cell to this one) and u_i = x_i - x_i0 A, a pair outside the
        // candidates (|r0| >= rc + skin) has |r| >= sigma_min(A) (rc + skin) - |u_i| - |u_j|, so the lists stay
        // complete while every |u_i| <= (sigma_min(A) (rc + skin) - rc) / 2; sigma_min(A) >= 1 - |A - I|_F.
        // Same cell bit for bit: A = I and the bound is skin / 2.  Open directions keep the strict rule.
        double thr = 0.5 * (a.rc_list - a.rc_phys);
        for (int k = 0; k < 9; k++) Aff[k] = (k % 4 == 0) ? 1.0 : 0.0;
        if (!cell_same) {
            if (a.pbc[0] && a.pbc[1] && a.pbc[2]) {
                const double *iv = c0 + 9;
                double fro = 0.0;
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) {
                        const double v = iv[3 * r] * cc[c] + iv[3 * r + 1] * cc[3 + c] + iv[3 * r + 2] * cc[6 + c];
                        Aff[3 * r + c] = v;
                        const double d = v - (r == c ? 1.0 : 0.0);
                        fro += d * d;
                    }
                thr = 0.5 * ((1.0 - sqrt(fro)) * a.rc_list - a.rc_phys);
                if (!(fro < 1.0)) thr = -1.0;  // (also catches NaN: a degenerate build-time cell)
            } else
                thr = -1.0;
        }
        thr2 = thr > 0.0 ? thr * thr : -1.0;
        if (wg == 0) {
            *a.grid = g;
            *(NlGrid *)(cache + 256 * a.parity) = g;
            for (int k = 0; k < 9; k++) ((double *)(cache + 256 * a.parity + 128))[k] = cc[k];
            // rebuild decision, part 1: forced, or the cell moved too far from the one the candidates were built
            // in (below); the flag of the step after the next is cleared (nobody reads it during this one)
            if (a.force != 0 || !(thr2 > 0.0)) atomicMax(&a.flag[a.fslot], 1);
            a.flag[a.fclear] = 0;
        }
    }
    const int gsz = gridDim.x * 256, gid = wg * 256 + tid;
    // this atom's position is requested before the barrier: the gather (perm -> pos_in) and the grid
    // set-up by thread 0 are independent latency chains
    const int i = gid;
    double x = 0.0, y = 0.0, z = 0.0, x0 = 0.0, y0 = 0.0, z0 = 0.0;
    int slot_i = 0;
    if (i < a.N) {
        const int c = a.perm ? a.perm[i] : i;
        x = a.pos_in[3 * c]; y = a.pos_in[3 * c + 1]; z = a.pos_in[3 * c + 2];
        slot_i = a.slot[i];
        x0 = a.pos0[3 * i]; y0 = a.pos0[3 * i + 1]; z0 = a.pos0[3 * i + 2];  // cold miss, off the chain behind the barrier
    }
    for (int k = gid; k < a.n_zero_a; k += gsz) a.zero_a[k] = 0.0;
    for (int k = gid; k < a.n_zero_b; k += gsz) a.zero_b[k] = 0.0;
    __syncthreads();
    if (i >= a.N) return;
    a.pos[3 * i] = x; a.pos[3 * i + 1] = y; a.pos[3 * i + 2] = z;
    {   // rebuild decision, part 2: has this atom moved more than half the skin since the candidates were built?
        const double dx = x - (x0 * Aff[0] + y0 * Aff[3] + z0 * Aff[6]), dy = y - (x0 * Aff[1] + y0 * Aff[4] + z0 * Aff[7]),
                     dz = z - (x0 * Aff[2] + y0 * Aff[5] + z0 * Aff[8]);
        if (!(dx * dx + dy * dy + dz * dz <= thr2)) atomicMax(&a.flag[a.fslot], 1);
    }
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
    }
    const int bin = (bidx[0] * g.nb[1] + bidx[1]) * g.nb[2] + bidx[2];
    a.bin_of[i] = bin;
    if (slot_i >= a.S) {  // a ghost (species outside the model's table, option "ignore_unknown_species"): nobody's neighbour
        a.kslot[i] = -1;
        return;
    }
    const int k = atomicAdd(&a.bin_count[(size_t)bin * SGPR_BIN_STRIDE], 1);
    a.kslot[i] = k;
    if (max(max(abs(w[0]), abs(w[1])), abs(w[2])) > 32767) atomicMax(&a.stat[3], 1);  // atoms > 32767 cells away
    if (k < a.cap) {
        const size_t e = (size_t)bin * a.cap + k;
        BinRec r;
        r.x = x; r.y = y; r.z = z; r.idx = i; r.pad = 0;
        a.b_rec[e] = r;
        BinAux ax;
        ax.w0 = (short)w[0]; ax.w1 = (short)w[1]; ax.w2 = (short)w[2]; ax.slot = (short)slot_i;
        a.b_aux[e] = ax;
    } else
        atomicMax(&a.stat[1], k + 1);  // rare: capacity exceeded, the host grows it and reruns
}

void launch_neighbor_bin(const NlParams &p, const int *perm, const double *pos_in, double *pos, const double *cell,
                         double rc, NlScratch s, double *zero_a, int n_zero_a, double *zero_b, int n_zero_b,
                         hipStream_t st)
{
    if (p.N <= 0) return;
    BinArgs a = {};
    a.N = p.N; a.cap = s.cap; a.S = p.S; a.perm = perm; a.pos_in = pos_in; a.cell = cell; a.rc = rc;
    for (int k = 0; k < 3; k++) a.pbc[k] = p.pbc[k];
    a.grid = s.grid; a.pos = pos; a.bin_count = s.bin_count; a.b_rec = s.b_rec; a.b_aux = s.b_aux;
    a.slot = s.slot; a.bin_of = s.bin_of; a.kslot = s.kslot; a.stat = s.stat;
    a.zero_a = zero_a; a.n_zero_a = n_zero_a; a.zero_b = zero_b; a.n_zero_b = n_zero_b;
    a.flag = s.flag; a.parity = s.parity; a.fslot = s.fslot; a.fclear = s.fclear; a.force = s.force; a.half_skin2 = 0.25 * s.skin * s.skin; a.pos0 = s.pos0;
    a.cell0 = s.cell0; a.rc_list = rc; a.rc_phys = rc - s.skin;
    hipLaunchKernelGGL(nl_bin_kernel, dim3((p.N + 255) / 256), dim3(256), 0, st, a);
}
