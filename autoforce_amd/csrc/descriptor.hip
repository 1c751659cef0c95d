// descriptor.hip — per-atom SeSoap descriptor, forward and reverse, hand-written for gfx950.
//
// What it computes (reference: descriptor/sesoap.py:161-260, descriptor/ylm.py:113-190,
// descriptor/cutoff.py:20-48, similarity/universal.py:100-107; reverse pass = the derivative
// torch.autograd takes at calculator/active.py:587-599):
//   x_j = r_j/u_j, d_j = |x_j|, g_j = [u d < rc](1 - u d/rc)^2 exp(-d^2/2), f_nj = g_j d_j^(2n)
//   c[s][n][lm] = sum_{j in species s} f_nj R_lm(x~_j)      (x~ = sheared x, ylm.py:10-23)
//   p[u][v][l]  = nnl sum_m c[u][l,m] c[v][l,m],  p^ = p/(|p|+eps)
// Mapping: ONE WAVE64 PER ATOM.  Neighbours are processed in tiles of 64 (lane = neighbour),
// staged in LDS; the c accumulation runs with lane = (n,lm) output slot (64 slots for the
// default lmax=nmax=3) reading the staged tile; the power spectrum and its norm are formed
// in-wave (DPP/shuffle reduction) and the packed row is written coalesced.
//
// Solid harmonics are evaluated in Cartesian form (polynomials in x,y,z): R_l0 = q_l0,
// R_lm^c = sqrt2 q_lm Re(x+iy)^m, R_lm^s = sqrt2 q_lm Im(x+iy)^m with the reference's
// recurrence coefficients for q (ylm.py:57-77,146-159).  Only sum_m c c* enters any output,
// so this real basis is equivalent to the reference's packed complex layout.
#include "sgpr_internal.h"

__constant__ HarmCoef c_hc;

void upload_harm_coef(const HarmCoef &hc) { (void)hipMemcpyToSymbol(HIP_SYMBOL(c_hc), &hc, sizeof(HarmCoef)); }

#define SQRT2 1.4142135623730951

__device__ __forceinline__ void wave_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// wave-uniform value into SGPRs (keeps cell / centre position out of the VGPR budget)
__device__ __forceinline__ double uniform(double v)
{
    const int lo = __builtin_amdgcn_readfirstlane(__double2loint(v));
    const int hi = __builtin_amdgcn_readfirstlane(__double2hiint(v));
    return __hiloint2double(hi, lo);
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------- solid harmonics
template <int LMAX>
struct Harm {
    static constexpr int L1 = LMAX + 1, LL = L1 * L1;
    double A[L1], B[L1], q[L1][L1];
    double x, y, z, rho;

    __device__ __forceinline__ void eval(double x_, double y_, double z_, double *Y)
    {
        x = x_; y = y_; z = z_;
        rho = x * x + y * y + z * z;
        A[0] = 1.0; B[0] = 0.0;
#pragma unroll
        for (int m = 1; m <= LMAX; m++) {
            A[m] = x * A[m - 1] - y * B[m - 1];
            B[m] = y * A[m - 1] + x * B[m - 1];
        }
        q[0][0] = c_hc.y00;
#pragma unroll
        for (int l = 1; l <= LMAX; l++) {
#pragma unroll
            for (int m = 0; m <= l - 2; m++)
                q[l][m] = c_hc.al[l][m] * (z * q[l - 1][m] + rho * c_hc.bl[l][m] * q[l - 2][m]);
            q[l][l - 1] = c_hc.cl[l] * z * q[l - 1][l - 1];
            q[l][l] = c_hc.dl[l] * q[l - 1][l - 1];
        }
#pragma unroll
        for (int l = 0; l <= LMAX; l++) {
            Y[l * l] = q[l][0];
#pragma unroll
            for (int m = 1; m <= l; m++) {
                Y[l * l + 2 * m - 1] = SQRT2 * q[l][m] * A[m];
                Y[l * l + 2 * m] = SQRT2 * q[l][m] * B[m];
            }
        }
    }

    // reverse pass of eval(): gY = dE/dY  ->  (gx,gy,gz) = dE/d(x,y,z); eval() must have run.
    __device__ __forceinline__ void backward(const double *gY, double &gx, double &gy, double &gz)
    {
        double gq[L1][L1], gA[L1], gB[L1];
#pragma unroll
        for (int m = 0; m <= LMAX; m++) { gA[m] = 0.0; gB[m] = 0.0; }
#pragma unroll
        for (int l = 0; l <= LMAX; l++) {
            gq[l][0] = gY[l * l];
#pragma unroll
            for (int m = 1; m <= l; m++) {
                const double gc = SQRT2 * gY[l * l + 2 * m - 1], gs = SQRT2 * gY[l * l + 2 * m];
                gq[l][m] = gc * A[m] + gs * B[m];
                gA[m] += q[l][m] * gc;
                gB[m] += q[l][m] * gs;
            }
        }
        double gzz = 0.0, grho = 0.0;
#pragma unroll
        for (int l = LMAX; l >= 1; l--) {
            gq[l - 1][l - 1] += c_hc.dl[l] * gq[l][l];
            gzz += c_hc.cl[l] * q[l - 1][l - 1] * gq[l][l - 1];
            gq[l - 1][l - 1] += c_hc.cl[l] * z * gq[l][l - 1];
#pragma unroll
            for (int m = 0; m <= l - 2; m++) {
                const double g = c_hc.al[l][m] * gq[l][m];
                gzz += q[l - 1][m] * g;
                grho += c_hc.bl[l][m] * q[l - 2][m] * g;
                gq[l - 1][m] += z * g;
                gq[l - 2][m] += rho * c_hc.bl[l][m] * g;
            }
        }
        double gxx = 0.0, gyy = 0.0;
#pragma unroll
        for (int m = LMAX; m >= 1; m--) {
            gxx += gA[m] * A[m - 1] + gB[m] * B[m - 1];
            gyy += -gA[m] * B[m - 1] + gB[m] * A[m - 1];
            gA[m - 1] += x * gA[m] + y * gB[m];
            gB[m - 1] += -y * gA[m] + x * gB[m];
        }
        gx = gxx + 2.0 * x * grho;
        gy = gyy + 2.0 * y * grho;
        gz = gzz + 2.0 * z * grho;
    }
};

// ---------------------------------------------------------------- kernel arguments
struct DescArgs {
    int N, Nall, first, stride, maxnn, S, Dc, Dpad, CS;
    int phase;              // reverse pass: 0 = dE/dc then pair kernel, 1 = dE/dc only, 2 = pair kernel only
    double rc;
    const double *pos;      // [Nall][3] (sorted order)
    const double *cell;     // [9]
    const int *slot;        // [Nall]
    const double *radii;    // [S] (device; ENV path)
    double radii_v[SGPR_MAX_S];  // the same by value: unit lookup is a select chain, no load
    const int *nn;          // [Nall]
    const int *nbr_j;       // [Nall][maxnn]
    const int *nbr_shift;   // [Nall][maxnn]
    const int64_t *env_ptr; // ENV mode: [N+1]
    const int *env_slot;    // ENV mode
    const double *env_r;    // ENV mode [..][3]
    const PackEntry *pack;  // [Dc]
    double *Pn;             // [N][Dpad]
    double *norm;           // [N]
    double *C;              // [N][CS]
    double *dC;             // [N][CS] dE/dc (reverse pass)
    int *shear;             // [N]
    const double *W;        // backward: [N][Dpad]
    double *Fnbr;           // backward: [Nall][3] (atomic)
    double *Fself;          // backward: [Nall][3] (plain store, one writer)
    double *vir_part;       // backward: [gridDim][4 waves][9]
};

// length unit of species slot s (wave-divergent s): select chain over the by-value table
template <int ST>
__device__ __forceinline__ double unit_of(const DescArgs &a, int s)
{
    double u = a.radii_v[0];
#pragma unroll
    for (int q = 1; q < ST; q++) u = (s == q) ? a.radii_v[q] : u;
    return u;
}

// neighbour t of atom (global sorted index gi / local index ia): displacement, species slot
template <bool ENV>
__device__ __forceinline__ void load_neighbor(const DescArgs &a, int gi, int ia, int t, const double *pi,
                                              const double *cell, double r[3], int &s, int &j)
{
    if constexpr (ENV) {
        const int64_t e = a.env_ptr[ia] + t;
        r[0] = a.env_r[3 * e]; r[1] = a.env_r[3 * e + 1]; r[2] = a.env_r[3 * e + 2];
        s = a.env_slot[e];
        j = -1;
    } else {
        const size_t e = (size_t)gi * a.maxnn + t;
        j = a.nbr_j[e];
        const int code = a.nbr_shift[e];
        const double s0 = (double)(int)(int8_t)(code & 0xff);
        const double s1 = (double)(int)(int8_t)((code >> 8) & 0xff);
        const double s2 = (double)(int)(int8_t)((code >> 16) & 0xff);
#pragma unroll
        for (int k = 0; k < 3; k++)
            r[k] = a.pos[3 * (size_t)j + k] - pi[k] + (s0 * cell[k] + s1 * cell[3 + k] + s2 * cell[6 + k]);
        s = (code >> 24) & 0xff;  // species slot of the neighbour (packed by nl_build)
    }
}

// radial weights f_n = g d^(2n) (and dg/dd for the reverse pass)
template <int NMAX>
__device__ __forceinline__ void radial(double d, double u, double rc, double *f, double &g, double &dg)
{
    // reciprocals instead of fp64 divisions (each is a ~12-instruction sequence on the VALU): the
    // pair term had 14 of them per evaluation
    const double ud = u * d;
    const double irc = 1.0 / rc;  // wave-uniform
    const double step = ud < rc ? 1.0 : 0.0;
    const double qq = 1.0 - ud * irc;
    const double cut = step * qq * qq;
    const double dcut = step * (-2.0 * qq * irc) * u;
    const double ex = exp(-0.5 * d * d);
    g = cut * ex;
    dg = dcut * ex - d * g;
    const double rho = d * d;
    double pw = g;
#pragma unroll
    for (int n = 0; n <= NMAX; n++) { f[n] = pw; pw *= rho; }
}

template <int LMAX, int NMAX>
struct WaveLds {
    static constexpr int L1 = LMAX + 1, N1 = NMAX + 1, LL = L1 * L1, LLP = LL + 1, NSLOT = N1 * LL;
    // doubles per wave for the neighbour tile
    // neighbours per tile of the forward kernel: 48, not 64 — a 64-neighbour tile costs 50 KB of LDS per
    // workgroup (3 workgroups per CU, so 1024 workgroups run as 1.33 rounds); 48 fits four per CU and
    // the whole grid is resident at once.  Lists longer than 48 take another tile.
    static constexpr int CH = 48;
    static constexpr int TILE_D = CH * N1 + CH * LLP;
};

// =========================================================================== forward
template <int LMAX, int NMAX, int ST, bool ENV>
__global__ __launch_bounds__(256) void desc_fwd_kernel(DescArgs a)
{
    using WL = WaveLds<LMAX, NMAX>;
    constexpr int N1 = WL::N1, LL = WL::LL, LLP = WL::LLP, NSLOT = WL::NSLOT;
    constexpr int SPL = (NSLOT + 63) / 64;
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ia = blockIdx.x * 4 + wave;
    if (ia >= a.N) return;
    const int gi = a.first + ia * a.stride;
    constexpr int CH = WL::CH;
    const int perwave = WL::TILE_D + ST * NSLOT + 64 / 2;  // + 64 ints
    double *fl = smem + (size_t)wave * perwave;  // [CH][N1]
    double *Yl = fl + CH * N1;                   // [CH][LLP]
    double *cl = Yl + CH * LLP;                  // [ST][NSLOT]
    int *sl = (int *)(cl + ST * NSLOT);          // [64]

    int nn;
    double pi[3] = {0, 0, 0}, cell[9];
    if constexpr (ENV) {
        nn = (int)(a.env_ptr[ia + 1] - a.env_ptr[ia]);
    } else {
        nn = a.nn[gi];
#pragma unroll
        for (int k = 0; k < 3; k++) pi[k] = a.pos[3 * (size_t)gi + k];
#pragma unroll
        for (int k = 0; k < 9; k++) cell[k] = a.cell[k];
    }

    // Does any neighbour sit inside the z cone? (ylm.py:10-23: then the whole environment shears.)
    // Environments that fit one tile decide it inside the main pass; larger ones
    // need a pass of their own first.
    bool shear = false;
    if (nn > CH) {
        bool near = false;
        for (int t0 = 0; t0 < nn; t0 += 64) {
            const int t = t0 + lane;
            if (t < nn) {
                double r[3]; int s, j;
                load_neighbor<ENV>(a, gi, ia, t, pi, cell, r, s, j);
                const double u = unit_of<ST>(a, s);
                const double tol = SGPR_TINY_ANGLE * fabs(r[2] / u);
                near |= (fabs(r[0] / u) < tol) && (fabs(r[1] / u) < tol);
            }
        }
        shear = __any(near);
    }

    double acc[ST][SPL];
#pragma unroll
    for (int s = 0; s < ST; s++)
#pragma unroll
        for (int k = 0; k < SPL; k++) acc[s][k] = 0.0;

    for (int t0 = 0; t0 < nn; t0 += CH) {
        const int cnt = min(CH, nn - t0);
        const int t = lane < cnt ? t0 + lane : nn;  // lanes past the tile sit the neighbour phase out
        wave_sync();
        double r[3] = {1.0, 0.0, 0.0};
        int s = 0, j = 0;
        if (t < nn) load_neighbor<ENV>(a, gi, ia, t, pi, cell, r, s, j);
        const double u = unit_of<ST>(a, s);
        const double iu = 1.0 / u;
        const double x = r[0] * iu, y = r[1] * iu, z = r[2] * iu;
        if (nn <= CH) {
            const double tol = SGPR_TINY_ANGLE * fabs(z);
            shear = __any(t < nn && fabs(x) < tol && fabs(y) < tol);
        }
        const double ang = shear ? SGPR_TINY_ANGLE : 0.0;
        if (t < nn) {
            const double d = sqrt(x * x + y * y + z * z);
            double f[N1], g, dg;
            radial<NMAX>(d, u, a.rc, f, g, dg);
            double Y[LL];
            Harm<LMAX> h;
            h.eval(x, y - ang * z, ang * y + z, Y);
#pragma unroll
            for (int n = 0; n < N1; n++) fl[lane * N1 + n] = f[n];
#pragma unroll
            for (int k = 0; k < LL; k++) Yl[lane * LLP + k] = Y[k];
            sl[lane] = s;
        }
        wave_sync();
        // lane = output slot (n,lm): c[s][slot] += f[t][n] * Y[t][lm]
        // The device neighbour list is sorted by (sorted atom index, image) and atoms are sorted by
        // species, so the species of a tile's neighbours is non-decreasing: one uniform loop per
        // species segment, no per-neighbour select (28 % fewer wave instructions in this kernel).
        // Caller-ordered environments (ENV) and anything non-monotone take the select loop.
        const int sv = t < nn ? s : ST;
        const int sprev = __shfl_up(sv, 1, 64);
        const bool sorted = !ENV && !__any(lane > 0 && lane < cnt && sprev > sv);
        if (sorted) {
            int beg = 0;
#pragma unroll
            for (int q = 0; q < ST; q++) {
                const int end = beg + __popcll(__ballot(t < nn && s == q));
#pragma unroll 4
                for (int tt = beg; tt < end; tt++) {
#pragma unroll
                    for (int k = 0; k < SPL; k++) {
                        const int slot = lane + 64 * k;
                        if (SPL * 64 == NSLOT || slot < NSLOT)
                            acc[q][k] += fl[tt * N1 + slot / LL] * Yl[tt * LLP + slot % LL];
                    }
                }
                beg = end;
            }
        } else {
#pragma unroll 4
            for (int tt = 0; tt < cnt; tt++) {
                const int s = __builtin_amdgcn_readfirstlane(sl[tt]);
#pragma unroll
                for (int k = 0; k < SPL; k++) {
                    const int slot = lane + 64 * k;
                    if (SPL * 64 == NSLOT || slot < NSLOT) {
                        const double v = fl[tt * N1 + slot / LL] * Yl[tt * LLP + slot % LL];
#pragma unroll
                        for (int q = 0; q < ST; q++)
                            if (s == q) acc[q][k] += v;
                    }
                }
            }
        }
    }
    wave_sync();
#pragma unroll
    for (int s = 0; s < ST; s++)
#pragma unroll
        for (int k = 0; k < SPL; k++) {
            const int slot = lane + 64 * k;
            if (SPL * 64 == NSLOT || slot < NSLOT) {
                cl[s * NSLOT + slot] = acc[s][k];
                if (a.C && s < a.S) a.C[(size_t)ia * a.CS + s * NSLOT + slot] = acc[s][k];
            }
        }
    wave_sync();
    // packed power spectrum: entry e = pair(u<=v)*L1 + l : coef * sum_{lm in l} c[u][lm] c[v][lm].
    // One lane per (u,v) pair, the l shells and their m sums statically unrolled (offsets into the
    // two LDS rows are compile-time constants; the per-entry form spent most of its instructions on
    // index arithmetic and a dynamic m loop).
    double nrm2 = 0.0;
    constexpr int L1 = LMAX + 1;
    constexpr int UMAX = ST * N1;
    constexpr int MAXP = ((UMAX * (UMAX + 1)) / 2 + 63) / 64;
    const int npair = a.Dc / L1;
    double pv[MAXP][L1];
#pragma unroll
    for (int k = 0; k < MAXP; k++) {
        const int pr = lane + 64 * k;
#pragma unroll
        for (int l = 0; l < L1; l++) pv[k][l] = 0.0;
        if (pr < npair) {
            const PackEntry p0 = a.pack[pr * L1];
            const double *cu = cl + (p0.u / N1) * NSLOT + (p0.u % N1) * LL;
            const double *cv = cl + (p0.v / N1) * NSLOT + (p0.v % N1) * LL;
#pragma unroll
            for (int l = 0; l < L1; l++) {
                double sacc = 0.0;
#pragma unroll
                for (int mm = 0; mm < 2 * l + 1; mm++) sacc += cu[l * l + mm] * cv[l * l + mm];
                pv[k][l] = sacc * a.pack[pr * L1 + l].coef;
                nrm2 += pv[k][l] * pv[k][l];
            }
        }
    }
    nrm2 = wave_sum(nrm2);
    const double nrm = sqrt(nrm2);
    const double inv = nn > 0 ? 1.0 / (nrm + SGPR_EPS) : 0.0;
#pragma unroll
    for (int k = 0; k < MAXP; k++) {
        const int pr = lane + 64 * k;
        if (pr < npair) {
#pragma unroll
            for (int l = 0; l < L1; l++) a.Pn[(size_t)ia * a.Dpad + pr * L1 + l] = pv[k][l] * inv;
        }
    }
    for (int e = a.Dc + lane; e < a.Dpad; e += 64) a.Pn[(size_t)ia * a.Dpad + e] = 0.0;
    if (lane == 0) {
        a.norm[ia] = nn > 0 ? nrm : 0.0;
        if (a.shear) a.shear[ia] = shear ? 1 : 0;
    }
}

// =========================================================================== backward
// Reverse pass in two kernels.
//  desc_dc_kernel    per atom: G~ = dE/dp^ (packed, from the W GEMM) -> dE/dp -> dE/dc, stored.
//  desc_pair_kernel  per atom j (one wave), lane = neighbour t at r = x_i - x_j + off.cell:
//      g_t  = dE_j/dr_jt            (own environment, dE/dc of j from LDS)
//    MIRROR (single process): the same lane also evaluates the mirrored pair, i.e. atom j seen
//      from i's environment at -r, with dE/dc of i gathered from global memory:
//      g'_t = dE_i/dr_ij(-r)   ->   F_j = sum_t (g_t - g'_t)     no atomics, deterministic.
//    !MIRROR (atoms sharded over ranks: dE/dc of remote atoms is not available): F_i -= g_t by
//      fp64 atomics into the all-atom force buffer that the ranks then all-reduce.
//    Virial: sum_t r (x) g_t from the own terms (each ordered pair once).
template <int LMAX, int NMAX, int ST>
__global__ __launch_bounds__(256) void desc_dc_kernel(DescArgs a)
{
    using WL = WaveLds<LMAX, NMAX>;
    constexpr int N1 = WL::N1, LL = WL::LL, NSLOT = WL::NSLOT;
    constexpr int SPL = (NSLOT + 63) / 64;
    extern __shared__ double smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ia = blockIdx.x * 4 + wave;
    if (ia >= a.N) return;
    const int Ur = a.S * N1;  // channels of the packed layout (pack table built with the real S)
    // Up to 4 species the packed gradient is expanded into the full symmetric [u][v][l] array in LDS,
    // which turns the contraction below into stride-1 addressing; 8 slots would need 32 KB per wave
    // and keep the packed form with computed pair indices.
    constexpr bool EXPAND = ST <= 4;
    constexpr int UT = ST * N1, L1 = LMAX + 1;
    constexpr int GS = EXPAND ? UT * UT * L1 : 0;
    const int perwave = ST * NSLOT + (EXPAND ? GS : a.Dpad);
    double *cl = smem + (size_t)wave * perwave;  // [ST][NSLOT]  c  ( = [u][lm], u = s*N1+n )
    double *gl = cl + ST * NSLOT;                // EXPAND: [UT][UT][L1] else [Dpad]:  dE/dp~ * coef * (1 or 2)
    const int gi = a.first + ia * a.stride;
    const int nn = a.nn[gi];
    const double nrm = a.norm[ia];
    double *dC = a.dC + (size_t)ia * a.CS;
    if (!(nn > 0 && nrm > 0.0)) {
        for (int k = lane; k < a.CS; k += 64) dC[k] = 0.0;
        return;
    }
    const double sden = nrm + SGPR_EPS;
    constexpr int MAXE = ((UT * (UT + 1)) / 2 * L1 + 63) / 64;
    const double *Wi = a.W + (size_t)ia * a.Dpad, *Pi = a.Pn + (size_t)ia * a.Dpad;
    if constexpr (MAXE <= 10) {
        // all global reads of this atom are issued up front (W, p^, pack entries, c): one memory
        // latency instead of three dependent ones
        double wv[MAXE], pv[MAXE];
        PackEntry pe[MAXE];
    #pragma unroll
        for (int k = 0; k < MAXE; k++) {
            const int e = lane + 64 * k;
            const bool in = e < a.Dc;
            wv[k] = in ? Wi[e] : 0.0;
            pv[k] = in ? Pi[e] : 0.0;
            pe[k] = a.pack[in ? e : 0];
        }
        double cin[ST][SPL];
    #pragma unroll
        for (int s = 0; s < ST; s++)
    #pragma unroll
            for (int k = 0; k < SPL; k++) {
                const int slot = lane + 64 * k;
                cin[s][k] = (s < a.S && (SPL * 64 == NSLOT || slot < NSLOT)) ? a.C[(size_t)ia * a.CS + s * NSLOT + slot] : 0.0;
            }
        // dE/dp~ = (W - p^ (p^.W) sden/nrm) / sden
        double pw = 0.0;
    #pragma unroll
        for (int k = 0; k < MAXE; k++) pw += wv[k] * pv[k];
        pw = wave_sum(pw);
        const double corr = pw * sden / nrm;
        const double isden = 1.0 / sden;
    #pragma unroll
        for (int k = 0; k < MAXE; k++) {
            const int e = lane + 64 * k;
            if (e < a.Dc) {
                const double gv = (wv[k] - pv[k] * corr) * isden * pe[k].coef * (pe[k].u == pe[k].v ? 2.0 : 1.0);
                if constexpr (EXPAND) {
                    gl[(pe[k].u * UT + pe[k].v) * L1 + pe[k].l] = gv;
                    gl[(pe[k].v * UT + pe[k].u) * L1 + pe[k].l] = gv;
                } else
                    gl[e] = gv;
            }
        }
    #pragma unroll
        for (int s = 0; s < ST; s++)
    #pragma unroll
            for (int k = 0; k < SPL; k++) {
                const int slot = lane + 64 * k;
                if (SPL * 64 == NSLOT || slot < NSLOT) cl[s * NSLOT + slot] = cin[s][k];
            }

    } else {
        // many species / high lmax: too many entries per lane to hold in registers
        double pw = 0.0;
        for (int e = lane; e < a.Dc; e += 64) pw += Wi[e] * Pi[e];
        pw = wave_sum(pw);
        const double corr = pw * sden / nrm;
        for (int e = lane; e < a.Dc; e += 64) {
            const PackEntry pe = a.pack[e];
            const double gv = (Wi[e] - Pi[e] * corr) / sden * pe.coef * (pe.u == pe.v ? 2.0 : 1.0);
            if constexpr (EXPAND) {
                gl[(pe.u * UT + pe.v) * L1 + pe.l] = gv;
                gl[(pe.v * UT + pe.u) * L1 + pe.l] = gv;
            } else
                gl[e] = gv;
        }
#pragma unroll
        for (int s = 0; s < ST; s++)
#pragma unroll
            for (int k = 0; k < SPL; k++) {
                const int slot = lane + 64 * k;
                if (SPL * 64 == NSLOT || slot < NSLOT)
                    cl[s * NSLOT + slot] = s < a.S ? a.C[(size_t)ia * a.CS + s * NSLOT + slot] : 0.0;
            }
    }
    wave_sync();
    // dE/dc[u][lm] = sum_v G[u][v][l] c[v][lm]
#pragma unroll
    for (int s = 0; s < ST; s++)
#pragma unroll
        for (int k = 0; k < SPL; k++) {
            const int slot = lane + 64 * k;
            if ((SPL * 64 == NSLOT || slot < NSLOT) && s < a.S) {
                const int n = slot / LL, lm = slot % LL;
                int l = 0;
#pragma unroll
                for (int q = 1; q <= LMAX; q++) l += (lm >= q * q) ? 1 : 0;
                const int u = s * N1 + n;
                double d = 0.0;
                if constexpr (EXPAND) {
                    const double *gu = gl + u * UT * L1 + l;
#pragma unroll 4
                    for (int v = 0; v < Ur; v++) d += gu[v * L1] * cl[v * LL + lm];
                } else {
                    for (int v = 0; v < Ur; v++) {
                        const int lo = min(u, v), hi = max(u, v);
                        const int pair = lo * Ur - (lo * (lo - 1)) / 2 + (hi - lo);
                        d += gl[pair * L1 + l] * cl[v * LL + lm];
                    }
                }
                dC[s * NSLOT + slot] = d;
            }
        }
}

// dE/dr of ONE pair term: neighbour at displacement r (unscaled), unit u, environment shear `ang`,
// dc = dE/dc[species slot of that neighbour][n][lm] of the environment's centre.
template <int LMAX, int NMAX, typename Fetch>
__device__ __forceinline__ void pair_grad(const double r[3], double u, double rc, double ang, Fetch fetch,
                                          double gr[3])
{
    constexpr int N1 = NMAX + 1, LL = (LMAX + 1) * (LMAX + 1);
    const double iu = 1.0 / u;
    const double x = r[0] * iu, y = r[1] * iu, z = r[2] * iu;
    const double d = sqrt(x * x + y * y + z * z);
    const double id = 1.0 / d;
    double f[N1], g, dg;
    radial<NMAX>(d, u, rc, f, g, dg);
    double Y[LL], gY[LL];
    Harm<LMAX> h;
    h.eval(x, y - ang * z, ang * y + z, Y);
    double dEdd = 0.0;
#pragma unroll
    for (int k = 0; k < LL; k++) gY[k] = 0.0;
    const double rho = d * d;
    double rpow = 1.0;  // rho^n
    // rolled over the radial channels on purpose: fully unrolled, the scheduler hoisted all 64
    // coefficient loads and the kernel needed > 256 VGPRs (1 wave per SIMD)
#pragma unroll 1
    for (int n = 0; n < N1; n++) {
        const double *dcn = fetch(n);  // channel n of dE/dc (LDS)
        const double fn = g * rpow;    // f_n = g d^(2n)
        double dEdf = 0.0;
#pragma unroll
        for (int k = 0; k < LL; k++) {
            const double dck = dcn[k];
            dEdf += dck * Y[k];
            gY[k] += fn * dck;
        }
        // d f_n/dd = dg rho^n + g 2n d^(2n-1)
        const double dfn = dg * rpow + (n ? g * 2.0 * n * rpow * id : 0.0);
        dEdd += dEdf * dfn;
        rpow *= rho;
    }
    double gxs, gys, gzs;
    h.backward(gY, gxs, gys, gzs);
    // inverse shear (ylm.py:203-213) + radial part, then 1/u
    const double rad = dEdd * id;
    gr[0] = (gxs + rad * x) * iu;
    gr[1] = (gys + ang * gzs + rad * y) * iu;
    gr[2] = (-ang * gys + gzs + rad * z) * iu;
}

// The mirrored pass of desc_pair_kernel: as pair_grad, with the coefficient rows of the 64 lanes'
// neighbours (dC[j][slot][n][:], one row per lane) gathered cooperatively — GR lanes per row, whole
// cache lines per group of lanes — one radial channel at a time, and SOFTWARE-PIPELINED: channel
// n+1 is in flight in registers while channel n is contracted out of LDS.  (As a lambda capturing
// the register array this ended in scratch; here the array is a local of the function that owns
// the rolled loop.)
template <int LMAX, int NMAX>
__device__ __forceinline__ void pair_grad_gathered(const double r[3], double u, double rc, double ang,
                                                   const double *dC, int CS, int chan_off /*slot*NSLOT*/, int j,
                                                   double *stage, int lane, double gr[3])
{
    constexpr int N1 = NMAX + 1, LL = (LMAX + 1) * (LMAX + 1);
    constexpr int SP = LL + 2, G = (LL % 2 == 0) ? 2 : 1, GR = LL / G;
    const double iu = 1.0 / u;
    const double x = r[0] * iu, y = r[1] * iu, z = r[2] * iu;
    const double d = sqrt(x * x + y * y + z * z);
    const double id = 1.0 / d;
    double f[N1], g, dg;
    radial<NMAX>(d, u, rc, f, g, dg);
    double Y[LL], gY[LL];
    Harm<LMAX> h;
    h.eval(x, y - ang * z, ang * y + z, Y);
    double dEdd = 0.0;
#pragma unroll
    for (int k = 0; k < LL; k++) gY[k] = 0.0;
    const double rho = d * d;
    double rpow = 1.0;
    double pre[GR][G];
#define SGPR_GATHER(N)                                                                           \
    _Pragma("unroll") for (int q = 0; q < GR; q++) {                                            \
        const int idx = q * 64 + lane;                                                          \
        const int jr = __shfl(j, idx / GR, 64);                                                 \
        const double *src = dC + (size_t)jr * CS + chan_off + (N) * LL + (idx % GR) * G;         \
        if constexpr (G == 2) {                                                                 \
            const double2 t2 = *(const double2 *)src;                                           \
            pre[q][0] = t2.x; pre[q][G - 1] = t2.y;                                             \
        } else                                                                                  \
            pre[q][0] = *src;                                                                   \
    }
    SGPR_GATHER(0)
#pragma unroll 1
    for (int n = 0; n < N1; n++) {
        wave_sync();
#pragma unroll
        for (int q = 0; q < GR; q++) {
            const int idx = q * 64 + lane;
            double *dst = stage + (idx / GR) * SP + (idx % GR) * G;
            if constexpr (G == 2)
                *(double2 *)dst = make_double2(pre[q][0], pre[q][G - 1]);
            else
                *dst = pre[q][0];
        }
        wave_sync();
        if (n + 1 < N1) { SGPR_GATHER(n + 1) }
        const double *dcn = stage + lane * SP;
        const double fn = g * rpow;
        double dEdf = 0.0;
#pragma unroll
        for (int k = 0; k < LL; k++) {
            const double dck = dcn[k];
            dEdf += dck * Y[k];
            gY[k] += fn * dck;
        }
        const double dfn = dg * rpow + (n ? g * 2.0 * n * rpow * id : 0.0);
        dEdd += dEdf * dfn;
        rpow *= rho;
    }
#undef SGPR_GATHER
    double gxs, gys, gzs;
    h.backward(gY, gxs, gys, gzs);
    const double rad = dEdd * id;
    gr[0] = (gxs + rad * x) * iu;
    gr[1] = (gys + ang * gzs + rad * y) * iu;
    gr[2] = (-ang * gys + gzs + rad * z) * iu;
}

// PASS 0: own terms + atomic scatter (sharded form); 3: own then mirrored terms in one launch (the
// single-process form).  (The two halves as separate launches measured 17.5 + 17.5 us against 29.8 us
// fused: both halves still need ~220 VGPRs, so splitting buys no occupancy.)
template <int LMAX, int NMAX, int ST, int PASS>
__global__ __launch_bounds__(256, 2) void desc_pair_kernel(DescArgs a)
{
    using WL = WaveLds<LMAX, NMAX>;
    constexpr int NSLOT = WL::NSLOT, LL = WL::LL;
    constexpr int SP = LL + 2;                // staged row stride (doubles): 16-B aligned, conflict-free
    constexpr int G = (LL % 2 == 0) ? 2 : 1;  // doubles per load granule
    constexpr int GR = LL / G;                // granules per row
    extern __shared__ double smem[];
    __shared__ double vred[4][9];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int ia = blockIdx.x * 4 + wave;
    double *dcl = smem + (size_t)wave * (ST * NSLOT + 64 * SP + 12 * 64);  // [S][NSLOT] dE/dc of this atom
    double *stage = dcl + ST * NSLOT;                            // [64][SP] one channel of 64 mirrored rows
    double *lv = stage + 64 * SP;                                // [9][64] per-lane virial accumulators
#pragma unroll
    for (int k = 0; k < 9; k++) lv[k * 64 + lane] = 0.0;
    double fsum[3] = {0, 0, 0};
    const bool active = ia < a.N;
    const int gi = a.first + (active ? ia : 0) * a.stride;
    const int nn = active ? a.nn[gi] : 0;
    constexpr bool MIRROR = PASS == 3;
    if (active && nn > 0) {
        for (int k = lane; k < a.CS; k += 64) dcl[k] = a.dC[(size_t)ia * a.CS + k];
        wave_sync();
        const double ang = a.shear[ia] ? SGPR_TINY_ANGLE : 0.0;
        const int sc = a.slot[gi];
        const double uc = unit_of<ST>(a, sc);
        double pi[3], cell[9];
#pragma unroll
        for (int k = 0; k < 3; k++) pi[k] = uniform(a.pos[3 * (size_t)gi + k]);
#pragma unroll
        for (int k = 0; k < 9; k++) cell[k] = uniform(a.cell[k]);
        // pass 1: own terms g_t = dE_j/dr_jt (dE/dc of this atom from LDS)
        for (int t0 = 0; t0 < nn; t0 += 64) {
            const int t = t0 + lane;
            if (t < nn) {
                double r[3], gr[3];
                int s, j;
                load_neighbor<false>(a, gi, ia, t, pi, cell, r, s, j);
                pair_grad<LMAX, NMAX>(r, unit_of<ST>(a, s), a.rc, ang,
                                      [&](int n) { return (const double *)(dcl + s * NSLOT + n * LL); }, gr);
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int q = 0; q < 3; q++) lv[(3 * p + q) * 64 + lane] += r[p] * gr[q];
#pragma unroll
                for (int k = 0; k < 3; k++) fsum[k] += gr[k];
                if constexpr (PASS == 0) {
#pragma unroll
                    for (int k = 0; k < 3; k++) unsafeAtomicAdd(&a.Fnbr[3 * (size_t)j + k], -gr[k]);
                }
            }
        }
        if constexpr (MIRROR) {
            // pass 2: mirrored terms g'_t = dE_i/dr_ij(-r): this atom as a neighbour of i, our
            // species slot, i's shear state.  The 64 rows dC[i][our slot][n][:] are fetched
            // cooperatively, one radial channel at a time, GR lanes per row (whole cache lines per
            // quad of lanes), and handed to their lanes through LDS: a per-lane 512-B gather costs
            // 4x the TA cycles.
            for (int t0 = 0; t0 < nn; t0 += 64) {
                const int t = t0 + lane;
                const bool on = t < nn;
                double r[3] = {-1.0, 0.0, 0.0}, gm[3];
                int s = sc, j = gi;
                if (on) {
                    load_neighbor<false>(a, gi, ia, t, pi, cell, r, s, j);
                    r[0] = -r[0]; r[1] = -r[1]; r[2] = -r[2];
                }
                const double angm = a.shear[j] ? SGPR_TINY_ANGLE : 0.0;
                pair_grad_gathered<LMAX, NMAX>(r, uc, a.rc, angm, a.dC, a.CS, sc * NSLOT, j, stage, lane, gm);
                if (on) {
#pragma unroll
                    for (int k = 0; k < 3; k++) fsum[k] -= gm[k];
                }
            }
        }
    }
    // 12 wave sums (force 3 + virial 9) through LDS: the per-lane virial terms already live there as
    // [9][64]; the force terms join as rows 9..11, then 48 lanes each add a quarter of a row and two
    // shuffles finish it (12 x 6 shuffle steps on 64-bit values were 144 ds_bpermute per wave)
#pragma unroll
    for (int k = 0; k < 3; k++) lv[(9 + k) * 64 + lane] = fsum[k];
    wave_sync();
    {
        const int k = lane >> 2, part = lane & 3;
        double sacc = 0.0;
        if (lane < 48) {
            const double *src = lv + k * 64 + part * 16;
#pragma unroll
            for (int i = 0; i < 16; i++) sacc += src[i];
        }
        sacc += __shfl_xor(sacc, 1, 64);
        sacc += __shfl_xor(sacc, 2, 64);
        if (lane < 48 && part == 0) {
            if (k < 9) vred[wave][k] = sacc;
            else if (active) a.Fself[3 * (size_t)gi + (k - 9)] = sacc;
        }
    }
    __syncthreads();
    if (threadIdx.x < 9)
        a.vir_part[(size_t)threadIdx.x * gridDim.x + blockIdx.x] =
            vred[0][threadIdx.x] + vred[1][threadIdx.x] + vred[2][threadIdx.x] + vred[3][threadIdx.x];
}

// =========================================================================== unpack (tests)
__global__ void unpack_kernel(int n, int S, int L1, int N1, int Dc, int Dpad, const PackEntry *pack,
                              const double *Pp, double *Pd)
{
    const int i = blockIdx.x;
    const int D = N1 * N1 * L1;
    for (int e = threadIdx.x; e < Dc; e += blockDim.x) {
        const PackEntry pe = pack[e];
        const int su = pe.u / N1, nu = pe.u % N1, sv = pe.v / N1, nv = pe.v % N1;
        const double v = Pp[(size_t)i * Dpad + e] / (pe.u == pe.v ? 1.0 : SQRT2);
        // reference layout: block [sb][sa] holds sum_m c[sa][n1] c[sb][n2] at [n1][n2][l]
        double *base = Pd + (size_t)i * S * S * D;
        base[((size_t)sv * S + su) * D + (nu * N1 + nv) * L1 + pe.l] = v;
        base[((size_t)su * S + sv) * D + (nv * N1 + nu) * L1 + pe.l] = v;
    }
}

void launch_unpack_descriptors(int n, int S, int lmax, int nmax, int Dc, int Dpad, const PackEntry *pack,
                               const double *Pp, double *Pdense, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(unpack_kernel, dim3(n), dim3(128), 0, st, n, S, lmax + 1, nmax + 1, Dc, Dpad, pack, Pp, Pdense);
}

// =========================================================================== dispatch
template <int LMAX, int NMAX, int ST>
static size_t fwd_lds_bytes()
{
    using WL = WaveLds<LMAX, NMAX>;
    return sizeof(double) * 4 * (size_t)(WL::TILE_D + ST * WL::NSLOT + 32);
}

template <int LMAX, int NMAX, int ST, bool ENV>
static int run_fwd(const DescArgs &a, hipStream_t st)
{
    if (a.N <= 0) return 0;
    const size_t lds = fwd_lds_bytes<LMAX, NMAX, ST>();
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)desc_fwd_kernel<LMAX, NMAX, ST, ENV>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((desc_fwd_kernel<LMAX, NMAX, ST, ENV>), dim3((a.N + 3) / 4), dim3(256), lds, st, a);
    return 0;
}

template <int LMAX, int NMAX, int ST>
static int run_bwd(const DescArgs &a, hipStream_t st)
{
    if (a.N <= 0) return 0;
    using WL = WaveLds<LMAX, NMAX>;
    const size_t lds1 = sizeof(double) * 4 * (size_t)(ST * WL::NSLOT + (ST <= 4 ? (ST * WL::N1) * (ST * WL::N1) * (LMAX + 1) : a.Dpad));
    static size_t attr_set = 0;
    if (attr_set < lds1) {
        (void)hipFuncSetAttribute((const void *)desc_dc_kernel<LMAX, NMAX, ST>,
                                  hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds1);
        attr_set = lds1;
    }
    if (a.phase != 2) hipLaunchKernelGGL((desc_dc_kernel<LMAX, NMAX, ST>), dim3((a.N + 3) / 4), dim3(256), lds1, st, a);
    if (a.phase == 1) return 0;
    const size_t lds2 = sizeof(double) * 4 * (size_t)(ST * WL::NSLOT + 64 * (WL::LL + 2) + 12 * 64);
    if (a.stride == 1 && a.first == 0 && a.N == a.Nall) {
        hipLaunchKernelGGL((desc_pair_kernel<LMAX, NMAX, ST, 3>), dim3((a.N + 3) / 4), dim3(256), lds2, st, a);
    } else
        hipLaunchKernelGGL((desc_pair_kernel<LMAX, NMAX, ST, 0>), dim3((a.N + 3) / 4), dim3(256), lds2, st, a);
    return 0;
}

static int st_of(int S) { return S <= 1 ? 1 : S <= 2 ? 2 : S <= 3 ? 3 : S <= 4 ? 4 : 8; }

#define DISPATCH_LNS(FN, ...)                                                              \
    do {                                                                                   \
        const int stv = st_of(p.S);                                                        \
        if (p.lmax == 3 && p.nmax == 3) {                                                  \
            if (stv == 1) return FN(3, 3, 1, __VA_ARGS__);                                  \
            if (stv == 2) return FN(3, 3, 2, __VA_ARGS__);                                  \
            if (stv == 3) return FN(3, 3, 3, __VA_ARGS__);                                  \
            if (stv == 4) return FN(3, 3, 4, __VA_ARGS__);                                  \
            return FN(3, 3, 8, __VA_ARGS__);                                                \
        }                                                                                  \
        if (p.lmax == 2 && p.nmax == 2) {                                                  \
            if (stv <= 2) return FN(2, 2, 2, __VA_ARGS__);                                  \
            if (stv <= 4) return FN(2, 2, 4, __VA_ARGS__);                                  \
            return FN(2, 2, 8, __VA_ARGS__);                                                \
        }                                                                                  \
        if (p.lmax == 4 && p.nmax == 4) {                                                  \
            if (stv <= 2) return FN(4, 4, 2, __VA_ARGS__);                                  \
            if (stv <= 4) return FN(4, 4, 4, __VA_ARGS__);                                  \
        }                                                                                  \
        return -6;                                                                         \
    } while (0)

static DescArgs make_args(const DescParams &p)
{
    DescArgs a = {};
    a.N = p.N; a.Nall = p.Nall; a.first = p.first; a.stride = p.stride > 0 ? p.stride : 1; a.maxnn = p.maxnn; a.S = p.S; a.Dc = p.Dc; a.Dpad = p.Dpad; a.CS = p.CS;
    a.rc = p.rc;
    for (int k = 0; k < SGPR_MAX_S; k++) a.radii_v[k] = p.radii_v[k];
    return a;
}

#define FWD_NL(L, N, S, a, st) run_fwd<L, N, S, false>(a, st)
#define FWD_ENV(L, N, S, a, st) run_fwd<L, N, S, true>(a, st)
#define BWD(L, N, S, a, st) run_bwd<L, N, S>(a, st)

int launch_descriptor_forward(const DescParams &p, const double *pos, const double *cell, const int *slot,
                              const double *radii, const int *nn, const int *nbr_j, const int *nbr_shift,
                              const PackEntry *pack, double *Pn, double *norm, double *C, int *shear,
                              hipStream_t st)
{
    DescArgs a = make_args(p);
    a.pos = pos; a.cell = cell; a.slot = slot; a.radii = radii; a.nn = nn; a.nbr_j = nbr_j;
    a.nbr_shift = nbr_shift; a.pack = pack; a.Pn = Pn; a.norm = norm; a.C = C; a.shear = shear;
    DISPATCH_LNS(FWD_NL, a, st);
}

int launch_descriptor_forward_env(const DescParams &p, const int64_t *env_ptr, const int *env_slot,
                                  const double *env_r, const double *radii, const PackEntry *pack, double *Pn,
                                  double *norm, hipStream_t st)
{
    DescArgs a = make_args(p);
    a.env_ptr = env_ptr; a.env_slot = env_slot; a.env_r = env_r; a.radii = radii; a.pack = pack;
    a.Pn = Pn; a.norm = norm; a.C = nullptr; a.shear = nullptr;
    DISPATCH_LNS(FWD_ENV, a, st);
}

int launch_descriptor_backward(const DescParams &p, const double *pos, const double *cell, const int *slot,
                               const double *radii, const int *nn, const int *nbr_j, const int *nbr_shift,
                               const PackEntry *pack, const double *Pn, const double *norm, const double *C,
                               const int *shear, const double *W, double *dC, double *F, double *virial,
                               int phase, hipStream_t st)
{
    DescArgs a = make_args(p);
    a.pos = pos; a.cell = cell; a.slot = slot; a.radii = radii; a.nn = nn; a.nbr_j = nbr_j;
    a.nbr_shift = nbr_shift; a.pack = pack; a.Pn = (double *)Pn; a.norm = (double *)norm; a.C = (double *)C;
    a.shear = (int *)shear; a.W = W; a.dC = dC; a.phase = phase;
    // F points at [Fnbr | Fself], virial at the per-wave partial array (see api.hip)
    a.Fnbr = F;
    a.Fself = F + 3 * (size_t)p.Nall;
    a.vir_part = virial;
    DISPATCH_LNS(BWD, a, st);
}
