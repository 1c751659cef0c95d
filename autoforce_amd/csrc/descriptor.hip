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
#include "gemm_tile.inc"  // the covloss tiles of a step ride in the reverse kernel's launch

__constant__ HarmCoef c_hc;

void upload_harm_coef(const HarmCoef &hc) { (void)hipMemcpyToSymbol(HIP_SYMBOL(c_hc), &hc, sizeof(HarmCoef)); }

#define SQRT2 1.4142135623730951
typedef double v4d __attribute__((ext_vector_type(4)));

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

// opaque copy of a wave-uniform pointer: loads through it cannot be hoisted above this point
template <typename T>
__device__ __forceinline__ const T *launder(const T *p)
{
    asm volatile("" : "+s"(p));
    return p;
}

__device__ __forceinline__ double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// Which quad of atoms workgroup b works on.  The dispatcher sends workgroup b to XCD b % 8, and the tile tables of the GEMMs
// put row tile t on XCD t % 8 (api.hip::build_tiles): with R quads per row tile, the 8R workgroups b = 8R g + 8 s + x take
// the quads 8R g + R x + s — the rows of a GEMM tile are produced (forward pass -> K_nm) and consumed (W -> reverse pass)
// on the XCD whose L2 holds them.  A last partial group keeps the identity.
__device__ __forceinline__ int xcd_quad(int b, int n, int R)
{
    if (R <= 0) return b;
    const int grp = 8 * R, g0 = b / grp * grp;
    if (g0 + grp > n) return b;
    const int r = b - g0;
    return g0 + (r & 7) * R + (r >> 3);
}

// ---------------------------------------------------------------- solid harmonics
template <int LMAX>
struct Harm {
    static constexpr int L1 = LMAX + 1, LL = L1 * L1;
    double A[L1], B[L1], q[L1][L1];
    double x, y, z, rho;
    // recurrence coefficients: constant memory, read through a pointer so that a kernel short of
    // SGPRs can re-derive it (launder()) and have the scalar loads issued next to their use instead
    // of hoisted across its whole tile loop
    const HarmCoef *hc = &c_hc;

    // recurrence state (A, B, q) at (x, y, z); eval()/dot()/backward() build on it
    __device__ __forceinline__ void prepare(double x_, double y_, double z_)
    {
        x = x_; y = y_; z = z_;
        rho = x * x + y * y + z * z;
        A[0] = 1.0; B[0] = 0.0;
#pragma unroll
        for (int m = 1; m <= LMAX; m++) {
            A[m] = x * A[m - 1] - y * B[m - 1];
            B[m] = y * A[m - 1] + x * B[m - 1];
        }
        q[0][0] = hc->y00;
#pragma unroll
        for (int l = 1; l <= LMAX; l++) {
#pragma unroll
            for (int m = 0; m <= l - 2; m++)
                q[l][m] = hc->al[l][m] * (z * q[l - 1][m] + rho * hc->bl[l][m] * q[l - 2][m]);
            q[l][l - 1] = hc->cl[l] * z * q[l - 1][l - 1];
            q[l][l] = hc->dl[l] * q[l - 1][l - 1];
        }
    }

    __device__ __forceinline__ void eval(double x_, double y_, double z_, double *Y)
    {
        prepare(x_, y_, z_);
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

    // Y and its gradient with respect to (x, y, z), forward mode; prepare() must have run.  The recurrence state is
    // q[l][m](z, rho) and A + iB = (x + iy)^m: dq/dz and dq/drho follow the same recurrences, dA/dx = m A[m-1],
    // dA/dy = -m B[m-1], dB/dx = m B[m-1], dB/dy = m A[m-1], and rho = x^2 + y^2 + z^2 brings 2 (x, y, z) dq/drho.
    // (The reverse-mode twin is backward(); the training-rows kernel needs the Jacobian itself, rows16.inc.)
    __device__ __forceinline__ void eval_grad(double *Y, double *Yx, double *Yy, double *Yz) const
    {
        double qz[L1][L1], qr[L1][L1];
#pragma unroll
        for (int l = 0; l <= LMAX; l++)
#pragma unroll
            for (int m = 0; m <= LMAX; m++) { qz[l][m] = 0.0; qr[l][m] = 0.0; }
#pragma unroll
        for (int l = 1; l <= LMAX; l++) {
#pragma unroll
            for (int m = 0; m <= l - 2; m++) {
                qz[l][m] = hc->al[l][m] * (q[l - 1][m] + z * qz[l - 1][m] + rho * hc->bl[l][m] * qz[l - 2][m]);
                qr[l][m] = hc->al[l][m] * (z * qr[l - 1][m] + hc->bl[l][m] * (q[l - 2][m] + rho * qr[l - 2][m]));
            }
            qz[l][l - 1] = hc->cl[l] * q[l - 1][l - 1];  // q[l-1][l-1] is a constant
        }
#pragma unroll
        for (int l = 0; l <= LMAX; l++) {
            {
                const double dz = qz[l][0] + 2.0 * z * qr[l][0];
                Y[l * l] = q[l][0];
                Yx[l * l] = 2.0 * x * qr[l][0];
                Yy[l * l] = 2.0 * y * qr[l][0];
                Yz[l * l] = dz;
            }
#pragma unroll
            for (int m = 1; m <= l; m++) {
                const double qq = SQRT2 * q[l][m], qrr = SQRT2 * qr[l][m], dz = SQRT2 * (qz[l][m] + 2.0 * z * qr[l][m]);
                const int kc = l * l + 2 * m - 1, ks = l * l + 2 * m;
                Y[kc] = qq * A[m];
                Y[ks] = qq * B[m];
                Yx[kc] = qq * m * A[m - 1] + 2.0 * x * qrr * A[m];
                Yy[kc] = -qq * m * B[m - 1] + 2.0 * y * qrr * A[m];
                Yz[kc] = dz * A[m];
                Yx[ks] = qq * m * B[m - 1] + 2.0 * x * qrr * B[m];
                Yy[ks] = qq * m * A[m - 1] + 2.0 * y * qrr * B[m];
                Yz[ks] = dz * B[m];
            }
        }
    }

    // sum_k Y_k w[k] without materialising Y; prepare() must have run.
    __device__ __forceinline__ double dot(const double *w) const
    {
        double s = 0.0;
#pragma unroll
        for (int l = 0; l <= LMAX; l++) {
            s += q[l][0] * w[l * l];
#pragma unroll
            for (int m = 1; m <= l; m++)
                s += SQRT2 * q[l][m] * (A[m] * w[l * l + 2 * m - 1] + B[m] * w[l * l + 2 * m]);
        }
        return s;
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
            gq[l - 1][l - 1] += hc->dl[l] * gq[l][l];
            gzz += hc->cl[l] * q[l - 1][l - 1] * gq[l][l - 1];
            gq[l - 1][l - 1] += hc->cl[l] * z * gq[l][l - 1];
#pragma unroll
            for (int m = 0; m <= l - 2; m++) {
                const double g = hc->al[l][m] * gq[l][m];
                gzz += q[l - 1][m] * g;
                grho += hc->bl[l][m] * q[l - 2][m] * g;
                gq[l - 1][m] += z * g;
                gq[l - 2][m] += rho * hc->bl[l][m] * g;
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
    double rc, irc;         // cutoff and its reciprocal
    const double *pos;      // [Nall][3] (sorted order)
    const double *cell;     // [9]
    const int *slot;        // [Nall]
    const double *radii;    // [S] (device; ENV path)
    double radii_v[SGPR_MAX_S];  // the same by value: unit lookup is a select chain, no load
    double radii_iv[SGPR_MAX_S]; // reciprocals
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
    long long *stamps;      // diagnostic build (-DSGPR_PHASE_STAMPS) only: [N][8] s_memtime per phase
    double *prec;           // [Nall][maxnn][4] pair records (r_x, r_y, r_z, exp(-d^2/2)): forward -> reverse pass
    double *G;              // reverse pass, gather form: [Nall][maxnn][4]: G[j][rev] = gradient of the pair (i -> j)
    const int *aux;         // [Nall][maxnn] bin-sweep id of each CANDIDATE    } reverse index of the candidate build:
    const unsigned short *T; // [Nall][t_stride]                               } T[i][aux[i][c]] = position of i among
    int t_stride;           //                                                   the candidates of its candidate c
    const int *cidx;        // [Nall][maxnn] candidate position of each list entry of this step
    const unsigned long long *hm;  // [Nall][hmw] this step's hit mask over the candidates
    int hmw;
    int rsz;                // reverse pass: doubles of the per-wave scratch region
    int xq;                 // quads of atoms per GEMM row tile (8 or 16; 0: workgroup b works on quad b) — see xcd_quad
    int *shear;             // [N]
    // training rows (sgpr_kernel_rows): blockIdx.y = column of the batch; the reverse-pass seed of column q is
    // W_i = Aw[i][q] * Pm[q][:], formed on the fly (no W array), and every output is strided by the batch index
    const double *rows_aw;  // [N][rows_ld] d k(i,q) / d(dot), or null (predict path: W below)
    const double *rows_pm;  // [m][Dpad]
    const int *rows_cols;   // [gridDim.y] species-sorted inducing index of each column of the batch
    const int *rows_colslot; // [m] species slot of the inducing LCEs
    int rows_ld, batch;
    size_t g_stride, f_stride, v_stride;  // doubles between the batch entries of G, F (= [Fnbr | Fself]), vir_part
    const double *W;        // backward: [N][Dpad]
    double *Fnbr;           // backward: [Nall][3] 64-bit fixed-point sums (integer atomics; SGPR_FIX_SCALE)
    int *stat;              // sticky status words (neighbor.hip): [3] = 3 when a force contribution leaves the fixed-point range
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

template <int ST>
__device__ __forceinline__ double inv_unit_of(const DescArgs &a, int s)
{
    double u = a.radii_iv[0];
#pragma unroll
    for (int q = 1; q < ST; q++) u = (s == q) ? a.radii_iv[q] : u;
    return u;
}

// Species slots in use.  The default instantiations (lmax = nmax = 3, up to four slots) are dispatched one per species
// count (DISPATCH_LNS: st_of(S) = S), so there the count is a COMPILE-TIME constant: the loops over species and channels
// unroll and the `v < channels` guards fold away — they were scalar branches around single LDS reads (a third of the
// instruction stream of the reverse kernel was scalar: profiles/r03_pmc_sq_summary.txt).
template <int LMAX, int NMAX, int ST>
__device__ __forceinline__ int species_in_use(const DescArgs &a)
{
    if constexpr (LMAX == 3 && NMAX == 3 && ST <= 4) return ST;
    else return a.S;
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

// radial weights f_n = g d^(2n) (and dg/dd for the reverse pass); ex = exp(-d^2/2)
template <int NMAX>
__device__ __forceinline__ void radial_ex(double d, double u, double rc, double irc, double ex, double *f, double &g,
                                          double &dg)
{
    // reciprocals instead of fp64 divisions (each is a ~12-instruction sequence on the VALU)
    const double ud = u * d;
    const double step = ud < rc ? 1.0 : 0.0;
    const double qq = 1.0 - ud * irc;
    const double cut = step * qq * qq;
    const double dcut = step * (-2.0 * qq * irc) * u;
    g = cut * ex;
    dg = dcut * ex - d * g;
    const double rho = d * d;
    double pw = g;
#pragma unroll
    for (int n = 0; n <= NMAX; n++) { f[n] = pw; pw *= rho; }
}

template <int NMAX>
__device__ __forceinline__ void radial(double d, double u, double rc, double *f, double &g, double &dg, double &ex)
{
    ex = exp(-0.5 * d * d);
    radial_ex<NMAX>(d, u, rc, 1.0 / rc, ex, f, g, dg);
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
            double f[N1], g, dg, ex;
            radial<NMAX>(d, u, a.rc, f, g, dg, ex);
            if constexpr (!ENV) {
                // pair record for the reverse pass: displacement and exp(-d^2/2), read back coalesced
                // (no second gather of the neighbour's position, no second exp)
                if (a.prec) {
                    double2 *dst = (double2 *)(a.prec + ((size_t)gi * a.maxnn + t) * 4);
                    dst[0] = make_double2(r[0], r[1]);
                    dst[1] = make_double2(r[2], ex);
                }
            }
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

// =========================================================================== list build + forward
// One wave64 per atom i (reference: ase NeighborList as driven by descriptor/atoms.py:348-363,:402, then
// descriptor/sesoap.py:161-260).
//  candidates (only when the rebuild flag of this step is set — an atom moved more than skin/2 since the
//          last build, the cell changed, or the frame is new; neighbor.hip decides): sweep of the
//          (2R+1)^3 neighbouring bins for |r| < rc + skin, FOUR bins x 16 slots per step (lane = bin of the
//          group, slot), bins beyond reach dropped; hits compacted into LDS and SORTED by (j, image): the
//          placement order inside a bin comes from atomics and is not reproducible, the sorted list is.
//          Written once per build: the candidate list and the reverse-index ingredients — atom i found j as
//          candidate (q, k) = (bin offset index, slot in that bin); j finds i under the mirrored offset
//          nbox-1-q at i's own slot, so i writes its candidate position c into T[j][(nbox-1-q)*cap + k_i]
//          and keeps aux[i][c] = q*cap + k_j: T[i][aux[i][c]] is the position of i among j's candidates.
//  list    every step: one pass over the ~51 candidates (lane = candidate): displacement from the current
//          positions, |r| < rc, ballot compaction — a subsequence of a sorted sequence, so the list has
//          exactly the pairs AND the order of a from-scratch build.  The hit mask hm[i] lets the reverse
//          pass turn a candidate position into a list position by a popcount.
//  forward lane = neighbour (tiles of 48): radial weights, solid harmonics, staged in LDS;
//          c[lm][(s,n)] += sum_t Y[t][lm] f[t][n] [s_t = s] on v_mfma_f64_16x16x4 (K = four neighbours);
//          power spectrum with one lane per (u,v) pair; norm in-wave.  The displacement and exp(-d^2/2)
//          of every pair are left in `prec` for the reverse pass.
#define NL_SORT_MAX 256  // candidate lists up to this length are sorted (longer ones keep sweep order beyond it)
#ifndef NL_STEPS
#define NL_STEPS 4
#endif

struct NlArgs {
    const NlGrid *grid;
    const int *bin_of, *kslot, *bin_count;
    int cap;
    const BinRec *b_rec;
    const BinAux *b_aux;
    int *nn, *nn_local, *nn_raw, *nbr_j, *nbr_shift, *aux;
    unsigned short *T;
    int t_stride;
    int *stat;
    // Verlet candidates
    const int *flag;       // this step's rebuild flag (neighbor.hip)
    double rc_list;        // rc + skin
    int *ncand, *cand_j, *cand_code, *cidx;   // [N], [N][maxnn] x2, per-step list entry -> candidate position
    unsigned long long *hm;                   // [N][hmw] hit mask over the candidates
    int hmw;
};

#ifdef SGPR_PHASE_STAMPS
#define PHASE_STAMP(K) if (a.stamps && lane == 0) a.stamps[(size_t)ia * 8 + (K)] = (long long)__builtin_amdgcn_s_memtime()
#else
#define PHASE_STAMP(K)
#endif

template <int LMAX, int NMAX, int ST>
struct FwdLds {
    using WL = WaveLds<LMAX, NMAX>;
    static constexpr int CH = WL::CH;
    // list-build view of the shared region: keys | candidate ids | bins worth visiting | hits of the first tile
    static constexpr int BINL = NL_SORT_MAX + NL_SORT_MAX / 2, HIT0 = BINL + 2 * 64;
    static constexpr int NLV = HIT0 + 4 * CH;
    static constexpr int FWV = CH * WL::N1 + CH * WL::LLP + CH / 2;              // radial rows | harmonic rows | species
    static constexpr int P4V = ST * WL::NSLOT;                                   // c for the power spectrum
    static constexpr int RA = (NLV > FWV ? (NLV > P4V ? NLV : P4V) : (FWV > P4V ? FWV : P4V));
    static constexpr int PW = 4 * CH + RA;  // + the hits of the second tile
};

template <int LMAX, int NMAX, int ST>
__global__ __launch_bounds__(256, 4) void nl_fwd_kernel(DescArgs a, NlArgs n)
{
    using WL = WaveLds<LMAX, NMAX>;
    using FL = FwdLds<LMAX, NMAX, ST>;
    constexpr int N1 = WL::N1, LL = WL::LL, LLP = WL::LLP, NSLOT = WL::NSLOT, CH = WL::CH;
    constexpr int UC = ST * N1, CBS = (UC + 15) / 16, RBL = (LL + 15) / 16;
    extern __shared__ double smem[];
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int ia = xcd_quad((int)blockIdx.x, (int)gridDim.x, a.xq) * 4 + wave;
    if (ia >= a.N) return;  // (no workgroup barrier in this kernel)
    const int i = a.first + ia * a.stride;
    double *wbase = smem + (size_t)wave * FL::PW;
    double *hit1 = wbase;                                          // [CH][4] hits of the second tile (r, species)
    double *RA = wbase + 4 * CH;
    unsigned long long *keys = (unsigned long long *)RA;           // [NL_SORT_MAX]   } list build view
    int *hq = (int *)(RA + NL_SORT_MAX);                           // [NL_SORT_MAX]   }
    int4 *binl = (int4 *)(RA + FL::BINL);                          // [64]            }
    double *hit0 = RA + FL::HIT0;                                  // [CH][4]         }
    double *fl = RA;                                               // [CH][N1]        } forward view
    double *Yl = fl + CH * N1;                                     // [CH][LLP]       }
    int *sl = (int *)(Yl + CH * LLP);                              // [CH]            }
    double *cl = RA;                                               // [ST][NSLOT]       power-spectrum view

    PHASE_STAMP(0);
    const int cap = n.cap, maxnn = a.maxnn;
    double h[9];
#pragma unroll
    for (int k = 0; k < 9; k++) h[k] = a.cell[k];
    const double xi = uniform(a.pos[3 * (size_t)i]), yi = uniform(a.pos[3 * (size_t)i + 1]), zi = uniform(a.pos[3 * (size_t)i + 2]);
    const int ki = __builtin_amdgcn_readfirstlane(n.kslot[i]);
    const bool ghost = ki < 0;  // species outside the model's table (option "ignore_unknown_species"): no environment
    const unsigned long long lt = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    const bool rebuild = __builtin_amdgcn_readfirstlane(*n.flag) != 0;
    int ncand = 0;
    int j_first = 0, cd_first = 0;  // reuse steps: candidate 0..63 of this atom (set below)
    if (rebuild) {
        // -------------------------------------------------------------- candidates: sweep + sort
        const NlGrid g = *n.grid;
        const int bi = __builtin_amdgcn_readfirstlane(n.bin_of[i]);
        int wi0 = 0, wi1 = 0, wi2 = 0;
        if (ki >= 0 && ki < cap) {  // (ki >= cap: the bin overflowed, the host grows it and reruns the step)
            const BinAux own = n.b_aux[(size_t)bi * cap + ki];
            wi0 = own.w0; wi1 = own.w1; wi2 = own.w2;
        }
        // index arithmetic without integer division (a ~30-instruction sequence each on this ISA): small
        // non-negative operands, so floor((q + 1/2) * (1/w)) in fp32 is exact
        auto fdiv = [](int q, float inv) { return (int)(((float)q + 0.5f) * inv); };
        const float i_n2 = 1.0f / (float)g.nb[2], i_n1 = 1.0f / (float)g.nb[1];
        const int bq = fdiv(bi, i_n2);
        const int b2 = bi - bq * g.nb[2];
        const int b0 = fdiv(bq, i_n1);
        const int b1 = bq - b0 * g.nb[1];
        const int w0 = 2 * g.rng[0] + 1, w1 = 2 * g.rng[1] + 1, w2 = 2 * g.rng[2] + 1;
        const float i_w2 = 1.0f / (float)w2, i_w1 = 1.0f / (float)w1;
        const double r_n0 = 1.0 / g.nb[0], r_n1 = 1.0 / g.nb[1], r_n2 = 1.0 / g.nb[2];
        const int nbox = w0 * w1 * w2;
        // reverse-index table: row stride nbox*cap entries; a smaller allocation is reported (sticky) and
        // the host grows it and reruns, like the other capacities
        const bool t_ok = n.T != nullptr && (long long)nbox * cap <= (long long)n.t_stride && cap <= 4096 && maxnn <= 65535 &&
                          ki < cap && !ghost;
        if (n.T != nullptr && !t_ok && ki < cap && !ghost && lane == 0)
            atomicMax(&n.stat[2], cap <= 4096 && maxnn <= 65535 ? nbox * cap : 0x7fffffff);
        const int grp = lane >> 4, s16 = lane & 15;
        const double rl2 = n.rc_list * n.rc_list;
        int base = 0;
        // one candidate per lane: distance test, compaction, key
        auto consume = [&](bool valid, const BinRec &rec, const BinAux &ax, int code, int q, int k) {
            bool hit = false;
            int f0 = 0, f1 = 0, f2 = 0;
            if (valid) {
                f0 = (int)(int8_t)(code & 0xff) - ax.w0 + wi0;
                f1 = (int)(int8_t)((code >> 8) & 0xff) - ax.w1 + wi1;
                f2 = (int)(int8_t)((code >> 16) & 0xff) - ax.w2 + wi2;
                const double dx = rec.x - xi + (f0 * h[0] + f1 * h[3] + f2 * h[6]);
                const double dy = rec.y - yi + (f0 * h[1] + f1 * h[4] + f2 * h[7]);
                const double dz = rec.z - zi + (f0 * h[2] + f1 * h[5] + f2 * h[8]);
                hit = dx * dx + dy * dy + dz * dz < rl2 && !(rec.idx == i && f0 == 0 && f1 == 0 && f2 == 0);
            }
            const unsigned long long m = __ballot(hit);
            if (hit) {
                const int slot = base + __popcll(m & lt);
                const int j = rec.idx, sj = ax.slot;
                // key: neighbour index (24 bits), the image triple biased to sort as unsigned (24), species
                // slot (4), sweep ordinal (12: finds the candidate id again after the sort)
                const unsigned img = (unsigned)((f0 + 128) & 0xff) << 16 | (unsigned)((f1 + 128) & 0xff) << 8 |
                                     (unsigned)((f2 + 128) & 0xff);
                const unsigned long long key = ((unsigned long long)(unsigned)j << 40) | ((unsigned long long)img << 16) |
                                               ((unsigned long long)(unsigned)sj << 12) | (unsigned)(slot & 0xfff);
                if (max(max(abs(f0), abs(f1)), abs(f2)) > 127) atomicMax(&n.stat[3], 1);  // image shift beyond the packed code
                if (slot < NL_SORT_MAX) {
                    keys[slot] = key;
                    hq[slot] = (q << 12) | k;
                } else if (slot < maxnn) {  // very long lists: keep sweep order beyond the sortable part
                    const size_t e = (size_t)i * maxnn + slot;
                    n.cand_j[e] = j;
                    n.cand_code[e] = (f0 & 0xff) | ((f1 & 0xff) << 8) | ((f2 & 0xff) << 16) | (sj << 24);
                    n.aux[e] = 0;
                    if (t_ok) {
                        n.aux[e] = q * cap + k;
                        n.T[(size_t)j * n.t_stride + (size_t)(nbox - 1 - q) * cap + ki] = (unsigned short)slot;
                    }
                }
            }
            base += __popcll(m);
        };
        constexpr int STEPS = NL_STEPS;  // steps (of four bins) whose loads are requested together
        // position of the atom inside its own bin, in bin units: bounds the distance to the neighbouring bins
        double tb[3];
#pragma unroll
        for (int k = 0; k < 3; k++) {
            const double fk = xi * g.inv[k] + yi * g.inv[3 + k] + zi * g.inv[6 + k];
            const double uk = (fk - floor(fk)) * g.nb[k] - (k == 0 ? b0 : k == 1 ? b1 : b2);
            tb[k] = g.rng[k] > 0 ? fmin(fmax(uk, 0.0), 1.0) : 0.0;
        }
        const double rl2_skip = rl2 * (1.0 + 1e-9);
        for (int q0 = 0; q0 < (ghost ? 0 : nbox); q0 += 64) {
            // lane = neighbouring bin: index, image code and population of up to 64 bins with one load; bins that
            // are empty or whose nearest point lies beyond reach are dropped (about 19 of 27 remain)
            int nbin_l = 0, code_l = 0, cnt_l = 0;
            bool keep = false;
            {
                const int q = q0 + lane;
                if (q < nbox) {
                    const int qa = fdiv(q, i_w2), qb = fdiv(qa, i_w1);
                    const int o2 = q - qa * w2 - g.rng[2], o1 = qa - qb * w1 - g.rng[1], o0 = qb - g.rng[0];
                    const int t0 = b0 + o0, t1 = b1 + o1, t2 = b2 + o2;
                    const int c0 = (int)floor((double)t0 * r_n0 + 1e-9), c1 = (int)floor((double)t1 * r_n1 + 1e-9),
                              c2 = (int)floor((double)t2 * r_n2 + 1e-9);
                    nbin_l = ((t0 - c0 * g.nb[0]) * g.nb[1] + (t1 - c1 * g.nb[1])) * g.nb[2] + (t2 - c2 * g.nb[2]);
                    code_l = (c0 & 0xff) | ((c1 & 0xff) << 8) | ((c2 & 0xff) << 16);
                    cnt_l = min(n.bin_count[(size_t)nbin_l * SGPR_BIN_STRIDE], cap);
                    // gap between the atom and the bin along each plane normal (bin units -> length)
                    const double g0 = (o0 > 0 ? o0 - tb[0] : o0 < 0 ? tb[0] - o0 - 1.0 : 0.0) * g.w[0];
                    const double g1 = (o1 > 0 ? o1 - tb[1] : o1 < 0 ? tb[1] - o1 - 1.0 : 0.0) * g.w[1];
                    const double g2 = (o2 > 0 ? o2 - tb[2] : o2 < 0 ? tb[2] - o2 - 1.0 : 0.0) * g.w[2];
                    const double gm = fmax(g0, fmax(g1, g2));
                    const double lb2 = g.ortho ? g0 * g0 + g1 * g1 + g2 * g2 : gm * gm;
                    keep = cnt_l > 0 && lb2 < rl2_skip;
                }
            }
            const unsigned long long km = __ballot(keep);
            const int nlist = __popcll(km);
            wave_sync();  // the previous chunk's readers of binl are done
            if (keep) binl[__popcll(km & lt)] = make_int4(nbin_l, code_l, cnt_l, q0 + lane);
            wave_sync();
            const int nsteps = (nlist + 3) >> 2;
            for (int u0 = 0; u0 < nsteps; u0 += STEPS) {
                // step u of the chunk: the 16 lanes of group grp take entry 4*(u0+u) + grp of the bin list
                int nbin[STEPS], code[STEPS], cnt[STEPS], qq[STEPS], cmax = 0;
#pragma unroll
                for (int u = 0; u < STEPS; u++) {
                    const int e = 4 * (u0 + u) + grp;
                    const int4 bb = binl[e < nlist ? e : 0];
                    nbin[u] = bb.x; code[u] = bb.y; qq[u] = bb.w;
                    cnt[u] = e < nlist ? bb.z : 0;
                    cmax = max(cmax, cnt[u]);
                }
                // 16 slots of every bin per pass: one pass unless a bin holds more than 16 atoms
                for (int s0 = 0; __any(s0 < cmax); s0 += 16) {
                    BinRec rec[STEPS];
                    BinAux ax[STEPS];
#pragma unroll
                    for (int u = 0; u < STEPS; u++) {
                        const size_t e = (size_t)nbin[u] * cap + (s0 + s16 < cnt[u] ? s0 + s16 : 0);
                        rec[u] = n.b_rec[e];
                        ax[u] = n.b_aux[e];
                    }
#pragma unroll
                    for (int u = 0; u < STEPS; u++)
                        if (__any(s0 + s16 < cnt[u])) consume(s0 + s16 < cnt[u], rec[u], ax[u], code[u], qq[u], s0 + s16);
                }
            }
        }
        PHASE_STAMP(1);
        // bitonic sort of the first min(base, NL_SORT_MAX) keys.  Up to 64 keys sort in registers, one key per
        // lane, partners by cross-lane shuffle: a third of the instructions of the LDS network and no barriers.
        const int ns = min(base, NL_SORT_MAX);
        wave_sync();
        if (ns <= 64) {
            unsigned long long key = lane < ns ? keys[lane] : ~0ull;
#pragma unroll
            for (int k2s = 2; k2s <= 64; k2s <<= 1)
#pragma unroll
                for (int j2 = k2s >> 1; j2 > 0; j2 >>= 1) {
                    const unsigned lo = __shfl_xor((unsigned)key, j2, 64), hi = __shfl_xor((unsigned)(key >> 32), j2, 64);
                    const unsigned long long other = ((unsigned long long)hi << 32) | lo;
                    const bool lower = (lane & j2) == 0, up = (lane & k2s) == 0;
                    const bool take_min = lower == up;  // the lower lane of a pair keeps the smaller key in an ascending block
                    key = take_min ? (other < key ? other : key) : (other > key ? other : key);
                }
            wave_sync();  // all lanes have read their unsorted key
            if (lane < ns) keys[lane] = key;
        } else {
            int np2 = 1;
            while (np2 < ns) np2 <<= 1;
            for (int t = ns + lane; t < np2; t += 64) keys[t] = ~0ull;
            wave_sync();
            for (int k2s = 2; k2s <= np2; k2s <<= 1)
                for (int j2 = k2s >> 1; j2 > 0; j2 >>= 1) {
                    for (int t = lane; t < np2; t += 64) {
                        const int p = t ^ j2;
                        if (p > t) {
                            const unsigned long long a0 = keys[t], a1 = keys[p];
                            const bool up = (t & k2s) == 0;
                            if ((a0 > a1) == up) { keys[t] = a1; keys[p] = a0; }
                        }
                    }
                    wave_sync();
                }
        }
        wave_sync();
        // candidate list out (kept until the next rebuild)
        ncand = base < maxnn ? base : maxnn;
        for (int c = lane; c < ns && c < maxnn; c += 64) {
            const unsigned long long key = keys[c];
            const unsigned img = (unsigned)(key >> 16) & 0xffffffu;
            const int f0 = (int)((img >> 16) & 0xff) - 128, f1 = (int)((img >> 8) & 0xff) - 128, f2 = (int)(img & 0xff) - 128,
                      sj = (int)(key >> 12) & 0xf, j = (int)(key >> 40);
            const size_t e = (size_t)i * maxnn + c;
            n.cand_j[e] = j;
            n.cand_code[e] = (f0 & 0xff) | ((f1 & 0xff) << 8) | ((f2 & 0xff) << 16) | (sj << 24);
            int hs = 0;
            if (t_ok) {
                const int hv = hq[(int)key & 0xfff];
                const int qv = hv >> 12, kk = hv & 0xfff;
                hs = qv * cap + kk;
                n.T[(size_t)j * n.t_stride + (size_t)(nbox - 1 - qv) * cap + ki] = (unsigned short)c;
            }
            n.aux[e] = hs;
        }
        if (lane == 0) {
            n.ncand[i] = ncand;
            n.nn_raw[ia] = base;  // unclamped: finalize reduces the max for the overflow check
        }
        if (ncand > NL_SORT_MAX) __threadfence();  // rare: the unsorted tail written during the sweep is read back below
        PHASE_STAMP(2);
    } else {
        // reuse step: the first 64 candidates are requested together with their count (the row has maxnn slots), not
        // behind it — one round trip less on the chain count -> candidates -> positions
        const int nc_v = n.ncand[i];
        if (lane < maxnn) {
            const size_t e = (size_t)i * maxnn + lane;
            j_first = n.cand_j[e];
            cd_first = n.cand_code[e];
        }
        ncand = __builtin_amdgcn_readfirstlane(nc_v);
        if (lane == 0) n.nn_raw[ia] = ncand;
    }
    // ------------------------------------------------------------------ this step's list: filter the candidates
    const double rc2_lo = a.rc * a.rc * (1.0 - 1e-14), rc2_hi = a.rc * a.rc * (1.0 + 1e-14);
    int nn = 0;
    bool near = false;
    for (int c0 = 0; c0 < ncand; c0 += 64) {
        const int c = c0 + lane;
        bool hit = false;
        int j = 0, cd = 0;
        double r0 = 1.0, r1 = 0.0, r2 = 0.0;
        if (c < ncand) {
            if (rebuild && c < NL_SORT_MAX) {
                const unsigned long long key = keys[c];
                const unsigned img = (unsigned)(key >> 16) & 0xffffffu;
                const int f0 = (int)((img >> 16) & 0xff) - 128, f1 = (int)((img >> 8) & 0xff) - 128, f2 = (int)(img & 0xff) - 128;
                j = (int)(key >> 40);
                cd = (f0 & 0xff) | ((f1 & 0xff) << 8) | ((f2 & 0xff) << 16) | (((int)(key >> 12) & 0xf) << 24);
            } else if (!rebuild && c0 == 0) {
                j = j_first; cd = cd_first;
            } else {
                const size_t e = (size_t)i * maxnn + c;
                j = n.cand_j[e];
                cd = n.cand_code[e];
            }
            const int f0 = (int)(int8_t)(cd & 0xff), f1 = (int)(int8_t)((cd >> 8) & 0xff), f2 = (int)(int8_t)((cd >> 16) & 0xff);
            r0 = a.pos[3 * (size_t)j] - xi + (f0 * h[0] + f1 * h[3] + f2 * h[6]);
            r1 = a.pos[3 * (size_t)j + 1] - yi + (f0 * h[1] + f1 * h[4] + f2 * h[7]);
            r2 = a.pos[3 * (size_t)j + 2] - zi + (f0 * h[2] + f1 * h[5] + f2 * h[8]);
            // |r| < rc on the squared distance; only within a few ulp of the cutoff the square root decides
            // (the pair rule of the reference list is on |r| itself)
            const double d2 = r0 * r0 + r1 * r1 + r2 * r2;
            hit = d2 < rc2_lo || (d2 < rc2_hi && sqrt(d2) < a.rc);
        }
        const unsigned long long m = __ballot(hit);
        if (lane == 0) n.hm[(size_t)i * n.hmw + (c0 >> 6)] = m;
        if (hit) {
            const int t = nn + __popcll(m & lt);
            const int sj = (cd >> 24) & 0xff;
            const size_t e = (size_t)i * maxnn + t;
            n.nbr_j[e] = j;
            n.nbr_shift[e] = cd;
            n.cidx[e] = c;
            double *dst = t < CH ? hit0 + 4 * t : (t < 2 * CH ? hit1 + 4 * (t - CH) : nullptr);
            if (dst) { dst[0] = r0; dst[1] = r1; dst[2] = r2; dst[3] = __hiloint2double(0, sj); }
            // does the neighbour sit inside the z cone? (ylm.py:10-23: then the whole environment shears;
            // the length unit scales all three components alike)
            const double tol = SGPR_TINY_ANGLE * fabs(r2);
            near |= (fabs(r0) < tol) && (fabs(r1) < tol);
        }
        nn += __popcll(m);
    }
    for (int w = (ncand + 63) / 64 + lane; w < n.hmw; w += 64) n.hm[(size_t)i * n.hmw + w] = 0ull;
    const bool shear = __any(near);
    if (lane == 0) {
        n.nn[i] = nn;
        n.nn_local[ia] = nn;
    }
    wave_sync();
    // tile 0 keeps its hits in registers (the region is about to be reused); tile 1 reads hit1; further tiles
    // (> 96 neighbours) read the list back from memory
    double h0r[3] = {1.0, 0.0, 0.0};
    int h0s = 0;
    if (lane < min(nn, CH)) {
        h0r[0] = hit0[4 * lane]; h0r[1] = hit0[4 * lane + 1]; h0r[2] = hit0[4 * lane + 2];
        h0s = __double2loint(hit0[4 * lane + 3]);
    }
    if (nn > 2 * CH) __threadfence();  // rare: the list entries written above are read back below
    PHASE_STAMP(3);
    // ------------------------------------------------------------------ forward
    v4d D[RBL][CBS];
#pragma unroll
    for (int rb = 0; rb < RBL; rb++)
#pragma unroll
        for (int cb = 0; cb < CBS; cb++) D[rb][cb] = (v4d){0.0, 0.0, 0.0, 0.0};
    const double ang = shear ? SGPR_TINY_ANGLE : 0.0;
    for (int t0 = 0; t0 < nn; t0 += CH) {
        const int cnt = min(CH, nn - t0);
        const bool on = lane < cnt;
        const int t = t0 + lane;
        double r[3] = {1.0, 0.0, 0.0};
        int s = 0;
        if (on) {
            if (t0 == 0) { r[0] = h0r[0]; r[1] = h0r[1]; r[2] = h0r[2]; s = h0s; }
            else if (t0 == CH) {
                r[0] = hit1[4 * lane]; r[1] = hit1[4 * lane + 1]; r[2] = hit1[4 * lane + 2];
                s = __double2loint(hit1[4 * lane + 3]);
            } else {
                const size_t e = (size_t)i * maxnn + t;
                const int j = n.nbr_j[e];
                const int cd = n.nbr_shift[e];
                const int f0 = (int)(int8_t)(cd & 0xff), f1 = (int)(int8_t)((cd >> 8) & 0xff), f2 = (int)(int8_t)((cd >> 16) & 0xff);
                s = (cd >> 24) & 0xff;
                r[0] = a.pos[3 * (size_t)j] - xi + (f0 * h[0] + f1 * h[3] + f2 * h[6]);
                r[1] = a.pos[3 * (size_t)j + 1] - yi + (f0 * h[1] + f1 * h[4] + f2 * h[7]);
                r[2] = a.pos[3 * (size_t)j + 2] - zi + (f0 * h[2] + f1 * h[5] + f2 * h[8]);
            }
        }
        const double u = unit_of<ST>(a, s);
        const double iu = inv_unit_of<ST>(a, s);
        const double x = r[0] * iu, y = r[1] * iu, z = r[2] * iu;
        const double d = sqrt(x * x + y * y + z * z);
        double f[N1], gg, dg;
        const double ex = exp(-0.5 * d * d);
        radial_ex<NMAX>(d, u, a.rc, a.irc, ex, f, gg, dg);
        double Y[LL];
        Harm<LMAX> hm;
        hm.eval(x, y - ang * z, ang * y + z, Y);
        if (on && a.prec) {
            // pair record for the reverse pass: displacement and exp(-d^2/2), read back coalesced
            double2 *dst = (double2 *)(a.prec + ((size_t)i * maxnn + t) * 4);
            dst[0] = make_double2(r[0], r[1]);
            dst[1] = make_double2(r[2], ex);
        }
        wave_sync();  // the region's previous user (list build / previous tile) is done
        if (lane < CH) {
#pragma unroll
            for (int q = 0; q < N1; q++) fl[lane * N1 + q] = on ? f[q] : 0.0;
#pragma unroll
            for (int k = 0; k < LL; k++) Yl[lane * LLP + k] = on ? Y[k] : 0.0;
            sl[lane] = on ? s : -1;
        }
        wave_sync();
        // c[lm][(s,n)] += sum over the tile's neighbours, four per MFMA: A = Y[t][lm], B = f[t][n] [s_t = s]
        // (four k-steps per trip, their operands requested together: the rows beyond the count hold zeros, CH is a
        // multiple of 16; one k-step per trip was a chain of LDS round trips, one per MFMA)
        for (int g16 = 0; g16 < cnt; g16 += 16) {
            double av[4][RBL], bv[4][CBS];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int tr = g16 + 4 * q + (lane >> 4);
                const int st = sl[tr];
#pragma unroll
                for (int rb = 0; rb < RBL; rb++) {
                    const int lm = 16 * rb + (lane & 15);
                    av[q][rb] = (RBL * 16 == LL || lm < LL) ? Yl[tr * LLP + (lm < LL ? lm : 0)] : 0.0;
                }
#pragma unroll
                for (int cb = 0; cb < CBS; cb++) {
                    const int c = 16 * cb + (lane & 15);
                    const double fv = fl[tr * N1 + (c < UC ? c % N1 : 0)];
                    bv[q][cb] = (c < UC && st == c / N1) ? fv : 0.0;
                }
            }
#pragma unroll
            for (int q = 0; q < 4; q++)
#pragma unroll
                for (int rb = 0; rb < RBL; rb++)
#pragma unroll
                    for (int cb = 0; cb < CBS; cb++)
                        D[rb][cb] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[q][rb], bv[q][cb], D[rb][cb], 0, 0, 0);
        }
    }
    wave_sync();
    PHASE_STAMP(4);
    // c out of the accumulators (C/D map: col = lane & 15, row = (lane >> 4) + 4 * reg): LDS for the power
    // spectrum, memory for the reverse pass
    constexpr int L1 = LMAX + 1;
    constexpr int UMAX = ST * N1;
    constexpr bool PMF = UMAX <= 16;             // the power spectrum as C_l . C_l^T on the matrix pipe (below)
    constexpr int CLS = PMF ? LL + 1 : LL;       // row stride of c in LDS (padded: conflict-free operand reads)
#pragma unroll
    for (int rb = 0; rb < RBL; rb++)
#pragma unroll
        for (int cb = 0; cb < CBS; cb++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int lm = 16 * rb + (lane >> 4) + 4 * q, c = 16 * cb + (lane & 15);
                if ((RBL * 16 == LL || lm < LL) && (CBS * 16 == UC || c < UC)) {
                    const int idx = (c / N1) * NSLOT + (c % N1) * LL + lm;
                    cl[PMF ? c * CLS + lm : idx] = D[rb][cb][q];
                }
            }
    wave_sync();
    // c for the reverse pass goes out from LDS, lane = consecutive entry: in the accumulator layout neighbouring lanes
    // hold neighbouring CHANNELS (16 doubles apart in memory) and every store instruction touched 64 cache lines
    if (a.C)
        for (int idx = lane; idx < species_in_use<LMAX, NMAX, ST>(a) * NSLOT; idx += 64)
            a.C[(size_t)ia * a.CS + idx] = cl[PMF ? (idx / LL) * CLS + idx % LL : idx];
    // packed power spectrum: entry e = pair(u<=v)*L1 + l : coef * sum_{lm in l} c[u][lm] c[v][lm].
    double nrm2 = 0.0;
    const int npair = a.Dc / L1;
    if constexpr (PMF) {
        // P_l = C_l . C_l^T, C_l = c[:, l^2 .. l^2 + 2l]: rows and columns are the channels u, v (<= 16), k runs over the
        // m of the shell, so A[row = lane & 15][k = lane >> 4] and B[k][col = lane & 15] are the SAME value in every lane
        // — one ds_read_b64 per MFMA, 1 + 1 + 2 + 2 of them for l = 0..3, where one lane per (u, v) pair read 32 values
        // of c (64 reads per lane with two pairs: every wave of the CU does this at the same moment and LDS bytes bound it).
        // The lane gets P_l[u = (lane >> 4) + 4 r][v = lane & 15] and keeps u <= v < U; the coefficient of the entry is
        // requested first (its address depends on the lane only).
        const int Ur = species_in_use<LMAX, NMAX, ST>(a) * N1, vch = lane & 15;
        double cf[L1][4];
        int eidx[4];
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const int u = (lane >> 4) + 4 * r;
            const bool keep = u <= vch && vch < Ur;
            eidx[r] = keep ? (u * Ur - (u * (u - 1)) / 2 + (vch - u)) * L1 : -1;
#pragma unroll
            for (int l = 0; l < L1; l++) cf[l][r] = keep ? a.pack[eidx[r] + l].coef : 0.0;
        }
        v4d P[L1];
#pragma unroll
        for (int l = 0; l < L1; l++) {
            P[l] = (v4d){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int ks = 0; ks < (2 * l + 4) / 4; ks++) {
                const int kk = 4 * ks + (lane >> 4);
                const double av = (vch < Ur && kk < 2 * l + 1) ? cl[vch * CLS + l * l + kk] : 0.0;
                P[l] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, av, P[l], 0, 0, 0);
            }
        }
#pragma unroll
        for (int l = 0; l < L1; l++)
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const double pvv = P[l][r] * cf[l][r];
                cf[l][r] = pvv;
                nrm2 += pvv * pvv;
            }
        nrm2 = wave_sum(nrm2);
        const double nrm = sqrt(nrm2);
        const double inv = nn > 0 ? 1.0 / (nrm + SGPR_EPS) : 0.0;
#pragma unroll
        for (int r = 0; r < 4; r++)
            if (eidx[r] >= 0) {
#pragma unroll
                for (int l = 0; l < L1; l++) a.Pn[(size_t)ia * a.Dpad + eidx[r] + l] = cf[l][r] * inv;
            }
        for (int e = a.Dc + lane; e < a.Dpad; e += 64) a.Pn[(size_t)ia * a.Dpad + e] = 0.0;
        if (lane == 0) {
            a.norm[ia] = nn > 0 ? nrm : 0.0;
            if (a.shear) a.shear[ia] = shear ? 1 : 0;
        }
        PHASE_STAMP(5);
        return;
    }
    // (more than 16 channels) one lane per (u,v) pair, the l shells and their m sums statically unrolled.
    constexpr int MAXP = ((UMAX * (UMAX + 1)) / 2 + 63) / 64;
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
    PHASE_STAMP(5);
}

// =========================================================================== backward
// Reverse pass: ONE kernel, one wave64 per atom i (reference: the torch.autograd pass of
// calculator/active.py:587-599 through descriptor/sesoap.py:161-260).
//  phase A  G~ = dE/dp^ (packed, from the W GEMM) -> dE/dp -> dE/dc[s][n][lm] of this atom, kept in LDS.
//  phase B  lane = neighbour t at r = x_j - x_i + off.cell:  g_t = dE_i/dr_it  (ONE evaluation per
//           ordered pair).  The contraction with the coefficients,
//               gY[t][lm] = sum_n f_n(t)  dE/dc[s_t][n][lm],   hY[t][lm] = sum_n f_n'(t) dE/dc[s_t][n][lm],
//           is [neighbours x radial] . [radial x lm]: it runs on v_mfma_f64_16x16x4 (rows = 16
//           neighbours, K = radial channel, columns = 16 lm; rows of other species are masked in the A
//           operand), and the per-neighbour rows come back to their lanes through LDS.  The lane then
//           needs only the harmonic recurrence state (q, A, B) next to the streamed row: 128 VGPRs,
//           four waves per SIMD, the whole 4096-atom grid resident at once.
//  output   GATHER: g_t is handed to the neighbour at ITS list position, G[j_t][rev_t] = g_t, with the
//           reverse index from the neighbour-list build (neighbor.hip); the wave keeps sum_t g_t.  The
//           step's last kernel forms  F_i = sum_t g_it - sum_t' G[i][t']  from one coalesced row: 32 B
//           per pair, no atomics, fixed summation order.
//           !GATHER (atoms sharded over ranks): F_j -= g_t by fp64 atomics into the all-atom buffer the
//           ranks all-reduce, F_i += sum_t g_t by the wave.
//           Virial: sum_t r (x) g_t, one partial per workgroup.
template <int LMAX, int NMAX>
struct RevDims {
    static constexpr int N1 = NMAX + 1, L1 = LMAX + 1, LL = L1 * L1, NSLOT = N1 * LL;
    static constexpr int KS = (N1 + 3) / 4;      // MFMA K-steps over the radial channels
    static constexpr int CB = (LL + 15) / 16;    // MFMA column blocks over lm
    static constexpr int CH = 48, RB = CH / 16;  // neighbours per tile, MFMA row blocks
    static constexpr int SP = LL | 1;            // staged row stride (odd: conflict-free ds_read_b64 by row)
    static constexpr int FS = 8 * KS + 1;        // radial staging row [f | f'] (odd stride)
};

// doubles of the per-wave scratch region that phase A and the stages of phase B share
template <int LMAX, int NMAX, int ST>
static int rev_region_doubles(int Dpad)
{
    using RD = RevDims<LMAX, NMAX>;
    const int UT = ST * RD::N1;
    const int dc = ST * RD::NSLOT + (ST <= 4 ? UT * UT * RD::L1 : Dpad);
    int r = RD::CH * RD::SP;
    if (RD::CH * RD::FS > r) r = RD::CH * RD::FS;
    if (12 * RD::CH > r) r = 12 * RD::CH;
    if (dc > r) r = dc;
    return (r + 1) & ~1;
}

// ROWS: the training-rows form (a batch of inducing columns over blockIdx.y, the seed formed from Aw and P^m);
// compiled apart so that the predict path carries none of its bookkeeping
// COV: the first `n_cov` workgroups of the launch are not atoms but tiles of the covloss product |choli k_i|^2
// (gemm_tile.inc, row-square epilogue): they depend on K_nm only, as this kernel depends on W only, and they are bound by
// the matrix pipe where the reverse pass is bound by vector issue and latency — in one launch the two fill each other's
// gaps, and the W product that used to share a launch with them gets shorter (no event hops: the side-stream form of
// the same idea paid more for its two events than the overlap returned).
template <int LMAX, int NMAX, int ST, bool GATHER, bool ROWS, bool COV = false>
__global__ __launch_bounds__(256, 4) void desc_rev_kernel(DescArgs a, GemmArgs gc, int n_cov)
{
    extern __shared__ double smem[];
    const int S_use = species_in_use<LMAX, NMAX, ST>(a);
    if constexpr (COV) {
        if ((int)blockIdx.x < n_cov) {
            using GL = GemmLds<EPI_ROWSQ, 1, 16>;
            gemm_tile_body<EPI_ROWSQ, 1, 16>(gc, (int)blockIdx.x, smem, smem + GL::NBUF * GL::ASZ);
            return;
        }
    }
    const int bx = COV ? (int)blockIdx.x - n_cov : (int)blockIdx.x;      // workgroup of the reverse pass
    const int nbx = COV ? (int)gridDim.x - n_cov : (int)gridDim.x;
    const double *const rows_aw = ROWS ? a.rows_aw : nullptr;
    using RD = RevDims<LMAX, NMAX>;
    constexpr int N1 = RD::N1, L1 = RD::L1, LL = RD::LL, NSLOT = RD::NSLOT;
    constexpr int KS = RD::KS, CB = RD::CB, CH = RD::CH, RB = RD::RB, SP = RD::SP, FS = RD::FS;
    constexpr int SPL = (NSLOT + 63) / 64;
    // atoms (waves) per workgroup: four; ONE for the sixteen-slot instantiation, whose per-wave region — dE/dc and a W row of
    // 8320 doubles — is 83 KB (launched with 64 threads: run_bwd)
    constexpr int WPW = SGPR_REV_WPW(ST);
    __shared__ double vred[4][9];
    // wave index through readfirstlane: everything derived from it (atom index, LDS bases, row
    // addresses) is then scalar and stays out of the VGPR budget
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int qd = xcd_quad(bx, nbx, a.xq);  // (the virial partial below stays at the quad's index: same sums as ever)
    const int ia = qd * WPW + wave;
    const bool active = ia < a.N;
    const int perwave = ST * NSLOT + a.rsz + CH / 2;
    double *dcl = smem + (size_t)wave * perwave;  // [ST][NSLOT] dE/dc of this atom
    double *R = dcl + ST * NSLOT;                 // shared scratch region (a.rsz doubles)
    int *sl = (int *)(R + a.rsz);                 // [CH] species slot per tile row
    const int gi = a.first + (active ? ia : 0) * a.stride;
    // batch entry (training rows): column, seed scale and output bases
    const int bq = ROWS ? a.rows_cols[blockIdx.y] : 0;
    // training rows: k(i, q) is identically zero for an atom of another species than the column's — the wave has no
    // pair gradient to hand over (the last kernel only reads the slots written by atoms of the column's species) and
    // goes straight to the tail, where it contributes zeros to the own-force and virial sums
    const bool off_species = ROWS && active && a.slot[gi] != a.rows_colslot[bq];
    const int nn = (active && !off_species) ? a.nn[gi] : 0;
    const double aw_q = (ROWS && active) ? rows_aw[(size_t)ia * a.rows_ld + bq] : 1.0;
    double *Gb = a.G, *Fnbr_b = a.Fnbr, *Fself_b = a.Fself, *vir_b = a.vir_part;
    if constexpr (ROWS) {
        if (Gb) Gb += blockIdx.y * a.g_stride;
        if (Fnbr_b) Fnbr_b += blockIdx.y * a.f_stride;
        if (Fself_b) Fself_b += blockIdx.y * a.f_stride;
        vir_b += blockIdx.y * a.v_stride;
    }
    double tot = 0.0;  // lanes < 48 with (lane & 3) == 0: running sum of virial component / force component lane >> 2
    bool zero_dc = false;  // dE/dp^ of this atom is identically zero (training rows: the column belongs to another
                           // species): every pair gradient is zero, only the hand-over slots have to be cleared

    PHASE_STAMP(0);
    // Everything phase A reads of this atom (norm, W, p^, pack table, c) is requested NOW, next to the neighbour count:
    // the addresses depend on the atom index only, and behind `nn > 0` and `nrm > 0` they were the third cold miss
    // of a chain of three (nn -> norm -> rows): 15k cycles for 36 FMAs per lane.
    constexpr int MAXE_H = ((ST * RD::N1 * (ST * RD::N1 + 1)) / 2 * RD::L1 + 63) / 64;
    constexpr bool HOIST = MAXE_H <= 10;
    const double nrm_h = active ? a.norm[ia] : 0.0;
    double wv_h[HOIST ? MAXE_H : 1], pv_h[HOIST ? MAXE_H : 1], cv_h[HOIST ? ST * SPL : 1];
    PackEntry pe_h[HOIST ? MAXE_H : 1];
    if constexpr (HOIST) {
        const double *Wi = ROWS ? a.rows_pm + (size_t)bq * a.Dpad : a.W + (size_t)(active ? ia : 0) * a.Dpad;
        const double *Pi = a.Pn + (size_t)(active ? ia : 0) * a.Dpad;
#pragma unroll
        for (int k = 0; k < MAXE_H; k++) {
            const int e = lane + 64 * k;
            const bool in = e < a.Dc && active;
            wv_h[k] = in ? aw_q * Wi[e] : 0.0;
            pv_h[k] = in ? Pi[e] : 0.0;
            pe_h[k] = a.pack[in ? e : 0];
        }
#pragma unroll
        for (int s = 0; s < ST; s++)
#pragma unroll
            for (int k = 0; k < SPL; k++) {
                const int slot = lane + 64 * k;
                cv_h[s * SPL + k] = (active && s < S_use && (SPL * 64 == NSLOT || slot < NSLOT)) ? a.C[(size_t)ia * a.CS + s * NSLOT + slot] : 0.0;
            }
    }
    if (nn > 0) {
        // ---------------------------------------------------------------- phase A: dE/dc -> dcl
        const double nrm = nrm_h;
        if (!(nrm > 0.0) || (ROWS && aw_q == 0.0)) {  // (aw_q: wave-uniform; 0 for atoms of other species)
            for (int k = lane; k < ST * NSLOT; k += 64) dcl[k] = 0.0;
            zero_dc = true;
        } else {
            const int Ur = S_use * N1;  // channels of the packed layout (pack table built with the real S)
            // Up to 4 species the packed gradient is expanded into the full symmetric [u][v][l] array in
            // LDS (stride-1 addressing in the contraction); 8 slots keep the packed form.
            constexpr bool EXPAND = ST <= 4;
            constexpr int UT = ST * N1;
            double *cl = R;                 // [ST][NSLOT]  c  ( = [u][lm], u = s*N1+n )
            double *gl = R + ST * NSLOT;    // EXPAND: [UT][UT][L1] else [Dpad]:  dE/dp~ * coef * (1 or 2)
            const double sden = nrm + SGPR_EPS;
            constexpr int MAXE = ((UT * (UT + 1)) / 2 * L1 + 63) / 64;
            const double *Wi = ROWS ? a.rows_pm + (size_t)bq * a.Dpad : a.W + (size_t)ia * a.Dpad;
            const double *Pi = a.Pn + (size_t)ia * a.Dpad;
            if constexpr (MAXE <= 10) {
                // (the global reads of this atom — W, p^, pack entries, c — were issued at the top of the kernel)
                static_assert(MAXE == MAXE_H, "hoisted loads");
                double (&wv)[MAXE] = wv_h, (&pv)[MAXE] = pv_h;
                PackEntry (&pe)[MAXE] = pe_h;
#pragma unroll
                for (int s = 0; s < ST; s++)
#pragma unroll
                    for (int k = 0; k < SPL; k++) {
                        const int slot = lane + 64 * k;
                        if (SPL * 64 == NSLOT || slot < NSLOT) cl[s * NSLOT + slot] = cv_h[s * SPL + k];
                    }
                if constexpr (ROWS) {
                    bool nz = false;
#pragma unroll
                    for (int k = 0; k < MAXE; k++) nz |= wv[k] != 0.0;
                    zero_dc = !__any(nz);
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
                    if (e < a.Dc && !zero_dc) {
                        const double gv = (wv[k] - pv[k] * corr) * isden * pe[k].coef * (pe[k].u == pe[k].v ? 2.0 : 1.0);
                        if constexpr (EXPAND) {
                            gl[(pe[k].u * UT + pe[k].v) * L1 + pe[k].l] = gv;
                            gl[(pe[k].v * UT + pe[k].u) * L1 + pe[k].l] = gv;
                        } else
                            gl[e] = gv;
                    }
                }
            } else {
                // many species / high lmax: too many entries per lane to hold in registers
                double pw = 0.0;
                for (int e = lane; e < a.Dc; e += 64) pw += aw_q * Wi[e] * Pi[e];
                pw = wave_sum(pw);
                const double corr = pw * sden / nrm;
                for (int e = lane; e < a.Dc; e += 64) {
                    const PackEntry pe = a.pack[e];
                    const double gv = (aw_q * Wi[e] - Pi[e] * corr) / sden * pe.coef * (pe.u == pe.v ? 2.0 : 1.0);
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
                            cl[s * NSLOT + slot] = s < S_use ? a.C[(size_t)ia * a.CS + s * NSLOT + slot] : 0.0;
                    }
            }
            wave_sync();
            // dE/dc[u][lm] = sum_v G[u][v][l] c[v][lm]
            // (the column of c is read once per slot and kept for all species: this contraction is bound by LDS bytes —
            // every wave of the CU is here at the same time — and the stores to dcl kept the compiler from doing it)
            if (!zero_dc)
#pragma unroll
            for (int k = 0; k < SPL; k++) {
                const int slot_c = lane + 64 * k;
                double cvv[EXPAND ? UT : 1];
                if constexpr (EXPAND) {
                    const int lm_c = (SPL * 64 == NSLOT || slot_c < NSLOT) ? slot_c % LL : 0;
#pragma unroll
                    for (int v = 0; v < UT; v++) cvv[v] = v < Ur ? cl[v * LL + lm_c] : 0.0;
                }
#pragma unroll
                for (int s = 0; s < ST; s++) {
                    const int slot = lane + 64 * k;
                    if (SPL * 64 == NSLOT || slot < NSLOT) {
                        double d = 0.0;
                        if (s < S_use) {
                            const int n = slot / LL, lm = slot % LL;
                            int l = 0;
#pragma unroll
                            for (int q = 1; q <= LMAX; q++) l += (lm >= q * q) ? 1 : 0;
                            const int u = s * N1 + n;
                            if constexpr (EXPAND) {
                                const double *gu = gl + u * UT * L1 + l;
#pragma unroll
                                for (int v = 0; v < UT; v++) d += (v < Ur ? gu[v * L1] : 0.0) * cvv[v];
                            } else {
                                for (int v = 0; v < Ur; v++) {
                                    const int lo = min(u, v), hi = max(u, v);
                                    const int pair = lo * Ur - (lo * (lo - 1)) / 2 + (hi - lo);
                                    d += gl[pair * L1 + l] * cl[v * LL + lm];
                                }
                            }
                        }
                        dcl[s * NSLOT + slot] = d;
                    }
                }
            }
        }
        PHASE_STAMP(1);
        // ---------------------------------------------------------------- phase B: pair terms
        const double ang = a.shear[ia] ? SGPR_TINY_ANGLE : 0.0;
        double *stage = R;  // [CH][SP] one staged row set (hY, then gY); earlier in a tile: [CH][FS] radial rows
        // training rows: an atom of another species than the column's has k(i,q) = 0 identically: no pair terms, and
        // nothing to hand over — the finalize of a rows batch only adds the hand-overs of neighbours of the column's
        // species (their species is in the list's code word), so the slots this atom would have cleared are never read
        int nn_b = nn;
        if constexpr (ROWS) {
            if (a.slot[gi] != a.rows_colslot[bq]) nn_b = 0;
        }
        for (int t0 = 0; t0 < nn_b; t0 += CH) {
            const int cnt = min(CH, nn_b - t0);
            const bool on = lane < cnt;
            const int t = t0 + lane;
            double r[3] = {1.0, 0.0, 0.0}, ex = 0.0;
            int s = 0, j = 0, rvp = 0;
            if (on) {
                // the forward pass left (r, exp(-d^2/2)) of every pair: one coalesced 32-B read per lane
                const size_t e = (size_t)gi * a.maxnn + t;
                const double2 *src = (const double2 *)(a.prec + e * 4);
                const double2 p0 = src[0], p1 = src[1];
                r[0] = p0.x; r[1] = p0.y; r[2] = p1.x; ex = p1.y;
                s = (a.nbr_shift[e] >> 24) & 0xff;
                j = a.nbr_j[e];
                if constexpr (GATHER) {
                    // where atom j keeps this pair: the position of i in j's list of THIS step = the number of
                    // j's candidates before i that are inside the cutoff now (popcount of j's hit mask).
                    // (Clamped: an attempt that overflowed a capacity leaves these words unwritten; the host
                    // discards its results.)
                    const int c = min(max(a.cidx[e], 0), a.maxnn - 1);
                    const int hs = min(max(a.aux[(size_t)gi * a.maxnn + c], 0), a.t_stride - 1);
                    const int cr = min((int)a.T[(size_t)gi * a.t_stride + hs], a.maxnn - 1);
                    const unsigned long long *hj = a.hm + (size_t)j * a.hmw;
                    int pcount = __popcll(hj[cr >> 6] & ((1ull << (cr & 63)) - 1ull));
                    for (int w = 0; w < (cr >> 6); w++) pcount += __popcll(hj[w]);
                    rvp = min(pcount, a.maxnn - 1);
                }
            }
            if (zero_dc) {  // wave-uniform
                if constexpr (GATHER) {
                    if (on) {
                        double2 *dst = (double2 *)(Gb + ((size_t)j * a.maxnn + rvp) * 4);
                        dst[0] = make_double2(0.0, 0.0);
                        dst[1] = make_double2(0.0, 0.0);
                    }
                }
                continue;
            }
            const double u = unit_of<ST>(a, s);
            const double iu = inv_unit_of<ST>(a, s);
            const double x = r[0] * iu, y = r[1] * iu, z = r[2] * iu;
            const double d = sqrt(x * x + y * y + z * z);
            const double id = 1.0 / d;
            double f[N1], g, dg;
            radial_ex<NMAX>(d, u, a.rc, a.irc, ex, f, g, dg);
            wave_sync();  // the previous user of the region (phase A / the previous tile) is done
            if (on) {
                // f_n = g d^(2n);  f_n' = dg d^(2n) + 2n g d^(2n-1)
                double rpow = 1.0;
                const double rho = d * d;
#pragma unroll
                for (int n = 0; n < 4 * KS; n++) {
                    stage[lane * FS + n] = n < N1 ? f[n < N1 ? n : 0] : 0.0;
                    stage[lane * FS + 4 * KS + n] = n < N1 ? dg * rpow + (n ? g * 2.0 * n * rpow * id : 0.0) : 0.0;
                    rpow *= rho;
                }
            }
            if (lane < CH) sl[lane] = on ? s : -1;
            wave_sync();
            // MFMA A operands: lane = (row i = lane & 15, k = lane >> 4) of each row block
            double af[RB][KS], ah[RB][KS];
            int sr[RB];
#pragma unroll
            for (int rb = 0; rb < RB; rb++) {
                const int row = 16 * rb + (lane & 15);
                sr[rb] = sl[row];
#pragma unroll
                for (int ks = 0; ks < KS; ks++) {
                    af[rb][ks] = stage[row * FS + 4 * ks + (lane >> 4)];
                    ah[rb][ks] = stage[row * FS + 4 * KS + 4 * ks + (lane >> 4)];
                }
            }
            wave_sync();  // radial rows are in registers: the region is free for the staged outputs
            Harm<LMAX> h;
            h.hc = launder(&c_hc);
            h.prepare(x, y - ang * z, ang * y + z);
            double dEdd = 0.0, gxs = 0.0, gys = 0.0, gzs = 0.0;
#pragma unroll
            for (int pass = 0; pass < 2; pass++) {
                // pass 0: hY = f'.dC (radial derivative part), pass 1: gY = f.dC (angular part)
#pragma unroll
                for (int cb = 0; cb < CB; cb++) {
                    // row block outermost: one accumulator tile (4 doubles) live at a time; the B
                    // fragment is re-read from LDS for every row block (one ds_read_b64)
                    const int n_b = lane >> 4, lm = 16 * cb + (lane & 15);
#pragma unroll
                    for (int rb = 0; rb < RB; rb++) {
                        v4d D = (v4d){0.0, 0.0, 0.0, 0.0};
                        for (int sp = 0; sp < S_use; sp++) {
                            if (__ballot(sr[rb] == sp) == 0ull) continue;
#pragma unroll
                            for (int ks = 0; ks < KS; ks++) {
                                const int n = 4 * ks + n_b;
                                const bool ok = n < N1 && lm < LL;
                                const double bv = dcl[sp * NSLOT + (ok ? n * LL + lm : 0)];
                                const double av = pass == 0 ? ah[rb][ks] : af[rb][ks];
                                D = __builtin_amdgcn_mfma_f64_16x16x4f64(sr[rb] == sp ? av : 0.0, ok ? bv : 0.0, D, 0, 0, 0);
                            }
                        }
                        // C/D map of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg
                        if (CB * 16 == LL || lm < LL) {
#pragma unroll
                            for (int q = 0; q < 4; q++) stage[(16 * rb + (lane >> 4) + 4 * q) * SP + lm] = D[q];
                        }
                    }
                }
                wave_sync();
                if (on) {
                    if (pass == 0) dEdd = h.dot(stage + lane * SP);
                    else {
                        h.hc = launder(h.hc);
                        h.backward(stage + lane * SP, gxs, gys, gzs);
                    }
                }
                wave_sync();
            }
            // inverse shear (ylm.py:203-213) + radial part, then 1/u
            const double rad = dEdd * id;
            double gr[3];
            gr[0] = (gxs + rad * x) * iu;
            gr[1] = (gys + ang * gzs + rad * y) * iu;
            gr[2] = (-ang * gys + gzs + rad * z) * iu;
            if (on) {
                if constexpr (GATHER) {
                    // handed to atom j at ITS list position: the last kernel reads whole rows, coalesced
                    double2 *dst = (double2 *)(Gb + ((size_t)j * a.maxnn + rvp) * 4);
                    dst[0] = make_double2(gr[0], gr[1]);
                    dst[1] = make_double2(gr[2], 0.0);
                } else {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        // (fixed point: the same bits whatever order the ranks' waves arrive in — sgpr_internal.h)
                        if (!(fabs(gr[k]) < SGPR_FIX_LIMIT) && a.stat) atomicMax(&a.stat[3], 3);
                        atomicAdd((unsigned long long *)&Fnbr_b[3 * (size_t)j + k], (unsigned long long)__double2ll_rn(-gr[k] * SGPR_FIX_SCALE));
                    }
                }
            }
            // 9 virial sums (+ 3 force sums in the sharded form) through the region: [12][CH], then 48
            // lanes each add a quarter of a row and two shuffles finish it
            if (lane < CH) {
                const double rv[3] = {x * u, y * u, z * u};  // = r to rounding (r itself is not kept live)
#pragma unroll
                for (int p = 0; p < 3; p++)
#pragma unroll
                    for (int q = 0; q < 3; q++) stage[(3 * p + q) * CH + lane] = on ? rv[p] * gr[q] : 0.0;
#pragma unroll
                for (int k = 0; k < 3; k++) stage[(9 + k) * CH + lane] = on ? gr[k] : 0.0;
            }
            wave_sync();
            if (lane < 48) {
                const double *src = stage + (lane >> 2) * CH + (lane & 3) * (CH / 4);
                double sacc = 0.0;
#pragma unroll
                for (int i = 0; i < CH / 4; i++) sacc += src[i];
                tot += sacc;
            }
        }
    }
    PHASE_STAMP(2);
    tot += __shfl_xor(tot, 1, 64);
    tot += __shfl_xor(tot, 2, 64);
    if (lane < 48 && (lane & 3) == 0) {
        const int k = lane >> 2;
        if (k < 9) vred[wave][k] = tot;
        else if (active) Fself_b[3 * (size_t)gi + (k - 9)] = tot;
    }
    __syncthreads();
    if (wave == 0 && lane < 9)
        vir_b[(size_t)lane * nbx + qd] = WPW == 1 ? vred[0][lane] : vred[0][lane] + vred[1][lane] + vred[2][lane] + vred[3][lane];
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
static int run_bwd(DescArgs a, hipStream_t st, const GemmParams *cov = nullptr)
{
    if (a.N <= 0) return 0;
    using RD = RevDims<LMAX, NMAX>;
    a.rsz = rev_region_doubles<LMAX, NMAX, ST>(a.Dpad);
    constexpr int WPW = SGPR_REV_WPW(ST);
    if (WPW == 1) a.xq = 0;
    size_t lds = sizeof(double) * WPW * (size_t)(ST * RD::NSLOT + a.rsz + RD::CH / 2);
    static size_t attr_set[6] = {0, 0, 0, 0, 0, 0};
    const bool gather = a.G != nullptr, rows = a.rows_aw != nullptr;
    const bool with_cov = WPW == 4 && cov && !rows && cov->tiles && cov->ntiles > 0 && cov->bm == 32 && cov->kd == 16;
    GemmArgs gc = {};
    int n_cov = 0;
    if (with_cov) {
        gc.p = *cov;
        gc.ieta = -1;
        n_cov = cov->ntiles;
        lds = std::max(lds, sizeof(double) * (size_t)GemmLds<EPI_ROWSQ, 1, 16>::DOUBLES);
    }
    const int which = with_cov ? 4 + gather : 2 * gather + rows;
    const void *fn = with_cov ? (gather ? (const void *)desc_rev_kernel<LMAX, NMAX, ST, true, false, true>
                                        : (const void *)desc_rev_kernel<LMAX, NMAX, ST, false, false, true>)
                   : gather ? (rows ? (const void *)desc_rev_kernel<LMAX, NMAX, ST, true, true>
                                    : (const void *)desc_rev_kernel<LMAX, NMAX, ST, true, false>)
                            : (rows ? (const void *)desc_rev_kernel<LMAX, NMAX, ST, false, true>
                                    : (const void *)desc_rev_kernel<LMAX, NMAX, ST, false, false>);
    if (attr_set[which] < lds) {
        (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set[which] = lds;
    }
    const dim3 grid((a.N + WPW - 1) / WPW + n_cov, rows ? a.batch : 1), blk(64 * WPW);
    if (with_cov && gather) hipLaunchKernelGGL((desc_rev_kernel<LMAX, NMAX, ST, true, false, true>), grid, blk, lds, st, a, gc, n_cov);
    else if (with_cov) hipLaunchKernelGGL((desc_rev_kernel<LMAX, NMAX, ST, false, false, true>), grid, blk, lds, st, a, gc, n_cov);
    else if (gather && rows) hipLaunchKernelGGL((desc_rev_kernel<LMAX, NMAX, ST, true, true>), grid, blk, lds, st, a, gc, n_cov);
    else if (gather) hipLaunchKernelGGL((desc_rev_kernel<LMAX, NMAX, ST, true, false>), grid, blk, lds, st, a, gc, n_cov);
    else if (rows) hipLaunchKernelGGL((desc_rev_kernel<LMAX, NMAX, ST, false, true>), grid, blk, lds, st, a, gc, n_cov);
    else hipLaunchKernelGGL((desc_rev_kernel<LMAX, NMAX, ST, false, false>), grid, blk, lds, st, a, gc, n_cov);
    return 0;
}

static int st_of(int S) { return S <= 1 ? 1 : S <= 2 ? 2 : S <= 3 ? 3 : S <= 4 ? 4 : S <= 8 ? 8 : 16; }

#define DISPATCH_LNS(FN, ...)                                                              \
    do {                                                                                   \
        const int stv = st_of(p.S);                                                        \
        if (p.lmax == 3 && p.nmax == 3) {                                                  \
            if (stv == 1) return FN(3, 3, 1, __VA_ARGS__);                                  \
            if (stv == 2) return FN(3, 3, 2, __VA_ARGS__);                                  \
            if (stv == 3) return FN(3, 3, 3, __VA_ARGS__);                                  \
            if (stv == 4) return FN(3, 3, 4, __VA_ARGS__);                                  \
            if (stv == 8) return FN(3, 3, 8, __VA_ARGS__);                                  \
            return FN(3, 3, 16, __VA_ARGS__);                                               \
        }                                                                                  \
        if (stv > 8) return -6;   /* nine to sixteen species: the reference's default (3, 3) only */                                                                                  \
        if (p.lmax == 2 && p.nmax == 2) {                                                  \
            if (stv <= 2) return FN(2, 2, 2, __VA_ARGS__);                                  \
            if (stv <= 4) return FN(2, 2, 4, __VA_ARGS__);                                  \
            return FN(2, 2, 8, __VA_ARGS__);                                                \
        }                                                                                  \
        if (p.lmax == 4 && p.nmax == 4) {                                                  \
            if (stv <= 2) return FN(4, 4, 2, __VA_ARGS__);                                  \
            if (stv <= 4) return FN(4, 4, 4, __VA_ARGS__);                                  \
            return FN(4, 4, 8, __VA_ARGS__);                                                \
        }                                                                                  \
        /* the other (lmax, nmax) pairs of {2,3,4}^2 (similarity/sesoap.py:10-24 takes any): */ \
        /* two instantiations each, four and eight species slots                           */ \
        if (stv <= 4) {                                                                    \
            if (p.lmax == 2 && p.nmax == 3) return FN(2, 3, 4, __VA_ARGS__);                \
            if (p.lmax == 2 && p.nmax == 4) return FN(2, 4, 4, __VA_ARGS__);                \
            if (p.lmax == 3 && p.nmax == 2) return FN(3, 2, 4, __VA_ARGS__);                \
            if (p.lmax == 3 && p.nmax == 4) return FN(3, 4, 4, __VA_ARGS__);                \
            if (p.lmax == 4 && p.nmax == 2) return FN(4, 2, 4, __VA_ARGS__);                \
            if (p.lmax == 4 && p.nmax == 3) return FN(4, 3, 4, __VA_ARGS__);                \
        } else {                                                                           \
            if (p.lmax == 2 && p.nmax == 3) return FN(2, 3, 8, __VA_ARGS__);                \
            if (p.lmax == 2 && p.nmax == 4) return FN(2, 4, 8, __VA_ARGS__);                \
            if (p.lmax == 3 && p.nmax == 2) return FN(3, 2, 8, __VA_ARGS__);                \
            if (p.lmax == 3 && p.nmax == 4) return FN(3, 4, 8, __VA_ARGS__);                \
            if (p.lmax == 4 && p.nmax == 2) return FN(4, 2, 8, __VA_ARGS__);                \
            if (p.lmax == 4 && p.nmax == 3) return FN(4, 3, 8, __VA_ARGS__);                \
        }                                                                                  \
        return -6;                                                                         \
    } while (0)

static DescArgs make_args(const DescParams &p)
{
    DescArgs a = {};
    a.N = p.N; a.Nall = p.Nall; a.first = p.first; a.stride = p.stride > 0 ? p.stride : 1; a.maxnn = p.maxnn; a.S = p.S; a.Dc = p.Dc; a.Dpad = p.Dpad; a.CS = p.CS;
    a.rc = p.rc;
    a.xq = p.xq;
    a.stat = p.stat;
    a.irc = 1.0 / p.rc;
    for (int k = 0; k < SGPR_MAX_S; k++) { a.radii_v[k] = p.radii_v[k]; a.radii_iv[k] = 1.0 / p.radii_v[k]; }
    return a;
}

#define FWD_ENV(L, N, S, a, st) run_fwd<L, N, S, true>(a, st)
#define BWD(L, N, S, a, st, cov) run_bwd<L, N, S>(a, st, cov)

template <int LMAX, int NMAX, int ST>
static int run_list_fwd(const DescArgs &a, const NlArgs &n, hipStream_t st)
{
    if (a.N <= 0) return 0;
    const size_t lds = sizeof(double) * 4 * (size_t)FwdLds<LMAX, NMAX, ST>::PW;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void *)nl_fwd_kernel<LMAX, NMAX, ST>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                  (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((nl_fwd_kernel<LMAX, NMAX, ST>), dim3((a.N + 3) / 4), dim3(256), lds, st, a, n);
    return 0;
}
#define LISTFWD(L, N, S, a, n, st) run_list_fwd<L, N, S>(a, n, st)

int launch_list_forward(const DescParams &p, const NlScratch &nl, const double *pos, const double *cell,
                        const PackEntry *pack, int *nn, int *nn_local, int *nbr_j, int *nbr_shift, double *Pn,
                        double *norm, double *C, int *shear, double *prec, hipStream_t st)
{
    DescArgs a = make_args(p);
    a.stamps = p.stamps;
    a.pos = pos; a.cell = cell; a.pack = pack; a.Pn = Pn; a.norm = norm; a.C = C; a.shear = shear; a.prec = prec;
    NlArgs n = {};
    n.grid = nl.grid; n.bin_of = nl.bin_of; n.kslot = nl.kslot; n.bin_count = nl.bin_count; n.cap = nl.cap;
    n.b_rec = nl.b_rec; n.b_aux = nl.b_aux; n.nn = nn; n.nn_local = nn_local; n.nn_raw = nl.nn_raw; n.nbr_j = nbr_j;
    n.nbr_shift = nbr_shift; n.aux = nl.aux; n.T = nl.T; n.t_stride = nl.t_stride; n.stat = nl.stat;
    n.flag = nl.flag + nl.fslot; n.rc_list = p.rc + nl.skin; n.ncand = nl.ncand; n.cand_j = nl.cand_j;
    n.cand_code = nl.cand_code; n.cidx = nl.cidx; n.hm = nl.hm; n.hmw = nl.hmw;
    DISPATCH_LNS(LISTFWD, a, n, st);
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
                               const int *shear, const double *W, const double *prec, double *G, const int *aux,
                               const unsigned short *T, int t_stride, const int *cidx, const unsigned long long *hm,
                               int hmw, double *F, double *virial, hipStream_t st, const RowsBatch *rows, const GemmParams *cov)
{
    DescArgs a = make_args(p);
    a.pos = pos; a.cell = cell; a.slot = slot; a.radii = radii; a.nn = nn; a.nbr_j = nbr_j;
    a.nbr_shift = nbr_shift; a.pack = pack; a.Pn = (double *)Pn; a.norm = (double *)norm; a.C = (double *)C;
    a.shear = (int *)shear; a.W = W; a.prec = (double *)prec;
    a.stamps = p.stamps ? p.stamps + 8 * (size_t)p.Nall : nullptr;
    // gather form (G != null): pair gradients go to G[Nall][maxnn][4], the step's last kernel sums them.
    // scatter form: F points at [Fnbr | Fself] (64-bit fixed-point integer atomics into Fnbr; sharded frames).
    a.G = G; a.aux = aux; a.T = T; a.t_stride = t_stride; a.cidx = cidx; a.hm = hm; a.hmw = hmw;
    a.Fnbr = F;
    a.Fself = F ? F + 3 * (size_t)p.Nall : nullptr;
    a.vir_part = virial;
    if (rows) {
        a.rows_aw = rows->aw; a.rows_pm = rows->pm; a.rows_cols = rows->cols; a.rows_colslot = rows->col_slot; a.rows_ld = rows->ld; a.batch = rows->batch;
        a.g_stride = rows->g_stride; a.f_stride = rows->f_stride; a.v_stride = rows->v_stride;
    }
    DISPATCH_LNS(BWD, a, st, cov);
}

#include "rows16.inc"
