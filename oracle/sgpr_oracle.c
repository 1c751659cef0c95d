/*
 * oracle/sgpr_oracle.c — CPU restatement of the reference's SGPR predict/solve hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library; the product path (autoforce_amd/) never does.
 *
 * Parity status: PINNED — every function below is checked by tests/test_oracle_golden.py
 * against vectors captured from the imported reference (the .npz files under tests/golden/, generator
 * tests/golden/gen/make_golden.py) and against the reference's own numeric KAT
 * (theforce/descriptor/soap.py:488-525).
 *
 * Each function cites the reference file:line (under /root/reference/theforce) it follows.
 * The algorithm is restated from the maths, in plain C with explicit loops; the reference's
 * torch-autograd force path (calculator/active.py:587-611) is restated as a hand-derived
 * reverse pass (exact derivative of the forward function, incl. the eps in the norm).
 *
 * Build: see oracle/Makefile  (gcc -O2 -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define ORC_MAXL 8
#define ORC_EPS 2.220446049250313e-16 /* torch.finfo(float64).eps, sesoap.py:250 */
#define ORC_TINY_ANGLE 1e-2           /* ylm.py:10 */

/* --------------------------------------------------------------------------------------
 * Ylm — descriptor/ylm.py:113-225.  Y = r^l Y_lm(theta,phi) in the packed (L+1)x(L+1)
 * layout: real part of (l,m) at Y[l][l-m], imaginary part (m>0) at Y[l-m][l]  (:124-132).
 * `shear` != 0 applies split_and_rotate_tiny_if_too_close_to_zaxis (:10-23) to the vector
 * first; dY (optional) is the Cartesian gradient w.r.t. the UNSHEARED input (:203-216).
 * The (l,m) coefficient of :103-111 is evaluated in fp64 here (the reference rounds it
 * through fp32; its autograd path, which is the parity target for forces, does not use it).
 * ------------------------------------------------------------------------------------ */
static void ylm_one(int lmax, const double v[3], int shear, double *Y /*[L1*L1]*/,
                    double *dY /*[L1*L1*3] or NULL*/)
{
    const int L1 = lmax + 1;
    double x = v[0], y = v[1], z = v[2];
    const double ang = shear ? ORC_TINY_ANGLE : 0.0;
    if (shear) {
        const double y0 = y, z0 = z;
        y = y0 - ang * z0;
        z = ang * y0 + z0;
    }
    const double rxy_sq = x * x + y * y;
    const double rxy = sqrt(rxy_sq);
    const double r = sqrt(rxy_sq + z * z);
    const double sin_t = rxy / r, cos_t = z / r;
    const double sin_p = y / rxy, cos_p = x / rxy;
    const double r2 = r * r, rs = r * sin_t, rc = r * cos_t;
    double alp[ORC_MAXL + 1][ORC_MAXL + 1];
    alp[0][0] = sqrt(1.0 / (4.0 * M_PI));
    for (int l = 1; l <= lmax; l++) {
        for (int m = 0; m < l - 1; m++) {
            const double al = sqrt((4.0 * l * l - 1.0) / (l * l - m * m));
            const double bl = -sqrt(((l - 1.0) * (l - 1.0) - m * m) / (4.0 * (l - 1.0) * (l - 1.0) - 1.0));
            alp[l][m] = al * (rc * alp[l - 1][m] + r2 * bl * alp[l - 2][m]);
        }
        alp[l][l - 1] = sqrt(2.0 * l + 1.0) * rc * alp[l - 1][l - 1];
        alp[l][l] = -sqrt(1.0 + 1.0 / (2.0 * l)) * rs * alp[l - 1][l - 1];
    }
    double sn[ORC_MAXL + 1], cs[ORC_MAXL + 1];
    sn[0] = 0.0; cs[0] = 1.0;
    if (lmax >= 1) { sn[1] = sin_p; cs[1] = cos_p; }
    for (int m = 2; m <= lmax; m++) {
        sn[m] = sin_p * cs[m - 1] + cos_p * sn[m - 1];
        cs[m] = cos_p * cs[m - 1] - sin_p * sn[m - 1];
    }
    for (int i = 0; i < L1 * L1; i++) Y[i] = 0.0;
    for (int l = 0; l <= lmax; l++)
        for (int m = 0; m <= l; m++) {
            Y[l * L1 + (l - m)] = alp[l][m] * cs[m];
            if (m > 0) Y[(l - m) * L1 + l] = alp[l][m] * sn[m];
        }
    if (!dY) return;
    /* :192-216 — spherical partials, then to Cartesian, then the inverse shear */
    for (int i = 0; i < L1; i++)
        for (int j = 0; j < L1; j++) {
            /* l,m tables (:84-93): l = max(i,j) , m = |i-j| */
            const int l = i > j ? i : j, m = i > j ? i - j : j - i;
            const double Yij = Y[i * L1 + j];
            const double Y_r = l * Yij / r;
            double Y_t = cos_t * l * Yij / sin_t;
            if (i >= 1 && j >= 1) {
                const double coef = sqrt((double)((l - m) * (l + m) * (2 * l + 1)) / (double)(2 * l - 1));
                Y_t -= r * Y[(i - 1) * L1 + (j - 1)] * coef / sin_t;
            }
            /* Y_phi = transpose(Y) * sign * m ; sign = +1 on and above the diagonal, -1 below */
            const double sign = (j >= i) ? 1.0 : -1.0;
            const double Y_p = Y[j * L1 + i] * sign * m;
            const double F_r = Y_r, F_t = Y_t / r, F_p = Y_p / (r * sin_t);
            const double cx = sin_t * cos_p * F_r + cos_t * cos_p * F_t - sin_p * F_p;
            const double cy = sin_t * sin_p * F_r + cos_t * sin_p * F_t + cos_p * F_p;
            const double cz = cos_t * F_r - sin_t * F_t;
            double *g = dY + (i * L1 + j) * 3;
            g[0] = cx;
            g[1] = cy + ang * cz;
            g[2] = -ang * cy + cz;
        }
}

/* ylm.py:10-23: the whole batch is sheared if ANY vector is within the cone */
static int needs_shear(int n, const double *xyz)
{
    for (int j = 0; j < n; j++) {
        const double tol = ORC_TINY_ANGLE * fabs(xyz[3 * j + 2]);
        if (fabs(xyz[3 * j]) < tol && fabs(xyz[3 * j + 1]) < tol) return 1;
    }
    return 0;
}

/* Batch form with the reference's [L1][L1][n] / [L1][L1][n][3] output layout. */
void orc_ylm(int lmax, int n, const double *xyz, double *Y, double *dY)
{
    const int L1 = lmax + 1, LL = L1 * L1;
    const int shear = needs_shear(n, xyz);
    double y1[(ORC_MAXL + 1) * (ORC_MAXL + 1)], d1[(ORC_MAXL + 1) * (ORC_MAXL + 1) * 3];
    for (int j = 0; j < n; j++) {
        ylm_one(lmax, xyz + 3 * j, shear, y1, dY ? d1 : NULL);
        for (int k = 0; k < LL; k++) {
            Y[(size_t)k * n + j] = y1[k];
            if (dY)
                for (int a = 0; a < 3; a++) dY[((size_t)k * n + j) * 3 + a] = d1[k * 3 + a];
        }
    }
}

/* --------------------------------------------------------------------------------------
 * SeSoap descriptor — descriptor/sesoap.py:102-260 (radial: cutoff.py:20-48 PolyCut n=2).
 * Environment = nn neighbour vectors r[nn][3], species slot s[nn] in 0..S-1, length unit
 * u[nn] (radii, sesoap.py:84-99,162).
 * flags: bit0 gaussian exp(-d^2/2) (SeSoap: on; AbsSeriesSoap soap.py:148-165: off)
 *        bit1 multiply by nnl (sesoap.py:116-128,248)     bit2 normalise (:249-251)
 * Output p[S][S][D], D=(nmax+1)^2 (lmax+1), flattened [n1][n2][l], with
 *   p[sb][sa][n1][n2][l] = nnl * sum_m w_m c[sa][n1][l][m] c*[sb][n2][l][m]
 * which is exactly where the reference's COO block (ab[0],ab[1]) = (species[b],species[a])
 * lands (sesoap.py:165-171,195-203).
 * ------------------------------------------------------------------------------------ */
typedef struct {
    int lmax, nmax, S, nn, L1, N1, D, shear;
    double rc;
    int flags;
    /* per neighbour */
    double *f;   /* [N1][nn]   f_n = cut * gauss * d^(2n)                 (:176-184) */
    double *df;  /* [N1][nn]   d f_n / d d                                */
    double *Y;   /* [nn][L1*L1] */
    double *dY;  /* [nn][L1*L1][3]  (w.r.t. the scaled, unsheared vector) */
    double *c;   /* [S][N1][L1*L1] packed real/imag like Y               (:188-194) */
    double *nnl; /* [N1][N1][L1] */
    double *p;   /* [S][S][D] un-normalised */
    double norm;
} env_t;

static double fact(int n) { double f = 1; for (int i = 2; i <= n; i++) f *= i; return f; }

static void env_free(env_t *e)
{
    free(e->f); free(e->df); free(e->Y); free(e->dY); free(e->c); free(e->nnl); free(e->p);
}

static void env_forward(env_t *e, int lmax, int nmax, double rc, int flags, int S, int nn,
                        const double *r, const int *s, const double *u, int want_grad)
{
    const int L1 = lmax + 1, N1 = nmax + 1, LL = L1 * L1, D = N1 * N1 * L1;
    e->lmax = lmax; e->nmax = nmax; e->S = S; e->nn = nn; e->L1 = L1; e->N1 = N1; e->D = D;
    e->rc = rc; e->flags = flags;
    e->f = (double *)calloc((size_t)N1 * (nn + 1), sizeof(double));
    e->df = (double *)calloc((size_t)N1 * (nn + 1), sizeof(double));
    e->Y = (double *)calloc((size_t)(nn + 1) * LL, sizeof(double));
    e->dY = (double *)calloc((size_t)(nn + 1) * LL * 3, sizeof(double));
    e->c = (double *)calloc((size_t)S * N1 * LL, sizeof(double));
    e->nnl = (double *)calloc((size_t)N1 * N1 * L1, sizeof(double));
    e->p = (double *)calloc((size_t)S * S * D, sizeof(double));
    double *xs = (double *)calloc((size_t)3 * (nn + 1), sizeof(double));
    for (int j = 0; j < nn; j++)
        for (int a = 0; a < 3; a++) xs[3 * j + a] = r[3 * j + a] / u[j]; /* :172 */
    e->shear = needs_shear(nn, xs);
    for (int j = 0; j < nn; j++) {
        const double *x = xs + 3 * j;
        const double d = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        const double ud = u[j] * d;
        /* cutoff.py:20-48 */
        const double step = ud < rc ? 1.0 : 0.0;
        const double q = 1.0 - ud / rc;
        const double cut = step * q * q;
        const double dcut = step * (-2.0 * q / rc) * u[j]; /* d/dd, :178 */
        const double ex = (flags & 1) ? exp(-0.5 * d * d) : 1.0;
        const double dex = (flags & 1) ? -d * ex : 0.0;
        const double rr = cut * ex, drr = dcut * ex + cut * dex;
        for (int n = 0; n < N1; n++) {
            const double dn = pow(d, 2.0 * n);
            e->f[n * nn + j] = rr * dn;
            e->df[n * nn + j] = drr * dn + (n ? rr * 2.0 * n * pow(d, 2.0 * n - 1.0) : 0.0);
        }
        ylm_one(lmax, x, e->shear, e->Y + (size_t)j * LL, want_grad ? e->dY + (size_t)j * LL * 3 : NULL);
        for (int n = 0; n < N1; n++)
            for (int k = 0; k < LL; k++)
                e->c[((size_t)s[j] * N1 + n) * LL + k] += e->f[n * nn + j] * e->Y[(size_t)j * LL + k];
    }
    free(xs);
    for (int n1 = 0; n1 < N1; n1++)
        for (int n2 = 0; n2 < N1; n2++)
            for (int l = 0; l < L1; l++) {
                const double a1 = 1.0 / ((2 * l + 1) * pow(2.0, 2 * n1 + l) * fact(n1) * fact(n1 + l));
                const double a2 = 1.0 / ((2 * l + 1) * pow(2.0, 2 * n2 + l) * fact(n2) * fact(n2 + l));
                e->nnl[(n1 * N1 + n2) * L1 + l] = (flags & 2) ? sqrt(a1 * a2) : 1.0;
            }
    /* power spectrum, :195-203: weights Yr = 2*tril - eye (row sums), Yi = 2*triu(1) (col sums) */
    double nrm2 = 0.0;
    for (int sb = 0; sb < S; sb++)
        for (int sa = 0; sa < S; sa++)
            for (int n1 = 0; n1 < N1; n1++)
                for (int n2 = 0; n2 < N1; n2++) {
                    const double *ca = e->c + ((size_t)sa * N1 + n1) * LL;
                    const double *cb = e->c + ((size_t)sb * N1 + n2) * LL;
                    for (int l = 0; l < L1; l++) {
                        double acc = 0.0;
                        for (int j = 0; j <= l; j++) acc += (j == l ? 1.0 : 2.0) * ca[l * L1 + j] * cb[l * L1 + j];
                        for (int i = 0; i < l; i++) acc += 2.0 * ca[i * L1 + l] * cb[i * L1 + l];
                        const double v = acc * e->nnl[(n1 * N1 + n2) * L1 + l];
                        e->p[((size_t)sb * S + sa) * D + (n1 * N1 + n2) * L1 + l] = v;
                        nrm2 += v * v;
                    }
                }
    e->norm = sqrt(nrm2);
}

/* Reverse pass: given G = dE/d(p-hat) [S][S][D] (or dE/dp when not normalising), return
 * dE/dr[nn][3].  Exact derivative of env_forward (what torch.autograd gives the reference,
 * calculator/active.py:587-599), including d/dp of p/(|p|+eps). */
static void env_backward(const env_t *e, const int *s, const double *u, const double *r,
                         const double *G, double *dr)
{
    const int S = e->S, N1 = e->N1, L1 = e->L1, LL = L1 * L1, D = e->D, nn = e->nn;
    const size_t SSD = (size_t)S * S * D;
    double *Gp = (double *)malloc(sizeof(double) * SSD);
    if (e->flags & 4) {
        const double sden = e->norm + ORC_EPS;
        double pg = 0.0;
        for (size_t k = 0; k < SSD; k++) pg += e->p[k] * G[k];
        const double coef = e->norm > 0.0 ? pg / (sden * sden * e->norm) : 0.0;
        for (size_t k = 0; k < SSD; k++) Gp[k] = G[k] / sden - e->p[k] * coef;
    } else
        memcpy(Gp, G, sizeof(double) * SSD);
    /* dE/dc */
    double *dc = (double *)calloc((size_t)S * N1 * LL, sizeof(double));
    for (int sb = 0; sb < S; sb++)
        for (int sa = 0; sa < S; sa++)
            for (int n1 = 0; n1 < N1; n1++)
                for (int n2 = 0; n2 < N1; n2++) {
                    const double *ca = e->c + ((size_t)sa * N1 + n1) * LL;
                    const double *cb = e->c + ((size_t)sb * N1 + n2) * LL;
                    double *da = dc + ((size_t)sa * N1 + n1) * LL;
                    double *db = dc + ((size_t)sb * N1 + n2) * LL;
                    for (int l = 0; l < L1; l++) {
                        const double g = Gp[((size_t)sb * S + sa) * D + (n1 * N1 + n2) * L1 + l] *
                                         e->nnl[(n1 * N1 + n2) * L1 + l];
                        for (int j = 0; j <= l; j++) {
                            const double w = (j == l ? 1.0 : 2.0) * g;
                            da[l * L1 + j] += w * cb[l * L1 + j];
                            db[l * L1 + j] += w * ca[l * L1 + j];
                        }
                        for (int i = 0; i < l; i++) {
                            const double w = 2.0 * g;
                            da[i * L1 + l] += w * cb[i * L1 + l];
                            db[i * L1 + l] += w * ca[i * L1 + l];
                        }
                    }
                }
    for (int j = 0; j < nn; j++) {
        const double x[3] = {r[3 * j] / u[j], r[3 * j + 1] / u[j], r[3 * j + 2] / u[j]};
        const double d = sqrt(x[0] * x[0] + x[1] * x[1] + x[2] * x[2]);
        double gx[3] = {0, 0, 0};
        for (int n = 0; n < N1; n++) {
            const double *dcn = dc + ((size_t)s[j] * N1 + n) * LL;
            double dEdf = 0.0;
            for (int k = 0; k < LL; k++) {
                dEdf += dcn[k] * e->Y[(size_t)j * LL + k];
                const double w = dcn[k] * e->f[n * nn + j];
                for (int a = 0; a < 3; a++) gx[a] += w * e->dY[((size_t)j * LL + k) * 3 + a];
            }
            for (int a = 0; a < 3; a++) gx[a] += dEdf * e->df[n * nn + j] * x[a] / d;
        }
        for (int a = 0; a < 3; a++) dr[3 * j + a] = gx[a] / u[j];
    }
    free(dc);
    free(Gp);
}

/* public: descriptor value (and optional vector-Jacobian product) of one environment */
void orc_descriptor(int lmax, int nmax, double rc, int flags, int S, int nn, const double *r,
                    const int *s, const double *u, double *p_out /*[S][S][D]*/,
                    const double *G /*[S][S][D] or NULL*/, double *dr /*[nn][3] or NULL*/)
{
    env_t e;
    env_forward(&e, lmax, nmax, rc, flags, S, nn, r, s, u, G != NULL);
    const size_t SSD = (size_t)S * S * e.D;
    const double sden = (flags & 4) ? e.norm + ORC_EPS : 1.0;
    for (size_t k = 0; k < SSD; k++) p_out[k] = e.p[k] / sden;
    if (G && dr) env_backward(&e, s, u, r, G, dr);
    env_free(&e);
}

/* --------------------------------------------------------------------------------------
 * Neighbour list — stands in for ase.neighborlist as called at descriptor/atoms.py:348-363
 * (radii rc/2, skin 0, bothways, no self interaction; periodic self-images kept).
 * Pair rule: |x_j - x_i + off.cell| < rc.  Brute force over images.  Two-pass: call with
 * j_out == NULL to get counts in ptr[N+1], then again with buffers.
 * Order within an atom: (j ascending, then off lexicographic) — same as the generator.
 * ------------------------------------------------------------------------------------ */
static void cross3(const double *a, const double *b, double *c)
{
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

static double det3(const double *h)
{
    return h[0] * (h[4] * h[8] - h[5] * h[7]) - h[1] * (h[3] * h[8] - h[5] * h[6]) +
           h[2] * (h[3] * h[7] - h[4] * h[6]);
}

/* ase.geometry.complete_cell as ASE's neighbour list applies it: a zero cell vector along an OPEN
 * direction (slab / wire cells such as [a, b, 0] with pbc = TTF) is replaced by a unit vector orthogonal
 * to the others, for the purpose of measuring heights and fractional coordinates only. */
static void complete_cell(const double *cell, const int *pbc, double *h)
{
    int zero[3], nz = 0;
    for (int k = 0; k < 9; k++) h[k] = cell[k];
    for (int k = 0; k < 3; k++) {
        zero[k] = h[3 * k] * h[3 * k] + h[3 * k + 1] * h[3 * k + 1] + h[3 * k + 2] * h[3 * k + 2] < 1e-24;
        nz += zero[k];
    }
    if (nz == 0 || nz == 3) return;
    for (int k = 0; k < 3; k++) {
        if (!zero[k] || pbc[k]) continue;
        const double *p = h + 3 * ((k + 1) % 3), *q = h + 3 * ((k + 2) % 3);
        double v[3];
        if (!zero[(k + 1) % 3] && !zero[(k + 2) % 3]) cross3(p, q, v);
        else {
            const double *w = zero[(k + 1) % 3] ? q : p;
            const int a = fabs(w[0]) <= fabs(w[1]) && fabs(w[0]) <= fabs(w[2]) ? 0 : (fabs(w[1]) <= fabs(w[2]) ? 1 : 2);
            double e[3] = {0, 0, 0};
            e[a] = 1.0;
            cross3(w, e, v);
        }
        const double nv = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
        if (nv > 1e-12) { h[3 * k] = v[0] / nv; h[3 * k + 1] = v[1] / nv; h[3 * k + 2] = v[2] / nv; zero[k] = 0; }
    }
}

int64_t orc_neighbors(int N, const double *pos, const double *cell_in, const int *pbc, double rc,
                      int64_t *ptr, int32_t *j_out, int32_t *off_out)
{
    int nmax[3] = {0, 0, 0};
    double hc[9];
    complete_cell(cell_in, pbc, hc);
    const double *cell = hc;  /* heights and fractional coordinates; image shifts use cell_in below */
    const double V = fabs(det3(cell));
    if (V > 1e-12) {
        /* fractional span of the positions (atoms may sit outside the cell) */
        double inv[9];
        const double *a = cell, *b = cell + 3, *c = cell + 6;
        double bc[3], ca[3], ab[3];
        cross3(b, c, bc); cross3(c, a, ca); cross3(a, b, ab);
        const double dt = det3(cell);
        for (int k = 0; k < 3; k++) { inv[3 * k + 0] = bc[k] / dt; inv[3 * k + 1] = ca[k] / dt; inv[3 * k + 2] = ab[k] / dt; }
        const double hgt[3] = {V / sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]),
                               V / sqrt(ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]),
                               V / sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2])};
        for (int k = 0; k < 3; k++) {
            double lo = 1e300, hi = -1e300;
            for (int i = 0; i < N; i++) {
                const double f = pos[3 * i] * inv[k] + pos[3 * i + 1] * inv[3 + k] + pos[3 * i + 2] * inv[6 + k];
                if (f < lo) lo = f;
                if (f > hi) hi = f;
            }
            nmax[k] = pbc[k] ? (int)ceil(rc / hgt[k] + (hi - lo)) : 0;
        }
    }
    int64_t total = 0;
    ptr[0] = 0;
    for (int i = 0; i < N; i++) {
        int64_t cnt = 0;
        for (int j = 0; j < N; j++)
            for (int o0 = -nmax[0]; o0 <= nmax[0]; o0++)
                for (int o1 = -nmax[1]; o1 <= nmax[1]; o1++)
                    for (int o2 = -nmax[2]; o2 <= nmax[2]; o2++) {
                        if (j == i && !o0 && !o1 && !o2) continue;
                        double d[3];
                        for (int a = 0; a < 3; a++)
                            d[a] = pos[3 * j + a] + (o0 * cell_in[a] + o1 * cell_in[3 + a] + o2 * cell_in[6 + a]) - pos[3 * i + a];
                        const double rr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                        if (rr < rc) {
                            if (j_out) {
                                j_out[total + cnt] = j;
                                off_out[3 * (total + cnt) + 0] = o0;
                                off_out[3 * (total + cnt) + 1] = o1;
                                off_out[3 * (total + cnt) + 2] = o2;
                            }
                            cnt++;
                        }
                    }
        total += cnt;
        ptr[i + 1] = total;
    }
    return total;
}

/* --------------------------------------------------------------------------------------
 * The same list by the linked-cell method, for frames where the brute-force double loop above is too
 * slow (4096 and 16384 atoms): positions are wrapped into the cell along the periodic directions,
 * sorted into cells at least rc wide (perpendicular heights), and every atom visits the cells within
 * range of its own, each under the periodic image it is reached by.  Same pair rule, same output
 * order (j ascending, then off lexicographic, off relative to the UNWRAPPED positions as ASE reports
 * them).  tests/test_oracle_golden.py holds it equal to orc_neighbors on every golden frame and on
 * random cells; OpenMP over atoms.
 * ------------------------------------------------------------------------------------ */
typedef struct { int32_t j, o0, o1, o2; } nbr_t;

static int nbr_cmp(const void *pa, const void *pb)
{
    const nbr_t *a = (const nbr_t *)pa, *b = (const nbr_t *)pb;
    if (a->j != b->j) return a->j < b->j ? -1 : 1;
    if (a->o0 != b->o0) return a->o0 < b->o0 ? -1 : 1;
    if (a->o1 != b->o1) return a->o1 < b->o1 ? -1 : 1;
    if (a->o2 != b->o2) return a->o2 < b->o2 ? -1 : 1;
    return 0;
}

static int floor_div(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

int64_t orc_neighbors_cells(int N, const double *pos, const double *cell_in, const int *pbc, double rc,
                            int64_t *ptr, int32_t *j_out, int32_t *off_out)
{
    if (N <= 0) { ptr[0] = 0; return 0; }
    double hc[9];
    complete_cell(cell_in, pbc, hc);
    const double *cell = hc;  /* heights and fractional coordinates; image shifts use cell_in below */
    const double V = fabs(det3(cell));
    double inv[9] = {0};
    double hgt[3] = {0, 0, 0};
    const int usable = V > 1e-12;
    if (usable) {
        const double *a = cell, *b = cell + 3, *c = cell + 6;
        double bc[3], ca[3], ab[3];
        cross3(b, c, bc); cross3(c, a, ca); cross3(a, b, ab);
        const double dt = det3(cell);
        for (int k = 0; k < 3; k++) { inv[3 * k + 0] = bc[k] / dt; inv[3 * k + 1] = ca[k] / dt; inv[3 * k + 2] = ab[k] / dt; }
        hgt[0] = V / sqrt(bc[0] * bc[0] + bc[1] * bc[1] + bc[2] * bc[2]);
        hgt[1] = V / sqrt(ca[0] * ca[0] + ca[1] * ca[1] + ca[2] * ca[2]);
        hgt[2] = V / sqrt(ab[0] * ab[0] + ab[1] * ab[1] + ab[2] * ab[2]);
    }
    int nc[3] = {1, 1, 1}, rg[3] = {0, 0, 0};
    for (int k = 0; k < 3; k++)
        if (usable && pbc[k]) {
            nc[k] = (int)floor(hgt[k] / rc);
            if (nc[k] < 1) nc[k] = 1;
            if (nc[k] > 64) nc[k] = 64;
            rg[k] = (int)ceil(rc * nc[k] / hgt[k]);
        }
    const int ncell = nc[0] * nc[1] * nc[2];
    int *wrap = (int *)calloc((size_t)3 * N, sizeof(int));
    int *cof = (int *)malloc(sizeof(int) * (size_t)N);
    int *head = (int *)malloc(sizeof(int) * (size_t)ncell);
    int *next = (int *)malloc(sizeof(int) * (size_t)N);
    for (int c = 0; c < ncell; c++) head[c] = -1;
    for (int i = 0; i < N; i++) {
        int ci[3] = {0, 0, 0};
        for (int k = 0; k < 3; k++)
            if (usable && pbc[k]) {
                double f = pos[3 * i] * inv[k] + pos[3 * i + 1] * inv[3 + k] + pos[3 * i + 2] * inv[6 + k];
                const double fl = floor(f);
                wrap[3 * i + k] = (int)fl;
                f -= fl;
                int b = (int)(f * nc[k]);
                if (b >= nc[k]) b = nc[k] - 1;
                if (b < 0) b = 0;
                ci[k] = b;
            }
        cof[i] = (ci[0] * nc[1] + ci[1]) * nc[2] + ci[2];
    }
    for (int i = N - 1; i >= 0; i--) { next[i] = head[cof[i]]; head[cof[i]] = i; }
    /* pass 1: counts; pass 2: sorted entries */
    for (int pass = 0; pass < (j_out ? 2 : 1); pass++) {
#pragma omp parallel
        {
            int cap = 256, n = 0;
            nbr_t *buf = (nbr_t *)malloc(sizeof(nbr_t) * (size_t)cap);
#pragma omp for schedule(dynamic, 16)
            for (int i = 0; i < N; i++) {
                n = 0;
                const int c2 = cof[i] % nc[2], c1 = (cof[i] / nc[2]) % nc[1], c0 = cof[i] / (nc[1] * nc[2]);
                for (int o0 = -rg[0]; o0 <= rg[0]; o0++)
                    for (int o1 = -rg[1]; o1 <= rg[1]; o1++)
                        for (int o2 = -rg[2]; o2 <= rg[2]; o2++) {
                            const int t0 = c0 + o0, t1 = c1 + o1, t2 = c2 + o2;
                            const int m0 = floor_div(t0, nc[0]), m1 = floor_div(t1, nc[1]), m2 = floor_div(t2, nc[2]);
                            const int cc = ((t0 - m0 * nc[0]) * nc[1] + (t1 - m1 * nc[1])) * nc[2] + (t2 - m2 * nc[2]);
                            for (int j = head[cc]; j >= 0; j = next[j]) {
                                const int s0 = m0 - wrap[3 * j] + wrap[3 * i], s1 = m1 - wrap[3 * j + 1] + wrap[3 * i + 1],
                                          s2 = m2 - wrap[3 * j + 2] + wrap[3 * i + 2];
                                if (j == i && !s0 && !s1 && !s2) continue;
                                double d[3];
                                for (int a = 0; a < 3; a++)
                                    d[a] = pos[3 * j + a] + (s0 * cell_in[a] + s1 * cell_in[3 + a] + s2 * cell_in[6 + a]) - pos[3 * i + a];
                                const double rr = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
                                if (rr < rc) {
                                    if (n == cap) { cap *= 2; buf = (nbr_t *)realloc(buf, sizeof(nbr_t) * (size_t)cap); }
                                    buf[n].j = j; buf[n].o0 = s0; buf[n].o1 = s1; buf[n].o2 = s2;
                                    n++;
                                }
                            }
                        }
                if (pass == 0) ptr[i + 1] = n;
                else {
                    qsort(buf, (size_t)n, sizeof(nbr_t), nbr_cmp);
                    for (int t = 0; t < n; t++) {
                        const int64_t e = ptr[i] + t;
                        j_out[e] = buf[t].j;
                        off_out[3 * e] = buf[t].o0; off_out[3 * e + 1] = buf[t].o1; off_out[3 * e + 2] = buf[t].o2;
                    }
                }
            }
            free(buf);
        }
        if (pass == 0) {
            ptr[0] = 0;
            for (int i = 0; i < N; i++) ptr[i + 1] += ptr[i];
        }
    }
    free(wrap); free(cof); free(head); free(next);
    return ptr[N];
}

/* --------------------------------------------------------------------------------------
 * Kernel entry — similarity/universal.py:109-122 + similarity/similarity.py:41-43,94-103:
 *   k(i,q) = [Z_i == Z_q] (p_i . p_q)^eta  (0 if either has no neighbours)
 *          + 1 if both have no neighbours and Z_i == Z_q.
 * ------------------------------------------------------------------------------------ */
static double ipow_or_pow(double x, double eta)
{
    if (eta == floor(eta) && eta >= 0 && eta <= 64) {
        double y = 1.0;
        for (int k = 0; k < (int)eta; k++) y *= x;
        return y;
    }
    return pow(x, eta);
}

void orc_kernel_matrix(int n1, const int *z1, const int *nn1, const double *P1, int n2, const int *z2,
                       const int *nn2, const double *P2, int SSD, double eta, double *K /*[n1][n2]*/)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n1; i++)
        for (int q = 0; q < n2; q++) {
            double k = 0.0;
            if (z1[i] == z2[q]) {
                if (nn1[i] > 0 && nn2[q] > 0) {
                    double dot = 0.0;
                    for (int t = 0; t < SSD; t++) dot += P1[(size_t)i * SSD + t] * P2[(size_t)q * SSD + t];
                    k = ipow_or_pow(dot, eta);
                } else if (nn1[i] == 0 && nn2[q] == 0)
                    k = 1.0;
            }
            K[(size_t)i * n2 + q] = k;
        }
}

/* --------------------------------------------------------------------------------------
 * One frame through the predict path — calculator/active.py:425-502, :548-611, :781-804:
 *   descriptors (atoms.py:365-382 + universal.py:100-107) -> K_nm (gppotential.py:63-84)
 *   -> E = sum(K mu) (+ mean, gppotential.py:219-227) -> F = -dE/dx, dE/dcell (reverse pass)
 *   -> stress (active.py:604-610) -> beta (active.py:781-792).
 * Inputs: neighbour list in CSR form (ptr,j,off); species table zs[S] with radii; inducing
 * descriptors Pm[m][S*S*D] with central numbers zm[m] and neighbour counts nnm[m].
 * choli may be NULL (beta skipped).  All outputs caller-allocated.
 * ------------------------------------------------------------------------------------ */
static int slot_of(int S, const int *zs, int z)
{
    for (int k = 0; k < S; k++)
        if (zs[k] == z) return k;
    return -1;
}

int orc_frame(int lmax, int nmax, double rc, double eta, int S, const int *zs, const double *radii,
              int N, const int *numbers, const double *pos, const double *cell, const int64_t *ptr,
              const int32_t *nj, const int32_t *noff, int m, const int *zm, const int *nnm,
              const double *Pm, const double *mu, const double *choli, double *P /*[N][SSD] or NULL*/,
              double *K /*[N][m]*/, double *E_out, double *F /*[N][3]*/, double *dcell /*[3][3]*/,
              double *stress /*[6]*/, double *beta /*[N] or NULL*/)
{
    const int L1 = lmax + 1, N1 = nmax + 1, D = N1 * N1 * L1;
    const int SSD = S * S * D;
    double E = 0.0;
    double *gx = (double *)calloc((size_t)3 * N, sizeof(double));
    double dc[9] = {0};
    int bad = 0;
#pragma omp parallel
    {
        double *gx_t = (double *)calloc((size_t)3 * N, sizeof(double));
        double dc_t[9] = {0};
        double E_t = 0.0;
        double *G = (double *)malloc(sizeof(double) * SSD);
        double *ph = (double *)malloc(sizeof(double) * SSD);
#pragma omp for schedule(dynamic, 4)
        for (int i = 0; i < N; i++) {
            const int nn = (int)(ptr[i + 1] - ptr[i]);
            double *r = (double *)malloc(sizeof(double) * 3 * (nn + 1));
            double *u = (double *)malloc(sizeof(double) * (nn + 1));
            int *s = (int *)malloc(sizeof(int) * (nn + 1));
            double *dr = (double *)calloc((size_t)3 * (nn + 1), sizeof(double));
            for (int t = 0; t < nn; t++) {
                const int64_t e = ptr[i] + t;
                const int j = nj[e];
                for (int a = 0; a < 3; a++)
                    r[3 * t + a] = pos[3 * j + a] - pos[3 * i + a] +
                                   (noff[3 * e] * cell[a] + noff[3 * e + 1] * cell[3 + a] + noff[3 * e + 2] * cell[6 + a]);
                s[t] = slot_of(S, zs, numbers[j]);
                if (s[t] < 0) { bad = 1; s[t] = 0; }
                u[t] = radii[s[t]];
            }
            env_t e;
            env_forward(&e, lmax, nmax, rc, 7, S, nn, r, s, u, 1);
            const double sden = e.norm + ORC_EPS;
            for (int k = 0; k < SSD; k++) ph[k] = nn > 0 ? e.p[k] / sden : 0.0;
            if (P) memcpy(P + (size_t)i * SSD, ph, sizeof(double) * SSD);
            memset(G, 0, sizeof(double) * SSD);
            for (int q = 0; q < m; q++) {
                double k = 0.0;
                if (numbers[i] == zm[q]) {
                    if (nn > 0 && nnm[q] > 0) {
                        double dot = 0.0;
                        const double *pq = Pm + (size_t)q * SSD;
                        for (int t = 0; t < SSD; t++) dot += ph[t] * pq[t];
                        k = ipow_or_pow(dot, eta);
                        /* d k / d p-hat = eta dot^(eta-1) p_q */
                        const double w = mu[q] * eta * ipow_or_pow(dot, eta - 1.0);
                        for (int t = 0; t < SSD; t++) G[t] += w * pq[t];
                    } else if (nn == 0 && nnm[q] == 0)
                        k = 1.0;
                }
                K[(size_t)i * m + q] = k;
                E_t += k * mu[q];
            }
            if (nn > 0) {
                env_backward(&e, s, u, r, G, dr);
                for (int t = 0; t < nn; t++) {
                    const int64_t ee = ptr[i] + t;
                    const int j = nj[ee];
                    for (int a = 0; a < 3; a++) {
                        gx_t[3 * j + a] += dr[3 * t + a];
                        gx_t[3 * i + a] -= dr[3 * t + a];
                        for (int k = 0; k < 3; k++) dc_t[3 * k + a] += noff[3 * ee + k] * dr[3 * t + a];
                    }
                }
            }
            env_free(&e);
            free(r); free(u); free(s); free(dr);
        }
#pragma omp critical
        {
            E += E_t;
            for (int k = 0; k < 3 * N; k++) gx[k] += gx_t[k];
            for (int k = 0; k < 9; k++) dc[k] += dc_t[k];
        }
        free(gx_t); free(G); free(ph);
    }
    for (int k = 0; k < 3 * N; k++) F[k] = -gx[k];
    for (int k = 0; k < 9; k++) dcell[k] = dc[k];
    *E_out = E;
    /* active.py:604-610 */
    double st[9];
    for (int a = 0; a < 3; a++)
        for (int b = 0; b < 3; b++) {
            double s1 = 0.0, s2 = 0.0;
            for (int i = 0; i < N; i++) s1 -= F[3 * i + b] * pos[3 * i + a];
            for (int k = 0; k < 3; k++) s2 += dc[3 * k + b] * cell[3 * k + a];
            st[3 * a + b] = s1 + s2;
        }
    double vol = fabs(det3(cell));
    if (!(vol > 0.0)) vol = -2.0;
    const int voigt[6] = {0, 4, 8, 5, 2, 1};
    for (int k = 0; k < 6; k++) stress[k] = st[voigt[k]] / vol;
    /* active.py:781-792 */
    if (beta && choli) {
#pragma omp parallel for schedule(static)
        for (int i = 0; i < N; i++) {
            double c = 0.0;
            for (int a = 0; a < m; a++) {
                double b = 0.0;
                for (int q = 0; q < m; q++) b += choli[(size_t)a * m + q] * K[(size_t)i * m + q];
                c += b * b;
            }
            const double v = 1.0 - c;
            beta[i] = sqrt(v > 0.0 ? v : 0.0);
        }
    }
    free(gx);
    return bad ? -1 : 0;
}

/* --------------------------------------------------------------------------------------
 * Linear algebra — regression/algebra.py:29-47 (jitcholesky) and
 * regression/gppotential.py:1232-1263 (choli = L^-1, mu by QR least squares).
 * ------------------------------------------------------------------------------------ */
/* plain lower Cholesky; returns 0 on success, k+1 if the k-th pivot is not positive
 * (LAPACK potrf semantics, which is what torch.linalg.cholesky raises on). */
static int chol_lower(int n, const double *A, double ridge, double *L)
{
    memset(L, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) {
        double d = A[(size_t)j * n + j] + ridge;
        for (int k = 0; k < j; k++) d -= L[(size_t)j * n + k] * L[(size_t)j * n + k];
        if (!(d > 0.0)) return j + 1;
        const double ljj = sqrt(d);
        L[(size_t)j * n + j] = ljj;
        for (int i = j + 1; i < n; i++) {
            double v = A[(size_t)i * n + j];
            for (int k = 0; k < j; k++) v -= L[(size_t)i * n + k] * L[(size_t)j * n + k];
            L[(size_t)i * n + j] = v / ljj;
        }
    }
    return 0;
}

/* algebra.py:29-47.  Returns 0 ok, -1 "cholesky was not successful!". */
int orc_jitcholesky(int n, const double *A, double *L, double *ridge_out)
{
    double ridge = 0.0;
    if (chol_lower(n, A, 0.0, L) == 0) { *ridge_out = 0.0; return 0; }
    double scale = 0.0;
    for (int i = 0; i < n; i++) scale += A[(size_t)i * n + i];
    scale /= n;
    if (scale == 0.0) scale = ORC_EPS;
    ridge = 1e-6 * scale;
    for (;;) {
        int done = chol_lower(n, A, ridge, L) == 0;
        if (!done) ridge *= 2.0;
        if (ridge > scale) { *ridge_out = ridge; return -1; }
        if (done) break;
    }
    *ridge_out = ridge;
    return 0;
}

/* inverse of a lower-triangular matrix (gppotential.py:1234: choli = L.inverse()) */
void orc_tril_inverse(int n, const double *L, double *Li)
{
    memset(Li, 0, sizeof(double) * n * n);
    for (int j = 0; j < n; j++) {
        Li[(size_t)j * n + j] = 1.0 / L[(size_t)j * n + j];
        for (int i = j + 1; i < n; i++) {
            double v = 0.0;
            for (int k = j; k < i; k++) v -= L[(size_t)i * n + k] * Li[(size_t)k * n + j];
            Li[(size_t)i * n + j] = v / L[(size_t)i * n + i];
        }
    }
}

/* min |A x - y|_2 by Householder QR, A[rows][cols] row-major, rows >= cols
 * (gppotential.py:1261-1263: Q,R = qr(A); mu = R^-1 Q^T y). */
int orc_lstsq_qr(int rows, int cols, const double *A_in, const double *y_in, double *x)
{
    double *A = (double *)malloc(sizeof(double) * rows * cols);
    double *y = (double *)malloc(sizeof(double) * rows);
    memcpy(A, A_in, sizeof(double) * rows * cols);
    memcpy(y, y_in, sizeof(double) * rows);
    for (int k = 0; k < cols; k++) {
        double nrm = 0.0;
        for (int i = k; i < rows; i++) nrm += A[(size_t)i * cols + k] * A[(size_t)i * cols + k];
        nrm = sqrt(nrm);
        if (nrm == 0.0) { free(A); free(y); return -1; }
        const double alpha = A[(size_t)k * cols + k] > 0 ? -nrm : nrm;
        A[(size_t)k * cols + k] -= alpha; /* v = x - alpha e1, stored in place */
        double vv = 0.0;
        for (int i = k; i < rows; i++) vv += A[(size_t)i * cols + k] * A[(size_t)i * cols + k];
        for (int j = k + 1; j < cols; j++) {
            double dot = 0.0;
            for (int i = k; i < rows; i++) dot += A[(size_t)i * cols + k] * A[(size_t)i * cols + j];
            const double t = 2.0 * dot / vv;
            for (int i = k; i < rows; i++) A[(size_t)i * cols + j] -= t * A[(size_t)i * cols + k];
        }
        double dot = 0.0;
        for (int i = k; i < rows; i++) dot += A[(size_t)i * cols + k] * y[i];
        const double t = 2.0 * dot / vv;
        for (int i = k; i < rows; i++) y[i] -= t * A[(size_t)i * cols + k];
        A[(size_t)k * cols + k] = alpha; /* R_kk */
    }
    for (int k = cols - 1; k >= 0; k--) {
        double v = y[k];
        for (int j = k + 1; j < cols; j++) v -= A[(size_t)k * cols + j] * x[j];
        x[k] = v / A[(size_t)k * cols + k];
    }
    free(A);
    free(y);
    return 0;
}

/* gppotential.py:1204-1339 with optimize=False, same_sigma=True:
 *   L,ridge = jitcholesky(M); choli = L^-1; sigma = noise0 * 0.99 * mean(diag M)
 *   (to_0_1(to_inf_inf(noise0)) == noise0, :1219-1222,:1245-1247)
 *   mu = lstsq([Ke;Kf;Kv; sigma L^T], [E - mean; F; V*stress; 0])   (:1255-1263,:1337-1338)
 * K = [Ke;Kf;Kv] stacked [rows][m]; Y = matching targets [rows]. */
int orc_regression(int m, const double *M, int rows, const double *K, const double *Y, double noise0,
                   double *mu, double *choli, double *L, double *ridge, double *sigma_out)
{
    const int rc = orc_jitcholesky(m, M, L, ridge);
    if (rc) return rc;
    orc_tril_inverse(m, L, choli);
    double dm = 0.0;
    for (int i = 0; i < m; i++) dm += M[(size_t)i * m + i];
    dm /= m;
    const double sigma = noise0 * (dm * 0.99);
    *sigma_out = sigma;
    double *A = (double *)malloc(sizeof(double) * (size_t)(rows + m) * m);
    double *y = (double *)calloc((size_t)rows + m, sizeof(double));
    memcpy(A, K, sizeof(double) * (size_t)rows * m);
    memcpy(y, Y, sizeof(double) * rows);
    for (int i = 0; i < m; i++)
        for (int j = 0; j < m; j++) A[(size_t)(rows + i) * m + j] = sigma * L[(size_t)j * m + i];
    const int r2 = orc_lstsq_qr(rows + m, m, A, y, mu);
    free(A);
    free(y);
    return r2;
}

/* gppotential.py:644-649: vscale[z] = mean over inducing q of species z of mu_q (M mu)_q */
void orc_vscale(int m, const double *M, const double *mu, const int *zm, int S, const int *zs, double *vscale)
{
    for (int k = 0; k < S; k++) {
        double acc = 0.0;
        int cnt = 0;
        for (int q = 0; q < m; q++) {
            if (zm[q] != zs[k]) continue;
            double v = 0.0;
            for (int t = 0; t < m; t++) v += M[(size_t)q * m + t] * mu[t];
            acc += mu[q] * v;
            cnt++;
        }
        vscale[k] = cnt ? acc / cnt : INFINITY;
    }
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* Tests that compare decision sequences need run-to-run identical sums: with one thread the
 * dynamic schedule + critical-section merge of orc_frame is a fixed summation order. */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
    omp_set_num_threads(n > 0 ? n : 1);
#else
    (void)n;
#endif
}
