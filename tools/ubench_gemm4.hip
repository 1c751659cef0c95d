// Three LDS buffers, one barrier per stage at the TOP of the stage, first fragments of the next stage read before
// the barrier (they were made visible one barrier earlier): does the MFMA stream run through the stage boundary?
// Compared with the two-buffer loop of gemm.hip (V = 0).  TM = 1: 32x64 tiles, TM = 2: 64x64 tiles.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double v4d __attribute__((ext_vector_type(4)));
#define KS 32
#define LD 34
template <int V, int TM>
__global__ __launch_bounds__(256) void k(const double *A, const double *B, double *C, int K, int lda, int ldb, int row_tiles)
{
    constexpr int BMT = 32 * TM, NB = V == 0 ? 2 : 3;
    __shared__ double As[NB][BMT * LD];
    __shared__ double Bs[NB][64 * LD];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int rt = blockIdx.x % row_tiles, ct = blockIdx.x / row_tiles;
    const int row0 = rt * BMT, col0 = ct * 64, kbeg = 0, kend = K;
    v4d acc[TM][2] = {};
    const int lr = tid >> 2, lk = (tid & 3) * 8;
    const double *Ag = A + (size_t)(row0 + (TM == 2 ? lr : (lr & 31))) * lda + lk;
    const double *Bg = B + (size_t)(col0 + lr) * ldb + lk;
    double2 pa0, pa1, pa2, pa3, pb0, pb1, pb2, pb3, qa0, qa1, qa2, qa3, qb0, qb1, qb2, qb3;
    pa0 = pa1 = pa2 = pa3 = pb0 = pb1 = pb2 = pb3 = make_double2(1.0, 2.0);
    qa0 = qa1 = qa2 = qa3 = qb0 = qb1 = qb2 = qb3 = make_double2(1.0, 2.0);
#define GLOAD(S, K0)                                                                             \
    if ((K0) < kend) {                                                                           \
        if (TM == 2 || lr < BMT) {                                                               \
        S##a0 = *(const double2 *)(Ag + (K0)); S##a1 = *(const double2 *)(Ag + (K0) + 2);        \
        S##a2 = *(const double2 *)(Ag + (K0) + 4); S##a3 = *(const double2 *)(Ag + (K0) + 6);    \
        }                                                                                        \
        S##b0 = *(const double2 *)(Bg + (K0)); S##b1 = *(const double2 *)(Bg + (K0) + 2);        \
        S##b2 = *(const double2 *)(Bg + (K0) + 4); S##b3 = *(const double2 *)(Bg + (K0) + 6);    \
    }
#define LSTORE(S, BUF)                                                                           \
    {                                                                                            \
        double *da = &As[BUF][lr * LD + lk], *db = &Bs[BUF][lr * LD + lk];                       \
        if (TM == 2 || lr < BMT) {                                                               \
        *(double2 *)(da) = S##a0; *(double2 *)(da + 2) = S##a1;                                  \
        *(double2 *)(da + 4) = S##a2; *(double2 *)(da + 6) = S##a3;                              \
        }                                                                                        \
        *(double2 *)(db) = S##b0; *(double2 *)(db + 2) = S##b1;                                  \
        *(double2 *)(db + 4) = S##b2; *(double2 *)(db + 6) = S##b3;                              \
    }
    const int fa = (wr * 16 * TM + (lane & 15)) * LD + (lane >> 4);
    const int fb = (wc * 32 + (lane & 15)) * LD + (lane >> 4);
    if (V == 0) {
        auto compute = [&](int buf) {
#pragma unroll
            for (int kk = 0; kk < KS; kk += 4) {
                const double b0 = Bs[buf][fb + kk], b1 = Bs[buf][fb + 16 * LD + kk];
#pragma unroll
                for (int tm = 0; tm < TM; tm++) {
                    const double a0 = As[buf][fa + tm * 16 * LD + kk];
                    acc[tm][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[tm][0], 0, 0, 0);
                    acc[tm][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[tm][1], 0, 0, 0);
                }
            }
        };
        GLOAD(p, kbeg); GLOAD(q, kbeg + KS); LSTORE(p, 0); GLOAD(p, kbeg + 2 * KS);
        __syncthreads();
        for (int k0 = kbeg; k0 < kend; k0 += 2 * KS) {
            if (k0 + KS < kend) LSTORE(q, 1);
            GLOAD(q, k0 + 3 * KS);
            compute(0);
            __syncthreads();
            if (k0 + KS >= kend) break;
            if (k0 + 2 * KS < kend) LSTORE(p, 0);
            GLOAD(p, k0 + 4 * KS);
            compute(1);
            __syncthreads();
        }
    } else {
        // stage s lives in buffer s % 3.  Iteration s: barrier (everyone is done reading stage s-1, and the stores of
        // stage s+1 issued during iteration s-1 are visible), store stage s+2 into the buffer stage s-1 leaves,
        // request stage s+4, compute stage s with the first fragments already in registers, then read the first
        // fragments of stage s+1 (visible since this iteration's barrier).
        double fb0, fb1, fa0[TM];
        auto first = [&](int buf) {
            fb0 = Bs[buf][fb]; fb1 = Bs[buf][fb + 16 * LD];
#pragma unroll
            for (int tm = 0; tm < TM; tm++) fa0[tm] = As[buf][fa + tm * 16 * LD];
        };
        auto compute = [&](int buf, int nbuf) {
#pragma unroll
            for (int kk = 0; kk < KS; kk += 4) {
                double b0, b1, a0[TM];
                if (kk == 0) { b0 = fb0; b1 = fb1; }
                else { b0 = Bs[buf][fb + kk]; b1 = Bs[buf][fb + 16 * LD + kk]; }
#pragma unroll
                for (int tm = 0; tm < TM; tm++) a0[tm] = kk == 0 ? fa0[tm] : As[buf][fa + tm * 16 * LD + kk];
                if (kk == KS - 8) first(nbuf);
#pragma unroll
                for (int tm = 0; tm < TM; tm++) {
                    acc[tm][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[tm], b0, acc[tm][0], 0, 0, 0);
                    acc[tm][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0[tm], b1, acc[tm][1], 0, 0, 0);
                }
            }
        };
        GLOAD(p, kbeg); GLOAD(q, kbeg + KS);
        LSTORE(p, 0); GLOAD(p, kbeg + 2 * KS);
        LSTORE(q, 1); GLOAD(q, kbeg + 3 * KS);
        __syncthreads();
        first(0);
        // unrolled by six = lcm(2 register stages, 3 buffers), written as a loop over s with static selection
        int s = 0;
        for (int k0 = kbeg; k0 < kend;) {
#define STEP(S, BUF, NBUF, SBUF)                                                                 \
            __syncthreads();                                                                     \
            if (k0 + 2 * KS < kend) LSTORE(S, SBUF);                                             \
            GLOAD(S, k0 + 4 * KS);                                                               \
            compute(BUF, NBUF);                                                                  \
            k0 += KS; if (k0 >= kend) break;
            STEP(p, 0, 1, 2) STEP(q, 1, 2, 0) STEP(p, 2, 0, 1) STEP(q, 0, 1, 2) STEP(p, 1, 2, 0) STEP(q, 2, 0, 1)
        }
        (void)s;
    }
    double sum = 0;
    for (int i = 0; i < TM; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 4; r++) sum += acc[i][j][r];
    C[(size_t)blockIdx.x * 256 + tid] = sum;
}

// V2: three LDS buffers addressed dynamically (period 2 = the register stages), branch-free steady state so that the
// compiler can count vmcnt, stores and loads placed by hand in the shadow of the MFMAs (sched_barrier keeps them there).
template <int TM>
__global__ __launch_bounds__(256, 2) void k2(const double *A, const double *B, double *C, int K, int lda, int ldb, int row_tiles)
{
    constexpr int BMT = 32 * TM, ASZ = BMT * LD, BSZ = 64 * LD, NA = 2 * TM;
    __shared__ double As[3 * ASZ];
    __shared__ double Bs[3 * BSZ];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int wr = wave >> 1, wc = wave & 1;
    const int rt = blockIdx.x % row_tiles, ct = blockIdx.x / row_tiles;
    const int row0 = rt * BMT, col0 = ct * 64, kbeg = 0, kend = K;
    v4d acc[TM][2] = {};
    // global side: B (64 rows) 4 threads per row, 64 B each; A the same for 64 rows, 8 threads per row (32 B each) for 32
    const int lrb = tid >> 2, lkb = (tid & 3) * 8;
    const int lra = TM == 2 ? (tid >> 2) : (tid >> 3), lka = TM == 2 ? (tid & 3) * 8 : (tid & 7) * 4;
    const double *Ag = A + (size_t)(row0 + lra) * lda + lka + kbeg;
    const double *Bg = B + (size_t)(col0 + lrb) * ldb + lkb + kbeg;
    const int la = lra * LD + lka, lb = lrb * LD + lkb;
    double2 pa[NA], pb[4], qa[NA], qb[4];
    const int fa = (wr * 16 * TM + (lane & 15)) * LD + (lane >> 4);
    const int fb = (wc * 32 + (lane & 15)) * LD + (lane >> 4);
    double fra[2][TM], frb[2][2];
    const int nst = (kend - kbeg + KS - 1) / KS;
#define GLA(S, ST, I0, I1) _Pragma("unroll") for (int i = I0; i < I1; i++) S##a[i] = *(const double2 *)(Ag + (ST) * KS + 2 * i);
#define GLB(S, ST, I0, I1) _Pragma("unroll") for (int i = I0; i < I1; i++) S##b[i] = *(const double2 *)(Bg + (ST) * KS + 2 * i);
#define LSA(S, BUF, I0, I1) _Pragma("unroll") for (int i = I0; i < I1; i++) *(double2 *)(As + (BUF) * ASZ + la + 2 * i) = S##a[i];
#define LSB(S, BUF, I0, I1) _Pragma("unroll") for (int i = I0; i < I1; i++) *(double2 *)(Bs + (BUF) * BSZ + lb + 2 * i) = S##b[i];
#define RDF(SL, BUF, KK)                                                                         \
    {                                                                                            \
        frb[SL][0] = Bs[(BUF) * BSZ + fb + (KK)]; frb[SL][1] = Bs[(BUF) * BSZ + fb + 16 * LD + (KK)]; \
        _Pragma("unroll") for (int tm = 0; tm < TM; tm++) fra[SL][tm] = As[(BUF) * ASZ + fa + tm * 16 * LD + (KK)]; \
    }
#define STAGE(R, DO_ST, DO_LD, SLD)                                                              \
    {                                                                                            \
        __syncthreads();                                                                         \
        _Pragma("unroll") for (int j = 0; j < 8; j++) {                                          \
            const int sl = j & 1;                                                                \
            if (j < 7) RDF(sl ^ 1, cur, 4 * (j + 1)) else RDF(sl ^ 1, nxt, 0)                    \
            _Pragma("unroll") for (int tm = 0; tm < TM; tm++) {                                  \
                acc[tm][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(fra[sl][tm], frb[sl][0], acc[tm][0], 0, 0, 0); \
                acc[tm][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(fra[sl][tm], frb[sl][1], acc[tm][1], 0, 0, 0); \
            }                                                                                    \
            if (TM == 1) {                                                                       \
                if (DO_ST) { if (j == 0) LSA(R, stb, 0, 2) if (j == 1) LSB(R, stb, 0, 2) if (j == 2) LSB(R, stb, 2, 4) } \
                if (DO_LD) { if (j == 3) GLA(R, SLD, 0, 2) if (j == 4) GLB(R, SLD, 0, 2) if (j == 5) GLB(R, SLD, 2, 4) } \
            } else {                                                                             \
                if (DO_ST) { if (j == 0) LSA(R, stb, 0, 2) if (j == 1) LSA(R, stb, 2, 4) if (j == 2) LSB(R, stb, 0, 2) if (j == 3) LSB(R, stb, 2, 4) } \
                if (DO_LD) { if (j == 4) GLA(R, SLD, 0, 4) if (j == 5) GLB(R, SLD, 0, 4) }        \
            }                                                                                    \
            __builtin_amdgcn_sched_barrier(0);                                                   \
        }                                                                                        \
        const int t_ = cur; cur = nxt; nxt = stb; stb = t_;                                      \
    }
    for (int i = 0; i < NA; i++) pa[i] = qa[i] = make_double2(0.0, 0.0);
    for (int i = 0; i < 4; i++) pb[i] = qb[i] = make_double2(0.0, 0.0);
    int cur = 0, nxt = 1, stb = 2;
    // prologue: stages 0, 1 -> LDS buffers 0, 1; stages 2, 3 in flight in p, q
    int s = 0;
    if (nst >= 6) {
        // the steady state is entered from a prologue without conditions, so that the wait counts of the loop
        // are exact (a conditional load before the loop makes the compiler wait for vmcnt(0) inside it)
        GLA(p, 0, 0, NA) GLB(p, 0, 0, 4)
        GLA(q, 1, 0, NA) GLB(q, 1, 0, 4)
        LSA(p, 0, 0, NA) LSB(p, 0, 0, 4)
        GLA(p, 2, 0, NA) GLB(p, 2, 0, 4)
        LSA(q, 1, 0, NA) LSB(q, 1, 0, 4)
        GLA(q, 3, 0, NA) GLB(q, 3, 0, 4)
        __syncthreads();
        RDF(0, 0, 0)
        for (; s + 5 < nst; s += 2) {  // both stages of the pair store (s+2, s+3) and load (s+4, s+5)
            STAGE(p, 1, 1, s + 4)
            STAGE(q, 1, 1, s + 5)
        }
    } else {
        GLA(p, 0, 0, NA) GLB(p, 0, 0, 4)
        if (nst > 1) { GLA(q, 1, 0, NA) GLB(q, 1, 0, 4) }
        LSA(p, 0, 0, NA) LSB(p, 0, 0, 4)
        if (nst > 2) { GLA(p, 2, 0, NA) GLB(p, 2, 0, 4) }
        LSA(q, 1, 0, NA) LSB(q, 1, 0, 4)
        if (nst > 3) { GLA(q, 3, 0, NA) GLB(q, 3, 0, 4) }
        __syncthreads();
        RDF(0, 0, 0)
    }
    // at most five stages left; one of them may still load
    for (; s < nst; s += 2) {
        if (s + 4 < nst) STAGE(p, 1, 1, s + 4)
        else if (s + 2 < nst) STAGE(p, 1, 0, 0)
        else STAGE(p, 0, 0, 0)
        if (s + 1 >= nst) break;
        if (s + 3 < nst) STAGE(q, 1, 0, 0)
        else STAGE(q, 0, 0, 0)
    }
    double sum = 0;
    for (int i = 0; i < TM; i++) for (int j = 0; j < 2; j++) for (int r = 0; r < 4; r++) sum += acc[i][j][r];
    C[(size_t)blockIdx.x * 256 + tid] = sum;
}

template <int TM> float run2(const double *A, const double *B, double *C, int K, int rt, int ct)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
        hipEventRecord(e0);
        k2<TM><<<rt * ct, 256>>>(A, B, C, K, K, K, rt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3;
}
template <int V, int TM> float run(const double *A, const double *B, double *C, int K, int rt, int ct)
{
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 20; rep++) {
        hipEventRecord(e0);
        k<V, TM><<<rt * ct, 256>>>(A, B, C, K, K, K, rt);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    return best * 1e3;
}
int main()
{
    const int M = 4096, N = 512, K = 320;
    double *A, *B, *C, *h;
    hipMalloc(&A, sizeof(double) * M * K); hipMalloc(&B, sizeof(double) * N * K); hipMalloc(&C, sizeof(double) * 2048 * 256);
    h = (double *)malloc(sizeof(double) * M * K);
    for (int i = 0; i < M * K; i++) h[i] = (double)((i * 2654435761u) >> 20) / 4096.0 - 0.5;
    hipMemcpy(A, h, sizeof(double) * M * K, hipMemcpyHostToDevice);
    hipMemcpy(B, h + 1000, sizeof(double) * N * K, hipMemcpyHostToDevice);
    // same sums from both variants (the stage order is the same)
    double *c0 = (double *)malloc(sizeof(double) * 2048 * 256), *c1 = (double *)malloc(sizeof(double) * 2048 * 256);
    k<0, 1><<<128 * 3, 256>>>(A, B, C, K, K, K, 128); hipMemcpy(c0, C, sizeof(double) * 384 * 256, hipMemcpyDeviceToHost);
    k<1, 1><<<128 * 3, 256>>>(A, B, C, K, K, K, 128); hipMemcpy(c1, C, sizeof(double) * 384 * 256, hipMemcpyDeviceToHost);
    double d = 0; for (int i = 0; i < 384 * 256; i++) d = fmax(d, fabs(c0[i] - c1[i]));
    printf("max |two-buffer - three-buffer| = %g (TM=1)\n", d);
    k<0, 2><<<64 * 3, 256>>>(A, B, C, K, K, K, 64); hipMemcpy(c0, C, sizeof(double) * 192 * 256, hipMemcpyDeviceToHost);
    k<1, 2><<<64 * 3, 256>>>(A, B, C, K, K, K, 64); hipMemcpy(c1, C, sizeof(double) * 192 * 256, hipMemcpyDeviceToHost);
    d = 0; for (int i = 0; i < 192 * 256; i++) d = fmax(d, fabs(c0[i] - c1[i]));
    printf("max |two-buffer - three-buffer| = %g (TM=2)\n", d);
    k2<2><<<64 * 3, 256>>>(A, B, C, K, K, K, 64); hipMemcpy(c1, C, sizeof(double) * 192 * 256, hipMemcpyDeviceToHost);
    d = 0; for (int i = 0; i < 192 * 256; i++) d = fmax(d, fabs(c0[i] - c1[i]));
    printf("max |two-buffer - V2| = %g (TM=2)\n", d);
    k<0, 1><<<128 * 3, 256>>>(A, B, C, K, K, K, 128); hipMemcpy(c0, C, sizeof(double) * 384 * 256, hipMemcpyDeviceToHost);
    k2<1><<<128 * 3, 256>>>(A, B, C, K, K, K, 128); hipMemcpy(c1, C, sizeof(double) * 384 * 256, hipMemcpyDeviceToHost);
    d = 0; for (int i = 0; i < 384 * 256; i++) d = fmax(d, fabs(c0[i] - c1[i]));
    printf("max |two-buffer - V2| = %g (TM=1)\n", d);
    for (int ct : {2, 3, 4, 8}) {
        printf("64x64 tiles=%4d: two-buffer %.1f  three-buffer %.1f  V2 %.1f us\n", 64 * ct, run<0, 2>(A, B, C, K, 64, ct), run<1, 2>(A, B, C, K, 64, ct), run2<2>(A, B, C, K, 64, ct));
        printf("32x64 tiles=%4d: two-buffer %.1f  three-buffer %.1f  V2 %.1f us\n", 128 * ct, run<0, 1>(A, B, C, K, 128, ct), run<1, 1>(A, B, C, K, 128, ct), run2<1>(A, B, C, K, 128, ct));
    }
    return 0;
}
