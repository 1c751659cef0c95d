// gemm.hip — fp64 MFMA GEMM family for the dense contractions of the SGPR predict path.
//
//   C[M][N] = A[M][K] . B[N][K]^T      (both operands row-major, K contiguous)
// on v_mfma_f64_16x16x4_f64, with three fused epilogues:
//   EPI_KERNEL  K_nm: k = [Z_i==Z_q] dot^eta (+ lone-atom term), writes K, Aw = mu_q eta dot^(eta-1)
//               and per-block energy partials   (similarity/universal.py:109-122,
//               similarity/similarity.py:94-103, calculator/active.py:549-560)
//   EPI_STORE   W = Aw . Pm  (dE/dp-hat, the reverse-pass seed of calculator/active.py:587-599)
//   EPI_ROWSQ   c_i = |choli . k_i|^2 without materialising choli.K^T (calculator/active.py:782-783)
//
// Operands are species-sorted, so K_nm, K_mm, choli are block-diagonal: each 64x64 tile works
// only on the species blocks it touches (tile skip + trimmed reduction range).
// All leading dimensions are multiples of 16 and all row counts are padded to 64 with zeros
// by the allocator (api.hip), so the main loop carries no bounds checks.
//
// Tile: 64x64 (TM = 2) or 32x64 (TM = 1) per 256-thread workgroup, 4 waves as 2x2, each wave TM x 2 MFMA tiles;
// K stages through LDS, fetched ahead through registers.  The three products of a step run on the 32-row
// form with 16-deep stages (KD = 16): THREE stages in 36 KB of XOR-swizzled LDS, one barrier per stage, stores and
// loads issued in the shadow of the MFMAs, four (W + covloss: four register sets, loads 3.5 stages ahead) or two
// (K_nm: the eight-wave form of gemm_tile.inc, six sets, 5.5 stages ahead) workgroups per CU; K_mm, the Cholesky
// trailing update and dense launches keep the 64-row form with two 32-deep stages.  Every form accumulates a dot product over k in the same
// order: the results do not depend on the tile shape (tests/test_hip_paths.py::test_gemm_tile_shapes_agree_bit_for_bit).
// Measured alternatives on the 4096 x 512 x 320 K_nm product: LDS-free fragment-shaped direct loads 35 us (TA-bound:
// each quad of lanes touches four cache lines); a 16-row panel against 256 columns per workgroup (one wave per SIMD)
// 19 us (tools/ubench_gemm5.hip).  Built with -mllvm -amdgpu-mfma-vgpr-form=1 (build.sh): accumulators stay in VGPRs.
#include "sgpr_internal.h"

#include "gemm_tile.inc"

template <int EPI, int TM = 2, int KD = 32>
__global__ __launch_bounds__(256, KD == 16 ? 4 : 2) void gemm_nt_kernel(GemmArgs g)
{
    using L = GemmLds<EPI, TM, KD>;
    __shared__ double As[L::NBUF * L::ASZ];
    __shared__ double Bs[L::NBUF * L::BSZ];
    if constexpr (EPI == EPI_WCOV && TM == 1 && KD == 16) {
        // (a tile table entry may name a successor — position + 1 in the high half of .y —: the tiles that would otherwise
        // wait for a free slot run behind the first tile to finish on the least loaded CUs, api.hip::build_tiles)
        for (int bid = (int)blockIdx.x;;) {
            gemm_tile_body<EPI, TM, KD>(g, bid, As, Bs);
            const int nxt = (int)((unsigned)g.p.tiles[bid].y >> 16) - 1;
            if (nxt < 0) break;
            __syncthreads();
            bid = nxt;
        }
    } else {
        gemm_tile_body<EPI, TM, KD>(g, (int)blockIdx.x, As, Bs);
    }
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel8(GemmArgs g)
{
    using L = GemmLds<EPI, 1, 16>;
    __shared__ double As[L::NBUF * L::ASZ];
    __shared__ double Bs[L::NBUF * L::BSZ];
    gemm_tile_body8<EPI>(g, (int)blockIdx.x, As, Bs);
}

// the half tile: 16 x 64 on four waves, one per SIMD (gemm_tile_body8<EPI, true>)
template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel4h(GemmArgs g)
{
    using L = GemmLds<EPI, 1, 16>;
    __shared__ double As[L::NBUF * L::ASZ];
    __shared__ double Bs[L::NBUF * L::BSZ];
    gemm_tile_body8<EPI, true>(g, (int)blockIdx.x, As, Bs);
}

template <int EPI>
__global__ __launch_bounds__(512, 2) void gemm_nt_kernel8r64(GemmArgs g)
{
    __shared__ double As[3 * 64 * 16];
    __shared__ double Bs[3 * 64 * 16];
    gemm_tile_body8r64<EPI>(g, (int)blockIdx.x, As, Bs);
}

// the same tile on three register stage sets: <= 80 VGPRs, THREE workgroups (24 waves) per CU (48 KB of LDS each)
template <int EPI>
__global__ __launch_bounds__(512, 6) void gemm_nt_kernel8r64x3(GemmArgs g)
{
    __shared__ double As[3 * 64 * 16];
    __shared__ double Bs[3 * 64 * 16];
    gemm_tile_body8r64<EPI, 3>(g, (int)blockIdx.x, As, Bs);
}

void launch_gemm_wcov(const GemmParams &pw, const GemmParams &pc, const int4 *tiles, int ntiles, hipStream_t st, int grid)
{
    if (ntiles <= 0) return;
    const bool chained = grid > 0 && grid < ntiles;
    GemmArgs g = {};
    g.p = pw;
    g.p2 = pc;
    g.p.tiles = tiles;
    g.p.ntiles = ntiles;
    g.ieta = -1;
    g.stamps = pw.stamps;
    if (pw.bm == 16) hipLaunchKernelGGL(gemm_nt_kernel4h<EPI_WCOV>, dim3(ntiles), dim3(256), 0, st, g);
    else if (pw.bm == 64 && pw.waves == 8 && pw.wgs == 3) hipLaunchKernelGGL(gemm_nt_kernel8r64x3<EPI_WCOV>, dim3(ntiles), dim3(512), 0, st, g);
    else if (pw.bm == 64 && pw.waves == 8) hipLaunchKernelGGL(gemm_nt_kernel8r64<EPI_WCOV>, dim3(ntiles), dim3(512), 0, st, g);
    else if (pw.bm == 32 && pw.kd == 16) hipLaunchKernelGGL((gemm_nt_kernel<EPI_WCOV, 1, 16>), dim3(chained ? grid : ntiles), dim3(256), 0, st, g);
    else if (pw.bm == 32) hipLaunchKernelGGL((gemm_nt_kernel<EPI_WCOV, 1>), dim3(ntiles), dim3(256), 0, st, g);
    else hipLaunchKernelGGL((gemm_nt_kernel<EPI_WCOV, 2>), dim3(ntiles), dim3(256), 0, st, g);
}

void launch_gemm_fused(const GemmParams &pk, const GemmParams &pw, const GemmParams &pc, const int4 *tiles, int ntiles,
                       double *Epart, int *panel_cnt, int epoch, int *err, hipStream_t st)
{
    if (ntiles <= 0) return;
    GemmArgs g = {};
    g.p = pw; g.p2 = pc; g.p3 = pk;
    g.p.tiles = tiles;
    g.p.ntiles = ntiles;
    g.ieta = (pk.eta == (double)(int)pk.eta && pk.eta >= 1.0 && pk.eta <= 64.0) ? (int)pk.eta : -1;
    g.row_slot = pk.row_slot;
    g.col_slot = pk.col_slot;
    g.Epart = Epart;
    g.panel_cnt = panel_cnt; g.epoch = epoch; g.fuse_err = err;
    hipLaunchKernelGGL((gemm_nt_kernel<EPI_FUSED, 1, 16>), dim3(ntiles), dim3(256), 0, st, g);
}

void launch_gemm_nt(const GemmParams &p, GemmEpilogue epi, hipStream_t st)
{
    if (p.M <= 0 || p.N <= 0) return;
    GemmArgs g = {};
    g.p = p;
    g.ieta = (p.eta == (double)(int)p.eta && p.eta >= 1.0 && p.eta <= 64.0) ? (int)p.eta : -1;
    g.row_slot = p.row_slot;
    g.col_slot = p.col_slot;
    g.Epart = p.Esum;
    g.stamps = p.stamps;
    g.row_tiles = (p.M + BM - 1) / BM;
    g.col_tiles = (p.N + BN - 1) / BN;
    g.rows_pad = (g.row_tiles + 7) / 8 * 8;
    dim3 grid(p.tiles ? p.ntiles : g.rows_pad * g.col_tiles), block(256);
    if (p.tiles && p.ntiles <= 0) return;
    if (p.bm == 16 && p.tiles && (epi == EPI_KERNEL || epi == EPI_STORE || epi == EPI_ROWSQ)) {
        if (epi == EPI_KERNEL) hipLaunchKernelGGL(gemm_nt_kernel4h<EPI_KERNEL>, grid, dim3(256), 0, st, g);
        else if (epi == EPI_STORE) hipLaunchKernelGGL(gemm_nt_kernel4h<EPI_STORE>, grid, dim3(256), 0, st, g);
        else hipLaunchKernelGGL(gemm_nt_kernel4h<EPI_ROWSQ>, grid, dim3(256), 0, st, g);
        return;
    }
    if (p.waves == 8 && p.bm == 64 && p.wgs == 3 && p.tiles && (epi == EPI_KERNEL || epi == EPI_STORE || epi == EPI_ROWSQ)) {
        if (epi == EPI_KERNEL) hipLaunchKernelGGL(gemm_nt_kernel8r64x3<EPI_KERNEL>, grid, dim3(512), 0, st, g);
        else if (epi == EPI_STORE) hipLaunchKernelGGL(gemm_nt_kernel8r64x3<EPI_STORE>, grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL(gemm_nt_kernel8r64x3<EPI_ROWSQ>, grid, dim3(512), 0, st, g);
        return;
    }
    if (p.waves == 8 && p.bm == 64 && p.tiles && (epi == EPI_KERNEL || epi == EPI_STORE || epi == EPI_ROWSQ)) {
        if (epi == EPI_KERNEL) hipLaunchKernelGGL(gemm_nt_kernel8r64<EPI_KERNEL>, grid, dim3(512), 0, st, g);
        else if (epi == EPI_STORE) hipLaunchKernelGGL(gemm_nt_kernel8r64<EPI_STORE>, grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL(gemm_nt_kernel8r64<EPI_ROWSQ>, grid, dim3(512), 0, st, g);
        return;
    }
    if (p.waves == 8 && p.bm == 32 && p.tiles && p.kd == 16 && (epi == EPI_KERNEL || epi == EPI_STORE)) {
        if (epi == EPI_KERNEL) hipLaunchKernelGGL(gemm_nt_kernel8<EPI_KERNEL>, grid, dim3(512), 0, st, g);
        else hipLaunchKernelGGL(gemm_nt_kernel8<EPI_STORE>, grid, dim3(512), 0, st, g);
        return;
    }
    if (epi == EPI_STORE && p.bm == 32 && p.tiles && p.kd == 16) hipLaunchKernelGGL((gemm_nt_kernel<EPI_STORE, 1, 16>), grid, block, 0, st, g);
    else if (epi == EPI_ROWSQ && p.bm == 32 && p.tiles && p.kd == 16) hipLaunchKernelGGL((gemm_nt_kernel<EPI_ROWSQ, 1, 16>), grid, block, 0, st, g);
    else if (epi == EPI_STORE && p.bm == 32 && p.tiles) hipLaunchKernelGGL((gemm_nt_kernel<EPI_STORE, 1>), grid, block, 0, st, g);
    else if (epi == EPI_ROWSQ && p.bm == 32 && p.tiles) hipLaunchKernelGGL((gemm_nt_kernel<EPI_ROWSQ, 1>), grid, block, 0, st, g);
    else if (epi == EPI_STORE) hipLaunchKernelGGL(gemm_nt_kernel<EPI_STORE>, grid, block, 0, st, g);
    else if (epi == EPI_KERNEL && p.bm == 32 && p.tiles && p.kd == 16) hipLaunchKernelGGL((gemm_nt_kernel<EPI_KERNEL, 1, 16>), grid, block, 0, st, g);
    else if (epi == EPI_KERNEL && p.bm == 32 && p.tiles) hipLaunchKernelGGL((gemm_nt_kernel<EPI_KERNEL, 1>), grid, block, 0, st, g);
    else if (epi == EPI_KERNEL) hipLaunchKernelGGL(gemm_nt_kernel<EPI_KERNEL>, grid, block, 0, st, g);
    else if (epi == EPI_SUBLOWER) hipLaunchKernelGGL(gemm_nt_kernel<EPI_SUBLOWER>, grid, block, 0, st, g);
    else hipLaunchKernelGGL(gemm_nt_kernel<EPI_ROWSQ>, grid, block, 0, st, g);
}
