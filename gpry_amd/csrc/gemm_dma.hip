// FP64 MFMA GEMM with LDS-DMA staging and the hand-placed software pipeline of the sweep
// contraction (sweep_gemm.hip, variant 3), generalised to the products of the factor chain:
//   NT  C -= A B^T        Cholesky trailing update (SYRK, lower tiles only)
//   NN  C  = -+ A B       V = L^-1 recursion levels
//   TN  C  = A^T B        K^-1 = V^T V
// The register-staged engine (gemm_f64.hip) keeps one workgroup per CU only ~45 % busy (60 us for
// a 128x128x256 tile whose MFMAs take 27 us): most launches of the factor chain have fewer tiles
// than the GPU has workgroup slots, so nothing hides its per-slab global -> VGPR -> LDS round trip.
// Here each wave issues its eight 1-KiB DMA pieces of slab s+2 in the shadow of the MFMAs of slab s.
//
// Operand images in LDS (per slab of 16 k):
//   KC (k contiguous in memory: A of NN/NT, B of NT): [128 rows][8 pieces of 16 B], unpadded, bank
//      conflicts removed by an XOR swizzle on the SOURCE side (LDS piece p' of row r holds global
//      piece p' ^ ((r >> 1) & 7));
//   MC (m/n contiguous: B of NN/TN, A of TN): [16 k-rows][144 doubles] (128 + pad).
// Requirements (checked by gemm_dma_usable): M, N multiples of 128, every k-range a multiple of 32
// and >= 32, leading dimensions and offsets even (16-byte pieces).  Everything else takes the
// register-staged kernel.
#include "common.h"

#include "gemm_dma_body.h"

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(GemmArgs g) {
    constexpr int A_SZ = !AT ? KC_DOUBLES : MC_DOUBLES, B_SZ = BT ? KC_DOUBLES : MC_DOUBLES;
    __shared__ __attribute__((aligned(16))) double smem[2 * (A_SZ + B_SZ)];
    gemm_dma_tile_body<AT, BT, EPI>(g, smem, (int)blockIdx.x, (int)blockIdx.y, (int)blockIdx.z);
}

// can this product go through the DMA kernel?  (host side; batch items are checked by the caller
// that builds them: see chol.hip)
bool gemm_dma_usable(const GemmArgs& g, int M, int N, int K) {
    if (M <= 0 || N <= 0 || K < 32) return false;
    if (M % BM || N % BN || K % 32) return false;
    if ((g.lda | g.ldb | g.ldc) & 1) return false;
    if (g.kskew || g.stagger || g.diag) return false;
    // triangular k-ranges start / end on tile boundaries (multiples of 128); split-K works on slab pairs
    return true;
}

template <bool AT, bool BT>
static int gd_launch_epi(gpry_ctx* ctx, const GemmArgs& g, int epi, dim3 grid) {
    hipStream_t st = g.stream ? g.stream : ctx->stream;
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_dma_kernel<AT, BT, EPI_STORE>), grid, dim3(256), (size_t)g.extra_lds, st, g); break;
        case EPI_STORE_NEG: hipLaunchKernelGGL((gemm_dma_kernel<AT, BT, EPI_STORE_NEG>), grid, dim3(256), (size_t)g.extra_lds, st, g); break;
        case EPI_SUB: hipLaunchKernelGGL((gemm_dma_kernel<AT, BT, EPI_SUB>), grid, dim3(256), (size_t)g.extra_lds, st, g); break;
        default: return gpry_fail(ctx, -1, "gemm_dma: bad epilogue %d", epi);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// the product only (same grid as gemm_f64_launch); the caller runs the split-K reduce
int gemm_dma_launch_product(gpry_ctx* ctx, const GemmArgs& g0, bool a_trans, bool b_trans, int epi, dim3 grid) {
    GemmArgs g = g0;
    if ((g.tile_map & 15) == TM_ROWMAJOR) {
        g.tile_map = TM_BALANCED;
        const int tm = g.M / BM, tn = g.N / BN;      // batched launches: the largest item
        if (g.lower_only && tm == tn && g.batch == nullptr) grid.x = (unsigned)(tm * (tm + 1) / 2);
    }
    if (!a_trans && !b_trans) return gd_launch_epi<false, false>(ctx, g, epi, grid);
    if (!a_trans && b_trans) return gd_launch_epi<false, true>(ctx, g, epi, grid);
    if (a_trans && !b_trans) return gd_launch_epi<true, false>(ctx, g, epi, grid);
    return gpry_fail(ctx, -1, "gemm_dma: the TT layout is not built");
}
