// 64 x 64 output tiles for the products of the factor chain that have only a handful of 128 x 128 tiles: the levels
// of V = L^-1 below 512 and K^-1 = V^T V of a few hundred training points (gpry/gpr.py:1456-1457,
// sklearn:_gpr.py:640-642 at the sizes where GPry spends most of its iterations).  A 128 x 128 x 128 product is ONE
// workgroup of the engines in gemm_f64.hip / gemm_dma.hip: 14-20 us of one CU's matrix pipe while 255 CUs idle, and
// a level of the recursion is two such launches in series.  Here a workgroup (4 waves, 2 x 2, each 32 x 32 = 2 x 2
// MFMA tiles) owns a 64 x 64 tile, so the same product is spread over four times as many CUs; k advances in slabs of
// 32 (global -> registers -> LDS, double buffered), which halves the number of exposed load latencies of these
// short k-ranges.  Same operation order per output element as the 128-tile engines (k ascending in MFMA steps of 4
// from a zero accumulator -- EPI_SUB: from -C --; the extra leading / trailing steps a coarser tile origin brings in multiply structural
// zeros of the triangular operand), hence the same bits.  Requires every M, N a multiple of 64 and K of 32
// (GemmArgs.small64, set by the callers that guarantee it).
#include "common.h"

#define SB_T 64
#define SB_K 32
#define SKC2 34     // row stride of a k-contiguous image [64][34]: b64 fragment reads of 32 lanes hit 64 distinct banks
#define SMC2 80     // row stride of an m-contiguous image [32][80]: rows gq, gq+1 are 32 banks apart
#define KC_DOUBLES (SB_T * SKC2)
#define MC_DOUBLES (SB_K * SMC2)
#define OP_DOUBLES 2560          // >= max(KC_DOUBLES = 2176, MC_DOUBLES = 2560)

__device__ __forceinline__ void s64_load_kc(const double* __restrict__ P, int64_t ld, int row0, int k0, double (&r)[8]) {
    const int t = threadIdx.x;
    const double2* p = reinterpret_cast<const double2*>(P + (int64_t)(row0 + (t >> 2)) * ld + k0 + (t & 3) * 8);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y; r[4] = c.x; r[5] = c.y; r[6] = d.x; r[7] = d.y;
}
__device__ __forceinline__ void s64_store_kc(double* lds, const double (&r)[8]) {
    const int t = threadIdx.x;
    double2* p = reinterpret_cast<double2*>(lds + (t >> 2) * SKC2 + (t & 3) * 8);
    p[0] = make_double2(r[0], r[1]); p[1] = make_double2(r[2], r[3]);
    p[2] = make_double2(r[4], r[5]); p[3] = make_double2(r[6], r[7]);
}
__device__ __forceinline__ void s64_load_mc(const double* __restrict__ P, int64_t ld, int col0, int k0, double (&r)[8]) {
    const int t = threadIdx.x;
    const double2* p = reinterpret_cast<const double2*>(P + (int64_t)(k0 + (t >> 3)) * ld + col0 + (t & 7) * 8);
    const double2 a = p[0], b = p[1], c = p[2], d = p[3];
    r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y; r[4] = c.x; r[5] = c.y; r[6] = d.x; r[7] = d.y;
}
__device__ __forceinline__ void s64_store_mc(double* lds, const double (&r)[8]) {
    const int t = threadIdx.x;
    double2* p = reinterpret_cast<double2*>(lds + (t >> 3) * SMC2 + (t & 7) * 8);
    p[0] = make_double2(r[0], r[1]); p[1] = make_double2(r[2], r[3]);
    p[2] = make_double2(r[4], r[5]); p[3] = make_double2(r[6], r[7]);
}

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256) void gemm64_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double smem[4 * OP_DOUBLES];
    int bx, tb, bz;
    if (!gemm_block_order(g, &bx, &tb, &bz)) return;
    if (g.info != nullptr && *bset(g.info, tb, g.bstride) != 0) return;
    const double* A = bset(g.A, tb, g.bstride); const double* B = bset(g.B, tb, g.bstride); double* C = bset(g.C, tb, g.bstride);
    int M = g.M, N = g.N, K = g.K;
    if (g.batch != nullptr) {
        const GemmBatchItem it = g.batch[bz];
        A += it.a_off; B += it.b_off; C += it.c_off;
        M = it.M; N = it.N; K = it.K;
    }
    const int tiles_n = N / SB_T;
    const int ti = bx / tiles_n, tj = bx - ti * tiles_n;
    if (ti >= M / SB_T || (g.lower_only && tj > ti)) return;
    const int row0 = ti * SB_T, col0 = tj * SB_T;
    int kbeg = 0, kend = K;
    if (g.kmode == KM_A_LOWER) kend = min(K, row0 + SB_T);
    else if (g.kmode == KM_B_LOWER) kbeg = min(K, col0);
    else if (g.kmode == KM_AT_LOWER_B_LOWER) kbeg = min(K, max(row0, col0));
    else if (g.kmode == KM_B_UPPER) kend = min(K, col0 + SB_T);
    else if (g.kmode == KM_AT_LOWER) kbeg = min(K, row0);
    const int nslab = kend > kbeg ? (kend - kbeg) / SB_K : 0;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, gq = lane >> 4;
    // EPI_SUB: one chain that starts at -C (see gemm_dma_body.h)
    v4d acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) {
            acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (EPI == EPI_SUB) {
#pragma unroll
                for (int q = 0; q < 4; q++)
                    acc[i][j][q] = -C[(int64_t)(row0 + wr * 32 + i * 16 + gq + 4 * q) * g.ldc + col0 + wc * 32 + j * 16 + r];
            }
        }

    double ra[8], rb[8];
    auto load_slab = [&](int s) {
        const int k0 = kbeg + s * SB_K;
        if (AT) s64_load_mc(A, g.lda, row0, k0, ra); else s64_load_kc(A, g.lda, row0, k0, ra);
        if (BT) s64_load_kc(B, g.ldb, col0, k0, rb); else s64_load_mc(B, g.ldb, col0, k0, rb);
    };
    auto store_slab = [&](int buf) {
        double* As = smem + buf * 2 * OP_DOUBLES;
        double* Bs = As + OP_DOUBLES;
        if (AT) s64_store_mc(As, ra); else s64_store_kc(As, ra);
        if (BT) s64_store_kc(Bs, rb); else s64_store_mc(Bs, rb);
    };
    if (nslab > 0) { load_slab(0); store_slab(0); }
    __syncthreads();
    for (int s = 0; s < nslab; s++) {
        const int buf = s & 1;
        if (s + 1 < nslab) load_slab(s + 1);
        const double* As = smem + buf * 2 * OP_DOUBLES;
        const double* Bs = As + OP_DOUBLES;
#pragma unroll
        for (int kk = 0; kk < SB_K / 4; kk++) {
            double a[2], b[2];
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
                a[mi] = AT ? As[(kk * 4 + gq) * SMC2 + wr * 32 + mi * 16 + r] : As[(wr * 32 + mi * 16 + r) * SKC2 + kk * 4 + gq];
#pragma unroll
            for (int ni = 0; ni < 2; ni++)
                b[ni] = BT ? Bs[(wc * 32 + ni * 16 + r) * SKC2 + kk * 4 + gq] : Bs[(kk * 4 + gq) * SMC2 + wc * 32 + ni * 16 + r];
#pragma unroll
            for (int mi = 0; mi < 2; mi++)
#pragma unroll
                for (int ni = 0; ni < 2; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (s + 1 < nslab) store_slab(buf ^ 1);
        __syncthreads();
    }
    // epilogue.  f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4 * reg
#pragma unroll
    for (int mi = 0; mi < 2; mi++)
#pragma unroll
        for (int ni = 0; ni < 2; ni++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double* p = C + (int64_t)(row0 + wr * 32 + mi * 16 + gq + 4 * q) * g.ldc + col0 + wc * 32 + ni * 16 + r;
                const double v = acc[mi][ni][q];
                if (EPI == EPI_STORE) *p = v;
                else *p = -v;           // EPI_STORE_NEG; EPI_SUB: the chain ran on -C
            }
}

template <bool AT, bool BT>
static int launch64(gpry_ctx* ctx, const GemmArgs& g, int epi, dim3 grid) {
    hipStream_t st = g.stream ? g.stream : ctx->stream;
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm64_kernel<AT, BT, EPI_STORE>), grid, dim3(256), 0, st, g); break;
        case EPI_STORE_NEG: hipLaunchKernelGGL((gemm64_kernel<AT, BT, EPI_STORE_NEG>), grid, dim3(256), 0, st, g); break;
        case EPI_SUB: hipLaunchKernelGGL((gemm64_kernel<AT, BT, EPI_SUB>), grid, dim3(256), 0, st, g); break;
        default: return gpry_fail(ctx, -1, "gemm64: bad epilogue %d", epi);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// g.M / g.N: the largest item of a batched launch (as for gemm_f64_launch)
int gemm64_launch(gpry_ctx* ctx, const GemmArgs& g0, bool a_trans, bool b_trans, int epi) {
    GemmArgs g = g0;
    gemm_fill_batch(ctx, &g);
    if (g.M % SB_T || g.N % SB_T || (g.batch == nullptr && g.K % SB_K))
        return gpry_fail(ctx, -1, "gemm64: M, N must be multiples of 64 and K of 32");
    if (a_trans && b_trans) return gpry_fail(ctx, -1, "gemm64: A^T B^T is not built");
    const dim3 grid((unsigned)((g.M / SB_T) * (g.N / SB_T)), 1, gemm_grid_z(g));
    if (!a_trans && !b_trans) return launch64<false, false>(ctx, g, epi, grid);
    if (!a_trans && b_trans) return launch64<false, true>(ctx, g, epi, grid);
    return launch64<true, false>(ctx, g, epi, grid);
}
