// Sweep contraction with direct global->LDS staging (LDS-DMA, global_load_lds_dwordx4):
//   per 128x128 tile:  S[ti][m] = sum_i ( sum_{k<=i} V[i][k] * Kst[k][m] )^2
// Same tiling, MFMA schedule and SUMSQ epilogue as gemm_f64_kernel<NN,SUMSQ>; what changes
// is the staging: no VGPR round trip and no ds_write pass -- each wave issues eight 1-KiB
// DMA pieces per slab straight into the other LDS buffer and then only waits (vmcnt(0))
// before the slab barrier.  The DMA destination is linear (wave base + lane*16 B), so
//   * the K*^T slab keeps the padded [16][144] image (one piece = one 1-KiB k-row), and
//   * the V slab is an UNPADDED [128 rows][8 pieces of 16 B] image whose bank conflicts are
//     removed by an XOR swizzle applied on the SOURCE side: LDS piece p' of row r holds
//     global piece p = p' ^ ((r >> 1) & 7); the fragment read applies the same involution.
// Requires M, N multiples of 128 (true for the sweep: Np and the padded chunk).
#include "common.h"

#define BM 128
#define BN 128
#define BK 16
#define SMC 144
#define A_DOUBLES 2048   // 128 x 16, unpadded
#define B_DOUBLES 2304   // 16 x 144
#define BUF_DOUBLES (A_DOUBLES + B_DOUBLES)

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void dma16(const double* g, double* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

__global__ __launch_bounds__(256, 2) void sweep_gemm_dma_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double smem[2 * BUF_DOUBLES];
    const int M = g.M, N = g.N, K = g.K;
    const int tiles_m = M / BM, tiles_n = N / BN;
    // XCD-aware super-tile map (see gemm_f64.hip)
    int ti, tj;
    {
        const int b = blockIdx.x;
        const int a = (g.tile_map >> 4) & 15, c = 6 - a;
        const int nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
        const int xcd = b & 7, q = b >> 3;
        const int s = (q >> 6) * 8 + xcd, within = q & 63;
        if (s >= nsi * nsj) return;
        const int si = nsi - 1 - s / nsj, sj = s % nsj;
        ti = (si << a) + (within >> c);
        tj = (sj << c) + (within & ((1 << c) - 1));
        if (ti >= tiles_m || tj >= tiles_n) return;
    }
    const int row0 = ti * BM, col0 = tj * BN;
    const int kend = min(K, row0 + BM);          // V is lower triangular
    const int nslab = kend / BK;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, gq = lane >> 4;

    // per-lane source addresses of this wave's 4 + 4 DMA pieces of a slab (k0 added per slab)
    const double* srcA[4];
    int dstA[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = wave * 4 + j;                       // piece index: rows q*8 .. q*8+7
        const int row = q * 8 + (lane >> 3), pp = lane & 7;
        const int p = pp ^ ((row >> 1) & 7);              // swizzle on the source side
        srcA[j] = g.A + (int64_t)(row0 + row) * g.lda + 2 * p;
        dstA[j] = q * 128;                                // doubles, wave-uniform
    }
    const double* srcB = g.B + (int64_t)(wave * 4) * g.ldb + col0 + 2 * lane;

    auto issue = [&](int s, int buf) {
        double* As = smem + buf * BUF_DOUBLES;
        double* Bs = As + A_DOUBLES;
        const int k0 = s * BK;
#pragma unroll
        for (int j = 0; j < 4; j++) dma16(srcA[j] + k0, As + dstA[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) dma16(srcB + (int64_t)(k0 + j) * g.ldb, Bs + (wave * 4 + j) * SMC);
    };

    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // fragment offsets of the swizzled A image that do not depend on the slab
    int aoff[4][2];   // [mi][k parity pair]: row*16 ; swizzle key
#pragma unroll
    for (int mi = 0; mi < 4; mi++) {
        const int row = wr * 64 + mi * 16 + r;
        aoff[mi][0] = row * 16;
        aoff[mi][1] = (row >> 1) & 7;
    }

    issue(0, 0);
    __syncthreads();   // emits s_waitcnt vmcnt(0): the DMA pieces of slab 0 have landed

    for (int s = 0; s < nslab; s++) {
        const int buf = s & 1;
        if (s + 1 < nslab) issue(s + 1, buf ^ 1);
        const double* As = smem + buf * BUF_DOUBLES;
        const double* Bs = As + A_DOUBLES;
#pragma unroll
        for (int kk = 0; kk < BK / 4; kk++) {
            const int k = kk * 4 + gq;
            double a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                a[mi] = As[aoff[mi][0] + 2 * ((k >> 1) ^ aoff[mi][1]) + (k & 1)];
#pragma unroll
            for (int ni = 0; ni < 4; ni++) b[ni] = Bs[k * SMC + wc * 64 + ni * 16 + r];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        __syncthreads();   // vmcnt(0) + barrier: slab s+1 landed, everybody done with slab s
    }

    // column sums of squares over the 128 rows of the tile (C/D: col = lane&15, row = (lane>>4)+4q)
    double cs[4];
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
        double sacc = 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int q = 0; q < 4; q++) sacc = fma(acc[mi][ni][q], acc[mi][ni][q], sacc);
        sacc += __shfl_xor(sacc, 16);
        sacc += __shfl_xor(sacc, 32);
        cs[ni] = sacc;
    }
    double* red = smem;
    if (gq == 0) {
#pragma unroll
        for (int ni = 0; ni < 4; ni++) red[wr * 128 + wc * 64 + ni * 16 + r] = cs[ni];
    }
    __syncthreads();
    if (threadIdx.x < 128)
        g.C[(int64_t)ti * g.ldc + col0 + threadIdx.x] = red[threadIdx.x] + red[128 + threadIdx.x];
}

int sweep_gemm_dma_launch(gpry_ctx* ctx, const GemmArgs& g) {
    if (g.M % BM || g.N % BN || g.K % BK) return gpry_fail(ctx, -1, "sweep_gemm_dma: dims must be multiples of 128");
    const int tiles_m = g.M / BM, tiles_n = g.N / BN;
    const int a = (g.tile_map >> 4) & 15, c = 6 - a;
    int64_t nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
    int64_t ns = (nsi * nsj + 7) / 8 * 8;
    hipLaunchKernelGGL(sweep_gemm_dma_kernel, dim3((unsigned)(ns * 64)), dim3(256), 0, ctx->stream, g);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------
// Variant 2: 128 x 256 tile, 8 waves (2 x 4), one workgroup per CU, three-stage LDS ring.
// A slab of V (128 x 16) now serves 256 candidates, i.e. 25 % fewer staged bytes per MFMA
// than the 128 x 128 tile, and a DMA piece has two slab times to land: at step s the wave
// waits only for its own pieces of slab s (s_waitcnt vmcnt(6) leaves slab s+1 in flight),
// crosses a raw s_barrier, issues slab s+2 into the buffer that was read in step s-1, and
// multiplies slab s.
#define BN2 256
#define SB2 272
#define B2_DOUBLES (16 * SB2)
#define STAGE2 (A_DOUBLES + B2_DOUBLES)

__global__ __launch_bounds__(512, 2) void sweep_gemm_dma256_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double smem[3 * STAGE2];
    const int M = g.M, N = g.N, K = g.K;
    const int tiles_m = M / BM, tiles_n = N / BN2;
    int ti, tj;
    {
        const int b = blockIdx.x;
        const int a = (g.tile_map >> 4) & 15, c = 5 - a;      // 32 tiles per XCD wave of workgroups
        const int nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
        const int xcd = b & 7, q = b >> 3;
        const int s = (q >> 5) * 8 + xcd, within = q & 31;
        if (s >= nsi * nsj) return;
        const int si = nsi - 1 - s / nsj, sj = s % nsj;
        ti = (si << a) + (within >> c);
        tj = (sj << c) + (within & ((1 << c) - 1));
        if (ti >= tiles_m || tj >= tiles_n) return;
    }
    const int row0 = ti * BM, col0 = tj * BN2;
    const int kend = min(K, row0 + BM);
    const int nslab = kend / BK;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 2, wc = wave & 3;
    const int r = lane & 15, gq = lane >> 4;

    const double* srcA[2];
    int dstA[2];
#pragma unroll
    for (int j = 0; j < 2; j++) {
        const int q = wave * 2 + j;
        const int row = q * 8 + (lane >> 3), pp = lane & 7;
        const int p = pp ^ ((row >> 1) & 7);
        srcA[j] = g.A + (int64_t)(row0 + row) * g.lda + 2 * p;
        dstA[j] = q * 128;
    }
    const double* srcB[4];
    int dstB[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int idx = wave * 4 + j, krow = idx >> 1, half = idx & 1;
        srcB[j] = g.B + (int64_t)krow * g.ldb + col0 + half * 128 + 2 * lane;
        dstB[j] = krow * SB2 + half * 128;
    }
    auto issue = [&](int s, int buf) {
        double* As = smem + buf * STAGE2;
        double* Bs = As + A_DOUBLES;
        const int k0 = s * BK;
#pragma unroll
        for (int j = 0; j < 2; j++) dma16(srcA[j] + k0, As + dstA[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) dma16(srcB[j] + (int64_t)k0 * g.ldb, Bs + dstB[j]);
    };

    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    int aoff[4][2];
#pragma unroll
    for (int mi = 0; mi < 4; mi++) {
        const int row = wr * 64 + mi * 16 + r;
        aoff[mi][0] = row * 16;
        aoff[mi][1] = (row >> 1) & 7;
    }

    issue(0, 0);
    if (nslab > 1) issue(1, 1);
    int buf = 0;
    for (int s = 0; s < nslab; s++) {
        // own pieces of slab s have landed (the newest six, slab s+1, may still be in flight)
        if (s + 1 < nslab) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        int nb = buf + 2; if (nb >= 3) nb -= 3;
        if (s + 2 < nslab) issue(s + 2, nb);
        const double* As = smem + buf * STAGE2;
        const double* Bs = As + A_DOUBLES;
#pragma unroll
        for (int kk = 0; kk < BK / 4; kk++) {
            const int k = kk * 4 + gq;
            double a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                a[mi] = As[aoff[mi][0] + 2 * ((k >> 1) ^ aoff[mi][1]) + (k & 1)];
#pragma unroll
            for (int ni = 0; ni < 4; ni++) b[ni] = Bs[k * SB2 + wc * 64 + ni * 16 + r];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's LDS reads of slab s are done
        buf = buf + 1; if (buf >= 3) buf = 0;
    }
    __syncthreads();
    double cs[4];
#pragma unroll
    for (int ni = 0; ni < 4; ni++) {
        double sacc = 0.0;
#pragma unroll
        for (int mi = 0; mi < 4; mi++)
#pragma unroll
            for (int q = 0; q < 4; q++) sacc = fma(acc[mi][ni][q], acc[mi][ni][q], sacc);
        sacc += __shfl_xor(sacc, 16);
        sacc += __shfl_xor(sacc, 32);
        cs[ni] = sacc;
    }
    double* red = smem;
    if (gq == 0) {
#pragma unroll
        for (int ni = 0; ni < 4; ni++) red[wr * 256 + wc * 64 + ni * 16 + r] = cs[ni];
    }
    __syncthreads();
    if (threadIdx.x < 256)
        g.C[(int64_t)ti * g.ldc + col0 + threadIdx.x] = red[threadIdx.x] + red[256 + threadIdx.x];
}

int sweep_gemm_dma256_launch(gpry_ctx* ctx, const GemmArgs& g) {
    if (g.M % BM || g.N % BN2 || g.K % BK) return gpry_fail(ctx, -1, "sweep_gemm_dma256: bad dims");
    const int tiles_m = g.M / BM, tiles_n = g.N / BN2;
    const int a = (g.tile_map >> 4) & 15, c = 5 - a;
    if (c < 0) return gpry_fail(ctx, -1, "sweep_gemm_dma256: tile map exponent must be <= 5");
    int64_t nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
    int64_t ns = (nsi * nsj + 7) / 8 * 8;
    hipLaunchKernelGGL(sweep_gemm_dma256_kernel, dim3((unsigned)(ns * 32)), dim3(512), 0, ctx->stream, g);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
