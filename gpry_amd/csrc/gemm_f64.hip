// FP64 MFMA GEMM for gfx950 (v_mfma_f64_16x16x4_f64), the contraction engine behind
//   - the NORA sweep  M = V * K*^T   (gpry/gpr.py:1204 dtrmm + :1208 einsum, fused)
//   - Cholesky trailing updates, V = L^-1 recursion, K^-1 = V^T V  (gpry/gpr.py:1456-1457,
//     sklearn:_gpr.py:640-642)
//
// Workgroup: 256 threads = 4 waves (64 lanes) in a 2x2 arrangement, each wave owns a
// 64x64 block of the 128x128 output tile as 4x4 MFMA tiles (16 accumulators of 4 f64).
// K is consumed in slabs of 16 staged global -> registers -> LDS, double buffered, one
// barrier per slab.  LDS images are padded so that every ds_read_b64 fragment read is
// bank-conflict free (strides 18 and 144 doubles).
#include "common.h"

#define BM 128
#define BN 128
#define BK 16
#define SKC 18    // row stride (doubles) of a K-contiguous tile image   [128][18]
#define SMC 144   // row stride (doubles) of an MN-contiguous tile image [16][144]
#define TILE_DOUBLES 2304  // 128*18 == 16*144

struct TileCoord { int ti, tj, valid; };

__device__ __forceinline__ TileCoord map_tile(const GemmArgs& g, int tiles_m, int tiles_n) {
    TileCoord t;
    int b = blockIdx.x;
    if ((g.tile_map & 15) == TM_SWEEP) {
        // XCD-aware super-tiles: blocks b, b+8, b+16.. share an XCD (round-robin
        // dispatch); give each XCD 64 consecutive slots = one (2^a x 64/2^a) super-tile so
        // that the V row-panels and K* column-panels of a super-tile are served by that
        // XCD's L2.  Super-tiles are ordered by descending row index (longest k first).
        // a = bits 4..7 of tile_map (default 3: 8 row-tiles x 8 candidate tiles).
        const int a = (g.tile_map >> 4) & 15, c = 6 - a;
        int nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
        int xcd = b & 7, q = b >> 3;
        int s = (q >> 6) * 8 + xcd;
        int within = q & 63;
        if (s >= nsi * nsj) { t.valid = 0; t.ti = t.tj = 0; return t; }
        int si = nsi - 1 - s / nsj, sj = s % nsj;
        t.ti = (si << a) + (within >> c);
        t.tj = (sj << c) + (within & ((1 << c) - 1));
    } else {
        t.ti = b / tiles_n;
        t.tj = b - t.ti * tiles_n;
    }
    t.valid = (t.ti < tiles_m && t.tj < tiles_n);
    if (g.lower_only && t.tj > t.ti) t.valid = 0;
    return t;
}

// Stage loaders.  KC: operand stored with k contiguous, 128 rows x 16 k per slab.
//                 MC: operand stored with m/n contiguous, 16 k-rows x 128 per slab.
__device__ __forceinline__ void load_kc(const double* __restrict__ P, int64_t ld, int row0,
                                        int rows, int k0, double (&r)[8]) {
    int t = threadIdx.x;
    int row = row0 + (t >> 1);
    if (row < rows) {
        const double2* p = reinterpret_cast<const double2*>(P + (int64_t)row * ld + k0 + (t & 1) * 8);
        double2 a = p[0], b = p[1], c = p[2], d = p[3];
        r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
        r[4] = c.x; r[5] = c.y; r[6] = d.x; r[7] = d.y;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = 0.0;
    }
}
__device__ __forceinline__ void store_kc(double* lds, const double (&r)[8]) {
    int t = threadIdx.x;
    double2* p = reinterpret_cast<double2*>(lds + (t >> 1) * SKC + (t & 1) * 8);
    p[0] = make_double2(r[0], r[1]); p[1] = make_double2(r[2], r[3]);
    p[2] = make_double2(r[4], r[5]); p[3] = make_double2(r[6], r[7]);
}
__device__ __forceinline__ void load_mc(const double* __restrict__ P, int64_t ld, int col0,
                                        int cols, int k0, double (&r)[8]) {
    int t = threadIdx.x;
    int col = col0 + (t & 15) * 8;
    if (col < cols) {
        const double2* p = reinterpret_cast<const double2*>(P + (int64_t)(k0 + (t >> 4)) * ld + col);
        double2 a = p[0], b = p[1], c = p[2], d = p[3];
        r[0] = a.x; r[1] = a.y; r[2] = b.x; r[3] = b.y;
        r[4] = c.x; r[5] = c.y; r[6] = d.x; r[7] = d.y;
    } else {
#pragma unroll
        for (int i = 0; i < 8; i++) r[i] = 0.0;
    }
}
__device__ __forceinline__ void store_mc(double* lds, const double (&r)[8]) {
    int t = threadIdx.x;
    double2* p = reinterpret_cast<double2*>(lds + (t >> 4) * SMC + (t & 15) * 8);
    p[0] = make_double2(r[0], r[1]); p[1] = make_double2(r[2], r[3]);
    p[2] = make_double2(r[4], r[5]); p[3] = make_double2(r[6], r[7]);
}

template <bool AT, bool BT, int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double smem[4 * TILE_DOUBLES];
    int tb, bz;
    gemm_block_z(g, (int)blockIdx.z, &tb, &bz);
    if (g.info != nullptr && *bset(g.info, tb, g.bstride) != 0) return;

    double* const C0 = bset(g.C, tb, g.bstride);
    const double* A = bset(g.A, tb, g.bstride); const double* B = bset(g.B, tb, g.bstride); double* C = C0;
    int M = g.M, N = g.N, K = g.K;
    if (g.batch != nullptr) {
        GemmBatchItem it = g.batch[bz];
        A += it.a_off; B += it.b_off; C += it.c_off;
        M = it.M; N = it.N; K = it.K;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    TileCoord tc = map_tile(g, tiles_m, tiles_n);
    if (!tc.valid) return;
    const int row0 = tc.ti * BM, col0 = tc.tj * BN;

    int kbeg = 0, kend = K;
    if (g.kmode == KM_A_LOWER) kend = min(K, row0 + BM);
    else if (g.kmode == KM_B_LOWER) kbeg = min(K, col0);
    else if (g.kmode == KM_AT_LOWER_B_LOWER) kbeg = min(K, max(row0, col0));
    else if (g.kmode == KM_B_UPPER) kend = min(K, col0 + BN);
    else if (g.kmode == KM_AT_LOWER) kbeg = min(K, row0);
    // split-K (grid.y = g.nsplit > 1): this workgroup takes a contiguous share of the tile's slabs and
    // stores its partial product (EPI_STORE) into slice blockIdx.y of g.split_buf; the reduce kernel
    // below adds the slices in a fixed order and applies the sign.  Shortens the critical path of
    // launches that have fewer tiles than the GPU has slots (V = L^-1 top levels, K^-1 = V^T V).
    if (g.nsplit > 1) {
        const int all = (kend - kbeg + BK - 1) / BK;
        const int per = (all + g.nsplit - 1) / g.nsplit;
        const int lo = min(all, (int)blockIdx.y * per), hi = min(all, lo + per);
        kend = min(kend, kbeg + hi * BK);
        kbeg = kbeg + lo * BK;
        C = bset(g.split_buf, tb, g.bstride) + (int64_t)blockIdx.y * g.split_stride + (C - C0);
    }
    const int nslab = (kend > kbeg) ? (kend - kbeg + BK - 1) / BK : 0;

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, gq = lane >> 4;

    // EPI_SUB: one chain that starts at -C (see gemm_dma_body.h)
    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
            acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
            if (EPI == EPI_SUB && g.nsplit <= 1) {
#pragma unroll
                for (int q = 0; q < 4; q++) {
                    const int row = row0 + wr * 64 + i * 16 + gq + 4 * q, col = col0 + wc * 64 + j * 16 + r;
                    if (row < M && col < N) acc[i][j][q] = -C[(int64_t)row * g.ldc + col];
                }
            }
        }

    double ra[8], rb[8];
    auto load_slab = [&](int s) {
        int k0 = kbeg + s * BK;
        if (AT) load_mc(A, g.lda, row0, M, k0, ra); else load_kc(A, g.lda, row0, M, k0, ra);
        if (BT) load_kc(B, g.ldb, col0, N, k0, rb); else load_mc(B, g.ldb, col0, N, k0, rb);
    };
    auto store_slab = [&](int buf) {
        double* As = smem + buf * 2 * TILE_DOUBLES;
        double* Bs = As + TILE_DOUBLES;
        if (AT) store_mc(As, ra); else store_kc(As, ra);
        if (BT) store_kc(Bs, rb); else store_mc(Bs, rb);
    };

    if (nslab > 0) {
        load_slab(0);
        store_slab(0);
    }
    __syncthreads();

    for (int s = 0; s < nslab; s++) {
        const int buf = s & 1;
        if (s + 1 < nslab) load_slab(s + 1);
        const double* As = smem + buf * 2 * TILE_DOUBLES;
        const double* Bs = As + TILE_DOUBLES;
#pragma unroll
        for (int kk = 0; kk < BK / 4; kk++) {
            double a[4], b[4];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
                a[mi] = AT ? As[(kk * 4 + gq) * SMC + wr * 64 + mi * 16 + r]
                           : As[(wr * 64 + mi * 16 + r) * SKC + kk * 4 + gq];
#pragma unroll
            for (int ni = 0; ni < 4; ni++)
                b[ni] = BT ? Bs[(wc * 64 + ni * 16 + r) * SKC + kk * 4 + gq]
                           : Bs[(kk * 4 + gq) * SMC + wc * 64 + ni * 16 + r];
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int ni = 0; ni < 4; ni++)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
        }
        if (s + 1 < nslab) store_slab(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue.  f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4*reg.
    if (EPI == EPI_SUMSQ) {
        // column sums of squares over the 128 rows of this tile -> C[ti*ldc + col]
        double cs[4];
#pragma unroll
        for (int ni = 0; ni < 4; ni++) {
            double s = 0.0;
#pragma unroll
            for (int mi = 0; mi < 4; mi++)
#pragma unroll
                for (int q = 0; q < 4; q++) s = fma(acc[mi][ni][q], acc[mi][ni][q], s);
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            cs[ni] = s;
        }
        double* red = smem;  // all slab reads are behind the loop's final barrier
        if (gq == 0) {
#pragma unroll
            for (int ni = 0; ni < 4; ni++) red[wr * 128 + wc * 64 + ni * 16 + r] = cs[ni];
        }
        __syncthreads();
        if (threadIdx.x < 128) {
            int col = col0 + threadIdx.x;
            if (col < N) C[(int64_t)tc.ti * g.ldc + col] = red[threadIdx.x] + red[128 + threadIdx.x];
        }
        return;
    }
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int row = row0 + wr * 64 + mi * 16 + gq + 4 * q;
                int col = col0 + wc * 64 + ni * 16 + r;
                if (row < M && col < N) {
                    double* p = C + (int64_t)row * g.ldc + col;
                    double v = acc[mi][ni][q];
                    if (EPI == EPI_STORE || g.nsplit > 1) *p = v;
                    else *p = -v;       // EPI_STORE_NEG; EPI_SUB: the chain ran on -C
                }
            }
}

// C = sign * sum over the split-K slices, same tile map / batch / lower_only logic as the product
// kernel (one workgroup per output tile, 256 threads, 16-byte accesses).
__global__ __launch_bounds__(256) void gemm_split_reduce_kernel(GemmArgs g, double sign) {
    int tb, bz;
    gemm_block_z(g, (int)blockIdx.z, &tb, &bz);
    if (g.info != nullptr && *bset(g.info, tb, g.bstride) != 0) return;
    double* C = bset(g.C, tb, g.bstride);
    const double* split_buf = bset(g.split_buf, tb, g.bstride);
    int M = g.M, N = g.N;
    int64_t coff = 0;
    if (g.batch != nullptr) {
        GemmBatchItem it = g.batch[bz];
        coff = it.c_off; M = it.M; N = it.N;
    }
    const int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    TileCoord tc = map_tile(g, tiles_m, tiles_n);
    if (!tc.valid) return;
    const int row0 = tc.ti * BM, col0 = tc.tj * BN;
    // grid.y = 8 row slices of 16 rows per tile: one workgroup per tile left the launch at ~1 TB/s
    // (a few hundred workgroups, each walking 32 dependent iterations)
    const int rbeg = (int)blockIdx.y * (BM / 8);
    for (int e = threadIdx.x; e < (BM / 8) * (BN / 2); e += 256) {
        const int rr = rbeg + e / (BN / 2), cc = (e % (BN / 2)) * 2;
        const int row = row0 + rr, col = col0 + cc;
        if (row >= M || col >= N) continue;
        const int64_t off = coff + (int64_t)row * g.ldc + col;
        double2 acc = make_double2(0.0, 0.0);
        for (int sidx = 0; sidx < g.nsplit; sidx++) {
            const double2 v = *reinterpret_cast<const double2*>(split_buf + (int64_t)sidx * g.split_stride + off);
            acc.x += v.x; acc.y += v.y;
        }
        *reinterpret_cast<double2*>(C + off) = make_double2(sign * acc.x, sign * acc.y);
    }
}

template <bool AT, bool BT>
static int launch_epi(gpry_ctx* ctx, const GemmArgs& g, int epi, dim3 grid) {
    hipStream_t st = g.stream ? g.stream : ctx->stream;
    switch (epi) {
        case EPI_STORE: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, EPI_STORE>), grid, dim3(256), 0, st, g); break;
        case EPI_STORE_NEG: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, EPI_STORE_NEG>), grid, dim3(256), 0, st, g); break;
        case EPI_SUB: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, EPI_SUB>), grid, dim3(256), 0, st, g); break;
        case EPI_SUMSQ: hipLaunchKernelGGL((gemm_f64_kernel<AT, BT, EPI_SUMSQ>), grid, dim3(256), 0, st, g); break;
        default: return gpry_fail(ctx, -1, "gemm: bad epilogue %d", epi);
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int gemm_f64_launch(gpry_ctx* ctx, const GemmArgs& g0, bool a_trans, bool b_trans, int epi) {
    GemmArgs g = g0;
    gemm_fill_batch(ctx, &g);
    int M = g.M, N = g.N;
    if (g.batch == nullptr && (M <= 0 || N <= 0)) return 0;
    int tiles_m = (M + BM - 1) / BM, tiles_n = (N + BN - 1) / BN;
    int64_t nblk;
    if ((g.tile_map & 15) == TM_SWEEP) {
        const int a = (g.tile_map >> 4) & 15, c = 6 - a;
        int64_t nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
        int64_t ns = (nsi * nsj + 7) / 8 * 8;
        nblk = ns * 64;
    } else {
        nblk = (int64_t)tiles_m * tiles_n;
    }
    // a handful of tiles: four times as many workgroups with 64 x 64 tiles (gemm_small.hip)
    if (g.small64 && ctx->opt_gemm_small > 0 && epi != EPI_SUMSQ && g.nsplit <= 1 && (g.tile_map & 15) == TM_ROWMAJOR &&
        !(a_trans && b_trans) && nblk * (g.batch ? g.n_batch : 1) <= ctx->opt_gemm_small)
        return gemm64_launch(ctx, g, a_trans, b_trans, epi);
    dim3 grid((unsigned)nblk, (unsigned)(g.nsplit > 1 ? g.nsplit : 1), (unsigned)(g.bz_div * g.bn));
    if (g.nsplit > 1 && !(epi == EPI_STORE || epi == EPI_STORE_NEG))
        return gpry_fail(ctx, -1, "gemm: split-K only with the store epilogues");
    int rc;
    bool dma = ctx->opt_gemm_dma && epi != EPI_SUMSQ && !(a_trans && b_trans);
    if (dma) dma = g.batch ? (g.dma_ok == 1 && gemm_dma_usable(g, BM, BN, 32)) : gemm_dma_usable(g, M, N, g.K);
    if (dma) rc = gemm_dma_launch_product(ctx, g, a_trans, b_trans, epi, grid);
    else if (!a_trans && !b_trans) rc = launch_epi<false, false>(ctx, g, epi, grid);
    else if (!a_trans && b_trans) rc = launch_epi<false, true>(ctx, g, epi, grid);
    else if (a_trans && !b_trans) rc = launch_epi<true, false>(ctx, g, epi, grid);
    else rc = launch_epi<true, true>(ctx, g, epi, grid);
    if (rc != 0 || g.nsplit <= 1 || g.skip_reduce) return rc;
    dim3 rgrid((unsigned)nblk, 8, (unsigned)(g.bz_div * g.bn));
    hipLaunchKernelGGL(gemm_split_reduce_kernel, rgrid, dim3(256), 0, g.stream ? g.stream : ctx->stream, g,
                       epi == EPI_STORE_NEG ? -1.0 : 1.0);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// scratch for split-K partial products: nsplit slices of `slice` doubles each
int gemm_split_scratch(gpry_ctx* ctx, int nsplit, int64_t slice, double** buf) {
    const int64_t need = (int64_t)nsplit * slice;
    // a batched evaluation works in the arena: its slices were sized up front and are never re-allocated
    if (need > ctx->split_cap && ctx->bpar) return gpry_fail(ctx, -1, "batched chain: split-K scratch of %lld doubles exceeds the %lld planned", (long long)need, (long long)ctx->split_cap);
    if (need > ctx->split_cap) {
        if (ctx->dsplit) GPRY_TRY(dev_free(ctx, ctx->dsplit));
        ctx->dsplit = nullptr; ctx->split_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dsplit, need));
        ctx->split_cap = need;
    }
    *buf = ctx->dsplit;
    return 0;
}
