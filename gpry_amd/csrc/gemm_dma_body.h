// Device-side body of the LDS-DMA pipelined FP64 MFMA GEMM (see gemm_dma.hip for the design notes): shared
// between gemm_dma_kernel and the fused Cholesky step (chol_panel.hip), which runs trailing-update tiles in
// the same launch as a panel step.
#pragma once
#include "common.h"

#define BM 128
#define BN 128
#define BK 16
#define SMC 144
#define KC_DOUBLES 2048   // 128 x 16, unpadded
#define MC_DOUBLES 2304   // 16 x 144
#define KC_MI_BYTES (16 * 16 * 8)       // 16 rows of a KC image
#define MC_KSTEP_BYTES (4 * SMC * 8)    // 4 k-rows of an MC image

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

__device__ __forceinline__ void gd_dma16(const double* g, double* l) {
    __builtin_amdgcn_global_load_lds((gptr_t)g, (lptr_t)l, 16, 0, 0);
}

struct GdFrag { double a0, a1, a2, a3, b0, b1, b2, b3; };

// Eight ds_read_b64 of k-step KK from buffer BUF.  KC operands: the address register already holds the
// (swizzled) k-step, the four 16-row blocks are immediates; MC operands: one address register, k-step
// and blocks are immediates.  Inline asm: the compiler's waitcnt pass would treat the pending LDS-DMA
// as a flat access and drain the counter (lgkmcnt(0)) in front of every use.
template <bool AKC, bool BKC, int BUF, int KK>
__device__ __forceinline__ void gd_load_frag(GdFrag& f, unsigned addrA, unsigned addrB) {
    constexpr int a_sz = (AKC ? KC_DOUBLES : MC_DOUBLES) * 8, b_sz = (BKC ? KC_DOUBLES : MC_DOUBLES) * 8;
    // the DS offset field has 16 bits: with two MC images the second buffer's B fragments lie beyond
    // it, so there the buffer offset goes into the address register (one v_add per operand and k-step)
    constexpr int buf_all = BUF * (a_sz + b_sz);
    constexpr int top_a = buf_all + (AKC ? 3 * KC_MI_BYTES : 3 * MC_KSTEP_BYTES + 3 * 128);
    constexpr int top_b = buf_all + a_sz + (BKC ? 3 * KC_MI_BYTES : 3 * MC_KSTEP_BYTES + 3 * 128);
    constexpr bool far = top_a > 65535 || top_b > 65535;
    constexpr int buf = far ? 0 : buf_all;
    if (far) { addrA += buf_all; addrB += buf_all; }
    constexpr int oa = buf + (AKC ? 0 : KK * MC_KSTEP_BYTES), sa = AKC ? KC_MI_BYTES : 128;
    constexpr int ob = buf + a_sz + (BKC ? 0 : KK * MC_KSTEP_BYTES), sb = BKC ? KC_MI_BYTES : 128;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a0) : "v"(addrA), "n"(oa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a1) : "v"(addrA), "n"(oa + sa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a2) : "v"(addrA), "n"(oa + 2 * sa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a3) : "v"(addrA), "n"(oa + 3 * sa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b0) : "v"(addrB), "n"(ob) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b1) : "v"(addrB), "n"(ob + sb) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b2) : "v"(addrB), "n"(ob + 2 * sb) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b3) : "v"(addrB), "n"(ob + 3 * sb) : "memory");
}
template <int NPEND>
__device__ __forceinline__ void gd_wait_frag(GdFrag& f) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(f.a0), "+v"(f.a1), "+v"(f.a2), "+v"(f.a3), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2), "+v"(f.b3)
                 : "n"(NPEND) : "memory");
}
__device__ __forceinline__ void gd_mma_frag(v4d (&acc)[4][4], const GdFrag& f) {
    const double a[4] = {f.a0, f.a1, f.a2, f.a3};
    const double b[4] = {f.b0, f.b1, f.b2, f.b3};
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
}
__device__ __forceinline__ void gd_mma_row(v4d (&acc)[4][4], const GdFrag& f, int mi) {
    const double a = mi == 0 ? f.a0 : mi == 1 ? f.a1 : mi == 2 ? f.a2 : f.a3;
    acc[mi][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b0, acc[mi][0], 0, 0, 0);
    acc[mi][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b1, acc[mi][1], 0, 0, 0);
    acc[mi][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b2, acc[mi][2], 0, 0, 0);
    acc[mi][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b3, acc[mi][3], 0, 0, 0);
}

// AT: A(i,k) stored at A[k*lda + i] (an MC operand); BT: B(k,j) stored at B[j*ldb + k] (a KC operand)
// One output tile (or split-K share of it) of the product; `smem` = 2 x (A image + B image) doubles of LDS,
// bx / by / bz = the block coordinates the tile map, the split-K share and the batch item are taken from
// (the kernel passes its own block index; the fused Cholesky step of chol_panel.hip passes a tile number).
// `part` (nullable): stream-K launch (gemm_dma_parts_kernel) -- the tile, the share of its k-range (in slab pairs,
// relative to the tile's own range) and the slice of g.split_buf the partial product goes to (slice < 0: the whole
// range, final epilogue straight into C) come from the part instead of from the block coordinates.
// `tb`: theta index of a batched launch (gpry_ctx::bn; 0 otherwise), bz_: the batch item
template <bool AT, bool BT, int EPI>
__device__ __forceinline__ void gemm_dma_tile_body(const GemmArgs& g, double* smem, const int bx, const int by, const int bz_,
                                                   const GemmPart* part = nullptr, const int tb = 0) {
    const int bz = part ? part->bz : bz_;
    constexpr bool AKC = !AT, BKC = BT;
    constexpr int A_SZ = AKC ? KC_DOUBLES : MC_DOUBLES, B_SZ = BKC ? KC_DOUBLES : MC_DOUBLES;
    constexpr int BUF_SZ = A_SZ + B_SZ;
    if (g.info != nullptr && *bset(g.info, tb, g.bstride) != 0) return;

    double* const C0 = bset(g.C, tb, g.bstride);
    double* const split0 = bset(g.split_buf, tb, g.bstride);
    const double* A = bset(g.A, tb, g.bstride); const double* B = bset(g.B, tb, g.bstride); double* C = C0;
    int M = g.M, N = g.N, K = g.K;
    if (g.batch != nullptr) {
        GemmBatchItem it = g.batch[bz];
        A += it.a_off; B += it.b_off; C += it.c_off;
        M = it.M; N = it.N; K = it.K;
    }
    const int tiles_m = M / BM, tiles_n = N / BN;
    int ti, tj;
    if (part) { ti = part->ti; tj = part->tj; }
    else {
        const int b = bx;
        if ((g.tile_map & 15) == TM_SWEEP) {
            const int a = (g.tile_map >> 4) & 15, c = 6 - a;
            const int nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
            const int xcd = b & 7, q = b >> 3;
            const int s = (q >> 6) * 8 + xcd, within = q & 63;
            if (s >= nsi * nsj) return;
            const int si = nsi - 1 - s / nsj, sj = s % nsj;
            ti = (si << a) + (within >> c);
            tj = (sj << c) + (within & ((1 << c) - 1));
        } else if ((g.tile_map & 15) == TM_BALANCED) {
            // With the row-major map XCD = tj mod 8 (tiles_n is a multiple of 8 at the sizes that matter):
            // in K^-1 = V^T V (k-length (tiles - ti) * 8 slabs, lower tiles only) XCD 0 then holds twice the
            // work of XCD 7 and, worse, its longest tiles in pairs on the same CUs.
            if (g.lower_only && tiles_m == tiles_n) {
                int i = (int)((sqrt(8.0 * (double)b + 1.0) - 1.0) * 0.5);       // b = i (i + 1) / 2 + j, j <= i
                while ((i + 1) * (i + 2) / 2 <= b) i++;
                while (i * (i + 1) / 2 > b) i--;
                ti = i; tj = b - i * (i + 1) / 2;
                if (ti >= tiles_m) return;
            } else if (g.kmode == KM_B_LOWER || g.kmode == KM_B_UPPER) {
                tj = b / tiles_m;
                ti = b - tj * tiles_m;
                if (g.kmode == KM_B_UPPER) tj = tiles_n - 1 - tj;
            } else {
                ti = b / tiles_n;
                tj = b - ti * tiles_n;
                if (g.kmode == KM_A_LOWER) ti = tiles_m - 1 - ti;
            }
        } else {
            ti = b / tiles_n;
            tj = b - ti * tiles_n;
        }
        if (ti < 0 || tj < 0 || ti >= tiles_m || tj >= tiles_n) return;
        if (g.lower_only && tj > ti) return;
    }
    const int row0 = ti * BM, col0 = tj * BN;
    int kbeg = 0, kend = K;
    if (g.kmode == KM_A_LOWER) kend = min(K, row0 + BM);
    else if (g.kmode == KM_B_LOWER) kbeg = min(K, col0);
    else if (g.kmode == KM_AT_LOWER_B_LOWER) kbeg = min(K, max(row0, col0));
    else if (g.kmode == KM_B_UPPER) kend = min(K, col0 + BN);
    else if (g.kmode == KM_AT_LOWER) kbeg = min(K, row0);
    bool partial = g.nsplit > 1;
    if (part) {
        kend = kbeg + part->hi * 2 * BK;
        kbeg = kbeg + part->lo * 2 * BK;
        partial = part->slice >= 0;
        if (partial) C = split0 + (int64_t)part->slice * g.split_stride + (C - C0);
    } else if (g.nsplit > 1) {      // split-K over grid.y, in units of slab pairs
        const int all = (kend - kbeg) / (2 * BK);
        const int per = (all + g.nsplit - 1) / g.nsplit;
        const int lo = min(all, by * per), hi = min(all, lo + per);
        kend = kbeg + hi * 2 * BK;
        kbeg = kbeg + lo * 2 * BK;
        C = split0 + (int64_t)by * g.split_stride + (C - C0);
    }
    const int nslab = (kend - kbeg) / BK;      // even; 0 only for an empty split-K share

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, gq = lane >> 4;

    // EPI_SUB (the trailing updates of the Cholesky): C - sum_k a b as ONE chain that starts at C -- the accumulators take -C,
    // the products are added, the result is stored negated (round-to-nearest is symmetric in sign: the bits of C - a b - ...).
    // A chain cut at a multiple of 4 k and resumed from the stored tile is the same chain, so every schedule of the
    // factorisation -- riding 64 x 64 tiles, column blocks with one launch behind each, left-looking panel steps -- gives the
    // same factor bit for bit (chol_panel.hip).  The old values go out first: loads return in order, so the wait for the
    // first slab of the DMA below covers them.
    double* const cbase = C + (int64_t)(row0 + wr * 64 + gq) * g.ldc + col0 + wc * 64 + r;
    v4d acc[4][4];
    if (EPI == EPI_SUB && !partial) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++)
#pragma unroll
                for (int q = 0; q < 4; q++) acc[i][j][q] = -cbase[(int64_t)(i * 16 + 4 * q) * g.ldc + j * 16];
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int j = 0; j < 4; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};
    }

    if (nslab > 0) {
    // per-lane source addresses of this wave's 4 + 4 DMA pieces of a slab
    const double* srcA[4];
    const double* srcB[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = wave * 4 + j;                       // KC piece: rows q*8 .. q*8+7; MC piece: k-row q
        const int row = q * 8 + (lane >> 3), pp = lane & 7;
        const int p = pp ^ ((row >> 1) & 7);              // swizzle on the source side
        srcA[j] = AKC ? A + (int64_t)(row0 + row) * g.lda + kbeg + 2 * p
                      : A + (int64_t)(kbeg + q) * g.lda + row0 + 2 * lane;
        srcB[j] = BKC ? B + (int64_t)(col0 + row) * g.ldb + kbeg + 2 * p
                      : B + (int64_t)(kbeg + q) * g.ldb + col0 + 2 * lane;
    }
    const int64_t stepA = AKC ? BK : (int64_t)BK * g.lda, stepB = BKC ? BK : (int64_t)BK * g.ldb;
    // one eighth of a slab's DMA (piece j of A for j < 4, piece j-4 of B otherwise)
    auto issue_piece = [&](int s, int buf, int j) {
        double* As = smem + buf * BUF_SZ;
        double* Bs = As + A_SZ;
        if (j < 4) gd_dma16(srcA[j] + s * stepA, As + (AKC ? (wave * 4 + j) * 128 : (wave * 4 + j) * SMC));
        else gd_dma16(srcB[j - 4] + s * stepB, Bs + (BKC ? (wave * 4 + j - 4) * 128 : (wave * 4 + j - 4) * SMC));
    };
    auto issue = [&](int s, int buf) {
#pragma unroll
        for (int j = 0; j < 8; j++) issue_piece(s, buf, j);
    };

    // LDS byte addresses of this lane's fragments.  KC image: row (w*64 + blk*16 + r) at 128 B per row,
    // 16-B piece ((k>>1) ^ key), key = (r>>1)&7 for every block; k = kk*4 + gq  =>  piece =
    // ((gq>>1) ^ key) ^ (2*kk).  MC image: k-row (kk*4 + gq) at SMC doubles, column w*64 + blk*16 + r.
    const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned addrA[4], addrB[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++) {
        addrA[kk] = AKC ? smem_base + (wr * 64 + r) * 128 + 16 * ((((gq >> 1) ^ (r >> 1)) & 7) ^ (2 * kk)) + 8 * (gq & 1)
                        : smem_base + gq * (SMC * 8) + (wr * 64 + r) * 8;
        addrB[kk] = BKC ? smem_base + (wc * 64 + r) * 128 + 16 * ((((gq >> 1) ^ (r >> 1)) & 7) ^ (2 * kk)) + 8 * (gq & 1)
                        : smem_base + gq * (SMC * 8) + (wc * 64 + r) * 8;
    }

    issue(0, 0);
    issue(1, 1);                                             // nslab >= 2
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // slab 0 landed, slab 1 in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    GdFrag f0, f1;
    gd_load_frag<AKC, BKC, 0, 0>(f0, addrA[0], addrB[0]);

#define GD_SLAB_STEP(BUF, S)                                                                 \
    {                                                                                        \
        gd_load_frag<AKC, BKC, BUF, 1>(f1, addrA[1], addrB[1]);                              \
        gd_wait_frag<8>(f0);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_mma_frag(acc, f0);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_load_frag<AKC, BKC, BUF, 2>(f0, addrA[2], addrB[2]);                              \
        gd_wait_frag<8>(f1);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_mma_frag(acc, f1);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_load_frag<AKC, BKC, BUF, 3>(f1, addrA[3], addrB[3]);                              \
        gd_wait_frag<8>(f0);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_mma_frag(acc, f0);                                                                \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        /* every wave holds its last fragments of slab S; slab S+1 has landed everywhere */  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
        gd_wait_frag<0>(f1);                                                                 \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        /* branch-free tail: the last slabs re-fetch slab nslab-1 into the (dead) buffer */  \
        const int s2 = min((S) + 2, nslab - 1);                                              \
        gd_mma_row(acc, f1, 0);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_load_frag<AKC, BKC, (BUF) ^ 1, 0>(f0, addrA[0], addrB[0]);                        \
        issue_piece(s2, BUF, 0); issue_piece(s2, BUF, 1);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_mma_row(acc, f1, 1);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        issue_piece(s2, BUF, 2); issue_piece(s2, BUF, 3); issue_piece(s2, BUF, 4);           \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_mma_row(acc, f1, 2);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        issue_piece(s2, BUF, 5); issue_piece(s2, BUF, 6); issue_piece(s2, BUF, 7);           \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        gd_mma_row(acc, f1, 3);                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    }
    for (int s = 0; s < nslab; s += 2) {
        GD_SLAB_STEP(0, s)
        GD_SLAB_STEP(1, s + 1)
    }
#undef GD_SLAB_STEP
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // trailing DMA / fragment reads
    }  // nslab > 0

    // ---- epilogue.  f64 16x16x4 C/D layout: col = lane & 15, row = (lane >> 4) + 4*reg.
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double* p = cbase + (int64_t)(mi * 16 + 4 * q) * g.ldc + ni * 16;
                const double v = acc[mi][ni][q];
                if (EPI == EPI_STORE || partial) *p = v;
                else *p = -v;           // EPI_STORE_NEG; EPI_SUB: the chain ran on -C
            }
}
