// Sweep contraction of the NORA acquisition (gpry/gpr.py:1204 dtrmm + :1208 einsum, fused), per 128 x 128 tile:
//     S[ti][m] = sum_{i in tile ti} ( sum_{k <= i} V[i][k] * Kst[k][m] )^2
// FP64 MFMA (v_mfma_f64_16x16x4_f64), 256 threads = 4 waves (2 x 2), each wave 64 x 64 = 4 x 4 MFMA tiles; the N x M product
// is never stored (SUMSQ epilogue: per-tile column sums of squares, reduced registers -> shuffles -> LDS in a fixed order).
// Staging is direct global -> LDS (LDS-DMA, global_load_lds_dwordx4): no VGPR round trip, no ds_write pass -- each wave issues
// eight 1-KiB DMA pieces per slab of 16 k straight into the other LDS buffer.  The DMA destination is linear (wave base +
// lane * 16 B), so
//   * the K*^T slab keeps a padded [16][144] image (one piece = one 1-KiB k-row), and
//   * the V slab is an UNPADDED [128 rows][8 pieces of 16 B] image whose bank conflicts are removed by an XOR swizzle applied
//     on the SOURCE side: LDS piece p' of row r holds global piece p = p' ^ ((r >> 1) & 7); the fragment read applies the
//     same involution.
// Requires M, N multiples of 128 (true for the sweep: Np and the padded chunk).  The register-staged engine of gemm_f64.hip
// (option "gemm_dma" = 0) is the comparator; the measured history of the variants that lost (plain LDS-DMA without the software
// pipeline, a 128 x 256 / 8-wave ring, persistent workgroups with per-XCD tickets, k-skewed and staggered starts) is in
// profiles/HISTORY.md.
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

// ---------------------------------------------------------------------------------------
// The 128 x 128 / 4-wave tile with an explicit software pipeline.
// The compiler treats a pending LDS-DMA as a "flat" access and therefore turns every wait on
// a fragment read into s_waitcnt lgkmcnt(0), placed right behind the newest ds_read: the
// MFMA pipe drains while the wave sits out the LDS latency, once per k-step.  Here the
// fragment reads are inline asm (invisible to the waitcnt pass) with hand-placed counters:
//   * fragments are double-buffered in registers; the eight LDS reads of k-step kk+1 are
//     issued BEFORE the 16 MFMAs of k-step kk and waited for with lgkmcnt(8) semantics
//     (only the older fragment set must have arrived);
//   * the slab barrier sits before the LAST k-step of a slab: behind it the wave first
//     requests the fragments of the next slab's k-step 0 and the DMA of slab s+2 (into the
//     buffer everybody has just finished reading), then multiplies k-step 3 -- neither the
//     LDS latency after a barrier nor the DMA issue sits in front of an idle MFMA pipe, and
//     a DMA piece has 1.25 slab times to land instead of 1.
struct Frag { double a0, a1, a2, a3, b0, b1, b2, b3; };

#define BUF_BYTES (BUF_DOUBLES * 8)        // 0x8800
#define B_BASE_BYTES (A_DOUBLES * 8)       // 0x4000
#define B_KSTEP_BYTES (4 * SMC * 8)        // 4 k-rows
#define A_MI_BYTES (16 * 16 * 8)           // 16 rows of the unpadded V image

// Eight ds_read_b64 per k-step.  ds_read_b64 banks on (addr/4) mod 64 in 32-lane groups, for
// which both images are conflict-free; the ds_read2(st64)_b64 the compiler would merge the V
// reads into banks on mod 32 in 16-lane groups and is 2-way conflicted on the swizzled image
// (rows r, r+1 share a 16-B piece; SQ_LDS_BANK_CONFLICT = 40 % of the LDS cycles).
template <int BUF, int KK>
__device__ __forceinline__ void load_frag(Frag& f, unsigned addrA, unsigned addrB) {
    constexpr int oa = BUF * BUF_BYTES;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a0) : "v"(addrA), "n"(oa) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a1) : "v"(addrA), "n"(oa + A_MI_BYTES) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a2) : "v"(addrA), "n"(oa + 2 * A_MI_BYTES) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.a3) : "v"(addrA), "n"(oa + 3 * A_MI_BYTES) : "memory");
    constexpr int ob = B_BASE_BYTES + BUF * BUF_BYTES + KK * B_KSTEP_BYTES;
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b0) : "v"(addrB), "n"(ob) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b1) : "v"(addrB), "n"(ob + 128) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b2) : "v"(addrB), "n"(ob + 256) : "memory");
    asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(f.b3) : "v"(addrB), "n"(ob + 384) : "memory");
}

// wait until at most NPEND LDS reads are outstanding; the fragment is threaded through the
// statement so that its consumers cannot be moved in front of the wait
template <int NPEND>
__device__ __forceinline__ void wait_frag(Frag& f) {
    asm volatile("s_waitcnt lgkmcnt(%8)"
                 : "+v"(f.a0), "+v"(f.a1), "+v"(f.a2), "+v"(f.a3), "+v"(f.b0), "+v"(f.b1), "+v"(f.b2), "+v"(f.b3)
                 : "n"(NPEND) : "memory");
}

__device__ __forceinline__ void mma_frag(v4d (&acc)[4][4], const Frag& f) {
    const double a[4] = {f.a0, f.a1, f.a2, f.a3};
    const double b[4] = {f.b0, f.b1, f.b2, f.b3};
#pragma unroll
    for (int mi = 0; mi < 4; mi++)
#pragma unroll
        for (int ni = 0; ni < 4; ni++)
            acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
}

__device__ __forceinline__ void mma_row(v4d (&acc)[4][4], const Frag& f, int mi) {
    const double a = mi == 0 ? f.a0 : mi == 1 ? f.a1 : mi == 2 ? f.a2 : f.a3;
    acc[mi][0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b0, acc[mi][0], 0, 0, 0);
    acc[mi][1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b1, acc[mi][1], 0, 0, 0);
    acc[mi][2] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b2, acc[mi][2], 0, 0, 0);
    acc[mi][3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, f.b3, acc[mi][3], 0, 0, 0);
}

// One workgroup per tile.  XCD-aware block map: blocks b, b + 8, ... share an XCD (round-robin dispatch), 64 consecutive
// slots of an XCD form one (2^ta x 64 / 2^ta) super-tile (ta = bits 4..7 of g.tile_map: 8 row tiles of V x 8 candidate tiles)
// so that the V row panels and K*^T column panels of a super-tile are served by that XCD's L2; super-tiles are ordered by
// descending row index (longest k first).
__global__ __launch_bounds__(256, 2) void sweep_gemm_dma_sp_kernel(GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double smem[2 * BUF_DOUBLES];
    const int M = g.M, N = g.N, K = g.K;
    const int tiles_m = M / BM, tiles_n = N / BN;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int r = lane & 15, gq = lane >> 4;

    const int vx = blockIdx.x & 7;                       // XCD
    const int ta = (g.tile_map >> 4) & 15, tc = 6 - ta;
    const int nsi = (tiles_m + (1 << ta) - 1) >> ta, nsj = (tiles_n + (1 << tc) - 1) >> tc;
    const int q = blockIdx.x >> 3;                       // position in the XCD's queue
    int ti, tj;
    bool valid, desc;
    {
        const int s = (q >> 6) * 8 + vx, within = q & 63;
        int si = nsi - 1 - s / nsj;
        int sj = s % nsj;
        int rowin = within >> tc;
        // Alternating k walk.  The 64 tiles of a super-tile fill the 64 slots of an XCD, and the
        // eight tiles of a row tile run in lockstep (equal length), so the V panel is fetched once per XCD; the K*^T
        // panel is shared across the eight ROW tiles only while these sit at the same k, and row tiles differ by
        // eight slabs in length.  A super-tile whose tiles start together and walk k upwards is aligned and frees its
        // slots row by row, shortest first, eight slab times apart; if the next super-tile of the XCD hands those
        // slots to its rows LONGEST first and walks k DOWNWARDS, its rows meet at the same k again (start skew =
        // length difference) and finish together, and the one after that starts aligned.  So the super-tiles of an
        // XCD alternate between two adjacent super-rows: odd positions walk down with their rows reversed.  The
        // direction is a function of the row tile alone, so a candidate's bits do not depend on where its
        // column falls in a launch (sharded = unsharded).
        const bool alt = true;
        if (!(nsi & 1) && (nsj & 7) == 0) {
            const int n = q >> 6, per_pair = nsj >> 2;        // 2 super-rows x nsj/8 super-tiles per XCD
            const int pair = n / per_pair, n2 = n - pair * per_pair, odd = n2 & 1;
            si = nsi - 1 - 2 * pair - odd;
            sj = (n2 >> 1) * 8 + vx;
            if (odd) rowin = (1 << ta) - 1 - rowin;
        }
        desc = alt && ((nsi - 1 - si) & 1);
        ti = (si << ta) + rowin;
        tj = (sj << tc) + (within & ((1 << tc) - 1));
        valid = s < nsi * nsj && si >= 0 && ti < tiles_m && tj < tiles_n;
        if (nsi == 1) {
            // A model of up to 1024 rows is ONE super-row whose row tiles differ by up to 8x in length: every super-tile mixes
            // them, and the launch ends with long tiles running beside idle slots (a k = 1024 tile takes ~ 200 us of a 1.9-ms
            // launch at M = 1e5).  Longest row tile first over the WHOLE launch instead: the tail is made of the shortest tiles;
            // consecutive candidate tiles go to different XCDs, each of which keeps the one V row panel in its L2.
            const int row = (int)blockIdx.x / tiles_n;
            ti = tiles_m - 1 - row;
            tj = (int)blockIdx.x - row * tiles_n;
            valid = row < tiles_m;
            desc = false;           // (as above for si = 0: the walk direction is a function of the row tile alone)
        }
    }
    if (valid) {
    const int row0 = ti * BM, col0 = tj * BN;
    const int kend = min(K, row0 + BM);   // V is lower triangular
    const int nslab = kend / BK;          // multiple of 8
    const int kfirst = desc ? (nslab - 1) * BK : 0, kstep = desc ? -BK : BK;

    const double* srcA[4];
    int dstA[4];
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int q = wave * 4 + j;
        const int row = q * 8 + (lane >> 3), pp = lane & 7;
        const int p = pp ^ ((row >> 1) & 7);
        srcA[j] = g.A + (int64_t)(row0 + row) * g.lda + 2 * p;
        dstA[j] = q * 128;
    }
    const double* srcB = g.B + (int64_t)(wave * 4) * g.ldb + col0 + 2 * lane;

    auto issue = [&](int s, int buf) {
        double* As = smem + buf * BUF_DOUBLES;
        double* Bs = As + A_DOUBLES;
        const int k0 = kfirst + s * kstep;
#pragma unroll
        for (int j = 0; j < 4; j++) dma16(srcA[j] + k0, As + dstA[j]);
#pragma unroll
        for (int j = 0; j < 4; j++) dma16(srcB + (int64_t)(k0 + j) * g.ldb, Bs + (wave * 4 + j) * SMC);
    };
    // one eighth of a slab's DMA (piece j of V for j < 4, piece j-4 of K*^T otherwise)
    auto issue_piece = [&](int s, int buf, int j) {
        double* As = smem + buf * BUF_DOUBLES;
        double* Bs = As + A_DOUBLES;
        const int k0 = kfirst + s * kstep;
        if (j < 4) dma16(srcA[j] + k0, As + dstA[j]);
        else dma16(srcB + (int64_t)(k0 + j - 4) * g.ldb, Bs + (wave * 4 + j - 4) * SMC);
    };

    v4d acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = (v4d){0.0, 0.0, 0.0, 0.0};

    // LDS byte addresses of this lane's fragments (buffer and k-step of B are immediates).
    // A image: row (wr*64 + mi*16 + r) at 128 B per row, 16-B piece ((k>>1) ^ key), key =
    // (row>>1)&7 = (r>>1)&7 for every mi; k = kk*4 + gq  =>  piece = ((gq>>1) ^ key) ^ (2*kk).
    const unsigned smem_base = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned addrA[4];
#pragma unroll
    for (int kk = 0; kk < 4; kk++)
        addrA[kk] = smem_base + (wr * 64 + r) * 128 + 16 * ((((gq >> 1) ^ (r >> 1)) & 7) ^ (2 * kk)) + 8 * (gq & 1);
    const unsigned addrB = smem_base + gq * (SMC * 8) + (wc * 64 + r) * 8;

    issue(0, 0);
    issue(1, 1);                                             // nslab >= 8
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // slab 0 landed, slab 1 in flight
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);

    Frag f0, f1;
    load_frag<0, 0>(f0, addrA[0], addrB);

#define SLAB_STEP(BUF, S)                                                                    \
    {                                                                                        \
        load_frag<BUF, 1>(f1, addrA[1], addrB);                                              \
        wait_frag<8>(f0);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        mma_frag(acc, f0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        load_frag<BUF, 2>(f0, addrA[2], addrB);                                              \
        wait_frag<8>(f1);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        mma_frag(acc, f1);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        load_frag<BUF, 3>(f1, addrA[3], addrB);                                              \
        wait_frag<8>(f0);                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        mma_frag(acc, f0);                                                                   \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        /* every wave holds its last fragments of slab S; slab S+1 has landed everywhere */  \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                     \
        wait_frag<0>(f1);                                                                    \
        __builtin_amdgcn_s_barrier();                                                        \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        /* branch-free tail: the last slabs re-fetch slab nslab-1 into the (dead) buffer and  \
           read fragments nobody uses, so that the DMA issue and the fragment reads of the    \
           next slab can sit in the shadow of k-step 3's MFMAs */                              \
        const int s2 = min((S) + 2, nslab - 1);                                              \
        mma_row(acc, f1, 0);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        load_frag<(BUF) ^ 1, 0>(f0, addrA[0], addrB);                                        \
        issue_piece(s2, BUF, 0); issue_piece(s2, BUF, 1);                                    \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        mma_row(acc, f1, 1);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        issue_piece(s2, BUF, 2); issue_piece(s2, BUF, 3); issue_piece(s2, BUF, 4);           \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        mma_row(acc, f1, 2);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        issue_piece(s2, BUF, 5); issue_piece(s2, BUF, 6); issue_piece(s2, BUF, 7);           \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        mma_row(acc, f1, 3);                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                   \
    }

    for (int s = 0; s < nslab; s += 2) {
        SLAB_STEP(0, s)
        SLAB_STEP(1, s + 1)
    }
#undef SLAB_STEP

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
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // trailing DMA / fragment reads
    __syncthreads();
    double* red = smem;
    if (gq == 0) {
#pragma unroll
        for (int ni = 0; ni < 4; ni++) red[wr * 128 + wc * 64 + ni * 16 + r] = cs[ni];
    }
    __syncthreads();
    if (threadIdx.x < 128)
        g.C[(int64_t)ti * g.ldc + col0 + threadIdx.x] = red[threadIdx.x] + red[128 + threadIdx.x];
    }  // valid
}

int sweep_gemm_dma_sp_launch(gpry_ctx* ctx, const GemmArgs& g) {
    if (g.M % BM || g.N % BN || g.K % BM) return gpry_fail(ctx, -1, "sweep_gemm_dma_sp: dims must be multiples of 128");
    const int tiles_m = g.M / BM, tiles_n = g.N / BN;
    const int a = (g.tile_map >> 4) & 15, c = 6 - a;
    const int64_t nsi = (tiles_m + (1 << a) - 1) >> a, nsj = (tiles_n + (1 << c) - 1) >> c;
    const dim3 grid(nsi == 1 ? (unsigned)(tiles_m * tiles_n) : (unsigned)(((nsi * nsj + 7) / 8 * 8) * 64));
    hipLaunchKernelGGL(sweep_gemm_dma_sp_kernel, grid, dim3(256), 0, ctx->stream, g);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
