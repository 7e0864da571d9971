// Fused Cholesky panel step for gfx950: ONE launch per 64 columns does, for every 64-row
// block of the panel (workgroup = block, 4 waves):
//     B   <- A[rows, j0:j0+64] - A[rows, K0:j0] * A[j0:j0+64, K0:j0]^T     (left-looking update)
//     Lkk <- chol(D)   with D the same update of the diagonal block (every workgroup
//                       recomputes it: 64x64, cheaper than a second launch + dependency)
//     X   <- B * Lkk^-T                                                     (dtrsm R,L,T,N)
// Inside the workgroup everything is blocked by 16: the 16x16 diagonal factors run in a
// single wave and the 16-wide triangular solves are lane-per-row substitutions, both with DPP
// row broadcasts as multiply-add operands (same operation order as LAPACK dpotf2/dtrsm: scale
// by the reciprocal pivot); all rank-16 updates are v_mfma_f64_16x16x4_f64; the four waves run
// the 64x64 factor as a dataflow on LDS flags.  LDS images use a row stride of 66 doubles:
// MFMA fragment reads (16 rows x {k, k+1}) then hit 32 distinct bank pairs.
#include "common.h"
#include "gemm_dma_body.h"
#include <algorithm>

#define PLD 66

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// acc += sign * A[16 x K] * B[16 x K]^T ; A, B point at row 0 of their 16-row blocks in LDS.  K = 16, 32, 48 or 64: all
// fragment reads of the product are issued before the first MFMA (a load / wait / MFMA loop exposed the LDS latency in
// every step: 2.3-2.8k cycles per 16-MFMA tile in the panel step instead of ~1.1k); same MFMA sequence, k ascending.
template <bool NEG, int KQ>
__device__ __forceinline__ v4d mfma_nt16_all(v4d acc, const double* A, const double* B, int lane) {
    const int r = lane & 15, g = lane >> 4;
    double a[KQ], b[KQ];
#pragma unroll
    for (int u = 0; u < KQ; u++) { a[u] = A[r * PLD + 4 * u + g]; b[u] = B[r * PLD + 4 * u + g]; }
    // (without this the scheduler interleaves reads and MFMAs with a full s_waitcnt lgkmcnt(0) every other MFMA)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < KQ; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a[u] : a[u], b[u], acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    return acc;
}
template <bool NEG>
__device__ __forceinline__ v4d mfma_nt16(v4d acc, const double* A, const double* B, int K, int lane) {
    switch (K) {
        case 16: return mfma_nt16_all<NEG, 4>(acc, A, B, lane);
        case 32: return mfma_nt16_all<NEG, 8>(acc, A, B, lane);
        case 48: return mfma_nt16_all<NEG, 12>(acc, A, B, lane);
        case 64: return mfma_nt16_all<NEG, 16>(acc, A, B, lane);
        default: break;
    }
    const int r = lane & 15, g = lane >> 4;
    for (int k0 = 0; k0 < K; k0 += 4) {
        double a = A[r * PLD + k0 + g];
        double b = B[r * PLD + k0 + g];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(NEG ? -a : a, b, acc, 0, 0, 0);
    }
    return acc;
}
// C/D fragment (row = g + 4q, col = r) <-> LDS tile at T (row 0, col 0 of the tile)
__device__ __forceinline__ v4d tile_load(const double* T, int lane) {
    const int r = lane & 15, g = lane >> 4;
    v4d v;
#pragma unroll
    for (int q = 0; q < 4; q++) v[q] = T[(g + 4 * q) * PLD + r];
    return v;
}
__device__ __forceinline__ void tile_store(double* T, v4d v, int lane) {
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; q++) T[(g + 4 * q) * PLD + r] = v[q];
}

// broadcast of a double from a lane known at compile time: two v_readlane_b32 (a few cycles)
// instead of a ds_bpermute round trip through the LDS crossbar
__device__ __forceinline__ double readlane_f64(double v, int src_lane) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_readlane(lo, src_lane);
    hi = __builtin_amdgcn_readlane(hi, src_lane);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(x) for a positive, normal x: v_rsq_f64 (~26 good bits) + ONE third-order step
//   e = 1 - x r^2,  r <- r + r e (1/2 + 3/8 e)          (error O(e^3): below the rounding of a double)
// = four dependent FP64 operations on the 16-step pivot chain instead of the six of two Newton steps
// (libm's rsqrt is a sqrt followed by a division -- ~250 cycles).
__device__ __forceinline__ double pivot_rsqrt(double x) {
    const double r = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * r, r, 1.0);
    const double p = fma(0.375, e, 0.5);
    const double q = r * e;
    return fma(q, p, r);
}

// acc += a[lane c of this lane's 16-lane row] * b   (v_fmac_f64 with a DPP row_newbcast source; c is a
// constant after unrolling, the switch folds)
#define FMAC_BCAST_CASE(C) case C: asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                                : "+v"(acc) : "v"(a), "v"(b)); break;
__device__ __forceinline__ void fmac_row_bcast(double& acc, double a, double b, int c) {
    switch (c) {
        FMAC_BCAST_CASE(0) FMAC_BCAST_CASE(1) FMAC_BCAST_CASE(2) FMAC_BCAST_CASE(3) FMAC_BCAST_CASE(4) FMAC_BCAST_CASE(5)
        FMAC_BCAST_CASE(6) FMAC_BCAST_CASE(7) FMAC_BCAST_CASE(8) FMAC_BCAST_CASE(9) FMAC_BCAST_CASE(10)
        FMAC_BCAST_CASE(11) FMAC_BCAST_CASE(12) FMAC_BCAST_CASE(13) FMAC_BCAST_CASE(14) FMAC_BCAST_CASE(15)
        default: break;
    }
}
#undef FMAC_BCAST_CASE
// acc -= a[lane c of this lane's 16-lane row] * b   (negation as the source modifier of the DPP operand)
#define FMSUB_BCAST_CASE(C) case C: asm volatile("v_fmac_f64_dpp %0, -%1, %2 row_newbcast:" #C " row_mask:0xf bank_mask:0xf" \
                                                 : "+v"(acc) : "v"(a), "v"(b)); break;
__device__ __forceinline__ void fmsub_row_bcast(double& acc, double a, double b, int c) {
    switch (c) {
        FMSUB_BCAST_CASE(0) FMSUB_BCAST_CASE(1) FMSUB_BCAST_CASE(2) FMSUB_BCAST_CASE(3) FMSUB_BCAST_CASE(4) FMSUB_BCAST_CASE(5)
        FMSUB_BCAST_CASE(6) FMSUB_BCAST_CASE(7) FMSUB_BCAST_CASE(8) FMSUB_BCAST_CASE(9) FMSUB_BCAST_CASE(10)
        FMSUB_BCAST_CASE(11) FMSUB_BCAST_CASE(12) FMSUB_BCAST_CASE(13) FMSUB_BCAST_CASE(14) FMSUB_BCAST_CASE(15)
        default: break;
    }
}
#undef FMSUB_BCAST_CASE

// Cholesky of the 16x16 block at S (LDS, stride PLD) by one wave, one ROW per lane (lanes 16..63 repeat
// lanes 0..15): column j takes the pivot through v_readlane and the scaled column entries L[c][j] of the
// other rows as DPP row broadcasts inside the multiply-add of the rank-1 update: no LDS shuffles, no
// per-element selects.  A lone wave issues in order, one FP64 instruction every ~12 cycles whatever its
// active lanes, so what counts is the instruction count: per column 2 readlanes + 1/sqrt (5) + scale (2)
// + (15 - j) fused multiply-adds + 6 selects = ~23 on average (64 in the 4-lanes-per-row
// layout with five ds_bpermute broadcasts per column, 38 with v_readlane broadcasts).  Same arithmetic
// per element (reciprocal-pivot scaling as dpotf2).
// Writes L (lower, zeros above) and the reciprocal pivots.  Returns the first failing column + 1 (0 if ok).
__device__ __forceinline__ int chol16_wave(double* S, double* rd, int lane) {
    const int i = lane & 15;
    double x[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = S[i * PLD + c];
    int bad = 0;
    double my_d = 1.0, my_r = 1.0;      // this lane's own pivot and its reciprocal root (row i = column i)
#pragma unroll
    for (int j = 0; j < 16; j++) {
        const double djj = readlane_f64(x[j], j);              // S[j][j], wave-uniform
        if (!(djj > 0.0) && bad == 0) bad = j + 1;              // dpotf2: ajj <= 0 or NaN
        const double rinv = pivot_rsqrt(djj);
        if (i == j) { my_d = djj; my_r = rinv; }
        double lij = x[j] * rinv;                               // L[i][j] (rows i < j: unused values)
        // x[c] -= L[i][j] * L[c][j]: the factor of row c is lane c of every 16-lane DPP row (the four
        // rows of the wave are replicas), fused into the multiply-add as a row_newbcast operand, the sign as its
        // source modifier
        asm volatile("s_nop 1" : "+v"(lij));                    // VALU write -> DPP read of the same VGPR
#pragma unroll
        for (int c = j + 1; c < 16; c++) fmsub_row_bcast(x[c], lij, lij, c);   // rows i < c: entries nobody reads
        x[j] = lij;                                             // (rows i <= j: values nobody reads; the diagonal follows below)
    }
    // L[i][i] = sqrt(d_i) = d_i / sqrt(d_i) refined by one Newton step (to the last bit), all sixteen at
    // once, one per lane, instead of four FP64 instructions in every column step
    double piv = my_d * my_r;
    piv = fma(fma(-piv, piv, my_d), 0.5 * my_r, piv);
    if (lane < 16) {
#pragma unroll
        for (int c = 0; c < 16; c++) S[i * PLD + c] = x[c];
        S[i * PLD + i] = piv;
        rd[i] = my_r;                                           // reciprocal pivots for the solves
    }
    return bad;
}

// x <- x * L^-T for the rows at Xr against the 16x16 lower factor at L (both LDS): one row of X per lane
// (ROWS = 16: lanes 0..15, the other three 16-lane rows repeat them; ROWS = 64: all lanes).  Forward
// substitution with reciprocal-pivot scaling.  Lane l also holds row (l & 15) of the factor and its
// reciprocal pivot: L[c2][c] and 1/L[c][c] reach the arithmetic as DPP row broadcasts inside the
// multiply-adds -- 16 + 120 FP64 instructions and no LDS traffic in the 16-step chain (the version
// with 152 broadcast LDS reads, even fetched ahead in two batches, took 3.0k cycles per call).
template <int ROWS = 16>
__device__ __forceinline__ void trsm16_rows(double* Xr, const double* L, const double* rd, int lane) {
    const int i = ROWS == 64 ? lane : (lane & 15);    // 64: one row per lane; 16: four copies
    const int li = lane & 15;
    double x[16], lr[16];
#pragma unroll
    for (int c = 0; c < 16; c++) x[c] = Xr[i * PLD + c];
#pragma unroll
    for (int c = 0; c < 16; c++) lr[c] = L[li * PLD + c];
    const double myr = rd[li];
#pragma unroll
    for (int c = 0; c < 16; c++) {
        double xc = 0.0;
        fmac_row_bcast(xc, myr, x[c], c);                       // x[c] / L[c][c]  (v_mul_f64 has no DPP form)
        x[c] = xc;
        const double nx = -xc;
#pragma unroll
        for (int c2 = c + 1; c2 < 16; c2++) fmac_row_bcast(x[c2], lr[c], nx, c2);   // x[c2] -= x[c] L[c2][c]
    }
    if (lane < ROWS) {
#pragma unroll
        for (int c = 0; c < 16; c++) Xr[i * PLD + c] = x[c];
    }
}

// Loads of a panel step (512 threads).  D and P_t (what the factor waits for) are dealt to all eight waves, 4 pieces of 16
// bytes per thread and block; the workgroup's own rows B and P_o to the four worker waves (threads 256..511, 8 pieces per
// thread and block), which keep them in registers until the first tasks (the update of D) are done.  The chain waves' lanes
// load one fixed element for those: no branch around the loads (register sets loaded behind branches went to scratch), one
// cache line of extra traffic.  Named registers instead of arrays: arrays that live across other work were left in scratch
// memory by the compiler.
#define D4_FOR(X) X(0) X(1) X(2) X(3)
#define D4_DECL(i) double2 rD##i, rPt##i;
#define D4_LOAD(i) { const int e_ = t + 512 * i; const int64_t off_ = (int64_t)(e_ >> 5) * ld + (e_ & 31) * 2;        \
                     rD##i = *reinterpret_cast<const double2*>(gD_ + off_); rPt##i = *reinterpret_cast<const double2*>(gPt_ + off_); }
#define D4_COMMIT(i) { const int e_ = t + 512 * i; const int o_ = (e_ >> 5) * PLD + (e_ & 31) * 2;                     \
                       *reinterpret_cast<double2*>(sD + o_) = rD##i; *reinterpret_cast<double2*>(sPt + o_) = rPt##i; }
#define B8_FOR(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define B8_DECL(i) double2 rB##i, rPo##i;
#define B8_LOAD(i) { const int e_ = t - 256 + 256 * i; const bool ok_ = t >= 256;                                     \
                     const int64_t off_ = ok_ ? (int64_t)(e_ >> 5) * ld + (e_ & 31) * 2 : 0;                          \
                     rB##i = *reinterpret_cast<const double2*>(gB_ + off_); rPo##i = *reinterpret_cast<const double2*>(gPo_ + off_); }
#define B8_COMMIT(i) { const int e_ = t - 256 + 256 * i; const int o_ = (e_ >> 5) * PLD + (e_ & 31) * 2;               \
                       *reinterpret_cast<double2*>(sB + o_) = rB##i; *reinterpret_cast<double2*>(sPo + o_) = rPo##i; }
__device__ __forceinline__ void store_block(double* __restrict__ G, int64_t ld, const double* S, int t,
                                            bool lower_only) {
    for (int e = t; e < 64 * 32; e += (int)blockDim.x) {
        int row = e >> 5, c2 = (e & 31) * 2;
        double2 v = *reinterpret_cast<const double2*>(S + row * PLD + c2);
        double* p = G + (int64_t)row * ld + c2;
        if (!lower_only || c2 + 1 <= row) *reinterpret_cast<double2*>(p) = v;
        else if (c2 <= row) p[0] = v.x;
    }
}

// ---------------------------------------------------------------------------------------------
// V_bb = L_bb^-1 for every 64x64 diagonal block of L (first stage of V = L^-1): one workgroup of four
// waves per block.  Wave w inverts the 16x16 diagonal block w (lane c: forward substitution for column c,
// true divisions), then two doubling levels (16 -> 32 -> 64)
// of V21 = -V22 (L21 V11) as 16x16 MFMA tiles in LDS; V is kept together with its transpose because
// the tile product takes both operands row-wise.  (The single-wave version -- lane c solving L x = e_c
// with 2016 multiply-adds and 64 divisions from broadcast LDS reads -- took 39 us, a quarter of the
// factorisation latency at N <= 128.)
__device__ __forceinline__ void tile_store_t(double* T, v4d v, int lane) {       // T[col][row] = v
    const int r = lane & 15, g = lane >> 4;
#pragma unroll
    for (int q = 0; q < 4; q++) T[r * PLD + g + 4 * q] = v[q];
}
// zero_right: the block right of the diagonal block is cleared as well (its last columns for the last block excepted) -- all that the
// tile-walking products of the chain ever read above the diagonal (see trtri_lower)
__global__ __launch_bounds__(256) void trtri_diag64_kernel(const double* __restrict__ L_, double* __restrict__ V_,
                                                           int64_t ld, const int* info, int64_t bstride, int zero_right) {
    __shared__ __attribute__((aligned(16))) double sL[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sV[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sVt[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sTt[64 * PLD];
    const int tb = (int)blockIdx.z;             // theta of a batched launch (gpry_ctx::bn)
    if (*bset(info, tb, bstride) != 0) return;
    const double* __restrict__ L = bset(L_, tb, bstride);
    double* __restrict__ V = bset(V_, tb, bstride);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    for (int e = t; e < 64 * 64; e += 256) {
        const int i = e >> 6, j = e & 63;
        sL[i * PLD + j] = (j <= i) ? L[(b0 + i) * ld + b0 + j] : 0.0;
        sV[i * PLD + j] = 0.0;
        sVt[i * PLD + j] = 0.0;
    }
    __syncthreads();
    {   // wave w, lane c < 16: column c of L_ww^-1 by forward substitution with true divisions (the
        // reciprocal-multiply form is one rounding per entry further from the reference's dtrsm: on the
        // cond(K) = 5e15 matrix of config 1 that alone moved the posterior mean by 3e-5 of its range)
        const double* Lw = sL + (w * 16) * PLD + w * 16;
        if (lane < 16) {
            double x[16];
#pragma unroll
            for (int i = 0; i < 16; i++) {
                double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; k++) sacc = fma(-Lw[i * PLD + k], x[k], sacc);
                x[i] = (i >= lane) ? sacc / Lw[i * PLD + i] : 0.0;
            }
#pragma unroll
            for (int i = 0; i < 16; i++) {
                sV[(w * 16 + i) * PLD + w * 16 + lane] = x[i];
                sVt[(w * 16 + lane) * PLD + w * 16 + i] = x[i];
            }
        }
    }
    __syncthreads();
    for (int h = 16; h <= 32; h *= 2) {
        const int hb = h / 16, ntile = (64 / (2 * h)) * hb * hb;
        for (int id = w; id < ntile; id += 4) {        // T = L21 V11, stored transposed
            const int p = id / (hb * hb), m = (id % (hb * hb)) / hb, n = id % hb;
            const int lo = p * 2 * h, mid = lo + h;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            acc = mfma_nt16<false>(acc, sL + (mid + 16 * m) * PLD + lo, sVt + (lo + 16 * n) * PLD + lo, h, lane);
            tile_store_t(sTt + (lo + 16 * n) * PLD + 16 * m, acc, lane);
        }
        __syncthreads();
        for (int id = w; id < ntile; id += 4) {        // V21 = -V22 T (and its transpose)
            const int p = id / (hb * hb), m = (id % (hb * hb)) / hb, n = id % hb;
            const int lo = p * 2 * h, mid = lo + h;
            v4d acc = {0.0, 0.0, 0.0, 0.0};
            acc = mfma_nt16<true>(acc, sV + (mid + 16 * m) * PLD + mid, sTt + (lo + 16 * n) * PLD, h, lane);
            tile_store(sV + (mid + 16 * m) * PLD + lo + 16 * n, acc, lane);
            tile_store_t(sVt + (lo + 16 * n) * PLD + mid + 16 * m, acc, lane);
        }
        __syncthreads();
    }
    for (int e = t; e < 64 * 64; e += 256) {
        const int i = e >> 6, j = e & 63;
        V[(b0 + i) * ld + b0 + j] = sV[i * PLD + j];
    }
    if (zero_right && b0 + 128 <= ld) {
        for (int e = t; e < 64 * 64; e += 256) {
            const int i = e >> 6, j = e & 63;
            V[(b0 + i) * ld + b0 + 64 + j] = 0.0;
        }
    }
}
int launch_trtri_diag(gpry_ctx* ctx, const double* L, double* V, int64_t Np, hipStream_t st, bool zero_right) {
    hipLaunchKernelGGL(trtri_diag64_kernel, dim3((unsigned)(Np / 64), 1, (unsigned)ctx->bn), dim3(256), 0, st, L, V, Np, ctx->dinfo,
                       ctx->bstride, zero_right ? 1 : 0);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
// diagonal blocks blk0 .. blk0 + nblk - 1 only (pipelined factor chain, chol.hip)
int launch_trtri_diag_range(gpry_ctx* ctx, const double* L, double* V, int64_t Np, int blk0, int nblk, hipStream_t st) {
    const int64_t off = (int64_t)blk0 * 64 * (Np + 1);
    hipLaunchKernelGGL(trtri_diag64_kernel, dim3((unsigned)nblk, 1, (unsigned)ctx->bn), dim3(256), 0, st, L + off, V + off, Np, ctx->dinfo,
                       ctx->bstride, 0);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// One panel step of the factorisation (the body of the fused launch below).
// `arrive` / `target`: the diagonal workgroup overwrites D with its factor in place, while every
// other workgroup of the launch reads D.  Workgroups count in on `arrive` once their loads have
// landed, and the diagonal workgroup stores only when all of them have (target = arrivals
// expected up to and including this launch).  Without this the result depends on all workgroups
// starting before the first one finishes -- not true when another stream shares the GPU.
struct PanelArgs {
    double* A; int64_t ld, j0, K0, n_real;      // j0, K0: columns relative to A (K0 = j0 - 64, or j0 for the first strip); n_real: real (unpadded) size of the WHOLE matrix
    int64_t col0;           // A is the trailing submatrix from column col0 of the whole matrix on (tail of the large schedule)
    int* info; int* arrive; int target;
    int64_t bstride;        // batched launch (gpry_ctx::bn): A, info and arrive of theta blockIdx.z lie this many doubles further on
    int flags;              // experiments (GPRY_PANEL_FLAGS): 1: the workers do not yield to their SIMD partners; 16 / 32 (host): the compact / the roomy step for every launch
    // THE INVERSE FACTOR AS EXTRA ROWS (potrf_stacked below): workgroups bx >= n_top own row block bx - n_top of a second matrix, the
    // appended block, that lies aug_delta doubles behind A (same leading dimension); outside that schedule n_top is "all of them"
    int n_top; int64_t aug_delta;
};
#define PANEL_SMEM_DOUBLES (4 * 64 * PLD + 64 + 16)

#ifdef GPRY_PANEL_STAMPS
// diagnostic build only (tools/r05/build_stamps.sh): s_memtime stamps of three workgroups per panel step
#define STAMP_STEPS 160
#define STAMP_SLOTS 8
__device__ long long g_panel_stamps[STAMP_STEPS][3][8][STAMP_SLOTS];
#define PANEL_STAMP(slot) do { if (stamp_wg >= 0 && lane == 0) g_panel_stamps[stamp_step][stamp_wg][w][slot] = (long long)__builtin_amdgcn_s_memtime(); } while (0)
// time per kind of own-row task (0: U, 1: M, 2: T incl. its waits, 3: count of tasks), per wave
__device__ long long g_panel_acc[STAMP_STEPS][3][8][4];
#define PANEL_ACC_DECL long long acc_t0_ = 0, acc_[4] = {0, 0, 0, 0};
#define PANEL_ACC_BEGIN do { acc_t0_ = (long long)__builtin_amdgcn_s_memtime(); } while (0)
#define PANEL_ACC_END(kind) do { acc_[kind] += (long long)__builtin_amdgcn_s_memtime() - acc_t0_; acc_[3]++; } while (0)
#define PANEL_ACC_FLUSH do { if (stamp_wg >= 0 && lane == 0) for (int q_ = 0; q_ < 4; q_++) g_panel_acc[stamp_step][stamp_wg][w][q_] = acc_[q_]; } while (0)
extern "C" int gpry_debug_panel_acc(long long* out, int n) {
    if (n > STAMP_STEPS * 3 * 8 * 4) n = STAMP_STEPS * 3 * 8 * 4;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_acc), sizeof(long long) * n);
}
extern "C" int gpry_debug_panel_stamps(long long* out, int n) {
    if (n > STAMP_STEPS * 3 * 8 * STAMP_SLOTS) n = STAMP_STEPS * 3 * 8 * STAMP_SLOTS;
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_panel_stamps), sizeof(long long) * n);
}
// progress codes of every wave of the first workgroups, written straight to mapped host memory: readable while a kernel hangs
__device__ int* g_panel_progress = nullptr;
#define PANEL_PROGRESS(code) do { if (g_panel_progress && lane == 0 && bx < 4 && tb == 0) \
    __hip_atomic_store(g_panel_progress + ((stamp_step & 15) * 4 + bx) * 8 + w, (code), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); } while (0)
extern "C" int* gpry_debug_progress_buffer() {
    int* h = nullptr;
    if (hipHostMalloc((void**)&h, 16 * 4 * 8 * sizeof(int), hipHostMallocMapped) != hipSuccess) return nullptr;
    memset(h, 0, 16 * 4 * 8 * sizeof(int));
    int* d = nullptr;
    if (hipHostGetDevicePointer((void**)&d, h, 0) != hipSuccess) return nullptr;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_panel_progress), &d, sizeof(d)) != hipSuccess) return nullptr;
    return h;
}
extern "C" int gpry_debug_read_info(gpry_ctx* ctx, int* out4) {
    return (int)hipMemcpy(out4, ctx->dinfo, 4 * sizeof(int), hipMemcpyDeviceToHost);
}
#else
#define PANEL_STAMP(slot) do { } while (0)
#define PANEL_PROGRESS(code) do { } while (0)
#define PANEL_ACC_DECL
#define PANEL_ACC_BEGIN do { } while (0)
#define PANEL_ACC_END(kind) do { } while (0)
#define PANEL_ACC_FLUSH do { } while (0)
#endif

// waits on the LDS words of the panel step are bounded: a wave that has spun for ~0.2 s (or sees that another one has)
// gives up, the step is reported as failed (info) and marked in info[3] -- never a hung GPU
#define PANEL_SPIN_CAP (1 << 21)
// (every value a wave branches on goes through v_readfirstlane: the conditions are wave-uniform by construction, and with
// that the compiler emits plain scalar loops -- left to its divergence analysis, the task loop below came out as nested
// exec-masked loops that re-ran a pulled task without pulling the next one)
__device__ __forceinline__ int lds_peek(int* f, bool acquire) {
    const int v = acquire ? __hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP)
                          : __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    return __builtin_amdgcn_readfirstlane(v);
}
__device__ __forceinline__ void lds_wait_ge(int* f, int v, int* s_abort, int id) {       // id: which wait (reported in info[3])
    for (int n = 0;; n++) {
        if (lds_peek(f, true) >= v) return;
        __builtin_amdgcn_s_sleep(1);
        if (lds_peek(s_abort, false) != 0) return;
        if (n > PANEL_SPIN_CAP) {
            __hip_atomic_store(s_abort, id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            return;
        }
    }
}
// Wave w of the chain and worker w + 4 share a SIMD (a workgroup's waves are dealt to the four SIMDs in turn), and the FP64
// MFMAs of one hold up the FP64 vector instructions of the other (chol16 of wave 0: 4.5k cycles alone, 6.4k beside a worker's
// tiles, stamped build).  The chain wave that everything waits for is the one that factors next, wave s_flag[0]: its partner
// pulls no new task until that block is factored.  (Measured against the alternatives, tools/r05/gpu_panel_flags.sh and
// profiles/r05_potrf.md: no yield -- the chain is 15 % slower and the tail behind it as much shorter; priority for the chain
// waves -- no effect; the chain on two SIMDs of its own and the workers on the other two -- 3 % slower, the workers' MFMAs
// then share two matrix pipes.)
__device__ __forceinline__ void worker_yield(int* blocks_factored, int w, int* s_abort) {
    for (int n = 0; n < PANEL_SPIN_CAP; n++) {
        if (lds_peek(blocks_factored, false) != w - 4 || lds_peek(s_abort, false) != 0) return;
        __builtin_amdgcn_s_sleep(2);
    }
}
__device__ __forceinline__ void lds_publish(int* f, int v, int lane) {         // after a wave_fence(): this wave's LDS writes come first
    if (lane == 0) __hip_atomic_store(f, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void lds_count(int* f, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(f, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// One panel step for the 64-row block `bx` of the panel: the strip before its own (columns [K0, j0), 64 of them) is applied
// first (left-looking), then the 64 x 64 factor and the row solves.  Workgroups of EIGHT waves: 0..3 are the factor chain
// (wave w owns block row w of D), 4..7 are workers.
//
// Round 5: what does not feed the factor is off the chain.  Until round 4 every workgroup (four waves) loaded its four
// 64 x 64 blocks, applied the previous strip to D AND to its own rows (5.9k cycles), factored D (21k), solved its rows (3k)
// and stored: 38k cycles per step plus the launch, one after the other.  Now
//   * only D and the previous strip's rows of the diagonal block (P_t) are waited for and committed to LDS before the factor
//     starts; the workgroup's own rows B and P_o are loaded by the worker waves, stay in their registers over the first
//     tasks and are committed when those are done (they have landed by then);
//   * wave 0 applies the previous strip to tile (0, 0) of D and starts the 16 x 16 factor chain; the other nine tiles of D
//     are the first task queue (waves 1..7 pull them; wave w of the chain waits for its block row);
//   * everything for the workgroup's own rows -- the left-looking update of B (16 tiles of 16 MFMAs), the updates with the
//     columns solved so far and the four 16-wide solves -- is a second queue of 32 tasks in dependency order, pulled from an
//     LDS counter by the workers and by the chain waves once their part of the chain is done: the own-row work runs beside
//     the factor, two waves to a SIMD (a lone wave issues one FP64 instruction per 8 cycles, the SIMD takes one per 4), and
//     only the last 16-wide solve is left behind the factor.
// Every tile sees the same operations in the same order as before (who computes it and when is all that changed; an MFMA chain
// cut at a multiple of 4 k and resumed from the stored tile is the same chain): factors bit-identical to round 4's.
__device__ __forceinline__ void panel_step_body(const PanelArgs& pa, double* smem, const int bx, const int tb) {
    double* __restrict__ A = bset(pa.A, tb, pa.bstride);
    const int64_t ld = pa.ld, j0 = pa.j0, K0 = pa.K0, n_real = pa.n_real;
    int* info = bset(pa.info, tb, pa.bstride); int* arrive = bset(pa.arrive, tb, pa.bstride); const int target = pa.target;
    double* sD = smem;
    double* sB = sD + 64 * PLD;
    double* sPt = sB + 64 * PLD;
    double* sPo = sPt + 64 * PLD;
    double* sRd = sPo + 64 * PLD;
    int* s_int = reinterpret_cast<int*>(sRd + 64);
    int& s_bad = s_int[0];
    int* s_abort = s_int + 1;
    int* s_flag = s_int + 2;        // [0]: 16 x 16 blocks factored, [1 + w]: columns solved by wave w, [5]: own-row column blocks solved
    int* s_commit = s_int + 10;     // worker waves that have committed their share of B / P_o
    int* s_task = s_int + 11;       // next own-row task x 64 (every lane of a pulling wave adds one)
    int* s_udone = s_int + 12;      // [c]: tiles of column block c that have the previous strip applied
    int* s_mdone = s_int + 16;      // [c]: tiles of column block c that have the solved columns applied
    int* s_dtask = s_int + 20;      // next tile of the update of D x 64
    int* s_ddone = s_int + 21;      // [w]: tiles of block row w of D that have the previous strip applied (w = 1..3)
    if (*info != 0) return;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);   // (scalar: the wave's role decides its control flow)
    const bool is_diag = bx == 0, worker = w >= 4;
    const int row = w;                              // chain waves: the block row of D they own
    // the workgroup's own rows: block bx below the diagonal block -- or block bx - n_top of the appended matrix (Aown)
    const bool appended = bx >= pa.n_top;
    const int64_t R = appended ? 64 * (int64_t)(bx - pa.n_top) : j0 + 64 * (int64_t)bx;
    double* __restrict__ Aown = appended ? A + pa.aug_delta : A;
    const bool has_prev = j0 > K0;                 // false for the first strip of a segment
#ifdef GPRY_PANEL_STAMPS
    const int stamp_step = (int)((pa.col0 + j0) / 64) < STAMP_STEPS ? (int)((pa.col0 + j0) / 64) : STAMP_STEPS - 1;
    const int stamp_wg = (tb == 0) ? (bx == 0 ? 0 : bx == 1 ? 1 : bx == 8 ? 2 : -1) : -1;
#endif
    PANEL_STAMP(0);
    if (t < 32) s_int[t] = 0;
    D4_FOR(D4_DECL)
    B8_FOR(B8_DECL)
    {
        const double* __restrict__ gD_ = A + j0 * ld + j0;
        const double* __restrict__ gPt_ = A + j0 * ld + K0;
        D4_FOR(D4_LOAD)
    }
    // (what the factor waits for goes out first: loads return in order, and the scheduler had put the own-row loads in front)
    __builtin_amdgcn_sched_barrier(0);
    {
        const double* __restrict__ gB_ = Aown + R * ld + j0;
        const double* __restrict__ gPo_ = Aown + R * ld + K0;
        B8_FOR(B8_LOAD)
    }
    D4_FOR(D4_COMMIT)
    __syncthreads();
    if (t == 256) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (a worker: wave 0 starts the chain)
    PANEL_STAMP(1);
    PANEL_PROGRESS(1);
    // ---- the previous strip onto D (only the ten tiles on and below the diagonal are ever read): tile (0, 0) by wave 0,
    // which starts the chain; the other nine from a queue (waves 1..3 and 5..7; block row 1 first)
    if (has_prev) {
        if (w == 0) {
            v4d acc = tile_load(sD, lane);
            acc = mfma_nt16<true>(acc, sPt, sPt, 64, lane);
            tile_store(sD, acc, lane);
            wave_fence();
        } else if (w != 4 || (pa.flags & 1)) {       // (wave 4 is the partner of wave 0, which is on the chain from the first cycle)
            for (;;) {
                const int dt = __builtin_amdgcn_readfirstlane(
                    __hip_atomic_fetch_add(s_dtask, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 6;
                if (dt >= 9) break;
                const int rw = dt < 2 ? 1 : dt < 5 ? 2 : 3;
                const int n = dt - (rw == 1 ? 0 : rw == 2 ? 2 : 5);
                double* T = sD + (rw * 16) * PLD + n * 16;
                v4d acc = tile_load(T, lane);
                acc = mfma_nt16<true>(acc, sPt + (rw * 16) * PLD, sPt + (n * 16) * PLD, 64, lane);
                tile_store(T, acc, lane);
                wave_fence();
                lds_count(&s_ddone[rw], lane);
            }
        }
    }
    if (!is_diag && worker) {
        B8_FOR(B8_COMMIT)
        wave_fence();
        lds_count(s_commit, lane);
    }
    PANEL_STAMP(2);
    PANEL_PROGRESS(2);
    if (!worker) {
        // ---- Cholesky of the 64x64 diagonal block, blocked by 16, as a dataflow between the four chain waves
        // (wave w owns block row w) instead of three workgroup barriers per block column:
        // (block row w below = `row`)
        //   for cb < w:  wait chol(cb);  T(w,cb): D[w][cb] <- D[w][cb] L[cb][cb]^-T;  publish;
        //                for cc in cb+1..w: (cc < w: wait T(cc,cb))  D[w][cc] -= D[w][cb] D[cc][cb]^T
        //   chol(w); publish.
        // Wave cb+1 starts chol(cb+1) as soon as ITS row is done, while the rows below still work on block
        // column cb: the chain is 4 chol16 + 3 (solve + one tile update).
        // Flags are LDS words written by lane 0 after a wave fence (LDS requests of a wave retire in order).
        if (has_prev && row > 0) lds_wait_ge(&s_ddone[row], row + 1, s_abort, 10);
        for (int cb = 0; cb < row; cb++) {
            PANEL_PROGRESS(10 + cb);
            lds_wait_ge(&s_flag[0], cb + 1, s_abort, 1);
            PANEL_PROGRESS(14 + cb);
            trsm16_rows(sD + (row * 16) * PLD + cb * 16, sD + (cb * 16) * PLD + cb * 16, sRd + cb * 16, lane);
            wave_fence();
            lds_publish(&s_flag[1 + row], cb + 1, lane);
            for (int cc = cb + 1; cc <= row; cc++) {
                if (cc < row) lds_wait_ge(&s_flag[1 + cc], cb + 1, s_abort, 2);
                double* T = sD + (row * 16) * PLD + cc * 16;
                v4d acc = tile_load(T, lane);
                acc = mfma_nt16<true>(acc, sD + (row * 16) * PLD + cb * 16, sD + (cc * 16) * PLD + cb * 16, 16, lane);
                tile_store(T, acc, lane);
            }
            wave_fence();
        }
        // a failed pivot (not positive definite) still publishes: nobody may wait forever; the first
        // failing column wins (the chol16 calls are ordered by the chain itself)
        PANEL_PROGRESS(20);
        const int bad = chol16_wave(sD + (row * 16) * PLD + row * 16, sRd + row * 16, lane);
        PANEL_PROGRESS(21);
        if (bad && lane == 0 && s_bad == 0) s_bad = row * 16 + bad;
        wave_fence();
        lds_publish(&s_flag[0], row + 1, lane);
    }
    PANEL_STAMP(3);
    PANEL_PROGRESS(22);
    if (!is_diag) {
        // ---- the workgroup's own rows: X = (B - P_o P_t^T) Lkk^-T, 16 columns at a time.  Tasks in dependency order,
        //   column block c:  U(rt, c), rt = 0..3: the previous strip onto tile (rt, c)          [needs: B, P_o committed]
        //                    M(rt, c), rt = 0..3 (c > 0): the solved columns onto tile (rt, c)  [U(., c), the solves < c, row c of the factor]
        //                    T(c): the 16-wide solve of all 64 rows, one row per lane           [M(., c), chol(c)]
        // pulled from s_task by whichever wave is free.  A wave that waits for a dependency waits for a task pulled earlier
        // or for the factor chain, which waits for none of this: no deadlock.
        lds_wait_ge(s_commit, 4, s_abort, 3);
        PANEL_ACC_DECL
        for (;;) {
            if (worker && !(pa.flags & 1)) worker_yield(&s_flag[0], w, s_abort);
            // every lane adds one (the compiler folds that into ONE ds_add of 64 and a per-lane offset): no value flows out
            // of an `if (lane == 0)` -- with the pull under such a branch the loop was compiled into a nest of exec-masked
            // loops that re-ran a pulled task without pulling the next one
            const int task = __builtin_amdgcn_readfirstlane(
                __hip_atomic_fetch_add(s_task, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 6;
            PANEL_PROGRESS(100 + task);
            if (task >= 32) break;
            // column block 0: tasks 0..4 (4 U, T); column block c > 0: 5 + 9 (c - 1) .. (4 U, 4 M, T)
            const int c = task < 5 ? 0 : 1 + (task - 5) / 9;
            const int k = task < 5 ? task : (task - 5) % 9;
            PANEL_ACC_BEGIN;
            if (k < 4) {                                    // U(k, c)
                if (has_prev) {
                    double* U = sB + (k * 16) * PLD + c * 16;
                    v4d acb = tile_load(U, lane);
                    acb = mfma_nt16<true>(acb, sPo + (k * 16) * PLD, sPt + (c * 16) * PLD, 64, lane);
                    tile_store(U, acb, lane);
                    wave_fence();
                }
                lds_count(&s_udone[c], lane);
                PANEL_ACC_END(0);
            } else if (c > 0 && k < 8) {                    // M(k - 4, c)
                const int rt = k - 4;
                lds_wait_ge(&s_udone[c], 4, s_abort, 4);
                lds_wait_ge(&s_flag[5], c, s_abort, 5);                // X[:, 0 .. 16 c) final
                lds_wait_ge(&s_flag[1 + c], c, s_abort, 6);            // L[c][0 .. c) final
                double* T = sB + (rt * 16) * PLD + c * 16;
                v4d acc = tile_load(T, lane);
                acc = mfma_nt16<true>(acc, sB + (rt * 16) * PLD, sD + (c * 16) * PLD, c * 16, lane);
                tile_store(T, acc, lane);
                wave_fence();
                lds_count(&s_mdone[c], lane);
                PANEL_ACC_END(1);
            } else {                                        // T(c)
                if (c > 0) lds_wait_ge(&s_mdone[c], 4, s_abort, 7); else lds_wait_ge(&s_udone[0], 4, s_abort, 8);
                lds_wait_ge(&s_flag[0], c + 1, s_abort, 9);
                trsm16_rows<64>(sB + c * 16, sD + (c * 16) * PLD + c * 16, sRd + c * 16, lane);
                wave_fence();
                lds_publish(&s_flag[5], c + 1, lane);
                PANEL_ACC_END(2);
            }
        }
        PANEL_ACC_FLUSH;
    }
    PANEL_STAMP(4);
    PANEL_PROGRESS(40);
    __syncthreads();
    PANEL_STAMP(5);
    PANEL_PROGRESS(41);
    if ((pa.flags & 64) && is_diag && t == 0) *s_abort = 0x7E;      // test hook ("panel_debug" = 64): as if a bounded wait had run out
    if (pa.flags & 64) __syncthreads();
    if (*s_abort != 0) {            // a bounded wait ran out (never seen; kept so that a protocol error cannot hang the GPU)
        if (t == 0) { atomicCAS(info, 0, (int)(pa.col0 + j0 + 1 <= n_real ? pa.col0 + j0 + 1 : n_real)); info[3] = 0x5A00 + *s_abort; }
        return;
    }
    if (s_bad) {
        if (is_diag && t == 0) {
            int64_t col = pa.col0 + j0 + s_bad;             // 1-based failing column of the whole matrix
            atomicCAS(info, 0, (int)(col <= n_real ? col : n_real));
        }
        return;
    }
    if (is_diag) {
        if (t == 0) {
            int n = 0;
            while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++n > PANEL_SPIN_CAP) { atomicCAS(info, 0, (int)(pa.col0 + j0 + 1 <= n_real ? pa.col0 + j0 + 1 : n_real)); info[3] = 0x5AFF; break; }
            }
        }
        __syncthreads();
        store_block(A + j0 * ld + j0, ld, sD, t, true);
        PANEL_STAMP(6);
        PANEL_PROGRESS(60);
        return;
    }
    store_block(Aown + R * ld + j0, ld, sB, t, false);
    PANEL_STAMP(6);
    PANEL_PROGRESS(61);
}

// ---------------------------------------------------------------------------------------------
// One 64 x 64 tile of a trailing update, C -= A_r P^T with the 128 columns of one panel: both operand
// images (64 rows x 128 k, row stride 130 doubles: conflict-free fragment reads) sit in LDS at once; the eight waves
// (round 5; four until then, a 32 x 32 quadrant each) own a 16 x 32 piece each (1 x 2 MFMA tiles, 32 k-steps): two waves to
// a SIMD, one's fragment reads under the other's MFMAs.  Per element the continuation of ONE MFMA chain that
// starts at the covariance entry, k ascending -- as in the SYRK launches of the GEMM engines (EPI_SUB) and the
// left-looking update of the panel step: every schedule of the factorisation gives the same bits.  Small on purpose: shorter than a panel step, so that tiles
// riding in a panel launch never set its length (a lone 128 x 128 x 128 tile takes 32-36 us).
#define S64 130
// n consecutive panels (128 columns apart) in one visit, then (half != 0) the 64 columns at ha_off / hb_off: the first
// half of the panel that has only just been completed (its second half is applied by the next panel step itself)
struct TileItem { int64_t a_off, b_off, c_off, ha_off, hb_off; int32_t n, half; };
__device__ __forceinline__ void syrk64_tile_body(double* __restrict__ A, int64_t ld, const TileItem it, double* smem,
                                                 const int* info, const int64_t aug_delta) {
    if (*info != 0) return;
    double* sA = smem;
    double* sB = smem + 64 * S64;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // (half: bit 0: the visit ends with the half panel; bit 1: the tile and its row operand lie in the appended matrix of
    // potrf_stacked, aug_delta doubles behind A -- a_off, c_off and ha_off count from there)
    const bool appended = (it.half & 2) != 0, with_half = (it.half & 1) != 0;
    double* __restrict__ Ar = appended ? A + aug_delta : A;
    const bool diag = !appended && it.a_off == it.b_off;
    const int wr = w >> 1, wc = w & 1, r = lane & 15, g = lane >> 4;
    // C/D fragment: row = g + 4q, col = r.  The 8 old values of this lane stay in registers over the visit.
    double* cbase = Ar + it.c_off + (int64_t)(wr * 16 + g) * ld + wc * 32 + r;
    // ONE MFMA chain per element over all panels it ever receives, k ascending, started at the covariance entry (round 6): the
    // accumulators take -C, the products of this visit's panels are added, -acc goes back -- the continuation of the chain the
    // earlier visits left, whatever launch they rode in (see gemm_dma_body.h: EPI_SUB).
    v4d acc[2];
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[ni][q] = -cbase[(int64_t)(4 * q) * ld + ni * 16];
    if (diag) sB = sA;
    // A lagging tile takes several pending panels in one visit, in order.
    const int nvisit = (int)it.n + (with_half ? 1 : 0);
#pragma unroll 1
    for (int u = 0; u < nvisit; u++) {
        const bool half = u >= (int)it.n;       // the last visit of a tile with `half`: 64 k instead of 128
        const double* Ag = half ? Ar + it.ha_off : Ar + it.a_off + (int64_t)u * 128;
        const double* Bg = half ? A + it.hb_off : A + it.b_off + (int64_t)u * 128;
        const int kd = half ? 64 : 128;
        if (u) __syncthreads();                 // everybody has read the previous panel's images
        // LDS-DMA: one wave instruction moves one 1-KiB row (128 k) of an operand straight into its padded LDS
        // row -- no staging registers, all 8 (+8) rows of a wave in flight at once (a half row: lanes 0-31)
        if (!half || lane < 32) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int row = w * 8 + i;
                gd_dma16(Ag + (int64_t)row * ld + 2 * lane, sA + row * S64);
            }
            if (!diag) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int row = w * 8 + i;
                    gd_dma16(Bg + (int64_t)row * ld + 2 * lane, sB + row * S64);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const double* pa = sA + (wr * 16 + r) * S64 + g;
        const double* pb = sB + (wc * 32 + r) * S64 + g;
#pragma unroll 1
        for (int kh = 0; kh < kd; kh += 64) {       // k ascending, 64 at a time (one or two passes).  (All fragment reads of a pass
                                                    // in front of its MFMAs, as in mfma_nt16: 1415 instead of 1382 us at N = 4096.)
#pragma unroll 8
            for (int k0 = kh; k0 < kh + 64; k0 += 4) {
                const double a0 = pa[k0], b0 = pb[k0], b1 = pb[16 * S64 + k0];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int q = 0; q < 4; q++) cbase[(int64_t)(4 * q) * ld + ni * 16] = -acc[ni][q];
}

// Loads of the COMPACT panel step (512 threads, below).  D and P_t (what the factor waits for) go to LDS through all eight
// waves as in the step above (D4_*).  The workgroup's own rows never pass through LDS on their way in: worker i (wave 4 + i) takes
// row tile i of P_o straight into the A-operand layout of the MFMA (lane (r, g): P_o[16 i + r][4 u + g], u = 0..15) and row
// tile i of B into the accumulator layout of its four column tiles (lane (r, g): B[16 i + g + 4 q][16 c + r]) -- named
// registers, not arrays: arrays that live across other work were left in scratch memory by the compiler.
#define U16_FOR(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15)
#define PO_DECL(u) double po##u = 0.0;
#define PO_LOAD(u) po##u = gPo_[4 * u];
#define PT_DECL(u) double pt##u;
#define PT_READ(u) pt##u = ptc_[4 * u];
#define U_MFMA(u) acc_ = __builtin_amdgcn_mfma_f64_16x16x4f64(-po##u, pt##u, acc_, 0, 0, 0);
#define PANEL_COMPACT_SMEM_DOUBLES (2 * 64 * PLD + 64 + 16)

// THE COMPACT FORM of the panel step (batched evaluations; the single evaluation keeps panel_step_body above).  One panel step
// for the 64-row block `bx` of the panel: the strip before its own (columns [K0, j0), 64 of them) is applied
// first (left-looking), then the 64 x 64 factor and the row solves.  Workgroups of EIGHT waves: 0..3 are the factor chain
// (wave w owns block row w of D), 4..7 are workers (worker i owns row tile i of the workgroup's own rows).  68 KB of LDS.
//
// Round 5: what does not feed the factor is off the chain, and the step no longer needs a CU to itself.  Until round 4
// every workgroup (four waves, 135 KB of LDS) loaded its four 64 x 64 blocks, applied the previous strip to D AND to its own
// rows (5.9k cycles), factored D (21k), solved its rows (3k) and stored: 38k cycles per step plus the launch, one after the other.
// Now
//   * only D and the previous strip's rows of the diagonal block (P_t) go to LDS and are waited for before the factor starts;
//   * wave w of the chain applies the previous strip to ITS block row of D (w + 1 tiles; wave 0 starts the 16 x 16 factor
//     chain behind one);
//   * the workgroup's own rows B and P_o stay in the registers of the workers in MFMA operand layout: worker i applies the
//     previous strip to its row tile (64 MFMAs, A operand from registers, B operand = P_t from LDS) beside the factor, and
//     writes the result to LDS only when nobody reads P_t any more -- INTO the space of P_t.  Two blocks of LDS instead of
//     four: a panel workgroup fits beside a GEMM workgroup of another stream on the same CU (or beside a second panel
//     workgroup), so that work overlapped with the chain no longer starves it of whole CUs;
//   * what then remains for the own rows -- the updates with the columns solved so far (M) and the four 16-wide solves (T),
//     16 tasks in dependency order -- is pulled from an LDS counter by the workers and by the chain waves once their block
//     row is factored; only the last solve is left behind the factor.
// Every tile sees the same operations in the same order as before (who computes it, from where and when is all that changed; an
// MFMA chain cut at a multiple of 4 k and resumed from the stored tile is the same chain): factors bit-identical to round 4's.
__device__ __forceinline__ void panel_step_compact(const PanelArgs& pa, double* smem, const int bx, const int tb) {
    double* __restrict__ A = bset(pa.A, tb, pa.bstride);
    const int64_t ld = pa.ld, j0 = pa.j0, K0 = pa.K0, n_real = pa.n_real;
    int* info = bset(pa.info, tb, pa.bstride); int* arrive = bset(pa.arrive, tb, pa.bstride); const int target = pa.target;
    double* sD = smem;
    double* sPt = sD + 64 * PLD;
    double* sB = sPt;               // the own rows, with the previous strip applied, once P_t is dead (s_ptdone)
    double* sRd = sPt + 64 * PLD;
    int* s_int = reinterpret_cast<int*>(sRd + 64);
    int& s_bad = s_int[0];
    int* s_abort = s_int + 1;
    int* s_flag = s_int + 2;        // [0]: 16 x 16 blocks factored, [1 + w]: columns solved by wave w, [5]: own-row column blocks solved
    int* s_ptdone = s_int + 10;     // waves that will not read P_t again (8: the space becomes B)
    int* s_task = s_int + 11;       // next own-row task x 64 (every lane of a pulling wave adds one)
    int* s_bdone = s_int + 12;      // workers that have written their row tile of B to LDS
    int* s_mdone = s_int + 16;      // [c]: tiles of column block c that have the solved columns applied
    if (*info != 0) return;
    const int t = threadIdx.x, lane = t & 63, w = __builtin_amdgcn_readfirstlane(t >> 6);   // (scalar: the wave's role decides its control flow)
    const bool is_diag = bx == 0, worker = w >= 4;
    const int row = w & 3;          // chain waves: the block row of D they own; workers: the row tile of the own rows
    const bool appended = bx >= pa.n_top;          // (the appended matrix of potrf_stacked: see PanelArgs)
    const int64_t R = appended ? 64 * (int64_t)(bx - pa.n_top) : j0 + 64 * (int64_t)bx;
    double* __restrict__ Aown = appended ? A + pa.aug_delta : A;
    const bool has_prev = j0 > K0;                 // false for the first strip of a segment
    const int r = lane & 15, g = lane >> 4;
#ifdef GPRY_PANEL_STAMPS
    const int stamp_step = (int)((pa.col0 + j0) / 64) < STAMP_STEPS ? (int)((pa.col0 + j0) / 64) : STAMP_STEPS - 1;
    const int stamp_wg = (tb == 0) ? (bx == 0 ? 0 : bx == 1 ? 1 : bx == 8 ? 2 : -1) : -1;
#endif
    PANEL_STAMP(0);
    if (t < 32) s_int[t] = 0;
    if (pa.flags & 2) { if (worker) __builtin_amdgcn_s_setprio(1); else __builtin_amdgcn_s_setprio(3); }   // experiment: panel waves above the tile waves they share a CU with
    D4_FOR(D4_DECL)
    {
        const double* __restrict__ gD_ = A + j0 * ld + j0;
        const double* __restrict__ gPt_ = A + j0 * ld + K0;
        D4_FOR(D4_LOAD)
    }
    // (what the factor waits for goes out first: loads return in order, and the scheduler had put the own-row loads in front)
    __builtin_amdgcn_sched_barrier(0);
    U16_FOR(PO_DECL)
    v4d bt0 = {0.0, 0.0, 0.0, 0.0}, bt1 = bt0, bt2 = bt0, bt3 = bt0;
    if (worker && !is_diag) {
        const double* __restrict__ gPo_ = Aown + (R + 16 * row + r) * ld + K0 + g;
        if (has_prev) { U16_FOR(PO_LOAD) }
        const double* __restrict__ gB_ = Aown + (R + 16 * row + g) * ld + j0 + r;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            bt0[q] = gB_[(int64_t)(4 * q) * ld]; bt1[q] = gB_[(int64_t)(4 * q) * ld + 16];
            bt2[q] = gB_[(int64_t)(4 * q) * ld + 32]; bt3[q] = gB_[(int64_t)(4 * q) * ld + 48];
        }
    }
    D4_FOR(D4_COMMIT)
    __syncthreads();
    if (t == 256) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (a worker: wave 0 starts the chain)
    PANEL_STAMP(1);
    PANEL_PROGRESS(1);
    if (!worker) {
        // ---- the previous strip onto block row `row` of D (only the tiles on and below the diagonal are ever read)
        if (has_prev) {
#pragma unroll 1
            for (int n = 0; n <= row; n++) {
                double* T = sD + (row * 16) * PLD + n * 16;
                v4d acc = tile_load(T, lane);
                acc = mfma_nt16<true>(acc, sPt + (row * 16) * PLD, sPt + (n * 16) * PLD, 64, lane);
                tile_store(T, acc, lane);
            }
            wave_fence();
        }
        lds_count(s_ptdone, lane);
        PANEL_STAMP(2);
        PANEL_PROGRESS(2);
        // ---- Cholesky of the 64x64 diagonal block, blocked by 16, as a dataflow between the four chain waves
        // (wave w owns block row w = `row`) instead of three workgroup barriers per block column:
        //   for cb < w:  wait chol(cb);  T(w,cb): D[w][cb] <- D[w][cb] L[cb][cb]^-T;  publish;
        //                for cc in cb+1..w: (cc < w: wait T(cc,cb))  D[w][cc] -= D[w][cb] D[cc][cb]^T
        //   chol(w); publish.
        // Wave cb+1 starts chol(cb+1) as soon as ITS row is done, while the rows below still work on block
        // column cb: the chain is 4 chol16 + 3 (solve + one tile update).
        // Flags are LDS words written by lane 0 after a wave fence (LDS requests of a wave retire in order).
        for (int cb = 0; cb < row; cb++) {
            PANEL_PROGRESS(10 + cb);
            lds_wait_ge(&s_flag[0], cb + 1, s_abort, 1);
            PANEL_PROGRESS(14 + cb);
            trsm16_rows(sD + (row * 16) * PLD + cb * 16, sD + (cb * 16) * PLD + cb * 16, sRd + cb * 16, lane);
            wave_fence();
            lds_publish(&s_flag[1 + row], cb + 1, lane);
            for (int cc = cb + 1; cc <= row; cc++) {
                if (cc < row) lds_wait_ge(&s_flag[1 + cc], cb + 1, s_abort, 2);
                double* T = sD + (row * 16) * PLD + cc * 16;
                v4d acc = tile_load(T, lane);
                acc = mfma_nt16<true>(acc, sD + (row * 16) * PLD + cb * 16, sD + (cc * 16) * PLD + cb * 16, 16, lane);
                tile_store(T, acc, lane);
            }
            wave_fence();
        }
        // a failed pivot (not positive definite) still publishes: nobody may wait forever; the first
        // failing column wins (the chol16 calls are ordered by the chain itself)
        PANEL_PROGRESS(20);
        const int bad = chol16_wave(sD + (row * 16) * PLD + row * 16, sRd + row * 16, lane);
        PANEL_PROGRESS(21);
        if (bad && lane == 0 && s_bad == 0) s_bad = row * 16 + bad;
        wave_fence();
        lds_publish(&s_flag[0], row + 1, lane);
    } else if (!is_diag) {
        // ---- worker `row`: the previous strip onto row tile `row` of the own rows, B(row, c) -= P_o(row) P_t(c)^T for the four
        // column tiles: 16 MFMAs each, k ascending, A operand from registers, B operand from LDS -- the chain of round 4's
        // U tiles.  (The partner of wave 0 waits for the first 16 x 16 factor: FP64 MFMAs of one wave of a SIMD hold up the
        // FP64 vector instructions of the other.)
        if (has_prev) {
            if (w == 4 && !(pa.flags & 1)) lds_wait_ge(&s_flag[0], 1, s_abort, 12);
#define U_TILE(c, BT) { const double* ptc_ = sPt + ((c) * 16 + r) * PLD + g; U16_FOR(PT_DECL) U16_FOR(PT_READ)               \
                        __builtin_amdgcn_sched_barrier(0); v4d acc_ = BT; U16_FOR(U_MFMA) BT = acc_; __builtin_amdgcn_sched_barrier(0); }
            U_TILE(0, bt0) U_TILE(1, bt1) U_TILE(2, bt2) U_TILE(3, bt3)
#undef U_TILE
            wave_fence();
        }
        lds_count(s_ptdone, lane);
        PANEL_STAMP(2);
        lds_wait_ge(s_ptdone, 8, s_abort, 13);          // nobody reads P_t any more: its space takes the updated own rows
        tile_store(sB + (row * 16) * PLD, bt0, lane);
        tile_store(sB + (row * 16) * PLD + 16, bt1, lane);
        tile_store(sB + (row * 16) * PLD + 32, bt2, lane);
        tile_store(sB + (row * 16) * PLD + 48, bt3, lane);
        wave_fence();
        lds_count(s_bdone, lane);
    }
    PANEL_STAMP(3);
    PANEL_PROGRESS(22);
    if (!is_diag) {
        // ---- the workgroup's own rows: X = B' Lkk^-T (B' = B - P_o P_t^T, above), 16 columns at a time.  Tasks in dependency
        // order, column block c:  M(rt, c), rt = 0..3 (c > 0): the solved columns onto tile (rt, c)  [the solves < c, row c of the factor]
        //                         T(c): the 16-wide solve of all 64 rows, one row per lane           [M(., c), chol(c)]
        // pulled from s_task by whichever wave is free.  A wave that waits for a dependency waits for a task pulled earlier
        // or for the factor chain, which waits for none of this: no deadlock.
        lds_wait_ge(s_bdone, 4, s_abort, 3);
        PANEL_ACC_DECL
        for (;;) {
            if (worker && !(pa.flags & 1)) worker_yield(&s_flag[0], w, s_abort);
            // every lane adds one (the compiler folds that into ONE ds_add of 64 and a per-lane offset): no value flows out
            // of an `if (lane == 0)` -- with the pull under such a branch the loop was compiled into a nest of exec-masked
            // loops that re-ran a pulled task without pulling the next one
            const int task = __builtin_amdgcn_readfirstlane(
                __hip_atomic_fetch_add(s_task, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) >> 6;
            PANEL_PROGRESS(100 + task);
            if (task >= 16) break;
            // task 0: T(0); column block c > 0: 1 + 5 (c - 1) .. (4 M, T)
            const int c = task < 1 ? 0 : 1 + (task - 1) / 5;
            const int k = task < 1 ? 4 : (task - 1) % 5;
            PANEL_ACC_BEGIN;
            if (k < 4) {                                    // M(k, c)
                const int rt = k;
                lds_wait_ge(&s_flag[5], c, s_abort, 5);                // X[:, 0 .. 16 c) final
                lds_wait_ge(&s_flag[1 + c], c, s_abort, 6);            // L[c][0 .. c) final
                double* T = sB + (rt * 16) * PLD + c * 16;
                v4d acc = tile_load(T, lane);
                acc = mfma_nt16<true>(acc, sB + (rt * 16) * PLD, sD + (c * 16) * PLD, c * 16, lane);
                tile_store(T, acc, lane);
                wave_fence();
                lds_count(&s_mdone[c], lane);
                PANEL_ACC_END(1);
            } else {                                        // T(c)
                if (c > 0) lds_wait_ge(&s_mdone[c], 4, s_abort, 7);
                lds_wait_ge(&s_flag[0], c + 1, s_abort, 9);
                trsm16_rows<64>(sB + c * 16, sD + (c * 16) * PLD + c * 16, sRd + c * 16, lane);
                wave_fence();
                lds_publish(&s_flag[5], c + 1, lane);
                PANEL_ACC_END(2);
            }
        }
        PANEL_ACC_FLUSH;
    }
    PANEL_STAMP(4);
    PANEL_PROGRESS(40);
    __syncthreads();
    PANEL_STAMP(5);
    PANEL_PROGRESS(41);
    if ((pa.flags & 64) && is_diag && t == 0) *s_abort = 0x7E;      // test hook ("panel_debug" = 64): as if a bounded wait had run out
    if (pa.flags & 64) __syncthreads();
    if (*s_abort != 0) {            // a bounded wait ran out (never seen; kept so that a protocol error cannot hang the GPU)
        if (t == 0) { atomicCAS(info, 0, (int)(pa.col0 + j0 + 1 <= n_real ? pa.col0 + j0 + 1 : n_real)); info[3] = 0x5A00 + *s_abort; }
        return;
    }
    if (s_bad) {
        if (is_diag && t == 0) {
            int64_t col = pa.col0 + j0 + s_bad;             // 1-based failing column of the whole matrix
            atomicCAS(info, 0, (int)(col <= n_real ? col : n_real));
        }
        return;
    }
    if (is_diag) {
        if (t == 0) {
            int n = 0;
            while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(8);
                if (++n > PANEL_SPIN_CAP) { atomicCAS(info, 0, (int)(pa.col0 + j0 + 1 <= n_real ? pa.col0 + j0 + 1 : n_real)); info[3] = 0x5AFF; break; }
            }
        }
        __syncthreads();
        store_block(A + j0 * ld + j0, ld, sD, t, true);
        PANEL_STAMP(6);
        PANEL_PROGRESS(60);
        return;
    }
    store_block(Aown + R * ld + j0, ld, sB, t, false);
    PANEL_STAMP(6);
    PANEL_PROGRESS(61);
}

// ---------------------------------------------------------------------------------------------
// The same tile for the compact launches, 64 k at a time: the operand images
// (64 rows x 64 k each, row stride 66 doubles: conflict-free fragment reads) take 68 KB of LDS, so TWO tile workgroups share a
// CU and one's LDS-DMA runs under the other's MFMAs (until round 4 both operands of all 128 k sat in LDS at once, 133 KB: one
// workgroup per CU, which waited for its own DMA).  The eight waves own a 16 x 32 piece each (1 x 2 MFMA tiles).  Per element
// the continuation of the element's one MFMA chain (k ascending), as in syrk64_tile_body: bit-identical.
#define S64C 66
__device__ __forceinline__ void syrk64_tile_compact(double* __restrict__ A, int64_t ld, const TileItem it, double* smem,
                                                 const int* info, const int64_t aug_delta) {
    if (*info != 0) return;
    double* sA = smem;
    double* sB = smem + 64 * S64C;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // (half: bit 0: the visit ends with the half panel; bit 1: the tile and its row operand lie in the appended matrix of
    // potrf_stacked, aug_delta doubles behind A -- a_off, c_off and ha_off count from there)
    const bool appended = (it.half & 2) != 0, with_half = (it.half & 1) != 0;
    double* __restrict__ Ar = appended ? A + aug_delta : A;
    const bool diag = !appended && it.a_off == it.b_off;
    const int wr = w >> 1, wc = w & 1, r = lane & 15, g = lane >> 4;
    // C/D fragment: row = g + 4q, col = r.  The 8 old values of this lane stay in registers over the visit.
    double* cbase = Ar + it.c_off + (int64_t)(wr * 16 + g) * ld + wc * 32 + r;
    v4d acc[2];         // the element's one MFMA chain, resumed at -C (see syrk64_tile_body)
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int q = 0; q < 4; q++) acc[ni][q] = -cbase[(int64_t)(4 * q) * ld + ni * 16];
    if (diag) sB = sA;
    const double* pa = sA + (wr * 16 + r) * S64C + g;
    const double* pb = sB + (wc * 32 + r) * S64C + g;
    // A lagging tile takes several pending panels in one visit, in order.
    const int nvisit = (int)it.n + (with_half ? 1 : 0);
    bool first = true;
#pragma unroll 1
    for (int u = 0; u < nvisit; u++) {
        const bool half = u >= (int)it.n;       // the last visit of a tile with `half`: 64 k instead of 128
        const double* Ag = half ? Ar + it.ha_off : Ar + it.a_off + (int64_t)u * 128;
        const double* Bg = half ? A + it.hb_off : A + it.b_off + (int64_t)u * 128;
        const int kd = half ? 64 : 128;
#pragma unroll 1
        for (int kc = 0; kc < kd; kc += 64) {       // k ascending, 64 at a time
            if (!first) __syncthreads();        // everybody has read the previous images
            first = false;
            // LDS-DMA: the lower half of a wave moves one 512-byte row (64 k) of an operand straight into its padded LDS row;
            // the 8 (+8) rows of a wave are in flight at once
            if (lane < 32) {
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int rw = w * 8 + i;
                    gd_dma16(Ag + (int64_t)rw * ld + kc + 2 * lane, sA + rw * S64C);
                }
                if (!diag) {
#pragma unroll
                    for (int i = 0; i < 8; i++) {
                        const int rw = w * 8 + i;
                        gd_dma16(Bg + (int64_t)rw * ld + kc + 2 * lane, sB + rw * S64C);
                    }
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
#pragma unroll 8
            for (int k0 = 0; k0 < 64; k0 += 4) {
                const double a0 = pa[k0], b0 = pb[k0], b1 = pb[16 * S64C + k0];
                acc[0] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b0, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f64_16x16x4f64(a0, b1, acc[1], 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int ni = 0; ni < 2; ni++)
#pragma unroll
        for (int q = 0; q < 4; q++) cbase[(int64_t)(4 * q) * ld + ni * 16] = -acc[ni][q];
}

// Fused step: the first P workgroups are the panel step, the others each take one 64 x 64 tile of an
// EARLIER panel's trailing update.  The panel chain is one workgroup's latency and leaves most of the GPU
// idle; the tiles fill it.  In-order launches on one stream: no cross-stream events, and a panel workgroup
// (135 KB of LDS) never waits behind tile workgroups for a CU because it comes first in the dispatch order.
__global__ __launch_bounds__(512) void chol_fused_kernel(PanelArgs pa, const TileItem* __restrict__ items, int P) {
    __shared__ __attribute__((aligned(16))) double smem[PANEL_SMEM_DOUBLES];
    const int bx = (int)blockIdx.x;
    const int tb = (int)blockIdx.z;             // theta of a batched launch (gpry_ctx::bn)
    if (bx < P) {
        panel_step_body(pa, smem, bx, tb);
    } else {
        syrk64_tile_body(bset(pa.A, tb, pa.bstride), pa.ld, items[bx - P], smem, bset(pa.info, tb, pa.bstride), pa.aug_delta);
    }
}

// The compact form of the same launch (batched evaluations): 68 KB of LDS and 110 VGPRs, two workgroups per CU.
__global__ __launch_bounds__(512) void chol_fused_compact_kernel(PanelArgs pa, const TileItem* __restrict__ items, int P) {
    __shared__ __attribute__((aligned(16))) double smem[PANEL_COMPACT_SMEM_DOUBLES];
    // Workgroups are dispatched in the order of their index, x first, then z.  With the thetas in z the panel workgroups of the last
    // thetas came behind all tiles of the thetas before them -- six rounds of the GPU's workgroup slots in a step of 32 thetas with
    // 81 tiles each -- and a launch lasted its tiles PLUS one panel latency (32 thetas at N = 1024: 24 + 1.05 us per tile and theta,
    // tools/r05/gpu_trace_batch.sh).  The index is therefore taken apart again: first the panel workgroups of all thetas (one
    // workgroup's latency each, and everything else waits for them), then the tiles, item by item for all thetas (the plan has
    // the longest visits first).  32 thetas: 1.88 -> 1.80 ms at N = 1024, 6.04 -> 5.83 at 1600, 9.78 -> 9.45 at 2048.
    const int bn = (int)gridDim.z;
    const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.z;
    if (lin < P * bn) {
        const int tb = lin / P;                 // theta of a batched launch (gpry_ctx::bn)
        panel_step_compact(pa, smem, lin - tb * P, tb);
    } else {
        const int k = lin - P * bn, ii = k / bn, tb = k - ii * bn;
        syrk64_tile_compact(bset(pa.A, tb, pa.bstride), pa.ld, items[ii], smem, bset(pa.info, tb, pa.bstride), pa.aug_delta);
    }
}

// ---------------------------------------------------------------------------------------------
// Host side.  A = L L^T in place (lower; the strict upper triangle is left untouched), panel steps of 64 columns.
// The factorisation is a sequence of SEGMENTS, each a chain of `ncols` strips on the trailing block that starts at its first
// column (nrows x nrows, leading dimension Np):
//   * up to Np = LARGE_TAIL one segment: the whole matrix;
//   * beyond that, outer blocks of up to HEAD_BLOCK columns first -- behind each of them ONE MFMA SYRK launch applies the block to
//     everything right of it (where the trailing matrix is large these fill the GPU; the trailing matrix is read and written
//     once per block) -- and the last LARGE_TAIL columns, an independent factorisation of the updated trailing block, as the
//     final segment (its SYRK launches would be one 35-us tile latency each).
// Inside a segment every step applies the strip before its own (64 k, left-looking); what a strip needs from the older strips
// of its segment reaches it as 64 x 64 tiles -- riding in the panel launches by a deadline-driven plan (potrf_lower_overlap,
// the default), or as launches of their own (potrf_lower_fused, "chol_overlap" = 0: the comparator).  Same updates, same
// order, same arithmetic per element in both: bit-identical factors.
// Round 4: potrf 6.65 -> 5.46 ms at Np = 8192, 4.96 -> 4.1 at 7168 (segments; in-block updates as riding tiles instead of
// up to three extra chunks in the panel steps), 1.70 -> 1.54 at 4096 (half-panel riding): profiles/r04_potrf.md.
// C[r0:, c0:c0+nc] -= A[r0:, K0:K0+kdepth] A[c0:c0+nc, K0:K0+kdepth]^T on the n x n matrix at A (leading dimension ld); lower
// tiles only when the block is square on the diagonal
static int trailing_update(gpry_ctx* ctx, double* A, int64_t ld, int64_t n, int64_t K0, int64_t r0, int64_t c0, int64_t nc, int kdepth) {
    if (r0 >= n || nc <= 0) return 0;
    GemmArgs g = {};
    g.A = A + r0 * ld + K0; g.lda = ld;
    g.B = A + c0 * ld + K0; g.ldb = ld;
    g.C = A + r0 * ld + c0; g.ldc = ld;
    g.M = (int)(n - r0); g.N = (int)nc; g.K = kdepth;
    // (tiles above the diagonal of a block that starts on it are never read: skipped, for a block column as for the square)
    g.kmode = KM_FULL; g.lower_only = (r0 == c0) ? 1 : 0; g.tile_map = TM_ROWMAJOR; g.info = ctx->dinfo;
    return gemm_f64_launch(ctx, g, false, true, EPI_SUB);
}
struct ChainState { int arrivals = 0; };
// one panel step of the n x n block at A (columns col0.. of the whole matrix), tiles riding along
// (n_aug > 0: the appended matrix of potrf_stacked at A + aug_delta contributes its first n_aug row blocks to this step)
static int panel_launch(gpry_ctx* ctx, ChainState& cs, double* A, int64_t ld, int64_t n, int64_t col0, int64_t j0, int64_t Kfrom,
                        const TileItem* items, int n_items, int n_aug = 0, int64_t aug_delta = 0) {
    const int n_top = (int)((n - j0) / 64);
    const int P = n_top + n_aug;
    cs.arrivals += P;
    static const int env_flags = getenv("GPRY_PANEL_FLAGS") ? atoi(getenv("GPRY_PANEL_FLAGS")) : 0;
    const int panel_flags = env_flags | ctx->opt_panel_debug;
    PanelArgs pa = {A, ld, j0, Kfrom, ctx->N, col0, ctx->dinfo, ctx->dinfo + 2, cs.arrivals, ctx->bstride, panel_flags,
                    n_aug > 0 ? n_top : (1 << 30), aug_delta};
    // Two forms of the step, the same operations on every tile in the same order (bit-identical factors; every theta of a
    // batch is compared with a single evaluation in tests/test_lml_batch_gpu.py).  A single evaluation takes the one whose
    // workgroups have a CU to themselves (135 KB of LDS): beside a second workgroup -- riding tiles of the same launch, GEMMs
    // of the side stream -- the FP64 vector instructions of the chain queue behind the other's FP64 MFMAs (the compact form
    // alone: potrf + 1-2 %, the pipelined chain + 4 %).  A batched evaluation (thetas x panel rows workgroups per step)
    // takes the compact one, two workgroups per CU: 32 thetas at N = 1024 1.99 -> 1.86 ms, N = 2048 10.35 -> 9.72 ms
    // (profiles/r05_potrf.md).  GPRY_PANEL_FLAGS bit 4 / bit 5: the compact / the roomy form for everything (experiments).
    const bool compact = (panel_flags & 16) ? true : (panel_flags & 32) ? false : ctx->bn > 1;
    if (compact)
        hipLaunchKernelGGL(chol_fused_compact_kernel, dim3((unsigned)(P + n_items), 1, (unsigned)ctx->bn), dim3(512), 0, ctx->stream, pa, items, P);
    else
        hipLaunchKernelGGL(chol_fused_kernel, dim3((unsigned)(P + n_items), 1, (unsigned)ctx->bn), dim3(512), 0, ctx->stream, pa, items, P);
    return trtri_pipeline_step(ctx, (int)((col0 + j0) / 64) + 1);
}

static const int64_t LARGE_TAIL = 3584, HEAD_BLOCK = 768;       // grid of both in profiles/r04_potrf.md
struct Segment { int64_t K0; int nrows, ncols; int first_launch; int naug = 0; bool aug_dense = false; };     // strips of 64; first_launch: index into the plan's per-launch lists;
                                                                                       // naug: row blocks of the appended matrix (potrf_stacked);
                                                                                       // aug_dense: without use of their zero structure (the comparator)
// (tail / block: LARGE_TAIL / HEAD_BLOCK for a single evaluation -- the latency optimum --, "tp_tail" / "tp_block" for the
// throughput schedule of gpry_lml_batch, where the thetas of a call fill the GPU and what counts is that the multiply-adds run
// in 128 x 128 tiles of the SYRK engine rather than in riding 64 x 64 tiles, which are bound by the L2.  Since every element is
// ONE chain of MFMAs whatever launch its pieces ride in -- syrk64_tile_body, gemm_dma_body.h -- the cut changes no bit.)
static std::vector<Segment> segments_of(int64_t Np, int64_t tail = LARGE_TAIL, int64_t block = HEAD_BLOCK) {
    std::vector<Segment> seg;
    int64_t K0 = 0;
    int launches = 0;
    if (Np > tail) {
        // the fewest outer blocks of at most `block` columns that leave at most `tail`, all of the same width
        const int64_t head = Np - tail, nblk = (head + block - 1) / block;
        const int64_t ob = round_up((head + nblk - 1) / nblk, 128);
        for (int64_t b = 0; b < nblk; b++, K0 += ob) {
            seg.push_back({K0, (int)((Np - K0) / 64), (int)(ob / 64), launches});
            launches += (int)(ob / 64);
        }
    }
    seg.push_back({K0, (int)((Np - K0) / 64), (int)((Np - K0) / 64), launches});
    return seg;
}

// The comparator of a segment: behind the two strips (2p, 2p + 1) of a panel, strip 2p goes onto strip 2p + 2 (64 k) and the whole
// panel (128 k) onto every strip right of that, one column strip per launch past the first (the riding tiles never touch a
// tile above the diagonal; a rectangular launch over several strips would)
static int separate_chain(gpry_ctx* ctx, ChainState& cs, double* A, int64_t ld, const Segment& sg) {
    const int64_t n = (int64_t)sg.nrows * 64;
    for (int c = 0; c < sg.ncols; c += 2) {
        const int64_t K0 = (int64_t)c * 64;
        for (int s = 0; s < 2 && c + s < sg.ncols; s++) {
            const int64_t j0 = K0 + 64 * s;
            GPRY_TRY(panel_launch(ctx, cs, A, ld, n, sg.K0, j0, j0 >= 64 ? j0 - 64 : 0, nullptr, 0));
        }
        if (c + 2 < sg.ncols) GPRY_TRY(trailing_update(ctx, A, ld, n, K0, K0 + 128, K0 + 128, 64, 64));
        if (sg.ncols == sg.nrows) {
            if (c + 3 < sg.ncols) GPRY_TRY(trailing_update(ctx, A, ld, n, K0, K0 + 192, K0 + 192, n - (K0 + 192), 128));
        } else {
            for (int cc = c + 3; cc < sg.ncols; cc++) GPRY_TRY(trailing_update(ctx, A, ld, n, K0, (int64_t)cc * 64, (int64_t)cc * 64, 64, 128));
        }
    }
    return 0;
}
// behind an outer block: the whole block onto everything right of it
static int block_update(gpry_ctx* ctx, double* A, int64_t ld, const Segment& sg) {
    if (sg.ncols == sg.nrows) return 0;
    const int64_t n = (int64_t)sg.nrows * 64, ob = (int64_t)sg.ncols * 64;
    return trailing_update(ctx, A, ld, n, 0, ob, ob, n - ob, (int)ob);
}

int potrf_lower_fused(gpry_ctx* ctx, double* A, int64_t Np) {
    if (!ctx->info_cleared) HIP_TRY(ctx, hipMemsetAsync(ctx->dinfo, 0, 4 * sizeof(int), ctx->stream));
    ctx->info_cleared = false;
    ChainState cs;
    for (const Segment& sg : segments_of(Np)) {
        double* As = A + sg.K0 * Np + sg.K0;
        GPRY_TRY(separate_chain(ctx, cs, As, Np, sg));
        GPRY_TRY(block_update(ctx, As, Np, sg));
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Riding tiles of a segment.  One launch per 64-column strip c (launch c factors strip c); a panel of the trailing update =
// the 128 columns of two strips (2p, 2p + 1), cut into 64 x 64 tiles (r, c).  Strip c of block bc = c / 2 receives, in this
// order,
//   * the panels p <= bc - 2 (c even) / p <= bc - 1 (c odd) as riding tiles, 128 k per visit, in any launch after the
//     panel is complete and before launch c, most urgent first (slack = launches left - updates left), one round of the CUs
//     the panel step leaves free per launch; a tile far from its deadline waits until two panels are pending and applies
//     both in one visit, C staying in registers in between;
//   * c even: the FIRST half of panel bc - 1 (strip c - 2, 64 k) as a riding tile in launch c - 1, the launch that
//     factors the second half (round 4; until then the panel step of launch c applied all 128 columns of panel bc - 1
//     itself, 17.6k cycles of its ~ 55k, on the panel chain);
//   * the strip before it (c - 1) by the panel step of launch c itself: left-looking, 64 k -- every step the same.
// A tile is visited by ONE workgroup per launch (pending panels and the half in the same visit).  Only the `ncols` strips of
// the segment receive tiles (an outer block: its SYRK launch takes care of the rest), rows run to the end of the matrix.  The
// plan depends on Np only and is cached on the device.
struct OverlapPlan {
    int64_t Np = 0;
    TileItem* d_items = nullptr;
    std::vector<Segment> seg;
    std::vector<int> first, count;      // per launch: slice of d_items
};
// plans per context: [0] the factorisation alone, [1] with the inverse factor as appended rows (potrf_stacked), [2] the same
// without use of the zero structure of the appended rows (its comparator), [3] the factorisation alone in the column blocks
// of the throughput schedule
struct OverlapPlans { OverlapPlan p[4]; };
void overlap_plan_free(gpry_ctx* ctx) {
    OverlapPlans* pl = static_cast<OverlapPlans*>(ctx->chol_plan);
    if (!pl) return;
    for (int i = 0; i < 4; i++) if (pl->p[i].d_items) (void)hipFree(pl->p[i].d_items);
    delete pl;
    ctx->chol_plan = nullptr;
}
// appends the launches of one segment; returns false if a deadline cannot be met
static bool plan_segment(const Segment& sg, int64_t ld, int ncu, std::vector<TileItem>& items, std::vector<int>& first, std::vector<int>& count) {
    // Rows nrows .. nrows + naug - 1 are the row blocks of the appended matrix (potrf_stacked: the identity, which the chain
    // turns into L^-T).  Appended row block a is zero left of column block a: its tiles exist from column a on, the panels left
    // of strip a leave them as they are (never visited), and its panel workgroup joins the chain at step a.
    // (aug_dense: the appended rows as ordinary rows -- every panel, every step, zeros included: what the skipping is compared with)
    const int ntop = sg.nrows, naug = sg.naug, nrows = ntop + naug, ncols = sg.ncols;
    const bool skip = !sg.aug_dense;
    std::vector<int> done((size_t)nrows * ncols, 0), last((size_t)nrows * ncols, -1);
    std::vector<char> halfdone((size_t)nrows * ncols, 0);
    for (int a = 0; skip && a < naug; a++)
        for (int c = a; c < ncols; c++) {
            const int bc = c / 2, need = (c & 1) ? bc : bc - 1;
            const int p0 = a / 2;            // whole panels that are all zero in row block a (a half-zero panel is applied: it adds exact zeros)
            done[(size_t)(ntop + a) * ncols + c] = need < 0 ? 0 : p0 < need ? p0 : need;
            if (!(c & 1) && c >= 2 && a >= c - 1) halfdone[(size_t)(ntop + a) * ncols + c] = 1;      // strip c - 2 is zero in this row block
        }
    struct Cand { int slack, c, r, p, n, half; };
    std::vector<Cand> cand;
    auto item = [&](int r, int c, int p, int nn, int half) {
        TileItem it;
        const int rr = r >= ntop ? r - ntop : r;                 // (appended rows count from the start of their own matrix: TileItem)
        it.a_off = (int64_t)rr * 64 * ld + (int64_t)p * 128;     // 64 rows from tile row r, the 128 columns of panel p
        it.b_off = (int64_t)c * 64 * ld + (int64_t)p * 128;
        it.c_off = (int64_t)rr * 64 * ld + (int64_t)c * 64;
        it.ha_off = (int64_t)rr * 64 * ld + (int64_t)(c - 2) * 64;    // the 64 columns of strip c - 2
        it.hb_off = (int64_t)c * 64 * ld + (int64_t)(c - 2) * 64;
        it.n = nn; it.half = (half ? 1 : 0) | (r >= ntop ? 2 : 0);
        return it;
    };
    const int multi = 2;                         // panels per visit of a lagging tile
    // tile rounds (of the CUs the panel step leaves free) per launch: with every panel step the same length, one round each
    // (measured at N = 4096: 1570 us with 1 / 1, 1601 with 2 / 1, 1625 with 2 / 2, 1632 with 3 / 2; no difference up to 2048)
    for (int l = 0; l < ncols; l++) {
        const int b = l / 2;
        const int P = (ntop - l) + (skip && naug > l + 1 ? l + 1 : naug);      // panel workgroups of launch l
        const int cap = ncu > P ? ncu - P : 0;
        cand.clear();
        // columns not yet factored: c >= 2b (+1 in the block's second launch: its first 64 columns are done)
        for (int c = 2 * b + (l & 1) > 2 ? 2 * b + (l & 1) : 2; c < ncols; c++) {
            const int bc = c / 2;
            const int need = (c & 1) ? bc : bc - 1;              // whole panels that ride
            // c even: the first half of panel bc - 1 (strip c - 2) rides in launch c - 1, behind the whole panels
            const bool half_now = !(c & 1) && l == c - 1;
            for (int r = c; r < nrows; r++) {
                if (skip && r >= ntop && r - ntop > c) continue;         // (zero tile of the appended matrix)
                const int p = done[(size_t)r * ncols + c];
                if (last[(size_t)r * ncols + c] >= l) continue;
                if (half_now && halfdone[(size_t)r * ncols + c]) continue;       // (appended rows whose strip c - 2 is zero)
                if (half_now) {
                    // everything this strip still waits for goes into ONE visit: the pending whole panels, then the half
                    const int nn = need - p;                     // (all of them are complete: p < need <= bc - 1 <= b)
                    cand.push_back({0, c, r, p, nn, 1});
                    continue;
                }
                if (p >= need || p > b - 1) continue;
                const int avail = (need < b ? need : b) - p;     // panels p .. p + avail - 1 are complete and wanted
                int slack;
                if ((c & 1) && p == bc - 1) slack = (2 * bc) - l;                           // must run in launch 2 bc
                else slack = (2 * bc - l) - ((bc - 1) - p);                                  // older updates: before block bc
                int nn = avail < multi ? avail : multi;
                // far tiles wait until `multi` panels are pending (fewer, longer visits); near ones cannot
                if (nn < multi && slack > 2 * multi) continue;
                cand.push_back({slack, c, r, p, nn, 0});
            }
        }
        std::sort(cand.begin(), cand.end(), [](const Cand& x, const Cand& y) {
            if (x.slack != y.slack) return x.slack < y.slack;
            if (x.c != y.c) return x.c < y.c;
            return x.r < y.r;
        });
        first.push_back((int)items.size());
        int n_taken = 0;
        for (const Cand& q : cand) {
            if (n_taken >= cap && q.slack > 1) continue;          // not urgent and the launch is full
            items.push_back(item(q.r, q.c, q.p, q.n, q.half));
            done[(size_t)q.r * ncols + q.c] = q.p + q.n; last[(size_t)q.r * ncols + q.c] = l;
            if (q.half) halfdone[(size_t)q.r * ncols + q.c] = 1;
            n_taken++;
        }
        count.push_back(n_taken);
        // longest visits first: when a launch has more tiles than free CUs, the workgroups that start late are the short ones
        std::stable_sort(items.end() - n_taken, items.end(), [](const TileItem& x, const TileItem& y) {
            return 2 * x.n + (x.half ? 1 : 0) > 2 * y.n + (y.half ? 1 : 0);
        });
        // what the NEXT launch's panel step reads must be complete now
        const int cnext = l + 1;                                 // the 64-column strip factored next
        if (cnext >= 2 && cnext < ncols) {
            const int bc = cnext / 2, need = (cnext & 1) ? bc : bc - 1;
            for (int r = cnext; r < nrows; r++) {
                if (skip && r >= ntop && r - ntop > cnext) continue;
                if (done[(size_t)r * ncols + cnext] != need || (!(cnext & 1) && !halfdone[(size_t)r * ncols + cnext])) return false;
            }
        }
    }
    return true;
}
// returns 1 if no valid plan exists (the caller takes the schedule with separate trailing launches)
static int overlap_plan_get(gpry_ctx* ctx, int64_t Np, OverlapPlan** out, int stacked = 0) {       // stacked: 0 no, 1 yes, 2 yes with dense appended rows, 3: no, throughput schedule
    if (!ctx->chol_plan) ctx->chol_plan = new OverlapPlans();
    OverlapPlan& pl = static_cast<OverlapPlans*>(ctx->chol_plan)->p[stacked];
    if (pl.Np == Np) { *out = &pl; return 0; }
    if (pl.d_items) { (void)hipFree(pl.d_items); pl.d_items = nullptr; }
    pl = OverlapPlan();
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    std::vector<TileItem> items;
    pl.seg = stacked == 3 ? segments_of(Np, ctx->opt_tp_tail, ctx->opt_tp_block) : segments_of(Np);
    if (stacked == 1 || stacked == 2) {
        if (pl.seg.size() != 1) { pl = OverlapPlan(); return 1; }      // (one segment: Np <= LARGE_TAIL)
        pl.seg[0].naug = pl.seg[0].nrows;
        pl.seg[0].aug_dense = stacked == 2;
    }
    for (const Segment& sg : pl.seg)
        if (!plan_segment(sg, Np, ncu, items, pl.first, pl.count)) { pl = OverlapPlan(); return 1; }
    if (!items.empty()) {
        hipError_t e = hipMalloc((void**)&pl.d_items, items.size() * sizeof(TileItem));
        if (e == hipSuccess) e = hipMemcpy(pl.d_items, items.data(), items.size() * sizeof(TileItem), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            if (pl.d_items) (void)hipFree(pl.d_items);
            pl = OverlapPlan();
            return gpry_fail(ctx, -2, "Cholesky overlap plan (Np = %lld): %s", (long long)Np, hipGetErrorString(e));
        }
    }
    pl.Np = Np;
    *out = &pl;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// THE INVERSE FACTOR AS EXTRA ROWS OF THE PANEL CHAIN (late in round 5; Np <= "chol_stacked").  The panel step solves
// X L(j,j)^T = B for every row block below the diagonal block.  Append the identity to the matrix -- [K; I], 2 Np x Np -- and the
// appended rows come out as U with U L^T = I: U = L^-T = V^T, the reference's own formulation of the inverse factor
// (solve_triangular(L, I), gpry/gpr.py:1456-1457), computed by the launches of the factorisation itself: the same panel step,
// the same riding tiles, the same deadline plan, batched launches included.  The recursive V = L^-1 behind potrf -- eleven
// dependent launches at N = 1024, 0.13 of an evaluation's 0.49 ms; 0.26 of 0.83 ms at N = 2048 -- is gone; the extra N^3 / 6
// multiply-adds ride in the four fifths of the GPU that the panel chain leaves idle.
//   * The appended block is a matrix of its own (`U`, same leading dimension): PanelArgs::aug_delta / TileItem::half bit 1.
//   * Its zero structure is used: row block a of I is zero left of column block a, so its panel workgroup joins the chain at
//     step a and its tiles skip the panels left of strip a (plan_segment).  The caller has written the identity (its tiles on and
//     right of the diagonal and the one left of each diagonal tile: the others are never read).
//   * The caller transposes U into the lower-triangular V that every consumer reads (transpose_upper_launch below).
// V differs from the recursive inverse by rounding (another summation order; the same in every schedule that takes this path:
// single, batched).  L is the factor of the other schedules bit for bit.
int potrf_stacked(gpry_ctx* ctx, double* A, double* U, int64_t Np) {
    // "chol_stacked_dense" = 1: the comparator -- the appended rows as ordinary rows (all of them in every step, every panel
    // applied, zeros included).  Skipping exact zeros changes no bit: tests/test_hip_parity.py compares the two.
    const bool dense = ctx->opt_chol_stacked_dense != 0;
    OverlapPlan* pl = nullptr;
    const int prc = overlap_plan_get(ctx, Np, &pl, dense ? 2 : 1);
    if (prc) return prc;            // (1: no plan -- the caller takes the other path)
    if (!ctx->info_cleared) HIP_TRY(ctx, hipMemsetAsync(ctx->dinfo, 0, 4 * sizeof(int), ctx->stream));
    ctx->info_cleared = false;
    ChainState cs;
    const Segment& sg = pl->seg[0];
    const int64_t n = (int64_t)sg.nrows * 64;
    for (int c = 0; c < sg.ncols; c++) {
        const int64_t j0 = (int64_t)c * 64;
        GPRY_TRY(panel_launch(ctx, cs, A, Np, n, 0, j0, j0 >= 64 ? j0 - 64 : 0, pl->d_items + pl->first[c], pl->count[c],
                              !dense && c + 1 < sg.naug ? c + 1 : sg.naug, (int64_t)(U - A)));
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
bool potrf_stacked_usable(const gpry_ctx* ctx, int64_t Np) {
    // (the throughput schedule of gpry_lml_batch takes the recursive inverse: in a batch its products are plain GEMM launches that
    // fill the GPU, while the appended rows double the panel workgroups of every step -- profiles/r06_batch.md)
    return ctx->opt_chol == 0 && ctx->opt_chol_overlap && Np <= ctx->opt_chol_stacked && Np <= LARGE_TAIL && Np >= 128 && !ctx->tp;
}

// V <- U^T behind potrf_stacked: V lower triangular with zeros above, as the recursive inverse leaves it.  64 x 64 tiles
// through LDS; thetas of a batched launch in z.  (U starts as the identity: written by the covariance build, kernel_build.hip.)
__global__ __launch_bounds__(256) void transpose_upper_kernel(const double* __restrict__ U_, double* __restrict__ V_, int64_t ld, int64_t bstride) {
    __shared__ double tile[64][65];
    const double* __restrict__ U = bset(U_, (int)blockIdx.z, bstride);
    double* __restrict__ V = bset(V_, (int)blockIdx.z, bstride);
    const int nb = (int)(ld / 64);
    const int bi = (int)blockIdx.x / nb, bj = (int)blockIdx.x - bi * nb;        // tile (bi, bj) of V
    const int t = threadIdx.x;
    if (bj > bi) {                                                             // above the diagonal: zeros
        for (int e = t; e < 64 * 32; e += 256) {
            const int i = e >> 5, j2 = e & 31;
            *reinterpret_cast<double2*>(V + ((int64_t)bi * 64 + i) * ld + (int64_t)bj * 64 + 2 * j2) = make_double2(0.0, 0.0);
        }
        return;
    }
    for (int e = t; e < 64 * 32; e += 256) {                                   // tile (bj, bi) of U
        const int i = e >> 5, j2 = e & 31;
        const double2 v = *reinterpret_cast<const double2*>(U + ((int64_t)bj * 64 + i) * ld + (int64_t)bi * 64 + 2 * j2);
        tile[i][2 * j2] = v.x; tile[i][2 * j2 + 1] = v.y;
    }
    __syncthreads();
    for (int e = t; e < 64 * 32; e += 256) {
        const int i = e >> 5, j2 = e & 31;
        double a = tile[2 * j2][i], b = tile[2 * j2 + 1][i];
        if (bi == bj) { if (2 * j2 > i) a = 0.0; if (2 * j2 + 1 > i) b = 0.0; }
        *reinterpret_cast<double2*>(V + ((int64_t)bi * 64 + i) * ld + (int64_t)bj * 64 + 2 * j2) = make_double2(a, b);
    }
}
int transpose_upper_launch(gpry_ctx* ctx, const double* U, double* V, int64_t Np) {
    const unsigned nb = (unsigned)(Np / 64);
    hipLaunchKernelGGL(transpose_upper_kernel, dim3(nb * nb, 1, (unsigned)ctx->bn), dim3(256), 0, ctx->stream, U, V, Np, ctx->bstride);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int potrf_lower_overlap(gpry_ctx* ctx, double* A, int64_t Np) {
    OverlapPlan* pl = nullptr;
    // (the column blocks pay from three thetas per chain on; one or two take the cut of the single evaluation -- the factor is
    // the same, bit for bit, either way: profiles/r06_tp.md)
    const int prc = overlap_plan_get(ctx, Np, &pl, (ctx->tp && ctx->bn >= 3) || ctx->opt_chol_tp_segments ? 3 : 0);
    if (prc == 1) return potrf_lower_fused(ctx, A, Np);
    if (prc) return prc;
    if (!ctx->info_cleared) HIP_TRY(ctx, hipMemsetAsync(ctx->dinfo, 0, 4 * sizeof(int), ctx->stream));
    ctx->info_cleared = false;
    ChainState cs;
    // "tp_left" = 1 (comparator; measured 1.5 % behind at N = 4096, equal at 2048: profiles/r06_tp.md): the column blocks LEFT-looking
    // -- in front of a block ONE launch brings its columns up to date with ALL columns left of it (k from 0: deep tiles, C read
    // and written once) instead of one launch behind every block onto everything right of it (k = the block's width, C read and
    // written once per block).  The element's chain is the same chain, k ascending from the covariance entry: same bits.
    const bool left = ((ctx->tp && ctx->bn >= 3) || ctx->opt_chol_tp_segments) && ctx->opt_tp_left;
    for (const Segment& sg : pl->seg) {
        double* As = A + sg.K0 * Np + sg.K0;
        const int64_t n = (int64_t)sg.nrows * 64;
        if (left && sg.K0 > 0)
            GPRY_TRY(trailing_update(ctx, A, Np, Np, 0, sg.K0, sg.K0, sg.ncols == sg.nrows ? n : (int64_t)sg.ncols * 64, (int)sg.K0));
        for (int c = 0; c < sg.ncols; c++) {
            const int64_t j0 = (int64_t)c * 64;
            const int l = sg.first_launch + c;
            // every step applies the strip before its own (64 k, left-looking) itself
            GPRY_TRY(panel_launch(ctx, cs, As, Np, n, sg.K0, j0, j0 >= 64 ? j0 - 64 : 0, pl->d_items + pl->first[l], pl->count[l]));
        }
        if (!left) GPRY_TRY(block_update(ctx, As, Np, sg));
    }
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

#ifdef GPRY_PROTOTYPES
// PROTOTYPE (tools/r05/build_proto.sh; not part of the product library): the inverse factor as extra rows of the panel chain.
// The stacked matrix [K; I] (2 Np x Np, one allocation) goes through ONE rectangular segment of the overlap schedule; its
// bottom block comes out as U = L^-T.  No use of the zero structure of I (the appended rows are treated as dense: three times
// the multiply-adds of a triangular inverse) -- this measures what the existing machinery gives, DESIGN.md section 7 (iii).
extern "C" int gpry_proto_potrf_stacked(gpry_ctx* ctx, const double* K_host, int64_t Np, double* L_host, double* U_host, int reps,
                                        double* ms_stacked, double* ms_square) {
    if (Np % 64) return gpry_fail(ctx, -1, "prototype: Np must be a multiple of 64");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int nb = (int)(Np / 64);
    double *dA = nullptr, *dIn = nullptr;
    HIP_TRY(ctx, hipMalloc((void**)&dA, sizeof(double) * 2 * Np * Np));
    HIP_TRY(ctx, hipMalloc((void**)&dIn, sizeof(double) * 2 * Np * Np));
    {
        std::vector<double> h((size_t)(2 * Np * Np), 0.0);
        memcpy(h.data(), K_host, sizeof(double) * Np * Np);
        for (int64_t i = 0; i < Np; i++) h[(size_t)((Np + i) * Np + i)] = 1.0;
        HIP_TRY(ctx, hipMemcpy(dIn, h.data(), sizeof(double) * 2 * Np * Np, hipMemcpyHostToDevice));
    }
    int ncu = 256;
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ncu = prop.multiProcessorCount; }
    hipEvent_t e0, e1;
    HIP_TRY(ctx, hipEventCreate(&e0)); HIP_TRY(ctx, hipEventCreate(&e1));
    for (int form = 0; form < 2; form++) {           // 0: the stacked matrix, 1: the square one alone
        Segment sg = {0, form == 0 ? 2 * nb : nb, nb, 0};
        std::vector<TileItem> items; std::vector<int> first, count;
        if (!plan_segment(sg, Np, ncu, items, first, count)) return gpry_fail(ctx, -1, "prototype: no plan");
        TileItem* d_items = nullptr;
        if (!items.empty()) {
            HIP_TRY(ctx, hipMalloc((void**)&d_items, items.size() * sizeof(TileItem)));
            HIP_TRY(ctx, hipMemcpy(d_items, items.data(), items.size() * sizeof(TileItem), hipMemcpyHostToDevice));
        }
        double total = 0.0;
        for (int rep = 0; rep < reps + 1; rep++) {
            HIP_TRY(ctx, hipMemcpyAsync(dA, dIn, sizeof(double) * 2 * Np * Np, hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemsetAsync(ctx->dinfo, 0, 4 * sizeof(int), ctx->stream));
            HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
            ChainState cs;
            for (int c = 0; c < sg.ncols; c++) {
                const int64_t j0 = (int64_t)c * 64;
                GPRY_TRY(panel_launch(ctx, cs, dA, Np, (int64_t)sg.nrows * 64, 0, j0, j0 >= 64 ? j0 - 64 : 0, d_items + first[c], count[c]));
            }
            HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            float ms = 0.f;
            HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
            if (rep > 0) total += ms;
        }
        *(form == 0 ? ms_stacked : ms_square) = total / reps;
        if (form == 0) {
            HIP_TRY(ctx, hipMemcpy(L_host, dA, sizeof(double) * Np * Np, hipMemcpyDeviceToHost));
            HIP_TRY(ctx, hipMemcpy(U_host, dA + Np * Np, sizeof(double) * Np * Np, hipMemcpyDeviceToHost));
        }
        if (d_items) (void)hipFree(d_items);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    (void)hipFree(dA); (void)hipFree(dIn);
    return 0;
}
#endif
