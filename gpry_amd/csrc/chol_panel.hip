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

#define PLD 66

__device__ __forceinline__ void wave_fence() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// acc += sign * A[16 x K] * B[16 x K]^T ; A, B point at row 0 of their 16-row blocks in LDS
template <bool NEG>
__device__ __forceinline__ v4d mfma_nt16(v4d acc, const double* A, const double* B, int K, int lane) {
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
        double nlij = -lij;
        // x[c] -= L[i][j] * L[c][j]: the factor of row c is lane c of every 16-lane DPP row (the four
        // rows of the wave are replicas), fused into the multiply-add as a row_newbcast operand
        asm volatile("s_nop 1" : "+v"(lij), "+v"(nlij));        // VALU write -> DPP read of the same VGPR
#pragma unroll
        for (int c = j + 1; c < 16; c++) fmac_row_bcast(x[c], lij, nlij, c);   // rows i < c: entries nobody reads
        x[j] = i > j ? lij : 0.0;                               // the diagonal entry follows below
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

// X[:, 16c .. 16c+15] of the 64-row block sB against the factor in sD, by ONE wave: the update with
// the columns solved so far (MFMA) and the 16-wide substitution with one row per lane.  Runs on a
// wave that would otherwise idle while another wave factors the next 16x16 diagonal block.
__device__ __forceinline__ void solve_block_cols(double* sB, const double* sD, const double* sRd, int c, int lane) {
    if (c) {
#pragma unroll 1
        for (int rt = 0; rt < 4; rt++) {
            double* T = sB + (rt * 16) * PLD + c * 16;
            v4d acc = tile_load(T, lane);
            acc = mfma_nt16<true>(acc, sB + (rt * 16) * PLD, sD + (c * 16) * PLD, c * 16, lane);
            tile_store(T, acc, lane);
        }
        wave_fence();
    }
    trsm16_rows<64>(sB + c * 16, sD + (c * 16) * PLD + c * 16, sRd + c * 16, lane);
    wave_fence();
}

// 64x64 block global -> LDS in two phases: all eight 16-byte loads of a thread are in flight before
// the first LDS store (a load/store pair per iteration paid one memory round trip each: the four
// blocks of a panel step cost 13k cycles that way)
__device__ __forceinline__ void load_block_issue(const double* __restrict__ G, int64_t ld, int t, double2 (&r)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e = t + 256 * i, row = e >> 5, c2 = (e & 31) * 2;
        r[i] = *reinterpret_cast<const double2*>(G + (int64_t)row * ld + c2);
    }
}
__device__ __forceinline__ void load_block_commit(double* S, int t, const double2 (&r)[8]) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int e = t + 256 * i, row = e >> 5, c2 = (e & 31) * 2;
        *reinterpret_cast<double2*>(S + row * PLD + c2) = r[i];
    }
}
__device__ __forceinline__ void store_block(double* __restrict__ G, int64_t ld, const double* S, int t,
                                            bool lower_only) {
    for (int e = t; e < 64 * 32; e += 256) {
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
__global__ __launch_bounds__(256) void trtri_diag64_kernel(const double* __restrict__ L, double* __restrict__ V,
                                                           int64_t ld, const int* info) {
    __shared__ __attribute__((aligned(16))) double sL[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sV[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sVt[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sTt[64 * PLD];
    if (*info != 0) return;
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
}
int launch_trtri_diag(gpry_ctx* ctx, const double* L, double* V, int64_t Np, hipStream_t st) {
    hipLaunchKernelGGL(trtri_diag64_kernel, dim3((unsigned)(Np / 64)), dim3(256), 0, st, L, V, Np, ctx->dinfo);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// `arrive` / `target`: the diagonal workgroup overwrites D with its factor in place, while every
// other workgroup of the launch reads D.  Workgroups count in on `arrive` once their loads have
// landed, and the diagonal workgroup stores only when all of them have (target = arrivals
// expected up to and including this launch).  Without this the result depends on all workgroups
// starting before the first one finishes -- not true when another stream shares the GPU.
__global__ __launch_bounds__(256) void chol_panel_kernel(double* __restrict__ A, int64_t ld, int64_t j0,
                                                         int64_t K0, int64_t n_real, int* info,
                                                         int* arrive, int target,
                                                         unsigned long long* dbg, int dbg_block) {
    __shared__ __attribute__((aligned(16))) double sD[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sB[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sPt[64 * PLD];
    __shared__ __attribute__((aligned(16))) double sPo[64 * PLD];
    __shared__ double sRd[64];
    __shared__ int s_bad;
    __shared__ int s_flag[8];
    if (*info != 0) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const bool is_diag = blockIdx.x == 0;
    // optional section stamps of one workgroup (dbg != NULL): loads, update, factor, solve, store
    const bool stamp = dbg != nullptr && (int)blockIdx.x == dbg_block && t == 0;
    unsigned long long ts[6] = {0, 0, 0, 0, 0, 0};
    if (stamp) ts[0] = __builtin_amdgcn_s_memtime();
    const int64_t R = j0 + 64 * (int64_t)blockIdx.x;
    const int kprev = (int)(j0 - K0);          // columns of the outer block already factorised: 0, 64, 128, ...
    if (t == 0) s_bad = 0;
    {
        // all four blocks unconditionally (the diagonal workgroup has R == j0 and the first step of an
        // outer block K0 == j0: those loads repeat D and land in buffers nobody reads): with the loads
        // behind branches the 3 x 128-byte register sets went through scratch memory
        double2 rD[8], rB[8], rPt[8], rPo[8];
        load_block_issue(A + j0 * ld + j0, ld, t, rD);
        load_block_issue(A + R * ld + j0, ld, t, rB);
        load_block_issue(A + j0 * ld + K0, ld, t, rPt);
        load_block_issue(A + R * ld + K0, ld, t, rPo);
        load_block_commit(sD, t, rD);
        load_block_commit(sB, t, rB);
        load_block_commit(sPt, t, rPt);
        load_block_commit(sPo, t, rPo);
    }
    __syncthreads();
    if (stamp) ts[1] = __builtin_amdgcn_s_memtime();
    // ---- left-looking update with the previous columns of the outer block, 64 at a time (the first
    // chunk came in with the loads above; an outer block of 256 columns has up to three)
    for (int c0 = 0; c0 < kprev; c0 += 64) {
        if (c0) {
            __syncthreads();                       // everybody is done with the previous chunk
            double2 rPt[8], rPo[8];
            load_block_issue(A + j0 * ld + K0 + c0, ld, t, rPt);
            if (!is_diag) load_block_issue(A + R * ld + K0 + c0, ld, t, rPo);
            load_block_commit(sPt, t, rPt);
            if (!is_diag) load_block_commit(sPo, t, rPo);
            __syncthreads();
        }
        // D: only the ten 16x16 tiles on and below the diagonal are ever read (the factor works on the
        // lower triangle); they are dealt round-robin to the waves (3, 3, 2, 2) instead of a full row each
#pragma unroll 1
        for (int tl = w; tl < 10; tl += 4) {
            const int rw = tl < 1 ? 0 : tl < 3 ? 1 : tl < 6 ? 2 : 3;
            const int n = tl - rw * (rw + 1) / 2;
            double* T = sD + (rw * 16) * PLD + n * 16;
            v4d acc = tile_load(T, lane);
            acc = mfma_nt16<true>(acc, sPt + (rw * 16) * PLD, sPt + (n * 16) * PLD, 64, lane);
            tile_store(T, acc, lane);
        }
        if (!is_diag) {
#pragma unroll
            for (int n = 0; n < 4; n++) {
                double* U = sB + (w * 16) * PLD + n * 16;
                v4d acb = tile_load(U, lane);
                acb = mfma_nt16<true>(acb, sPo + (w * 16) * PLD, sPt + (n * 16) * PLD, 64, lane);
                tile_store(U, acb, lane);
            }
        }
    }
    if (t == 0) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (stamp) ts[2] = __builtin_amdgcn_s_memtime();
    // ---- Cholesky of the 64x64 diagonal block, blocked by 16, as a dataflow between the four waves
    // (wave w owns block row w) instead of three workgroup barriers per block column:
    //   for cb < w:  wait chol(cb);  T(w,cb): D[w][cb] <- D[w][cb] L[cb][cb]^-T;  publish;
    //                for cc in cb+1..w: (cc < w: wait T(cc,cb))  D[w][cc] -= D[w][cb] D[cc][cb]^T
    //   chol(w); publish;  then (w < 3, off-diagonal workgroups) the own-row solve of column block w.
    // Wave cb+1 starts chol(cb+1) as soon as ITS row is done, while the rows below still work on block
    // column cb: the chain is 4 chol16 + 3 (solve + one tile update) = ~35k cycles instead of 44k.
    // Same operations on every tile in the same order as the barrier version: bit-identical factors.
    // Flags are LDS words written by lane 0 after a wave fence (LDS requests of a wave retire in order).
    if (w == 0 && lane < 8) s_flag[lane] = 0;      // [0]: blocks factored, [1 + w]: columns solved by wave w, [5]: own-row blocks solved
    __syncthreads();
    {
        for (int cb = 0; cb < w; cb++) {
            while (__hip_atomic_load(&s_flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= cb) __builtin_amdgcn_s_sleep(1);
            trsm16_rows(sD + (w * 16) * PLD + cb * 16, sD + (cb * 16) * PLD + cb * 16, sRd + cb * 16, lane);
            wave_fence();
            if (lane == 0) __hip_atomic_store(&s_flag[1 + w], cb + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int cc = cb + 1; cc <= w; cc++) {
                if (cc < w)
                    while (__hip_atomic_load(&s_flag[1 + cc], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= cb) __builtin_amdgcn_s_sleep(1);
                double* T = sD + (w * 16) * PLD + cc * 16;
                v4d acc = tile_load(T, lane);
                acc = mfma_nt16<true>(acc, sD + (w * 16) * PLD + cb * 16, sD + (cc * 16) * PLD + cb * 16, 16, lane);
                tile_store(T, acc, lane);
            }
            wave_fence();
        }
        // a failed pivot (not positive definite) still publishes: nobody may wait forever; the first
        // failing column wins (the chol16 calls are ordered by the chain itself)
        const int bad = chol16_wave(sD + (w * 16) * PLD + w * 16, sRd + w * 16, lane);
        if (bad && lane == 0 && s_bad == 0) s_bad = w * 16 + bad;
        wave_fence();
        if (lane == 0) __hip_atomic_store(&s_flag[0], w + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        if (!is_diag && w < 3) {
            // this wave has nothing left to do in the factor: it solves the workgroup's own rows against
            // its block column (needs the own-row blocks 0..w-1, solved by the waves before it)
            while (__hip_atomic_load(&s_flag[5], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) < w) __builtin_amdgcn_s_sleep(1);
            solve_block_cols(sB, sD, sRd, w, lane);
            if (lane == 0) __hip_atomic_store(&s_flag[5], w + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __syncthreads();
    if (s_bad) {
        if (is_diag && t == 0) {
            int64_t col = j0 + s_bad;                       // 1-based failing column
            atomicCAS(info, 0, (int)(col <= n_real ? col : n_real));
        }
        return;
    }
    if (stamp) ts[3] = __builtin_amdgcn_s_memtime();
    if (is_diag) {
        if (t == 0)
            while (__hip_atomic_load(arrive, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target)
                __builtin_amdgcn_s_sleep(8);
        __syncthreads();
        if (stamp) ts[4] = __builtin_amdgcn_s_memtime();
        store_block(A + j0 * ld + j0, ld, sD, t, true);
        if (stamp) {
            ts[5] = __builtin_amdgcn_s_memtime();
            for (int i = 0; i < 5; i++) atomicAdd(&dbg[i], ts[i + 1] - ts[i]);
            atomicAdd(&dbg[5], 1ull);
        }
        return;
    }
    // ---- X = B Lkk^-T: the column blocks 0..2 were solved inside the factor loop (by the waves
    // idling there); the last one is done here, wave w on its own 16-row strip
    {
        const int cb = 3;
        double* T = sB + (w * 16) * PLD + cb * 16;
        v4d acc = tile_load(T, lane);
        acc = mfma_nt16<true>(acc, sB + (w * 16) * PLD, sD + (cb * 16) * PLD, cb * 16, lane);
        tile_store(T, acc, lane);
        wave_fence();
        trsm16_rows(T, sD + (cb * 16) * PLD + cb * 16, sRd + cb * 16, lane);
        wave_fence();
    }
    __syncthreads();
    if (stamp) ts[4] = __builtin_amdgcn_s_memtime();
    store_block(A + R * ld + j0, ld, sB, t, false);
    if (stamp) {
        ts[5] = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 5; i++) atomicAdd(&dbg[i], ts[i + 1] - ts[i]);
        atomicAdd(&dbg[5], 1ull);
    }
}

// A = L L^T in place (lower; the strict upper triangle is left untouched).  Outer blocks of
// 128 columns: two fused panel steps, then one MFMA SYRK (K = 128) on the trailing matrix.
// Trailing update C -= P P^T (lower tiles only) of the rows/cols [r0, Np) x [c0, c0 + nc) with the
// 128-column panel P = A[:, K0:K0+128].
static int trailing_update(gpry_ctx* ctx, double* A, int64_t Np, int64_t K0, int64_t r0, int64_t c0,
                           int64_t nc, hipStream_t st, int kdepth = 128) {
    if (r0 >= Np || nc <= 0) return 0;
    GemmArgs g = {};
    g.A = A + r0 * Np + K0; g.lda = Np;
    g.B = A + c0 * Np + K0; g.ldb = Np;
    g.C = A + r0 * Np + c0; g.ldc = Np;
    g.M = (int)(Np - r0); g.N = (int)nc; g.K = kdepth;
    g.kmode = KM_FULL; g.lower_only = 1; g.tile_map = TM_ROWMAJOR; g.info = ctx->dinfo; g.stream = st;
    g.extra_lds = ctx->opt_syrk_lds;
    return gemm_f64_launch(ctx, g, false, true, EPI_SUB);
}

int potrf_lower_fused(gpry_ctx* ctx, double* A, int64_t Np) {
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemsetAsync(ctx->dinfo, 0, 4 * sizeof(int), st));
    int arrivals = 0;
    // Look-ahead: after panel k only the next panel's 128 columns of the trailing matrix are
    // updated on the main stream; the rest of the update runs on stream2 underneath panel k+1
    // (the panel chain is latency-bound and leaves the machine empty).  Every element still
    // receives its rank-128 updates in the same order from the same kernel: bit-identical.
    // Outer block: the trailing matrix is read and written once per outer block, and a trailing update
    // costs >= 40 us however small it is, so wider blocks halve both; the panel steps pay for it with
    // up to three extra 64-column chunks in their left-looking update.  With the DMA-pipelined
    // trailing update (36 us for a lone tile) the crossover is at Np ~ 6144 (tools/prof_factor.py N d reps ab:
    // 128 / 256 columns: 1.16 / 1.23 ms at 2048, 2.61 / 2.71 at 4096, 4.80 / 4.80 at 6144, 8.12 / 7.84 at
    // 8192); the look-ahead schedule keeps 128.
    const bool la = ctx->opt_chol_lookahead && ctx->stream2 != nullptr && Np > 512;
    const int64_t OB = la ? 128 : (ctx->opt_chol_outer > 0 ? ctx->opt_chol_outer : (Np > 6144 ? 256 : 128));
    bool rest_pending = false;
    if (la) {
        const size_t need = 2 * (size_t)(Np / 128);
        while (ctx->ev_pool.size() < need) {
            hipEvent_t ev;
            HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->ev_pool.push_back(ev);
        }
    }
    int step = 0;
    hipEvent_t ev_rest_prev = nullptr;
    for (int64_t K0 = 0; K0 < Np; K0 += OB, step++) {
        const int64_t ob = (Np - K0 < OB) ? Np - K0 : OB;
        for (int64_t j0 = K0; j0 < K0 + ob; j0 += 64) {
            unsigned nblk = (unsigned)((Np - j0) / 64);
            arrivals += (int)nblk;
            unsigned long long* dbg = nullptr;
            if (ctx->opt_chol_dbg) { if (!ctx->dsel) GPRY_TRY(dev_alloc(ctx, &ctx->dsel, 64)); dbg = ctx->dsel + 16; }
            hipLaunchKernelGGL(chol_panel_kernel, dim3(nblk), dim3(256), 0, st, A, Np, j0, K0, ctx->N, ctx->dinfo,
                               ctx->dinfo + 2, arrivals, dbg, ctx->opt_chol_dbg - 1);
        }
        const int64_t r0 = K0 + ob;
        if (r0 >= Np) break;
        if (!la) {
            GPRY_TRY(trailing_update(ctx, A, Np, K0, r0, r0, Np - r0, st, (int)ob));
            continue;
        }
        hipEvent_t ev_panel = ctx->ev_pool[2 * step], ev_rest = ctx->ev_pool[2 * step + 1];
        HIP_TRY(ctx, hipEventRecord(ev_panel, st));                            // panel k done
        if (rest_pending) HIP_TRY(ctx, hipStreamWaitEvent(st, ev_rest_prev, 0));      // rest(k-1) done
        GPRY_TRY(trailing_update(ctx, A, Np, K0, r0, r0, 128, st));            // next panel's columns
        if (r0 + 128 < Np) {
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream2, ev_panel, 0));
            GPRY_TRY(trailing_update(ctx, A, Np, K0, r0 + 128, r0 + 128, Np - r0 - 128, ctx->stream2));
            HIP_TRY(ctx, hipEventRecord(ev_rest, ctx->stream2));
            ev_rest_prev = ev_rest;
            rest_pending = true;
        }
    }
    if (rest_pending) HIP_TRY(ctx, hipStreamWaitEvent(st, ev_rest_prev, 0));
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
