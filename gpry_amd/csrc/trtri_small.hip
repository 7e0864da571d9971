// V_bb = L_bb^-1 for every 128 x 128 diagonal block of L in one launch: first stage of V = L^-1 (gpry/gpr.py:1457) up to
// Np = 1024.  The 64 x 64 stage of chol_panel.hip (trtri_diag64_kernel) leaves the 64 -> 128 level to two GEMM launches;
// at a few hundred training points these are three dependent dispatches (27 us) for what one workgroup per block does in
// its own LDS in 12: the block is loaded once, its eight 16 x 16 diagonal tiles are inverted by forward substitution with
// true divisions (as the 64 x 64 stage: the reciprocal-multiply form is one rounding per entry further from the
// reference's dtrsm), and three doubling levels [[L11, 0], [L21, L22]]^-1 = [[V11, 0], [-V22 (L21 V11), V22]] run IN PLACE
// over the tiles -- the scheme of the single-launch objective (lml_small.hip, phases E and F), whose building blocks
// (chol16.h) this file shares.
#include "common.h"
#include "chol16.h"

#define TS_NP 128
#define TS_LD 130
#define TS_NB 8
#define TS_CLEAR 8      // workgroups that clear the rows of a block right of it (clear_right)

__global__ __launch_bounds__(512) void trtri_diag128_kernel(const double* __restrict__ L_, double* __restrict__ V_,
                                                            int64_t ld, const int* info, int clear_right, int64_t bstride) {
    __shared__ __attribute__((aligned(16))) double M[TS_NP * TS_LD];
    const int tb = (int)blockIdx.z;             // theta of a batched launch (gpry_ctx::bn)
    if (*bset(info, tb, bstride) != 0) return;
    const double* __restrict__ L = bset(L_, tb, bstride);
    double* __restrict__ V = bset(V_, tb, bstride);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nblk = (int)(ld / TS_NP);
    if ((int)blockIdx.x >= nblk) {
        // The rows of a block, columns right of it: stands in for a memset of V (ld = the matrix dimension).  Workgroups of
        // their own, sixteen rows each, beside the ones that invert (until round 5 the inverting workgroup cleared its 128 rows
        // first: 0.9 MB through ONE workgroup at N = 1024 -- 25 of the launch's 37 us on the critical path of every evaluation).
        const int id = (int)blockIdx.x - nblk, b = id / TS_CLEAR, sl = id - b * TS_CLEAR;
        const int64_t c0 = (int64_t)(b + 1) * TS_NP, r0 = (int64_t)b * TS_NP + sl * (TS_NP / TS_CLEAR);
        const int64_t ncol2 = (ld - c0) >> 1;
        const double2 zero2 = make_double2(0.0, 0.0);
        for (int64_t e = t; e < (TS_NP / TS_CLEAR) * ncol2; e += 512) {
            const int64_t i = e / ncol2, j2 = e - i * ncol2;
            *reinterpret_cast<double2*>(V + (r0 + i) * ld + c0 + 2 * j2) = zero2;
        }
        return;
    }
    const int64_t b0 = (int64_t)blockIdx.x * TS_NP;
    for (int e = t; e < TS_NP * (TS_NP / 2); e += 512) {
        const int i = e >> 6, j = (e & 63) * 2;
        const double2 v = *reinterpret_cast<const double2*>(L + (b0 + i) * ld + b0 + j);
        M[i * TS_LD + j] = j <= i ? v.x : 0.0;
        M[i * TS_LD + j + 1] = j + 1 <= i ? v.y : 0.0;
    }
    __syncthreads();
    // diagonal tiles: wave w, lane c < 16: column c of L_ww^-1 by forward substitution (true divisions)
    double winv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) winv[i] = 0.0;
    if (lane < 16) {
        const double* Lw = M + (w * 16) * TS_LD + w * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
            for (int k = 0; k < i; k++) sacc = fma(-Lw[i * TS_LD + k], winv[k], sacc);
            winv[i] = (i >= lane) ? sacc / Lw[i * TS_LD + i] : 0.0;
        }
    }
    __syncthreads();
    if (lane < 16) {
        double* Lw = M + (w * 16) * TS_LD + w * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) Lw[i * TS_LD + lane] = winv[i];
    }
    __syncthreads();
    // doubling over the tiles (h = 1, 2, 4 tiles): T(m, n) = sum_{k = n}^{mid - 1} L(m, k) V11(k, n) replaces L21, then
    // V21(m, n) = -sum_{k = mid}^{m} V22(m, k) T(k, n) replaces T; every output tile is a task, dealt to the eight waves
    // (two per wave on the top level, long and short k-ranges paired), written only when everybody has read
    for (int lg = 0; lg < 3; lg++) {
        const int h = 1 << lg, ntask = (TS_NB >> (lg + 1)) << (2 * lg);
#pragma unroll
        for (int prod = 0; prod < 2; prod++) {
            v4d acc[2];
            double* dst[2] = {nullptr, nullptr};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
                const int id = w + TS_NB * q;
                if (id < ntask) {
                    const int p = id >> (2 * lg), e = id & ((1 << (2 * lg)) - 1);
                    int mi = e >> lg, ni = e & (h - 1);
                    if (q == 1) { ni = h - 1 - ni; mi = (h + h / 2 - 1) - mi; }
                    const int lo = 2 * h * p, mid = lo + h, m = mid + mi, n = lo + ni;
                    dst[q] = M + (m * 16) * TS_LD + n * 16;
                    if (prod == 0)
                        acc[q] = c16::mfma_nn<TS_LD, false>(acc[q], M + (m * 16) * TS_LD + n * 16, M + (n * 16) * TS_LD + n * 16,
                                                            16 * (mid - n), lane);
                    else
                        acc[q] = c16::mfma_nn<TS_LD, true>(acc[q], M + (m * 16) * TS_LD + mid * 16, M + (mid * 16) * TS_LD + n * 16,
                                                           16 * (m - mid + 1), lane);
                }
            }
            __syncthreads();
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (dst[q]) c16::tile_store<TS_LD>(dst[q], acc[q], lane);
            __syncthreads();
        }
    }
    for (int e = t; e < TS_NP * (TS_NP / 2); e += 512) {
        const int i = e >> 6, j = (e & 63) * 2;
        const double a = j <= i ? M[i * TS_LD + j] : 0.0, b = j + 1 <= i ? M[i * TS_LD + j + 1] : 0.0;
        *reinterpret_cast<double2*>(V + (b0 + i) * ld + b0 + j) = make_double2(a, b);
    }
}

int launch_trtri_diag128(gpry_ctx* ctx, const double* L, double* V, int64_t Np, hipStream_t st, bool clear_right) {
    const unsigned nblk = (unsigned)(Np / TS_NP);
    hipLaunchKernelGGL(trtri_diag128_kernel, dim3(clear_right ? nblk * (1 + TS_CLEAR) : nblk, 1, (unsigned)ctx->bn), dim3(512), 0, st, L, V, Np, ctx->dinfo,
                       clear_right ? 1 : 0, ctx->bstride);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
