// Kernel algebra shared by the covariance kernels (kernel_build.hip) and the resident predict kernel (server.hip):
// parameter blocks, correlation functions restated from sklearn:kernels.py:1553-1580 (RBF), :1708-1768 (Matern),
// :1239-1291 (Constant), and the per-slice posterior mean of one point.
#pragma once
#include "common.h"

struct KernParams {
    double C;
    int d, dpad, has_aff;
    int64_t N;
};
struct AffParams {
    double ls[GPRY_MAX_DIM];     // length scales (unused slots = 1)
    double lo[GPRY_MAX_DIM];
    double span[GPRY_MAX_DIM];
};

#define SQRT3 1.7320508075688772
#define SQRT5 2.23606797749979

template <int KID>
__device__ __forceinline__ double corr_r2(double r2) {
    if (KID == GPRY_RBF) return exp(-0.5 * r2);
    if (KID == GPRY_MATERN12) return exp(-sqrt(r2));
    if (KID == GPRY_MATERN32) { double t = sqrt(r2) * SQRT3; return (1.0 + t) * exp(-t); }
    double t = sqrt(r2) * SQRT5;
    return (1.0 + t + t * t * (1.0 / 3.0)) * exp(-t);
}
// Straight-line sqrt / exp for the cross-kernel panel (4.1e9 evaluations per 1e6 candidates at
// N = 4096: the panel build is VALU-bound, and libm's versions carry range checks, denormal
// scaling and ~10 register moves each).  Valid for the arguments that occur here: x >= 0 not
// denormal; t >= 0.  Accuracy ~1 ulp.
__device__ __forceinline__ double fast_sqrt_pos(double x) {
    const double s = __builtin_amdgcn_rsq(x);
    double g = x * s, h = 0.5 * s;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    double e = fma(-g, g, x);
    g = fma(e, h, g);
    e = fma(-g, g, x);
    g = fma(e, h, g);
    return x == 0.0 ? 0.0 : g;
}
__device__ __forceinline__ double fast_exp_neg(double t) {      // exp(-t)
    const double y = -t;
    const double n = __builtin_rint(y * 1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, y);
    r = fma(n, -1.90821492927058770002e-10, r);
    // exp(r), |r| <= ln2/2: Taylor to degree 13 (truncation 4e-18 relative)
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    const double nn = fmax(n, -1100.0);      // exp(-t) underflows to 0 through ldexp
    return __builtin_ldexp(p, (int)nn);
}
template <int KID>
__device__ __forceinline__ double corr_r2_fast(double r2) {
    if (KID == GPRY_RBF) return fast_exp_neg(0.5 * r2);
    if (KID == GPRY_MATERN12) return fast_exp_neg(fast_sqrt_pos(r2));
    if (KID == GPRY_MATERN32) { double t = fast_sqrt_pos(r2) * SQRT3; return (1.0 + t) * fast_exp_neg(t); }
    double t = fast_sqrt_pos(r2) * SQRT5;
    return (1.0 + t + t * t * (1.0 / 3.0)) * fast_exp_neg(t);
}
// ---- the correlation from a PRE-SCALED squared distance u = s r^2 (round 5, the MFMA panel of the sweep: 4.1e9
// evaluations per 1e6 candidates at N = 4096, FP64-issue-bound).  s = corr_scale<KID>(): 1/2 (RBF: k = e^-u), 1 (Matern-1/2:
// k = e^-sqrt(u)), 3 and 5 (Matern-3/2, -5/2: t = sqrt(u)); the caller folds s into the norms it adds anyway, which saves
// the multiplication by sqrt(3) / sqrt(5) behind the root.  Against corr_r2_fast per evaluation of Matern-5/2 (vector
// instructions, profiles/r05_kern_math.md): root 12 -> 8 (no zero test: u >= 1e-290 is the caller's clamp, where k = 1 to
// the last bit; one correction step instead of two), polynomial 3 -> 2 (Horner), exponential 20 -> 18 (degree 11 instead
// of 13 on |r| <= ln 2 / 2: interpolant at the Chebyshev nodes, 1.5e-16 relative including the rounding of its Horner form
// -- the Taylor polynomial needs two more terms for that).  Root and exponential within 1 ulp of the correctly rounded
// values on 4e6 log-uniform arguments, like libm's and corr_r2_fast's (tools/r05/kern_math_check.hip); not the same bits.
template <int KID>
__host__ __device__ constexpr double corr_scale() {
    return KID == GPRY_RBF ? 0.5 : KID == GPRY_MATERN12 ? 1.0 : KID == GPRY_MATERN32 ? 3.0 : 5.0;
}
__device__ __forceinline__ double fast_sqrt_nz(double x) {         // x >= 1e-290, not denormal
    const double s = __builtin_amdgcn_rsq(x);
    double g = x * s, h = 0.5 * s;
    const double r = fma(-h, g, 0.5);
    g = fma(g, r, g); h = fma(h, r, h);
    const double e = fma(-g, g, x);
    return fma(e, h, g);        // ONE correction behind the coupled step: <= 1 ulp over 4e6 arguments, as with two (kern_math_check)
}
__device__ __forceinline__ double fast_exp_neg11(double t) {       // exp(-t), t >= 0
    const double n = __builtin_rint(t * -1.4426950408889634074);
    double r = fma(n, -6.93147180369123816490e-01, -t);
    r = fma(n, -1.90821492927058770002e-10, r);
    double p = 2.5107785307679967e-08;
    p = fma(p, r, 2.7632571010826906e-07);
    p = fma(p, r, 2.7557247099675525e-06);
    p = fma(p, r, 2.480148569156966e-05);
    p = fma(p, r, 0.00019841269885093954);
    p = fma(p, r, 0.0013888888952097655);
    p = fma(p, r, 0.008333333333320085);
    p = fma(p, r, 0.04166666666648896);
    p = fma(p, r, 0.16666666666666685);
    p = fma(p, r, 0.5000000000000018);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)fmax(n, -1100.0));             // underflows to 0 through ldexp
}
template <int KID>
__device__ __forceinline__ double corr_scaled_fast(double u) {
    if (KID == GPRY_RBF) return fast_exp_neg11(u);
    const double t = fast_sqrt_nz(u);
    if (KID == GPRY_MATERN12) return fast_exp_neg11(t);
    if (KID == GPRY_MATERN32) return (1.0 + t) * fast_exp_neg11(t);
    return fma(fma(t, 1.0 / 3.0, 1.0), t, 1.0) * fast_exp_neg11(t);
}
// returns k(r) in *kval and h with d k / d log l_k = h * D_k   (both without the factor C)
template <int KID>
__device__ __forceinline__ double corr_and_h(double r2, double* kval) {
    if (KID == GPRY_RBF) { double e = fast_exp_neg(0.5 * r2); *kval = e; return e; }
    if (KID == GPRY_MATERN12) {
        double r = fast_sqrt_pos(r2); double e = fast_exp_neg(r); *kval = e;
        return r != 0.0 ? e / r : 0.0;
    }
    if (KID == GPRY_MATERN32) {
        double t = fast_sqrt_pos(r2) * SQRT3; double e = fast_exp_neg(t);
        *kval = (1.0 + t) * e;
        return 3.0 * e;
    }
    double t = fast_sqrt_pos(r2) * SQRT5; double e = fast_exp_neg(t);
    *kval = (1.0 + t + t * t * (1.0 / 3.0)) * e;
    return (5.0 / 3.0) * (t + 1.0) * e;
}

static KernParams make_kp(gpry_ctx* ctx) {
    KernParams kp;
    kp.C = exp(ctx->theta[0]);
    kp.d = ctx->d; kp.dpad = ctx->dpad;
    kp.has_aff = ctx->tf.has_x_affine; kp.N = ctx->N;
    return kp;
}
static AffParams make_ap(gpry_ctx* ctx, bool use_affine) {
    AffParams ap;
    for (int k = 0; k < GPRY_MAX_DIM; k++) {
        // sklearn divides by the length scale (x / l); we keep the division, not a reciprocal
        // multiply, so that scaled coordinates round identically.
        ap.ls[k] = k < ctx->d ? exp(ctx->theta[1 + k]) : 1.0;
        ap.lo[k] = (use_affine && k < ctx->d) ? ctx->tf.x_lo[k] : 0.0;
        ap.span[k] = (use_affine && k < ctx->d) ? ctx->tf.x_span[k] : 1.0;
    }
    return ap;
}

#define DISPATCH_KID(kid, CALL)                         \
    switch (kid) {                                      \
        case GPRY_RBF: { CALL(GPRY_RBF); break; }       \
        case GPRY_MATERN12: { CALL(GPRY_MATERN12); break; } \
        case GPRY_MATERN32: { CALL(GPRY_MATERN32); break; } \
        default: { CALL(GPRY_MATERN52); break; }        \
    }


// ------------------------------------------------------------------------------------
// One slice of the posterior mean of ONE point: sum over the training rows [row_lo, row_lo + rows_per_split) of
// alpha_j C k(x, X_j) in transformed units, by one workgroup of 256 threads (result valid in thread 0).  `x`: the d
// raw coordinates of the point (any address space the caller can read: mapped host memory, LDS, global).  Shared by
// the one-launch kernel (predict_mean_small_kernel) and the resident one (server.hip): whichever of the two serves
// a call, the bits are the same.
#define MEAN_SLICE_CH 4096        // rows per LDS chunk of the one-launch kernel
// CH: rows per chunk (the size of r2s).  `base`: Xs / alpha_ hold the rows from `base` on (0: the arrays of the context
// in global memory; row_lo: a copy of this slice that the resident kernel keeps in LDS) -- the values and the order of
// every sum are the same either way.
template <int DP, int KID, int CH = MEAN_SLICE_CH>
__device__ __forceinline__ double mean_slice(const double* x, const double* __restrict__ Xs, const double* __restrict__ alpha_,
                                             int64_t row_lo, int64_t rows_per_split, const KernParams& kp, const AffParams& ap,
                                             double* r2s /*[CH]*/, double* red /*[256]*/, int64_t base = 0) {
    constexpr int P = DP / 2;             // lanes per training row: one 16-byte piece each
    const int t = threadIdx.x, sub = t % P, rloc = t / P;
    // this lane's two scaled coordinates of the point
    double x0 = 0.0, x1 = 0.0;
    {
        const int k0 = 2 * sub, k1 = 2 * sub + 1;
        if (k0 < kp.d) { double v = x[k0]; if (kp.has_aff) v = (v - ap.lo[k0]) / ap.span[k0]; x0 = v / ap.ls[k0]; }
        if (k1 < kp.d) { double v = x[k1]; if (kp.has_aff) v = (v - ap.lo[k1]) / ap.span[k1]; x1 = v / ap.ls[k1]; }
    }
    const bool piece_ok = 2 * sub < kp.dpad;
    double acc = 0.0;
    const int64_t row_hi = (row_lo + rows_per_split < kp.N) ? row_lo + rows_per_split : kp.N;
    for (int64_t c0 = row_lo; c0 < row_hi; c0 += CH) {
        const int nrow = (int)((row_hi - c0 < CH) ? row_hi - c0 : CH);
        // phase 1: squared distances, rows read as whole cache lines (P lanes per row); eight
        // passes are loaded back to back so that their memory latencies overlap
        constexpr int RP = 256 / P;       // rows per pass
        for (int r0 = 0; r0 < nrow; r0 += 8 * RP) {
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = r0 + u * RP + rloc;
                v[u] = make_double2(0.0, 0.0);
                if (row < nrow && piece_ok)
                    v[u] = *reinterpret_cast<const double2*>(Xs + (c0 - base + row) * kp.dpad + 2 * sub);
            }
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int row = r0 + u * RP + rloc;
                const double d0 = x0 - v[u].x, d1 = x1 - v[u].y;
                double part = piece_ok ? fma(d1, d1, d0 * d0) : 0.0;
#pragma unroll
                for (int o = 1; o < P; o <<= 1) part += __shfl_xor(part, o);
                if (sub == 0 && row < nrow) r2s[row] = part;
            }
        }
        __syncthreads();
        // phase 2: one row per lane, four independent chains
        for (int j0 = t; j0 < nrow; j0 += 1024) {
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int j = j0 + 256 * q;
                v[q] = j < nrow ? alpha_[c0 - base + j] * (kp.C * corr_r2_fast<KID>(r2s[j])) : 0.0;
            }
            acc += (v[0] + v[1]) + (v[2] + v[3]);
        }
        __syncthreads();
    }
    // The binary tree over the 256 partial sums -- red[t] += red[t + s] for s = 128, 64, ..., 1 -- with ONE barrier: the
    // two cross-wave levels are read back by wave 0, the six levels inside it are cross-lane adds with the same pairs
    // (t, t + s), so the result is the tree's to the last bit (the eight barriers of the loop form were 0.8 us of a
    // request the resident kernel answers in ~3).
    red[t] = acc;
    __syncthreads();
    double out = 0.0;
    if (t < 64) {
        double v = (red[t] + red[t + 128]) + (red[t + 64] + red[t + 192]);
        v += __shfl_down(v, 32);
        v += __shfl_down(v, 16);
        v += __shfl_down(v, 8);
        v += __shfl_down(v, 4);
        v += __shfl_down(v, 2);
        v += __shfl_down(v, 1);
        out = v;
    }
    __syncthreads();          // `red` may be reused by the caller's next point
    return out;
}
