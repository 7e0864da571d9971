// Log marginal likelihood + gradient of a SMALL training set (N <= 128, d <= 16) in ONE launch of ONE workgroup.
//
// sklearn:_gpr.py:574-652 as reached from gpry/gpr.py:876-881 (the objective of fit_gpr_hyperparameters,
// gpry/gpr.py:883-994).  The first iterations of every GPry run -- and whole runs in low dimension -- live at a few
// dozen to a few hundred training points.  There the chain of the general path (csrc/api.hip: gpry_lml) is 19
// dependent kernels, nine of them at the 4-5 us floor of a dependent dispatch: 173 us per evaluation whatever N is
// (profiles/r02_trace_small_lml.txt).  Here the whole evaluation stays in the LDS of one CU (8 waves):
//
//   A  x / l (true divisions, as scale_train_kernel)                    -> xs[k][row]          16 KB
//   B  K = C k(r) + diag(noise), lower 16 x 16 tiles                    -> M[128][130]        133 KB
//   C  M <- chol(M): dataflow between the waves, wave w owns block row w (chol16_wave / trsm16_rows / MFMA rank-16
//      updates: the building blocks of chol_panel.hip), only the ceil(N / 16) blocks that hold training rows
//   D  log L_ii;  E  diagonal blocks W_jj = L_jj^-1 (column substitution with the factor's reciprocal pivots), by the
//      wave that has just factored block j, into registers, while the chain goes on; stored after the barrier
//   F  V = L^-1 IN PLACE by recursive doubling over the tiles: T = L21 V11 replaces L21, V21 = -V22 T replaces T
//   G  z = V y, quad = z.z, alpha = V^T z
//   H  per lower 16 x 16 tile: K^-1 tile = sum_k V[k][i] V[k][j] (MFMA, accumulators only: K^-1 is never stored),
//      W = alpha alpha^T - K^-1 contracted at once with dK/dtheta (distances recomputed from xs, as lml_traces_kernel)
//   I  results + factorisation status as stamped 16-byte units into the mapped host buffer (LsUnit below)
//
// One row stride (130 doubles) keeps the row-wise MFMA fragment reads conflict-free and the k-wise ones 2-way.  The
// factor does NOT travel back to HBM: gpry_factorize cannot adopt it (lml_cache stays false) and
// runs the general chain once per fit -- the prediction factor keeps its operation order (profiles/HISTORY.md section 4.2).
#include "kern_math.h"
#include "chol16.h"

#define LS_NP 128
#define LS_LD 130
#define LS_NT 512
#define LS_NB 8

// Results travel as self-validating 16-byte units {value, stamp}, each written by ONE store instruction (one PCIe write
// inside a cache line), so that the host may pick them up the moment they land instead of waiting for the end-of-kernel
// signal: round 2 measured that wait at 15-18 us per evaluation, and learned that a status word written "last" does not
// arrive last (inbound writes to different cache lines are not ordered on this platform, profiles/HISTORY.md section 4.5).  No
// order between the units is assumed: the host reads a value only from a unit whose stamp is this evaluation's.
struct LsUnit { unsigned long long payload, stamp; };
#define LS_INFO_UNIT 40
typedef unsigned int ls_u4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void ls_store_unit(LsUnit* p, double value, unsigned long long stamp) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(value);
    ls_u4 v = {(unsigned)b, (unsigned)(b >> 32), (unsigned)stamp, (unsigned)(stamp >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" :: "v"(p), "v"(v) : "memory");
}

struct LmlSmallArgs {
    const double* X;        // N x d transformed training rows (ctx->dX)
    const double* y;        // Np (zero padded)
    const double* noise;    // Np
    LsUnit* host_res;       // mapped host memory, 16-byte units {value, stamp}: [0] sum log L_ii, [1] quad, [2 .. 2 + d] gradient,
                            // [LS_INFO_UNIT] factorisation status (0 or the 1-based failing column, as a double)
    unsigned long long seq; // stamp of this evaluation
    int want_grad;
    const double* batch;       // nullable: several thetas in ONE launch, workgroup b evaluates [C, l_1 .. l_16] = batch[17 b ..] (the values
                               // the host computes for a single evaluation: same bits) and reports into host_res + LS_UNITS * b
};
#define LS_UNITS (LS_INFO_UNIT + 1)

template <int DP, int KID>
__global__ __launch_bounds__(LS_NT) void lml_small_kernel(LmlSmallArgs a, KernParams kp, AffParams ap) {
    __shared__ __attribute__((aligned(16))) double M[LS_NP * LS_LD];
    __shared__ __attribute__((aligned(16))) double xs[DP * LS_NP];      // k-major: xs[k * 128 + row]
    __shared__ double sy[LS_NP], sz[LS_NP], sal[LS_NP], srd[LS_NP], slog[LS_NP];
    __shared__ double red[LS_NB][DP + 2];
    __shared__ double red2[4][LS_NP];
    __shared__ int s_flag[16];
    __shared__ int s_bad;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int N = (int)kp.N;
    if (a.batch != nullptr) {      // this workgroup's theta and result units (the restarts of a fit evaluated side by side)
        const double* bp = a.batch + 17 * (int64_t)blockIdx.x;
        kp.C = bp[0];
#pragma unroll
        for (int k = 0; k < DP; k++) ap.ls[k] = bp[1 + k];
        a.host_res += LS_UNITS * (int64_t)blockIdx.x;
    }
    const int nb = (N + 15) >> 4;            // 16-row blocks that hold training rows (the others are identity padding)

    // ---- A: scaled coordinates, targets
#pragma unroll
    for (int k = 0; k < DP; k++) {
        if (t < LS_NP) {
            double v = 0.0;
            if (t < N && k < kp.d) v = a.X[(int64_t)t * kp.d + k] / ap.ls[k];
            xs[k * LS_NP + t] = v;
        }
    }
    if (t < LS_NP) { sy[t] = a.y[t]; srd[t] = 1.0; slog[t] = 0.0; }
    if (t < 16) s_flag[t] = 0;
    if (t == 0) s_bad = 0;
    __syncthreads();

    // ---- B: covariance matrix, lower tiles (tile list: (bi, bj), bj <= bi), MFMA C-layout per wave:
    // lane (g, r) owns rows 16 bi + g + 4q (q = 0..3), column 16 bj + r
    for (int tl = w; tl < LS_NB * (LS_NB + 1) / 2; tl += LS_NB) {
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= tl) bi++;
        const int bj = tl - bi * (bi + 1) / 2;
        const int j = 16 * bj + r;
        double r2[4] = {0.0, 0.0, 0.0, 0.0};
        if (bi < nb) {
#pragma unroll
            for (int k = 0; k < DP; k++) {
                const double xj = xs[k * LS_NP + j];
#pragma unroll
                for (int q = 0; q < 4; q++) { const double df = xs[k * LS_NP + 16 * bi + g + 4 * q] - xj; r2[q] = fma(df, df, r2[q]); }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = 16 * bi + g + 4 * q;
            double v = kp.C * corr_r2<KID>(r2[q]);               // libm exp / sqrt, as the training build of the general path
            if (i == j) v = kp.C + (i < N ? a.noise[i] : 0.0);
            if (i >= N || j >= N) v = (i == j) ? 1.0 : 0.0;      // identity padding
            M[i * LS_LD + j] = v;
        }
    }
    __syncthreads();

    // ---- C: Cholesky, blocked by 16, as a dataflow between the waves (wave w owns block row w).  A wave that has
    // factored its diagonal block is done with the chain: it inverts that block (E) into registers while the chain
    // goes on -- the tile itself is still read by the rows below and is replaced after the barrier.
    double winv[16];
#pragma unroll
    for (int i = 0; i < 16; i++) winv[i] = 0.0;
    if (w < nb) {
        for (int cb = 0; cb < w; cb++) {
            while (__hip_atomic_load(&s_flag[0], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= cb) __builtin_amdgcn_s_sleep(1);
            c16::trsm16_rows<LS_LD>(M + (w * 16) * LS_LD + cb * 16, M + (cb * 16) * LS_LD + cb * 16, srd + cb * 16, lane);
            c16::wave_fence();
            if (lane == 0) __hip_atomic_store(&s_flag[1 + w], cb + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
            for (int cc = cb + 1; cc <= w; cc++) {
                if (cc < w)
                    while (__hip_atomic_load(&s_flag[1 + cc], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= cb) __builtin_amdgcn_s_sleep(1);
                double* T = M + (w * 16) * LS_LD + cc * 16;
                v4d acc = c16::tile_load<LS_LD>(T, lane);
                acc = c16::mfma_nt<LS_LD, true>(acc, M + (w * 16) * LS_LD + cb * 16, M + (cc * 16) * LS_LD + cb * 16, 16, lane);
                c16::tile_store<LS_LD>(T, acc, lane);
            }
            c16::wave_fence();
        }
        // a failed pivot (not positive definite) still publishes: nobody may wait forever
        const int bad = c16::chol16_wave<LS_LD>(M + (w * 16) * LS_LD + w * 16, srd + w * 16, lane);
        if (bad && lane == 0) atomicMin(&s_bad, 0 - (1 << 20) + w * 16 + bad);      // first failing column wins
        c16::wave_fence();
        if (lane == 0) __hip_atomic_store(&s_flag[0], w + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        // ---- D: log-determinant terms;  E: W_ww = L_ww^-1, column `lane` by forward substitution (reciprocal pivots)
        if (lane < 16) {
            const double* Lw = M + (w * 16) * LS_LD + w * 16;
            const int idx = w * 16 + lane;
            slog[idx] = idx < N ? log(Lw[lane * LS_LD + lane]) : 0.0;
#pragma unroll
            for (int i = 0; i < 16; i++) {
                double sacc = (i == lane) ? 1.0 : 0.0;
#pragma unroll
                for (int k = 0; k < i; k++) sacc = fma(-Lw[i * LS_LD + k], winv[k], sacc);
                winv[i] = (i >= lane) ? sacc * srd[w * 16 + i] : 0.0;
            }
        }
    }
    __syncthreads();
    if (s_bad != 0) {      // sklearn:_gpr.py:586-589: the caller turns info > 0 into (-inf, 0)
        if (t == 0) {
            int col = s_bad + (1 << 20);                        // 1-based failing column
            if (col > N) col = N;
            ls_store_unit(a.host_res + LS_INFO_UNIT, (double)col, a.seq);
        }
        return;
    }
    if (w < nb && lane < 16) {
        double* Lw = M + (w * 16) * LS_LD + w * 16;
#pragma unroll
        for (int i = 0; i < 16; i++) Lw[i * LS_LD + lane] = winv[i];
    }
    __syncthreads();

    // ---- F: V = L^-1 in place by recursive doubling over the 16 x 16 tiles: [[L11, 0], [L21, L22]]^-1 =
    // [[V11, 0], [-V22 (L21 V11), V22]].  Per level (block size h = 1, 2, 4 tiles) and node (lo, mid, hi): first
    // T(m, n) = sum_{k=n}^{mid-1} L(m, k) V11(k, n) replaces L21, then V21(m, n) = -sum_{k=mid}^{m} V22(m, k) T(k, n)
    // replaces T.  Every output tile is a task; the tasks of a product are dealt to the eight waves (two per wave on the
    // top level, long and short k-ranges paired), accumulated in registers, and written only when everybody has read.
    for (int lg = 0; lg < 3; lg++) {
        const int h = 1 << lg, ntask = (LS_NB >> (lg + 1)) << (2 * lg);      // (shifts: h is a run-time value)
#pragma unroll
        for (int prod = 0; prod < 2; prod++) {
            v4d acc[2];
            double* dst[2] = {nullptr, nullptr};
#pragma unroll
            for (int q = 0; q < 2; q++) {
                acc[q] = (v4d){0.0, 0.0, 0.0, 0.0};
                const int id = w + LS_NB * q;
                if (id < ntask) {
                    const int p = id >> (2 * lg), e = id & ((1 << (2 * lg)) - 1);
                    int mi = e >> lg, ni = e & (h - 1);
                    if (q == 1) { ni = h - 1 - ni; mi = (h + h / 2 - 1) - mi; }      // top level: pair long with short
                    const int lo = 2 * h * p, mid = lo + h, m = mid + mi, n = lo + ni;
                    if (m < nb) {
                        dst[q] = M + (m * 16) * LS_LD + n * 16;
                        if (prod == 0)
                            acc[q] = c16::mfma_nn<LS_LD, false>(acc[q], M + (m * 16) * LS_LD + n * 16, M + (n * 16) * LS_LD + n * 16,
                                                                16 * (mid - n), lane);
                        else
                            acc[q] = c16::mfma_nn<LS_LD, true>(acc[q], M + (m * 16) * LS_LD + mid * 16, M + (mid * 16) * LS_LD + n * 16,
                                                               16 * (m - mid + 1), lane);
                    }
                }
            }
            __syncthreads();                       // all reads of this product are done
#pragma unroll
            for (int q = 0; q < 2; q++)
                if (dst[q]) c16::tile_store<LS_LD>(dst[q], acc[q], lane);
            __syncthreads();
        }
    }

    // ---- G: z = V y, alpha = V^T z, quad = z.z, log-determinant
    {   // z: four threads per row, 32 terms each (walking a row with the 64 lanes and a butterfly sum per row was
        // twice as slow: 16 x 6 dependent cross-lane steps per wave)
        const int row = t >> 2, seg = t & 3;
        double s0 = 0.0, s1 = 0.0;
        const double* mr = M + row * LS_LD + seg * 32;
#pragma unroll
        for (int k = 0; k < 32; k += 2) {          // (above the diagonal M holds no data)
            const int kk = seg * 32 + k;
            s0 = fma(kk <= row ? mr[k] : 0.0, sy[kk], s0);
            s1 = fma(kk + 1 <= row ? mr[k + 1] : 0.0, sy[kk + 1], s1);
        }
        double sacc = s0 + s1;
        sacc += __shfl_xor(sacc, 1);
        sacc += __shfl_xor(sacc, 2);
        if (seg == 0) sz[row] = sacc;
    }
    __syncthreads();
    {
        const int col = t & 127, seg = t >> 7;
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int i = 0; i < 32; i += 2) {
            const int ii = seg * 32 + i;
            s0 = fma(ii >= col ? M[ii * LS_LD + col] : 0.0, sz[ii], s0);
            s1 = fma(ii + 1 >= col ? M[(ii + 1) * LS_LD + col] : 0.0, sz[ii + 1], s1);
        }
        red2[seg][col] = s0 + s1;
    }
    __syncthreads();
    if (t < LS_NP) sal[t] = (red2[0][t] + red2[1][t]) + (red2[2][t] + red2[3][t]);
    if (w == 7) {          // a wave that has nothing to do in the alpha pass: fixed-order sums of 128 terms
        double sl = slog[lane] + slog[lane + 64];
        double sq = fma(sz[lane], sz[lane], sz[lane + 64] * sz[lane + 64]);
        for (int off = 32; off >= 1; off >>= 1) { sl += __shfl_xor(sl, off); sq += __shfl_xor(sq, off); }
        if (lane == 0) { red[0][DP + 1] = sl; red[1][DP + 1] = sq; }
    }
    __syncthreads();
    if (!a.want_grad) {
        if (t == 0) {
            const double sl = red[0][DP + 1], sq = red[1][DP + 1];
            ls_store_unit(a.host_res + 0, sl, a.seq);
            ls_store_unit(a.host_res + 1, sq, a.seq);
            ls_store_unit(a.host_res + LS_INFO_UNIT, 0.0, a.seq);
        }
        return;
    }

    // ---- H: traces 1/2 tr((alpha alpha^T - K^-1) dK/dtheta_k) over the lower tiles, off-diagonal tiles twice
    double gacc[DP + 1];
#pragma unroll
    for (int k = 0; k <= DP; k++) gacc[k] = 0.0;
    const int ntile = nb * (nb + 1) / 2;
    for (int q8 = 0; q8 * LS_NB < ntile; q8++) {
        const int tl = q8 * LS_NB + ((q8 & 1) ? LS_NB - 1 - w : w);      // snake over the waves: long tiles come first
        if (tl >= ntile) continue;
        int bi = 0;
        while ((bi + 1) * (bi + 2) / 2 <= tl) bi++;
        const int bj = tl - bi * (bi + 1) / 2;
        v4d kin = {0.0, 0.0, 0.0, 0.0};
        kin = c16::mfma_tn<LS_LD>(kin, M + (bi * 16) * LS_LD + bi * 16, M + (bi * 16) * LS_LD + bj * 16, (nb - bi) * 16, lane);
        const int j = 16 * bj + r;
        const double aj = sal[j];
        const double wt = bi == bj ? 1.0 : 2.0;
        double r2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < DP; k++) {
            const double xj = xs[k * LS_NP + j];
#pragma unroll
            for (int q = 0; q < 4; q++) { const double df = xs[k * LS_NP + 16 * bi + g + 4 * q] - xj; r2[q] = fma(df, df, r2[q]); }
        }
        double wh[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int i = 16 * bi + g + 4 * q;
            double wv = (sal[i] * aj - kin[q]) * wt;
            double kv;
            const double h = corr_and_h<KID>(r2[q], &kv);
            if (i == j) kv = 1.0;
            if (i >= N || j >= N) wv = 0.0;
            gacc[0] = fma(wv, kp.C * kv, gacc[0]);
            wh[q] = wv * kp.C * h;
        }
#pragma unroll
        for (int k = 0; k < DP; k++) {
            const double xj = xs[k * LS_NP + j];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const double df = xs[k * LS_NP + 16 * bi + g + 4 * q] - xj;
                gacc[1 + k] = fma(wh[q], df * df, gacc[1 + k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k <= DP; k++) {
        double v = gacc[k];
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[w][k] = v;
    }
    __syncthreads();
    // ---- I: results
    if (t <= kp.d) {
        double s = 0.0;
#pragma unroll
        for (int q = 0; q < LS_NB; q++) s += red[q][t];
        ls_store_unit(a.host_res + 2 + t, 0.5 * s, a.seq);
    }
    if (t == 0) {
        const double sl = red[0][DP + 1], sq = red[1][DP + 1];
        ls_store_unit(a.host_res + 0, sl, a.seq);
        ls_store_unit(a.host_res + 1, sq, a.seq);
        ls_store_unit(a.host_res + LS_INFO_UNIT, 0.0, a.seq);
    }
}

// N <= 128, d <= 16: the single-launch evaluation, for B thetas at once (B workgroups of one launch; B = 1: gpry_lml).
// params: B x 17 doubles [C, l_1 .. l_16] as the host computes them for a single evaluation (nullptr with B = 1: the
// context's theta); out: B x (2 + 33) doubles [sum log L_ii, quad, grad (1 + d) ...] per theta, info[B]: 0 or the failing
// column.  The results are picked up from the stamped units as they land (see LsUnit); the stream is only queried now and
// then, to notice a launch that failed.  Returns 1 if the model does not fit this kernel (the caller takes the general
// chain), 0 when `out` / `info` are filled, < 0 on errors.
#define LS_OUT_STRIDE (2 + 1 + GPRY_MAX_DIM)
int launch_lml_small_batch(gpry_ctx* ctx, int B, const double* params, int want_grad, double* out, int* info) {
    if (ctx->Np != LS_NP || ctx->d > 16) return 1;
    if (B < 1 || (B > 1 && !params)) return gpry_fail(ctx, -1, "lml_small: bad batch");
    KernParams kp;
    kp.C = exp(ctx->theta[0]); kp.d = ctx->d; kp.dpad = ctx->dpad; kp.has_aff = 0; kp.N = ctx->N;
    AffParams ap = make_ap(ctx, false);
    // the units live behind the first KiB of the staging buffer (the general chain writes plain doubles into that), the
    // parameter rows of a batch behind the units
    const int64_t unit_bytes = (int64_t)sizeof(LsUnit) * LS_UNITS * B;
    const int64_t par_off = round_up(1024 + unit_bytes, 256);
    GPRY_TRY(ensure_pinned(ctx, par_off + (int64_t)sizeof(double) * 17 * B + 256));
    LsUnit* hu = reinterpret_cast<LsUnit*>(static_cast<char*>(ctx->hpin) + 1024);
    LmlSmallArgs a;
    a.X = ctx->dX; a.y = ctx->dy; a.noise = ctx->dnoise;
    a.host_res = reinterpret_cast<LsUnit*>(static_cast<char*>(ctx->hpin_dev) + 1024);
    a.seq = ++ctx->lml_seq;
    a.want_grad = want_grad;
    a.batch = nullptr;
    if (params) {
        memcpy(static_cast<char*>(ctx->hpin) + par_off, params, sizeof(double) * 17 * B);
        a.batch = reinterpret_cast<const double*>(static_cast<char*>(ctx->hpin_dev) + par_off);
    }
#define LS2(DP, KID) hipLaunchKernelGGL((lml_small_kernel<DP, KID>), dim3((unsigned)B), dim3(LS_NT), 0, ctx->stream, a, kp, ap)
#define LS4(KID) { if (ctx->d <= 4) LS2(4, KID); else if (ctx->d <= 8) LS2(8, KID); else LS2(16, KID); }
    DISPATCH_KID(ctx->kernel_id, LS4)
#undef LS4
#undef LS2
    HIP_TRY(ctx, hipGetLastError());
    const unsigned long long seq = a.seq;
    const int need = 2 + (want_grad ? ctx->d + 1 : 0);
    long long spins = 0;
    bool finished = false;         // the stream has drained: whatever has not landed by now never will
    int done = 0;                  // thetas 0 .. done - 1 have been picked up
    for (;;) {
        while (done < B) {
            volatile LsUnit* vu = hu + (int64_t)LS_UNITS * done;
            if (vu[LS_INFO_UNIT].stamp != seq) break;
            unsigned long long b = vu[LS_INFO_UNIT].payload;
            double col; memcpy(&col, &b, 8);
            if (col != 0.0) { info[done] = (int)col; done++; continue; }
            bool all = true;
            for (int u = 0; u < need; u++) if (vu[u].stamp != seq) { all = false; break; }
            if (!all) break;
            __atomic_thread_fence(__ATOMIC_ACQUIRE);
            for (int u = 0; u < need; u++) { b = vu[u].payload; memcpy(&out[(int64_t)LS_OUT_STRIDE * done + u], &b, 8); }
            info[done] = 0;
            done++;
        }
        if (done == B) return 0;
        if (finished) return gpry_fail(ctx, -2, "lml: the single-launch evaluation did not deliver its results");
        if ((++spins & 0x3ff) == 0) {
            const hipError_t e = hipStreamQuery(ctx->stream);
            if (e == hipSuccess) finished = true;                       // one more look at the units, then give up
            else if (e != hipErrorNotReady) return gpry_fail(ctx, -2, "lml: %s", hipGetErrorString(e));
        }
    }
}
int launch_lml_small(gpry_ctx* ctx, int want_grad, double* out, int* info) {
    return launch_lml_small_batch(ctx, 1, nullptr, want_grad, out, info);
}
