// Covariance-matrix kernels for gfx950:
//   * train kernel build  K = C k(r) (+ diag noise)       gpry/gpr.py:1015-1016
//   * cross-kernel panel  K*^T (k-major) + mean partials   gpry/gpr.py:1179-1180
//   * LML gradient traces 1/2 tr((aa^T - K^-1) dK/dtheta)  sklearn:_gpr.py:625-649
// Kernel algebra restated from sklearn:kernels.py:1553-1580 (RBF), :1708-1768 (Matern),
// :1239-1291 (Constant), :931-966 (Product).  All kernels are templated on the kernel
// family so that the per-pair code has no branches.
#include "common.h"

#include "kern_math.h"

// ------------------------------------------------------------------------------------
// Batched launch (gpry_ctx::bn): theta blockIdx.z scales into ITS coordinate buffer with ITS length scales (bpar: the
// [C, l_1 .. l_d] rows the host computed, one per theta, bstride doubles apart like every per-theta buffer).
__global__ void scale_train_kernel(const double* __restrict__ X, double* __restrict__ Xs_,
                                   int64_t N, int64_t Np, int d, int dpad, AffParams ap, int* info_,
                                   const double* __restrict__ bpar, int64_t bstride) {
    const int tb = (int)blockIdx.z;
    double* __restrict__ Xs = bset(Xs_, tb, bstride);
    int* info = bset(info_, tb, bstride);
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // first kernel of every factorisation: it also clears the status / arrival words the panel chain starts from
    // (a memset in front of the first panel step is one more dependent dispatch)
    if (idx < 4) info[idx] = 0;
    if (idx >= Np * dpad) return;
    int64_t i = idx / dpad; int k = (int)(idx - i * dpad);
    double v = 0.0;
    if (i < N && k < d) v = X[i * d + k] / (bpar ? bset(bpar, tb, bstride)[1 + k] : ap.ls[k]);
    Xs[idx] = v;
}

int launch_scale_train(gpry_ctx* ctx) {
    AffParams ap = make_ap(ctx, false);
    int64_t n = ctx->Np * ctx->dpad;
    hipLaunchKernelGGL(scale_train_kernel, dim3((unsigned)((n + 255) / 256), 1, (unsigned)ctx->bn), dim3(256), 0,
                       ctx->stream, ctx->dX, ctx->dXs, ctx->N, ctx->Np, ctx->d, ctx->dpad, ap, ctx->dinfo, ctx->bpar, ctx->bstride);
    HIP_TRY(ctx, hipGetLastError());
    ctx->info_cleared = true;
    ctx->xs_foreign = false;
    return 0;
}

// An LML evaluation scales the training coordinates for ITS theta and used to put the prediction factor's back with one
// more launch at its end -- a dependent dispatch per evaluation of every fit, undone by the next evaluation.  It now only
// marks dXs; whoever reads it on behalf of the prediction factor (the launchers below, the resident predict kernel)
// restores it first.
int ensure_pred_xs(gpry_ctx* ctx) {
    if (!ctx->xs_foreign) return 0;
    GPRY_TRY(launch_scale_train(ctx));
    ctx->info_cleared = false;      // (this was not the start of a factorisation)
    return 0;
}

// decode linear index -> (bi >= bj) lower-triangular block pair
__device__ __forceinline__ void tri_decode(int64_t t, int* bi, int* bj) {
    int i = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
    while ((int64_t)(i + 1) * (i + 2) / 2 <= t) i++;
    while ((int64_t)i * (i + 1) / 2 > t) i--;
    *bi = i; *bj = (int)(t - (int64_t)i * (i + 1) / 2);
}

typedef double nt_v2d __attribute__((ext_vector_type(2)));
// streaming 16-byte store: K is written once and read next by another kernel
__device__ __forceinline__ void nt_store2(double2* p, double a, double b) {
    nt_v2d v = {a, b};
    __builtin_nontemporal_store(v, reinterpret_cast<nt_v2d*>(p));
}

// One 64 x 64 tile of K per workgroup (lower block triangle; off-diagonal tiles write both images), 256 threads = 4 waves.
// Algorithmic traffic: read X once (8 N d), write K once (8 N^2).
// Every wave owns a 32 x 32 quadrant; lane = (ly, lx) in an 8 x 8 grid; the thread's 4 x 4 patch has rows
// {16 a' + 2 ly + a''} and columns {16 b' + 2 lx + b''} of the quadrant.  With that shape BOTH images of an entry
// are stored straight from registers in whole 128-byte lines: the direct one as double2 over b'' (8 lx lanes = 16
// contiguous doubles of a row), the mirrored one as double2 over a'' (8 ly lanes = 16 contiguous doubles of the
// mirrored row).  No LDS transpose, no barrier behind the staging of the coordinates, and the stores of the first
// two patch rows leave while the last two are still being evaluated -- the round-2 kernel computed all 16 values,
// then stored, then transposed through LDS behind two barriers, so that VALU time (~20 us of FP64 issue at N = 4096)
// and store time (~29 us at the fill rate of the GPU) overlapped only between workgroups.
// The correlation keeps libm's exp / sqrt (corr_r2), NOT the straight-line versions of the cross-kernel panel: they
// differ from numpy's in the last bit of a few entries, and on the cond(K) = 5e15 matrix of BASELINE config 1 one ulp
// of K moves the posterior mean by 2e-5 of its range (tests/test_host_mirror_gpu.py::test_f9_config1_curved_degeneracy
// failed with them; the F1 goldens, 1e-13, would not have noticed).
template <int KID>
__global__ __launch_bounds__(256, 6) void kernel_train_q_kernel(
    const double* __restrict__ Xs_, const double* __restrict__ noise, double* __restrict__ K_,
    int64_t ld, KernParams kp, int add_noise, const double* __restrict__ bpar, int64_t bstride, double* __restrict__ U_, int u_full) {
    constexpr int TS = 64, WPR = TS / 32, NT = WPR * WPR * 64;
    extern __shared__ __attribute__((aligned(16))) double sm[];
    // batched launch (gpry_ctx::bn): theta blockIdx.z builds its own K from its own scaled coordinates and constant
    const double* __restrict__ Xs = bset(Xs_, (int)blockIdx.z, bstride);
    double* __restrict__ K = bset(K_, (int)blockIdx.z, bstride);
    if (bpar) kp.C = bset(bpar, (int)blockIdx.z, bstride)[0];
    // Coordinates of the 64 rows and the 64 columns, row-major as they lie in memory, row stride dp + 2 doubles: staged with
    // 16-byte loads and stores (whole 128-byte lines in, no bank conflicts), read back as double2 over k -- the rows of the
    // eight `ly` (columns of the eight `lx`) are 288 bytes apart = 8 banks, a conflict-free ds_read_b128.  (Round 3 staged
    // them transposed, [k][row], with 8-byte loads and a 16-way bank conflict on every LDS write: with all 2080 tiles of
    // N = 4096 resident at once, every workgroup sits in this prologue at the same time and nothing hides it -- 13 of the
    // 38 us of a launch, tools/r04/time_kernel_build.py with the arithmetic and the stores switched off.)
    const int dp = kp.dpad, ldx = dp + 2, dp2 = dp >> 1;
    double* Xi = sm;              // [TS][ldx]
    double* Xj = sm + TS * ldx;   // [TS][ldx]
    int bi, bj; tri_decode(blockIdx.x, &bi, &bj);
    const int t = threadIdx.x, w = t >> 6, lane = t & 63, lx = lane & 7, ly = lane >> 3;
    if (U_ != nullptr) {
        // (potrf_stacked, chol_panel.hip: the matrix that is appended to K starts as the identity -- tile (bj, bi) of it goes out
        // with tile (bi, bj) of K, first, so that the stores are under way while the coordinates are staged; further left of
        // its diagonal than one tile nothing is ever read)
        double* __restrict__ U = bset(U_, (int)blockIdx.z, bstride);
        for (int e = t; e < TS * (TS / 2); e += NT) {
            const int i = e >> 5, j2 = (e & 31) * 2;
            const double one = bi == bj ? 1.0 : 0.0;
            nt_store2(reinterpret_cast<double2*>(U + ((int64_t)bj * TS + i) * ld + (int64_t)bi * TS + j2), j2 == i ? one : 0.0, j2 + 1 == i ? one : 0.0);
            // (the tile left of a diagonal tile is read as well: row block a enters the chain at step a with the left-looking
            // update of the strip before it, and a panel of two strips may start one strip left of a; the comparator without
            // the zero structure reads everything)
            if ((u_full && bi != bj) || bi == bj + 1) nt_store2(reinterpret_cast<double2*>(U + ((int64_t)bi * TS + i) * ld + (int64_t)bj * TS + j2), 0.0, 0.0);
        }
    }
    for (int e = t; e < TS * dp2; e += NT) {
        const int row = e / dp2, k2 = (e - row * dp2) * 2;
        *reinterpret_cast<double2*>(Xi + row * ldx + k2) = *reinterpret_cast<const double2*>(Xs + ((int64_t)bi * TS + row) * dp + k2);
        *reinterpret_cast<double2*>(Xj + row * ldx + k2) = *reinterpret_cast<const double2*>(Xs + ((int64_t)bj * TS + row) * dp + k2);
    }
    __syncthreads();
    const int r0 = 32 * (w / WPR) + 2 * ly, c0 = 32 * (w % WPR) + 2 * lx;
    double r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) r2[a][b] = 0.0;
    const double* pi = Xi + r0 * ldx;
    const double* pj = Xj + c0 * ldx;
    for (int k = 0; k < dp; k += 2) {       // k ascending, one multiply-add per coordinate: the sums of the transposed staging, bit for bit
        double2 xi[4], xj[4];
#pragma unroll
        for (int a = 0; a < 4; a++) {
            xi[a] = *reinterpret_cast<const double2*>(pi + ((a >> 1) * 16 + (a & 1)) * ldx + k);
            xj[a] = *reinterpret_cast<const double2*>(pj + ((a >> 1) * 16 + (a & 1)) * ldx + k);
        }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) { const double df = xi[a].x - xj[b].x; r2[a][b] = fma(df, df, r2[a][b]); }
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) { const double df = xi[a].y - xj[b].y; r2[a][b] = fma(df, df, r2[a][b]); }
    }
    // only the tiles on the diagonal and on the padded edge have entries that are not C k(r)
    const bool special = bi == bj || ((int64_t)bi + 1) * TS > kp.N;
#pragma unroll
    for (int ap = 0; ap < 2; ap++) {
        double v[2][4];
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                double x = kp.C * corr_r2<KID>(r2[2 * ap + a2][b]);     // libm exp / sqrt: see above
                if (special) {
                    const int64_t i = (int64_t)bi * TS + r0 + 16 * ap + a2;
                    const int64_t j = (int64_t)bj * TS + c0 + 16 * (b >> 1) + (b & 1);
                    if (i == j) x = kp.C + (add_noise ? noise[i < kp.N ? i : 0] : 0.0);
                    if (i >= kp.N || j >= kp.N) x = (i == j) ? 1.0 : 0.0;   // identity padding
                }
                v[a2][b] = x;
            }
#pragma unroll
        for (int a2 = 0; a2 < 2; a2++) {
            const int64_t i = (int64_t)bi * TS + r0 + 16 * ap + a2;
            double2* p = reinterpret_cast<double2*>(K + i * ld + (int64_t)bj * TS + c0);
            nt_store2(p, v[a2][0], v[a2][1]);
            nt_store2(p + 8, v[a2][2], v[a2][3]);
        }
        if (bi != bj) {
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int64_t j = (int64_t)bj * TS + c0 + 16 * (b >> 1) + (b & 1);
                double2* q = reinterpret_cast<double2*>(K + j * ld + (int64_t)bi * TS + r0 + 16 * ap);
                nt_store2(q, v[0][b], v[1][b]);
            }
        }
    }
}

int launch_kernel_train(gpry_ctx* ctx, double* K, int add_noise, double* U) {
    KernParams kp = make_kp(ctx);
    const int64_t nbq = ctx->Np / 64, ntq = nbq * (nbq + 1) / 2;
    const size_t smq = sizeof(double) * (size_t)(2 * 64 * (ctx->dpad + 2));
    const dim3 gq((unsigned)ntq, 1, (unsigned)ctx->bn);
#define KQ(KID) hipLaunchKernelGGL((kernel_train_q_kernel<KID>), gq, dim3(256), smq, ctx->stream, ctx->dXs, ctx->dnoise, K, ctx->Np, kp, \
                                   add_noise, ctx->bpar, ctx->bstride, U, ctx->opt_chol_stacked_dense)
    DISPATCH_KID(ctx->kernel_id, KQ)
#undef KQ
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Rows [row0, row0 + k) of the training covariance matrix K + diag(noise) of the CURRENT training set
// (kp.N rows, scaled coordinates Xs), with the arithmetic of kernel_train_kernel (same corr_r2, diagonal
// forced to C + noise): the border of a factor that grows by k points (gpry_append_rows).
//   Bk[j * ldk + a] = K[row0 + a][j]   for j < row0 (0 for j >= row0 and a >= k):  the k-major image the
//                                       product U = V B wants, like the cross-kernel panel
//   Cb[a * 64 + b]  = K[row0 + a][row0 + b]                                        (k <= 64)
template <int KID>
__global__ __launch_bounds__(256) void kernel_rows_kernel(const double* __restrict__ Xs, const double* __restrict__ noise,
                                                          int64_t row0, int k, int64_t Np, int64_t ldk,
                                                          double* __restrict__ Bk, double* __restrict__ Cb, KernParams kp) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;     // (j, a): a fastest
    if (idx >= Np * ldk) return;
    const int64_t j = idx / ldk;
    const int a = (int)(idx - j * ldk);
    double v = 0.0;
    if (a < k && j < row0 + k) {
        const double* xa = Xs + (row0 + a) * kp.dpad;
        const double* xj = Xs + j * kp.dpad;
        double r2 = 0.0;
        for (int c = 0; c < kp.dpad; c++) { const double df = xa[c] - xj[c]; r2 = fma(df, df, r2); }
        v = kp.C * corr_r2<KID>(r2);            // the arithmetic of the training build (kernel_train_q_kernel)
        if (j == row0 + a) v = kp.C + noise[j];
        if (j >= row0) { Cb[a * 64 + (int)(j - row0)] = v; v = 0.0; }
    }
    Bk[idx] = v;
}
int launch_kernel_rows(gpry_ctx* ctx, int64_t row0, int k, int64_t ldk, double* Bk, double* Cb) {
    GPRY_TRY(ensure_pred_xs(ctx));
    KernParams kp = make_kp(ctx);
    const int64_t n = ctx->Np * ldk;
#define KR(KID) hipLaunchKernelGGL((kernel_rows_kernel<KID>), dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, \
                                   ctx->dXs, ctx->dnoise, row0, k, ctx->Np, ldk, Bk, Cb, kp)
    DISPATCH_KID(ctx->kernel_id, KR)
#undef KR
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Cross-kernel panel.  Thread <-> candidate (coalesced stores along m); a workgroup
// covers 256 candidates x 128 training rows; scaled candidate coordinates stay in
// registers, the training chunk sits in LDS and is read wave-uniformly (broadcast).
// Output Kst[j*ldk + m] = C k(x*_m, x_j) (rows j >= N are zero) and the per-chunk mean
// partial  mean_part[jc*mc + m] = sum_{j in chunk} alpha_[j] * Kst[j][m].
// The candidates of a chunk, mapped to the unit cube and divided by the length scales ONCE, coordinate-major:
// Xcs[k * ldm + ml] = ((x_k - lo_k) / span_k) / l_k (true divisions, in this order, as sklearn and the preprocessor do;
// zeros beyond the pool and beyond d).  Every one of the Np / 128 row-chunk workgroups of cross_build_kernel used to redo
// these 2 d FP64 divisions per candidate from row-major coordinates (one 128-byte line per lane and load).
__global__ __launch_bounds__(256) void scale_cand_kernel(const double* __restrict__ Xc, int64_t M, int64_t m0, int64_t mc, int d, int dsel,
                                                         int has_aff, AffParams ap, double* __restrict__ Xcs, int64_t ldm) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;      // (ml, k): k fastest -> the loads are whole lines
    if (idx >= mc * dsel) return;
    const int64_t ml = idx / dsel;
    const int k = (int)(idx - ml * dsel);
    const int64_t m = m0 + ml;
    double v = 0.0;
    if (k < d && m < M) {
        v = Xc[m * d + k];
        if (has_aff) v = (v - ap.lo[k]) / ap.span[k];
        v = v / ap.ls[k];
    }
    Xcs[(int64_t)k * ldm + ml] = v;
}

template <int DP, int KID>
__global__ __launch_bounds__(256) void cross_build_kernel(
    const double* __restrict__ Xcs, int64_t ldm, int64_t mc,
    const double* __restrict__ Xs, const double* __restrict__ alpha_,
    double* __restrict__ Kst, int64_t ldk, double* __restrict__ mean_part,
    KernParams kp) {
    __shared__ double Xl[128 * DP];
    __shared__ double al[128];
    const int t = threadIdx.x;
    const int64_t ml = (int64_t)blockIdx.x * 256 + t;   // local candidate index in the chunk
    const int jc = blockIdx.y;
    for (int e = t; e < 128 * DP; e += 256) {
        int row = e / DP, k = e - row * DP;
        Xl[e] = (k < kp.dpad) ? Xs[((int64_t)jc * 128 + row) * kp.dpad + k] : 0.0;
    }
    if (t < 128) al[t] = alpha_ ? alpha_[jc * 128 + t] : 0.0;
    double xs[DP];
#pragma unroll
    for (int k = 0; k < DP; k++) xs[k] = Xcs[(int64_t)k * ldm + ml];       // (the buffer covers the padded chunk)
    __syncthreads();
    double macc = 0.0;
    const bool in_chunk = ml < mc;
    // rows of this 128-chunk that are real training points (the rest is padding: zeros)
    const int64_t left = kp.N - (int64_t)jc * 128;
    const int nvalid = left >= 128 ? 128 : (left > 0 ? (int)left : 0);
    const int nfull = nvalid & ~3;
    for (int j0 = 0; j0 < nfull; j0 += 4) {      // branch-free: four rows in flight per lane
        double r2[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int k = 0; k < DP; k++) {
#pragma unroll
            for (int q = 0; q < 4; q++) {
                double df = xs[k] - Xl[(j0 + q) * DP + k];
                r2[q] = fma(df, df, r2[q]);
            }
        }
        double v[4];
#pragma unroll
        for (int q = 0; q < 4; q++) {
            v[q] = kp.C * corr_r2_fast<KID>(r2[q]);
            macc = fma(al[j0 + q], v[q], macc);
        }
        if (in_chunk) {                           // one predicated region for the four stores
#pragma unroll
            for (int q = 0; q < 4; q++) Kst[((int64_t)jc * 128 + j0 + q) * ldk + ml] = v[q];
        }
    }
    for (int jj = nfull; jj < 128; jj++) {       // ragged tail and padding rows
        double v = 0.0;
        if (jj < nvalid) {
            double r2 = 0.0;
#pragma unroll
            for (int k = 0; k < DP; k++) {
                double df = xs[k] - Xl[jj * DP + k];
                r2 = fma(df, df, r2);
            }
            v = kp.C * corr_r2_fast<KID>(r2);
            macc = fma(al[jj], v, macc);
        }
        if (in_chunk) Kst[((int64_t)jc * 128 + jj) * ldk + ml] = v;
    }
    if (in_chunk && mean_part) mean_part[(int64_t)jc * mc + ml] = macc;
}

// ------------------------------------------------------------------------------------
// The same panel with the distances on the matrix pipe (the NORA sweep and large predict batches; gpry/gpr.py:1179).
// cross_build_kernel above is FP64-VALU-issue-bound: ~75 wave-instructions per (candidate, training row), 2 d of them the
// differences and squares of the distance.  Here r^2 = |x|^2 + |y|^2 - 2 x.y with the dot products as
// v_mfma_f64_16x16x4_f64 (d / 4 instructions per 16 x 16 block of pairs, issued beside the vector work of the other
// waves): 3 vector instructions per pair instead of 2 d.  The expanded form cancels, so both sides are CENTRED first
// (x - c, y - c with c = the mean of the training rows per dimension: distances do not change, the norms shrink to the
// spread of the data): the absolute error of r^2 is <= 4 eps (|x - c|^2 + |y - c|^2), i.e. up to 4 eps C (1 + 6 R^2) in an
// entry of K* (R = the scaled radius of the training set: 1e-14 C at l = 0.3, d = 16, but 1e-10 C at l = 0.01) -- and
// the posterior mean multiplies that by the weights alpha_.  The caller (api.hip, where the panel form is chosen) therefore takes this form
// only while its estimate of that product stays a factor of four inside the 1e-6 the posterior mean is specified to, and the
// difference form otherwise.  The exact difference form also stays for gpry_kernel_cross (K* itself is
// compared at 1e-13), the small batches and the Kriging-believer registrations.
// Workgroup = 128 training rows x 256 candidates as above; wave w owns candidates 64 w .. 64 w + 63 as four MFMA column
// blocks in two pairs; per 16 training rows (A operand from LDS) 4 x d/4 MFMAs give each lane 4 x 4 pairs: rows g + 4 q
// (g = lane >> 4), two neighbouring candidates per block pair, stored as 16-byte pieces of 256-byte row segments.
// Ycs: Np x DP centred scaled training rows (zero rows for the padding), row-major; Xcs: DP x ldm centred scaled candidates.
// HYB (round 6): the HYBRID form for models whose error estimates rule the expanded form out (api.hip: run_sweep).  The error
// of the expanded r^2 is absolute -- 4 eps (2 r^2 + 4 R^2) whatever r -- while the kernel only listens to r^2 where r is small:
// beyond r^2 = 100 every smooth kernel here has |dk / d r^2| < 4e-9 C, so that even R^2 = 1.6e7 (all length scales at their
// lower bound) moves such an entry by < 1e-15 C.  A pair that comes out NEARER than that has its distance taken again from the
// coordinates themselves, sum_k ((x_k - c_k) / l_k - (y_k - c_k) / l_k)^2 -- the difference form's own arithmetic, on the
// centred coordinates the kernel already holds (the training row from LDS, the candidate's 16-byte pieces from L2).  Models
// that fail the gate because of SHORT length scales (large R^2: the bench's fitted model) have next to no near pairs: the
// branch is taken by the waves that hold a candidate on or next to a training point, and the panel costs what the expanded
// form costs.  (A model that fails it with ordinary length scales -- huge weights -- has nothing but near pairs; the caller
// then takes cross_build_kernel.)
template <int DP, int KID, bool HYB>
__global__ __launch_bounds__(256) void cross_build_mfma_kernel(
    const double* __restrict__ Xcs, int64_t ldm, int64_t mc,
    const double* __restrict__ Ycs, const double* __restrict__ alpha_,
    double* __restrict__ Kst, int64_t ldk, double* __restrict__ mean_part, KernParams kp, double ucut) {
    constexpr int S = DP + 2, KS = DP / 4;
    constexpr double SC = corr_scale<KID>();
    __shared__ __attribute__((aligned(16))) double Yl[128 * S];
    __shared__ double yn[128], al[128];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, r = lane & 15, g = lane >> 4;
    const int jc = blockIdx.y;
    const int64_t mb = (int64_t)blockIdx.x * 256 + 64 * w;      // first candidate of this wave (local index in the chunk)
    for (int e = t; e < 128 * (DP / 2); e += 256) {
        const int row = e / (DP / 2), k2 = (e - row * (DP / 2)) * 2;
        *reinterpret_cast<double2*>(Yl + row * S + k2) = *reinterpret_cast<const double2*>(Ycs + ((int64_t)jc * 128 + row) * DP + k2);
    }
    if (t < 128) al[t] = alpha_ ? alpha_[jc * 128 + t] : 0.0;
    // B operands: coordinate 4 kk + g of the candidate of lane r in block tl; |x - c|^2 of that candidate.  The blocks come in
    // pairs over 32 candidates: lane r holds candidate 2 r in the even block and 2 r + 1 in the odd one (one 16-byte load
    // gives both), so that a lane ends up with the values of two NEIGHBOURING candidates for every training row and stores
    // them as one 16-byte piece -- 256 contiguous bytes per row and instruction.  (Round 4 had candidate r / 16 + r in the
    // blocks and 8-byte stores: the kernel ran at 3.9 TB/s of panel writes whatever its instruction count, the rate of 8-byte
    // stores on this part -- MI355X_MICROARCH.md: 0.54-0.70 of the 16-byte rate.)
    double xb[4][KS], xn[4];
#pragma unroll
    for (int tp = 0; tp < 2; tp++) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int kk = 0; kk < KS; kk++) {
            const double2 v = *reinterpret_cast<const double2*>(Xcs + (int64_t)(4 * kk + g) * ldm + mb + 32 * tp + 2 * r);
            xb[2 * tp][kk] = v.x; xb[2 * tp + 1][kk] = v.y;
            s0 = fma(v.x, v.x, s0); s1 = fma(v.y, v.y, s1);
        }
        s0 += __shfl_xor(s0, 16); s0 += __shfl_xor(s0, 32);
        s1 += __shfl_xor(s1, 16); s1 += __shfl_xor(s1, 32);
        xn[2 * tp] = s0 * SC; xn[2 * tp + 1] = s1 * SC;     // (the scale of corr_scaled_fast folded into both norms and the product: u = s r^2)
    }
    __syncthreads();
    if (t < 128) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < DP; k++) s = fma(Yl[t * S + k], Yl[t * S + k], s);
        yn[t] = s * SC;
    }
    __syncthreads();
    const int64_t left = kp.N - (int64_t)jc * 128;
    const int nvalid = left >= 128 ? 128 : (left > 0 ? (int)left : 0);      // training rows of this chunk (the rest: zero padding)
    double macc[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 1
    for (int j0 = 0; j0 < 128; j0 += 16) {
        double a[KS], ynq[4], alq[4];
#pragma unroll
        for (int kk = 0; kk < KS; kk++) a[kk] = Yl[(j0 + r) * S + 4 * kk + g];
#pragma unroll
        for (int q = 0; q < 4; q++) { ynq[q] = yn[j0 + g + 4 * q]; alq[q] = al[j0 + g + 4 * q]; }
#pragma unroll
        for (int tp = 0; tp < 2; tp++) {
            v4d acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < KS; kk++) acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], xb[2 * tp][kk], acc0, 0, 0, 0);
#pragma unroll
            for (int kk = 0; kk < KS; kk++) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[kk], xb[2 * tp + 1][kk], acc1, 0, 0, 0);
            const int64_t mloc = mb + 32 * tp + 2 * r;      // this lane's pair of candidates (mc is a multiple of 128: both or neither)
            const bool in_chunk = mloc < mc;
#pragma unroll
            for (int q = 0; q < 4; q++) {
                const int jj = j0 + g + 4 * q;
                double u0 = fma(-2.0 * SC, acc0[q], xn[2 * tp] + ynq[q]);
                double u1 = fma(-2.0 * SC, acc1[q], xn[2 * tp + 1] + ynq[q]);
                if (HYB && (u0 < ucut || u1 < ucut)) {      // a near pair: its distance again, from the coordinates
                    double r0 = 0.0, r1 = 0.0;
#pragma unroll 2
                    for (int k = 0; k < DP; k++) {      // (rare: kept short in registers -- fully unrolled it cost the kernel a third of its occupancy)
                        const double yk = Yl[jj * S + k];
                        const double2 xk = *reinterpret_cast<const double2*>(Xcs + (int64_t)k * ldm + mloc);
                        const double d0 = xk.x - yk, d1 = xk.y - yk;
                        r0 = fma(d0, d0, r0); r1 = fma(d1, d1, r1);
                    }
                    u0 = r0 * SC; u1 = r1 * SC;
                }
                u0 = fmax(u0, 1e-290);         // (rounding may leave a small negative number; at 1e-290 the correlation is 1 exactly)
                u1 = fmax(u1, 1e-290);
                double v0 = kp.C * corr_scaled_fast<KID>(u0);
                double v1 = kp.C * corr_scaled_fast<KID>(u1);
                if (jj >= nvalid) { v0 = 0.0; v1 = 0.0; }
                macc[2 * tp] = fma(alq[q], v0, macc[2 * tp]);
                macc[2 * tp + 1] = fma(alq[q], v1, macc[2 * tp + 1]);
                if (in_chunk) *reinterpret_cast<double2*>(Kst + ((int64_t)jc * 128 + jj) * ldk + mloc) = make_double2(v0, v1);
            }
        }
    }
    if (mean_part) {
#pragma unroll
        for (int tl = 0; tl < 4; tl++) {
            double s = macc[tl];
            s += __shfl_xor(s, 16);
            s += __shfl_xor(s, 32);
            const int64_t ml = mb + 32 * (tl >> 1) + 2 * r + (tl & 1);
            if (g == 0 && ml < mc) mean_part[(int64_t)jc * mc + ml] = s;
        }
    }
}
// centred scaled training rows for cross_build_mfma_kernel: Ycs[i][k] = (X[i][k] - c_k) / l_k, zero rows / columns as padding
__global__ __launch_bounds__(256) void center_train_kernel(const double* __restrict__ X, double* __restrict__ Ycs, int64_t N, int64_t Np, int d,
                                                           int dsel, AffParams ap, AffParams cen) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Np * dsel) return;
    const int64_t i = idx / dsel;
    const int k = (int)(idx - i * dsel);
    Ycs[idx] = (i < N && k < d) ? (X[i * d + k] - cen.lo[k]) / ap.ls[k] : 0.0;
}
// ... and the candidates: Xcs[k * ldm + ml] = ((x_k - lo_k) / span_k - c_k) / l_k
__global__ __launch_bounds__(256) void center_cand_kernel(const double* __restrict__ Xc, int64_t M, int64_t m0, int64_t mc, int d, int dsel,
                                                          int has_aff, AffParams ap, AffParams cen, double* __restrict__ Xcs, int64_t ldm) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= mc * dsel) return;
    const int64_t ml = idx / dsel;
    const int k = (int)(idx - ml * dsel);
    const int64_t m = m0 + ml;
    double v = 0.0;
    if (k < d && m < M) {
        v = Xc[m * d + k];
        if (has_aff) v = (v - ap.lo[k]) / ap.span[k];
        v = (v - cen.lo[k]) / ap.ls[k];
    }
    Xcs[(int64_t)k * ldm + ml] = v;
}

int launch_cross_prepare(gpry_ctx* ctx) {
    const int dsel = ctx->d <= 4 ? 4 : ctx->d <= 8 ? 8 : ctx->d <= 16 ? 16 : ctx->d <= 24 ? 24 : 32;
    if (ctx->Np * dsel > ctx->ycs_cap) {
        if (ctx->dYcs) GPRY_TRY(dev_free(ctx, ctx->dYcs));
        ctx->dYcs = nullptr; ctx->ycs_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dYcs, ctx->Np * dsel));
        ctx->ycs_cap = ctx->Np * dsel;
    }
    AffParams ap = make_ap(ctx, false), cen;
    for (int k = 0; k < GPRY_MAX_DIM; k++) { cen.ls[k] = 1.0; cen.span[k] = 1.0; cen.lo[k] = k < ctx->d ? ctx->xcenter[k] : 0.0; }
    hipLaunchKernelGGL(center_train_kernel, dim3((unsigned)((ctx->Np * dsel + 255) / 256)), dim3(256), 0, ctx->stream, ctx->dX, ctx->dYcs,
                       ctx->N, ctx->Np, ctx->d, dsel, ap, cen);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// after launch_cross_prepare (same theta, same training set)
int launch_cross_build_mfma(gpry_ctx* ctx, const double* Xc, int64_t m0, int64_t mc, int64_t ldk, double* Kst, double* mean_part,
                            int raw_affine, int hybrid) {
    hipStream_t st = ctx->stream;
    KernParams kp = make_kp(ctx);
    kp.has_aff = raw_affine && ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff), cen;
    for (int k = 0; k < GPRY_MAX_DIM; k++) { cen.ls[k] = 1.0; cen.span[k] = 1.0; cen.lo[k] = k < ctx->d ? ctx->xcenter[k] : 0.0; }
    if (ctx->d > 32) return gpry_fail(ctx, -1, "d > 32 is not supported");
    const int dsel = ctx->d <= 4 ? 4 : ctx->d <= 8 ? 8 : ctx->d <= 16 ? 16 : ctx->d <= 24 ? 24 : 32;
    const int64_t ldm = round_up(mc, 256);
    if (dsel * ldm > ctx->xcs_cap) {
        if (ctx->dXcs) GPRY_TRY(dev_free(ctx, ctx->dXcs));
        ctx->dXcs = nullptr; ctx->xcs_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dXcs, dsel * ldm));
        ctx->xcs_cap = dsel * ldm;
    }
    hipLaunchKernelGGL(center_cand_kernel, dim3((unsigned)((ldm * dsel + 255) / 256)), dim3(256), 0, st, Xc, ctx->sw_M, m0, ldm, ctx->d, dsel,
                       kp.has_aff, ap, cen, ctx->dXcs, ldm);
    dim3 grid((unsigned)((mc + 255) / 256), (unsigned)(ctx->Np / 128));
    // near pairs of the hybrid form: r^2 < 100 (in units of the kernel's scaled argument u = corr_scale * r^2)
#define CM2(DP, KID) do { const double ucut = 100.0 * corr_scale<KID>();                                                                  \
                       if (hybrid) { hipLaunchKernelGGL((cross_build_mfma_kernel<DP, KID, true>), grid, dim3(256), 0, st, ctx->dXcs, ldm, mc, ctx->dYcs, \
                                                        ctx->dalpha_, Kst, ldk, mean_part, kp, ucut); }                                  \
                       else { hipLaunchKernelGGL((cross_build_mfma_kernel<DP, KID, false>), grid, dim3(256), 0, st, ctx->dXcs, ldm, mc, ctx->dYcs, \
                                                 ctx->dalpha_, Kst, ldk, mean_part, kp, ucut); } } while (0)
#define CM4(KID) { if (dsel == 4) CM2(4, KID); else if (dsel == 8) CM2(8, KID); else if (dsel == 16) CM2(16, KID); \
                   else if (dsel == 24) CM2(24, KID); else CM2(32, KID); }
    DISPATCH_KID(ctx->kernel_id, CM4)
#undef CM4
#undef CM2
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// (Round 3, measured and removed: reading the training rows of a chunk with scalar loads -- constant address space,
// s_load_dwordx16, SGPR operands of the v_add_f64 -- instead of broadcasting them from LDS: same bits, 11.2 vs 11.2 ms per
// 1e6 candidates at N = 4096, d = 16 with Matern-5/2, 10.4 vs 9.0 ms with RBF, 2.5 vs 1.9 ms at N = 1024: the LDS pipe
// is not what holds this kernel at 73 % VALU issue, and the scalar loads are waited for in full at the top of every
// two-row step.  tools/ab_cross_build.py, profiles/r03_ab_cross_build.log.)
// The same panel for a SMALL batch (a few hundred points: gpry_predict / gpry_predict_grad_batch): with one
// candidate per thread and 128 training rows per workgroup the launch has Np/128 workgroups whose four waves
// walk 128 rows each at a lone wave's FP64 rate (36 us whatever the batch size).  Here a workgroup is 64
// candidates x 4 row groups of 32 training rows: four times as many workgroups, loops four times shorter.
// Mean partials: one per (128-row chunk, row group): mean_part[(jc * 4 + rg) * mc + ml].
template <int DP, int KID>
__global__ __launch_bounds__(256) void cross_build_small_kernel(
    const double* __restrict__ Xc, int64_t M, int64_t m0, int64_t mc,
    const double* __restrict__ Xs, const double* __restrict__ alpha_,
    double* __restrict__ Kst, int64_t ldk, double* __restrict__ mean_part,
    KernParams kp, AffParams ap) {
    __shared__ double Xl[128 * DP];
    __shared__ double al[128];
    const int t = threadIdx.x, rg = t >> 6;
    const int64_t ml = (int64_t)blockIdx.x * 64 + (t & 63);
    const int64_t m = m0 + ml;
    const int jc = blockIdx.y;
    for (int e = t; e < 128 * DP; e += 256) {
        int row = e / DP, k = e - row * DP;
        Xl[e] = (k < kp.dpad) ? Xs[((int64_t)jc * 128 + row) * kp.dpad + k] : 0.0;
    }
    if (t < 128) al[t] = alpha_ ? alpha_[jc * 128 + t] : 0.0;
    double xs[DP];
#pragma unroll
    for (int k = 0; k < DP; k++) {
        double v = 0.0;
        if (k < kp.d && m < M) {
            v = Xc[m * kp.d + k];
            if (kp.has_aff) v = (v - ap.lo[k]) / ap.span[k];
            v = v / ap.ls[k];
        }
        xs[k] = v;
    }
    __syncthreads();
    double macc = 0.0;
    const bool in_chunk = ml < mc;
    const int64_t left = kp.N - (int64_t)jc * 128;
    const int nvalid = left >= 128 ? 128 : (left > 0 ? (int)left : 0);
    for (int jj = rg * 32; jj < rg * 32 + 32; jj++) {
        double v = 0.0;
        if (jj < nvalid) {
            double r2 = 0.0;
#pragma unroll
            for (int k = 0; k < DP; k++) {
                double df = xs[k] - Xl[jj * DP + k];
                r2 = fma(df, df, r2);
            }
            v = kp.C * corr_r2_fast<KID>(r2);
            macc = fma(al[jj], v, macc);
        }
        if (in_chunk) Kst[((int64_t)jc * 128 + jj) * ldk + ml] = v;
    }
    if (in_chunk && mean_part) mean_part[((int64_t)jc * 4 + rg) * mc + ml] = macc;
}
// mean_part then holds 4 * (Np / 128) partials per candidate (row stride mc)
int launch_cross_build_small(gpry_ctx* ctx, const double* Xc, int64_t m0, int64_t mc, int64_t ldk,
                             double* Kst, double* mean_part, int raw_affine) {
    GPRY_TRY(ensure_pred_xs(ctx));
    hipStream_t st = ctx->stream;
    KernParams kp = make_kp(ctx);
    kp.has_aff = raw_affine && ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff);
    dim3 grid((unsigned)((mc + 63) / 64), (unsigned)(ctx->Np / 128));
    int64_t M = ctx->sw_M;
    if (ctx->d > 32) return gpry_fail(ctx, -1, "d > 32 is not supported");
#define CS2(DP, KID) hipLaunchKernelGGL((cross_build_small_kernel<DP, KID>), grid, dim3(256), 0, st, Xc, M, \
                                        m0, mc, ctx->dXs, ctx->dalpha_, Kst, ldk, mean_part, kp, ap)
#define CS4(KID) { if (ctx->d <= 4) CS2(4, KID); else if (ctx->d <= 8) CS2(8, KID); \
                   else if (ctx->d <= 16) CS2(16, KID); else if (ctx->d <= 24) CS2(24, KID); else CS2(32, KID); }
    DISPATCH_KID(ctx->kernel_id, CS4)
#undef CS4
#undef CS2
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

int launch_cross_build(gpry_ctx* ctx, const double* Xc, int64_t m0, int64_t mc, int64_t ldk,
                       double* Kst, double* mean_part, int raw_affine, hipStream_t st) {
    GPRY_TRY(ensure_pred_xs(ctx));
    if (!st) st = ctx->stream;
    KernParams kp = make_kp(ctx);
    kp.has_aff = raw_affine && ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff);
    dim3 grid((unsigned)((mc + 255) / 256), (unsigned)(ctx->Np / 128));
    int64_t M = ctx->sw_M;
    if (ctx->d > 32) return gpry_fail(ctx, -1, "d > 32 is not supported");
    const int dsel = ctx->d <= 4 ? 4 : ctx->d <= 8 ? 8 : ctx->d <= 16 ? 16 : ctx->d <= 24 ? 24 : 32;
    const int64_t ldm = round_up(mc, 256);
    if (dsel * ldm > ctx->xcs_cap) {
        if (ctx->dXcs) GPRY_TRY(dev_free(ctx, ctx->dXcs));
        ctx->dXcs = nullptr; ctx->xcs_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dXcs, dsel * ldm));
        ctx->xcs_cap = dsel * ldm;
    }
    hipLaunchKernelGGL(scale_cand_kernel, dim3((unsigned)((ldm * dsel + 255) / 256)), dim3(256), 0, st, Xc, M, m0, ldm, ctx->d, dsel,
                       kp.has_aff, ap, ctx->dXcs, ldm);
#define CB2(DP, KID) hipLaunchKernelGGL((cross_build_kernel<DP, KID>), grid, dim3(256), 0, st, ctx->dXcs, ldm, \
                                        mc, ctx->dXs, ctx->dalpha_, Kst, ldk, mean_part, kp)
#define CB4(KID) { if (ctx->d <= 4) CB2(4, KID); else if (ctx->d <= 8) CB2(8, KID); \
                   else if (ctx->d <= 16) CB2(16, KID); else if (ctx->d <= 24) CB2(24, KID); else CB2(32, KID); }
    DISPATCH_KID(ctx->kernel_id, CB4)
#undef CB4
#undef CB2
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// LML gradient traces.  One 64x64 tile of pairs per workgroup (lower block triangle,
// off-diagonal tiles weighted twice).  W_ij = a_i a_j - Kinv_ij is read once (Kinv holds
// the lower triangle); distances are recomputed from the scaled coordinates, so the
// (N,N,d) tensor of the reference never exists.  Per-tile partial sums go to
// part[tile][DP+1]; reduce_traces_kernel sums them in a fixed order (deterministic).
template <int DP, int KID>
__global__ __launch_bounds__(256) void lml_traces_kernel(
    const double* __restrict__ Xs_, const double* __restrict__ Kinv_, int64_t ld,
    const double* __restrict__ alpha_, double* __restrict__ part_, KernParams kp,
    const double* __restrict__ bpar, int64_t bstride) {
    // batched launch (gpry_ctx::bn): the buffers and the constant of theta blockIdx.z
    const double* __restrict__ Xs = bset(Xs_, (int)blockIdx.z, bstride);
    const double* __restrict__ Kinv = bset(Kinv_, (int)blockIdx.z, bstride);
    const double* __restrict__ alpha = bset(alpha_, (int)blockIdx.z, bstride);
    double* __restrict__ part = bset(part_, (int)blockIdx.z, bstride);
    if (bpar) kp.C = bset(bpar, (int)blockIdx.z, bstride)[0];
    // coordinates of the 64 rows and the 64 columns, [k][row] with a row stride of 66 doubles: with 64 the sixteen k of a row --
    // neighbouring lanes of the staging loop -- fell into ONE pair of LDS banks (a 16-way conflict on every write, in a prologue
    // that all resident workgroups run at the same time); 66 spreads them over 32 banks and keeps the 16-byte reads aligned
    constexpr int XLD = 66;
    __shared__ __attribute__((aligned(16))) double Xi[DP * XLD];
    __shared__ __attribute__((aligned(16))) double Xj[DP * XLD];
    __shared__ double ai[64], aj[64];
    __shared__ double red[4][DP + 1];
    int bi, bj; tri_decode(blockIdx.x, &bi, &bj);
    const int t = threadIdx.x, tx = t & 15, ty = t >> 4;
    for (int e = t; e < 64 * DP; e += 256) {
        int row = e / DP, k = e - row * DP;
        double vi = 0.0, vj = 0.0;
        if (k < kp.dpad) {
            vi = Xs[((int64_t)bi * 64 + row) * kp.dpad + k];
            vj = Xs[((int64_t)bj * 64 + row) * kp.dpad + k];
        }
        Xi[k * XLD + row] = vi; Xj[k * XLD + row] = vj;
    }
    if (t < 64) { ai[t] = alpha[bi * 64 + t]; aj[t] = alpha[bj * 64 + t]; }
    __syncthreads();
    double g[DP + 1];
#pragma unroll
    for (int k = 0; k <= DP; k++) g[k] = 0.0;
    // A thread owns 4 x 4 pairs: rows ty * 4 + a of the i block, rows tx * 4 + b of the j block.  Round 5: the coordinate loop is
    // the OUTER one -- per coordinate four 16-byte LDS reads feed all sixteen pairs (until then each of the four a re-read its
    // coordinate and all four of the j side: twelve reads per coordinate, the kernel waited for LDS at a third of the FP64
    // vector rate: 79 us at N = 4096).  Every sum runs in the order it had: r2 over k ascending, g[0] and g[1 + k] over
    // (a, b) with a outer -- the same bits.
    double r2[4][4];
#pragma unroll
    for (int a = 0; a < 4; a++)
#pragma unroll
        for (int b = 0; b < 4; b++) r2[a][b] = 0.0;
    int ci = ty * 4, cj = tx * 4;
    // K^-1 of the sixteen pairs, on its way while the distances are summed (Kinv holds the lower triangle; off the diagonal
    // blocks a thread's four columns are 32 contiguous, aligned bytes of one row)
    double kin[4][4];
    if (bi != bj) {
#pragma unroll
        for (int a = 0; a < 4; a++) {
            const double2* pk = reinterpret_cast<const double2*>(Kinv + ((int64_t)bi * 64 + ci + a) * ld + (int64_t)bj * 64 + cj);
            const double2 k0 = pk[0], k1 = pk[1];
            kin[a][0] = k0.x; kin[a][1] = k0.y; kin[a][2] = k1.x; kin[a][3] = k1.y;
        }
    } else {
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) {
                const int64_t i = (int64_t)bi * 64 + ci + a, j = (int64_t)bj * 64 + cj + b;
                kin[a][b] = Kinv[(i > j ? i : j) * ld + (i > j ? j : i)];
            }
    }
#pragma unroll
    for (int k = 0; k < DP; k++) {
        // (two coordinates' reads in flight: the offsets are laundered per pair of coordinates and the scheduler is fenced
        // behind it -- left alone, the compiler issues all 4 DP reads of a pass first and spills them, 64 DP bytes per lane)
        if ((k & 1) == 0) asm volatile("" : "+v"(ci), "+v"(cj));
        const double2* pi = reinterpret_cast<const double2*>(Xi + k * XLD + ci);
        const double2* pj = reinterpret_cast<const double2*>(Xj + k * XLD + cj);
        const double2 i0 = pi[0], i1 = pi[1], j0 = pj[0], j1 = pj[1];
        const double xi[4] = {i0.x, i0.y, i1.x, i1.y}, xj[4] = {j0.x, j0.y, j1.x, j1.y};
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) { const double df = xi[a] - xj[b]; r2[a][b] = fma(df, df, r2[a][b]); }
        if (k & 1) {
            // (... and the sums pass through an empty asm: the multiply-adds of this pair of coordinates stay in front of it --
            // pure arithmetic is otherwise placed behind ALL reads of the unrolled loop)
            asm volatile("" : "+v"(r2[0][0]), "+v"(r2[0][1]), "+v"(r2[0][2]), "+v"(r2[0][3]), "+v"(r2[1][0]), "+v"(r2[1][1]), "+v"(r2[1][2]), "+v"(r2[1][3]),
                              "+v"(r2[2][0]), "+v"(r2[2][1]), "+v"(r2[2][2]), "+v"(r2[2][3]), "+v"(r2[3][0]), "+v"(r2[3][1]), "+v"(r2[3][2]), "+v"(r2[3][3]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // r2 becomes w C h (what the second pass weighs the squared differences with)
#pragma unroll
    for (int a = 0; a < 4; a++) {
        const int il = ci + a;
        const int64_t i = (int64_t)bi * 64 + il;
#pragma unroll
        for (int b = 0; b < 4; b++) {
            const int jl = cj + b;
            const int64_t j = (int64_t)bj * 64 + jl;
            double w = ai[il] * aj[jl] - kin[a][b];          // (the buffers are padded to Np, padding rows are zero)
            double kv;
            const double h = corr_and_h<KID>(r2[a][b], &kv);
            if (i == j) kv = 1.0;
            if (i >= kp.N || j >= kp.N) w = 0.0;
            g[0] = fma(w, kp.C * kv, g[0]);
            r2[a][b] = w * kp.C * h;
        }
        __builtin_amdgcn_sched_barrier(0);      // four kernel evaluations in flight, not sixteen (their temporaries: 256 VGPRs)
    }
    // (the offsets are laundered: otherwise the 8 DP coordinates of the first pass stay live across the kernel evaluations to be
    // reused here -- 256 VGPRs, one wave per SIMD)
    asm volatile("" : "+v"(ci), "+v"(cj));
#pragma unroll
    for (int k = 0; k < DP; k++) {
        // (two coordinates' reads in flight: the offsets are laundered per pair of coordinates and the scheduler is fenced
        // behind it -- left alone, the compiler issues all 4 DP reads of a pass first and spills them, 64 DP bytes per lane)
        if ((k & 1) == 0) asm volatile("" : "+v"(ci), "+v"(cj));
        const double2* pi = reinterpret_cast<const double2*>(Xi + k * XLD + ci);
        const double2* pj = reinterpret_cast<const double2*>(Xj + k * XLD + cj);
        const double2 i0 = pi[0], i1 = pi[1], j0 = pj[0], j1 = pj[1];
        const double xi[4] = {i0.x, i0.y, i1.x, i1.y}, xj[4] = {j0.x, j0.y, j1.x, j1.y};
#pragma unroll
        for (int a = 0; a < 4; a++)
#pragma unroll
            for (int b = 0; b < 4; b++) { const double df = xi[a] - xj[b]; g[1 + k] = fma(r2[a][b], df * df, g[1 + k]); }
        if (k & 1) {
            asm volatile("" : "+v"(g[k]), "+v"(g[1 + k]));
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const int lane = t & 63, wave = t >> 6;
#pragma unroll
    for (int k = 0; k <= DP; k++) {
        double v = g[k];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (t <= DP) {
        double s = (red[0][t] + red[1][t]) + (red[2][t] + red[3][t]);
        if (bi != bj) s *= 2.0;
        part[(int64_t)blockIdx.x * (DP + 1) + t] = s;
    }
}

// one workgroup per hyperparameter: strided partial sums then a fixed LDS tree
__global__ __launch_bounds__(256) void reduce_traces_kernel(const double* __restrict__ part_, int64_t ntile,
                                                            int stride, double* __restrict__ out_,
                                                            const int* __restrict__ info_, const double* __restrict__ lq_,
                                                            double* __restrict__ host_res, int info_at, int64_t bstride) {
    __shared__ double red[256];
    const int tb = (int)blockIdx.z;             // theta of a batched launch: its results go to their own row of host_res
    const double* __restrict__ part = bset(part_, tb, bstride);
    double* __restrict__ out = bset(out_, tb, bstride);
    const int* __restrict__ info = bset(info_, tb, bstride);
    const double* __restrict__ lq = bset(lq_, tb, bstride);
    if (host_res) host_res += (int64_t)tb * GPRY_BRES_STRIDE;
    const int k = blockIdx.x, t = threadIdx.x;
    double s = 0.0;
    for (int64_t i = t; i < ntile; i += 256) s += part[i * stride + k];
    red[t] = s;
    __syncthreads();
    for (int w = 128; w >= 1; w >>= 1) {
        if (t < w) red[t] += red[t + w];
        __syncthreads();
    }
    if (t == 0) {
        out[k] = 0.5 * red[0];
        // last kernel of an evaluation with gradient: results and factorisation status also go straight into the
        // mapped host buffer (read by the host after the stream wait: no copy-out operations)
        if (host_res) {
            host_res[2 + k] = 0.5 * red[0];
            if (k == 0) {
                host_res[0] = lq[0]; host_res[1] = lq[1];
                host_res[info_at] = (double)info[0]; host_res[info_at + 1] = (double)info[1];
                host_res[info_at + 2] = (double)info[3];       // (0x5A..: a bounded wait of the panel step ran out -- not a verdict on the matrix)
            }
        }
    }
}

int launch_lml_traces(gpry_ctx* ctx, const double* Kinv, const double* alpha, double* grad_out_dev, const double* lq_dev,
                      double* host_res, int info_at) {
    KernParams kp = make_kp(ctx);
    int64_t nb = ctx->Np / 64;
    int64_t ntile = nb * (nb + 1) / 2;
    if (ctx->d > 32) return gpry_fail(ctx, -1, "d > 32 is not supported");
    int DPsel = ctx->d <= 4 ? 4 : ctx->d <= 8 ? 8 : ctx->d <= 16 ? 16 : ctx->d <= 24 ? 24 : 32;
    int64_t need = ntile * (DPsel + 1);
    if (need > ctx->part_cap && ctx->bpar) return gpry_fail(ctx, -1, "batched chain: partial sums exceed the arena");
    if (need > ctx->part_cap) {
        if (ctx->dpart) dev_free(ctx, ctx->dpart);
        GPRY_TRY(dev_alloc(ctx, &ctx->dpart, need));
        ctx->part_cap = need;
    }
#define LT2(DP, KID) hipLaunchKernelGGL((lml_traces_kernel<DP, KID>), dim3((unsigned)ntile, 1, (unsigned)ctx->bn), dim3(256), 0, \
                                        ctx->stream, ctx->dXs, Kinv, ctx->Np, alpha, ctx->dpart, kp, ctx->bpar, ctx->bstride)
#define LT4(KID) { if (DPsel == 4) LT2(4, KID); else if (DPsel == 8) LT2(8, KID); \
                   else if (DPsel == 16) LT2(16, KID); else if (DPsel == 24) LT2(24, KID); else LT2(32, KID); }
    DISPATCH_KID(ctx->kernel_id, LT4)
#undef LT4
#undef LT2
    HIP_TRY(ctx, hipGetLastError());
    hipLaunchKernelGGL(reduce_traces_kernel, dim3((unsigned)(ctx->d + 1), 1, (unsigned)ctx->bn), dim3(256), 0, ctx->stream,
                       ctx->dpart, ntile, DPsel + 1, grad_out_dev, ctx->dinfo, lq_dev, host_res, info_at, ctx->bstride);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// x-gradients for one point (gpry/gpr.py:1236-1266 with
// gpry/kernels.py:257-278 RBF, :326-432 Matern, :687-699 product rule): for every training row j
//   kstar[j] = C k(x, X_j),   G[j][k] = C d k(x, X_j) / d x_k     (coordinates of the kernel)
// with diff = (x - X_j) / l, r = |diff|:
//   RBF         -exp(-r^2/2) diff/l
//   Matern 1/2  -exp(-r)/r diff/l        (r = 0: -1/l, the reference's fill value)
//   Matern 3/2  -3 exp(-sqrt3 r) diff/l
//   Matern 5/2  -(5/3) (1 + sqrt5 r) exp(-sqrt5 r) diff/l
// Thread <-> training row; rows j >= N give zeros.
template <int KID>
__global__ __launch_bounds__(256) void gradx_kernel(const double* __restrict__ x, const double* __restrict__ Xs,
                                                    double* __restrict__ kstar, double* __restrict__ G,
                                                    int64_t Np, KernParams kp, AffParams ap) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= Np) return;
    // fully unrolled over the maximum dimension with guards: run-time indices into `diff` and into the
    // by-value parameter arrays would put both in scratch memory
    double diff[32];      // set_train refuses d > 32
    double r2 = 0.0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        diff[k] = 0.0;
        if (k < kp.d) {
            double v = x[k];
            if (kp.has_aff) v = (v - ap.lo[k]) / ap.span[k];
            v = v / ap.ls[k];
            diff[k] = v - Xs[j * kp.dpad + k];
            r2 = fma(diff[k], diff[k], r2);
        }
    }
    const bool real = j < kp.N;
    double kv, coef;      // k(r) and the factor of diff/l
    if (KID == GPRY_RBF) { kv = exp(-0.5 * r2); coef = -kv; }
    else if (KID == GPRY_MATERN12) { double r = sqrt(r2); kv = exp(-r); coef = r != 0.0 ? -kv / r : 0.0; }
    else if (KID == GPRY_MATERN32) { double t = sqrt(r2) * SQRT3; double e = exp(-t); kv = (1.0 + t) * e; coef = -3.0 * e; }
    else { double t = sqrt(r2) * SQRT5; double e = exp(-t); kv = (1.0 + t + t * t * (1.0 / 3.0)) * e; coef = -(5.0 / 3.0) * (1.0 + t) * e; }
    kstar[j] = real ? kp.C * kv : 0.0;
#pragma unroll
    for (int k = 0; k < 32; k++) {
        if (k < kp.dpad) {
            double g = 0.0;
            if (real && k < kp.d) {
                g = kp.C * coef * diff[k] / ap.ls[k];
                if (KID == GPRY_MATERN12 && r2 == 0.0) g = -kp.C / ap.ls[k];
            }
            G[j * kp.dpad + k] = g;
        }
    }
}

// u = V kstar (V lower triangular, row-major): one wave per row
__global__ __launch_bounds__(256) void gx_trmv_lower_kernel(const double* __restrict__ V, int64_t ld,
                                                         const double* __restrict__ x, double* __restrict__ y, int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double s = 0.0;
    for (int64_t k = lane; k <= i; k += 64) s = fma(V[i * ld + k], x[k], s);
    for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) y[i] = s;
}
// partial sums of w = V^T u: block (cb, rb) covers columns cb*256.. and rows rb*128..; thread <->
// column (coalesced along the row); part[rb][j].  Deterministic (no atomics).
__global__ __launch_bounds__(256) void gx_trmv_lower_t_part_kernel(const double* __restrict__ V, int64_t ld,
                                                                const double* __restrict__ u,
                                                                double* __restrict__ part, int64_t n) {
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i0 = (int64_t)blockIdx.y * 128;
    double s = 0.0;
    if (j < n && i0 + 127 >= j) {
        // eight loads in flight per step (one load per multiply-add was a chain of up to 128 memory latencies: 15 us per
        // call whatever the size); rows above the diagonal enter as 0 * 0, which leaves the bits of the sum alone
        for (int64_t ib = i0; ib < i0 + 128; ib += 8) {
            double v[8], uu[8];
#pragma unroll
            for (int q = 0; q < 8; q++) {
                const int64_t i = ib + q;
                const bool in = i >= j && i < n;
                v[q] = in ? V[i * ld + j] : 0.0;
                uu[q] = in ? u[i] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; q++) s = fma(v[q], uu[q], s);
        }
    }
    if (j < n) part[(int64_t)blockIdx.y * n + j] = s;
}
__global__ void gx_colsum_kernel(const double* __restrict__ part, int nrow, int64_t n, double* __restrict__ out) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    double s = 0.0;
    for (int r = 0; r < nrow; r++) s += part[(int64_t)r * n + j];
    out[j] = s;
}
// out[k] = sum_j G[j][k] a[j] and out[dpad + k] = sum_j G[j][k] w[j]: one workgroup per k
__global__ __launch_bounds__(256) void gradx_contract_kernel(const double* __restrict__ G, int dpad,
                                                             const double* __restrict__ a, const double* __restrict__ w,
                                                             int64_t n, double* __restrict__ out) {
    __shared__ double ra[256], rw[256];
    const int k = blockIdx.x, t = threadIdx.x;
    double sa = 0.0, sw = 0.0;
    for (int64_t j = t; j < n; j += 256) {
        const double g = G[j * dpad + k];
        sa = fma(g, a[j], sa);
        if (w) sw = fma(g, w[j], sw);
    }
    ra[t] = sa; rw[t] = sw;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) { ra[t] += ra[t + s]; rw[t] += rw[t + s]; }
        __syncthreads();
    }
    if (t == 0) { out[k] = ra[0]; out[dpad + k] = rw[0]; }
}

// x: device pointer to the point (d doubles, raw or transformed as in gpry_predict).
// Outputs (device): kstar[Np], G[Np x dpad], out[2*dpad] = {G^T alpha_, G^T K^-1 kstar};
// scratch u, w (Np each), part ((Np/128) x Np).
int launch_gradx(gpry_ctx* ctx, const double* x, int raw_affine, int want_kinv, double* kstar, double* G,
                 double* u, double* w, double* part, double* out) {
    GPRY_TRY(ensure_pred_xs(ctx));
    KernParams kp = make_kp(ctx);
    kp.has_aff = raw_affine && ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff);
    const int64_t Np = ctx->Np;
    hipStream_t st = ctx->stream;
#define GX(KID) hipLaunchKernelGGL((gradx_kernel<KID>), dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, st, \
                                   x, ctx->dXs, kstar, G, Np, kp, ap)
    DISPATCH_KID(ctx->kernel_id, GX)
#undef GX
    if (want_kinv) {
        hipLaunchKernelGGL(gx_trmv_lower_kernel, dim3((unsigned)((Np + 3) / 4)), dim3(256), 0, st, ctx->dV, Np, kstar, u, Np);
        const int nrb = (int)(Np / 128);
        hipLaunchKernelGGL(gx_trmv_lower_t_part_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)nrb), dim3(256), 0, st,
                           ctx->dV, Np, u, part, Np);
        hipLaunchKernelGGL(gx_colsum_kernel, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, st, part, nrb, Np, w);
    }
    hipLaunchKernelGGL(gradx_contract_kernel, dim3((unsigned)ctx->dpad), dim3(256), 0, st, G, ctx->dpad,
                       ctx->dalpha_, want_kinv ? w : nullptr, Np, out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// x-gradients for a BATCH of points (the restarts of an acquisition optimiser evaluated side by side,
// gpry/gp_acquisition.py:270-389): workgroup <-> point i, threads over the training rows; G_i is never
// stored.  With w_i = K^-1 k*_i as column i of Wm (Np x ldw, may be NULL):
//     out[i][c]        = sum_j G_i[j][c] alpha_[j]          (-> mean gradient)
//     out[i][dpad + c] = sum_j G_i[j][c] Wm[j][i]           (-> std gradient)
// Same per-pair arithmetic as gradx_kernel (exact exp / sqrt), fixed-order tree reduction.
template <int KID>
__global__ __launch_bounds__(256) void gradx_batch_kernel(const double* __restrict__ Xb, const double* __restrict__ Xs,
                                                          const double* __restrict__ alpha_, const double* __restrict__ Wm,
                                                          int64_t ldw, double* __restrict__ out, KernParams kp, AffParams ap) {
    __shared__ double red[256];
    const int64_t i = blockIdx.x;
    const int t = threadIdx.x;
    double xs[32], sa[32], sw[32];
#pragma unroll
    for (int k = 0; k < 32; k++) {
        sa[k] = 0.0; sw[k] = 0.0; xs[k] = 0.0;
        if (k < kp.d) {
            double v = Xb[i * kp.d + k];
            if (kp.has_aff) v = (v - ap.lo[k]) / ap.span[k];
            xs[k] = v / ap.ls[k];
        }
    }
    for (int64_t j = t; j < kp.N; j += 256) {
        double diff[32];
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            diff[k] = 0.0;
            if (k < kp.d) { diff[k] = xs[k] - Xs[j * kp.dpad + k]; r2 = fma(diff[k], diff[k], r2); }
        }
        double coef;
        if (KID == GPRY_RBF) coef = -exp(-0.5 * r2);
        else if (KID == GPRY_MATERN12) { const double r = sqrt(r2); coef = r != 0.0 ? -exp(-r) / r : 0.0; }
        else if (KID == GPRY_MATERN32) coef = -3.0 * exp(-sqrt(r2) * SQRT3);
        else { const double tt = sqrt(r2) * SQRT5; coef = -(5.0 / 3.0) * (1.0 + tt) * exp(-tt); }
        const double a = alpha_[j], w = Wm ? Wm[j * ldw + i] : 0.0;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if (k < kp.d) {
                double g = kp.C * coef * diff[k] / ap.ls[k];
                if (KID == GPRY_MATERN12 && r2 == 0.0) g = -kp.C / ap.ls[k];
                sa[k] = fma(g, a, sa[k]);
                sw[k] = fma(g, w, sw[k]);
            }
        }
    }
    // 2 d block reductions (d <= 32): cheap next to the pair loop
    for (int k = 0; k < kp.d; k++) {
        for (int which = 0; which < 2; which++) {
            double v = 0.0;
#pragma unroll
            for (int q = 0; q < 32; q++) if (q == k) v = which ? sw[q] : sa[q];
            red[t] = v;
            __syncthreads();
            for (int s = 128; s >= 1; s >>= 1) {
                if (t < s) red[t] += red[t + s];
                __syncthreads();
            }
            if (t == 0) out[i * 2 * kp.dpad + which * kp.dpad + k] = red[0];
            __syncthreads();
        }
    }
}
int launch_gradx_batch(gpry_ctx* ctx, const double* Xb, int64_t m, int raw_affine, const double* Wm, int64_t ldw,
                       double* out) {
    GPRY_TRY(ensure_pred_xs(ctx));
    KernParams kp = make_kp(ctx);
    kp.has_aff = raw_affine && ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff);
#define GXB(KID) hipLaunchKernelGGL((gradx_batch_kernel<KID>), dim3((unsigned)m), dim3(256), 0, ctx->stream, \
                                    Xb, ctx->dXs, ctx->dalpha_, Wm, ldw, out, kp, ap)
    DISPATCH_KID(ctx->kernel_id, GXB)
#undef GXB
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Low-latency posterior mean for small batches (the per-point
// calls of nested samplers / MCMC, gpry/gp_acquisition.py:769-793, gpry/mc.py:387-391).
// One workgroup per point: mean = y_std * sum_j alpha_j C k(x, X_j) + y_mean, clipped and
// masked as gpry_predict does (gpry/gpr.py:1180-1201) -- one launch, no panel, no partials.
template <int DP, int KID>
__global__ __launch_bounds__(256) void predict_mean_small_kernel(
    const double* __restrict__ Xc, const double* __restrict__ Xs, const double* __restrict__ alpha_,
    double* __restrict__ part_out, int64_t rows_per_split, KernParams kp, AffParams ap) {
    __shared__ double r2s[MEAN_SLICE_CH];
    __shared__ double red[256];
    const int64_t m = blockIdx.x;
    // blockIdx.y: which slice of the training rows (the host adds the slices up)
    const double v = mean_slice<DP, KID>(Xc + m * kp.d, Xs, alpha_, (int64_t)blockIdx.y * rows_per_split, rows_per_split,
                                         kp, ap, r2s, red);
    if (threadIdx.x == 0) part_out[m * gridDim.y + blockIdx.y] = v;
}

// part_out: M x nsplit partial sums of K* alpha_ (transformed units); the caller adds them and
// applies y_std, y_mean, the clip and the mask.
int launch_predict_mean_small(gpry_ctx* ctx, const double* Xc, int64_t M, int nsplit, double* part_out) {
    GPRY_TRY(ensure_pred_xs(ctx));
    KernParams kp = make_kp(ctx);
    kp.has_aff = ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff);
    if (ctx->d > 32) return gpry_fail(ctx, -1, "d > 32 is not supported");
    const int64_t rows_per_split = round_up((ctx->N + nsplit - 1) / nsplit, 32);
#define PM2(DP, KID) hipLaunchKernelGGL((predict_mean_small_kernel<DP, KID>), dim3((unsigned)M, (unsigned)nsplit), \
                                        dim3(256), 0, ctx->stream, Xc, ctx->dXs, ctx->dalpha_, part_out,          \
                                        rows_per_split, kp, ap)
#define PM4(KID) { if (ctx->d <= 4) PM2(4, KID); else if (ctx->d <= 8) PM2(8, KID); \
                   else if (ctx->d <= 16) PM2(16, KID); else PM2(32, KID); }
    DISPATCH_KID(ctx->kernel_id, PM4)
#undef PM4
#undef PM2
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Mean AND std for a handful of points (M <= 16): the panel path costs ~1 ms at N = 4096
// whatever M is (its longest triangular tile runs 256 slabs on one workgroup); here
//   kstar_rows_kernel   k*[m][j] (thread <-> training row) + alpha-weighted partials of the mean
//   trmv_multi_kernel   u_m = V k*_m for all m at once: a wave owns 4 rows of V, reads them once
//                       and keeps 4 x M accumulators; per-workgroup partials of |u_m|^2
// and the host adds the partials in a fixed order (pinned, zero-copy buffers).
template <int DP, int KID>
__global__ __launch_bounds__(256) void kstar_rows_kernel(
    const double* __restrict__ Xc, const double* __restrict__ Xs, const double* __restrict__ alpha_,
    double* __restrict__ kstar, double* __restrict__ mean_part, int64_t Np, KernParams kp, AffParams ap) {
    __shared__ double xs_s[DP];
    __shared__ double red[256];
    const int m = blockIdx.y, t = threadIdx.x;
    if (t < DP) {
        double v = 0.0;
        if (t < kp.d) {
            v = Xc[(int64_t)m * kp.d + t];
            if (kp.has_aff) v = (v - ap.lo[t]) / ap.span[t];
            v = v / ap.ls[t];
        }
        xs_s[t] = v;
    }
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * 256 + t;
    double kv = 0.0;
    if (j < kp.N) {
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < DP; k++) {
            if (k < kp.dpad) {
                double df = xs_s[k] - Xs[j * kp.dpad + k];
                r2 = fma(df, df, r2);
            }
        }
        kv = kp.C * corr_r2_fast<KID>(r2);
    }
    if (j < Np) kstar[(int64_t)m * Np + j] = kv;
    red[t] = j < kp.N ? alpha_[j] * kv : 0.0;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if (t < s) red[t] += red[t + s];
        __syncthreads();
    }
    if (t == 0) mean_part[(int64_t)m * gridDim.x + blockIdx.x] = red[0];
}

template <int MM>
__global__ __launch_bounds__(256) void trmv_multi_kernel(const double* __restrict__ V, int64_t ld,
                                                         const double* __restrict__ kstar, int M,
                                                         double* __restrict__ ss_part, double* __restrict__ u_out = nullptr) {
    __shared__ double wsum[4][MM];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t i0 = (int64_t)blockIdx.x * 16 + w * 4;
    double acc[4][MM];
#pragma unroll
    for (int r = 0; r < 4; r++)
#pragma unroll
        for (int m = 0; m < MM; m++) acc[r][m] = 0.0;
    for (int64_t k = lane; k <= i0 + 3; k += 64) {
        double v[4];
#pragma unroll
        for (int r = 0; r < 4; r++) v[r] = V[(i0 + r) * ld + k];     // zero above the diagonal
#pragma unroll
        for (int m = 0; m < MM; m++) {
            if (m < M) {
                const double ks = kstar[(int64_t)m * ld + k];
#pragma unroll
                for (int r = 0; r < 4; r++) acc[r][m] = fma(v[r], ks, acc[r][m]);
            }
        }
    }
#pragma unroll
    for (int m = 0; m < MM; m++) {
        double ss = 0.0;
        if (m < M) {
#pragma unroll
            for (int r = 0; r < 4; r++) {
                double u = acc[r][m];
                for (int o = 32; o >= 1; o >>= 1) u += __shfl_xor(u, o);
                ss = fma(u, u, ss);
                if (m == 0 && u_out != nullptr && lane == 0) u_out[i0 + r] = u;      // (one-point calls that go on to V^T u)
            }
        }
        if (lane == 0) wsum[w][m] = ss;
    }
    __syncthreads();
    if (threadIdx.x < M)
        ss_part[(int64_t)threadIdx.x * gridDim.x + blockIdx.x] =
            (wsum[0][threadIdx.x] + wsum[1][threadIdx.x]) + (wsum[2][threadIdx.x] + wsum[3][threadIdx.x]);
}

// kstar: M x Np device scratch; mean_part: M x (Np/256), ss_part: M x (Np/16) (pinned host is fine)
int launch_predict_small_std(gpry_ctx* ctx, const double* Xc, int M, double* kstar, double* mean_part,
                             double* ss_part) {
    GPRY_TRY(ensure_pred_xs(ctx));
    KernParams kp = make_kp(ctx);
    kp.has_aff = ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff);
    const int64_t Np = ctx->Np;
    if (M > 16) return gpry_fail(ctx, -1, "predict_small_std: M > 16");
    dim3 grid((unsigned)(Np / 256 + (Np % 256 ? 1 : 0)), (unsigned)M);
#define KS2(DP, KID) hipLaunchKernelGGL((kstar_rows_kernel<DP, KID>), grid, dim3(256), 0, ctx->stream, Xc, ctx->dXs, \
                                        ctx->dalpha_, kstar, mean_part, Np, kp, ap)
#define KS4(KID) { if (ctx->d <= 4) KS2(4, KID); else if (ctx->d <= 8) KS2(8, KID); \
                   else if (ctx->d <= 16) KS2(16, KID); else KS2(32, KID); }
    DISPATCH_KID(ctx->kernel_id, KS4)
#undef KS4
#undef KS2
    if (M <= 4) hipLaunchKernelGGL(trmv_multi_kernel<4>, dim3((unsigned)(Np / 16)), dim3(256), 0, ctx->stream, ctx->dV, Np, kstar, M, ss_part);
    else hipLaunchKernelGGL(trmv_multi_kernel<16>, dim3((unsigned)(Np / 16)), dim3(256), 0, ctx->stream, ctx->dV, Np, kstar, M, ss_part);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// ONE point, everything the acquisition optimiser asks of it in one call (gpry/gpr.py:1236-1266 as driven by
// gpry/gp_acquisition.py:309-342: posterior mean, std and both x-gradients of a single point per L-BFGS step):
//   gradx_rows_kernel      k*[j], G[j][k] = d k*_j / d x_k (the arithmetic of gradx_kernel) + per-workgroup partials of k*.alpha
//   trmv_multi_kernel<4>   u = V k* (stored) + partials of |u|^2
//   gx_trmv_lower_t_part   partials of w = V^T u                       (only when the std gradient is wanted)
//   gradx_contract1_kernel G^T alpha and G^T w, w summed from its partials on the fly
// The partials and the 2 d contractions land in the pinned, device-mapped staging buffer; the host adds them in a fixed order.
// Four launches and one stream wait instead of predict (2 launches, a wait) + predict_grad (5 launches, two copies, a wait).
template <int KID>
__global__ __launch_bounds__(256) void gradx_rows_kernel(const double* __restrict__ x, const double* __restrict__ Xs,
                                                         const double* __restrict__ alpha_, double* __restrict__ kstar,
                                                         double* __restrict__ G, double* __restrict__ mean_part,
                                                         int64_t Np, KernParams kp, AffParams ap) {
    __shared__ double red[256];
    const int t = threadIdx.x;
    const int64_t j = (int64_t)blockIdx.x * 256 + t;
    double ks = 0.0;
    if (j < Np) {
        double diff[32];
        double r2 = 0.0;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            diff[k] = 0.0;
            if (k < kp.d) {
                double v = x[k];
                if (kp.has_aff) v = (v - ap.lo[k]) / ap.span[k];
                v = v / ap.ls[k];
                diff[k] = v - Xs[j * kp.dpad + k];
                r2 = fma(diff[k], diff[k], r2);
            }
        }
        const bool real = j < kp.N;
        double kv, coef;
        if (KID == GPRY_RBF) { kv = exp(-0.5 * r2); coef = -kv; }
        else if (KID == GPRY_MATERN12) { double r = sqrt(r2); kv = exp(-r); coef = r != 0.0 ? -kv / r : 0.0; }
        else if (KID == GPRY_MATERN32) { double tt = sqrt(r2) * SQRT3; double e = exp(-tt); kv = (1.0 + tt) * e; coef = -3.0 * e; }
        else { double tt = sqrt(r2) * SQRT5; double e = exp(-tt); kv = (1.0 + tt + tt * tt * (1.0 / 3.0)) * e; coef = -(5.0 / 3.0) * (1.0 + tt) * e; }
        ks = real ? kp.C * kv : 0.0;
        kstar[j] = ks;
#pragma unroll
        for (int k = 0; k < 32; k++) {
            if (k < kp.dpad) {
                double g = 0.0;
                if (real && k < kp.d) {
                    g = kp.C * coef * diff[k] / ap.ls[k];
                    if (KID == GPRY_MATERN12 && r2 == 0.0) g = -kp.C / ap.ls[k];
                }
                G[j * kp.dpad + k] = g;
            }
        }
    }
    red[t] = j < kp.N ? alpha_[j] * ks : 0.0;
    __syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (t < sft) red[t] += red[t + sft];
        __syncthreads();
    }
    if (t == 0) mean_part[blockIdx.x] = red[0];
}
// out[k] = sum_j G[j][k] alpha[j], out[dpad + k] = sum_j G[j][k] w[j] with w[j] = sum_r part[r][j] (ascending r, as gx_colsum_kernel)
__global__ __launch_bounds__(256) void gradx_contract1_kernel(const double* __restrict__ G, int dpad,
                                                              const double* __restrict__ a, const double* __restrict__ part,
                                                              int nrow, int64_t n, double* __restrict__ out) {
    __shared__ double ra[256], rw[256];
    const int k = blockIdx.x, t = threadIdx.x;
    double sa = 0.0, sw = 0.0;
    for (int64_t j = t; j < n; j += 256) {
        const double g = G[j * dpad + k];
        sa = fma(g, a[j], sa);
        if (part) {
            double w = 0.0;
            for (int r = 0; r < nrow; r++) w += part[(int64_t)r * n + j];
            sw = fma(g, w, sw);
        }
    }
    ra[t] = sa; rw[t] = sw;
    __syncthreads();
    for (int sft = 128; sft >= 1; sft >>= 1) {
        if (t < sft) { ra[t] += ra[t + sft]; rw[t] += rw[t + sft]; }
        __syncthreads();
    }
    if (t == 0) { out[k] = ra[0]; out[dpad + k] = rw[0]; }
}
// x, mean_part (Np/256 doubles), ss_part (Np/16), out (2 dpad): pinned, device-mapped; kstar, u: Np; G: Np x dpad; part: (Np/128) x Np
int launch_point_full(gpry_ctx* ctx, const double* x, int want_kinv, double* kstar, double* G, double* u, double* part,
                      double* mean_part, double* ss_part, double* out) {
    GPRY_TRY(ensure_pred_xs(ctx));
    KernParams kp = make_kp(ctx);
    kp.has_aff = ctx->tf.has_x_affine;
    AffParams ap = make_ap(ctx, kp.has_aff);
    const int64_t Np = ctx->Np;
    hipStream_t st = ctx->stream;
#define GR(KID) hipLaunchKernelGGL((gradx_rows_kernel<KID>), dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, st, \
                                   x, ctx->dXs, ctx->dalpha_, kstar, G, mean_part, Np, kp, ap)
    DISPATCH_KID(ctx->kernel_id, GR)
#undef GR
    hipLaunchKernelGGL(trmv_multi_kernel<4>, dim3((unsigned)(Np / 16)), dim3(256), 0, st, ctx->dV, Np, kstar, 1, ss_part, u);
    const int nrb = (int)(Np / 128);
    const double* wsrc = nullptr;
    int wrows = 0;
    if (want_kinv) {
        hipLaunchKernelGGL(gx_trmv_lower_t_part_kernel, dim3((unsigned)((Np + 255) / 256), (unsigned)nrb), dim3(256), 0, st,
                           ctx->dV, Np, u, part, Np);
        wsrc = part; wrows = nrb;
        if (nrb > 8) {      // many row blocks: add them up once (into k*, which is no longer needed) instead of once per dimension
            hipLaunchKernelGGL(gx_colsum_kernel, dim3((unsigned)((Np + 255) / 256)), dim3(256), 0, st, part, nrb, Np, kstar);
            wsrc = kstar; wrows = 1;
        }
    }
    hipLaunchKernelGGL(gradx_contract1_kernel, dim3((unsigned)ctx->dpad), dim3(256), 0, st, G, ctx->dpad, ctx->dalpha_,
                       wsrc, wrows, Np, out);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// Gates of the sweep on the device: per candidate
//   GPRY_MASK_OUTSIDE_TRUST   raw x outside the closed trust box (gpry/gpr.py:1107-1112, tools.py:263-290)
//   GPRY_MASK_CLASSIFIED_INF  the SVM says "infinite": decision = sum_i coef_i exp(-gamma |x_ - sv_i|^2)
//                             + intercept <= 0 in transformed coordinates (gpry/svm.py:308-347 ->
//                             libsvm _dense_predict, sklearn SVC(kernel="rbf"), two classes)
// ORed into mask[m] (which already holds the caller's bits or zeros).  Thread <-> candidate; the
// support vectors pass through LDS in chunks of 128.
template <int DP>
__global__ __launch_bounds__(256) void gates_kernel(const double* __restrict__ Xc, int64_t M, int d,
                                                    const double* __restrict__ sv, const double* __restrict__ coef,
                                                    int64_t n_sv, double gamma, double intercept, int positive_is_finite,
                                                    const double* __restrict__ trust, int has_trust, int has_aff,
                                                    AffParams ap, uint8_t* __restrict__ mask) {
    __shared__ double svl[128 * DP];
    __shared__ double cl[128];
    const int t = threadIdx.x;
    const int64_t m = (int64_t)blockIdx.x * 256 + t;
    const bool live = m < M;
    double x_[DP];
    unsigned bits = 0;
#pragma unroll
    for (int k = 0; k < DP; k++) {
        double v = 0.0;
        if (k < d && live) {
            v = Xc[m * d + k];
            if (has_trust && !(v >= trust[2 * k] && v <= trust[2 * k + 1])) bits |= GPRY_MASK_OUTSIDE_TRUST;
            if (has_aff) v = (v - ap.lo[k]) / ap.span[k];
        }
        x_[k] = v;
    }
    if (n_sv > 0) {
        double dec = 0.0;
        for (int64_t s0 = 0; s0 < n_sv; s0 += 128) {
            const int ns = (int)((n_sv - s0 < 128) ? n_sv - s0 : 128);
            __syncthreads();
            for (int e = t; e < ns * DP; e += 256) {
                const int r = e / DP, k = e - r * DP;
                svl[e] = k < d ? sv[(s0 + r) * d + k] : 0.0;
            }
            if (t < ns) cl[t] = coef[s0 + t];
            __syncthreads();
            for (int r = 0; r < ns; r++) {
                double r2 = 0.0;
#pragma unroll
                for (int k = 0; k < DP; k++) {
                    const double df = x_[k] - svl[r * DP + k];
                    r2 = fma(df, df, r2);
                }
                dec = fma(cl[r], fast_exp_neg(gamma * r2), dec);
            }
        }
        dec += intercept;
        const bool finite = positive_is_finite ? dec > 0.0 : !(dec > 0.0);
        if (!finite) bits |= GPRY_MASK_CLASSIFIED_INF;
    }
    if (live) mask[m] |= (uint8_t)bits;
}

int launch_gates(gpry_ctx* ctx, const double* Xc, int64_t M, uint8_t* mask) {
    AffParams ap = make_ap(ctx, ctx->tf.has_x_affine);
    const unsigned nb = (unsigned)((M + 255) / 256);
#define GT(DP) hipLaunchKernelGGL((gates_kernel<DP>), dim3(nb), dim3(256), 0, ctx->stream, Xc, M, ctx->d, ctx->gate_sv, \
                                  ctx->gate_coef, ctx->gate_nsv, ctx->gate_gamma, ctx->gate_intercept,                \
                                  ctx->gate_positive_finite, ctx->gate_trust, ctx->gate_has_trust,                   \
                                  ctx->tf.has_x_affine, ap, mask)
    if (ctx->d <= 4) GT(4); else if (ctx->d <= 8) GT(8); else if (ctx->d <= 16) GT(16); else GT(32);
#undef GT
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}
