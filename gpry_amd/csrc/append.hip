// gpry_append_rows: grow the factor by k training points at fixed hyper-parameters and frozen
// pre-processors -- what append_to_data(fit_gpr=False, fit_classifier=False) needs (gpry/gpr.py:577-753
// -> _update_model :996-1020, which REBUILDS K and refactorises, O(N^3); used for the "lies" of
// BatchOptimizer, gpry/gp_acquisition.py:488-491, and by RankedPool.cache_model :1550-1553).  With
//     K' = [[K, B], [B^T, C]],   U = L^-1 B = V B,   S = C - U^T U = L22 L22^T
//     L' = [[L, 0], [U^T, L22]],                V' = L'^-1 = [[V, 0], [-L22^-1 U^T V, L22^-1]]
// the update is two N x N x k products on the MFMA engine plus O(k^3) in one workgroup: O(k N^2).
#include "common.h"
#include <algorithm>

// dst (ld ldn, n_new x n_new) <- src (ld ldo, n_old x n_old) extended by an identity block
__global__ void relayout_kernel(const double* __restrict__ src, int64_t ldo, int64_t n_old, double* __restrict__ dst,
                                int64_t ldn, int64_t n_new) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_new * n_new) return;
    const int64_t i = idx / n_new, j = idx - i * n_new;
    dst[i * ldn + j] = (i < n_old && j < n_old) ? src[i * ldo + j] : (i == j ? 1.0 : 0.0);
}

// Make room for N_new training rows WITHOUT losing the factor (ensure_capacity drops everything).
static int grow_keep_factor(gpry_ctx* ctx, int64_t N_new) {
    const int64_t Np_old = ctx->Np, Np_new = round_up(N_new, GPRY_TILE);
    if (Np_new == Np_old) return 0;
    hipStream_t st = ctx->stream;
    const unsigned nb = (unsigned)((Np_new * Np_new + 255) / 256);
    if (Np_new <= ctx->cap) {
        // same buffers, wider leading dimension: through the LML scratch matrices
        hipLaunchKernelGGL(relayout_kernel, dim3(nb), dim3(256), 0, st, ctx->dA, Np_old, Np_old, ctx->dW, Np_new, Np_new);
        hipLaunchKernelGGL(relayout_kernel, dim3(nb), dim3(256), 0, st, ctx->dV, Np_old, Np_old, ctx->dW2, Np_new, Np_new);
        HIP_TRY(ctx, hipGetLastError());
        std::swap(ctx->dA, ctx->dW);
        std::swap(ctx->dV, ctx->dW2);
        // padding entries beyond the old padded size were never initialised
        HIP_TRY(ctx, hipMemsetAsync(ctx->dy + Np_old, 0, sizeof(double) * (Np_new - Np_old), st));
        HIP_TRY(ctx, hipMemsetAsync(ctx->dnoise + Np_old, 0, sizeof(double) * (Np_new - Np_old), st));
    } else {
        int64_t cap = std::max(Np_new, round_up(ctx->cap + ctx->cap / 8, GPRY_TILE));
        const int dp = ctx->dp_cap;
        double *nX = nullptr, *nXs = nullptr, *ny = nullptr, *nn = nullptr, *nA = nullptr, *nV = nullptr, *nW = nullptr,
               *nW2 = nullptr, *nW3 = nullptr, *na = nullptr, *nv = nullptr;
        double** fresh[] = {&nX, &nXs, &ny, &nn, &nA, &nV, &nW, &nW2, &nW3, &na, &nv};
        const int64_t counts[] = {cap * dp, cap * dp, cap, cap, cap * cap, cap * cap, cap * cap, cap * cap, cap * cap, cap,
                                  8 * cap + 4096};
        for (size_t i = 0; i < sizeof(fresh) / sizeof(fresh[0]); i++) {
            int rc = dev_alloc(ctx, fresh[i], counts[i]);
            if (rc) { for (auto f : fresh) if (*f) (void)hipFree(*f); return rc; }      // the old state is intact
        }
        HIP_TRY(ctx, hipMemcpyAsync(nX, ctx->dX, sizeof(double) * ctx->N * ctx->d, hipMemcpyDeviceToDevice, st));
        HIP_TRY(ctx, hipMemsetAsync(ny, 0, sizeof(double) * cap, st));
        HIP_TRY(ctx, hipMemsetAsync(nn, 0, sizeof(double) * cap, st));
        HIP_TRY(ctx, hipMemcpyAsync(ny, ctx->dy, sizeof(double) * ctx->N, hipMemcpyDeviceToDevice, st));
        HIP_TRY(ctx, hipMemcpyAsync(nn, ctx->dnoise, sizeof(double) * ctx->N, hipMemcpyDeviceToDevice, st));
        hipLaunchKernelGGL(relayout_kernel, dim3(nb), dim3(256), 0, st, ctx->dA, Np_old, Np_old, nA, Np_new, Np_new);
        hipLaunchKernelGGL(relayout_kernel, dim3(nb), dim3(256), 0, st, ctx->dV, Np_old, Np_old, nV, Np_new, Np_new);
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(st));
        double* old[] = {ctx->dX, ctx->dXs, ctx->dy, ctx->dnoise, ctx->dA, ctx->dV, ctx->dW, ctx->dW2, ctx->dW3,
                         ctx->dalpha_, ctx->dvec};
        for (double* o : old) if (o) (void)hipFree(o);
        ctx->dX = nX; ctx->dXs = nXs; ctx->dy = ny; ctx->dnoise = nn; ctx->dA = nA; ctx->dV = nV; ctx->dW = nW;
        ctx->dW2 = nW2; ctx->dW3 = nW3; ctx->dalpha_ = na; ctx->dvec = nv;
        ctx->cap = cap;
        ctx->kst_cap = 0; if (ctx->dKst) { (void)hipFree(ctx->dKst); ctx->dKst = nullptr; }
        ctx->kb_cap = 0; ctx->kb_ld = 0;
        if (ctx->dU) { (void)hipFree(ctx->dU); ctx->dU = nullptr; }
        if (ctx->dXkb) { (void)hipFree(ctx->dXkb); ctx->dXkb = nullptr; }
        ctx->bord_cap = 0; if (ctx->dbord) { (void)hipFree(ctx->dbord); ctx->dbord = nullptr; }
    }
    ctx->Np = Np_new;
    ctx->lml_cache = false;       // dW / dW2 were used as scratch (or replaced)
    ctx->kb_n = 0;                // u(x) vectors of a KB session have the old length
    return 0;
}

// S = C - U^T U for the k x k border block (lower triangle): one workgroup per entry, reduction over the rows
__global__ __launch_bounds__(256) void border_s_kernel(const double* __restrict__ U, int64_t ldu, int64_t nrow, int k,
                                                       const double* __restrict__ Cb, double* __restrict__ S) {
    const int a = blockIdx.x, b = blockIdx.y;
    if (b > a) return;
    __shared__ double red[4];
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < nrow; j += 256) acc = fma(U[j * ldu + a], U[j * ldu + b], acc);
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) S[a * 64 + b] = Cb[a * 64 + b] - (((red[0] + red[1]) + red[2]) + red[3]);
}

// L22 = chol(S) (dpotf2 order, reciprocal-pivot scaling), W22 = L22^-1 (column by column, true divisions),
// both written into the factor: L[r0+a][r0+b], V[r0+a][r0+b] (strict upper part of the block zeroed).  One
// workgroup; k <= 64.  info: first failing column (global, 1-based) if S is not positive definite.
__global__ __launch_bounds__(256) void border_chol_kernel(const double* __restrict__ S, int k, int64_t r0, double* __restrict__ L,
                                                          double* __restrict__ V, int64_t ld, double* __restrict__ W22,
                                                          int* info) {
    __shared__ double sL[64 * 65];
    __shared__ double sW[64 * 65];
    __shared__ int s_bad;
    const int t = threadIdx.x;
    if (t == 0) s_bad = 0;
    for (int e = t; e < 64 * 64; e += 256) {
        const int a = e >> 6, b = e & 63;
        sL[a * 65 + b] = (a < k && b <= a) ? S[a * 64 + b] : 0.0;
        sW[a * 65 + b] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < k; j++) {
        const double djj = sL[j * 65 + j];
        if (!(djj > 0.0)) { if (t == 0 && s_bad == 0) s_bad = j + 1; }
        __syncthreads();
        if (s_bad) break;
        const double piv = sqrt(djj), rinv = 1.0 / piv;
        if (t > j && t < k) sL[t * 65 + j] *= rinv;
        if (t == j) sL[j * 65 + j] = piv;
        __syncthreads();
        // trailing update of the lower triangle: entries (a, b), j < b <= a < k
        for (int e = t; e < k * k; e += 256) {
            const int a = e / k, b = e - a * k;
            if (b > j && a >= b) sL[a * 65 + b] = fma(-sL[a * 65 + j], sL[b * 65 + j], sL[a * 65 + b]);
        }
        __syncthreads();
    }
    if (s_bad) { if (t == 0) atomicCAS(info, 0, (int)(r0 + s_bad)); return; }
    if (t < k) {            // column t of L22^-1 by forward substitution
        for (int i = t; i < k; i++) {
            double acc = (i == t) ? 1.0 : 0.0;
            for (int c = t; c < i; c++) acc = fma(-sL[i * 65 + c], sW[c * 65 + t], acc);
            sW[i * 65 + t] = acc / sL[i * 65 + i];
        }
    }
    __syncthreads();
    for (int e = t; e < k * k; e += 256) {
        const int a = e / k, b = e - a * k;
        L[(r0 + a) * ld + r0 + b] = b <= a ? sL[a * 65 + b] : 0.0;
        V[(r0 + a) * ld + r0 + b] = b <= a ? sW[a * 65 + b] : 0.0;
        W22[a * 64 + b] = b <= a ? sW[a * 65 + b] : 0.0;
    }
}

// L[r0+a][j] = U[j][a],   V[r0+a][j] = -sum_b W22[a][b] T[b][j]      (j < r0)
__global__ __launch_bounds__(256) void border_rows_kernel(const double* __restrict__ U, int64_t ldu, const double* __restrict__ T,
                                                          int64_t ldt, const double* __restrict__ W22, int k, int64_t r0,
                                                          double* __restrict__ L, double* __restrict__ V, int64_t ld,
                                                          const int* info) {
    if (*info != 0) return;
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int a = blockIdx.y;
    if (j >= r0) return;
    L[(r0 + a) * ld + j] = U[j * ldu + a];
    double acc = 0.0;
    for (int b = 0; b <= a; b++) acc = fma(W22[a * 64 + b], T[(int64_t)b * ldt + j], acc);
    V[(r0 + a) * ld + j] = -acc;
}

static int append_chunk(gpry_ctx* ctx, const double* Xn, const double* yn, const double* an, int k, int* info_host) {
    hipStream_t st = ctx->stream;
    const int64_t N0 = ctx->N;
    GPRY_TRY(grow_keep_factor(ctx, N0 + k));
    const int64_t Np = ctx->Np;
    const int64_t W = 128;                                     // padded border width (one GEMM tile)
    const int64_t need = 2 * Np * W + W * Np + 3 * 64 * 64;
    if (need > ctx->bord_cap) {
        if (ctx->dbord) GPRY_TRY(dev_free(ctx, ctx->dbord));
        ctx->dbord = nullptr; ctx->bord_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dbord, need));
        ctx->bord_cap = need;
    }
    double* Bk = ctx->dbord;                 // Np x W   B[j][a] = K[N0+a][j]
    double* Um = Bk + Np * W;                // Np x W   U = V B
    double* Tm = Um + Np * W;                // W x Np   T = U^T V
    double* Cb = Tm + W * Np;                // 64 x 64  border block of K, then S
    double* Sm = Cb + 64 * 64;
    double* W22 = Sm + 64 * 64;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dX + N0 * ctx->d, Xn, sizeof(double) * k * ctx->d, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dy + N0, yn, sizeof(double) * k, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dnoise + N0, an, sizeof(double) * k, hipMemcpyHostToDevice, st));
    ctx->N = N0 + k;
    for (int a = 0; a < k; a++) if (an[a] < ctx->noise_min) ctx->noise_min = an[a];
    // centre of the MFMA panel build: the running sums continue in row order, so that a context that grew by border rows
    // holds the centre -- to the bit -- of one that received the whole set at once (members of a device group do)
    for (int a = 0; a < k; a++)
        for (int c = 0; c < ctx->d; c++) {
            const double v = Xn[(int64_t)a * ctx->d + c];
            ctx->xsum[c] += v;
            if (v < ctx->xlo[c]) ctx->xlo[c] = v;
            if (v > ctx->xhi[c]) ctx->xhi[c] = v;
        }
    for (int c = 0; c < ctx->d; c++) ctx->xcenter[c] = ctx->xsum[c] / (double)ctx->N;
    GPRY_TRY(launch_scale_train(ctx));
    GPRY_TRY(launch_kernel_rows(ctx, N0, k, W, Bk, Cb));
    auto splits = [&](int64_t tiles) { int n = 1; while (n < 16 && tiles * n * 2 <= 512 && Np / (n * 2) >= 64) n *= 2; return n; };
    {   // U = V B   (V lower triangular; its rows >= N0 are still identity rows, B is zero there)
        GemmArgs g = {};
        g.A = ctx->dV; g.lda = Np; g.B = Bk; g.ldb = W; g.C = Um; g.ldc = W;
        g.M = (int)Np; g.N = (int)W; g.K = (int)Np; g.kmode = KM_A_LOWER; g.tile_map = TM_ROWMAJOR;
        g.nsplit = splits(Np / 128);
        if (g.nsplit > 1) { GPRY_TRY(gemm_split_scratch(ctx, g.nsplit, Np * W, &g.split_buf)); g.split_stride = Np * W; }
        GPRY_TRY(gemm_f64_launch(ctx, g, false, false, EPI_STORE));
    }
    hipLaunchKernelGGL(border_s_kernel, dim3((unsigned)k, (unsigned)k), dim3(256), 0, st, Um, W, N0, k, Cb, Sm);
    hipLaunchKernelGGL(border_chol_kernel, dim3(1), dim3(256), 0, st, Sm, k, N0, ctx->dA, ctx->dV, Np, W22, ctx->dinfo);
    HIP_TRY(ctx, hipGetLastError());
    {   // T = U^T V   (V lower: k-range of column tile tj starts at tj * 128)
        GemmArgs g = {};
        g.A = Um; g.lda = W; g.B = ctx->dV; g.ldb = Np; g.C = Tm; g.ldc = Np;
        g.M = (int)W; g.N = (int)Np; g.K = (int)Np; g.kmode = KM_B_LOWER; g.tile_map = TM_ROWMAJOR;
        g.nsplit = splits(Np / 128);
        if (g.nsplit > 1) { GPRY_TRY(gemm_split_scratch(ctx, g.nsplit, W * Np, &g.split_buf)); g.split_stride = W * Np; }
        GPRY_TRY(gemm_f64_launch(ctx, g, true, false, EPI_STORE));
    }
    hipLaunchKernelGGL(border_rows_kernel, dim3((unsigned)((N0 + 255) / 256), (unsigned)k), dim3(256), 0, st, Um, W, Tm, Np,
                       W22, k, N0, ctx->dA, ctx->dV, Np, ctx->dinfo);
    HIP_TRY(ctx, hipGetLastError());
    int info[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(info, ctx->dinfo, 2 * sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    *info_host = info[0];
    return 0;
}

extern "C" int gpry_append_rows(gpry_ctx* ctx, const double* Xnew_, const double* ynew_, const double* alphanew,
                                int64_t k, int* info) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_append_rows: ctx is NULL");
    if (info) *info = 0;
    GPRY_TRY(serve_stop(ctx));
    if (ctx->N <= 0 || !ctx->have_theta || !ctx->factor_valid)
        return gpry_fail(ctx, -1, "append_rows: no factorised model to extend (call gpry_factorize)");
    if (k <= 0) return 0;
    if (!Xnew_ || !ynew_ || !alphanew) return gpry_fail(ctx, -1, "append_rows: Xnew_, ynew_ and alphanew must not be NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    StageScope scope(ctx, "append_rows");
    HIP_TRY(ctx, hipMemsetAsync(ctx->dinfo, 0, 4 * sizeof(int), ctx->stream));
    ctx->lml_cache = false;
    ctx->kb_n = 0;
    for (int64_t o = 0; o < k; o += 64) {
        const int kc = (int)std::min<int64_t>(64, k - o);
        int inf = 0;
        int rc = append_chunk(ctx, Xnew_ + o * ctx->d, ynew_ + o, alphanew + o, kc, &inf);
        if (rc) { ctx->factor_valid = false; return rc; }
        if (inf != 0) {
            // not positive definite: the factor is left incomplete; the caller refactorises the
            // enlarged training set (gpry_set_train + gpry_factorize), which reports the reference's error
            ctx->factor_valid = false;
            if (info) *info = inf;
            return 0;
        }
        ctx->n_border += kc;
    }
    // alpha_ = V^T (V y) over the enlarged set
    GPRY_TRY(solve_alpha(ctx, ctx->dV, ctx->dy, ctx->dvec, ctx->dalpha_, ctx->Np));
    ctx->alpha_l2 = -1.0;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}
