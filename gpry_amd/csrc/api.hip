// C-ABI entry points of the GP hot path (include/gpry_hip.h): factor, LML, predict,
// fused NORA sweep, shortlist selection and Kriging-believer support.
#include "common.h"
#include <algorithm>

static int require_model(gpry_ctx* ctx, bool need_factor) {
    if (ctx->N <= 0) return gpry_fail(ctx, -1, "no training set (call gpry_set_train)");
    if (!ctx->have_theta) return gpry_fail(ctx, -1, "no hyperparameters (call gpry_set_theta)");
    if (need_factor && !ctx->factor_valid) return gpry_fail(ctx, -1, "model not factorised (call gpry_factorize)");
    // an X map that was set for a model of fewer dimensions has zero spans in the new ones: every prediction would be NaN
    if (need_factor && ctx->tf.has_x_affine)
        for (int k = 0; k < ctx->d; k++)
            if (!(ctx->tf.x_span[k] != 0.0) || !isfinite(ctx->tf.x_span[k]) || !isfinite(ctx->tf.x_lo[k]))
                return gpry_fail(ctx, -1, "affine map of X: span %g, offset %g in dimension %d of %d (gpry_set_affine of another model?)",
                                 ctx->tf.x_span[k], ctx->tf.x_lo[k], k, ctx->d);
    return 0;
}

// temporary device buffer of an entry point: freed on EVERY return path (the error paths of the entry points
// below used to leak their scratch allocations)
template <typename T>
struct TmpBuf {
    T* p = nullptr;
    TmpBuf() = default;
    TmpBuf(const TmpBuf&) = delete;
    TmpBuf& operator=(const TmpBuf&) = delete;
    ~TmpBuf() { if (p) (void)hipFree(p); }
    int alloc(gpry_ctx* ctx, int64_t count) { return dev_alloc(ctx, &p, count); }
};

static int ensure_part(gpry_ctx* ctx, int64_t need) {
    if (need <= ctx->part_cap) return 0;
    if (ctx->dpart) GPRY_TRY(dev_free(ctx, ctx->dpart));
    ctx->dpart = nullptr; ctx->part_cap = 0;
    GPRY_TRY(dev_alloc(ctx, &ctx->dpart, need));
    ctx->part_cap = need;
    return 0;
}

// copy a device matrix with leading dimension ld to a dense host rows x cols array
static int copy_out_matrix(gpry_ctx* ctx, const double* dsrc, int64_t ld, int64_t rows, int64_t cols, double* hdst) {
    HIP_TRY(ctx, hipMemcpy2DAsync(hdst, sizeof(double) * cols, dsrc, sizeof(double) * ld,
                                  sizeof(double) * cols, rows, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// A panel step whose bounded waits ran out (chol_panel.hip: PANEL_SPIN_CAP) reports a failing column like a matrix that is
// not positive definite and leaves a marker in the fourth status word.  That is a scheduling failure, not a property of the
// matrix: the caller gets an error, never the -inf of sklearn's non-PD convention (an optimiser would go on with a wrong objective).
static bool panel_timed_out(int marker) { return (marker & 0xFF00) == 0x5A00; }
static int panel_timeout_error(gpry_ctx* ctx, int marker) {
    return gpry_fail(ctx, -2, "Cholesky panel step timed out (wait %d; the GPU is shared with work that starves the step?)", marker & 0xFF);
}
static int build_factor(gpry_ctx* ctx, double* A, double* V, double* T, int* info_host) {
    // A <- K + diag(alpha); A <- chol(A) (lower); V <- A^-1
    GPRY_TRY(launch_scale_train(ctx));      // X / l (N x d, microseconds)
    // (where V comes out of the Cholesky launches -- potrf_stacked below -- T starts as the identity: written by the same kernel)
    const bool want_stacked = ctx->opt_chol != 1 && potrf_stacked_usable(ctx, ctx->Np);
    {
        StageScope s(ctx, "kernel_build");   // the O(N^2 d) covariance build proper
        GPRY_TRY(launch_kernel_train(ctx, A, 1, want_stacked ? T : nullptr));
    }
    if (ctx->opt_chol == 1) {
        StageScope s(ctx, "potrf");
        GPRY_TRY(rocsolver_potrf_trtri(ctx, A, V, ctx->Np, 1));
    } else {
        // V = L^-1 phase by phase on stream2 while the panel chain is still running (chol.hip); "trtri" then
        // times what is left of it after potrf
        bool piped = false;
        struct PipeGuard {      // an error between begin and finish must not leave side-stream work in flight
            gpry_ctx* c; bool armed = false;
            ~PipeGuard() { if (armed) trtri_pipeline_abort(c); }
        } guard{ctx};
        // (a batched evaluation has thetas enough to fill the GPU: its V = L^-1 stays behind potrf on the main stream;
        // the two schedules give the same bits)
        if (ctx->opt_factor_pipeline && ctx->opt_chol == 0 && ctx->Np >= ctx->opt_factor_pipeline_min && ctx->bn == 1 && !ctx->tp &&
            !potrf_stacked_usable(ctx, ctx->Np)) {
            const int rc = trtri_pipeline_begin(ctx, A, V, T, ctx->Np);
            if (rc < 0) return rc;
            piped = rc == 0;
            guard.armed = piped;
        }
        // Up to Np = "chol_stacked" the inverse factor comes out of the Cholesky launches themselves (potrf_stacked, chol_panel.hip):
        // T takes the identity and comes back as L^-T, which is transposed into V.
        bool stacked = !piped && want_stacked;
        if (stacked) {
            StageScope s(ctx, "potrf");
            const int rc = potrf_stacked(ctx, A, T, ctx->Np);
            if (rc < 0) return rc;
            stacked = rc == 0;      // (1: no plan for this size)
        }
        if (stacked) {
            StageScope s(ctx, "trtri");
            GPRY_TRY(transpose_upper_launch(ctx, T, V, ctx->Np));
        } else {
            {
                StageScope s(ctx, "potrf");
                GPRY_TRY(ctx->opt_chol_overlap ? potrf_lower_overlap(ctx, A, ctx->Np) : potrf_lower_fused(ctx, A, ctx->Np));
            }
            {
                StageScope s(ctx, "trtri");
                if (piped) { GPRY_TRY(trtri_pipeline_finish(ctx)); guard.armed = false; }
                else GPRY_TRY(trtri_lower(ctx, A, V, T, ctx->Np));
            }
        }
    }
    ctx->info_cleared = false;
    if (!info_host) return 0;      // the caller fetches dinfo itself, together with its results
    int info[4] = {0, 0, 0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(info, ctx->dinfo, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (panel_timed_out(info[3])) return panel_timeout_error(ctx, info[3]);
    *info_host = info[0] != 0 ? info[0] : info[1];
    return 0;
}

// The launches of one objective evaluation (sklearn:_gpr.py:574-652) at the theta the context holds -- or, in a batched
// evaluation (gpry_ctx::bn), at the bn thetas whose [C, l...] rows sit in the arena: covariance build, factor, V = L^-1,
// alpha, log-det and quadratic form, and with want_grad K^-1 = V^T V and the traces.  The last kernel writes results and
// factorisation status into `dres` (device view of pinned host memory; theta tb at row tb of GPRY_BRES_STRIDE doubles).
constexpr int RES_INFO = 2 + 1 + GPRY_MAX_DIM;      // [logdet/2, quad, grad (1 + d) ..., info0, info1]
static_assert(RES_INFO + 3 <= GPRY_BRES_STRIDE, "result row too short");
static int lml_chain(gpry_ctx* ctx, int want_grad, double* dres) {
    GPRY_TRY(build_factor(ctx, ctx->dW, ctx->dW2, ctx->dW3, nullptr));
    double* dz = ctx->dvec;                 // z = V y
    double* da = ctx->dvec + ctx->Np;       // alpha
    double* dout = ctx->dvec + 2 * ctx->Np; // device copy: [logdet/2, quad, grad...]
    {
        StageScope s(ctx, "solve_alpha");
        GPRY_TRY(solve_alpha(ctx, ctx->dW2, ctx->dy, dz, da, ctx->Np));
        GPRY_TRY(logdet_and_quad(ctx, ctx->dW, dz, ctx->Np, dout, want_grad ? nullptr : dres, RES_INFO));
    }
    if (!want_grad) return 0;
    {
        StageScope s(ctx, "lauum");
        GPRY_TRY(lauum_lower(ctx, ctx->dW2, ctx->dW3, ctx->Np));
    }
    StageScope s(ctx, "lml_traces");
    return launch_lml_traces(ctx, ctx->dW3, da, dout + 2, dout, dres, RES_INFO);
}

// ---- B thetas in ONE chain of launches (128 < Np <= lml_batch) --------------------------------------------------
// The reference runs the optimiser restarts of a fit one after another (gpry/gpr.py:968-984, 10 + 2 d of them by default,
// gpry/run.py:315-325); stepped side by side (gpry_amd/lockstep.py) they hand a round's thetas to gpry_lml_batch.  Above
// N = 128 one evaluation is a chain of 20-60 launches that keeps a handful of CUs busy (0.001-0.02 of the FP64 peak at
// N = 256 ... 1024); here every launch of that chain carries all thetas (grid.z).  Same kernels, same launch geometry,
// same operands per theta as a single gpry_lml: the per-theta results are bit-identical to it.
struct BatchLayout {
    int64_t w, w2, w3, xs, vec, part, split, par, info;     // offsets (doubles) within a theta's set
    int64_t part_cap, split_cap, stride;
};
static int batch_layout(gpry_ctx* ctx, BatchLayout* L) {
    const int64_t Np = ctx->Np, nn = Np * Np;
    int slices = 0;
    GPRY_TRY(factor_chain_slices(ctx, Np, &slices));
    const int DPsel = ctx->d <= 4 ? 4 : ctx->d <= 8 ? 8 : ctx->d <= 16 ? 16 : ctx->d <= 24 ? 24 : 32;
    const int64_t nb = Np / 64, ntile = nb * (nb + 1) / 2;
    const int64_t p1 = ((Np + 255) / 256) * Np, p2 = ntile * (DPsel + 1);       // solve_alpha, launch_lml_traces
    int64_t o = 0;
    auto take = [&](int64_t n) { const int64_t at = o; o += round_up(n, 32); return at; };      // 256-byte pieces
    L->w = take(nn); L->w2 = take(nn); L->w3 = take(nn);
    L->xs = take(Np * ctx->dpad);
    L->vec = take(8 * Np + 4096);
    L->part_cap = p1 > p2 ? p1 : p2; L->part = take(L->part_cap);
    L->split_cap = (int64_t)slices * nn; L->split = take(L->split_cap > 0 ? L->split_cap : 1);
    L->par = take(1 + GPRY_MAX_DIM);
    L->info = take(8);                      // 16 status / arrival words
    L->stride = o + 32 * 9;                 // not a multiple of a large power of two: consecutive sets start on different channels
    return 0;
}
// returns 2 when the device has no room for the arena (the caller halves its chunk), < 0 on any other failure
static int ensure_batch_buffers(gpry_ctx* ctx, int64_t arena_doubles, int64_t res_bytes) {
    if (arena_doubles > ctx->barena_cap) {
        if (ctx->barena) GPRY_TRY(dev_free(ctx, ctx->barena));
        ctx->barena = nullptr; ctx->barena_cap = 0;
        const hipError_t e = hipMalloc((void**)&ctx->barena, sizeof(double) * (size_t)arena_doubles);
        if (e == hipErrorOutOfMemory) { (void)hipGetLastError(); ctx->barena = nullptr; return 2; }
        if (e != hipSuccess) { ctx->barena = nullptr; return gpry_fail(ctx, -2, "lml_batch: hipMalloc of the scratch arena failed: %s", hipGetErrorString(e)); }
        ctx->barena_cap = arena_doubles;
    }
    if (res_bytes > ctx->hbres_cap) {
        if (ctx->hbres) HIP_TRY(ctx, hipHostFree(ctx->hbres));
        ctx->hbres = nullptr; ctx->hbres_dev = nullptr; ctx->hbres_cap = 0;
        HIP_TRY(ctx, hipHostMalloc(&ctx->hbres, (size_t)res_bytes, hipHostMallocMapped | hipHostMallocPortable));
        HIP_TRY(ctx, hipHostGetDevicePointer(&ctx->hbres_dev, ctx->hbres, 0));
        memset(ctx->hbres, 0, (size_t)res_bytes);
        ctx->hbres_cap = res_bytes;
    }
    return 0;
}
// is the batched chain built for this context's size and options?  (rocSOLVER is not batched: one evaluation after another)
static bool lml_batch_usable(const gpry_ctx* ctx) {
    return ctx->N > 0 && ctx->Np > 128 && ctx->Np <= ctx->opt_lml_batch && ctx->d <= 32 && ctx->opt_chol == 0;
}
static int lml_batch_general(gpry_ctx* ctx, const double* thetas, int64_t B, int want_grad, double* lml, double* grad, int* info) {
    const int w = ctx->d + 1;
    for (int64_t i = 0; i < B * w; i++)
        if (!isfinite(thetas[i])) return gpry_fail(ctx, -1, "theta[%d] is not finite", (int)(i % w));
    GPRY_TRY(serve_stop(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // the chain below is the code of a single evaluation: it finds the buffers of theta 0 where the context keeps its own
    struct Scope {
        gpry_ctx* c;
        double *dW, *dW2, *dW3, *dXs, *dvec, *dpart, *dsplit; int* dinfo; hipStream_t stream;
        int64_t part_cap, split_cap; bool xs_foreign, have_theta;
        double theta[1 + GPRY_MAX_DIM];
        explicit Scope(gpry_ctx* ctx) : c(ctx), dW(ctx->dW), dW2(ctx->dW2), dW3(ctx->dW3), dXs(ctx->dXs), dvec(ctx->dvec), dpart(ctx->dpart),
                                        dsplit(ctx->dsplit), dinfo(ctx->dinfo), stream(ctx->stream), part_cap(ctx->part_cap), split_cap(ctx->split_cap),
                                        xs_foreign(ctx->xs_foreign), have_theta(ctx->have_theta) {
            memcpy(theta, ctx->theta, sizeof(theta));
        }
        ~Scope() {
            c->dW = dW; c->dW2 = dW2; c->dW3 = dW3; c->dXs = dXs; c->dvec = dvec; c->dpart = dpart; c->dsplit = dsplit; c->dinfo = dinfo;
            c->stream = stream;
            c->part_cap = part_cap; c->split_cap = split_cap; c->xs_foreign = xs_foreign; c->have_theta = have_theta;
            c->info_cleared = false;
            memcpy(c->theta, theta, sizeof(theta));
            c->bn = 1; c->bstride = 0; c->bpar = nullptr; c->tp = false;
        }
    } scope(ctx);
    // THE THROUGHPUT SCHEDULE (option "lml_schedule" = 1; gpry_ctx::opt_lml_schedule): whole-tile products only, the recursive
    // inverse at every size, the Cholesky in column blocks, the thetas of a chunk dealt over stream groups.  Everything a theta's
    // result depends on is a function of the model size alone -- never of B, of the chunking or of the grouping.
    const bool tp = ctx->opt_lml_schedule == 1;
    ctx->tp = tp;
    BatchLayout L;
    GPRY_TRY(batch_layout(ctx, &L));
    // thetas per chain: the diagonal workgroup of a panel step waits for the other workgroups of ITS theta to have read the
    // block it overwrites -- only those wait, one per theta, so any number well below the GPU's workgroup slots is safe
    int64_t chunk = B < 96 ? B : 96;
    int64_t mem_cap = (ctx->opt_lml_batch_mb << 20) / (8 * L.stride);
    {
        // ... and by what the device has free right now: a fit opens up to three such contexts per GPU beside the farm's,
        // and "lml_batch_mb" is sized for an empty 288-GB part.  The arena may take what it already holds plus 70 % of
        // the free memory.
        size_t free_b = 0, total_b = 0;
        if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
            const int64_t room = ctx->barena_cap * 8 + (int64_t)((double)free_b * 0.7);
            if (room / (8 * L.stride) < mem_cap) mem_cap = room / (8 * L.stride);
        }
    }
    if (chunk > mem_cap) chunk = mem_cap;
    // (latency schedule: below two sets the caller evaluates the thetas one after another -- the same bits; the throughput
    // schedule has no such twin: one set at a time is still its own chain)
    const int64_t min_chunk = tp ? 1 : 2;
    if (chunk < min_chunk) {
        if (!tp) return 1;
        return gpry_fail(ctx, -2, "lml_batch: no device memory for one scratch set (%lld MiB)", (long long)((8 * L.stride) >> 20));
    }
    // a failed allocation (another context took the memory in between) halves the chunk; below `min_chunk` sets the thetas go
    // through one after another instead of the fit failing (latency schedule).  Only an out-of-memory answer is retried.
    for (;;) {
        const int rc = ensure_batch_buffers(ctx, chunk * L.stride, (int64_t)sizeof(double) * chunk * (GPRY_BRES_STRIDE + 1 + GPRY_MAX_DIM));
        if (rc == 0) break;
        if (rc != 2) return rc;             // anything but "out of memory" is the caller's to see
        ctx->batch_shrinks++;
        chunk /= 2;
        if (chunk < min_chunk) {
            if (!tp) return 1;
            return gpry_fail(ctx, -2, "lml_batch: no device memory for one scratch set");
        }
    }
    double* hres = static_cast<double*>(ctx->hbres);
    double* hpar = hres + chunk * GPRY_BRES_STRIDE;         // [C, l_1 .. l_d] rows of a chunk, staged in the pinned buffer
    double* a0 = ctx->barena;
    const hipStream_t main_stream = scope.stream;
    // the sets of thetas [t0, t0 + n) become "the context's own buffers" for the launchers of the chain
    auto point_at = [&](int64_t t0, int64_t n, hipStream_t st) {
        double* a = a0 + t0 * L.stride;
        ctx->dW = a + L.w; ctx->dW2 = a + L.w2; ctx->dW3 = a + L.w3; ctx->dXs = a + L.xs; ctx->dvec = a + L.vec;
        ctx->dpart = a + L.part; ctx->part_cap = L.part_cap; ctx->dsplit = a + L.split; ctx->split_cap = L.split_cap;
        ctx->dinfo = reinterpret_cast<int*>(a + L.info);
        ctx->bstride = L.stride; ctx->bpar = a + L.par;
        ctx->bn = (int)n; ctx->stream = st;
    };
    ctx->have_theta = true;
    const double log2pi = log(2.0 * M_PI);
    for (int64_t b0 = 0; b0 < B; b0 += chunk) {
        const int64_t nb = B - b0 < chunk ? B - b0 : chunk;
        for (int64_t b = 0; b < nb; b++) {
            const double* th = thetas + (b0 + b) * w;
            double* row = hpar + b * (1 + GPRY_MAX_DIM);
            for (int k = 0; k <= GPRY_MAX_DIM; k++) row[k] = 1.0;
            row[0] = exp(th[0]);                                     // as make_kp / make_ap for one evaluation
            for (int k = 0; k < ctx->d; k++) row[1 + k] = exp(th[1 + k]);
            hres[b * GPRY_BRES_STRIDE + RES_INFO] = -1.0; hres[b * GPRY_BRES_STRIDE + RES_INFO + 1] = -1.0;     // "not written"
        }
        for (int k = 0; k < w; k++) ctx->theta[k] = thetas[b0 * w + k];      // (the launchers read the sizes, not these, in a batch)
        // test hook ("panel_debug" bit 7): every scratch set starts as NaNs -- a result that depended on what a set held before shows
        if (ctx->opt_panel_debug & 128) HIP_TRY(ctx, hipMemsetAsync(a0, 0xFF, sizeof(double) * (size_t)(nb * L.stride), main_stream));
        HIP_TRY(ctx, hipMemcpy2DAsync(a0 + L.par, sizeof(double) * L.stride, hpar, sizeof(double) * (1 + GPRY_MAX_DIM),
                                      sizeof(double) * (1 + GPRY_MAX_DIM), (size_t)nb, hipMemcpyHostToDevice, main_stream));
        // Stream groups (throughput schedule): the panel chain of the Cholesky is one workgroup's latency per step whatever the
        // number of thetas, and between two products of a group the GPU drains.  With the thetas dealt over G streams the host
        // queues group after group; the groups run out of step by themselves and one's panel steps sit underneath the others' GEMMs.
        int G = 1;
        if (tp) { G = ctx->opt_lml_streams; if (G > nb) G = (int)nb; if (G < 1) G = 1; }
        if (G > 1) {
            while ((int)ctx->tp_streams.size() < G - 1) {
                hipStream_t st;
                HIP_TRY(ctx, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
                ctx->tp_streams.push_back(st);
            }
            while ((int)ctx->tp_events.size() < G) {
                hipEvent_t ev;
                HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
                ctx->tp_events.push_back(ev);
            }
            HIP_TRY(ctx, hipEventRecord(ctx->tp_events[0], main_stream));        // the parameters are on the device
        }
        {
            StageScope s(ctx, "lml_batch", main_stream);
            int64_t t0 = 0;
            int rc = 0;
            for (int g = 0; g < G && rc == 0; g++) {
                const int64_t n = nb / G + (g < nb % G ? 1 : 0);
                const hipStream_t st = g == 0 ? main_stream : ctx->tp_streams[(size_t)g - 1];
                if (g > 0) {
                    const hipError_t e = hipStreamWaitEvent(st, ctx->tp_events[0], 0);
                    if (e != hipSuccess) { rc = gpry_fail(ctx, -2, "lml_batch: %s", hipGetErrorString(e)); break; }
                }
                point_at(t0, n, st);
                rc = lml_chain(ctx, want_grad, static_cast<double*>(ctx->hbres_dev) + t0 * GPRY_BRES_STRIDE);
                t0 += n;
            }
            // the main stream joins the groups (also on the error path: nothing of this call may still be in flight afterwards)
            for (int g = 1; g < G; g++) {
                const hipStream_t st = ctx->tp_streams[(size_t)g - 1];
                if (hipEventRecord(ctx->tp_events[(size_t)g], st) != hipSuccess || hipStreamWaitEvent(main_stream, ctx->tp_events[(size_t)g], 0) != hipSuccess)
                    (void)hipStreamSynchronize(st);
            }
            ctx->stream = main_stream;
            if (rc) { (void)hipStreamSynchronize(main_stream); return rc; }
        }
        HIP_TRY(ctx, hipStreamSynchronize(main_stream));
        for (int64_t b = 0; b < nb; b++) {
            const double* o = hres + b * GPRY_BRES_STRIDE;
            if (o[RES_INFO] < 0.0) return gpry_fail(ctx, -2, "lml_batch: evaluation %lld did not deliver its status", (long long)(b0 + b));
            const int i0 = (int)o[RES_INFO], i1 = (int)o[RES_INFO + 1];
            if (panel_timed_out((int)o[RES_INFO + 2])) return panel_timeout_error(ctx, (int)o[RES_INFO + 2]);
            const int inf = i0 != 0 ? i0 : i1;
            if (info) info[b0 + b] = inf;
            if (inf != 0) {   // sklearn:_gpr.py:586-589
                lml[b0 + b] = -INFINITY;
                if (want_grad) for (int k = 0; k < w; k++) grad[(b0 + b) * w + k] = 0.0;
                continue;
            }
            lml[b0 + b] = -0.5 * o[1] - o[0] - 0.5 * (double)ctx->N * log2pi;
            if (want_grad) for (int k = 0; k < w; k++) grad[(b0 + b) * w + k] = o[2 + k];
        }
    }
    return 0;
}

__global__ void zero_upper_copy_kernel(const double* __restrict__ src, double* __restrict__ dst, int64_t ld, int64_t n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int64_t i = idx / n, j = idx - i * n;
    dst[idx] = (j <= i) ? src[i * ld + j] : 0.0;
}

extern "C" {

int gpry_kernel_train(gpry_ctx* ctx, int add_alpha, double* K_out) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_kernel_train: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, false));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    GPRY_TRY(launch_scale_train(ctx));
    ctx->lml_cache = false;      // dW is about to be overwritten
    {
        StageScope s(ctx, "kernel_build");
        GPRY_TRY(launch_kernel_train(ctx, ctx->dW, add_alpha));
    }
    if (K_out) GPRY_TRY(copy_out_matrix(ctx, ctx->dW, ctx->Np, ctx->N, ctx->N, K_out));
    else HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int gpry_kernel_cross(gpry_ctx* ctx, const double* Xc_, int64_t M, double* K_out) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_kernel_cross: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, false));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (M <= 0) return 0;
    int64_t mp = round_up(M, 256);
    TmpBuf<double> bX, bK;
    GPRY_TRY(bX.alloc(ctx, M * ctx->d));
    GPRY_TRY(bK.alloc(ctx, ctx->Np * mp));
    double *dX = bX.p, *dK = bK.p;
    HIP_TRY(ctx, hipMemcpyAsync(dX, Xc_, sizeof(double) * M * ctx->d, hipMemcpyHostToDevice, ctx->stream));
    GPRY_TRY(launch_scale_train(ctx));
    int64_t saveM = ctx->sw_M; ctx->sw_M = M;
    int rc = launch_cross_build(ctx, dX, 0, mp, mp, dK, nullptr, 0);
    ctx->sw_M = saveM;
    if (rc) return rc;
    // dK is N x mp (k-major); the caller wants M x N: transpose on the host side copy
    std::vector<double> tmp((size_t)ctx->N * M);
    GPRY_TRY(copy_out_matrix(ctx, dK, mp, ctx->N, M, tmp.data()));
    for (int64_t j = 0; j < ctx->N; j++)
        for (int64_t m = 0; m < M; m++) K_out[m * ctx->N + j] = tmp[(size_t)j * M + m];
    return 0;
}

int gpry_factorize(gpry_ctx* ctx, int* info) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_factorize: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, false));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    ctx->factor_valid = false;
    ctx->kb_n = 0;
    int inf = 0;
    // The optimiser's last objective evaluation is normally at the theta it returns: its factor
    // (L in dW, V in dW2; same kernels, same inputs => the bits a fresh factorisation would give)
    // is adopted by swapping buffers instead of factorising again.
    bool hit = ctx->opt_lml_cache && ctx->lml_cache && ctx->lml_kernel_id == ctx->kernel_id;
    for (int k = 0; hit && k <= ctx->d; k++) hit = ctx->lml_theta[k] == ctx->theta[k];
    ctx->lml_cache = false;
    if (hit) {
        std::swap(ctx->dA, ctx->dW);
        std::swap(ctx->dV, ctx->dW2);
        GPRY_TRY(launch_scale_train(ctx));
    } else {
        GPRY_TRY(build_factor(ctx, ctx->dA, ctx->dV, ctx->dW, &inf));
    }
    if (info) *info = inf;
    if (inf != 0) return 0;
    // alpha_ = V^T (V y)
    GPRY_TRY(solve_alpha(ctx, ctx->dV, ctx->dy, ctx->dvec, ctx->dalpha_, ctx->Np));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->factor_valid = true;
    ctx->alpha_l2 = -1.0;
    return 0;
}

int gpry_get_factor(gpry_ctx* ctx, double* L, double* V, double* alpha_) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_get_factor: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    int64_t N = ctx->N;
    if (L) {
        TmpBuf<double> bt;
        GPRY_TRY(bt.alloc(ctx, N * N));
        double* tmp = bt.p;
        hipLaunchKernelGGL(zero_upper_copy_kernel, dim3((unsigned)((N * N + 255) / 256)), dim3(256), 0,
                           ctx->stream, ctx->dA, tmp, ctx->Np, N);
        HIP_TRY(ctx, hipMemcpyAsync(L, tmp, sizeof(double) * N * N, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    if (V) GPRY_TRY(copy_out_matrix(ctx, ctx->dV, ctx->Np, N, N, V));
    if (alpha_) {
        HIP_TRY(ctx, hipMemcpyAsync(alpha_, ctx->dalpha_, sizeof(double) * N, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return 0;
}

int gpry_lml(gpry_ctx* ctx, const double* theta, int want_grad, double* lml, double* grad, int* info) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_lml: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    if (ctx->N <= 0) return gpry_fail(ctx, -1, "no training set (call gpry_set_train)");
    if (!theta || !lml || (want_grad && !grad)) return gpry_fail(ctx, -1, "lml: theta, lml and (with want_grad) grad must not be NULL");
    for (int k = 0; k <= ctx->d; k++)
        if (!isfinite(theta[k])) return gpry_fail(ctx, -1, "theta[%d] is not finite", k);
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    // evaluate at `theta` without disturbing the prediction factor (dA, dV, dalpha_)
    double saved[1 + GPRY_MAX_DIM];
    bool had = ctx->have_theta;
    memcpy(saved, ctx->theta, sizeof(saved));
    for (int k = 0; k <= ctx->d; k++) ctx->theta[k] = theta[k];
    ctx->have_theta = true;
    // One stream synchronisation per evaluation: the factorisation status is not waited for in the
    // middle (every kernel behind a failed factorisation either exits on *dinfo != 0 or works on
    // values nobody reads); status and results come back together at the end.
    // Results and factorisation status land in the pinned, device-mapped staging buffer: the last kernels
    // of the evaluation write them there themselves (no copy-out operations behind the evaluation).
    int rc = ensure_pinned(ctx, 4096);
    double* hres = static_cast<double*>(ctx->hpin);
    double* dres = static_cast<double*>(ctx->hpin_dev);     // the same buffer as the device sees it
    if (rc == 0) { hres[RES_INFO] = -1.0; hres[RES_INFO + 1] = -1.0; }      // "not written" marker
    // N <= 128, d <= 16: the whole evaluation in one launch of one workgroup (lml_small.hip); it leaves no factor in
    // dW / dW2 and does not touch the scaled coordinates of the prediction factor
    bool fused = false;
    double fhost[2 + 1 + GPRY_MAX_DIM];
    int finfo = 0;
    if (rc == 0 && ctx->opt_lml_small && ctx->opt_chol == 0) {
        StageScope s(ctx, "lml_small");
        const int r = launch_lml_small(ctx, want_grad, fhost, &finfo);     // returns with the results in hand
        if (r < 0) rc = r;
        fused = r == 0;
    }
    const bool rescaled = rc == 0 && !fused;        // build_factor scales the training coordinates for THIS theta
    if (rescaled) rc = lml_chain(ctx, want_grad, dres);
    memcpy(ctx->theta, saved, sizeof(saved));
    ctx->have_theta = had;
    // the scaled training coordinates belong to the prediction factor: whoever needs them next restores them
    // (ensure_pred_xs; an optimiser's next evaluation does not)
    if (rescaled && had && ctx->factor_valid) ctx->xs_foreign = true;
    double host[2 + 1 + GPRY_MAX_DIM];
    int hinfo[2] = {0, 0};
    if (rc == 0 && fused) {
        memcpy(host, fhost, sizeof(double) * (2 + (want_grad ? ctx->d + 1 : 0)));
        hinfo[0] = finfo;
    } else if (rc == 0) {
        // (Polling the status word in the mapped buffer instead of waiting for the stream returns 15 us earlier and is
        // WRONG: inbound PCIe writes to different cache lines are not ordered here -- the host saw the status before
        // the last gradient entries in 2 % of the evaluations, tests/tools/stress_concurrent_fit.py; a loop over
        // hipStreamQuery is correct and no faster than hipStreamSynchronize.)
        hipError_t e = hipStreamSynchronize(ctx->stream);
        if (e != hipSuccess) rc = gpry_fail(ctx, -2, "lml: %s", hipGetErrorString(e));
        else if (hres[RES_INFO] < 0.0) rc = gpry_fail(ctx, -2, "lml: the evaluation did not deliver its status");
        else {
            memcpy(host, hres, sizeof(double) * (2 + (want_grad ? ctx->d + 1 : 0)));
            hinfo[0] = (int)hres[RES_INFO]; hinfo[1] = (int)hres[RES_INFO + 1];
            if (panel_timed_out((int)hres[RES_INFO + 2])) rc = panel_timeout_error(ctx, (int)hres[RES_INFO + 2]);
        }
    }
    const int inf = hinfo[0] != 0 ? hinfo[0] : hinfo[1];
    ctx->lml_cache = (rc == 0 && inf == 0 && !fused);
    if (ctx->lml_cache) {
        memset(ctx->lml_theta, 0, sizeof(ctx->lml_theta));
        for (int k = 0; k <= ctx->d; k++) ctx->lml_theta[k] = theta[k];
        ctx->lml_kernel_id = ctx->kernel_id;
    }
    if (rc) return rc;
    if (info) *info = inf;
    if (inf != 0) {   // sklearn:_gpr.py:586-589
        *lml = -INFINITY;
        if (want_grad && grad) for (int k = 0; k <= ctx->d; k++) grad[k] = 0.0;
        return 0;
    }
    *lml = -0.5 * host[1] - host[0] - 0.5 * (double)ctx->N * log(2.0 * M_PI);
    if (want_grad && grad) for (int k = 0; k <= ctx->d; k++) grad[k] = host[2 + k];
    return 0;
}

// B objective evaluations in one call: the optimiser runs of a multi-restart fit stepped side by side (gpry/gpr.py:883-994 runs
// them one after another).  N <= 128, d <= 16: ONE launch, one workgroup per theta (lml_small.hip); up to Np = lml_batch
// (default 4096): ONE chain of launches for all thetas (lml_batch_general) -- either way every theta gets the arithmetic of a
// single gpry_lml call, hence the same bits; otherwise the thetas are evaluated one after another.
int gpry_lml_batch(gpry_ctx* ctx, const double* thetas, int64_t B, int want_grad, double* lml, double* grad, int* info) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_lml_batch: ctx is NULL");
    if (B <= 0) return 0;
    if (!thetas || !lml || (want_grad && !grad)) return gpry_fail(ctx, -1, "lml_batch: thetas, lml and (with want_grad) grad must not be NULL");
    const int w = ctx->d + 1;
    bool fused = ctx->N > 0 && ctx->opt_lml_small && ctx->opt_chol == 0 && ctx->Np == 128 && ctx->d <= 16 && B <= 256;
    for (int64_t i = 0; fused && i < B * w; i++) fused = isfinite(thetas[i]);
    // (throughput schedule: a single theta goes through the chain of the many as well -- its result must not depend on B)
    if (!fused && (B >= 2 || ctx->opt_lml_schedule == 1) && lml_batch_usable(ctx)) {
        const int r = lml_batch_general(ctx, thetas, B, want_grad, lml, grad, info);
        if (r <= 0) return r;               // 1: no room for two sets, one after another below
    }
    if (!fused) {
        for (int64_t b = 0; b < B; b++) {
            int inf = 0;
            GPRY_TRY(gpry_lml(ctx, thetas + b * w, want_grad, lml + b, want_grad ? grad + b * w : nullptr, &inf));
            if (info) info[b] = inf;
        }
        return 0;
    }
    GPRY_TRY(serve_stop(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    std::vector<double> par((size_t)B * 17, 1.0), out((size_t)B * (2 + 1 + GPRY_MAX_DIM), 0.0);
    std::vector<int> inf((size_t)B, 0);
    for (int64_t b = 0; b < B; b++) {
        par[b * 17] = exp(thetas[b * w]);                                   // as make_kp / make_ap for one evaluation
        for (int k = 0; k < ctx->d; k++) par[b * 17 + 1 + k] = exp(thetas[b * w + 1 + k]);
    }
    {
        StageScope s(ctx, "lml_small");
        const int r = launch_lml_small_batch(ctx, (int)B, par.data(), want_grad, out.data(), inf.data());
        if (r < 0) return r;
        if (r != 0) return gpry_fail(ctx, -1, "lml_batch: the model does not fit the single-launch kernel");
    }
    ctx->lml_cache = false;
    for (int64_t b = 0; b < B; b++) {
        const double* o = out.data() + b * (2 + 1 + GPRY_MAX_DIM);
        if (info) info[b] = inf[b];
        if (inf[b] != 0) {   // sklearn:_gpr.py:586-589
            lml[b] = -INFINITY;
            if (want_grad) for (int k = 0; k < w; k++) grad[b * w + k] = 0.0;
            continue;
        }
        lml[b] = -0.5 * o[1] - o[0] - 0.5 * (double)ctx->N * log(2.0 * M_PI);
        if (want_grad) for (int k = 0; k < w; k++) grad[b * w + k] = o[2 + k];
    }
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// sweep
struct FinishParams {
    double C, y_mean, y_std, clip_hi, zeta, baseline, sigma_n;
    int want_std, want_acq;
};

// LogExp.f on one (mean, std) pair (gpry/acquisition_functions.py:1068-1074): log sqrt(0) = -inf and a
// mean of -inf give -inf, as numpy does under the errstate the reference sets (gp_acquisition.py:1099)
__device__ __forceinline__ double logexp_value(double y, double sd, double zeta, double baseline, double sigma_n) {
    // std**2 - noise**2 as numpy evaluates it: both squares rounded, then the difference.  Contracted
    // into one FMA the cancellation just above sigma_n moved the result by 1e-9 relative (found by the
    // reference's own F5 edge vectors).  -ffp-contract=fast fuses in the backend whatever the source
    // pragmas say, so the products are pinned behind empty asm statements.
    double s2 = sd * sd, n2 = sigma_n * sigma_n;
    asm volatile("" : "+v"(s2));
    asm volatile("" : "+v"(n2));
    double v = s2 - n2;
    if (v < 0.0) v = 0.0;
    double lin = (2.0 * zeta) * (y - baseline);
    asm volatile("" : "+v"(lin));
    return lin + log(sqrt(v));
}
__global__ void logexp_kernel(const double* __restrict__ mu, const double* __restrict__ sd, int64_t n, double zeta,
                              double baseline, double sigma_n, double* __restrict__ acq) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acq[i] = logexp_value(mu[i], sd[i], zeta, baseline, sigma_n);
}

// per candidate: reduce the partials, apply the reference's post-processing chain
// (gpry/gpr.py:1180-1231) and LogExp.f (gpry/acquisition_functions.py:1068-1074)
__global__ void sweep_finish_kernel(const double* __restrict__ mean_part, const double* __restrict__ ss_part,
                                    int nt_mean, int nt, int64_t ldp, int64_t m0, int64_t mc, const uint8_t* __restrict__ mask,
                                    double* __restrict__ y_all, double* __restrict__ sig_all,
                                    double* __restrict__ acq_all, FinishParams fp) {
    int64_t ml = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (ml >= mc) return;
    int64_t m = m0 + ml;
    double mu_ = 0.0;
#pragma unroll 8
    for (int t = 0; t < nt_mean; t++) mu_ += mean_part[(int64_t)t * ldp + ml];
    double y = mu_ * fp.y_std + fp.y_mean;
    y = fmin(y, fp.clip_hi);
    unsigned mk = mask ? mask[m] : 0u;
    if (mk) y = -INFINITY;
    y_all[m] = y;
    if (!fp.want_std) return;
    double ss = 0.0;
#pragma unroll 8
    for (int t = 0; t < nt; t++) ss += ss_part[(int64_t)t * ldp + ml];
    double var = fp.C - ss;
    if (var < 0.0) var = 0.0;
    double sd = sqrt(var) * fp.y_std;
    if (mk & GPRY_MASK_CLASSIFIED_INF) sd = 0.0;
    sig_all[m] = sd;
    if (!fp.want_acq) return;
    acq_all[m] = logexp_value(y, sd, fp.zeta, fp.baseline, fp.sigma_n);
}

// Split-K contraction of a small batch: the slices P[y] (Np x ldp each, `stride` doubles apart) hold
// partial products of u = V k*; per 128-row tile ti and candidate m
//     ss_part[ti][m] = sum_{i in tile} ( sum_y P[y][i][m] )^2
// -- the same per-tile partials the SUMSQ epilogue of the one-pass contraction leaves, slices added
// in a fixed order (deterministic).  Block = one row tile x 64 candidates, 4 waves x 32 rows.
template <int NS>
__global__ __launch_bounds__(1024) void splitk_sumsq_kernel(const double* __restrict__ P, int64_t stride,
                                                            int64_t ldp, double* __restrict__ ss_part) {
    // 1024 threads = 64 candidates x 16 row groups of 8 rows: every thread has its NS x 2 loads of two
    // rows in flight at once (a 256-thread version that walked 32 rows x NS slices per thread was a
    // chain of dependent memory round trips: 100+ us for a 20-us amount of data)
    __shared__ double red[16][64];
    const int ti = blockIdx.x, col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.y * 64 + col;
    double acc = 0.0;
#pragma unroll
    for (int rr = 0; rr < 8; rr += 2) {
        const int64_t off = ((int64_t)ti * 128 + rg * 8 + rr) * ldp + c;
        double v0[NS], v1[NS];
#pragma unroll
        for (int y = 0; y < NS; y++) { v0[y] = P[(int64_t)y * stride + off]; v1[y] = P[(int64_t)y * stride + off + ldp]; }
        double u0 = 0.0, u1 = 0.0;
#pragma unroll
        for (int y = 0; y < NS; y++) { u0 += v0[y]; u1 += v1[y]; }      // slices in a fixed order
        acc = fma(u0, u0, acc);
        acc = fma(u1, u1, acc);
    }
    red[rg][col] = acc;
    __syncthreads();
    if (rg == 0) {
        double s = 0.0;
#pragma unroll
        for (int k = 0; k < 16; k++) s += red[k][col];
        ss_part[(int64_t)ti * ldp + c] = s;
    }
}

static int ensure_sweep_buffers(gpry_ctx* ctx, int64_t M) {
    if (M > ctx->sw_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        void* old[] = {ctx->dXc, ctx->dmask, ctx->dy_all, ctx->dsig_all, ctx->dacq_all};
        for (void* p : old) if (p) GPRY_TRY(dev_free(ctx, p));
        int64_t cap = round_up(M, 1024);
        GPRY_TRY(dev_alloc(ctx, &ctx->dXc, cap * GPRY_MAX_DIM));
        GPRY_TRY(dev_alloc(ctx, &ctx->dmask, cap));
        GPRY_TRY(dev_alloc(ctx, &ctx->dy_all, cap));
        GPRY_TRY(dev_alloc(ctx, &ctx->dsig_all, cap));
        GPRY_TRY(dev_alloc(ctx, &ctx->dacq_all, cap));
        ctx->sw_cap = cap;
    }
    return 0;
}

// runs the chunked sweep over candidates resident in ctx->dXc
static int run_sweep(gpry_ctx* ctx, int64_t M, bool have_mask, bool want_std, bool want_acq,
                     double zeta, double baseline, double sigma_n, bool allow_split = false) {
    const int64_t Np = ctx->Np;
    const int nt = (int)(Np / 128);
    // candidates per chunk: the K* panel of a chunk (Np x chunk doubles) is 1 GiB at Np = 4096 and stays that size for smaller
    // models -- at Np = 1024 the 1e5 candidates of BASELINE configs[1] are ONE launch of 6256 tiles instead of three and a
    // ragged fourth (contraction 0.60 -> 0.71 of peak); a candidate's result does not depend on the chunking
    int64_t chunk = ctx->opt_sweep_chunk;
    if (chunk <= 0) chunk = Np < 4096 ? round_up(32768 * 4096 / Np, 1024) : 32768;
    if (chunk > round_up(M, 128)) chunk = round_up(M, 128);
    // "sweep_overlap" = 1 (round 6): the cross-kernel panel of chunk c + 1 is built on the side stream while the main stream
    // contracts chunk c -- two panels and two sets of partial sums, one event per hand-over.  Same kernels on the same data:
    // same bits.  Only for sweeps of several chunks with the one-pass contraction.
    const bool overlap = ctx->opt_sweep_overlap && ctx->stream2 != nullptr && !allow_split && want_std && M > chunk;
    const int nbuf = overlap ? 2 : 1;
    if (nbuf * Np * chunk > ctx->kst_cap) {
        if (ctx->dKst) GPRY_TRY(dev_free(ctx, ctx->dKst));
        ctx->dKst = nullptr; ctx->kst_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dKst, nbuf * Np * chunk));
        ctx->kst_cap = nbuf * Np * chunk;
    }
    // gpry_predict with a few hundred points: the panel comes from the small-batch kernel, which leaves
    // four mean partials per 128 training rows (kernel_build.hip: cross_build_small_kernel)
    const bool small_build = allow_split && M <= 512;
    const int nt_mean = small_build ? 4 * nt : nt;
    const int64_t part_stride = (int64_t)(nt_mean + nt) * chunk;
    GPRY_TRY(ensure_part(ctx, nbuf * part_stride));
    FinishParams fp;
    fp.C = exp(ctx->theta[0]); fp.y_mean = ctx->tf.y_mean; fp.y_std = ctx->tf.y_std;
    fp.clip_hi = ctx->tf.clip_hi; fp.zeta = zeta; fp.baseline = baseline; fp.sigma_n = sigma_n;
    fp.want_std = want_std; fp.want_acq = want_acq;
    ctx->sw_M = M;
    // distances of the panel from the matrix pipe (cross_build_mfma_kernel; "cross_mfma" = 0: the difference form)
    // (not for Matern-1/2: exp(-r) has a cusp at r = 0, where the rounding noise e of the expanded r^2 becomes sqrt(e) in r --
    // 1e-7 in k for a candidate on a training point; the smoother kernels see e itself)
    bool fast_panel = ctx->opt_cross_mfma && !small_build && ctx->kernel_id != GPRY_MATERN12;
    bool hybrid_panel = false;
    ctx->panel_form = small_build ? 3 : 2;
    ctx->panel_est[0] = ctx->panel_est[1] = ctx->panel_est[2] = 0.0; ctx->panel_est[3] = 2.5e-7;
    if (fast_panel) {
        // ... and not for a model that would amplify that noise beyond the posterior tolerance.  The expanded form has
        // |d r^2| <= 4 eps (|x - c|^2 + |y - c|^2) <= 4 eps (2 r^2 + 4 R^2), with R^2 the largest |y - c|^2 of a training row
        // (bounded below by the per-dimension extent of the training set) -- whatever the candidate: a far one has a large
        // r^2, and r^2 |dk / d r^2| <= C / 2, |dk / d r^2| <= 1.5 C for the three smooth kernels.  Every entry of K* is thus
        // off by at most e = 4 eps C (1 + 6 R^2) -- attained only by a candidate that sits on a training row at the rim of
        // the set; a random candidate sees a small fraction of it.
        //   * MEAN = k*^T alpha_: at worst e ||alpha_||_1 (est[1], reported); the rounding errors of different pairs being
        //     independent, in effect e ||alpha_||_2 (est[0], gated).
        //   * VARIANCE = C - ||V k*||^2: d var = -2 w^T dk with w = K^-1 k*, i.e. 2 e ||w||_2 in the same statistical sense.
        //     ||w||_2^2 = k*^T K^-2 k* <= ||K^-1||_2 k*^T K^-1 k* <= C / lambda_min(K) (the posterior variance is >= 0), and
        //     lambda_min(K) >= the smallest noise variance on the diagonal: ||w||_2 <= sqrt(C) / sigma_n,min for EVERY
        //     candidate (typical candidates have ||w||_2 = O(1); the bound is attained by a k* along the weakest eigenvector).
        //     Relative to C: est[2] = 2 e / (sigma_n,min sqrt(C)).  (Round 6: until then the gate had no variance term.)
        // Both are held below 2.5e-7 (est[3]) -- of the unit-variance normalised targets, resp. of the prior variance C --, a
        // quarter of the 1e-6 the posterior is specified to; on well-conditioned models they are 1e-12 ... 1e-10 (the parity
        // tests compare at 1e-8 / 1e-9 C).  A nearly singular K (tiny noise, long length scales: large alpha_) or length
        // scales far below the extent of the training set (large R^2) take the difference form, whose entries are good to
        // 1e-15 C.  (BASELINE configs[2] as the bench fits it -- several length scales at their lower bound of 1e-3, R^2 =
        // 4e5, ||alpha_||_2 = 64 -- comes to 1.3e-7 for the mean and, since round 6, fails on the variance term.)
        if (ctx->alpha_l2 < 0.0) {
            std::vector<double> ha((size_t)ctx->N);
            HIP_TRY(ctx, hipMemcpyAsync(ha.data(), ctx->dalpha_, sizeof(double) * ctx->N, hipMemcpyDeviceToHost, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            double ss = 0.0, s1 = 0.0;
            for (double v : ha) { ss += v * v; s1 += fabs(v); }
            ctx->alpha_l2 = sqrt(ss); ctx->alpha_l1 = s1;
        }
        double R2 = 0.0;
        for (int k = 0; k < ctx->d; k++) {
            const double a = ctx->xhi[k] - ctx->xcenter[k], b = ctx->xcenter[k] - ctx->xlo[k];
            const double r = (a > b ? a : b) * exp(-ctx->theta[1 + k]);
            R2 += r * r;
        }
        const double C = exp(ctx->theta[0]);
        const double e = 4.0 * 2.220446049250313e-16 * C * (1.0 + 6.0 * R2);
        ctx->panel_est[0] = e * ctx->alpha_l2;
        ctx->panel_est[1] = e * ctx->alpha_l1;
        ctx->panel_est[2] = ctx->noise_min > 0.0 ? 2.0 * e / (sqrt(ctx->noise_min) * sqrt(C)) : INFINITY;
        if (!(ctx->panel_est[0] <= ctx->panel_est[3]) || !(ctx->panel_est[2] <= ctx->panel_est[3])) fast_panel = false;
        if (ctx->opt_panel_debug & 32) fast_panel = true;       // test hook: the matrix-pipe form whatever the estimates say
        if (fast_panel) ctx->panel_form = 1;
        // THE HYBRID FORM (round 6; cross_build_mfma_kernel<.., HYB>): distances from the matrix pipe, and every pair that comes
        // out nearer than r^2 = 100 -- the only ones whose kernel value listens to r^2 at the 1e-15 level -- again from the
        // coordinates, as the difference form does.  Its entries are as good as the difference form's; what decides between the
        // two is cost: a model that failed the gate through R^2 (length scales far below the extent of the data: the bench's
        // fitted model) has next to no near pairs and pays the matrix-pipe price; one that failed it through its weights at
        // ordinary length scales has nothing else and takes the difference form.
        else if (ctx->opt_cross_hybrid && R2 >= 1000.0) { hybrid_panel = true; ctx->panel_form = 4; }
        if (getenv("GPRY_HIP_DEBUG_PANEL")) {
            fprintf(stderr, "gpry: panel form: C %.3g R2 %.3g |alpha|_2 %.3g |alpha|_1 %.3g min noise %.3g -> mean %.3g (l1 %.3g) var %.3g: %s; l =", C, R2,
                    ctx->alpha_l2, ctx->alpha_l1, ctx->noise_min, ctx->panel_est[0], ctx->panel_est[1], ctx->panel_est[2],
                    fast_panel ? "matrix pipe" : hybrid_panel ? "hybrid" : "difference form");
            for (int k = 0; k < ctx->d; k++) fprintf(stderr, " %.3g (%.3g..%.3g)", exp(ctx->theta[1 + k]), ctx->xlo[k], ctx->xhi[k]);
            fprintf(stderr, "\n");
        }
    }
    if (fast_panel || hybrid_panel) GPRY_TRY(launch_cross_prepare(ctx));
    // A fresh pool (gpry_sweep_logexp with a host array, option "sweep_upload"): the rows of chunk c go up on stream2 while
    // the main stream still works on chunk c - 1 -- 4.2 MB against 7.7 ms of kernels at N = 4096 -- and the main stream
    // waits for nothing but its own chunk (one event per chunk, never re-recorded within a call).  From pageable memory
    // hipMemcpyAsync returns when the rows are staged, so the host is one chunk ahead of the GPU, which is all it takes.
    const double* up_X = ctx->up_X;
    const size_t nchunk = (size_t)((M + chunk - 1) / chunk);
    if (up_X || overlap) {
        while (ctx->ev_pool.size() < 3 * nchunk + 1) {
            hipEvent_t ev;
            HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
            ctx->ev_pool.push_back(ev);
        }
    }
    const hipStream_t main_stream = ctx->stream, side = ctx->stream2;
    // the launchers queue on ctx->stream: for the work of the side stream it is swapped for the duration of the call
    struct StreamSwap {
        gpry_ctx* c; hipStream_t keep;
        StreamSwap(gpry_ctx* ctx, hipStream_t st) : c(ctx), keep(ctx->stream) { c->stream = st; }
        ~StreamSwap() { c->stream = keep; }
    };
    // upload (and gates) of chunk ci on the side stream; `ev_up` = ev_pool[ci]
    auto upload_chunk = [&](size_t ci, bool gates_on_side) -> int {
        const int64_t m0 = (int64_t)ci * chunk, mc = (M - m0 < chunk) ? M - m0 : chunk;
        HIP_TRY(ctx, hipMemcpyAsync(ctx->dXc + m0 * ctx->d, up_X + m0 * ctx->d, sizeof(double) * mc * ctx->d,
                                    hipMemcpyHostToDevice, side));
        if (gates_on_side && ctx->up_gates) {
            StreamSwap sw(ctx, side);
            StageScope s(ctx, "gates");
            GPRY_TRY(launch_gates(ctx, ctx->dXc + m0 * ctx->d, mc, ctx->dmask + m0));
        }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_pool[ci], side));
        return 0;
    };
    auto build_panel = [&](size_t ci, double* Kst, double* mean_part) -> int {
        const int64_t m0 = (int64_t)ci * chunk, mc = (M - m0 < chunk) ? M - m0 : chunk, mcp = round_up(mc, 128);
        StageScope s(ctx, "cross_build");
        if (small_build) return launch_cross_build_small(ctx, ctx->dXc, m0, mcp, mcp, Kst, mean_part, 1);
        if (fast_panel || hybrid_panel) return launch_cross_build_mfma(ctx, ctx->dXc, m0, mcp, mcp, Kst, mean_part, 1, hybrid_panel ? 1 : 0);
        return launch_cross_build(ctx, ctx->dXc, m0, mcp, mcp, Kst, mean_part, 1);
    };
    if (overlap) {
        // the side stream starts behind what the main stream has queued so far (the scaled / centred training rows, the mask)
        hipEvent_t ev0 = ctx->ev_pool[3 * nchunk];
        HIP_TRY(ctx, hipEventRecord(ev0, main_stream));
        HIP_TRY(ctx, hipStreamWaitEvent(side, ev0, 0));
        if (up_X) GPRY_TRY(upload_chunk(0, true));
        { StreamSwap sw(ctx, side); GPRY_TRY(build_panel(0, ctx->dKst, ctx->dpart)); }
        HIP_TRY(ctx, hipEventRecord(ctx->ev_pool[nchunk], side));
    }
    for (int64_t m0 = 0; m0 < M; m0 += chunk) {
        int64_t mc = (M - m0 < chunk) ? M - m0 : chunk;
        int64_t mcp = round_up(mc, 128);
        const size_t ci = (size_t)(m0 / chunk);
        const int buf = overlap ? (int)(ci & 1) : 0;
        double* Kst = ctx->dKst + (int64_t)buf * Np * chunk;
        double* mean_part = ctx->dpart + (int64_t)buf * part_stride;
        double* ss_part = mean_part + (int64_t)nt_mean * chunk;
        if (overlap) {
            // side stream: upload and panel of the NEXT chunk, into the buffers chunk ci - 1 has finished with
            if (ci + 1 < nchunk) {
                if (up_X) GPRY_TRY(upload_chunk(ci + 1, true));
                if (ci >= 1) HIP_TRY(ctx, hipStreamWaitEvent(side, ctx->ev_pool[2 * nchunk + ci - 1], 0));
                const int nb = (int)((ci + 1) & 1);
                { StreamSwap sw(ctx, side); GPRY_TRY(build_panel(ci + 1, ctx->dKst + (int64_t)nb * Np * chunk, ctx->dpart + (int64_t)nb * part_stride)); }
                HIP_TRY(ctx, hipEventRecord(ctx->ev_pool[nchunk + ci + 1], side));
            }
            HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pool[nchunk + ci], 0));
        } else {
            if (up_X) {
                GPRY_TRY(upload_chunk(ci, false));
                HIP_TRY(ctx, hipStreamWaitEvent(main_stream, ctx->ev_pool[ci], 0));
                if (ctx->up_gates) {        // the SVM / trust-region verdicts of this chunk, on top of the caller's bits
                    StageScope s(ctx, "gates");
                    GPRY_TRY(launch_gates(ctx, ctx->dXc + m0 * ctx->d, mc, ctx->dmask + m0));
                }
            }
            GPRY_TRY(build_panel(ci, Kst, mean_part));
        }
        // A batch of a few hundred to a few thousand points has fewer tiles than the GPU has workgroup
        // slots, and its longest tile walks all Np/16 slabs alone (1 ms at Np = 4096): split every
        // tile's k-range over grid.y so that ~512 workgroups share the contraction, keep the partial
        // products u_y in scratch and square their sum in a second, small kernel.
        // Only for gpry_predict: the NORA sweep keeps the one-pass contraction, whose result for a candidate
        // does not depend on which other candidates share its launch -- a pool sharded over several
        // contexts / GPUs then gives bit for bit what one context gives (tests/test_group_gpu.py).
        int nsplit = 1;
        if (allow_split && want_std && ctx->opt_predict_split && M <= chunk) {
            const int64_t tiles = (int64_t)nt * (mcp / 128);
            while (nsplit < 16 && tiles * nsplit * 2 <= 1024 && Np / (nsplit * 2) >= 64) nsplit *= 2;
        }
        if (want_std && nsplit > 1) {
            StageScope s(ctx, "sweep_gemm_splitk");
            double* sbuf = nullptr;
            GPRY_TRY(gemm_split_scratch(ctx, nsplit, Np * mcp, &sbuf));
            GemmArgs g = {};
            g.A = ctx->dV; g.lda = Np; g.B = Kst; g.ldb = mcp; g.C = sbuf; g.ldc = mcp;
            g.M = (int)Np; g.N = (int)mcp; g.K = (int)Np;
            g.kmode = KM_A_LOWER; g.tile_map = TM_ROWMAJOR;
            g.nsplit = nsplit; g.split_buf = sbuf; g.split_stride = Np * mcp; g.skip_reduce = 1;
            GPRY_TRY(gemm_f64_launch(ctx, g, false, false, EPI_STORE));
            const dim3 rg((unsigned)nt, (unsigned)(mcp / 64));
            switch (nsplit) {
                case 2: hipLaunchKernelGGL(splitk_sumsq_kernel<2>, rg, dim3(1024), 0, ctx->stream, sbuf, Np * mcp, mcp, ss_part); break;
                case 4: hipLaunchKernelGGL(splitk_sumsq_kernel<4>, rg, dim3(1024), 0, ctx->stream, sbuf, Np * mcp, mcp, ss_part); break;
                case 8: hipLaunchKernelGGL(splitk_sumsq_kernel<8>, rg, dim3(1024), 0, ctx->stream, sbuf, Np * mcp, mcp, ss_part); break;
                default: hipLaunchKernelGGL(splitk_sumsq_kernel<16>, rg, dim3(1024), 0, ctx->stream, sbuf, Np * mcp, mcp, ss_part); break;
            }
            HIP_TRY(ctx, hipGetLastError());
        } else if (want_std) {
            StageScope s(ctx, "sweep_gemm");
            GemmArgs g = {};
            g.A = ctx->dV; g.lda = Np; g.B = Kst; g.ldb = mcp; g.C = ss_part; g.ldc = mcp;
            g.M = (int)Np; g.N = (int)mcp; g.K = (int)Np;
            g.kmode = KM_A_LOWER; g.lower_only = 0; g.tile_map = TM_SWEEP | (3 << 4);     // super-tiles of 8 row tiles x 8 candidate tiles
            // LDS-DMA staging + software pipeline (sweep_gemm.hip); "gemm_dma" = 0: the register-staged engine (comparator)
            if (ctx->opt_gemm_dma) GPRY_TRY(sweep_gemm_dma_sp_launch(ctx, g));
            else GPRY_TRY(gemm_f64_launch(ctx, g, false, false, EPI_SUMSQ));
        }
        {
            StageScope s(ctx, "sweep_finish");
            hipLaunchKernelGGL(sweep_finish_kernel, dim3((unsigned)((mc + 255) / 256)), dim3(256), 0, ctx->stream,
                               mean_part, ss_part, nt_mean, nt, mcp, m0, mc, have_mask ? ctx->dmask : nullptr,
                               ctx->dy_all, ctx->dsig_all, ctx->dacq_all, fp);
            HIP_TRY(ctx, hipGetLastError());
        }
        if (overlap) HIP_TRY(ctx, hipEventRecord(ctx->ev_pool[2 * nchunk + ci], main_stream));
    }
    return 0;
}

static int upload_candidates(gpry_ctx* ctx, const double* X, int64_t M, const uint8_t* mask, bool upload_later = false) {
    GPRY_TRY(ensure_sweep_buffers(ctx, M));
    if (X) HIP_TRY(ctx, hipMemcpyAsync(ctx->dXc, X, sizeof(double) * M * ctx->d, hipMemcpyHostToDevice, ctx->stream));
    else if (upload_later) { }      // (the caller's rows reach dXc chunk by chunk inside run_sweep)
    else if (ctx->sw_M != M) return gpry_fail(ctx, -1, "X == NULL but no resident candidate set of size %lld", (long long)M);
    if (mask) HIP_TRY(ctx, hipMemcpyAsync(ctx->dmask, mask, (size_t)M, hipMemcpyHostToDevice, ctx->stream));
    return 0;
}

// gpry_predict works on its own candidate set: swap it in for the duration of the call
struct PredictSetGuard {
    gpry_ctx* c;
    explicit PredictSetGuard(gpry_ctx* ctx) : c(ctx) { swap(); }
    ~PredictSetGuard() { swap(); }
    void swap() {
        std::swap(c->sw_M, c->pr.M); std::swap(c->sw_cap, c->pr.cap);
        std::swap(c->dXc, c->pr.dXc); std::swap(c->dmask, c->pr.dmask);
        std::swap(c->dy_all, c->pr.dy); std::swap(c->dsig_all, c->pr.dsig);
        std::swap(c->dacq_all, c->pr.dacq);
    }
};

__global__ void count_nan_kernel(const double* __restrict__ a, int64_t n, unsigned long long* out) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long c = 0;
    for (; i < n; i += stride) c += (a[i] != a[i]) ? 1ull : 0ull;
    for (int off = 32; off >= 1; off >>= 1) c += __shfl_xor(c, off);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, c);
}

extern "C" {

int gpry_sweep_info(gpry_ctx* ctx, int* panel_form, double* est) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_sweep_info: ctx is NULL");
    if (panel_form) *panel_form = ctx->panel_form;
    if (est) for (int k = 0; k < 4; k++) est[k] = ctx->panel_est[k];
    return 0;
}

int gpry_predict(gpry_ctx* ctx, const double* X, int64_t M, const uint8_t* mask, double* mean, double* std) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_predict: ctx is NULL");
    GPRY_TRY(require_model(ctx, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (M <= 0) return 0;
    if (!X) return gpry_fail(ctx, -1, "predict: X is NULL");
    if (!std && M <= GPRY_SERVE_MAXM && ctx->opt_predict_serve && ctx->opt_predict_small > 0) {
        // Latency path of the point-by-point callers (gpry/gp_acquisition.py:766-771, gpry/mc.py:387-391): no kernel
        // launch, the request goes to the resident kernel (server.hip) -- same slices, same sums, same bits as the
        // one-launch path below.
        double part[GPRY_SERVE_MAXM * 8];
        unsigned gbits[GPRY_SERVE_MAXM];
        int nsplit = 1;
        GPRY_TRY(serve_predict_mean(ctx, X, M, part, &nsplit, gbits));
        for (int64_t m = 0; m < M; m++) {
            double mu_ = 0.0;
            for (int sidx = 0; sidx < nsplit; sidx++) mu_ += part[m * nsplit + sidx];
            double y = fmin(mu_ * ctx->tf.y_std + ctx->tf.y_mean, ctx->tf.clip_hi);
            if ((mask && mask[m]) || gbits[m]) y = -INFINITY;
            mean[m] = y;
        }
        return 0;
    }
    GPRY_TRY(serve_stop(ctx));
    // "predict_gates" = 1: the verdicts of gpry_set_gates are ORed into the caller's mask here as gpry_sweep_logexp
    // does (the small paths below post-process on the host: they get a merged host mask; the panel paths run the
    // gates kernel on their resident candidate buffers)
    const bool dev_gates = ctx->gates_on && ctx->opt_predict_gates;
    std::vector<uint8_t> merged;
    if (dev_gates && M <= 4096) {
        const int64_t xb = round_up(sizeof(double) * M * ctx->d, 256);
        GPRY_TRY(ensure_pinned(ctx, xb + round_up(M, 256)));
        char* h = (char*)ctx->hpin; char* hd = (char*)ctx->hpin_dev;
        memcpy(h, X, sizeof(double) * M * ctx->d);
        if (mask) memcpy(h + xb, mask, (size_t)M); else memset(h + xb, 0, (size_t)M);
        GPRY_TRY(launch_gates(ctx, (const double*)hd, M, (uint8_t*)(hd + xb)));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        merged.assign((uint8_t*)(h + xb), (uint8_t*)(h + xb) + M);
        mask = merged.data();
    }
    if (!std && M <= ctx->opt_predict_small) {   // (predict_small = 0 switches both small-batch paths off)
        // Latency path (samplers call this per point, gpry/gp_acquisition.py:769-793): the points,
        // the mask and the result live in one pinned host buffer that the kernel reads and
        // writes directly -- one launch and one stream synchronisation, no copies, no timers.
        // The training rows are split over up to 8 workgroups per point; the host adds the
        // slices in a fixed order and applies the affine map, the clip and the mask.
        int nsplit = (int)(ctx->N / 1024);
        if (nsplit < 1) nsplit = 1;
        if (nsplit > 8) nsplit = 8;
        const int64_t xb = round_up(sizeof(double) * M * ctx->d, 256);
        GPRY_TRY(ensure_pinned(ctx, xb + sizeof(double) * M * nsplit));
        char* h = (char*)ctx->hpin;
        char* hd = (char*)ctx->hpin_dev;           // the same buffer as the device sees it
        memcpy(h, X, sizeof(double) * M * ctx->d);
        double* hp = (double*)(h + xb);
        GPRY_TRY(launch_predict_mean_small(ctx, (const double*)hd, M, nsplit, (double*)(hd + xb)));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        for (int64_t m = 0; m < M; m++) {
            double mu_ = 0.0;
            for (int sidx = 0; sidx < nsplit; sidx++) mu_ += hp[m * nsplit + sidx];
            double y = fmin(mu_ * ctx->tf.y_std + ctx->tf.y_mean, ctx->tf.clip_hi);
            if (mask && mask[m]) y = -INFINITY;
            mean[m] = y;
        }
        return 0;
    }
    if (std && M <= 4 && ctx->opt_predict_small > 0) {
        // a handful of points with std: k* rows + one multi-vector triangular product; partial sums come
        // back through pinned memory.  54 us for one point at N = 4096 and ~16 us per further point; from
        // five points on the split-K contraction below (138 us for up to 128 points) is faster
        const int64_t Np = ctx->Np, nmb = (Np + 255) / 256, nsb = Np / 16;
        const int64_t xb = round_up(sizeof(double) * M * ctx->d, 256);
        GPRY_TRY(ensure_pinned(ctx, xb + sizeof(double) * M * (nmb + nsb)));
        if (M * Np > ctx->g_cap) {
            if (ctx->dG) GPRY_TRY(dev_free(ctx, ctx->dG));
            ctx->dG = nullptr; ctx->g_cap = 0;
            GPRY_TRY(dev_alloc(ctx, &ctx->dG, 16 * Np));
            ctx->g_cap = 16 * Np;
        }
        char* h = (char*)ctx->hpin;
        char* hd = (char*)ctx->hpin_dev;           // the same buffer as the device sees it
        memcpy(h, X, sizeof(double) * M * ctx->d);
        double* hm = (double*)(h + xb);
        double* hs = hm + M * nmb;
        GPRY_TRY(launch_predict_small_std(ctx, (const double*)hd, (int)M, ctx->dG, (double*)(hd + xb),
                                          (double*)(hd + xb) + M * nmb));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        const double C = exp(ctx->theta[0]);
        for (int64_t m = 0; m < M; m++) {
            double mu_ = 0.0, ss = 0.0;
            for (int64_t b = 0; b < nmb; b++) mu_ += hm[m * nmb + b];
            for (int64_t b = 0; b < nsb; b++) ss += hs[m * nsb + b];
            double y = fmin(mu_ * ctx->tf.y_std + ctx->tf.y_mean, ctx->tf.clip_hi);
            unsigned mk = mask ? mask[m] : 0u;
            if (mk) y = -INFINITY;
            mean[m] = y;
            double var = C - ss;
            if (var < 0.0) var = 0.0;
            double sd = sqrt(var) * ctx->tf.y_std;
            if (mk & GPRY_MASK_CLASSIFIED_INF) sd = 0.0;
            std[m] = sd;
        }
        return 0;
    }
    PredictSetGuard guard(ctx);
    if (M <= 4096) {
        // A few thousand points: candidates, mask and results live in the pinned, device-mapped staging
        // buffer that the kernels read and write directly.  Four pageable hipMemcpyAsync calls cost more
        // (~100 us together) than the whole contraction of such a batch.
        const int64_t xb = round_up(sizeof(double) * M * ctx->d, 256), mb = round_up(M, 256), ob = round_up(sizeof(double) * M, 256);
        GPRY_TRY(ensure_pinned(ctx, xb + mb + 3 * ob));
        char* h = (char*)ctx->hpin;
        char* hd = (char*)ctx->hpin_dev;
        memcpy(h, X, sizeof(double) * M * ctx->d);
        if (mask) memcpy(h + xb, mask, (size_t)M);
        struct Saved { double* X; uint8_t* m; double *y, *s, *a; int64_t cap; } sv =
            {ctx->dXc, ctx->dmask, ctx->dy_all, ctx->dsig_all, ctx->dacq_all, ctx->sw_cap};
        ctx->dXc = (double*)hd; ctx->dmask = (uint8_t*)(hd + xb);
        ctx->dy_all = (double*)(hd + xb + mb); ctx->dsig_all = (double*)(hd + xb + mb + ob);
        ctx->dacq_all = (double*)(hd + xb + mb + 2 * ob);
        int rc = run_sweep(ctx, M, mask != nullptr, std != nullptr, false, 0.0, 0.0, 0.0, true);
        hipError_t e = hipStreamSynchronize(ctx->stream);
        ctx->dXc = sv.X; ctx->dmask = sv.m; ctx->dy_all = sv.y; ctx->dsig_all = sv.s; ctx->dacq_all = sv.a; ctx->sw_cap = sv.cap;
        ctx->sw_M = 0;                       // the staging buffer is not a resident candidate set
        if (rc) return rc;
        if (e != hipSuccess) return gpry_fail(ctx, -2, "predict: %s", hipGetErrorString(e));
        memcpy(mean, h + xb + mb, sizeof(double) * M);
        if (std) memcpy(std, h + xb + mb + ob, sizeof(double) * M);
        return 0;
    }
    GPRY_TRY(upload_candidates(ctx, X, M, mask));
    bool have_mask = mask != nullptr;
    if (dev_gates) {
        if (!have_mask) HIP_TRY(ctx, hipMemsetAsync(ctx->dmask, 0, (size_t)M, ctx->stream));
        GPRY_TRY(launch_gates(ctx, ctx->dXc, M, ctx->dmask));
        have_mask = true;
    }
    GPRY_TRY(run_sweep(ctx, M, have_mask, std != nullptr, false, 0.0, 0.0, 0.0, true));
    HIP_TRY(ctx, hipMemcpyAsync(mean, ctx->dy_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    if (std) HIP_TRY(ctx, hipMemcpyAsync(std, ctx->dsig_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int gpry_predict_grad(gpry_ctx* ctx, const double* x, int want_kinv, double* kgrad, double* mean_grad,
                      double* kinvk_grad) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_predict_grad: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, want_kinv || mean_grad != nullptr));   // kgrad alone needs no factor
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!x) return gpry_fail(ctx, -1, "predict_grad: x is NULL");
    const int64_t Np = ctx->Np;
    const int dpad = ctx->dpad, d = ctx->d;
    if (Np * dpad > ctx->g_cap) {
        if (ctx->dG) GPRY_TRY(dev_free(ctx, ctx->dG));
        ctx->dG = nullptr; ctx->g_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dG, Np * dpad));
        ctx->g_cap = Np * dpad;
    }
    if (!ctx->factor_valid) GPRY_TRY(launch_scale_train(ctx));   // scaled rows are normally made by the factorisation
    if (want_kinv) GPRY_TRY(ensure_part(ctx, (Np / 128) * Np));
    double* kstar = ctx->dvec; double* u = ctx->dvec + Np; double* w = ctx->dvec + 2 * Np;
    double* out = ctx->dvec + 3 * Np; double* xdev = out + 2 * GPRY_MAX_DIM;
    HIP_TRY(ctx, hipMemcpyAsync(xdev, x, sizeof(double) * d, hipMemcpyHostToDevice, ctx->stream));
    {
        StageScope s(ctx, "predict_grad");
        GPRY_TRY(launch_gradx(ctx, xdev, 1, want_kinv, kstar, ctx->dG, u, w, ctx->dpart, out));
    }
    double h[2 * GPRY_MAX_DIM];
    HIP_TRY(ctx, hipMemcpyAsync(h, out, sizeof(double) * 2 * dpad, hipMemcpyDeviceToHost, ctx->stream));
    if (kgrad) GPRY_TRY(copy_out_matrix(ctx, ctx->dG, dpad, ctx->N, d, kgrad));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int k = 0; k < d; k++) {
        if (mean_grad) mean_grad[k] = h[k];
        if (kinvk_grad) kinvk_grad[k] = want_kinv ? h[dpad + k] : 0.0;
    }
    return 0;
}

// ONE point: posterior mean, std and the two x-gradient contractions in one call (what GaussianProcessRegressor.predict
// with return_mean_grad / return_std_grad asks for, once per L-BFGS step of the acquisition optimiser).  Mean and std are
// finalised as by gpry_predict (affine map of y, clipping, mask bits); the gradients are the raw contractions of
// gpry_predict_grad.
int gpry_predict_point(gpry_ctx* ctx, const double* x, int mask_bits, int want_kinv, double* mean, double* std,
                       double* mean_grad, double* kinvk_grad, int* verdict) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_predict_point: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (!x || !mean || !std || !mean_grad || (want_kinv && !kinvk_grad))
        return gpry_fail(ctx, -1, "predict_point: x, mean, std, mean_grad and (with want_kinv) kinvk_grad must not be NULL");
    const int64_t Np = ctx->Np;
    const int dpad = ctx->dpad, d = ctx->d;
    if (Np * dpad > ctx->g_cap) {
        if (ctx->dG) GPRY_TRY(dev_free(ctx, ctx->dG));
        ctx->dG = nullptr; ctx->g_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dG, Np * dpad));
        ctx->g_cap = Np * dpad;
    }
    if (want_kinv) GPRY_TRY(ensure_part(ctx, (Np / 128) * Np));
    const int64_t nmb = (Np + 255) / 256, nsb = Np / 16;
    const int64_t xb = round_up((int64_t)sizeof(double) * GPRY_MAX_DIM, 256);
    GPRY_TRY(ensure_pinned(ctx, xb + (int64_t)sizeof(double) * (nmb + nsb + 2 * GPRY_MAX_DIM) + 256));
    char* h = (char*)ctx->hpin;
    char* hd = (char*)ctx->hpin_dev;
    memcpy(h, x, sizeof(double) * d);
    double* hm = (double*)(h + xb);
    double* hs = hm + nmb;
    double* ho = hs + nsb;
    double* dm = (double*)(hd + xb);
    // "predict_gates" = 1: the classifier / trust-box verdict of gpry_set_gates for this point, ORed into mask_bits (one more
    // small launch in the same stream; the host-side libsvm call it replaces costs 40 us per step of the optimiser)
    const bool dev_gates = ctx->gates_on && ctx->opt_predict_gates;
    uint8_t* hmask = (uint8_t*)(ho + 2 * GPRY_MAX_DIM);
    if (dev_gates) {
        hmask[0] = 0;
        GPRY_TRY(launch_gates(ctx, (const double*)hd, 1, (uint8_t*)(hd + ((char*)hmask - h))));
    }
    {
        StageScope s(ctx, "predict_point");
        GPRY_TRY(launch_point_full(ctx, (const double*)hd, want_kinv, ctx->dvec, ctx->dG, ctx->dvec + Np, ctx->dpart,
                                   dm, dm + nmb, dm + nmb + nsb));
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (dev_gates) mask_bits |= hmask[0];
    if (verdict) *verdict = mask_bits;
    double mu_ = 0.0, ss = 0.0;
    for (int64_t b = 0; b < nmb; b++) mu_ += hm[b];
    for (int64_t b = 0; b < nsb; b++) ss += hs[b];
    double y = fmin(mu_ * ctx->tf.y_std + ctx->tf.y_mean, ctx->tf.clip_hi);
    if (mask_bits) y = -INFINITY;
    *mean = y;
    double var = exp(ctx->theta[0]) - ss;
    if (var < 0.0) var = 0.0;
    double sd = sqrt(var) * ctx->tf.y_std;
    if (mask_bits & GPRY_MASK_CLASSIFIED_INF) sd = 0.0;
    *std = sd;
    for (int k = 0; k < d; k++) {
        mean_grad[k] = ho[k];
        if (kinvk_grad) kinvk_grad[k] = want_kinv ? ho[dpad + k] : 0.0;
    }
    return 0;
}

// column sums of squares of U (Np x ld): ss[i] = |u_i|^2, one workgroup per column
__global__ __launch_bounds__(256) void colsumsq_kernel(const double* __restrict__ U, int64_t ld, int64_t nrow,
                                                       double* __restrict__ ss) {
    __shared__ double red[256];
    const int64_t i = blockIdx.x;
    double acc = 0.0;
    for (int64_t j = threadIdx.x; j < nrow; j += 256) { const double u = U[j * ld + i]; acc = fma(u, u, acc); }
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ss[i] = red[0];
}

int gpry_predict_grad_batch(gpry_ctx* ctx, const double* X, int64_t m, int want_kinv, double* mean, double* std,
                            double* mean_grad, double* kinvk_grad) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_predict_grad_batch: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (m <= 0) return 0;
    if (!X || !mean_grad || (want_kinv && !kinvk_grad)) return gpry_fail(ctx, -1, "predict_grad_batch: X, mean_grad and (with want_kinv) kinvk_grad must not be NULL");
    if (m > 4096) return gpry_fail(ctx, -1, "predict_grad_batch: at most 4096 points per call (got %lld)", (long long)m);
    hipStream_t st = ctx->stream;
    const int64_t Np = ctx->Np, mp = round_up(m, 128);
    const int nt = (int)(Np / 128), d = ctx->d, dpad = ctx->dpad;
    const bool need_u = want_kinv || std != nullptr;
    // workspace: points | k* panel (Np x mp) | U | W | mean partials (nt x mp) | |u|^2 (mp) | gradients (m x 2 dpad)
    const bool small_build = mp <= 512;
    const int ntm = small_build ? 4 * nt : nt;      // mean partials per point
    const int64_t need = round_up(m * d, 2) + 3 * Np * mp + (int64_t)ntm * mp + mp + m * 2 * dpad;
    if (need > ctx->g_cap) {
        if (ctx->dG) GPRY_TRY(dev_free(ctx, ctx->dG));
        ctx->dG = nullptr; ctx->g_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dG, need));
        ctx->g_cap = need;
    }
    double* dXb = ctx->dG;
    double* Kst = dXb + round_up(m * d, 2);
    double* Um = Kst + Np * mp;
    double* Wm = Um + Np * mp;
    double* mpart = Wm + Np * mp;
    double* ss = mpart + (int64_t)ntm * mp;
    double* gout = ss + mp;
    // A few hundred points (the rounds of an acquisition optimiser's restarts): the points and the three small result arrays
    // live in the pinned, device-mapped staging buffer that the kernels read and write directly -- one pageable upload and
    // three pageable downloads cost 40 us of a 130-us call.
    const bool zero_copy = small_build;
    double *hx = nullptr, *hmz = nullptr, *hsz = nullptr, *hgz = nullptr;
    if (zero_copy) {
        const int64_t o1 = round_up(m * d, 32), o2 = o1 + round_up((int64_t)ntm * mp, 32), o3 = o2 + round_up(mp, 32);
        GPRY_TRY(ensure_pinned(ctx, (int64_t)sizeof(double) * (o3 + m * 2 * dpad + 32)));
        double* hb = (double*)ctx->hpin;
        double* db = (double*)ctx->hpin_dev;
        hx = hb; hmz = hb + o1; hsz = hb + o2; hgz = hb + o3;
        memcpy(hx, X, sizeof(double) * m * d);
        dXb = db; mpart = db + o1; ss = db + o2; gout = db + o3;
    } else {
        HIP_TRY(ctx, hipMemcpyAsync(dXb, X, sizeof(double) * m * d, hipMemcpyHostToDevice, st));
    }
    StageScope scope(ctx, "predict_grad_batch");
    const int64_t saveM = ctx->sw_M; ctx->sw_M = m;
    int rc = small_build ? launch_cross_build_small(ctx, dXb, 0, mp, mp, Kst, mpart, 1)
                         : launch_cross_build(ctx, dXb, 0, mp, mp, Kst, mpart, 1);
    ctx->sw_M = saveM;
    if (rc) return rc;
    auto splits = [&](int64_t tiles) { int n = 1; while (n < 16 && tiles * n * 2 <= 1024 && Np / (n * 2) >= 64) n *= 2; return n; };
    if (need_u) {       // U = V K*^T
        GemmArgs g = {};
        g.A = ctx->dV; g.lda = Np; g.B = Kst; g.ldb = mp; g.C = Um; g.ldc = mp;
        g.M = (int)Np; g.N = (int)mp; g.K = (int)Np; g.kmode = KM_A_LOWER; g.tile_map = TM_ROWMAJOR;
        g.nsplit = splits((int64_t)nt * (mp / 128));
        if (g.nsplit > 1) { GPRY_TRY(gemm_split_scratch(ctx, g.nsplit, Np * mp, &g.split_buf)); g.split_stride = Np * mp; }
        GPRY_TRY(gemm_f64_launch(ctx, g, false, false, EPI_STORE));
        hipLaunchKernelGGL(colsumsq_kernel, dim3((unsigned)m), dim3(256), 0, st, Um, mp, Np, ss);
    }
    if (want_kinv) {    // W = V^T U = K^-1 K*^T
        GemmArgs g = {};
        g.A = ctx->dV; g.lda = Np; g.B = Um; g.ldb = mp; g.C = Wm; g.ldc = mp;
        g.M = (int)Np; g.N = (int)mp; g.K = (int)Np; g.kmode = KM_AT_LOWER; g.tile_map = TM_ROWMAJOR;
        g.nsplit = splits((int64_t)nt * (mp / 128));
        if (g.nsplit > 1) { GPRY_TRY(gemm_split_scratch(ctx, g.nsplit, Np * mp, &g.split_buf)); g.split_stride = Np * mp; }
        GPRY_TRY(gemm_f64_launch(ctx, g, true, false, EPI_STORE));
    }
    GPRY_TRY(launch_gradx_batch(ctx, dXb, m, 1, want_kinv ? Wm : nullptr, mp, gout));
    std::vector<double> hmv, hsv, hgv;
    const double *hm = hmz, *hs = hsz, *hg = hgz;
    if (!zero_copy) {
        hmv.resize((size_t)ntm * mp); hsv.resize((size_t)mp); hgv.resize((size_t)m * 2 * dpad);
        HIP_TRY(ctx, hipMemcpyAsync(hmv.data(), mpart, sizeof(double) * ntm * mp, hipMemcpyDeviceToHost, st));
        if (need_u) HIP_TRY(ctx, hipMemcpyAsync(hsv.data(), ss, sizeof(double) * mp, hipMemcpyDeviceToHost, st));
        HIP_TRY(ctx, hipMemcpyAsync(hgv.data(), gout, sizeof(double) * m * 2 * dpad, hipMemcpyDeviceToHost, st));
        hm = hmv.data(); hs = hsv.data(); hg = hgv.data();
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    const double C = exp(ctx->theta[0]);
    for (int64_t i = 0; i < m; i++) {
        if (mean) {
            double mu_ = 0.0;
            for (int t = 0; t < ntm; t++) mu_ += hm[(size_t)t * mp + i];
            mean[i] = fmin(mu_ * ctx->tf.y_std + ctx->tf.y_mean, ctx->tf.clip_hi);
        }
        if (std) {
            double var = C - hs[i];
            if (var < 0.0) var = 0.0;
            std[i] = sqrt(var) * ctx->tf.y_std;
        }
        for (int k = 0; k < d; k++) {
            mean_grad[i * d + k] = hg[(size_t)i * 2 * dpad + k];
            if (kinvk_grad) kinvk_grad[i * d + k] = want_kinv ? hg[(size_t)i * 2 * dpad + dpad + k] : 0.0;
        }
    }
    return 0;
}

int gpry_sweep_fetch(gpry_ctx* ctx, int64_t M, double* y_all, double* sigma_all, double* acq_all) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_sweep_fetch: ctx is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (M <= 0 || M != ctx->sw_M) return gpry_fail(ctx, -1, "sweep_fetch: the resident sweep has %lld candidates, not %lld",
                                                  (long long)ctx->sw_M, (long long)M);
    if (y_all) HIP_TRY(ctx, hipMemcpyAsync(y_all, ctx->dy_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    if (sigma_all) HIP_TRY(ctx, hipMemcpyAsync(sigma_all, ctx->dsig_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    if (acq_all) HIP_TRY(ctx, hipMemcpyAsync(acq_all, ctx->dacq_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

int gpry_set_gates(gpry_ctx* ctx, const double* sv, const double* coef, int64_t n_sv, double gamma,
                   double intercept, int positive_is_finite, const double* trust_bounds) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_set_gates: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (ctx->d <= 0) return gpry_fail(ctx, -1, "set_gates before set_train");
    if (n_sv < 0 || (n_sv > 0 && (!sv || !coef))) return gpry_fail(ctx, -1, "set_gates: bad support vectors");
    if (n_sv > ctx->gate_sv_cap) {
        if (ctx->gate_sv) GPRY_TRY(dev_free(ctx, ctx->gate_sv));
        if (ctx->gate_coef) GPRY_TRY(dev_free(ctx, ctx->gate_coef));
        ctx->gate_sv = ctx->gate_coef = nullptr; ctx->gate_sv_cap = 0;
        const int64_t cap = round_up(n_sv, 256);
        GPRY_TRY(dev_alloc(ctx, &ctx->gate_sv, cap * GPRY_MAX_DIM));
        GPRY_TRY(dev_alloc(ctx, &ctx->gate_coef, cap));
        ctx->gate_sv_cap = cap;
    }
    if (!ctx->gate_trust) GPRY_TRY(dev_alloc(ctx, &ctx->gate_trust, 2 * GPRY_MAX_DIM));
    if (n_sv > 0) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->gate_sv, sv, sizeof(double) * n_sv * ctx->d, hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(ctx->gate_coef, coef, sizeof(double) * n_sv, hipMemcpyHostToDevice, ctx->stream));
    }
    if (trust_bounds)
        HIP_TRY(ctx, hipMemcpyAsync(ctx->gate_trust, trust_bounds, sizeof(double) * 2 * ctx->d, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->gate_nsv = n_sv; ctx->gate_gamma = gamma; ctx->gate_intercept = intercept;
    ctx->gate_positive_finite = positive_is_finite; ctx->gate_has_trust = trust_bounds != nullptr;
    ctx->gates_on = (n_sv > 0 || trust_bounds != nullptr);
    return 0;
}

int gpry_sweep_logexp(gpry_ctx* ctx, const double* X, int64_t M, const uint8_t* mask, double zeta,
                      double baseline, double sigma_n, double* y_all, double* sigma_all, double* acq_all,
                      int64_t* n_nan) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_sweep_logexp: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (M <= 0) return gpry_fail(ctx, -1, "sweep: M must be > 0");
    // a pool that comes from the host goes up chunk by chunk underneath the sweep itself (run_sweep); "sweep_upload" = 0:
    // in one piece in front of it (the comparator)
    const bool piped = X != nullptr && ctx->opt_sweep_upload && ctx->stream2 != nullptr;
    GPRY_TRY(upload_candidates(ctx, piped ? nullptr : X, M, mask, piped));
    bool have_mask = mask != nullptr;
    if (ctx->gates_on) {
        // the SVM / trust-region verdicts are computed here, on top of the caller's bits
        if (!have_mask) HIP_TRY(ctx, hipMemsetAsync(ctx->dmask, 0, (size_t)M, ctx->stream));
        if (!piped) {
            StageScope s(ctx, "gates");
            GPRY_TRY(launch_gates(ctx, ctx->dXc, M, ctx->dmask));
        }
        have_mask = true;
    }
    struct UploadScope {        // (cleared on every way out: a later sweep of the resident pool must not upload again)
        gpry_ctx* c; bool done = false;
        ~UploadScope() {
            c->up_X = nullptr; c->up_gates = 0;
            // a sweep that did not complete leaves no resident pool behind: with the chunked upload part of dXc would be
            // stale, and a later call with X == NULL must not pass the size check; the side stream is drained as well
            if (!done) { c->sw_M = 0; if (c->stream2) (void)hipStreamSynchronize(c->stream2); (void)hipStreamSynchronize(c->stream); }
        }
    } upload_scope{ctx};
    if (piped) { ctx->up_X = X; ctx->up_gates = ctx->gates_on ? 1 : 0; }
    GPRY_TRY(run_sweep(ctx, M, have_mask, true, true, zeta, baseline, sigma_n));
    if (!ctx->dsel) GPRY_TRY(dev_alloc(ctx, &ctx->dsel, 64));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dsel, 0, 8, ctx->stream));
    hipLaunchKernelGGL(count_nan_kernel, dim3(1024), dim3(256), 0, ctx->stream, ctx->dacq_all, M, ctx->dsel);
    unsigned long long nn = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&nn, ctx->dsel, 8, hipMemcpyDeviceToHost, ctx->stream));
    if (y_all) HIP_TRY(ctx, hipMemcpyAsync(y_all, ctx->dy_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    if (sigma_all) HIP_TRY(ctx, hipMemcpyAsync(sigma_all, ctx->dsig_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    if (acq_all) HIP_TRY(ctx, hipMemcpyAsync(acq_all, ctx->dacq_all, sizeof(double) * M, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    upload_scope.done = true;
    if (n_nan) *n_nan = (int64_t)nn;
    return 0;
}

}  // extern "C"

// ------------------------------------------------------------------------------------
// shortlist selection: exact radix select on the 96-bit composite key
// (order-preserving image of acq, candidate index), 12 passes of 8 bits.
struct SelState { unsigned long long hi; unsigned int lo; unsigned int pad; unsigned long long k_rem; unsigned long long count_ge; };

__device__ __forceinline__ unsigned long long acq_key(double a) {
    unsigned long long b = (unsigned long long)__double_as_longlong(a);
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}

__global__ void make_keys_kernel(const double* __restrict__ acq, int64_t M, unsigned long long* __restrict__ keys) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < M) keys[i] = acq_key(acq[i]);
}
__global__ void exclude_keys_kernel(unsigned long long* keys, const int64_t* excl, int64_t n, int64_t M) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && excl[i] >= 0 && excl[i] < M) keys[excl[i]] = 0ull;   // below key(-inf)
}

// digit of pass p (0 = most significant byte of the acq key ... 7; 8..11 = index bytes)
__device__ __forceinline__ unsigned digit_of(unsigned long long key, unsigned int idx, int pass) {
    return pass < 8 ? (unsigned)((key >> (56 - 8 * pass)) & 0xFF) : (unsigned)((idx >> (24 - 8 * (pass - 8))) & 0xFF);
}
__device__ __forceinline__ bool prefix_match(unsigned long long key, unsigned int idx, const SelState& s, int pass) {
    if (pass == 0) return true;
    if (pass <= 8) {
        int sh = 64 - 8 * pass;
        return sh >= 64 ? true : ((key >> sh) == (s.hi >> sh));
    }
    if (key != s.hi) return false;
    int sh = 32 - 8 * (pass - 8);
    return (idx >> sh) == (s.lo >> sh);
}

__global__ __launch_bounds__(256) void select_hist_kernel(const unsigned long long* __restrict__ keys, int64_t M,
                                                          const SelState* __restrict__ st, int pass,
                                                          unsigned int* __restrict__ hist) {
    __shared__ unsigned int h[256];
    h[threadIdx.x] = 0;
    __syncthreads();
    SelState s = *st;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < M; i += stride) {
        unsigned long long k = keys[i];
        if (k == 0ull) continue;   // excluded
        if (prefix_match(k, (unsigned)i, s, pass)) atomicAdd(&h[digit_of(k, (unsigned)i, pass)], 1u);
    }
    __syncthreads();
    if (h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], h[threadIdx.x]);
}
__global__ void select_scan_kernel(unsigned int* hist, SelState* st, int pass) {
    if (threadIdx.x != 0) return;
    SelState s = *st;
    unsigned long long k = s.k_rem, acc = 0;
    int dsel = 0;
    for (int dgt = 255; dgt >= 0; dgt--) {
        unsigned long long c = hist[dgt];
        if (acc + c >= k) { dsel = dgt; break; }
        acc += c;
    }
    s.k_rem = k - acc;
    if (pass < 8) s.hi |= ((unsigned long long)dsel) << (56 - 8 * pass);
    else s.lo |= ((unsigned int)dsel) << (24 - 8 * (pass - 8));
    *st = s;
    for (int dgt = 0; dgt < 256; dgt++) hist[dgt] = 0;
}
// emit every candidate whose composite key >= threshold; track the best one below it
__global__ void select_emit_kernel(const unsigned long long* __restrict__ keys, int64_t M, const SelState* __restrict__ st,
                                   const double* __restrict__ acq, const double* __restrict__ y,
                                   const double* __restrict__ sig, gpry_cand* __restrict__ out, int64_t cap,
                                   unsigned long long* counters /*[0]=n_out, [1]=max key below*/) {
    SelState s = *st;
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    unsigned long long best_below = 0ull;
    for (; i < M; i += stride) {
        unsigned long long k = keys[i];
        if (k == 0ull) continue;
        bool ge = (k > s.hi) || (k == s.hi && (unsigned)i >= s.lo);
        if (ge) {
            unsigned long long pos = atomicAdd(&counters[0], 1ull);
            if ((int64_t)pos < cap) { gpry_cand c; c.acq = acq[i]; c.y = y[i]; c.sigma = sig[i]; c.idx = i; out[pos] = c; }
        } else if (k > best_below) best_below = k;
    }
    for (int off = 32; off >= 1; off >>= 1) {
        unsigned long long o = __shfl_xor(best_below, off);
        if (o > best_below) best_below = o;
    }
    if ((threadIdx.x & 63) == 0 && best_below) atomicMax(&counters[1], best_below);
}

// all sweep results as shortlist records (small pools: selected on the host)
__global__ void cand_records_kernel(const double* __restrict__ acq, const double* __restrict__ y, const double* __restrict__ sig,
                                    int64_t M, gpry_cand* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M) return;
    gpry_cand c;
    c.acq = acq[i]; c.y = y[i]; c.sigma = sig ? sig[i] : 0.0; c.idx = i;
    out[i] = c;
}

static double key_to_acq(unsigned long long k) {
    unsigned long long b = (k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k;
    double a; memcpy(&a, &b, 8); return a;
}

extern "C" int gpry_sweep_topk(gpry_ctx* ctx, int64_t Kp, const int64_t* exclude, int64_t n_exclude,
                               gpry_cand* top, int64_t* n_out, double* bound) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_sweep_topk: ctx is NULL");
    if (!top || !n_out || !bound) return gpry_fail(ctx, -1, "topk: top, n_out and bound must not be NULL");
    if (n_exclude > 0 && !exclude) return gpry_fail(ctx, -1, "topk: n_exclude > 0 but exclude is NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    const int64_t M = ctx->sw_M;
    if (M <= 0 || !ctx->dacq_all) return gpry_fail(ctx, -1, "topk: no sweep results resident");
    if (M > 0xFFFFFFFFll) return gpry_fail(ctx, -1, "topk: M too large");
    StageScope scope(ctx, "topk");
    hipStream_t st = ctx->stream;
    if (M <= ctx->opt_topk_host) {
        // Small pools (the first iterations of a run: a few thousand candidates): the radix select is 28 dependent
        // launches (0.22 ms whatever M is); one kernel writes all M records into the pinned, device-mapped staging
        // buffer and the host selects -- same total order (acq desc, idx desc; NaN first), same bound.
        GPRY_TRY(ensure_pinned(ctx, (int64_t)sizeof(gpry_cand) * M));
        gpry_cand* hrec = static_cast<gpry_cand*>(ctx->hpin);
        hipLaunchKernelGGL(cand_records_kernel, dim3((unsigned)((M + 255) / 256)), dim3(256), 0, st, ctx->dacq_all,
                           ctx->dy_all, ctx->dsig_all, M, static_cast<gpry_cand*>(ctx->hpin_dev));
        HIP_TRY(ctx, hipGetLastError());
        HIP_TRY(ctx, hipStreamSynchronize(st));
        std::vector<int64_t> ex;
        for (int64_t e = 0; e < n_exclude; e++) if (exclude[e] >= 0 && exclude[e] < M) ex.push_back(exclude[e]);
        std::sort(ex.begin(), ex.end());
        ex.erase(std::unique(ex.begin(), ex.end()), ex.end());
        std::vector<gpry_cand> v;
        v.reserve((size_t)M);
        size_t xi = 0;
        for (int64_t i = 0; i < M; i++) {
            if (xi < ex.size() && ex[xi] == i) { xi++; continue; }
            v.push_back(hrec[i]);
        }
        auto before = [](const gpry_cand& a, const gpry_cand& b) {
            unsigned long long ka, kb; double x = a.acq, y = b.acq;
            memcpy(&ka, &x, 8); memcpy(&kb, &y, 8);
            ka = (ka >> 63) ? ~ka : (ka | 0x8000000000000000ull);
            kb = (kb >> 63) ? ~kb : (kb | 0x8000000000000000ull);
            if (ka != kb) return ka > kb;
            return a.idx > b.idx;
        };
        // the device path counts the exclusions as given (n_valid = M - n_exclude)
        int64_t n_valid = M - (n_exclude > 0 ? n_exclude : 0);
        if (n_valid < 0) n_valid = 0;
        if (n_valid > (int64_t)v.size()) n_valid = (int64_t)v.size();
        const int64_t K = Kp < n_valid ? Kp : n_valid;
        *n_out = 0; *bound = -INFINITY;
        if (K <= 0) return 0;
        const int64_t take = std::min<int64_t>(K + 1, (int64_t)v.size());
        std::partial_sort(v.begin(), v.begin() + take, v.end(), before);
        for (int64_t k = 0; k < K; k++) top[k] = v[(size_t)k];
        *n_out = K;
        if ((int64_t)v.size() > K) *bound = v[(size_t)K].acq;
        return 0;
    }
    if (M > ctx->keys_cap) {
        if (ctx->dkeys) GPRY_TRY(dev_free(ctx, ctx->dkeys));
        GPRY_TRY(dev_alloc(ctx, &ctx->dkeys, round_up(M, 1024)));
        ctx->keys_cap = round_up(M, 1024);
    }
    if (!ctx->dhist) { GPRY_TRY(dev_alloc(ctx, &ctx->dhist, 256)); }
    if (!ctx->dsel) GPRY_TRY(dev_alloc(ctx, &ctx->dsel, 64));
    int64_t n_valid = M - (n_exclude > 0 ? n_exclude : 0);
    if (n_valid < 0) n_valid = 0;
    int64_t K = Kp < n_valid ? Kp : n_valid;
    if (K > ctx->cand_cap) {
        if (ctx->dcand) GPRY_TRY(dev_free(ctx, ctx->dcand));
        GPRY_TRY(dev_alloc(ctx, &ctx->dcand, round_up(K, 1024)));
        ctx->cand_cap = round_up(K, 1024);
    }
    unsigned nb = (unsigned)((M + 255) / 256);
    hipLaunchKernelGGL(make_keys_kernel, dim3(nb), dim3(256), 0, st, ctx->dacq_all, M, ctx->dkeys);
    TmpBuf<int64_t> bex;
    int64_t* dex = nullptr;
    if (n_exclude > 0) {
        GPRY_TRY(bex.alloc(ctx, n_exclude));
        dex = bex.p;
        HIP_TRY(ctx, hipMemcpyAsync(dex, exclude, sizeof(int64_t) * n_exclude, hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(exclude_keys_kernel, dim3((unsigned)((n_exclude + 255) / 256)), dim3(256), 0, st,
                           ctx->dkeys, dex, n_exclude, M);
    }
    *n_out = 0; *bound = -INFINITY;
    if (K <= 0) {
        HIP_TRY(ctx, hipStreamSynchronize(st));
        return 0;
    }
    SelState s0; memset(&s0, 0, sizeof(s0)); s0.k_rem = (unsigned long long)K;
    SelState* dst = reinterpret_cast<SelState*>(ctx->dsel);          // 32 bytes
    unsigned long long* dcnt = ctx->dsel + 4;                         // 2 counters after the state
    HIP_TRY(ctx, hipMemcpyAsync(dst, &s0, sizeof(s0), hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(dcnt, 0, 16, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dhist, 0, 256 * sizeof(unsigned int), st));
    unsigned nbs = nb < 2048 ? nb : 2048;
    for (int pass = 0; pass < 12; pass++) {
        hipLaunchKernelGGL(select_hist_kernel, dim3(nbs), dim3(256), 0, st, ctx->dkeys, M, dst, pass, ctx->dhist);
        hipLaunchKernelGGL(select_scan_kernel, dim3(1), dim3(64), 0, st, ctx->dhist, dst, pass);
    }
    hipLaunchKernelGGL(select_emit_kernel, dim3(nbs), dim3(256), 0, st, ctx->dkeys, M, dst, ctx->dacq_all,
                       ctx->dy_all, ctx->dsig_all, ctx->dcand, K, dcnt);
    HIP_TRY(ctx, hipGetLastError());
    unsigned long long cnt[2] = {0, 0};
    HIP_TRY(ctx, hipMemcpyAsync(cnt, dcnt, 16, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if ((int64_t)cnt[0] != K)
        return gpry_fail(ctx, -4, "topk: selected %llu candidates, expected %lld", cnt[0], (long long)K);
    HIP_TRY(ctx, hipMemcpy(top, ctx->dcand, sizeof(gpry_cand) * K, hipMemcpyDeviceToHost));
    // total order (acq desc, idx desc); NaN first as np.argsort(acq)[::-1] would put it
    std::sort(top, top + K, [](const gpry_cand& a, const gpry_cand& b) {
        unsigned long long ka, kb; double x = a.acq, y = b.acq;
        memcpy(&ka, &x, 8); memcpy(&kb, &y, 8);
        ka = (ka >> 63) ? ~ka : (ka | 0x8000000000000000ull);
        kb = (kb >> 63) ? ~kb : (kb | 0x8000000000000000ull);
        if (ka != kb) return ka > kb;
        return a.idx > b.idx;
    });
    *n_out = K;
    *bound = cnt[1] ? key_to_acq(cnt[1]) : -INFINITY;
    return 0;
}

// ------------------------------------------------------------------------------------
// Kriging-believer session
__global__ void scale_rows_kernel(const double* __restrict__ X, int64_t m, int d, int dpad, int has_aff,
                                  const double* __restrict__ lo, const double* __restrict__ span,
                                  const double* __restrict__ ls, double* __restrict__ out) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= m * dpad) return;
    int64_t i = idx / dpad; int k = (int)(idx - i * dpad);
    double v = 0.0;
    if (k < d) {
        v = X[i * d + k];
        if (has_aff) v = (v - lo[k]) / span[k];
        v = v / ls[k];
    }
    out[idx] = v;
}
// one wave per row x: out_dot[x] = U[x].U[p] ; out_k[x] = C k(|xs_x - xs_p|)
__device__ __forceinline__ double kb_corr(int kid, double r2) {
    switch (kid) {
        case GPRY_RBF: return exp(-0.5 * r2);
        case GPRY_MATERN12: return exp(-sqrt(r2));
        case GPRY_MATERN32: { double t = sqrt(r2) * 1.7320508075688772; return (1.0 + t) * exp(-t); }
        default: { double t = sqrt(r2) * 2.23606797749979; return (1.0 + t + t * t / 3.0) * exp(-t); }
    }
}
__global__ __launch_bounds__(256) void kb_gram_kernel(const double* __restrict__ U, int64_t ldu, int64_t n, int64_t p,
                                                      int64_t Np, const double* __restrict__ Xkb, int dpad,
                                                      int kid, double C, double* __restrict__ out_dot,
                                                      double* __restrict__ out_k) {
    const int lane = threadIdx.x & 63;
    const int64_t x = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (x >= n) return;
    const double* ux = U + x * ldu; const double* up = U + p * ldu;
    double s = 0.0;
    for (int64_t i = lane; i < Np; i += 64) s = fma(ux[i], up[i], s);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) {
        out_dot[x] = s;
        double r2 = 0.0;
        for (int k = 0; k < dpad; k++) { double df = Xkb[x * dpad + k] - Xkb[p * dpad + k]; r2 = fma(df, df, r2); }
        out_k[x] = C * kb_corr(kid, r2);
    }
}
__global__ __launch_bounds__(256) void kb_var0_kernel(const double* __restrict__ U, int64_t ldu, int64_t first,
                                                      int64_t m, int64_t Np, double C, double* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t x = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (x >= m) return;
    const double* ux = U + (first + x) * ldu;
    double s = 0.0;
    for (int64_t i = lane; i < Np; i += 64) s = fma(ux[i], ux[i], s);
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) out[x] = C - s;
}

extern "C" {

int gpry_kb_reset(gpry_ctx* ctx) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_kb_reset: ctx is NULL");
    GPRY_TRY(require_model(ctx, true));
    ctx->kb_n = 0;
    return 0;
}

int gpry_kb_register(gpry_ctx* ctx, const double* X, int64_t m, int64_t* first, double* var0) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_kb_register: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    GPRY_TRY(require_model(ctx, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (m <= 0) { if (first) *first = ctx->kb_n; return 0; }
    StageScope kb_scope(ctx, "kb_register");
    hipStream_t st = ctx->stream;
    const int64_t Np = ctx->Np;
    int64_t need = ctx->kb_n + round_up(m, 128);
    // rows registered earlier were written with the padded size of that time as their stride: every path that
    // changes Np (set_train, append_rows) ends the session, so a stride change can only meet an empty session
    if (ctx->kb_n > 0 && ctx->kb_ld < Np) return gpry_fail(ctx, -4, "kb_register: session rows have stride %lld, model needs %lld", (long long)ctx->kb_ld, (long long)Np);
    if (need > ctx->kb_cap || Np > ctx->kb_ld) {
        // (the row buffer used to be sized kb_cap x Np_at_allocation only: after the training set grew within
        // ctx->cap -- Np 1280 -> 1408, same buffers -- a session of ~900+ rows ran over its end)
        int64_t cap = std::max<int64_t>(need, std::max<int64_t>(1024, need > ctx->kb_cap ? 2 * ctx->kb_cap : ctx->kb_cap));
        double *nU = nullptr, *nX = nullptr, *nO = nullptr;
        GPRY_TRY(dev_alloc(ctx, &nU, cap * Np));
        GPRY_TRY(dev_alloc(ctx, &nX, cap * ctx->dpad));
        GPRY_TRY(dev_alloc(ctx, &nO, 2 * cap));
        if (ctx->kb_n > 0) {
            HIP_TRY(ctx, hipMemcpyAsync(nU, ctx->dU, sizeof(double) * ctx->kb_n * Np, hipMemcpyDeviceToDevice, st));
            HIP_TRY(ctx, hipMemcpyAsync(nX, ctx->dXkb, sizeof(double) * ctx->kb_n * ctx->dpad, hipMemcpyDeviceToDevice, st));
        }
        HIP_TRY(ctx, hipStreamSynchronize(st));
        if (ctx->dU) GPRY_TRY(dev_free(ctx, ctx->dU));
        if (ctx->dXkb) GPRY_TRY(dev_free(ctx, ctx->dXkb));
        if (ctx->dkbout) GPRY_TRY(dev_free(ctx, ctx->dkbout));
        ctx->dU = nU; ctx->dXkb = nX; ctx->dkbout = nO; ctx->kb_cap = cap; ctx->kb_ld = Np;
    }
    const int64_t mp = round_up(m, 128);
    // stage the candidates, build their cross-kernel panel, then U^T = K*  V^T
    TmpBuf<double> bX, bpar;
    GPRY_TRY(bX.alloc(ctx, m * ctx->d));
    GPRY_TRY(bpar.alloc(ctx, 3 * GPRY_MAX_DIM));
    double *dX = bX.p, *dpar = bpar.p;
    HIP_TRY(ctx, hipMemcpyAsync(dX, X, sizeof(double) * m * ctx->d, hipMemcpyHostToDevice, st));
    double hpar[3 * GPRY_MAX_DIM];
    for (int k = 0; k < GPRY_MAX_DIM; k++) {
        hpar[k] = k < ctx->d ? ctx->tf.x_lo[k] : 0.0;
        hpar[GPRY_MAX_DIM + k] = k < ctx->d ? ctx->tf.x_span[k] : 1.0;
        hpar[2 * GPRY_MAX_DIM + k] = k < ctx->d ? exp(ctx->theta[1 + k]) : 1.0;
    }
    HIP_TRY(ctx, hipMemcpyAsync(dpar, hpar, sizeof(hpar), hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((m * ctx->dpad + 255) / 256)), dim3(256), 0, st, dX, m,
                       ctx->d, ctx->dpad, ctx->tf.has_x_affine, dpar, dpar + GPRY_MAX_DIM, dpar + 2 * GPRY_MAX_DIM,
                       ctx->dXkb + ctx->kb_n * ctx->dpad);
    if (Np * mp > ctx->kst_cap) {
        if (ctx->dKst) GPRY_TRY(dev_free(ctx, ctx->dKst));
        ctx->dKst = nullptr; ctx->kst_cap = 0;
        GPRY_TRY(dev_alloc(ctx, &ctx->dKst, Np * mp));
        ctx->kst_cap = Np * mp;
    }
    int64_t saveM = ctx->sw_M; ctx->sw_M = m;
    int rc = mp <= 512 ? launch_cross_build_small(ctx, dX, 0, mp, mp, ctx->dKst, nullptr, 1)
                       : launch_cross_build(ctx, dX, 0, mp, mp, ctx->dKst, nullptr, 1);
    ctx->sw_M = saveM;
    if (rc) return rc;
    GemmArgs g = {};
    g.A = ctx->dKst; g.lda = mp;          // A(x, k) = Kst[k][x]
    g.B = ctx->dV; g.ldb = Np;            // B(k, i) = V[i][k]
    g.C = ctx->dU + ctx->kb_n * Np; g.ldc = Np;
    g.M = (int)mp; g.N = (int)Np; g.K = (int)Np;
    g.kmode = KM_B_UPPER; g.lower_only = 0; g.tile_map = TM_ROWMAJOR;
    {   // a shortlist is a few hundred points: 2 x Np/128 tiles, the longest walks all Np/16 slabs alone (0.8 ms
        // at Np = 4096): split the k-ranges so that the launch fills the GPU (as gpry_predict does, section 4.5)
        // The factor depends on Np ONLY: the conditioned variances must not depend on how many points were
        // registered together (a sharded pool merges a different number of shortlist points than one context
        // selects, and the proposals are compared bit for bit, tests/test_group_gpu.py).
        int ns = 1;
        while (ns < 8 && Np / (ns * 2) >= 256) ns *= 2;
        if (ns > 1) {
            GPRY_TRY(gemm_split_scratch(ctx, ns, mp * Np, &g.split_buf));
            g.nsplit = ns; g.split_stride = mp * Np;
        }
    }
    GPRY_TRY(gemm_f64_launch(ctx, g, true, true, EPI_STORE));
    if (var0) {
        hipLaunchKernelGGL(kb_var0_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, ctx->dU, Np, ctx->kb_n, m,
                           Np, exp(ctx->theta[0]), ctx->dkbout);
        HIP_TRY(ctx, hipMemcpyAsync(var0, ctx->dkbout, sizeof(double) * m, hipMemcpyDeviceToHost, st));
    }
    HIP_TRY(ctx, hipStreamSynchronize(st));
    if (first) *first = ctx->kb_n;
    ctx->kb_n += m;
    return 0;
}

int gpry_kb_gram(gpry_ctx* ctx, int64_t p, double* G, double* kvec, int64_t* n) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_kb_gram: ctx is NULL");
    GPRY_TRY(require_model(ctx, true));
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (p < 0 || p >= ctx->kb_n) return gpry_fail(ctx, -1, "kb_gram: index %lld out of range [0, %lld)", (long long)p, (long long)ctx->kb_n);
    StageScope kb_scope(ctx, "kb_gram");
    hipStream_t st = ctx->stream;
    int64_t nn = ctx->kb_n;
    // the two result vectors go straight into the pinned, device-mapped staging buffer (no copy-out operations:
    // the ranking calls this once per accepted point, 15 times per cycle)
    GPRY_TRY(ensure_pinned(ctx, (int64_t)sizeof(double) * 2 * nn));
    double* hres = static_cast<double*>(ctx->hpin);
    double* dres = static_cast<double*>(ctx->hpin_dev);
    hipLaunchKernelGGL(kb_gram_kernel, dim3((unsigned)((nn + 3) / 4)), dim3(256), 0, st, ctx->dU, ctx->Np, nn, p,
                       ctx->Np, ctx->dXkb, ctx->dpad, ctx->kernel_id, exp(ctx->theta[0]), dres, dres + nn);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipStreamSynchronize(st));
    memcpy(G, hres, sizeof(double) * nn);
    memcpy(kvec, hres + nn, sizeof(double) * nn);
    if (n) *n = nn;
    return 0;
}

}  // extern "C"

extern "C" int gpry_debug_gemm(gpry_ctx* ctx, const double* A, const double* B, double* C, int M, int N,
                               int K, int a_trans, int b_trans, int epi, int kmode, int lower_only,
                               int tile_map) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_debug_gemm: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    if (!A || !B || !C) return gpry_fail(ctx, -1, "debug_gemm: A, B and C must not be NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    if (M % 64 || N % 64 || K % 64) return gpry_fail(ctx, -1, "debug_gemm: dims must be multiples of 64");
    int64_t crow = (epi == EPI_SUMSQ) ? (M + 127) / 128 : M;
    TmpBuf<double> bA, bB, bC;
    GPRY_TRY(bA.alloc(ctx, (int64_t)M * K));
    GPRY_TRY(bB.alloc(ctx, (int64_t)K * N));
    GPRY_TRY(bC.alloc(ctx, crow * N));
    double *dA = bA.p, *dB = bB.p, *dC = bC.p;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(dA, A, sizeof(double) * M * K, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dB, B, sizeof(double) * K * N, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(dC, C, sizeof(double) * crow * N, hipMemcpyHostToDevice, st));
    GemmArgs g = {};
    g.A = dA; g.lda = a_trans ? M : K;
    g.B = dB; g.ldb = b_trans ? K : N;
    g.C = dC; g.ldc = N;
    g.M = M; g.N = N; g.K = K; g.kmode = kmode; g.lower_only = lower_only; g.tile_map = tile_map & 0xff;
    g.small64 = (tile_map >> 28) & 1;             // test hook: bit 28 = eligible for the 64 x 64 tiles of gemm_small.hip
    const int nsplit = (tile_map >> 8) & 15;      // test hook: bits 8..11 of tile_map = split-K factor
    if (nsplit > 1) {
        GPRY_TRY(gemm_split_scratch(ctx, nsplit, (int64_t)crow * N, &g.split_buf));
        g.nsplit = nsplit; g.split_stride = (int64_t)crow * N;
    }
    const int seg = (tile_map >> 16) & 0xfff;     // test hook: bits 16..27 = stream-K segment length in slab pairs
    if (seg > 0) {
        if (M % 128 || N % 128 || K % 32) return gpry_fail(ctx, -1, "debug_gemm: stream-K needs M, N multiples of 128, K of 32");
        GemmPartsPlan pl;
        std::vector<GemmShape> sh = {{M, N, K}};
        GPRY_TRY(gemm_parts_plan_build(ctx, kmode, lower_only, sh, seg, &pl));
        int rc = gemm_dma_parts_launch(ctx, g, a_trans != 0, b_trans != 0, epi, pl, (int64_t)crow * N);
        if (rc == 0 && hipStreamSynchronize(st) != hipSuccess) rc = gpry_fail(ctx, -2, "debug_gemm: stream-K launch failed");
        gemm_parts_plan_free(&pl);
        if (rc) return rc;
    } else {
        StageScope s(ctx, "debug_gemm");
        GPRY_TRY(gemm_f64_launch(ctx, g, a_trans != 0, b_trans != 0, epi));
    }
    HIP_TRY(ctx, hipMemcpyAsync(C, dC, sizeof(double) * crow * N, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

// model-free entry to the acquisition epilogue of the sweep (the F5 edge vectors go through it)
extern "C" int gpry_debug_logexp(gpry_ctx* ctx, const double* mu, const double* sigma, int64_t n, double zeta,
                                 double baseline, double sigma_n, double* acq) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_debug_logexp: ctx is NULL");
    if (n <= 0) return 0;
    if (!mu || !sigma || !acq) return gpry_fail(ctx, -1, "debug_logexp: mu, sigma and acq must not be NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    TmpBuf<double> bd;
    GPRY_TRY(bd.alloc(ctx, 3 * n));
    double* d = bd.p;
    hipStream_t st = ctx->stream;
    HIP_TRY(ctx, hipMemcpyAsync(d, mu, sizeof(double) * n, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(d + n, sigma, sizeof(double) * n, hipMemcpyHostToDevice, st));
    hipLaunchKernelGGL(logexp_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, d, d + n, n, zeta, baseline,
                       sigma_n, d + 2 * n);
    HIP_TRY(ctx, hipGetLastError());
    HIP_TRY(ctx, hipMemcpyAsync(acq, d + 2 * n, sizeof(double) * n, hipMemcpyDeviceToHost, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    return 0;
}

