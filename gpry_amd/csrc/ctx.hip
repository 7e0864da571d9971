// Context, memory, options, timing and micro-benchmarks for libgpry_hip.so.
#include "common.h"
#include <string>
#include <stdarg.h>
#include <dlfcn.h>

// per thread: concurrent optimiser restarts drive distinct contexts from distinct host threads
thread_local char g_last_error[1024] = {0};
void trtri_plan_free(gpry_ctx* ctx);
void overlap_plan_free(gpry_ctx* ctx);

int gpry_fail(gpry_ctx* ctx, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    if (ctx) { strncpy(ctx->err, g_last_error, sizeof(ctx->err) - 1); }
    return code;
}

template <typename T>
int dev_alloc(gpry_ctx* ctx, T** p, int64_t count) {
    *p = nullptr;
    if (count <= 0) count = 1;
    HIP_TRY(ctx, hipMalloc((void**)p, sizeof(T) * (size_t)count));
    return 0;
}
template int dev_alloc<double>(gpry_ctx*, double**, int64_t);
template int dev_alloc<int>(gpry_ctx*, int**, int64_t);
template int dev_alloc<uint8_t>(gpry_ctx*, uint8_t**, int64_t);
template int dev_alloc<unsigned int>(gpry_ctx*, unsigned int**, int64_t);
template int dev_alloc<unsigned long long>(gpry_ctx*, unsigned long long**, int64_t);
template int dev_alloc<gpry_cand>(gpry_ctx*, gpry_cand**, int64_t);
template int dev_alloc<int64_t>(gpry_ctx*, int64_t**, int64_t);

int dev_free(gpry_ctx* ctx, void* p) {
    if (p) HIP_TRY(ctx, hipFree(p));
    return 0;
}

// roctx ranges around the stages (the reference's Timer / TimerCounter, gpry/progress.py:243-284, as marker ranges that
// rocprofv3 --marker-trace shows next to the kernels): off unless GPRY_HIP_ROCTX=1; the roctx library is looked up at run time
// (dlopen), the library has no link-time dependency on it.  A range covers the host-side queueing of a stage's launches.
namespace {
struct Roctx {
    bool tried = false, on = false;
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
} g_roctx;
bool roctx_on() {
    if (!g_roctx.tried) {
        g_roctx.tried = true;
        const char* e = getenv("GPRY_HIP_ROCTX");
        if (e && atoi(e) != 0) {
            // rocprofv3 (rocprofiler-sdk) records the ranges of ITS roctx library; the roctracer one (libroctx64) is the fallback
            // for the older tools -- same entry points
            void* lib = nullptr;
            for (const char* name : {"librocprofiler-sdk-roctx.so", "librocprofiler-sdk-roctx.so.1", "/opt/rocm/lib/librocprofiler-sdk-roctx.so",
                                     "libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"}) {
                lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
                if (lib) break;
            }
            if (lib) {
                g_roctx.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
                g_roctx.pop = (int (*)())dlsym(lib, "roctxRangePop");
                g_roctx.on = g_roctx.push && g_roctx.pop;
            }
            if (!g_roctx.on) fprintf(stderr, "gpry: GPRY_HIP_ROCTX=1 but neither librocprofiler-sdk-roctx.so nor libroctx64.so could be loaded: no marker ranges\n");
        }
    }
    return g_roctx.on;
}
}  // namespace

StageScope::StageScope(gpry_ctx* c, const char* n, hipStream_t stream) : ctx(c), name(n) {
    st = stream ? stream : ctx->stream;
    if (roctx_on()) { g_roctx.push(name); marked = true; }
    if (!ctx->opt_timing) return;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
    (void)hipEventRecord(e0, st);
}
StageScope::~StageScope() {
    if (marked) g_roctx.pop();
    if (!e0) return;
    (void)hipEventRecord(e1, st);
    auto& tm = ctx->timers[name];
    tm.pending.push_back({e0, e1});
    if (tm.pending.size() >= 1024) timers_collect(ctx);   // bounded even if nobody ever reads the timers
}
void timers_collect(gpry_ctx* ctx) {
    for (auto& kv : ctx->timers) {
        for (auto& pr : kv.second.pending) {
            float ms = 0.f;
            if (hipEventSynchronize(pr.second) == hipSuccess &&
                hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                kv.second.total_ms += ms;
                kv.second.count += 1;
            }
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        kv.second.pending.clear();
    }
}

int ensure_pinned(gpry_ctx* ctx, int64_t bytes) {
    if (bytes <= ctx->hpin_cap) return 0;
    if (ctx->hpin) HIP_TRY(ctx, hipHostFree(ctx->hpin));
    ctx->hpin = nullptr; ctx->hpin_dev = nullptr; ctx->hpin_cap = 0;
    // mapped: the small-batch predict kernels read and write this buffer directly
    HIP_TRY(ctx, hipHostMalloc(&ctx->hpin, (size_t)bytes, hipHostMallocMapped | hipHostMallocPortable));
    HIP_TRY(ctx, hipHostGetDevicePointer(&ctx->hpin_dev, ctx->hpin, 0));
    // the stamped units of the single-launch objective (lml_small.hip) live in this buffer: a recycled allocation must
    // not hold words that look like a current stamp
    memset(ctx->hpin, 0, (size_t)bytes);
    ctx->hpin_cap = bytes;
    return 0;
}

int ensure_capacity(gpry_ctx* ctx, int64_t N, int d) {
    int64_t Np = round_up(N > 0 ? N : 1, GPRY_TILE);
    int dpad = (d + 1) & ~1;
    if (Np > ctx->cap || dpad > ctx->dp_cap) {
        int64_t cap = ctx->cap;
        if (Np > cap) {
            cap = Np;
            // grow geometrically: the training set gains d points per iteration
            if (ctx->cap > 0 && cap < ctx->cap + ctx->cap / 8) cap = round_up(ctx->cap + ctx->cap / 8, GPRY_TILE);
        }
        int dp = dpad > ctx->dp_cap ? dpad : ctx->dp_cap;
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        double** bufs[] = {&ctx->dX, &ctx->dXs, &ctx->dy, &ctx->dnoise, &ctx->dA, &ctx->dV, &ctx->dW,
                           &ctx->dW2, &ctx->dW3, &ctx->dalpha_, &ctx->dvec};
        const int64_t counts[] = {cap * dp, cap * dp, cap, cap, cap * cap, cap * cap, cap * cap,
                                  cap * cap, cap * cap, cap, 8 * cap + 4096};
        // nothing is valid while the buffers are being replaced: a failed allocation leaves an empty
        // context (cap = 0) that the next set_train allocates afresh, never stale or NULL pointers
        // behind the old capacity
        ctx->cap = 0; ctx->dp_cap = 0; ctx->N = 0; ctx->Np = 0;
        ctx->factor_valid = false; ctx->lml_cache = false;
        for (auto b : bufs) { if (*b) { (void)hipFree(*b); *b = nullptr; } }
        for (size_t i = 0; i < sizeof(bufs) / sizeof(bufs[0]); i++) {
            int rc = dev_alloc(ctx, bufs[i], counts[i]);
            if (rc) {
                for (auto b : bufs) { if (*b) { (void)hipFree(*b); *b = nullptr; } }
                return rc;
            }
        }
        ctx->cap = cap;
        ctx->dp_cap = dp;
        ctx->kst_cap = 0; if (ctx->dKst) { GPRY_TRY(dev_free(ctx, ctx->dKst)); ctx->dKst = nullptr; }
        ctx->kb_cap = 0; ctx->kb_n = 0; ctx->kb_ld = 0;
        if (ctx->dU) { GPRY_TRY(dev_free(ctx, ctx->dU)); ctx->dU = nullptr; }
        if (ctx->dXkb) { GPRY_TRY(dev_free(ctx, ctx->dXkb)); ctx->dXkb = nullptr; }
    }
    // the scratch arena of the batched objective is sized for the padded size it was last used with: a model of less than
    // half that footprint gives the memory back (the next gpry_lml_batch allocates what it needs).  A model whose size
    // moves back and forth across a padding boundary keeps its arena: freeing and allocating tens of GB per fit cost more
    // than the memory is worth.
    if (ctx->barena && 2 * Np * Np < ctx->Np * ctx->Np) {
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipFree(ctx->barena);
        ctx->barena = nullptr; ctx->barena_cap = 0;
    }
    ctx->N = N; ctx->Np = Np; ctx->d = d; ctx->dpad = dpad;
    return 0;
}

extern "C" {

int gpry_version(void) { return 100; }

int gpry_device_count(int* n) {
    int c = 0;
    hipError_t e = hipGetDeviceCount(&c);
    if (e != hipSuccess) { *n = 0; return gpry_fail(nullptr, -2, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    *n = c;
    return 0;
}

int gpry_device_info(int device, char* name, int name_len, int64_t* hbm_bytes, int* n_cu,
                     int* clock_khz, char* arch, int arch_len) {
    hipDeviceProp_t p;
    hipError_t e = hipGetDeviceProperties(&p, device);
    if (e != hipSuccess) return gpry_fail(nullptr, -2, "hipGetDeviceProperties: %s", hipGetErrorString(e));
    if (name && name_len > 0) { strncpy(name, p.name, name_len - 1); name[name_len - 1] = 0; }
    if (arch && arch_len > 0) { strncpy(arch, p.gcnArchName, arch_len - 1); arch[arch_len - 1] = 0; }
    if (hbm_bytes) *hbm_bytes = (int64_t)p.totalGlobalMem;
    if (n_cu) *n_cu = p.multiProcessorCount;
    if (clock_khz) *clock_khz = p.clockRate;
    return 0;
}

int gpry_ctx_create(int device, gpry_ctx** out) {
    if (!out) return gpry_fail(nullptr, -1, "gpry_ctx_create: out is NULL");
    *out = nullptr;
    hipError_t e = hipSetDevice(device);
    if (e != hipSuccess) return gpry_fail(nullptr, -2, "hipSetDevice(%d): %s", device, hipGetErrorString(e));
    gpry_ctx* ctx = new gpry_ctx();
    ctx->device = device;
    memset(&ctx->tf, 0, sizeof(ctx->tf));
    ctx->tf.y_std = 1.0; ctx->tf.clip_hi = INFINITY;
    e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return gpry_fail(nullptr, -2, "hipStreamCreate: %s", hipGetErrorString(e)); }
    e = hipStreamCreateWithFlags(&ctx->stream2, hipStreamNonBlocking);
    if (e != hipSuccess) { delete ctx; return gpry_fail(nullptr, -2, "hipStreamCreate: %s", hipGetErrorString(e)); }
    if (hipMalloc((void**)&ctx->dinfo, 16 * sizeof(int)) != hipSuccess) { delete ctx; return gpry_fail(nullptr, -2, "hipMalloc info"); }
    (void)hipMemset(ctx->dinfo, 0, 16 * sizeof(int));
    // GPRY_HIP_OPTIONS="key=value,key=value": options for every context of the process (A/B runs of
    // unmodified callers; an unknown key or a malformed entry fails the creation loudly)
    if (const char* env = getenv("GPRY_HIP_OPTIONS")) {
        std::string all(env);
        size_t pos = 0;
        while (pos < all.size()) {
            size_t end = all.find(',', pos);
            if (end == std::string::npos) end = all.size();
            const std::string item = all.substr(pos, end - pos);
            pos = end + 1;
            if (item.empty()) continue;
            const size_t eq = item.find('=');
            char* tail = nullptr;
            const long long v = eq == std::string::npos ? 0 : strtoll(item.c_str() + eq + 1, &tail, 10);
            if (eq == std::string::npos || eq == 0 || tail == item.c_str() + eq + 1 || *tail != '\0' ||
                gpry_ctx_set_option(ctx, item.substr(0, eq).c_str(), (int64_t)v) != 0) {
                (void)gpry_ctx_destroy(ctx);
                return gpry_fail(nullptr, -1, "GPRY_HIP_OPTIONS: bad entry '%s'", item.c_str());
            }
        }
    }
    *out = ctx;
    return 0;
}

int gpry_ctx_destroy(gpry_ctx* ctx) {
    if (!ctx) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    if (ctx->stream2) (void)hipStreamSynchronize(ctx->stream2);
    timers_collect(ctx);
    serve_free(ctx);
    trtri_plan_free(ctx);
    trtri_pipe_free(ctx);
    overlap_plan_free(ctx);
    void* bufs[] = {ctx->dX, ctx->dXs, ctx->dy, ctx->dnoise, ctx->dA, ctx->dV, ctx->dW, ctx->dW2, ctx->dW3,
                    ctx->dalpha_, ctx->dvec, ctx->dinfo, ctx->dparams, ctx->dXc, ctx->dmask, ctx->dy_all,
                    ctx->dsig_all, ctx->dacq_all, ctx->dKst, ctx->dpart, ctx->dkeys, ctx->dhist,
                    ctx->dcand, ctx->dsel, ctx->dU, ctx->dXkb, ctx->dkbout, ctx->pr.dXc, ctx->pr.dmask,
                    ctx->pr.dy, ctx->pr.dsig, ctx->pr.dacq, ctx->dG,
                    ctx->gate_sv, ctx->gate_coef, ctx->gate_trust, ctx->dsplit, ctx->dbord, ctx->barena, ctx->dXcs, ctx->dYcs};
    for (void* b : bufs) if (b) (void)hipFree(b);
    if (ctx->hpin) (void)hipHostFree(ctx->hpin);
    if (ctx->hbres) (void)hipHostFree(ctx->hbres);
    (void)hipStreamDestroy(ctx->stream);
    if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
    for (hipEvent_t ev : ctx->ev_pool) (void)hipEventDestroy(ev);
    for (hipStream_t st : ctx->tp_streams) (void)hipStreamDestroy(st);
    for (hipEvent_t ev : ctx->tp_events) (void)hipEventDestroy(ev);
    delete ctx;
    return 0;
}

const char* gpry_last_error(gpry_ctx* ctx) { return ctx ? ctx->err : g_last_error; }

int gpry_ctx_sync(gpry_ctx* ctx) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_ctx_sync: ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- options: ONE table (include/gpry_hip.h documents every key) -------------------------------------------------
namespace {
struct OptionSpec {
    const char* key;
    int64_t lo, hi;                         // accepted range
    int64_t (*get)(const gpry_ctx*);
    void (*set)(gpry_ctx*, int64_t);
};
#define OPT_INT(KEY, FIELD, LO, HI, AFTER)                                                         \
    {KEY, LO, HI, [](const gpry_ctx* c) -> int64_t { return (int64_t)c->FIELD; },                  \
     [](gpry_ctx* c, int64_t v) { c->FIELD = (decltype(c->FIELD))v; AFTER; }}
const int64_t BIG = (int64_t)1 << 40;
const OptionSpec OPTIONS[] = {
    OPT_INT("timing", opt_timing, 0, 1, (void)0),
    OPT_INT("chol", opt_chol, 0, 1, c->lml_cache = false),
    OPT_INT("chol_overlap", opt_chol_overlap, 0, 1, c->lml_cache = false),      // (the comparator takes the recursive inverse: other rounding)
    OPT_INT("factor_pipeline", opt_factor_pipeline, 0, 1, (void)0),
    OPT_INT("factor_pipeline_min", opt_factor_pipeline_min, 0, BIG, (void)0),
    OPT_INT("gemm_dma", opt_gemm_dma, 0, 1, (void)0),
    OPT_INT("gemm_streamk", opt_gemm_streamk, 0, BIG, trtri_plan_free(c)),       // the plan holds the stream-K parts
    OPT_INT("gemm_small", opt_gemm_small, 0, BIG, c->lml_cache = false),
    OPT_INT("cross_mfma", opt_cross_mfma, 0, 1, (void)0),
    OPT_INT("cross_hybrid", opt_cross_hybrid, 0, 1, (void)0),
    OPT_INT("sweep_chunk", opt_sweep_chunk, 0, BIG, c->opt_sweep_chunk = round_up(c->opt_sweep_chunk, 1024)),
    OPT_INT("topk_host", opt_topk_host, 0, BIG, (void)0),
    OPT_INT("lml_small", opt_lml_small, 0, 1, c->lml_cache = false),
    OPT_INT("lml_cache", opt_lml_cache, 0, 1, c->lml_cache = false),
    OPT_INT("lml_batch", opt_lml_batch, 0, BIG, (void)0),
    OPT_INT("lml_batch_mb", opt_lml_batch_mb, 1, BIG, (void)0),
    OPT_INT("lml_schedule", opt_lml_schedule, 0, 1, (void)0),
    OPT_INT("lml_streams", opt_lml_streams, 1, 8, (void)0),
    OPT_INT("chol_tp_segments", opt_chol_tp_segments, 0, 1, c->lml_cache = false),
    OPT_INT("panel_debug", opt_panel_debug, 0, 255, c->lml_cache = false),
    OPT_INT("tp_left", opt_tp_left, 0, 1, (void)0),
    OPT_INT("tp_block", opt_tp_block, 128, BIG, c->opt_tp_block = round_up(c->opt_tp_block, 128); overlap_plan_free(c)),
    OPT_INT("tp_tail", opt_tp_tail, 128, BIG, c->opt_tp_tail = round_up(c->opt_tp_tail, 128); overlap_plan_free(c)),
    OPT_INT("predict_small", opt_predict_small, 0, BIG, (void)0),
    OPT_INT("predict_split", opt_predict_split, 0, 1, (void)0),
    OPT_INT("sweep_upload", opt_sweep_upload, 0, 1, (void)0),
    OPT_INT("sweep_overlap", opt_sweep_overlap, 0, 1, (void)0),
    OPT_INT("chol_stacked", opt_chol_stacked, 0, BIG, c->lml_cache = false),
    OPT_INT("chol_stacked_dense", opt_chol_stacked_dense, 0, 1, c->lml_cache = false),
    OPT_INT("predict_gates", opt_predict_gates, 0, 1, (void)0),
    OPT_INT("predict_serve", opt_predict_serve, 0, 1, (void)0),
    OPT_INT("serve_idle_us", opt_serve_idle_us, 10, 1000000, (void)0),
};
#undef OPT_INT
const OptionSpec* find_option(const char* key) {
    for (const OptionSpec& o : OPTIONS) if (!strcmp(o.key, key)) return &o;
    return nullptr;
}
}  // namespace

int gpry_ctx_set_option(gpry_ctx* ctx, const char* key, int64_t value) {
    if (!ctx || !key) return gpry_fail(ctx, -1, "gpry_ctx_set_option: NULL argument");
    GPRY_TRY(serve_stop(ctx));
    const OptionSpec* o = find_option(key);
    if (!o) return gpry_fail(ctx, -1, "unknown option '%s'", key);
    if (value < o->lo || value > o->hi)
        return gpry_fail(ctx, -1, "%s must be in %lld..%lld (got %lld)", key, (long long)o->lo, (long long)o->hi, (long long)value);
    o->set(ctx, value);
    return 0;
}

int gpry_ctx_get_option(gpry_ctx* ctx, const char* key, int64_t* value) {
    if (!ctx || !key || !value) return gpry_fail(ctx, -1, "gpry_ctx_get_option: NULL argument");
    const OptionSpec* o = find_option(key);
    if (!o) return gpry_fail(ctx, -1, "unknown option '%s'", key);
    *value = o->get(ctx);
    return 0;
}

int gpry_set_train(gpry_ctx* ctx, const double* X_, const double* y_, const double* alpha,
                   int64_t N, int d) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_set_train: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    if (N <= 0 || d <= 0) return gpry_fail(ctx, -1, "set_train: need N > 0 and d > 0 (got %lld, %d)", (long long)N, d);
    if (d > GPRY_MAX_DIM) return gpry_fail(ctx, -1, "set_train: d=%d > GPRY_MAX_DIM=%d is not supported by this build", d, GPRY_MAX_DIM);
    if (!X_ || !y_ || !alpha) return gpry_fail(ctx, -1, "set_train: X_, y_ and alpha must not be NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    GPRY_TRY(ensure_capacity(ctx, N, d));
    hipStream_t st = ctx->stream;
    for (int k = 0; k < d; k++) {       // centre of the MFMA panel build (kernel_build.hip: cross_build_mfma_kernel)
        double s = 0.0, lo = X_[k], hi = X_[k];
        for (int64_t i = 0; i < N; i++) {                           // row order: gpry_append_rows continues the sums
            const double v = X_[i * d + k];
            s += v;
            lo = v < lo ? v : lo; hi = v > hi ? v : hi;
        }
        ctx->xsum[k] = s;
        ctx->xcenter[k] = s / (double)N;
        ctx->xlo[k] = lo; ctx->xhi[k] = hi;
    }
    {
        double nm = alpha[0];
        for (int64_t i = 1; i < N; i++) nm = alpha[i] < nm ? alpha[i] : nm;
        ctx->noise_min = nm;
    }
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dX, X_, sizeof(double) * N * d, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dy, 0, sizeof(double) * ctx->Np, st));
    HIP_TRY(ctx, hipMemsetAsync(ctx->dnoise, 0, sizeof(double) * ctx->Np, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dy, y_, sizeof(double) * N, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->dnoise, alpha, sizeof(double) * N, hipMemcpyHostToDevice, st));
    HIP_TRY(ctx, hipStreamSynchronize(st));
    ctx->factor_valid = false;
    ctx->lml_cache = false;
    ctx->kb_n = 0;
    return 0;
}

int gpry_set_theta(gpry_ctx* ctx, int kernel_id, const double* theta) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_set_theta: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    if (kernel_id < 0 || kernel_id > 3) return gpry_fail(ctx, -1, "unknown kernel id %d", kernel_id);
    if (ctx->d <= 0) return gpry_fail(ctx, -1, "set_theta before set_train");
    if (!theta) return gpry_fail(ctx, -1, "set_theta: theta is NULL");
    ctx->kernel_id = kernel_id;
    for (int k = 0; k <= ctx->d; k++) {
        if (!isfinite(theta[k])) return gpry_fail(ctx, -1, "theta[%d] is not finite", k);
        ctx->theta[k] = theta[k];
    }
    ctx->have_theta = true;
    ctx->factor_valid = false;
    ctx->kb_n = 0;
    return 0;
}

int gpry_set_affine(gpry_ctx* ctx, const gpry_affine* tf) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_set_affine: ctx is NULL");
    GPRY_TRY(serve_stop(ctx));
    if (!tf) return gpry_fail(ctx, -1, "set_affine: null");
    ctx->tf = *tf;
    return 0;
}

int gpry_timing_reset(gpry_ctx* ctx) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_timing_reset: ctx is NULL");
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    timers_collect(ctx);
    for (auto& kv : ctx->timers) { kv.second.total_ms = 0.0; kv.second.count = 0; }
    ctx->opt_timing = 1;      // whoever resets the timers wants them; a production loop never pays for events
    return 0;
}

int gpry_timing_get(gpry_ctx* ctx, const char* name, double* total_ms, int64_t* count) {
    if (!ctx) return gpry_fail(nullptr, -1, "gpry_timing_get: ctx is NULL");
    // (not a timer: how often a batched objective halved its chunk after an out-of-memory answer since the context was created)
    if (name && !strcmp(name, "lml_batch_shrinks")) { if (total_ms) *total_ms = 0.0; if (count) *count = ctx->batch_shrinks; return 0; }
    timers_collect(ctx);
    auto it = ctx->timers.find(name);
    if (it == ctx->timers.end()) { if (total_ms) *total_ms = 0.0; if (count) *count = 0; return -1; }
    if (total_ms) *total_ms = it->second.total_ms;
    if (count) *count = it->second.count;
    return 0;
}

}  // extern "C"

// ---- micro-benchmarks ----------------------------------------------------------------
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(double* out, int iters) {
    v4d acc[8];
#pragma unroll
    for (int i = 0; i < 8; i++) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 8; i++) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 8; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;   // keep the chain live without a store in practice
}
// same loop with the accumulators pinned to architectural VGPRs (as the GEMM kernels use them)
__global__ __launch_bounds__(256) void mfma_f64_peak_vgpr_kernel(double* out, int iters) {
    v4d acc[16];
#pragma unroll
    for (int i = 0; i < 16; i++) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < 16; i++)
            asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
    }
    double s = 0.0;
#pragma unroll
    for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678) out[0] = s;
}
// Do the matrix pipe and the vector ALU run FP64 work side by side?  Waves 0-3 of a workgroup (one per
// SIMD) loop over v_mfma_f64_16x16x4_f64, waves 4-7 (their SIMD neighbours) over v_fma_f64 with one
// wave-uniform (SGPR) factor -- the operand pattern of a contraction whose V entries come through the
// scalar cache.  mode bit 0: matrix waves run, bit 1: vector waves run.
__global__ __launch_bounds__(512) void mixed_f64_peak_kernel(double* out, int iters_mfma, int iters_valu,
                                                             int mode, double u) {
    const int wave = threadIdx.x >> 6;
    double s = 0.0;
    if (wave < 4) {
        if (!(mode & 1)) return;
        v4d acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = (v4d){0.0, 0.0, 0.0, 0.0};
        double a = 1.0 + 1e-9 * threadIdx.x, b = 1.0 - 1e-9 * threadIdx.x;
        for (int it = 0; it < iters_mfma; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        if (!(mode & 2)) return;
        double acc[16];
#pragma unroll
        for (int i = 0; i < 16; i++) acc[i] = 1e-9 * i;
        double b = 1.0 - 1e-9 * threadIdx.x;
        for (int it = 0; it < iters_valu; it++) {
#pragma unroll
            for (int i = 0; i < 16; i++)
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(acc[i]) : "s"(u), "v"(b));
        }
#pragma unroll
        for (int i = 0; i < 16; i++) s += acc[i];
    }
    if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void stream_copy_kernel(const double2* __restrict__ src,
                                                          double2* __restrict__ dst, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void xcc_probe_kernel(int* out) {
    // HW_REG_XCC_ID (id 20), bits 3:0 = XCC id
    int x = __builtin_amdgcn_s_getreg(20 | (0 << 6) | ((4 - 1) << 11));
    // keep the block alive for a while so that the whole grid is not drained by a few CUs
    for (int i = 0; i < 20; i++) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
__global__ __launch_bounds__(256, 2) void lds_alloc_probe_kernel(unsigned* out) {
    // HW_REG_LDS_ALLOC (id 6), whole register; 70 KB of LDS so that two workgroups share a CU
    __shared__ double pad[8960];
    unsigned x = __builtin_amdgcn_s_getreg(6 | (0 << 6) | ((32 - 1) << 11));
    pad[threadIdx.x] = (double)x;
    for (int i = 0; i < 20; i++) __builtin_amdgcn_s_sleep(64);
    if (threadIdx.x == 0) out[blockIdx.x] = x + (pad[5] < 0.0 ? 1u : 0u);
}
__global__ __launch_bounds__(256) void stream_fill_kernel(double2* __restrict__ dst, int64_t n, double v) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (; i < n; i += stride) dst[i] = make_double2(v, v + 1.0);
}

namespace {
struct EventPair {      // destroyed on every return path of the micro-benchmarks
    hipEvent_t a = nullptr, b = nullptr;
    ~EventPair() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
};
}  // namespace

extern "C" int gpry_microbench(gpry_ctx* ctx, int kind, int64_t bytes, double* value) {
    if (!ctx || !value) return gpry_fail(ctx, -1, "gpry_microbench: ctx and value must not be NULL");
    HIP_TRY(ctx, hipSetDevice(ctx->device));
    EventPair ev;
    HIP_TRY(ctx, hipEventCreate(&ev.a));
    HIP_TRY(ctx, hipEventCreate(&ev.b));
    hipEvent_t e0 = ev.a, e1 = ev.b;
    float ms = 0.f;
    if (kind == 0) {
        double* out = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&out, 64));
        // `bytes` selects the number of 4-wave workgroups per CU (waves per SIMD), default 1
        const int wps = (bytes >= 1 && bytes <= 8) ? (int)bytes : 1;
        const int iters = 40000 / wps, nblk = 256 * wps;
        hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(nblk), dim3(256), 0, ctx->stream, out, 10);
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(nblk), dim3(256), 0, ctx->stream, out, iters);
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)nblk * 4.0 * iters * 8.0 * 2048.0;  // 16*16*4*2 per MFMA
        *value = flops / (ms * 1e-3) / 1e12;
        (void)hipFree(out);
    } else if (kind == 2) {
        double* out = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&out, 64));
        const int wps = (bytes >= 1 && bytes <= 8) ? (int)bytes : 1;
        const int iters = 20000 / wps, nblk = 256 * wps;
        hipLaunchKernelGGL(mfma_f64_peak_vgpr_kernel, dim3(nblk), dim3(256), 0, ctx->stream, out, 10);
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(mfma_f64_peak_vgpr_kernel, dim3(nblk), dim3(256), 0, ctx->stream, out, iters);
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        double flops = (double)nblk * 4.0 * iters * 16.0 * 2048.0;
        *value = flops / (ms * 1e-3) / 1e12;
        (void)hipFree(out);
    } else if (kind == 1) {
        if (bytes < (1 << 20)) bytes = 1 << 20;
        int64_t n = bytes / 16;
        double2 *src = nullptr, *dst = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&src, (size_t)n * 16));
        HIP_TRY(ctx, hipMalloc((void**)&dst, (size_t)n * 16));
        HIP_TRY(ctx, hipMemsetAsync(src, 1, (size_t)n * 16, ctx->stream));
        // four workgroups per CU: the best copy shape of tools/r04/hbm_ceiling.hip (5.8 TB/s on 2 GB, 6.9 inside the Infinity Cache)
        hipLaunchKernelGGL(stream_copy_kernel, dim3(1024), dim3(256), 0, ctx->stream, src, dst, n);
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        const int reps = 5;
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL(stream_copy_kernel, dim3(1024), dim3(256), 0, ctx->stream, src, dst, n);
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        *value = (double)reps * 2.0 * (double)n * 16.0 / (ms * 1e-3) / 1e9;
        (void)hipFree(src); (void)hipFree(dst);
    } else if (kind == 7) {   // matrix + vector FP64 side by side; bytes = mode (1 matrix, 2 vector, 3 both) + 16 * workgroups per CU
        double* out = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&out, 64));
        const int mode = (int)(bytes & 3), wpc = (bytes >> 4) >= 1 ? (int)(bytes >> 4) : 1;
        const int im = 4000, iv = 16 * im, nblk = 256 * wpc;
        hipLaunchKernelGGL(mixed_f64_peak_kernel, dim3(nblk), dim3(512), 0, ctx->stream, out, 10, 160, mode, 0.999999);
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        hipLaunchKernelGGL(mixed_f64_peak_kernel, dim3(nblk), dim3(512), 0, ctx->stream, out, im, iv, mode, 0.999999);
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        double flops = 0.0;
        if (mode & 1) flops += (double)nblk * 4.0 * im * 16.0 * 2048.0;
        if (mode & 2) flops += (double)nblk * 4.0 * iv * 16.0 * 128.0;
        *value = flops / (ms * 1e-3) / 1e12;
        fprintf(stderr, "mixed_f64 mode %d, %d wg/CU: %.3f ms, %.2f TFLOP/s\n", mode, wpc, ms, *value);
        (void)hipFree(out);
    } else if (kind == 6) {   // covariance build of the resident model, `bytes` launches back to back: us per launch
        if (ctx->N <= 0 || !ctx->have_theta) return gpry_fail(ctx, -1, "microbench 6 needs set_train + set_theta");
        const int reps = bytes >= 1 ? (int)bytes : 20;
        ctx->lml_cache = false;   // dW is the scratch target
        GPRY_TRY(launch_scale_train(ctx));
        GPRY_TRY(launch_kernel_train(ctx, ctx->dW, 1));
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        for (int r = 0; r < reps; r++) GPRY_TRY(launch_kernel_train(ctx, ctx->dW, 1));
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        *value = (double)ms * 1e3 / reps;
    } else if (kind == 8) {   // covariance build + potrf (serial chain) of the resident model, us per repetition; bytes & 1: replayed
                              // as a captured hipGraph instead of launched kernel by kernel; bytes >> 1: repetitions (default 20)
        if (ctx->N <= 0 || !ctx->have_theta) return gpry_fail(ctx, -1, "microbench 8 needs set_train + set_theta");
        const bool graph = bytes & 1;
        const int reps = (bytes >> 1) >= 1 ? (int)(bytes >> 1) : 20;
        ctx->lml_cache = false;
        ctx->factor_valid = false;
        GPRY_TRY(launch_scale_train(ctx));
        auto once = [&]() -> int {
            GPRY_TRY(launch_kernel_train(ctx, ctx->dW, 1));
            ctx->info_cleared = false;
            GPRY_TRY(potrf_lower_overlap(ctx, ctx->dW, ctx->Np));
            return 0;
        };
        GPRY_TRY(once());                                   // plans, allocations
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        if (!graph) {
            HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
            for (int r = 0; r < reps; r++) GPRY_TRY(once());
            HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        } else {
            hipGraph_t g = nullptr; hipGraphExec_t ge = nullptr;
            HIP_TRY(ctx, hipStreamBeginCapture(ctx->stream, hipStreamCaptureModeRelaxed));
            const int rc = once();
            const hipError_t ee = hipStreamEndCapture(ctx->stream, &g);
            if (rc) return rc;
            HIP_TRY(ctx, ee);
            HIP_TRY(ctx, hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
            HIP_TRY(ctx, hipGraphLaunch(ge, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
            for (int r = 0; r < reps; r++) HIP_TRY(ctx, hipGraphLaunch(ge, ctx->stream));
            HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
            HIP_TRY(ctx, hipEventSynchronize(e1));
            (void)hipGraphExecDestroy(ge); (void)hipGraphDestroy(g);
        }
        HIP_TRY(ctx, hipEventSynchronize(e1));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        int inf[2] = {0, 0};
        HIP_TRY(ctx, hipMemcpy(inf, ctx->dinfo, sizeof(inf), hipMemcpyDeviceToHost));
        if (inf[0] != 0) return gpry_fail(ctx, -1, "microbench 8: the matrix is not positive definite (info %d)", inf[0]);
        *value = (double)ms * 1e3 / reps;
    } else if (kind == 4) {   // dispatch probe: fraction of blocks b that run on XCC (b % 8 + c) % 8
        const int nblk = 4096;
        int* d = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&d, nblk * sizeof(int)));
        hipLaunchKernelGGL(xcc_probe_kernel, dim3(nblk), dim3(256), 0, ctx->stream, d);
        std::vector<int> h(nblk);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(h.data(), d, nblk * sizeof(int), hipMemcpyDeviceToHost));
        int best = 0;
        for (int c = 0; c < 8; c++) {
            int ok = 0;
            for (int b = 0; b < nblk; b++) ok += ((h[b] & 15) == (b + c) % 8);
            if (ok > best) best = ok;
        }
        *value = (double)best / nblk;
        (void)hipFree(d);
    } else if (kind == 5) {   // HW_REG_LDS_ALLOC of co-resident 70-KB workgroups: histogram to stderr
        const int nblk = 2048;
        unsigned* d = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&d, nblk * sizeof(unsigned)));
        hipLaunchKernelGGL(lds_alloc_probe_kernel, dim3(nblk), dim3(256), 0, ctx->stream, d);
        std::vector<unsigned> h(nblk);
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        HIP_TRY(ctx, hipMemcpy(h.data(), d, nblk * sizeof(unsigned), hipMemcpyDeviceToHost));
        std::map<unsigned, int> hist;
        for (unsigned v : h) hist[v]++;
        for (auto& kv : hist) fprintf(stderr, "LDS_ALLOC 0x%08x : %d blocks\n", kv.first, kv.second);
        *value = (double)hist.size();
        (void)hipFree(d);
    } else if (kind == 3) {   // pure streaming write (the kernel build is write-dominated)
        if (bytes < (1 << 20)) bytes = 1 << 20;
        int64_t n = bytes / 16;
        double2* dst = nullptr;
        HIP_TRY(ctx, hipMalloc((void**)&dst, (size_t)n * 16));
        // one workgroup per CU, grid-stride: 6.5 TB/s on a 2-GB buffer where 8 workgroups per CU reach 4.6-4.7
        // (tools/r04/hbm_ceiling.hip, profiles/r04_hbm_ceiling.md)
        hipLaunchKernelGGL(stream_fill_kernel, dim3(256), dim3(256), 0, ctx->stream, dst, n, 1.0);
        HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
        const int reps = 5;
        for (int r = 0; r < reps; r++)
            hipLaunchKernelGGL(stream_fill_kernel, dim3(256), dim3(256), 0, ctx->stream, dst, n, 2.0 + r);
        HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
        HIP_TRY(ctx, hipEventSynchronize(e1));
        HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
        *value = (double)reps * (double)n * 16.0 / (ms * 1e-3) / 1e9;
        (void)hipFree(dst);
    } else {
        return gpry_fail(ctx, -1, "microbench: unknown kind %d", kind);
    }
    return 0;
}
