// Resident predict kernel ("server"): posterior mean of 1..8 points per request WITHOUT a kernel launch per call.
//
// Why.  Nested samplers and MCMC evaluate the surrogate one point at a time -- gpry/gp_acquisition.py:766-771
// (PolyChord: `gpr.predict(np.atleast_2d(X), return_std=False, validate=False)[0]`), :784-793 (UltraNest), gpry/mc.py:387-391
// -- 1e4..1e6 calls per acquisition step; SURVEY.md section 8(f)2 names "a persistent-kernel or pinned-buffer path".
// One launch + hipStreamSynchronize costs 15-20 us on this platform whatever the kernel does (round 2: 21 us per
// call through Python = the CPU's own 21 us at N = 64), the arithmetic of one point at N <= 1024 well under 2 us.
//
// How.  After the first small mean-only gpry_predict the context keeps ONE kernel running on a stream of its own:
// G = clamp(N / 1024, 1, 8) workgroups, each owning the slice of training rows the one-launch kernel gives
// blockIdx.y (same mean_slice(), same order of every sum: the bits do not depend on which path served a call).
// Host -> device: a mailbox in pinned, coherent, mapped host memory, made of 16-byte units {payload, stamp}; unit 0
// is the header {command, number of points}, units 1.. hold the coordinates.  The host writes payload before stamp
// (x86 stores are ordered); the leader workgroup polls units 0..d with ONE 16-byte system-scope load per lane -- a
// 16-byte unit inside a cache line is read atomically, so a unit whose stamp is the expected sequence number carries
// the payload of that request -- and republishes the request to its followers through device memory (release /
// acquire at agent scope).  Device -> host: per (slice, point) ONE 16-byte store {partial sum, stamp}: a single PCIe
// write inside one cache line.  (Round 2 learned that inbound writes to DIFFERENT lines are not ordered here,
// profiles/HISTORY.md section 4.5: a status word written "last" overtook its data in 2 % of the calls.  No ordering between
// units is assumed anywhere in this protocol; every unit proves its own freshness.)  The host spins on the stamps,
// adds the slices in the fixed order of the one-launch path and applies y_std, y_mean, clip and mask.
//
// Leaving.  The kernel ends (a) on a QUIT request -- every other entry point of the library posts one and waits
// (serve_stop) before it touches the model, because the kernel holds theta, the affine maps and the buffer addresses
// of its launch; (b) after `serve_idle_us` (default 2000) without a request; (c) after SRV_MAX_POLLS polls whatever
// the clock says (a second, independent bound: a resident kernel must never outlive its host).  The leader decides,
// tells the followers through device memory, and writes {EXITED, generation} to the host, which relaunches on the
// next request.  A request that arrives while the leader is leaving is served by the next generation (the host sees
// EXITED without its results and launches; results of two generations are the same bits).
#include "kern_math.h"
#include <chrono>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

#define SRV_MAXM 8                       // points per request
#define SRV_REQ_UNITS (1 + SRV_MAXM * GPRY_MAX_DIM)
#define SRV_MAX_SLICES 8
#define SRV_CMD_PREDICT 1ull
#define SRV_CMD_QUIT 2ull
#define SRV_EXITED 0x45584954ull          // "EXIT"
#define SRV_MAX_POLLS 4000000             // ~ seconds of polling: hard bound besides the idle clock

struct SrvUnit { unsigned long long payload, stamp; };      // 16 bytes, 16-byte aligned

struct SrvArgs {
    const SrvUnit* req;        // host mailbox (device view): SRV_REQ_UNITS units
    SrvUnit* res;              // host results (device view): [slice][point]
    SrvUnit* state;            // host: {SRV_EXITED, generation} when the kernel has left
    unsigned long long* dseq;  // device: sequence number of the last request the leader has republished (followers poll it)
    unsigned long long* dhdr;  // device: header of that request (number of points)
    unsigned long long* dexit; // device: generation whose leader has decided to leave (followers poll it)
    double* dx;                // device: coordinates of that request (SRV_MAXM * GPRY_MAX_DIM)
    const double* Xs; const double* alpha_;
    unsigned long long seq0;   // last sequence number served before this launch
    unsigned long long gen;
    long long idle_ticks;      // wall_clock64 ticks (100 MHz) without a request before the kernel leaves
    int64_t rows_per_split;
    int cached;                // this launch keeps its slices of Xs / alpha_ in LDS
    // gates (gpry_set_gates) evaluated per point by the leader when the context applies them to gpry_predict
    const double* gate_sv; const double* gate_coef; const double* gate_trust;
    int64_t gate_nsv; double gate_gamma, gate_intercept;
    int gate_positive_finite, gate_has_trust, gates;
};

// GPRY_MASK_* bits of one point (raw coordinates x in LDS): the trust box on the raw coordinates, the SVM decision
// function sum_r coef_r exp(-gamma |x_ - sv_r|^2) + intercept on the transformed ones -- the per-pair arithmetic of
// gates_kernel (kernel_build.hip), the support vectors dealt out over the 256 threads and their terms added by the
// fixed LDS tree.  Valid in every thread.
__device__ __forceinline__ unsigned srv_gate_bits(const double* x, const SrvArgs& a, const KernParams& kp, const AffParams& ap,
                                                  double* red) {
    const int t = threadIdx.x;
    unsigned bits = 0;
    if (a.gate_has_trust) {
        for (int k = 0; k < kp.d; k++) {
            const double v = x[k];
            if (!(v >= a.gate_trust[2 * k] && v <= a.gate_trust[2 * k + 1])) bits |= GPRY_MASK_OUTSIDE_TRUST;
        }
    }
    if (a.gate_nsv > 0) {
        double part = 0.0;
        for (int64_t r = t; r < a.gate_nsv; r += 256) {
            double r2 = 0.0;
            for (int k = 0; k < kp.d; k++) {
                double v = x[k];
                if (kp.has_aff) v = (v - ap.lo[k]) / ap.span[k];
                const double df = v - a.gate_sv[r * kp.d + k];
                r2 = fma(df, df, r2);
            }
            part = fma(a.gate_coef[r], fast_exp_neg(a.gate_gamma * r2), part);
        }
        red[t] = part;
        __syncthreads();
        for (int s2 = 128; s2 >= 1; s2 >>= 1) {
            if (t < s2) red[t] += red[t + s2];
            __syncthreads();
        }
        const double dec = red[0] + a.gate_intercept;
        __syncthreads();
        const bool finite = a.gate_positive_finite ? dec > 0.0 : !(dec > 0.0);
        if (!finite) bits |= GPRY_MASK_CLASSIFIED_INF;
    }
    return bits;
}

#define SRV_CACHE_ROWS 1024
#define SRV_CACHE_DOUBLES 16384           // 128 KB of the 160 KB of a CU
typedef unsigned int srv_u4 __attribute__((ext_vector_type(4)));


// one 16-byte unit, system scope (the mailbox is fine-grained host memory: nothing may be served from a cache)
__device__ __forceinline__ SrvUnit srv_load_sys(const SrvUnit* p) {
    srv_u4 v;
    asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(v) : "v"(p) : "memory");
    SrvUnit u;
    u.payload = ((unsigned long long)v.y << 32) | v.x;
    u.stamp = ((unsigned long long)v.w << 32) | v.z;
    return u;
}
__device__ __forceinline__ void srv_store_sys(SrvUnit* p, unsigned long long payload, unsigned long long stamp) {
    srv_u4 v = {(unsigned)payload, (unsigned)(payload >> 32), (unsigned)stamp, (unsigned)(stamp >> 32)};
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" :: "v"(p), "v"(v) : "memory");
}

template <int DP, int KID>
__global__ __launch_bounds__(256) void predict_server_kernel(SrvArgs a, KernParams kp, AffParams ap) {
    // The slice of the model this workgroup owns stays in LDS for the life of the kernel when it fits (<= 1024 rows,
    // rows x dpad <= 16384 doubles): a request then touches no global memory but the mailbox.
    __shared__ __attribute__((aligned(16))) double xs_c[SRV_CACHE_DOUBLES];
    __shared__ double al_c[SRV_CACHE_ROWS];
    __shared__ double r2s[SRV_CACHE_ROWS];
    __shared__ double red[256];
    __shared__ double s_x[SRV_MAXM * GPRY_MAX_DIM];
    __shared__ unsigned long long s_hdr;
    const int t = threadIdx.x, g = blockIdx.x;
    const int64_t row_lo = (int64_t)g * a.rows_per_split;
    if (a.cached) {
        const int64_t nrows = a.rows_per_split;
        for (int64_t e = t; e < nrows * kp.dpad; e += 256) {
            const int64_t row = row_lo + e / kp.dpad;
            xs_c[e] = row < kp.N ? a.Xs[row_lo * kp.dpad + e] : 0.0;
        }
        for (int64_t r = t; r < nrows; r += 256) al_c[r] = row_lo + r < kp.N ? a.alpha_[row_lo + r] : 0.0;
        __syncthreads();
    }
    unsigned long long last = a.seq0;
    long long t_last = wall_clock64();
    int polls = 0;
    for (;;) {
        // ---- wait for request last + 1 (wave 0 polls, the other waves sleep at the barrier)
        if (t < 64) {
            unsigned long long hdr = 0;
            if (g == 0) {
                const int nfirst = 1 + kp.d;               // header + the coordinates of one point
                for (;;) {
                    SrvUnit u = {0ull, ~0ull};
                    if (t < nfirst) u = srv_load_sys(a.req + t);
                    const bool fresh = t >= nfirst || u.stamp == last + 1;
                    if (__all(fresh)) {
                        hdr = __shfl(u.payload, 0);
                        if ((hdr >> 56) == SRV_CMD_PREDICT && ((hdr & 0xff) == 0 || (hdr & 0xff) > SRV_MAXM)) hdr = SRV_CMD_QUIT << 56;   // malformed: leave
                        const int nd = (hdr >> 56) == SRV_CMD_PREDICT ? (int)(hdr & 0xff) * kp.d : 0;
                        if (t >= 1 && t <= kp.d) s_x[t - 1] = __longlong_as_double((long long)u.payload);
                        // further points of the request: their units may still be on their way
                        bool all = true;
                        for (int e = kp.d + t; e < nd; e += 64) {
                            const SrvUnit w = srv_load_sys(a.req + 1 + e);
                            if (w.stamp != last + 1) all = false;
                            s_x[e] = __longlong_as_double((long long)w.payload);
                        }
                        if (__all(all)) break;
                    }
                    polls++;
                    if (wall_clock64() - t_last > a.idle_ticks || polls > SRV_MAX_POLLS) { hdr = SRV_CMD_QUIT << 56; break; }
                }
                if (gridDim.x > 1) {
                    if ((hdr >> 56) == SRV_CMD_PREDICT) {
                        // republish for the followers: coordinates and header, then the sequence number (release)
                        for (int e = t; e < (int)(hdr & 0xff) * kp.d; e += 64) a.dx[e] = s_x[e];
                        __threadfence();
                        if (t == 0) {
                            a.dhdr[0] = hdr;
                            __hip_atomic_store(a.dseq, last + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                        }
                    } else if (t == 0) {
                        __hip_atomic_store(a.dexit, a.gen, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            } else {
                hdr = SRV_CMD_QUIT << 56;
                for (;;) {
                    unsigned long long sq = 0, ex = 0;
                    if (t == 0) {
                        sq = __hip_atomic_load(a.dseq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                        ex = __hip_atomic_load(a.dexit, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
                    }
                    sq = __shfl(sq, 0); ex = __shfl(ex, 0);
                    if (sq == last + 1) {
                        hdr = __hip_atomic_load(a.dhdr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        for (int e = t; e < (int)(hdr & 0xff) * kp.d; e += 64)
                            s_x[e] = __hip_atomic_load(a.dx + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    if (ex == a.gen || ++polls > 16 * SRV_MAX_POLLS) break;      // the leader has left (or is lost)
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            if (t == 0) s_hdr = hdr;
        }
        __syncthreads();
        const unsigned long long hdr = s_hdr;
        if ((hdr >> 56) != SRV_CMD_PREDICT) {
            // QUIT from the host, or the leader's own decision: the host learns which generation has left
            if (g == 0 && t == 0) srv_store_sys(a.state, SRV_EXITED, a.gen);
            return;
        }
        const int M = (int)(hdr & 0xff);
        const unsigned long long seq = last + 1;
        for (int m = 0; m < M; m++) {
            // (the one-launch kernel walks chunks of 4096 rows; a slice of <= 1024 rows is one chunk either way: same sums)
            const double v = a.cached
                ? mean_slice<DP, KID, SRV_CACHE_ROWS>(s_x + m * kp.d, xs_c, al_c, row_lo, a.rows_per_split, kp, ap, r2s, red, row_lo)
                : mean_slice<DP, KID>(s_x + m * kp.d, a.Xs, a.alpha_, row_lo, a.rows_per_split, kp, ap, xs_c, red);   // (xs_c: free, serves as the 4096-row r2 buffer)
            if (t == 0) srv_store_sys(a.res + g * SRV_MAXM + m, (unsigned long long)__double_as_longlong(v), seq);
            if (a.gates && g == 0) {
                const unsigned bits = srv_gate_bits(s_x + m * kp.d, a, kp, ap, red);
                if (t == 0) srv_store_sys(a.res + SRV_MAX_SLICES * SRV_MAXM + m, (unsigned long long)bits, seq);
            }
        }
        last = seq;
        t_last = wall_clock64();
        polls = 0;
        __syncthreads();
    }
}

// ---- host side -----------------------------------------------------------------------------------
struct SrvHost {
    char* h = nullptr; char* d = nullptr;          // pinned, coherent, mapped: host / device view
    unsigned long long* dflags = nullptr;          // device: [0] dseq, [1] dhdr, [2] dexit
    double* dx = nullptr;
    hipStream_t stream = nullptr;
    bool running = false;
    unsigned long long seq = 0, gen = 0;           // last sequence number posted / last generation launched
    int nsplit = 1;
    int gates = 0;                                 // the running generation evaluates the gates of its launch
    int64_t launches = 0, requests = 0;
    // layout of the host buffer
    static constexpr size_t REQ = 0, RES = 8192, STATE = 8192 + 2048, BYTES = 16384;
    volatile SrvUnit* req() { return reinterpret_cast<volatile SrvUnit*>(h + REQ); }
    volatile SrvUnit* res() { return reinterpret_cast<volatile SrvUnit*>(h + RES); }
    volatile SrvUnit* state() { return reinterpret_cast<volatile SrvUnit*>(h + STATE); }
    bool exited() { volatile SrvUnit* st = state(); const unsigned long long g = st->stamp; return g == gen && st->payload == SRV_EXITED; }
};

static inline void srv_cpu_relax() {
#if defined(__x86_64__)
    _mm_pause();
#endif
}

static int srv_get(gpry_ctx* ctx, SrvHost** out) {
    if (!ctx->srv) {
        SrvHost* s = new SrvHost();
        ctx->srv = s;
        static_assert(sizeof(SrvUnit) == 16, "unit size");
        static_assert(sizeof(SrvUnit) * SRV_REQ_UNITS <= SrvHost::RES, "mailbox");
        static_assert(sizeof(SrvUnit) * (SRV_MAX_SLICES + 1) * SRV_MAXM <= 2048, "results (+ one row of gate bits)");
        if (hipHostMalloc((void**)&s->h, SrvHost::BYTES, hipHostMallocMapped | hipHostMallocCoherent | hipHostMallocPortable) != hipSuccess) {
            s->h = nullptr;
            return gpry_fail(ctx, -2, "serve: hipHostMalloc failed");
        }
        memset(s->h, 0, SrvHost::BYTES);
        HIP_TRY(ctx, hipHostGetDevicePointer((void**)&s->d, s->h, 0));
        HIP_TRY(ctx, hipMalloc((void**)&s->dflags, 64));
        HIP_TRY(ctx, hipMemset(s->dflags, 0, 64));
        HIP_TRY(ctx, hipMalloc((void**)&s->dx, sizeof(double) * SRV_MAXM * GPRY_MAX_DIM));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    }
    *out = static_cast<SrvHost*>(ctx->srv);
    return (*out)->stream ? 0 : gpry_fail(ctx, -2, "serve: not initialised");
}

static void srv_post(SrvHost* s, unsigned long long cmd, const double* X, int M, int d) {
    volatile SrvUnit* rq = s->req();
    const unsigned long long seq = ++s->seq;
    // payload before stamp in every unit; the units in any order (each proves its own freshness)
    for (int e = 0; e < M * d; e++) {
        unsigned long long b; memcpy(&b, X + e, 8);
        rq[1 + e].payload = b;
    }
    rq[0].payload = (cmd << 56) | (unsigned long long)M;
    __atomic_thread_fence(__ATOMIC_RELEASE);      // compiler + store ordering (x86 stores are ordered anyway)
    // the leader polls the header and the units of ONE point (0 .. d) and wants all of them fresh: a request without
    // points (QUIT) stamps them too
    const int n = M * d > d ? M * d : d;
    for (int e = 0; e <= n; e++) rq[e].stamp = seq;
    __atomic_thread_fence(__ATOMIC_SEQ_CST);      // drain the store buffer: the request is on its way now
}

// generation gen + 1, expecting the request that has just been posted (sequence number s->seq)
static int srv_launch(gpry_ctx* ctx, SrvHost* s) {
    KernParams kp;
    kp.C = exp(ctx->theta[0]); kp.d = ctx->d; kp.dpad = ctx->dpad; kp.has_aff = ctx->tf.has_x_affine; kp.N = ctx->N;
    AffParams ap = make_ap(ctx, kp.has_aff);
    int nsplit = (int)(ctx->N / 1024);            // the slices of the one-launch path (api.hip: gpry_predict)
    if (nsplit < 1) nsplit = 1;
    if (nsplit > SRV_MAX_SLICES) nsplit = SRV_MAX_SLICES;
    s->nsplit = nsplit;
    SrvArgs a;
    a.req = reinterpret_cast<const SrvUnit*>(s->d + SrvHost::REQ);
    a.res = reinterpret_cast<SrvUnit*>(s->d + SrvHost::RES);
    a.state = reinterpret_cast<SrvUnit*>(s->d + SrvHost::STATE);
    a.dseq = s->dflags; a.dhdr = s->dflags + 1; a.dexit = s->dflags + 2; a.dx = s->dx;
    GPRY_TRY(ensure_pred_xs(ctx));
    a.Xs = ctx->dXs; a.alpha_ = ctx->dalpha_;
    a.gen = ++s->gen;
    a.seq0 = s->seq - 1;
    a.idle_ticks = (long long)ctx->opt_serve_idle_us * 100;       // wall_clock64: 100 MHz
    a.rows_per_split = round_up((ctx->N + nsplit - 1) / nsplit, 32);
    a.cached = a.rows_per_split <= SRV_CACHE_ROWS && a.rows_per_split * ctx->dpad <= SRV_CACHE_DOUBLES;
    a.gates = ctx->gates_on && ctx->opt_predict_gates;
    s->gates = a.gates;
    a.gate_sv = ctx->gate_sv; a.gate_coef = ctx->gate_coef; a.gate_trust = ctx->gate_trust;
    a.gate_nsv = ctx->gate_nsv; a.gate_gamma = ctx->gate_gamma; a.gate_intercept = ctx->gate_intercept;
    a.gate_positive_finite = ctx->gate_positive_finite; a.gate_has_trust = ctx->gate_has_trust;
#define SV2(DP, KID) hipLaunchKernelGGL((predict_server_kernel<DP, KID>), dim3((unsigned)nsplit), dim3(256), 0, s->stream, a, kp, ap)
#define SV4(KID) { if (ctx->d <= 4) SV2(4, KID); else if (ctx->d <= 8) SV2(8, KID); \
                   else if (ctx->d <= 16) SV2(16, KID); else SV2(32, KID); }
    DISPATCH_KID(ctx->kernel_id, SV4)
#undef SV4
#undef SV2
    HIP_TRY(ctx, hipGetLastError());
    s->running = true;
    s->launches++;
    return 0;
}

// Every entry point that changes the model (or wants the GPU to itself) calls this first: the resident kernel holds
// theta, the affine maps and the buffer addresses of its launch.  No-op when nothing is running.
int serve_stop(gpry_ctx* ctx) {
    SrvHost* s = static_cast<SrvHost*>(ctx->srv);
    if (!s || !s->running) return 0;
    if (!s->exited()) srv_post(s, SRV_CMD_QUIT, nullptr, 0, ctx->d);
    s->running = false;
    HIP_TRY(ctx, hipStreamSynchronize(s->stream));     // at most one request + the idle time away
    return 0;
}

void serve_free(gpry_ctx* ctx) {
    SrvHost* s = static_cast<SrvHost*>(ctx->srv);
    if (!s) return;
    (void)serve_stop(ctx);
    if (s->stream) { (void)hipStreamSynchronize(s->stream); (void)hipStreamDestroy(s->stream); }
    if (s->dflags) (void)hipFree(s->dflags);
    if (s->dx) (void)hipFree(s->dx);
    if (s->h) (void)hipHostFree(s->h);
    delete s;
    ctx->srv = nullptr;
}

void serve_stats(gpry_ctx* ctx, int64_t* launches, int64_t* requests) {
    SrvHost* s = static_cast<SrvHost*>(ctx->srv);
    *launches = s ? s->launches : 0;
    *requests = s ? s->requests : 0;
}

// Mean of M <= SRV_MAXM points through the resident kernel; `part` receives M x nsplit partial sums in the layout of
// the one-launch path (part[m * nsplit + slice]).  gate_bits[m]: GPRY_MASK_* bits of the device gates (0 unless the
// context applies them to gpry_predict).  Returns 0 and *nsplit_out, or an error code.
int serve_predict_mean(gpry_ctx* ctx, const double* X, int64_t M, double* part, int* nsplit_out, unsigned* gate_bits) {
    if (M < 1 || M > SRV_MAXM) return gpry_fail(ctx, -1, "serve: 1..%d points per request", SRV_MAXM);
    SrvHost* s = nullptr;
    GPRY_TRY(srv_get(ctx, &s));
    if (s->running && s->exited()) s->running = false;        // left on its own (idle): the stream is free again
    srv_post(s, SRV_CMD_PREDICT, X, (int)M, ctx->d);
    s->requests++;
    if (!s->running) GPRY_TRY(srv_launch(ctx, s));
    const unsigned long long seq = s->seq;
    volatile SrvUnit* rs = s->res();
    const int nsplit = s->nsplit;
    const auto t0 = std::chrono::steady_clock::now();
    int64_t spins = 0;
    for (;;) {
        bool done = true;
        for (int g = 0; g < nsplit && done; g++)
            for (int m = 0; m < (int)M; m++)
                if (rs[g * SRV_MAXM + m].stamp != seq) { done = false; break; }
        if (done && s->gates)
            for (int m = 0; m < (int)M; m++)
                if (rs[SRV_MAX_SLICES * SRV_MAXM + m].stamp != seq) { done = false; break; }
        if (done) break;
        if (s->exited()) {
            // the generation left without this request (it was posted while the leader was leaving): the next one
            // starts with it.  (A generation never serves a request partially: the leader republishes, then serves.)
            s->running = false;
            GPRY_TRY(srv_launch(ctx, s));
        }
        srv_cpu_relax();
        if ((++spins & 0xfff) == 0) {
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (sec > 10.0) {
                (void)serve_stop(ctx);
                return gpry_fail(ctx, -2, "serve: no answer from the resident predict kernel within 10 s (generation %llu, request %llu)",
                                 s->gen, seq);
            }
        }
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);
    for (int m = 0; m < (int)M; m++)
        for (int g = 0; g < nsplit; g++) {
            const unsigned long long b = rs[g * SRV_MAXM + m].payload;
            memcpy(&part[m * nsplit + g], &b, 8);
        }
    for (int m = 0; m < (int)M; m++) gate_bits[m] = s->gates ? (unsigned)rs[SRV_MAX_SLICES * SRV_MAXM + m].payload : 0u;
    *nsplit_out = nsplit;
    return 0;
}

extern "C" int gpry_debug_serve_stats(gpry_ctx* ctx, int64_t* launches, int64_t* requests) {
    if (!ctx || !launches || !requests) return gpry_fail(ctx, -1, "gpry_debug_serve_stats: NULL argument");
    serve_stats(ctx, launches, requests);
    return 0;
}
