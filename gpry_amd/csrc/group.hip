// Single-process device group: k contexts (on k GPUs, or several on one GPU) driven from k host
// threads behind ONE call -- the form an unmodified gpry.Runner needs, which is one Python process
// without MPI (gpry/run.py:838-844 calls multi_add once; gpry/mpi.py:18-28 falls back to a 1-rank
// dummy when mpi4py is absent).  Same partitioning and merge rule as the one-process-per-GPU path
// (gpry_amd/gp_acquisition.py:NORA._shortlist): contiguous candidate shards, replicated model, one
// exchange of shortlist records, entries at or below the largest per-member bound held back.
// Replaces mpi.compute_y_parallel / step_split / merge_step_split (gpry/mpi.py:105-131,182-218) and
// the gather + merge of per-rank pools (gpry/gp_acquisition.py:1148-1191).
#include "common.h"
#include <rccl/rccl.h>
#include <algorithm>
#include <thread>
#include <stdarg.h>
#include <stdlib.h>

struct gpry_group {
    std::vector<gpry_ctx*> members;
    std::vector<char> owned;            // 0: adopted from the caller (member 0), not destroyed here
    std::vector<int64_t> lo, hi;        // shard [lo, hi) of the resident candidate set per member
    int64_t M = 0;
    char err[1024] = {0};
    // shortlist exchange: 0 = every member copies its records to the host, merged there;
    // 1 = RCCL all-gather between the members' devices (distinct devices only), one copy-out
    int transport = 0;
    std::vector<ncclComm_t> comms;
    std::vector<gpry_cand*> dsend, drecv;
    std::vector<hipStream_t> cstream;
    int64_t xcap = 0;
};

static int group_fail(gpry_group* g, int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
    if (g) { strncpy(g->err, g_last_error, sizeof(g->err) - 1); }
    return code;
}

// run fn(i) for every member on its own host thread; first non-zero status wins (its text is kept)
template <typename F>
static int for_members(gpry_group* g, F fn) {
    const int n = (int)g->members.size();
    std::vector<int> rc(n, 0);
    if (n == 1) { rc[0] = fn(0); }
    else {
        std::vector<std::thread> th;
        th.reserve(n);
        for (int i = 0; i < n; i++) th.emplace_back([&, i]() { rc[i] = fn(i); });
        for (auto& t : th) t.join();
    }
    for (int i = 0; i < n; i++)
        if (rc[i] != 0) return group_fail(g, rc[i], "member %d (device %d): %s", i, g->members[i]->device, g->members[i]->err);
    return 0;
}

static void group_shards(gpry_group* g, int64_t M) {
    const int64_t n = (int64_t)g->members.size();
    const int64_t per = (M + n - 1) / n;
    g->lo.assign(n, 0); g->hi.assign(n, 0);
    for (int64_t i = 0; i < n; i++) {
        g->lo[i] = std::min(i * per, M);
        g->hi[i] = std::min((i + 1) * per, M);
    }
    g->M = M;
}

static void group_rccl_teardown(gpry_group* g) {
    for (size_t i = 0; i < g->comms.size(); i++) {
        (void)hipSetDevice(g->members[i]->device);
        if (g->comms[i]) (void)ncclCommDestroy(g->comms[i]);
        if (i < g->dsend.size() && g->dsend[i]) (void)hipFree(g->dsend[i]);
        if (i < g->drecv.size() && g->drecv[i]) (void)hipFree(g->drecv[i]);
        if (i < g->cstream.size() && g->cstream[i]) (void)hipStreamDestroy(g->cstream[i]);
    }
    g->comms.clear(); g->dsend.clear(); g->drecv.clear(); g->cstream.clear(); g->xcap = 0;
    g->transport = 0;
}

extern "C" {

int gpry_group_create(int n, const int* devices, gpry_ctx* adopt0, gpry_group** out) {
    if (!out) return group_fail(nullptr, -1, "gpry_group_create: out is NULL");
    *out = nullptr;
    if (n < 1 || n > 64 || !devices) return group_fail(nullptr, -1, "gpry_group_create: need 1..64 devices");
    if (adopt0 && adopt0->device != devices[0])
        return group_fail(nullptr, -1, "gpry_group_create: the adopted context lives on device %d, member 0 is device %d",
                          adopt0->device, devices[0]);
    gpry_group* g = new gpry_group();
    for (int i = 0; i < n; i++) {
        gpry_ctx* c = nullptr;
        if (i == 0 && adopt0) { c = adopt0; g->owned.push_back(0); }
        else {
            int rc = gpry_ctx_create(devices[i], &c);
            if (rc) {
                for (size_t k = 0; k < g->members.size(); k++) if (g->owned[k]) gpry_ctx_destroy(g->members[k]);
                delete g;
                std::string why(g_last_error);
                return group_fail(nullptr, rc, "gpry_group_create: member %d (device %d): %s", i, devices[i], why.c_str());
            }
            g->owned.push_back(1);
        }
        g->members.push_back(c);
    }
    // The shortlists are a few KB that the HOST needs (it ranks them): by default every member copies its
    // records out and the host merges.  GPRY_GROUP_TRANSPORT=rccl routes them through an in-process RCCL
    // all-gather between the members' devices instead (only if every member has a device of its own: a
    // communicator cannot hold one device twice) -- opt-in, because that path cannot be exercised on the
    // 1-GPU development boxes and an unmodified Runner reaches the group by default.
    bool distinct = n > 1;
    for (int i = 0; i < n && distinct; i++)
        for (int k = 0; k < i; k++) if (devices[i] == devices[k]) { distinct = false; break; }
    const char* env = getenv("GPRY_GROUP_TRANSPORT");
    if (distinct && env && !strcmp(env, "rccl")) {
        g->comms.assign(n, nullptr);
        ncclResult_t r = ncclCommInitAll(g->comms.data(), n, devices);
        if (r != ncclSuccess) {
            // not silent: gpry_group_size reports transport 0 and the reason stays in the error text
            group_fail(g, 0, "ncclCommInitAll over %d devices failed (%s): shortlists are merged on the host", n,
                       ncclGetErrorString(r));
            g->comms.clear();
        } else {
            g->transport = 1;
            g->dsend.assign(n, nullptr); g->drecv.assign(n, nullptr); g->cstream.assign(n, nullptr);
            for (int i = 0; i < n; i++) {
                (void)hipSetDevice(devices[i]);
                if (hipStreamCreateWithFlags(&g->cstream[i], hipStreamNonBlocking) != hipSuccess) { group_rccl_teardown(g); break; }
            }
        }
    }
    *out = g;
    return 0;
}

int gpry_group_destroy(gpry_group* g) {
    if (!g) return 0;
    group_rccl_teardown(g);
    for (size_t k = 0; k < g->members.size(); k++) if (g->owned[k]) gpry_ctx_destroy(g->members[k]);
    delete g;
    return 0;
}

int gpry_group_size(gpry_group* g, int* n, int* transport) {
    if (!g) return group_fail(nullptr, -1, "gpry_group_size: group is NULL");
    if (n) *n = (int)g->members.size();
    if (transport) *transport = g->transport;
    return 0;
}

gpry_ctx* gpry_group_member(gpry_group* g, int i) {
    if (!g || i < 0 || i >= (int)g->members.size()) return nullptr;
    return g->members[i];
}

const char* gpry_group_last_error(gpry_group* g) { return g ? g->err : g_last_error; }

int gpry_group_set_model(gpry_group* g, const double* X_, const double* y_, const double* alpha, int64_t N, int d,
                         int kernel_id, const double* theta, const gpry_affine* tf, int* info) {
    if (!g) return group_fail(nullptr, -1, "gpry_group_set_model: group is NULL");
    if (info) *info = 0;
    std::vector<int> infos(g->members.size(), 0);
    GPRY_TRY(for_members(g, [&](int i) -> int {
        gpry_ctx* c = g->members[i];
        if (!g->owned[i]) {        // the owner keeps this one up to date itself; it must be ready
            if (!c->factor_valid) return gpry_fail(c, -1, "the adopted context holds no factorised model");
            return 0;
        }
        GPRY_TRY(gpry_set_train(c, X_, y_, alpha, N, d));
        GPRY_TRY(gpry_set_theta(c, kernel_id, theta));
        if (tf) GPRY_TRY(gpry_set_affine(c, tf));
        return gpry_factorize(c, &infos[i]);
    }));
    for (int v : infos) if (v != 0 && info && *info == 0) *info = v;
    return 0;
}

int gpry_group_set_gates(gpry_group* g, const double* sv, const double* coef, int64_t n_sv, double gamma,
                         double intercept, int positive_is_finite, const double* trust_bounds) {
    if (!g) return group_fail(nullptr, -1, "gpry_group_set_gates: group is NULL");
    return for_members(g, [&](int i) -> int {
        return gpry_set_gates(g->members[i], sv, coef, n_sv, gamma, intercept, positive_is_finite, trust_bounds);
    });
}

int gpry_group_sweep_logexp(gpry_group* g, const double* X, int64_t M, const uint8_t* mask, double zeta,
                            double baseline, double sigma_n, double* y_all, double* sigma_all, double* acq_all,
                            int64_t* n_nan) {
    if (!g) return group_fail(nullptr, -1, "gpry_group_sweep_logexp: group is NULL");
    if (M <= 0) return group_fail(g, -1, "group sweep: M must be > 0");
    if (!X && M != g->M) return group_fail(g, -1, "X == NULL but the resident candidate set has %lld rows, not %lld",
                                          (long long)g->M, (long long)M);
    if (X) group_shards(g, M);
    std::vector<int64_t> nn(g->members.size(), 0);
    GPRY_TRY(for_members(g, [&](int i) -> int {
        gpry_ctx* c = g->members[i];
        const int64_t lo = g->lo[i], m = g->hi[i] - lo;
        if (m <= 0) return 0;
        return gpry_sweep_logexp(c, X ? X + lo * c->d : nullptr, m, mask ? mask + lo : nullptr, zeta, baseline,
                                 sigma_n, y_all ? y_all + lo : nullptr, sigma_all ? sigma_all + lo : nullptr,
                                 acq_all ? acq_all + lo : nullptr, &nn[i]);
    }));
    if (n_nan) { *n_nan = 0; for (int64_t v : nn) *n_nan += v; }
    return 0;
}

int gpry_group_sweep_fetch(gpry_group* g, int64_t M, double* y_all, double* sigma_all, double* acq_all) {
    if (!g) return group_fail(nullptr, -1, "gpry_group_sweep_fetch: group is NULL");
    if (M <= 0 || M != g->M) return group_fail(g, -1, "group sweep_fetch: the resident sweep has %lld candidates, not %lld",
                                              (long long)g->M, (long long)M);
    return for_members(g, [&](int i) -> int {
        const int64_t lo = g->lo[i], m = g->hi[i] - lo;
        if (m <= 0) return 0;
        return gpry_sweep_fetch(g->members[i], m, y_all ? y_all + lo : nullptr, sigma_all ? sigma_all + lo : nullptr,
                                acq_all ? acq_all + lo : nullptr);
    });
}

int gpry_group_sweep_topk(gpry_group* g, int64_t Kp, const int64_t* exclude, int64_t n_exclude, gpry_cand* top,
                          int64_t* n_out, double* bound, int* exhausted) {
    if (!g) return group_fail(nullptr, -1, "gpry_group_sweep_topk: group is NULL");
    if (!top || !n_out || !bound) return group_fail(g, -1, "group topk: top, n_out and bound must not be NULL");
    if (Kp < 0) return group_fail(g, -1, "group topk: Kp < 0");
    if (g->M <= 0) return group_fail(g, -1, "group topk: no sweep results resident");
    const int n = (int)g->members.size();
    const int64_t stride = Kp + 1;           // Kp records + trailer {acq = bound, idx = count}
    std::vector<gpry_cand> rec((size_t)n * stride);
    // exclusions (sorted global rows) split by shard
    std::vector<std::vector<int64_t>> ex(n);
    for (int64_t e = 0; e < n_exclude; e++)
        for (int i = 0; i < n; i++)
            if (exclude[e] >= g->lo[i] && exclude[e] < g->hi[i]) { ex[i].push_back(exclude[e] - g->lo[i]); break; }
    GPRY_TRY(for_members(g, [&](int i) -> int {
        gpry_cand* mine = rec.data() + (size_t)i * stride;
        int64_t cnt = 0; double bd = -INFINITY;
        if (g->hi[i] > g->lo[i] && Kp > 0)
            GPRY_TRY(gpry_sweep_topk(g->members[i], Kp, ex[i].empty() ? nullptr : ex[i].data(), (int64_t)ex[i].size(),
                                     mine, &cnt, &bd));
        for (int64_t k = 0; k < cnt; k++) mine[k].idx += g->lo[i];
        for (int64_t k = cnt; k < Kp; k++) { mine[k].acq = -INFINITY; mine[k].y = mine[k].sigma = 0.0; mine[k].idx = -1; }
        mine[Kp].acq = bd; mine[Kp].y = mine[Kp].sigma = 0.0; mine[Kp].idx = cnt;
        return 0;
    }));
    if (g->transport == 1) {
        // every device ends up with every member's records (one all-gather over xGMI); the host reads
        // member 0's copy.  The records merged below are the gathered ones.
        const size_t bytes = (size_t)stride * sizeof(gpry_cand);
        if (stride > g->xcap) {
            for (int i = 0; i < n; i++) {
                if (hipSetDevice(g->members[i]->device) != hipSuccess) return group_fail(g, -2, "hipSetDevice");
                if (g->dsend[i]) (void)hipFree(g->dsend[i]);
                if (g->drecv[i]) (void)hipFree(g->drecv[i]);
                g->dsend[i] = g->drecv[i] = nullptr;
                if (hipMalloc((void**)&g->dsend[i], bytes) != hipSuccess || hipMalloc((void**)&g->drecv[i], bytes * n) != hipSuccess)
                    return group_fail(g, -2, "group topk: exchange buffers");
            }
            g->xcap = stride;
        }
        std::vector<gpry_cand> gathered((size_t)n * stride);
        int rc = for_members(g, [&](int i) -> int {
            gpry_ctx* c = g->members[i];
            HIP_TRY(c, hipSetDevice(c->device));
            hipStream_t st = g->cstream[i];
            HIP_TRY(c, hipMemcpyAsync(g->dsend[i], rec.data() + (size_t)i * stride, bytes, hipMemcpyHostToDevice, st));
            ncclResult_t r = ncclAllGather(g->dsend[i], g->drecv[i], bytes, ncclChar, g->comms[i], st);
            if (r != ncclSuccess) return gpry_fail(c, -5, "ncclAllGather: %s", ncclGetErrorString(r));
            if (i == 0) HIP_TRY(c, hipMemcpyAsync(gathered.data(), g->drecv[0], bytes * n, hipMemcpyDeviceToHost, st));
            HIP_TRY(c, hipStreamSynchronize(st));
            return 0;
        });
        if (rc) return rc;
        rec.swap(gathered);
    }
    std::vector<gpry_cand> merged;
    double gbound = -INFINITY;
    bool exh = true;
    for (int i = 0; i < n; i++) {
        const gpry_cand* mine = rec.data() + (size_t)i * stride;
        const int64_t cnt = mine[Kp].idx;
        merged.insert(merged.end(), mine, mine + cnt);
        if (mine[Kp].acq > gbound) gbound = mine[Kp].acq;
        exh = exh && cnt < Kp;
    }
    std::sort(merged.begin(), merged.end(), [](const gpry_cand& a, const gpry_cand& b) {
        unsigned long long ka, kb; double x = a.acq, y = b.acq;
        memcpy(&ka, &x, 8); memcpy(&kb, &y, 8);
        ka = (ka >> 63) ? ~ka : (ka | 0x8000000000000000ull);
        kb = (kb >> 63) ? ~kb : (kb | 0x8000000000000000ull);
        if (ka != kb) return ka > kb;
        return a.idx > b.idx;
    });
    // entries at or below the largest per-member bound may be preceded by candidates a member did
    // not send: held back until a later, longer shortlist (exactness of the descending stream)
    int64_t w = 0;
    for (const gpry_cand& c : merged) if (exh || c.acq > gbound) top[w++] = c;
    *n_out = w;
    *bound = exh ? -INFINITY : gbound;
    if (exhausted) *exhausted = exh ? 1 : 0;
    return 0;
}

int gpry_group_lml_batch(gpry_group* g, const double* thetas, int n_theta, int want_grad, double* lml, double* grad,
                         int* info) {
    if (!g) return group_fail(nullptr, -1, "gpry_group_lml_batch: group is NULL");
    if (n_theta <= 0) return 0;
    if (!thetas || !lml || (want_grad && !grad)) return group_fail(g, -1, "group lml: thetas, lml and (with want_grad) grad must not be NULL");
    const int n = (int)g->members.size();
    return for_members(g, [&](int i) -> int {
        gpry_ctx* c = g->members[i];
        const int w = c->d + 1;
        for (int t = i; t < n_theta; t += n) {      // round robin: member i takes thetas i, i+n, ...
            int inf = 0;
            GPRY_TRY(gpry_lml(c, thetas + (size_t)t * w, want_grad, lml + t, want_grad ? grad + (size_t)t * w : nullptr, &inf));
            if (info) info[t] = inf;
        }
        return 0;
    });
}

}  // extern "C"
