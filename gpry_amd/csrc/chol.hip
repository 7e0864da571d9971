// Dense FP64 factorisation kernels for gfx950 (row-major, lower triangular):
//   potrf       : A = L L^T          scipy.linalg.cholesky  at gpry/gpr.py:1456   (chol_panel.hip)
//   trtri_lower : V = L^-1           solve_triangular(L, I) at gpry/gpr.py:1457
//   lauum_lower : K^-1 = V^T V       cho_solve(L, I)        at sklearn:_gpr.py:640-642
//   solve_alpha : alpha_ = V^T (V y) cho_solve(L, y)        at gpry/gpr.py:1465
// The O(N^3) parts run on the FP64 MFMA GEMM (gemm_f64.hip); the 64x64 diagonal work
// runs in LDS / registers.  Matrices are padded to a multiple of 128 with an identity
// block, so no kernel needs edge handling.
#include "common.h"
#include <algorithm>
#include <dlfcn.h>

// ------------------------------------------------------------------------------------
// Inverse of every 64x64 diagonal block: lane c solves L x = e_c by forward substitution
// (L broadcast from LDS, x in registers).  Writes zeros above the diagonal.
__global__ __launch_bounds__(64) void trtri_diag_kernel(const double* __restrict__ L,
                                                        double* __restrict__ V, int64_t ld,
                                                        const int* info) {
    __shared__ double Lk[64 * 64];
    if (*info != 0) return;
    const int t = threadIdx.x;
    const int64_t b0 = (int64_t)blockIdx.x * 64;
    for (int e = t; e < 64 * 64; e += 64) {
        int i = e >> 6, j = e & 63;
        Lk[e] = L[(b0 + i) * ld + b0 + j];
    }
    __syncthreads();
    double x[64];
#pragma unroll
    for (int i = 0; i < 64; i++) {
        double s = (i == t) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 0; k < i; k++) s = fma(-Lk[i * 64 + k], x[k], s);
        x[i] = (i >= t) ? s / Lk[i * 64 + i] : 0.0;
    }
#pragma unroll
    for (int i = 0; i < 64; i++) V[(b0 + i) * ld + b0 + t] = x[i];
}

struct TriNode { int lo, mid, hi, level; };
static int build_tree(int lo, int hi, std::vector<TriNode>& out) {
    if (hi - lo <= 1) return 0;
    int mid = lo + (hi - lo + 1) / 2;
    int l1 = build_tree(lo, mid, out), l2 = build_tree(mid, hi, out);
    int lev = (l1 > l2 ? l1 : l2) + 1;
    out.push_back({lo, mid, hi, lev});
    return lev;
}

// cached per-Np batch descriptors of the trtri recursion
// The same recursion cut into phases for the pipelined factor chain (trtri_pipeline_*): phase p holds what
// can run once the first `blocks_done` 64-column blocks of L are final.
struct TrtriPhaseLevel {
    int lev = 0;                            // index into the per-level arrays of the plan
    int64_t t_first = 0, v_first = 0;       // offsets into d_phase_items
    int n_t = 0, n_v = 0, mM_t = 0, mN_t = 0, mM_v = 0, mN_v = 0;
    GemmPartsPlan sk_t, sk_v;               // stream-K plans of this subset (levels with seg_t > 0)
};
struct TrtriPhase {
    int blocks_done = 0, diag_lo = 0, diag_hi = 0;
    std::vector<TrtriPhaseLevel> levels;
};
struct TrtriPlan {
    int64_t Np = 0;
    int leaf = 1;                           // leaves of the recursion in 64-row blocks (2: 128 x 128 diagonal stage)
    std::vector<GemmBatchItem*> d_t, d_v;   // per level
    std::vector<int> count, maxM, maxN;
    std::vector<int> aligned;               // per level: every item is a multiple of 128 in M, N and K
    std::vector<TrtriPhase> phases;         // empty: too few blocks to cut
    GemmBatchItem* d_phase_items = nullptr;
    // stream-K (gemm_dma.hip) for the aligned levels with blocks >= 512: segment length per level and product,
    // taken from the WHOLE level so that a phase's subset is cut exactly like the full launch
    std::vector<int> seg_t, seg_v;          // per level; 0: the level is launched tile by tile
    std::vector<GemmPartsPlan> sk_t, sk_v;  // per level: the full launch
    GemmPartsPlan sk_lauum;                 // K^-1 = V^T V
    int seg_lauum = 0;
    void release() {
        for (auto p : d_t) if (p) (void)hipFree(p);
        for (auto p : d_v) if (p) (void)hipFree(p);
        if (d_phase_items) (void)hipFree(d_phase_items);
        for (auto& q : sk_t) gemm_parts_plan_free(&q);
        for (auto& q : sk_v) gemm_parts_plan_free(&q);
        for (auto& ph : phases) for (auto& pv : ph.levels) { gemm_parts_plan_free(&pv.sk_t); gemm_parts_plan_free(&pv.sk_v); }
        gemm_parts_plan_free(&sk_lauum);
    }
};
// the plan lives in its context (ctx->trtri_plan): distinct contexts may be driven from distinct threads
void trtri_plan_free(gpry_ctx* ctx) {
    TrtriPlan* pl = static_cast<TrtriPlan*>(ctx->trtri_plan);
    if (!pl) return;
    pl->release();
    delete pl;
    ctx->trtri_plan = nullptr;
}

static int trtri_plan_get(gpry_ctx* ctx, int64_t Np, TrtriPlan** out) {
    if (!ctx->trtri_plan) ctx->trtri_plan = new TrtriPlan();
    TrtriPlan& pl = *static_cast<TrtriPlan*>(ctx->trtri_plan);
    if (pl.Np == Np) { *out = &pl; return 0; }
    pl.release();
    pl = TrtriPlan();       // Np = 0: an incomplete plan is never taken for a finished one
    int slots = 512;        // two GEMM workgroups per CU
    { hipDeviceProp_t prop; if (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) slots = 2 * prop.multiProcessorCount; }
    // up to Np = 1024 the leaves are 128 x 128 blocks, inverted in LDS by one workgroup each (trtri_small.hip): the tree
    // is built over 128-row blocks and its coordinates scaled back to the 64-row units everything below works in
    pl.leaf = (Np > 128 && Np <= 1024) ? 2 : 1;
    std::vector<TriNode> nodes;
    int nlev = build_tree(0, (int)(Np / (64 * pl.leaf)), nodes);
    for (auto& nd : nodes) { nd.lo *= pl.leaf; nd.mid *= pl.leaf; nd.hi *= pl.leaf; }
    for (int lev = 1; lev <= nlev; lev++) {
        std::vector<GemmBatchItem> bt, bv;
        int mM = 0, mN = 0, al = 1;
        for (auto& nd : nodes) {
            if (nd.level != lev) continue;
            int64_t lo = nd.lo * 64, mid = nd.mid * 64, hi = nd.hi * 64;
            int m = (int)(hi - mid), n = (int)(mid - lo);
            GemmBatchItem a;  // T[mid:hi, lo:mid] = L[mid:hi, lo:mid] * V[lo:mid, lo:mid]
            a.a_off = mid * Np + lo; a.b_off = lo * Np + lo; a.c_off = mid * Np + lo;
            a.M = m; a.N = n; a.K = n; a.pad = 0;
            bt.push_back(a);
            GemmBatchItem b;  // V[mid:hi, lo:mid] = -V[mid:hi, mid:hi] * T[mid:hi, lo:mid]
            b.a_off = mid * Np + mid; b.b_off = mid * Np + lo; b.c_off = mid * Np + lo;
            b.M = m; b.N = n; b.K = m; b.pad = 0;
            bv.push_back(b);
            if (m > mM) mM = m;
            if (n > mN) mN = n;
            if (m % 128 || n % 128) al = 0;
        }
        GemmBatchItem *dt = nullptr, *dv = nullptr;
        size_t bytes = bt.size() * sizeof(GemmBatchItem);
        hipError_t e = hipMalloc(&dt, bytes);
        if (e == hipSuccess) e = hipMalloc(&dv, bytes);
        if (e == hipSuccess) e = hipMemcpy(dt, bt.data(), bytes, hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemcpy(dv, bv.data(), bytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {      // drop the partial plan: the next call starts over
            if (dt) (void)hipFree(dt);
            if (dv) (void)hipFree(dv);
            pl.release();
            pl = TrtriPlan();
            return gpry_fail(ctx, -2, "trtri plan (Np = %lld): %s", (long long)Np, hipGetErrorString(e));
        }
        pl.d_t.push_back(dt); pl.d_v.push_back(dv);
        pl.count.push_back((int)bt.size()); pl.maxM.push_back(mM); pl.maxN.push_back(mN);
        pl.aligned.push_back(al);
        // stream-K for the levels whose launches have few, long tiles
        pl.seg_t.push_back(0); pl.seg_v.push_back(0);
        pl.sk_t.emplace_back(); pl.sk_v.emplace_back();
        if (al && mN >= 512 && Np <= ctx->opt_gemm_streamk) {     // (built only for the sizes that launch them)
            std::vector<GemmShape> st_, sv_;
            for (auto& a : bt) st_.push_back({a.M, a.N, a.K});
            for (auto& b : bv) sv_.push_back({b.M, b.N, b.K});
            pl.seg_t.back() = gemm_parts_segment(KM_B_LOWER, 0, st_, slots);
            pl.seg_v.back() = gemm_parts_segment(KM_A_LOWER, 0, sv_, slots);
            int rc = gemm_parts_plan_build(ctx, KM_B_LOWER, 0, st_, pl.seg_t.back(), &pl.sk_t.back());
            if (rc == 0) rc = gemm_parts_plan_build(ctx, KM_A_LOWER, 0, sv_, pl.seg_v.back(), &pl.sk_v.back());
            if (rc) { pl.release(); pl = TrtriPlan(); return rc; }
        }
    }
    if (Np >= 512 && Np <= ctx->opt_gemm_streamk) {   // K^-1 = V^T V: lower tiles, k >= max(i, j) * 128
        std::vector<GemmShape> sh = {{(int)Np, (int)Np, (int)Np}};
        pl.seg_lauum = gemm_parts_segment(KM_AT_LOWER_B_LOWER, 1, sh, slots);
        int rc = gemm_parts_plan_build(ctx, KM_AT_LOWER_B_LOWER, 1, sh, pl.seg_lauum, &pl.sk_lauum);
        if (rc) { pl.release(); pl = TrtriPlan(); return rc; }
    }
    // Phases: checkpoints at 1/4, 1/2 and 3/4 of the blocks (see below).  Diagonal block b is ready once b + 1 blocks of L are final, the product
    // T = L21 V11 of a node once `mid` blocks are (columns lo..mid of L are final for ALL rows after their
    // panel step, and V11 is complete by then), its V21 = -V22 T once `hi` blocks are.
    const int nblk = (int)(Np / 64);
    if (nblk >= 8) {
        const TriNode root = nodes.back();
        std::vector<int> cps = {root.mid, nblk};
        for (auto& nd : nodes)
            if (nd.lo == root.lo && nd.hi == root.mid) cps.push_back(nd.mid);      // first quarter
        // ... and the split points down the right spine of the tree while a node has at least half of all blocks,
        // i.e. only 3/4.  Going on to 7/8, 15/16, ... leaves less for after the last panel on paper, but the
        // small phases are chains of 5-20 us launches that start one or two panel steps before the end and are
        // not finished when potrf is (exposed V = L^-1 at N = 4096: 0.49 ms with 3/4, 0.55 with 7/8, 0.62 with
        // 15/16, 0.68 down to the last block; tools/ab_factor_pipeline.py).
        const int spine = nblk / 2;
        for (int lo = root.mid, hi = root.hi; hi - lo >= spine && hi - lo > 1;) {
            int mid = -1;
            for (auto& nd : nodes) if (nd.lo == lo && nd.hi == hi) mid = nd.mid;
            if (mid < 0) break;
            cps.push_back(mid);
            lo = mid;
        }
        std::sort(cps.begin(), cps.end());
        cps.erase(std::unique(cps.begin(), cps.end()), cps.end());
        auto phase_of = [&](int blocks) { size_t p = 0; while (cps[p] < blocks) p++; return p; };
        std::vector<GemmBatchItem> all;
        int dlo = 0;
        for (size_t p = 0; p < cps.size(); p++) {
            TrtriPhase ph;
            ph.blocks_done = cps[p]; ph.diag_lo = dlo; ph.diag_hi = cps[p]; dlo = cps[p];
            for (int lev = 1; lev <= nlev; lev++) {
                TrtriPhaseLevel pv; pv.lev = lev - 1;
                std::vector<GemmBatchItem> bt, bv;
                for (auto& nd : nodes) {
                    if (nd.level != lev) continue;
                    int64_t lo = nd.lo * 64, mid = nd.mid * 64, hi = nd.hi * 64;
                    int m = (int)(hi - mid), n = (int)(mid - lo);
                    if (phase_of(nd.mid) == p) {
                        GemmBatchItem a;
                        a.a_off = mid * Np + lo; a.b_off = lo * Np + lo; a.c_off = mid * Np + lo;
                        a.M = m; a.N = n; a.K = n; a.pad = 0;
                        bt.push_back(a);
                        if (m > pv.mM_t) pv.mM_t = m;
                        if (n > pv.mN_t) pv.mN_t = n;
                    }
                    if (phase_of(nd.hi) == p) {
                        GemmBatchItem b;
                        b.a_off = mid * Np + mid; b.b_off = mid * Np + lo; b.c_off = mid * Np + lo;
                        b.M = m; b.N = n; b.K = m; b.pad = 0;
                        bv.push_back(b);
                        if (m > pv.mM_v) pv.mM_v = m;
                        if (n > pv.mN_v) pv.mN_v = n;
                    }
                }
                if (bt.empty() && bv.empty()) continue;
                if (pl.seg_t[lev - 1] > 0) {
                    std::vector<GemmShape> st_, sv_;
                    for (auto& a : bt) st_.push_back({a.M, a.N, a.K});
                    for (auto& b : bv) sv_.push_back({b.M, b.N, b.K});
                    int rc = gemm_parts_plan_build(ctx, KM_B_LOWER, 0, st_, pl.seg_t[lev - 1], &pv.sk_t);
                    if (rc == 0) rc = gemm_parts_plan_build(ctx, KM_A_LOWER, 0, sv_, pl.seg_v[lev - 1], &pv.sk_v);
                    if (rc) { ph.levels.push_back(pv); pl.phases.push_back(ph); pl.release(); pl = TrtriPlan(); return rc; }
                }
                pv.n_t = (int)bt.size(); pv.n_v = (int)bv.size();
                pv.t_first = (int64_t)all.size(); all.insert(all.end(), bt.begin(), bt.end());
                pv.v_first = (int64_t)all.size(); all.insert(all.end(), bv.begin(), bv.end());
                ph.levels.push_back(pv);
            }
            pl.phases.push_back(ph);
        }
        hipError_t e = hipMalloc(&pl.d_phase_items, all.size() * sizeof(GemmBatchItem));
        if (e == hipSuccess) e = hipMemcpy(pl.d_phase_items, all.data(), all.size() * sizeof(GemmBatchItem), hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            pl.release();
            pl = TrtriPlan();
            return gpry_fail(ctx, -2, "trtri plan (Np = %lld): %s", (long long)Np, hipGetErrorString(e));
        }
    }
    pl.Np = Np;             // committed only once every level is on the device
    *out = &pl;
    return 0;
}

// split-K factor of a level: from the figures of the WHOLE level, also when only part of its items is
// launched (pipelined chain), so that every product is summed in the same order in both schedules
static int trtri_level_nsplit(gpry_ctx* ctx, const TrtriPlan* pl, size_t lev) {
    // the top levels are a handful of long tiles: split their K range so that the launch fills
    // the GPU (512 resident workgroups) and its critical path shrinks accordingly
    if (ctx->tp) return 1;              // throughput schedule: whole tiles only (the thetas of the call fill the GPU)
    const int tm = (pl->maxM[lev] + 127) / 128, tn = (pl->maxN[lev] + 127) / 128;
    const int64_t tiles = (int64_t)tm * tn * pl->count[lev];
    int nsplit = 1;
    if (pl->maxN[lev] >= 512) {
        while (nsplit < 4 && tiles * nsplit * 2 <= 1024 && pl->maxN[lev] / (nsplit * 2) >= 128) nsplit *= 2;
    }
    return nsplit;
}

// one level's pair of batched products (all of the level, or the part of it that belongs to a phase)
static int trtri_level_products(gpry_ctx* ctx, const double* L, double* V, double* T, int64_t Np, int nsplit, int aligned,
                                const GemmBatchItem* d_t, int n_t, int mM_t, int mN_t,
                                const GemmBatchItem* d_v, int n_v, int mM_v, int mN_v, hipStream_t st,
                                const GemmPartsPlan* sk_t = nullptr, const GemmPartsPlan* sk_v = nullptr) {
    if (sk_t && sk_v && Np <= ctx->opt_gemm_streamk && !ctx->tp) {       // stream-K launches (gemm_dma.hip)
        if (n_t > 0) {
            GemmArgs g = {};
            g.A = L; g.lda = Np; g.B = V; g.ldb = Np; g.C = T; g.ldc = Np;
            g.kmode = KM_B_LOWER; g.batch = d_t; g.n_batch = n_t; g.info = ctx->dinfo; g.stream = st;
            GPRY_TRY(gemm_dma_parts_launch(ctx, g, false, false, EPI_STORE, *sk_t, Np * Np));
        }
        if (n_v > 0) {
            GemmArgs h = {};
            h.A = V; h.lda = Np; h.B = T; h.ldb = Np; h.C = V; h.ldc = Np;
            h.kmode = KM_A_LOWER; h.batch = d_v; h.n_batch = n_v; h.info = ctx->dinfo; h.stream = st;
            GPRY_TRY(gemm_dma_parts_launch(ctx, h, false, false, EPI_STORE_NEG, *sk_v, Np * Np));
        }
        return 0;
    }
    double* sbuf = nullptr;
    if (nsplit > 1) GPRY_TRY(gemm_split_scratch(ctx, nsplit, Np * Np, &sbuf));
    if (n_t > 0) {
        GemmArgs g = {};
        g.A = L; g.lda = Np; g.B = V; g.ldb = Np; g.C = T; g.ldc = Np;
        g.M = mM_t; g.N = mN_t; g.K = 0;
        g.kmode = KM_B_LOWER; g.lower_only = 0; g.tile_map = TM_ROWMAJOR;
        g.batch = d_t; g.n_batch = n_t; g.info = ctx->dinfo; g.stream = st;
        g.nsplit = nsplit; g.split_buf = sbuf; g.split_stride = Np * Np; g.dma_ok = aligned;
        g.small64 = 1;      // the tree is made of 64-row blocks
        GPRY_TRY(gemm_f64_launch(ctx, g, false, false, EPI_STORE));
    }
    if (n_v > 0) {
        GemmArgs h = {};
        h.A = V; h.lda = Np; h.B = T; h.ldb = Np; h.C = V; h.ldc = Np;
        h.M = mM_v; h.N = mN_v; h.K = 0;
        h.kmode = KM_A_LOWER; h.lower_only = 0; h.tile_map = TM_ROWMAJOR;
        h.batch = d_v; h.n_batch = n_v; h.info = ctx->dinfo; h.stream = st;
        h.nsplit = nsplit; h.split_buf = sbuf; h.split_stride = Np * Np; h.dma_ok = aligned;
        h.small64 = 1;
        GPRY_TRY(gemm_f64_launch(ctx, h, false, false, EPI_STORE_NEG));
    }
    return 0;
}

// V <- 0 for every theta of a batched launch (one memset per theta otherwise)
__global__ __launch_bounds__(256) void zero_sets_kernel(double2* __restrict__ p_, int64_t n2, int64_t bstride) {
    double2* __restrict__ p = bset(p_, (int)blockIdx.z, bstride);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n2; i += stride) p[i] = make_double2(0.0, 0.0);
}

// V = L^-1 by recursive doubling: [[L11,0],[L21,L22]]^-1 = [[V11,0],[-V22 L21 V11, V22]];
// every level is two batched MFMA GEMMs.  T is an Np x Np scratch.
int trtri_lower(gpry_ctx* ctx, const double* L, double* V, double* T, int64_t Np) {
    hipStream_t st = ctx->stream;
    // The four-wave kernels from Np = 256 on.  Up to Np = 128 the whole inverse IS this stage, and
    // the column-by-column substitution of the single-wave kernel is the operation order of the
    // reference's dtrsm: on the cond(K) = 5e15 matrix of BASELINE config 1 (N = 64) any other order moves
    // the posterior mean by 2..4e-5 of its range, beyond the one-ulp noise floor the golden test allows.
    const bool single_wave = Np <= 128;
    // V above its block diagonal has to be zero (the products walk whole tiles).  Up to Np = 1024 the workgroups of
    // the 128 x 128 diagonal stage clear the blocks to the right of their own (a memset is one more dependent dispatch, 5 us
    // of an evaluation that takes 200 at N = 256); above, one memset of the matrix is cheaper than their row strips.
    const bool clear_in_diag = !single_wave && Np <= 1024;
    // The throughput schedule of gpry_lml_batch works on scratch sets nobody reads as matrices: what the products (128 x 128 or
    // 64 x 64 tiles with 64-aligned origins and triangular k-ranges), K^-1 = V^T V and the alpha kernels touch above the
    // diagonal is the 64 x 64 block right of every diagonal block and nothing else.  The diagonal stage writes those zeros
    // itself (128-row leaves: the whole leaf block) -- no pass over the matrix (0.4 ms of 22 for 16 thetas at N = 4096).
    const bool tp_min_clear = ctx->tp && !single_wave;
    if (tp_min_clear) {
    } else if (!clear_in_diag && ctx->bn > 1) {
        const int64_t n2 = Np * Np / 2;
        int64_t nb = (n2 + 2047) / 2048;          // 8 double2 per thread
        if (nb > 4096) nb = 4096;
        hipLaunchKernelGGL(zero_sets_kernel, dim3((unsigned)nb, 1, (unsigned)ctx->bn), dim3(256), 0, st, reinterpret_cast<double2*>(V), n2,
                           ctx->bstride);
        HIP_TRY(ctx, hipGetLastError());
    } else if (!clear_in_diag) HIP_TRY(ctx, hipMemsetAsync(V, 0, sizeof(double) * Np * Np, st));
    TrtriPlan* pl = nullptr;
    GPRY_TRY(trtri_plan_get(ctx, Np, &pl));
    if (single_wave && ctx->bn > 1) return gpry_fail(ctx, -1, "batched chain: the single-wave diagonal stage is not batched");
    if (single_wave) {
        hipLaunchKernelGGL(trtri_diag_kernel, dim3((unsigned)(Np / 64)), dim3(64), 0, st, L, V, Np, ctx->dinfo);
        HIP_TRY(ctx, hipGetLastError());
    } else if (pl->leaf == 2) {         // (128 < Np <= 1024: exactly the sizes that clear in the diagonal stage)
        GPRY_TRY(launch_trtri_diag128(ctx, L, V, Np, st, !tp_min_clear));
    } else {
        GPRY_TRY(launch_trtri_diag(ctx, L, V, Np, st, tp_min_clear));
    }
    for (size_t lev = 0; lev < pl->count.size(); lev++)
        GPRY_TRY(trtri_level_products(ctx, L, V, T, Np, trtri_level_nsplit(ctx, pl, lev), pl->aligned[lev],
                                      pl->d_t[lev], pl->count[lev], pl->maxM[lev], pl->maxN[lev],
                                      pl->d_v[lev], pl->count[lev], pl->maxM[lev], pl->maxN[lev], st,
                                      pl->seg_t[lev] > 0 ? &pl->sk_t[lev] : nullptr, pl->seg_t[lev] > 0 ? &pl->sk_v[lev] : nullptr));
    return 0;
}

// ---- V = L^-1 underneath the Cholesky factorisation --------------------------------------------
// The panel chain of potrf is latency-bound and leaves most of the GPU idle (13.8 % of the matrix pipe at
// N = 4096), and three quarters of V = L^-1 depend on columns of L that are final long before the last
// panel: after every checkpoint (a quarter of the blocks) the potrf loop calls trtri_pipeline_step, which
// records an event on the main stream and queues that phase's products on stream2.  The last phase runs on
// the main stream once both are done.  Same products, same split-K factors, same operands as trtri_lower:
// V is bit-identical (tests/test_hip_parity.py::test_pipelined_factor_chain_is_bit_identical).
// The side work is not free for the chain: a panel-step workgroup needs a whole CU's LDS (135 KB) and waits
// while the CUs hold GEMM workgroups of the side stream (potrf 1.70 -> 1.94 ms at N = 4096, V = L^-1 after it
// 0.87 -> 0.49 ms: LML+gradient 3.35 -> 3.22 ms; 7.20 -> 6.94 at 6144, 13.9 -> 13.5 at 8192; neutral at 3072, a loss below
// that (default from Np = 4096; tools/ab_factor_pipeline.py).  A lowest-priority side stream changes nothing (no CU is held free
// for the urgent kernel); a side stream masked to 64 / 128 / 192 CUs (hipExtStreamCreateWithCUMask) doubles
// the time of the potrf launches themselves (4.2 / 3.7 / 3.7 ms): not kept.
struct TrtriPipe {
    const double* L = nullptr; double* V = nullptr; double* T = nullptr;
    int64_t Np = 0; TrtriPlan* pl = nullptr; size_t next = 0; bool active = false;
    hipStream_t side = nullptr;     // where the early phases run (ctx->stream2)
};
static TrtriPipe* pipe_of(gpry_ctx* ctx) {
    if (!ctx->trtri_pipe) ctx->trtri_pipe = new TrtriPipe();
    return static_cast<TrtriPipe*>(ctx->trtri_pipe);
}
void trtri_pipe_free(gpry_ctx* ctx) {
    delete static_cast<TrtriPipe*>(ctx->trtri_pipe);
    ctx->trtri_pipe = nullptr;
}
static int trtri_phase_run(gpry_ctx* ctx, TrtriPipe* pp, size_t p, hipStream_t st) {
    const TrtriPlan* pl = pp->pl;
    const TrtriPhase& ph = pl->phases[p];
    if (ph.diag_hi > ph.diag_lo)
        GPRY_TRY(launch_trtri_diag_range(ctx, pp->L, pp->V, pp->Np, ph.diag_lo, ph.diag_hi - ph.diag_lo, st));
    for (const TrtriPhaseLevel& pv : ph.levels)
        GPRY_TRY(trtri_level_products(ctx, pp->L, pp->V, pp->T, pp->Np, trtri_level_nsplit(ctx, pl, (size_t)pv.lev),
                                      pl->aligned[pv.lev], pl->d_phase_items + pv.t_first, pv.n_t, pv.mM_t, pv.mN_t,
                                      pl->d_phase_items + pv.v_first, pv.n_v, pv.mM_v, pv.mN_v, st,
                                      pl->seg_t[pv.lev] > 0 ? &pv.sk_t : nullptr, pl->seg_t[pv.lev] > 0 ? &pv.sk_v : nullptr));
    return 0;
}
// returns 1 when the chain is not cut for this size (the caller runs trtri_lower after potrf)
int trtri_pipeline_begin(gpry_ctx* ctx, const double* L, double* V, double* T, int64_t Np) {
    TrtriPipe* pp = pipe_of(ctx);
    pp->active = false;
    if (!ctx->stream2) return 1;
    TrtriPlan* pl = nullptr;
    GPRY_TRY(trtri_plan_get(ctx, Np, &pl));
    if (pl->phases.size() < 2 || pl->leaf != 1) return 1;      // (the phases are cut in 64-row diagonal blocks)
    while (ctx->ev_pool.size() < pl->phases.size() + 2) {
        hipEvent_t ev;
        HIP_TRY(ctx, hipEventCreateWithFlags(&ev, hipEventDisableTiming));
        ctx->ev_pool.push_back(ev);
    }
    // the split-K scratch is shared by all phases: size it before anything is queued
    int maxsplit = 1;
    for (size_t lev = 0; lev < pl->count.size(); lev++) {
        const int ns = trtri_level_nsplit(ctx, pl, lev);
        if (ns > maxsplit) maxsplit = ns;
    }
    if (Np <= ctx->opt_gemm_streamk) {
        for (auto& q : pl->sk_t) if (q.max_slices > maxsplit) maxsplit = q.max_slices;
        for (auto& q : pl->sk_v) if (q.max_slices > maxsplit) maxsplit = q.max_slices;
        for (auto& ph : pl->phases) for (auto& pv : ph.levels) {
            if (pv.sk_t.max_slices > maxsplit) maxsplit = pv.sk_t.max_slices;
            if (pv.sk_v.max_slices > maxsplit) maxsplit = pv.sk_v.max_slices;
        }
    }
    double* sbuf = nullptr;
    if (maxsplit > 1) GPRY_TRY(gemm_split_scratch(ctx, maxsplit, Np * Np, &sbuf));
    pp->L = L; pp->V = V; pp->T = T; pp->Np = Np; pp->pl = pl; pp->next = 0; pp->active = true;
    pp->side = ctx->stream2;
    // V may still be read by work queued earlier on the main stream
    hipEvent_t ev0 = ctx->ev_pool[pl->phases.size()];
    HIP_TRY(ctx, hipEventRecord(ev0, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(pp->side, ev0, 0));
    HIP_TRY(ctx, hipMemsetAsync(V, 0, sizeof(double) * Np * Np, pp->side));
    return 0;
}
// called by the potrf loops after the launch that makes 64-column block `blocks_done - 1` of L final
int trtri_pipeline_step(gpry_ctx* ctx, int blocks_done) {
    TrtriPipe* pp = static_cast<TrtriPipe*>(ctx->trtri_pipe);
    if (!pp || !pp->active) return 0;
    const TrtriPlan* pl = pp->pl;
    if (pp->next + 1 >= pl->phases.size() || pl->phases[pp->next].blocks_done != blocks_done) return 0;
    hipEvent_t ev = ctx->ev_pool[pp->next];
    HIP_TRY(ctx, hipEventRecord(ev, ctx->stream));
    HIP_TRY(ctx, hipStreamWaitEvent(pp->side, ev, 0));
    GPRY_TRY(trtri_phase_run(ctx, pp, pp->next, pp->side));
    pp->next++;
    return 0;
}
int trtri_pipeline_finish(gpry_ctx* ctx) {
    TrtriPipe* pp = static_cast<TrtriPipe*>(ctx->trtri_pipe);
    if (!pp || !pp->active) return gpry_fail(ctx, -1, "trtri_pipeline_finish without begin");
    pp->active = false;
    const TrtriPlan* pl = pp->pl;
    // phases the potrf loop did not reach (it returned early): run them now, in order
    hipEvent_t evd = ctx->ev_pool[pl->phases.size() + 1];
    HIP_TRY(ctx, hipEventRecord(evd, pp->side));
    HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, evd, 0));
    for (; pp->next < pl->phases.size(); pp->next++) GPRY_TRY(trtri_phase_run(ctx, pp, pp->next, ctx->stream));
    return 0;
}

// error path of the chain (build_factor): whatever the side stream still has queued on V / T / the split-K scratch
// is waited for, so that the caller's next operation on those buffers cannot race with it
void trtri_pipeline_abort(gpry_ctx* ctx) {
    TrtriPipe* pp = static_cast<TrtriPipe*>(ctx->trtri_pipe);
    if (!pp || !pp->active) return;
    pp->active = false;
    if (pp->side) (void)hipStreamSynchronize(pp->side);
}

// K^-1 = V^T V, lower triangle only (the traces kernel reads Kinv[max(i,j)][min(i,j)]).
int lauum_lower(gpry_ctx* ctx, const double* V, double* Kinv, int64_t Np) {
    // Stream-K (gemm_dma.hip) up to Np = gemm_streamk (default 5632), i.e. while the launch has few tiles per
    // workgroup slot: 509 -> 430 us at Np = 4096 (528 tiles), 871 -> 793 at 5120, but 1243 -> 1300 us at 6144 (1176
    // tiles) and 2830 -> 3000 us at 8192 (2080), where one workgroup per tile already fills the GPU for several
    // rounds and the longest-first tile order keeps the tail short (tools/ab_factor_pipeline.py; giving every XCD a
    // contiguous run of segments instead of every eighth changes nothing).  The plan lives with the V = L^-1 plan.
    if (Np >= 512 && Np <= ctx->opt_gemm_streamk && !ctx->tp) {
        TrtriPlan* pl = nullptr;
        GPRY_TRY(trtri_plan_get(ctx, Np, &pl));
        GemmArgs g = {};
        g.A = V; g.lda = Np; g.B = V; g.ldb = Np; g.C = Kinv; g.ldc = Np;
        g.M = (int)Np; g.N = (int)Np; g.K = (int)Np;
        g.kmode = KM_AT_LOWER_B_LOWER; g.lower_only = 1; g.info = ctx->dinfo;
        return gemm_dma_parts_launch(ctx, g, true, false, EPI_STORE, pl->sk_lauum, Np * Np);
    }
    GemmArgs g = {};
    g.A = V; g.lda = Np; g.B = V; g.ldb = Np; g.C = Kinv; g.ldc = Np;
    g.M = (int)Np; g.N = (int)Np; g.K = (int)Np;
    g.kmode = KM_AT_LOWER_B_LOWER; g.lower_only = 1; g.tile_map = TM_ROWMAJOR; g.info = ctx->dinfo;
    // tile (i, j) sums over k >= max(i, j) * 128: the first tiles run the whole K and are the critical
    // path of the launch.  Splitting them pays up to about two rounds of workgroups (Np = 2048: x4,
    // 332 -> 130 us with the balanced tile map; Np = 4096, 528 tiles: x2, 0.63 -> 0.51 ms); at
    // Np = 8192 (2080 tiles) it costs 5 %.
    const int64_t tiles = (Np / 128) * (Np / 128 + 1) / 2;
    int nsplit = 1;
    while (!ctx->tp && nsplit < 4 && tiles * nsplit * 2 <= 1100 && Np / (nsplit * 2) >= 256) nsplit *= 2;
    if (nsplit > 1) {
        double* sbuf = nullptr;
        GPRY_TRY(gemm_split_scratch(ctx, nsplit, Np * Np, &sbuf));
        g.nsplit = nsplit; g.split_buf = sbuf; g.split_stride = Np * Np;
    }
    g.small64 = 1;          // Np is a multiple of 128
    return gemm_f64_launch(ctx, g, true, false, EPI_STORE);
}

// Split-K slices (of Np x Np doubles each) that V = L^-1 followed by K^-1 = V^T V ask of gemm_split_scratch at this size:
// what a batched evaluation reserves per theta before the first launch (its arena is never re-allocated under way).
int factor_chain_slices(gpry_ctx* ctx, int64_t Np, int* slices) {
    if (ctx->tp) { *slices = 0; return 0; }     // throughput schedule: no partial slices anywhere
    TrtriPlan* pl = nullptr;
    GPRY_TRY(trtri_plan_get(ctx, Np, &pl));
    int m = 1;
    for (size_t lev = 0; lev < pl->count.size(); lev++) {
        if (pl->seg_t[lev] > 0 && Np <= ctx->opt_gemm_streamk) {
            if (pl->sk_t[lev].max_slices > m) m = pl->sk_t[lev].max_slices;
            if (pl->sk_v[lev].max_slices > m) m = pl->sk_v[lev].max_slices;
        } else {
            const int ns = trtri_level_nsplit(ctx, pl, lev);
            if (ns > m) m = ns;
        }
    }
    if (Np >= 512 && Np <= ctx->opt_gemm_streamk) {
        if (pl->sk_lauum.max_slices > m) m = pl->sk_lauum.max_slices;
    } else {
        const int64_t tiles = (Np / 128) * (Np / 128 + 1) / 2;
        int nsplit = 1;
        while (nsplit < 4 && tiles * nsplit * 2 <= 1100 && Np / (nsplit * 2) >= 256) nsplit *= 2;
        if (nsplit > m) m = nsplit;
    }
    *slices = m > 1 ? m : 0;
    return 0;
}

// ------------------------------------------------------------------------------------
// z = V y : one wave per row (row-contiguous, coalesced), fixed reduction tree.
// (batched launches, gpry_ctx::bn: V, z, part, alpha of theta blockIdx.z lie bstride doubles further on; y is shared)
__global__ __launch_bounds__(256) void trmv_lower_kernel(const double* __restrict__ V_, int64_t ld,
                                                         const double* __restrict__ y,
                                                         double* __restrict__ z_, int64_t n, int64_t bstride) {
    const double* __restrict__ V = bset(V_, (int)blockIdx.z, bstride);
    double* __restrict__ z = bset(z_, (int)blockIdx.z, bstride);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    double s = 0.0;
    // eight loads in flight per step, the multiply-adds in the order they had (one dependent load per multiply-add was a chain
    // of up to Np / 64 memory latencies: 27 us at N = 4096, 37 us for 32 thetas at N = 1024); terms right of the diagonal enter
    // as 0 * 0 behind the last real one, which leaves the sum's bits alone
    for (int64_t k0 = lane; k0 <= row; k0 += 512) {
        double v[8], yy[8];
#pragma unroll
        for (int u = 0; u < 8; u++) {
            const int64_t k = k0 + 64 * u;
            const bool in = k <= row;
            v[u] = in ? V[row * ld + k] : 0.0;
            yy[u] = in ? y[k] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; u++) s = fma(v[u], yy[u], s);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    if (lane == 0) z[row] = s;
}
// partial column sums of V^T z over 256-row chunks: part[chunk][col]
__global__ __launch_bounds__(256) void trmv_lower_t_kernel(const double* __restrict__ V_, int64_t ld,
                                                           const double* __restrict__ z_,
                                                           double* __restrict__ part_, int64_t n, int64_t bstride) {
    __shared__ double red[4][64];
    const double* __restrict__ V = bset(V_, (int)blockIdx.z, bstride);
    const double* __restrict__ z = bset(z_, (int)blockIdx.z, bstride);
    double* __restrict__ part = bset(part_, (int)blockIdx.z, bstride);
    const int c = threadIdx.x & 63, rq = threadIdx.x >> 6;
    const int64_t col = (int64_t)blockIdx.x * 64 + c;
    const int64_t r0 = (int64_t)blockIdx.y * 256;
    double s = 0.0;
    if (r0 + 255 >= (int64_t)blockIdx.x * 64) {
        // eight loads in flight per step (one dependent load per multiply-add made this a 64-deep latency chain:
        // 16 us whatever the size); terms above the diagonal enter as 0 * 0, which leaves the sum's bits alone
        for (int q0 = 0; q0 < 64; q0 += 8) {
            double v[8], zz[8];
#pragma unroll
            for (int u = 0; u < 8; u++) {
                const int64_t r = r0 + rq + 4 * (q0 + u);
                const bool in = r >= col && r < n;
                v[u] = in ? V[r * ld + col] : 0.0;
                zz[u] = in ? z[r] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 8; u++) s = fma(v[u], zz[u], s);
        }
    }
    red[rq][c] = s;
    __syncthreads();
    if (rq == 0) part[(int64_t)blockIdx.y * n + col] = (red[0][c] + red[1][c]) + (red[2][c] + red[3][c]);
}
__global__ void colsum_kernel(const double* __restrict__ part_, int64_t n, int nchunk, double* __restrict__ out_, int64_t bstride) {
    const double* __restrict__ part = bset(part_, (int)blockIdx.z, bstride);
    double* __restrict__ out = bset(out_, (int)blockIdx.z, bstride);
    int64_t col = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (col >= n) return;
    double s = 0.0;
    for (int ch = 0; ch < nchunk; ch++) s += part[(int64_t)ch * n + col];
    out[col] = s;
}

int solve_alpha(gpry_ctx* ctx, const double* V, const double* y, double* z, double* alpha, int64_t Np) {
    hipStream_t st = ctx->stream;
    const unsigned bn = (unsigned)ctx->bn;
    hipLaunchKernelGGL(trmv_lower_kernel, dim3((unsigned)((Np + 3) / 4), 1, bn), dim3(256), 0, st, V, Np, y, z, Np, ctx->bstride);
    int nchunk = (int)((Np + 255) / 256);
    int64_t need = (int64_t)nchunk * Np;
    if (need > ctx->part_cap && ctx->bpar) return gpry_fail(ctx, -1, "batched chain: partial sums exceed the arena");
    if (need > ctx->part_cap) {
        if (ctx->dpart) dev_free(ctx, ctx->dpart);
        GPRY_TRY(dev_alloc(ctx, &ctx->dpart, need));
        ctx->part_cap = need;
    }
    hipLaunchKernelGGL(trmv_lower_t_kernel, dim3((unsigned)(Np / 64), (unsigned)nchunk, bn), dim3(256), 0, st,
                       V, Np, z, ctx->dpart, Np, ctx->bstride);
    hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)((Np + 255) / 256), 1, bn), dim3(256), 0, st, ctx->dpart, Np,
                       nchunk, alpha, ctx->bstride);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// out[0] = sum_i log L_ii (i < n_real), out[1] = sum_i z_i^2   (single workgroup)
__global__ __launch_bounds__(1024) void logdet_quad_kernel(const double* __restrict__ L_, int64_t ld,
                                                           const double* __restrict__ z_,
                                                           int64_t n_real, double* __restrict__ out_,
                                                           const int* __restrict__ info_, double* __restrict__ host_res,
                                                           int info_at, int64_t bstride) {
    __shared__ double r0[1024], r1[1024];
    const int tb = (int)blockIdx.z;             // theta of a batched launch: its results go to their own row of host_res
    const double* __restrict__ L = bset(L_, tb, bstride);
    const double* __restrict__ z = bset(z_, tb, bstride);
    double* __restrict__ out = bset(out_, tb, bstride);
    const int* __restrict__ info = bset(info_, tb, bstride);
    if (host_res) host_res += (int64_t)tb * GPRY_BRES_STRIDE;
    const int t = threadIdx.x;
    double a = 0.0, b = 0.0;
    for (int64_t i = t; i < n_real; i += 1024) { a += log(L[i * ld + i]); b = fma(z[i], z[i], b); }
    r0[t] = a; r1[t] = b;
    __syncthreads();
    for (int s = 512; s >= 1; s >>= 1) {
        if (t < s) { r0[t] += r0[t + s]; r1[t] += r1[t + s]; }
        __syncthreads();
    }
    if (t == 0) {
        out[0] = r0[0]; out[1] = r1[0];
        // last kernel of a value-only evaluation: results and factorisation status also go straight into the mapped
        // host buffer (read by the host after the stream wait)
        if (host_res) {
            host_res[0] = r0[0]; host_res[1] = r1[0];
            host_res[info_at] = (double)info[0]; host_res[info_at + 1] = (double)info[1];
                host_res[info_at + 2] = (double)info[3];       // (0x5A..: a bounded wait of the panel step ran out -- not a verdict on the matrix)
        }
    }
}

int logdet_and_quad(gpry_ctx* ctx, const double* L, const double* z, int64_t Np, double* out2_dev, double* host_res,
                    int info_at) {
    (void)Np;
    hipLaunchKernelGGL(logdet_quad_kernel, dim3(1, 1, (unsigned)ctx->bn), dim3(1024), 0, ctx->stream, L, ctx->Np, z, ctx->N, out2_dev,
                       ctx->dinfo, host_res, info_at, ctx->bstride);
    HIP_TRY(ctx, hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------
// rocSOLVER comparator (option "chol"=1), loaded lazily so the default path has no
// dependency on rocBLAS/rocSOLVER.  Row-major lower == column-major upper.
typedef void* rb_handle;
static struct {
    bool tried = false, ok = false;
    void *lib_blas = nullptr, *lib_solver = nullptr;
    int (*create)(rb_handle*) = nullptr;
    int (*set_stream)(rb_handle, hipStream_t) = nullptr;
    int (*dpotrf)(rb_handle, int, int, double*, int, int*) = nullptr;
    int (*dtrtri)(rb_handle, int, int, int, double*, int, int*) = nullptr;
    rb_handle handle = nullptr;
} g_rs;

__global__ void zero_upper_kernel(double* A, int64_t n) {
    int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n * n) return;
    int64_t i = idx / n, j = idx - i * n;
    if (j > i) A[idx] = 0.0;
}

int rocsolver_potrf_trtri(gpry_ctx* ctx, double* A, double* V, int64_t Np, int want_v) {
    if (!g_rs.tried) {
        g_rs.tried = true;
        g_rs.lib_blas = dlopen("librocblas.so", RTLD_NOW | RTLD_GLOBAL);
        g_rs.lib_solver = dlopen("librocsolver.so", RTLD_NOW | RTLD_GLOBAL);
        if (g_rs.lib_blas && g_rs.lib_solver) {
            g_rs.create = (int (*)(rb_handle*))dlsym(g_rs.lib_blas, "rocblas_create_handle");
            g_rs.set_stream = (int (*)(rb_handle, hipStream_t))dlsym(g_rs.lib_blas, "rocblas_set_stream");
            g_rs.dpotrf = (int (*)(rb_handle, int, int, double*, int, int*))dlsym(g_rs.lib_solver, "rocsolver_dpotrf");
            g_rs.dtrtri = (int (*)(rb_handle, int, int, int, double*, int, int*))dlsym(g_rs.lib_solver, "rocsolver_dtrtri");
            g_rs.ok = g_rs.create && g_rs.set_stream && g_rs.dpotrf && g_rs.dtrtri;
            if (g_rs.ok && g_rs.create(&g_rs.handle) != 0) g_rs.ok = false;
        }
    }
    if (!g_rs.ok) return gpry_fail(ctx, -3, "rocSOLVER comparator unavailable (dlopen/dlsym failed)");
    g_rs.set_stream(g_rs.handle, ctx->stream);
    HIP_TRY(ctx, hipMemsetAsync(ctx->dinfo, 0, 2 * sizeof(int), ctx->stream));
    const int rocblas_fill_upper = 121, rocblas_diagonal_non_unit = 131;
    if (g_rs.dpotrf(g_rs.handle, rocblas_fill_upper, (int)Np, A, (int)Np, ctx->dinfo) != 0)
        return gpry_fail(ctx, -3, "rocsolver_dpotrf failed");
    if (want_v) {
        HIP_TRY(ctx, hipMemcpyAsync(V, A, sizeof(double) * Np * Np, hipMemcpyDeviceToDevice, ctx->stream));
        hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)((Np * Np + 255) / 256)), dim3(256), 0, ctx->stream, V, Np);
        if (g_rs.dtrtri(g_rs.handle, rocblas_fill_upper, rocblas_diagonal_non_unit, (int)Np, V, (int)Np, ctx->dinfo + 1) != 0)
            return gpry_fail(ctx, -3, "rocsolver_dtrtri failed");
    }
    return 0;
}
