// Internal definitions shared by the gfx950 kernels behind include/gpry_hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <math.h>
#include <string>
#include <vector>
#include <map>

#include "../../include/gpry_hip.h"

#define GPRY_TILE 128       // GEMM workgroup tile and padding quantum of N
#define GPRY_NB 64          // Cholesky / trtri base block

typedef double v4d __attribute__((ext_vector_type(4)));

struct StageTimer {
    double total_ms = 0.0;
    int64_t count = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
};

struct gpry_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;          // V = L^-1 phases of the pipelined factor chain run here, underneath potrf
    std::vector<hipEvent_t> ev_pool;        // one pair per Cholesky look-ahead step (never re-recorded within a call)
    char err[1024] = {0};

    // options
    int opt_chol = 0;
    int64_t opt_sweep_chunk = 0;         // 0: 32768 at Np >= 4096, proportionally more below (same panel bytes)
    int opt_timing = 0;          // per-stage HIP-event timers: off until gpry_timing_reset (or "timing" = 1) asks for them
    bool xs_foreign = false;     // dXs holds coordinates scaled for an LML evaluation's theta, not the prediction factor's (ensure_pred_xs)
    bool info_cleared = false;   // dinfo[0..3] were zeroed by launch_scale_train and nothing has touched them since
    int opt_gemm_small = 32;     // launches of at most this many 128 x 128 tiles run with 64 x 64 tiles (0: never)
    int64_t opt_predict_small = 2048;  // mean-only gpry_predict of at most this many points: one fused launch
    int opt_chol_overlap = 1;    // 1: trailing-update tiles ride in the panel launches (potrf_lower_overlap)

    // training set (transformed space)
    int64_t N = 0, Np = 0, cap = 0;  // cap: allocated padded size
    int d = 0, dpad = 0, dp_cap = 0;   // dp_cap: row width the X buffers were allocated for
    int kernel_id = GPRY_RBF;
    bool have_theta = false, factor_valid = false;
    double theta[1 + GPRY_MAX_DIM] = {0};
    gpry_affine tf;
    // factor of the last successful gpry_lml evaluation: still sitting in dW (L) / dW2 (V)
    bool lml_cache = false;
    int lml_kernel_id = -1;
    double lml_theta[1 + GPRY_MAX_DIM] = {0};
    int opt_lml_cache = 1;
    unsigned long long lml_seq = 0;     // stamp of the last single-launch evaluation
    int opt_lml_small = 1;       // N <= 128, d <= 16: LML + gradient in one launch of one workgroup (lml_small.hip)

    // Batched objective above N = 128 (gpry_lml_batch, api.hip): the B thetas of a call go through ONE chain of launches.
    // Every kernel of the chain takes the theta index from blockIdx.z; the per-theta buffers (W, W2, W3, scaled
    // coordinates, vectors, partial sums, split-K slices, [C, l...], status words) are the same layout repeated
    // `bstride` doubles apart in one arena, so a kernel moves every per-theta pointer by the same number of bytes
    // (bset below).  Outside a batched evaluation bn = 1, bstride = 0, bpar = NULL and nothing changes.
    int bn = 1;
    int64_t bstride = 0;
    const double* bpar = nullptr;      // device, set 0: [C, l_1 .. l_d] as the host computes them for one evaluation
    double* barena = nullptr; int64_t barena_cap = 0;     // doubles
    void* hbres = nullptr; void* hbres_dev = nullptr; int64_t hbres_cap = 0;   // results of a batch (pinned, device-mapped)
    int64_t opt_lml_batch = 4096;      // largest Np whose gpry_lml_batch runs as one chain (0: thetas one after another); measured
                                       // against the thread farm of three contexts: full fits 2.2x faster at N = 1024, 1.4x at 2048,
                                       // 1.1x at 4096, 0.9x at 8192 (tools/r04/time_fit_crossover.py, bench.py --workload farm)
    int64_t opt_lml_batch_mb = 49152;  // upper limit of the arena (MiB of the 288 GB): longer batches go through in chunks
    // Schedule of gpry_lml_batch (option "lml_schedule"): 0 = latency -- every theta gets the launches, and the bits, of a single
    // gpry_lml (stream-K / split-K partial sums, the inverse factor out of the Cholesky launches up to "chol_stacked"); 1 =
    // throughput -- the chain for MANY thetas at once: whole-tile products only (no partial slices to write and add), the
    // Cholesky in column blocks with one MFMA SYRK launch behind each, the thetas dealt over "lml_streams" streams so that one
    // group's panel chain runs underneath the other groups' products.  Its per-theta result does not depend on how many thetas
    // share the call (B = 1 included) and differs from the latency schedule's by rounding (another summation order).
    int opt_lml_schedule = 0;
    int opt_lml_streams = 2;           // throughput schedule: stream groups per call (1: one chain for all thetas)
    int64_t opt_tp_block = 512;        // throughput schedule: width of the column blocks of the Cholesky (multiple of 128)
    int opt_tp_left = 0;               // throughput schedule: 0 = the column blocks right-looking (one launch of the block's width behind each; default: measured ahead), 1 = left-looking (one deep launch in front of each); same bits
    int64_t opt_tp_tail = 1024;        // ... and the size of the last block, factored with riding tiles only
    int64_t batch_shrinks = 0;         // times a batched evaluation halved its chunk after an out-of-memory answer (diagnostic, gpry_timing_get "lml_batch_shrinks")
    int opt_chol_tp_segments = 0;      // 1: every factorisation of the context takes the column blocks of the throughput schedule (comparator: same bits)
    int opt_panel_debug = 0;           // ORed into GPRY_PANEL_FLAGS (chol_panel.hip); 64: the diagonal workgroup of every panel step reports a timed-out wait (test hook)
    bool tp = false;                   // a throughput-schedule chain is being queued (set by lml_batch_general only)
    std::vector<hipStream_t> tp_streams;   // the extra streams of the groups (created on first use)
    std::vector<hipEvent_t> tp_events;

    double* dX = nullptr;      // N x d raw transformed training rows (row-major, ld = d)
    double* dXs = nullptr;     // Np x dpad rows scaled by 1/l (pad rows = 0)
    double* dy = nullptr;      // Np
    double* dnoise = nullptr;  // Np (alpha = noise_^2; pad = 0)
    double* dA = nullptr;      // Np x Np : K then L (factor used for prediction)
    double* dV = nullptr;      // Np x Np : V = L^-1 (upper triangle exactly zero)
    double* dW = nullptr;      // Np x Np : scratch (lml: K/L ; T ; K^-1)
    double* dW2 = nullptr;     // Np x Np : scratch (lml: V)
    double* dW3 = nullptr;     // Np x Np : scratch (lml: T, then K^-1)
    double* dalpha_ = nullptr; // Np
    double* dvec = nullptr;    // small vectors / reductions (8 * Np + 4096 doubles)
    int* dinfo = nullptr;      // device status word(s)
    double* dparams = nullptr; // device copy of [C, 1/l..., lo..., span...] etc.

    // gates evaluated on the device inside gpry_sweep_logexp (gpry_set_gates)
    double* gate_sv = nullptr; double* gate_coef = nullptr; double* gate_trust = nullptr;
    int64_t gate_nsv = 0, gate_sv_cap = 0;
    double gate_gamma = 0.0, gate_intercept = 0.0;
    int gate_positive_finite = 1, gate_has_trust = 0, gates_on = 0;

    // sweep state
    int64_t sw_M = 0, sw_cap = 0;
    double* dXc = nullptr;     // M x d candidates (raw as given)
    uint8_t* dmask = nullptr;  // M
    double *dy_all = nullptr, *dsig_all = nullptr, *dacq_all = nullptr;  // M each
    // second candidate set, used by gpry_predict so that the resident NORA pool survives
    struct CandSet {
        int64_t M = 0, cap = 0;
        double* dXc = nullptr; uint8_t* dmask = nullptr;
        double *dy = nullptr, *dsig = nullptr, *dacq = nullptr;
    } pr;
    double* dXcs = nullptr;    // dsel x chunk: the candidates of the current chunk, scaled, coordinate-major (launch_cross_build)
    int64_t xcs_cap = 0;
    double* dYcs = nullptr;    // Np x dsel: centred scaled training rows of the MFMA panel build (launch_cross_prepare)
    int64_t ycs_cap = 0;
    double xcenter[GPRY_MAX_DIM] = {0};    // mean of the training rows per dimension: the centre both sides are shifted by
    double xsum[GPRY_MAX_DIM] = {0};       // its running sums, in row order (set_train, append_rows)
    double xlo[GPRY_MAX_DIM] = {0}, xhi[GPRY_MAX_DIM] = {0};     // smallest / largest training coordinate per dimension (error estimate of the MFMA panel)
    int opt_cross_mfma = 1;    // 1 (default): the sweep's cross-kernel panel takes its distances from the matrix pipe
    double alpha_l2 = -1.0;    // ||alpha_||_2 of the prediction factor (fetched when the panel form is chosen; < 0: not yet)
    double alpha_l1 = 0.0;     // ||alpha_||_1, fetched with it
    double noise_min = 0.0;    // smallest entry of the noise vector `alpha` of the training rows (set_train, append_rows): lambda_min(K) >= it
    int panel_form = 0;        // cross-kernel panel of the last sweep / panel predict: 0 none yet, 1 matrix pipe, 2 difference form, 3 small-batch kernel, 4 hybrid (matrix pipe, near pairs from the coordinates)
    int opt_cross_hybrid = 1;  // 1: a model that fails the gate of the matrix-pipe form because of SHORT length scales takes the hybrid form instead of the difference form
    double panel_est[4] = {0, 0, 0, 2.5e-7};   // its error estimates: mean (l2, gated), mean (l1, worst case), variance / C (gated), the gate
    double* dKst = nullptr;    // Np x chunk cross-kernel panel (k-major)
    int64_t kst_cap = 0;       // doubles allocated
    double* dG = nullptr;      // Np x dpad: d k(x, X_j)/dx of the last gpry_predict_grad
    int64_t g_cap = 0;
    double* dsplit = nullptr;  // split-K partial products of the factor GEMMs
    int64_t split_cap = 0;
    int opt_predict_split = 1; // split-K contraction for predict / sweeps of a few thousand points (one chunk)
    int opt_gemm_dma = 1;         // 128-aligned factor-chain products through gemm_dma_kernel
    double* dpart = nullptr;   // partial sums (sumsq per i-tile, mean per j-chunk)
    int64_t part_cap = 0;
    // top-k scratch
    unsigned long long* dkeys = nullptr; int64_t keys_cap = 0;
    unsigned int* dhist = nullptr;
    gpry_cand* dcand = nullptr; int64_t cand_cap = 0;
    unsigned long long* dsel = nullptr;  // select state

    // Kriging-believer session
    int64_t kb_n = 0, kb_cap = 0;
    int64_t kb_ld = 0;         // row length dU was allocated for (Np may grow within ctx->cap: set_train, gpry_append_rows)
    double* dU = nullptr;      // kb_cap x kb_ld : row x = u(x)^T, used with stride Np <= kb_ld (x-major, contiguous u)
    double* dXkb = nullptr;    // kb_cap x dpad scaled candidate rows
    double* dkbout = nullptr;  // 2 * kb_cap

    double* dbord = nullptr;   // border workspace of gpry_append_rows: B, U (Np x 128), T (128 x Np), S / L22 / W22
    int64_t bord_cap = 0;
    int64_t n_border = 0;      // rows appended by border updates since the last full factorisation (diagnostic)

    void* trtri_plan = nullptr;
    void* trtri_pipe = nullptr;   // state of a pipelined factor chain in flight (chol.hip)
    int64_t opt_gemm_streamk = 5632;   // up to this Np the V = L^-1 levels >= 512 and K^-1 = V^T V are stream-K launches
                                       // (gemm_dma.hip; a gain up to 5120, neutral at 6144, a loss at 8192); 0 = off
    int64_t opt_topk_host = 16384;    // pools up to this size are selected on the host (gpry_sweep_topk)
    int opt_factor_pipeline = 1;  // 1: V = L^-1 phases run on stream2 underneath potrf
    int opt_factor_pipeline_min = 1280;   // from this Np on (ahead by 2-6% from 1280 up, level below: tools/r04/ab_pipeline_now.py)
    void* chol_plan = nullptr;    // cached tile schedule of the fused Cholesky (chol_panel.hip)   // cached batch descriptors of the V = L^-1 recursion (chol.hip)

    // resident predict kernel (server.hip)
    void* srv = nullptr;
    int opt_predict_serve = 1;         // mean-only gpry_predict of <= 8 points goes through the resident kernel
    int64_t opt_serve_idle_us = 2000;  // the kernel leaves after this long without a request
    int64_t opt_chol_stacked = 2048;   // up to this Np the inverse factor comes out of the Cholesky launches themselves (potrf_stacked, chol_panel.hip); 0: never
    int opt_chol_stacked_dense = 0;    // 1: potrf_stacked without use of the zero structure of the appended rows (comparator)
    int opt_sweep_overlap = 0;         // 1: the cross-kernel panel of chunk c + 1 is built on the side stream underneath the contraction of chunk c (two panels)
    int opt_sweep_upload = 1;          // 1: a fresh candidate pool is uploaded chunk by chunk on stream2, chunk c + 1 underneath the kernels of chunk c
    const double* up_X = nullptr;      // host pool of the sweep in flight whose chunks are still to be uploaded (run_sweep)
    int up_gates = 0;                  // ... and the device gates are evaluated chunk by chunk behind each upload
    int opt_predict_gates = 0;         // 1: gpry_predict ORs the device gates (gpry_set_gates) into the caller's mask, as the sweep does

    // host pinned staging
    void* hpin = nullptr; void* hpin_dev = nullptr; int64_t hpin_cap = 0;   // host / device view of the same buffer

    std::map<std::string, StageTimer> timers;
};

extern thread_local char g_last_error[1024];

int gpry_fail(gpry_ctx* ctx, int code, const char* fmt, ...);

#define HIP_TRY(ctx, expr)                                                          \
    do {                                                                            \
        hipError_t _e = (expr);                                                     \
        if (_e != hipSuccess)                                                       \
            return gpry_fail(ctx, -2, "%s failed: %s (%s:%d)", #expr,               \
                             hipGetErrorString(_e), __FILE__, __LINE__);            \
    } while (0)

#define GPRY_TRY(expr)                 \
    do {                               \
        int _r = (expr);               \
        if (_r != 0) return _r;        \
    } while (0)

// scoped device timing of one stage on ctx->stream
struct StageScope {
    gpry_ctx* ctx; const char* name; hipEvent_t e0 = nullptr, e1 = nullptr; hipStream_t st = nullptr; bool marked = false;
    StageScope(gpry_ctx* c, const char* n, hipStream_t stream = nullptr);
    ~StageScope();
};
void timers_collect(gpry_ctx* ctx);

template <typename T>
int dev_alloc(gpry_ctx* ctx, T** p, int64_t count);
int dev_free(gpry_ctx* ctx, void* p);

static inline int64_t round_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }

// the buffer of theta `tb` in a batched launch (see gpry_ctx::bn): every per-theta pointer moves by tb * bstride doubles
// (pointer arithmetic, not integer arithmetic: the result keeps the provenance of a kernel argument, i.e. global loads
// and stores -- through an integer cast it became a flat pointer, and chol_fused_kernel went to scratch memory)
template <typename T>
__device__ __forceinline__ T* bset(T* p, int tb, int64_t bstride) {
    return (T*)((char*)p + (int64_t)tb * bstride * 8);
}
#define GPRY_BRES_STRIDE 40     // doubles per theta in the result buffer of a batch: [logdet/2, quad, grad (1 + 32), info0, info1]

// ---- GEMM (gemm_f64.hip) -----------------------------------------------------------
enum GemmKMode {
    KM_FULL = 0,      // k in [0, K)
    KM_A_LOWER = 1,   // A lower triangular (rows i, cols k): k < (ti+1)*TILE
    KM_B_LOWER = 2,   // B lower triangular (rows k, cols j): k >= tj*TILE
    KM_AT_LOWER_B_LOWER = 3, // C = A^T B with A, B lower: k >= max(ti, tj)*TILE
    KM_B_UPPER = 4,   // B upper triangular (rows k, cols j): k < (tj+1)*TILE
    KM_AT_LOWER = 5   // C = A^T B with A lower triangular (stored rows k, cols i): k >= ti*TILE
};
enum GemmEpi { EPI_STORE = 0, EPI_STORE_NEG = 1, EPI_SUB = 2, EPI_SUMSQ = 3 };
// TM_BALANCED (gemm_dma.hip): workgroups go to the XCDs round-robin by block index, so the enumeration
// decides the balance: tiles of equal k-length are neighbours (row-wise triangular order for lower-only
// square outputs, column-major where the length depends on the tile column), longest first
enum GemmTileMap { TM_ROWMAJOR = 0, TM_SWEEP = 1, TM_BALANCED = 2 };

struct GemmBatchItem { int64_t a_off, b_off, c_off; int M, N, K, pad; };

struct GemmArgs {
    const double* A; const double* B; double* C;
    int64_t lda, ldb, ldc;
    int M, N, K;
    int kmode;
    int lower_only;        // skip output tiles with tj > ti
    int tile_map;
    const GemmBatchItem* batch;  // nullable; grid.z = n_batch
    int n_batch;
    const int* info;       // nullable: if *info != 0 the kernel exits immediately
    hipStream_t stream;    // null: ctx->stream
    int nsplit;            // > 1: split-K over grid.y into split_buf (store epilogues only), then reduced
    double* split_buf; int64_t split_stride;
    int dma_ok;            // batched launches: 1 = every item meets gemm_dma_usable (checked by the caller)
    int skip_reduce;       // split-K: leave the slices in split_buf (the caller reduces them itself)
    int small64;           // 1: every M, N is a multiple of 64 and every K of 32: launches with few tiles may take gemm_small.hip
    // batched objective (filled in by the launchers from the context): grid.z = bz_div * bn, theta = blockIdx.z / bz_div,
    // A / B / C / info / split_buf of theta tb lie tb * bstride doubles behind those of theta 0
    int bn, bz_div;
    int64_t bstride;
};
// theta index and batch item of a GEMM workgroup
__device__ __forceinline__ void gemm_block_z(const GemmArgs& g, int z, int* tb, int* item) {
    if (g.bn > 1) { *tb = z / g.bz_div; *item = z - *tb * g.bz_div; }
    else { *tb = 0; *item = z; }
}
// The same for the launches of a batched evaluation (bn > 1), with the dispatch order taken apart again: workgroups start in
// the order of their linear index (x fastest, then z), and with the thetas in z the LONGEST tiles of the last theta -- the
// enumerations put long k-ranges first -- came behind all tiles of the thetas before it: a launch of K^-1 = V^T V for 16
// thetas at N = 4096 ended with a tail of up to 32 of its 187 slab units per workgroup slot.  Here the theta is the fastest
// index, then the batch item, then the tile: long tiles of ALL thetas first, the tail is made of the shortest ones; with a
// multiple of eight thetas all tiles of a theta also land on one XCD (round-robin dispatch) and share its L2.  Which
// workgroup computes a tile changes no bit.
// (With a multiple of eight thetas every theta sits on one XCD; any other count spreads them evenly.  Counting the thetas in
// eights -- surplus workgroups returning at once -- pins theta t to XCD t mod 8 for EVERY count, and was measured and dropped:
// two thetas then use two of the eight XCDs, nine load one XCD twice; a 42-restart fit at N = 4096, whose rounds run through
// all widths, went from 2.45 to 3.2 s.)
__device__ __forceinline__ bool gemm_block_order(const GemmArgs& g, int* bx, int* tb, int* item) {
    if (g.bn > 1) {
        const int lin = (int)blockIdx.x + (int)gridDim.x * (int)blockIdx.z;
        const int rest = lin / g.bn;
        *tb = lin - rest * g.bn;
        *bx = rest / g.bz_div;
        *item = rest - *bx * g.bz_div;
        return true;
    }
    *bx = (int)blockIdx.x; *tb = 0; *item = (int)blockIdx.z;
    return true;
}
static inline unsigned gemm_grid_z(const GemmArgs& g) { return (unsigned)(g.bz_div * g.bn); }
static inline void gemm_fill_batch(const gpry_ctx* ctx, GemmArgs* g) {
    g->bn = ctx->bn; g->bstride = ctx->bstride; g->bz_div = g->batch ? g->n_batch : 1;
}
// ---- stream-K launches of the DMA engine (gemm_dma.hip): the (tile, k) space of a launch is cut into segments of
// equal length, one workgroup per segment; a segment is a list of parts (tile, share of the tile's k-range).
struct GemmPart { int bz, ti, tj, lo, hi, slice; };     // lo / hi: slab pairs (32 k) within the tile's own k-range
struct GemmRedTile { int bz, ti, tj, nslice; };         // tiles whose product arrives in nslice >= 2 partial slices
struct GemmPartsPlan {
    GemmPart* d_parts = nullptr; int* d_first = nullptr; GemmRedTile* d_red = nullptr;
    int nwg = 0, nred = 0, max_slices = 0;
};
struct GemmShape { int M, N, K; };
// segment length (slab pairs) that gives `slots` segments over all tiles of the shapes
int gemm_parts_segment(int kmode, int lower_only, const std::vector<GemmShape>& shapes, int slots);
// plan for one launch (items = batch items in order; a non-batched launch has one); frees nothing on failure
int gemm_parts_plan_build(gpry_ctx* ctx, int kmode, int lower_only, const std::vector<GemmShape>& items, int seg_pairs,
                          GemmPartsPlan* out);
void gemm_parts_plan_free(GemmPartsPlan* pl);
int gemm_dma_parts_launch(gpry_ctx* ctx, const GemmArgs& g, bool a_trans, bool b_trans, int epi, const GemmPartsPlan& pl,
                          int64_t slice_stride);
// a_trans: A(i,k) stored at A[k*lda + i]; b_trans: B(k,j) stored at B[j*ldb + k]
int gemm_f64_launch(gpry_ctx* ctx, const GemmArgs& g, bool a_trans, bool b_trans, int epi);
// 64 x 64 tiles for launches with a handful of 128 x 128 tiles (gemm_small.hip); same bits
int gemm64_launch(gpry_ctx* ctx, const GemmArgs& g, bool a_trans, bool b_trans, int epi);
int gemm_split_scratch(gpry_ctx* ctx, int nsplit, int64_t slice, double** buf);
// gemm_dma.hip: LDS-DMA staged, software-pipelined variant for 128-aligned products (NN, NT, TN)
int launch_trtri_diag(gpry_ctx* ctx, const double* L, double* V, int64_t Np, hipStream_t st, bool zero_right = false);   // chol_panel.hip
// re-scales the training coordinates for the prediction factor if the last LML evaluation left its own in dXs
int ensure_pred_xs(gpry_ctx* ctx);   // kernel_build.hip
int launch_point_full(gpry_ctx* ctx, const double* x, int want_kinv, double* kstar, double* G, double* u, double* part,
                      double* mean_part, double* ss_part, double* out);
int launch_trtri_diag128(gpry_ctx* ctx, const double* L, double* V, int64_t Np, hipStream_t st, bool clear_right);   // trtri_small.hip
int launch_trtri_diag_range(gpry_ctx* ctx, const double* L, double* V, int64_t Np, int blk0, int nblk, hipStream_t st);
bool gemm_dma_usable(const GemmArgs& g, int M, int N, int K);
int gemm_dma_launch_product(gpry_ctx* ctx, const GemmArgs& g, bool a_trans, bool b_trans, int epi, dim3 grid);
int sweep_gemm_dma_sp_launch(gpry_ctx* ctx, const GemmArgs& g);  // sweep_gemm.hip: LDS-DMA staging + explicit software pipeline

// ---- kernel_build.hip --------------------------------------------------------------
int upload_params(gpry_ctx* ctx, const double* theta);
int launch_scale_train(gpry_ctx* ctx);                         // dXs from dX and theta
int launch_kernel_train(gpry_ctx* ctx, double* K, int add_noise, double* U = nullptr); // full symmetric K; U (nullable): the identity's tiles on and right of the diagonal (potrf_stacked)
int launch_kernel_rows(gpry_ctx* ctx, int64_t row0, int k, int64_t ldk, double* Bk, double* Cb);   // border rows of K
int launch_cross_build(gpry_ctx* ctx, const double* Xc, int64_t m0, int64_t mc,
                       int64_t ldk, double* Kst, double* mean_part, int raw_affine,
                       hipStream_t st = nullptr);
int launch_cross_prepare(gpry_ctx* ctx);               // centred scaled training rows for ...
int launch_cross_build_mfma(gpry_ctx* ctx, const double* Xc, int64_t m0, int64_t mc, int64_t ldk, double* Kst, double* mean_part,
                            int raw_affine, int hybrid = 0);           // ... the panel with MFMA distances (sweep, large predict batches); hybrid: near pairs from the coordinates
int launch_cross_build_small(gpry_ctx* ctx, const double* Xc, int64_t m0, int64_t mc, int64_t ldk,
                             double* Kst, double* mean_part, int raw_affine);     // 4 x (Np/128) mean partials
int launch_predict_mean_small(gpry_ctx* ctx, const double* Xc, int64_t M, int nsplit, double* part_out);
int launch_gates(gpry_ctx* ctx, const double* Xc, int64_t M, uint8_t* mask);
int launch_predict_small_std(gpry_ctx* ctx, const double* Xc, int M, double* kstar, double* mean_part,
                             double* ss_part);
int launch_gradx(gpry_ctx* ctx, const double* x, int raw_affine, int want_kinv, double* kstar, double* G,
                 double* u, double* w, double* part, double* out);
int launch_gradx_batch(gpry_ctx* ctx, const double* Xb, int64_t m, int raw_affine, const double* Wm, int64_t ldw,
                       double* out);
int launch_lml_small(gpry_ctx* ctx, int want_grad, double* out, int* info);   // lml_small.hip: N <= 128, d <= 16 in one launch, results picked up as they land; 1 = not applicable
int launch_lml_small_batch(gpry_ctx* ctx, int B, const double* params, int want_grad, double* out, int* info);   // out: B x (2 + 1 + GPRY_MAX_DIM)
int launch_lml_traces(gpry_ctx* ctx, const double* Kinv, const double* alpha,
                      double* grad_out_dev, const double* lq_dev, double* host_res, int info_at);
// host_res (nullable, mapped host memory): [logdet/2, quad, grad...] and dinfo[0..1] as doubles at info_at, status last

// ---- chol.hip ----------------------------------------------------------------------
int potrf_lower_fused(gpry_ctx* ctx, double* A, int64_t Np);       // panel steps, every trailing update its own launch (comparator)
int potrf_lower_overlap(gpry_ctx* ctx, double* A, int64_t Np);
int potrf_stacked(gpry_ctx* ctx, double* A, double* U, int64_t Np);      // A <- L, U <- L^-T (U: the identity on entry)
bool potrf_stacked_usable(const gpry_ctx* ctx, int64_t Np);
int transpose_upper_launch(gpry_ctx* ctx, const double* U, double* V, int64_t Np);     // panel step + earlier trailing tiles in ONE launch; above Np = 3584 segment by segment (default)
int trtri_lower(gpry_ctx* ctx, const double* L, double* V, double* T, int64_t Np);
// V = L^-1 queued phase by phase underneath potrf (chol.hip); begin returns 1 when the size is not cut
int trtri_pipeline_begin(gpry_ctx* ctx, const double* L, double* V, double* T, int64_t Np);
int trtri_pipeline_step(gpry_ctx* ctx, int blocks_done);
int trtri_pipeline_finish(gpry_ctx* ctx);
void trtri_pipeline_abort(gpry_ctx* ctx);   // error path: wait for the side stream, mark the chain inactive
void trtri_pipe_free(gpry_ctx* ctx);
int lauum_lower(gpry_ctx* ctx, const double* V, double* Kinv, int64_t Np);
int factor_chain_slices(gpry_ctx* ctx, int64_t Np, int* slices);   // split-K slices V = L^-1 and K^-1 = V^T V need at this size
int solve_alpha(gpry_ctx* ctx, const double* V, const double* y, double* z, double* alpha,
                int64_t Np);
int logdet_and_quad(gpry_ctx* ctx, const double* L, const double* z, int64_t Np,
                    double* out2_dev, double* host_res, int info_at);   // host_res: as above, value-only evaluation
int rocsolver_potrf_trtri(gpry_ctx* ctx, double* A, double* V, int64_t Np, int want_v);

// ---- server.hip: resident predict kernel --------------------------------------------
int serve_stop(gpry_ctx* ctx);          // no-op when nothing is running; every model-changing entry point calls it first
void serve_free(gpry_ctx* ctx);
void serve_stats(gpry_ctx* ctx, int64_t* launches, int64_t* requests);
int serve_predict_mean(gpry_ctx* ctx, const double* X, int64_t M, double* part, int* nsplit_out, unsigned* gate_bits);
#define GPRY_SERVE_MAXM 8

int ensure_capacity(gpry_ctx* ctx, int64_t N, int d);
int ensure_pinned(gpry_ctx* ctx, int64_t bytes);
