/*
 * gpry_hip.h -- C ABI of libgpry_hip.so: MI355X (gfx950) GP-regression + NORA sweep.
 *
 * This is the drop-in boundary for GPry's hot path.  GPry is pure Python; the calls
 * below are what a ctypes binding inside gpry/gpr.py and gpry/gp_acquisition.py would
 * bind to replace the numpy/scipy/scikit-learn arithmetic (reference file:line cited
 * per entry point; "sklearn:" = scikit-learn 1.7.2 sklearn/gaussian_process/).
 *
 * Conventions
 *  - plain C, no C++ types, no torch types; all host arrays are caller-owned,
 *    C-contiguous float64 / int64 / uint8; every call copies in/out synchronously and
 *    returns after the device work has completed (stream-synchronised).
 *  - return value: 0 ok; <0 API / HIP / RCCL error (text via gpry_last_error);
 *    numerical status (LAPACK-style "info") is returned through an out-parameter.
 *  - one gpry_ctx per device; a ctx is not re-entrant.
 *  - all floating point is IEEE float64.
 */
#ifndef GPRY_HIP_H
#define GPRY_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct gpry_ctx gpry_ctx;
typedef struct gpry_comm gpry_comm;
typedef struct gpry_group gpry_group;

/* kernel ids: Product(ConstantKernel, RBF | Matern(nu)) -- gpry/kernels.py:213,281,601,681 */
enum { GPRY_RBF = 0, GPRY_MATERN12 = 1, GPRY_MATERN32 = 2, GPRY_MATERN52 = 3 };

/* Largest dimension of the parameter space: ABI size of the per-dimension arrays in gpry_affine AND the
 * limit gpry_set_train enforces (the kernels of this build are instantiated for d <= 32; GPry's design
 * envelope is d < 20, README.rst:64). */
#define GPRY_MAX_DIM 32

/* Affine pre-/post-processing fused into the device path.
 * x_ = (x - x_lo) / x_span       gpry/preprocessing.py:380 (Normalize_bounds.transform)
 * y  = y_ * y_std + y_mean       gpry/preprocessing.py:620 (Normalize_y.inverse_transform)
 * s  = s_ * y_std                gpry/preprocessing.py:630 (inverse_transform_scale)
 * y  = min(y, clip_hi)           gpry/gpr.py:1187-1195   (pass +inf for "no clipping") */
typedef struct {
    int has_x_affine;                 /* 0: X is used as given */
    double x_lo[GPRY_MAX_DIM];
    double x_span[GPRY_MAX_DIM];
    double y_mean, y_std;
    double clip_hi;
} gpry_affine;

/* One shortlisted candidate of the NORA sweep (gpry/gp_acquisition.py:1328-1333 feeds
 * candidates to RankedPool.add_one in descending acquisition order). */
typedef struct {
    double acq, y, sigma;
    int64_t idx;
} gpry_cand;

/* bit flags of the per-candidate mask consumed by predict/sweep */
enum {
    GPRY_MASK_CLASSIFIED_INF = 1, /* gpry/gpr.py:1145,1172,1230: mean=-inf, std=0      */
    GPRY_MASK_OUTSIDE_TRUST = 2   /* gpry/gpr.py:1107,1201:      mean=-inf, std kept   */
};

/* ---- library / device ------------------------------------------------------------ */
int gpry_version(void);
int gpry_device_count(int* n);
/* name (<=name_len), HBM bytes, CU count, max clock kHz, gcn arch string (<=arch_len) */
int gpry_device_info(int device, char* name, int name_len, int64_t* hbm_bytes,
                     int* n_cu, int* clock_khz, char* arch, int arch_len);

int gpry_ctx_create(int device, gpry_ctx** out);
int gpry_ctx_destroy(gpry_ctx* ctx);
const char* gpry_last_error(gpry_ctx* ctx); /* ctx may be NULL: last global error */
int gpry_ctx_sync(gpry_ctx* ctx);
/* Options (all of them; unknown keys and values outside the stated range return -1).  One comparator is kept per stage --
 * rocSOLVER for the factorisation ("chol"), the schedule with separate trailing launches for the fused Cholesky
 * ("chol_overlap"), the register-staged GEMM engine for the LDS-DMA one ("gemm_dma") --, the variants that lost their A/B
 * runs in earlier rounds are gone (profiles/HISTORY.md keeps their numbers).
 *   measurement
 *     "timing" 0/1            per-stage HIP-event timers (gpry_timing_get); off by default, switched on by gpry_timing_reset
 *   factorisation (gpry/gpr.py:1453-1465)
 *     "chol" 0/1              0 (default): hand-written MFMA Cholesky + V = L^-1; 1: rocSOLVER dpotrf / dtrtri (comparator)
 *     "chol_overlap" 0/1      1 (default): trailing-update tiles ride in the Cholesky panel launches (above 3584 padded rows
 *                             segment by segment: outer blocks of up to 768 columns, each followed by one SYRK launch, then
 *                             the last 3584 columns); 0: every update a launch of its own (comparator; bit-identical factors)
 *     "factor_pipeline" 0/1   V = L^-1 is queued phase by phase on a second stream underneath the Cholesky panel chain
 *                             (default 1; bit-identical), from "factor_pipeline_min" padded rows on (default 1280)
 *     "gemm_dma" 0/1          1 (default): LDS-DMA staged, software-pipelined GEMM engine for the sweep contraction and the
 *                             128-aligned products of the factor chain; 0: register-staged engine (comparator)
 *     "gemm_streamk"          largest padded size at which the top levels of V = L^-1 and K^-1 = V^T V run as stream-K
 *                             launches (default 5632; 0 = off)
 *     "gemm_small"            launches of at most this many 128 x 128 tiles use 64 x 64 tiles instead (default 32; 0 = never)
 *   objective (sklearn:_gpr.py:574-652)
 *     "lml_small" 0/1         gpry_lml of N <= 128, d <= 16 in ONE launch of one workgroup (default 1; the factor of such an
 *                             evaluation is not kept for gpry_factorize)
 *     "lml_cache" 0/1         gpry_factorize adopts the factor of the last gpry_lml when theta is the same (default 1)
 *     "lml_batch"             largest padded size at which gpry_lml_batch runs all its thetas through ONE chain of launches
 *                             (default 4096: beyond it the host's thread farm of three contexts is faster; 0 = one after another)
 *     "lml_batch_mb"          upper limit of the scratch arena of such a batch in MiB (default 49152; longer batches go in chunks)
 *     "lml_schedule" 0/1      schedule of gpry_lml_batch above 128 rows.  0 (default) latency: every theta gets the launches -- and the
 *                             bits -- of a single gpry_lml.  1 throughput: the chain for many thetas at once (whole-tile products only,
 *                             the recursive inverse at every size, the Cholesky in column blocks with one SYRK launch behind each, the
 *                             thetas dealt over "lml_streams" streams); a theta's result does not depend on how many thetas share the
 *                             call (B = 1 included) and differs from the latency schedule's by rounding.  The host mirror sets it for
 *                             the multi-restart fits (gpry/gpr.py:968-984)
 *     "lml_streams" 1..8      throughput schedule: stream groups per call (default 2)
 *     "tp_block", "tp_tail"   throughput schedule: width of the column blocks of the Cholesky (default 512) and size of the last
 *                             block, factored with riding tiles only (default 1024); multiples of 128.  No bit depends on them
 *     "tp_left" 0/1           throughput schedule: 0 (default) the column blocks right-looking (one launch of the block's width behind
 *                             every block), 1 left-looking (ONE deep launch in front of a block brings its columns up to date with
 *                             all columns left of it; the comparator, measured 1.5 % behind at 4096 rows).  Same factor bit for bit
 *     "chol_tp_segments" 0/1  1: every factorisation of the context takes those column blocks (comparator: the same factor bit for bit)
 *   predict / sweep (gpry/gpr.py:1022-1273, gpry/gp_acquisition.py:971-1108)
 *     "sweep_chunk"           candidates per sweep chunk, rounded up to a multiple of 1024 (default 0 = 32768 from 4096 padded
 *                             training rows on and proportionally more below: the K* panel of a chunk stays 1 GiB)
 *     "cross_mfma" 0/1        1 (default): the cross-kernel panel of sweeps and large predict batches takes its squared
 *                             distances from the matrix pipe (centred coordinates, |x|^2 + |y|^2 - 2 x.y); 0: the difference
 *                             form (comparator; gpry_kernel_cross, the small batches and Matern-1/2 always use it).  The form a sweep
 *                             actually took, and the error estimates of the model that decided it: gpry_sweep_info
 *     "cross_hybrid" 0/1      1 (default): a model whose error estimates rule the matrix-pipe form out because its length scales are
 *                             far below the extent of its data takes the hybrid form (matrix-pipe distances, every pair nearer than
 *                             r^2 = 100 again from the coordinates) instead of the difference form; 0: always the difference form
 *     "panel_debug"           test hooks, ORed bits: 32 the matrix-pipe panel whatever the estimates say, 64 every Cholesky panel step
 *                             reports a timed-out wait, 128 the scratch sets of a batched objective start as NaNs (default 0)
 *     "topk_host"             largest pool that gpry_sweep_topk selects on the host from one kernel's records
 *                             (default 16384; 0 = always the device radix select)
 *     "predict_small"         mean-only gpry_predict of at most this many points is one fused launch (default 2048)
 *     "predict_split" 0/1     split-K contraction for gpry_predict batches of 5 ... a few thousand points (default 1)
 *     "sweep_upload" 0/1      gpry_sweep_logexp with a host pool: 1 (default) uploads it chunk by chunk on a copy stream, chunk
 *                             c + 1 underneath the kernels of chunk c (gpry/gp_acquisition.py:1023-1031 draws a fresh pool every
 *                             mc_every-th call); 0: one copy in front of the sweep (the comparator; same bits)
 *     "sweep_overlap" 0/1     1: the cross-kernel panel of chunk c + 1 is built on the side stream underneath the contraction of
 *                             chunk c (two panels; same bits).  Default 0: measured slower (profiles/r06_sweep.md), kept as the comparator
 *     "chol_stacked"          up to this padded training-set size (default 2048; at most 3584) the inverse factor V = L^-1 comes
 *                             out of the launches of the Cholesky factorisation itself (the identity appended to the matrix as
 *                             extra rows); 0: always the recursive inverse behind the factorisation.  Same L; V, and what is
 *                             computed from it, agree to rounding
 *     "chol_stacked_dense" 0/1  1: that schedule without use of the zero structure of the appended rows (every row block in
 *                             every step, every panel applied to every tile: three times the work; the comparator, same bits)
 *     "predict_gates" 0/1     gpry_predict applies the gates of gpry_set_gates itself (default 0; the Python mirror sets 1)
 *     "predict_serve" 0/1     mean-only gpry_predict of <= 8 points is answered by a RESIDENT kernel (no launch per call;
 *                             default 1); "serve_idle_us" = how long that kernel waits for the next request before it
 *                             leaves (10 ... 1000000, default 2000)
 * The environment variable GPRY_HIP_OPTIONS="key=value,key=value" applies options to every context the process
 * creates (gpry_ctx_create fails on an unknown key or a malformed entry). */
int gpry_ctx_set_option(gpry_ctx* ctx, const char* key, int64_t value);
/* current value of an option (the host mirror asks for "lml_batch" to decide whether the restarts of a fit -- gpry/gpr.py:968-984,
 * one after another there -- can be stepped side by side); unknown keys return -1 */
int gpry_ctx_get_option(gpry_ctx* ctx, const char* key, int64_t* value);

/* ---- model state ------------------------------------------------------------------ */
/* X_train_, y_train_, alpha = noise_^2 in the TRANSFORMED space, as assembled by
 * append_to_data (gpry/gpr.py:743-747).  X_ is N x d row-major. */
int gpry_set_train(gpry_ctx* ctx, const double* X_, const double* y_,
                   const double* alpha, int64_t N, int d);
/* kernel_.theta = [log C, log l_1..l_d] (sklearn:kernels.py:734-747) */
int gpry_set_theta(gpry_ctx* ctx, int kernel_id, const double* theta);
int gpry_set_affine(gpry_ctx* ctx, const gpry_affine* tf);

/* ---- a1/a2/a8: kernel matrices ---------------------------------------------------- */
/* K = kernel_(X_train_) [+ diag(alpha)] -> host N x N.  gpry/gpr.py:1015-1016,
 * sklearn:kernels.py:931-966,1553-1560,1708-1738. */
int gpry_kernel_train(gpry_ctx* ctx, int add_alpha, double* K_out);
/* K* = kernel_(Xc_, X_train_) -> host M x N (gpry/gpr.py:1179).  Xc_ already
 * transformed; meant for tests and small M (the sweep never materialises it on host). */
int gpry_kernel_cross(gpry_ctx* ctx, const double* Xc_, int64_t M, double* K_out);

/* ---- a3/a7: factor (gpry/gpr.py:996-1020, 1453-1465) ------------------------------ */
/* builds K+diag(alpha), L = chol(K), V = L^-1, alpha_ = K^-1 y_.
 * *info = 0 ok, k>0: leading minor of order k is not positive definite (dpotrf). */
int gpry_factorize(gpry_ctx* ctx, int* info);
/* copy-out for attribute parity (L_, V_, alpha_); any pointer may be NULL */
int gpry_get_factor(gpry_ctx* ctx, double* L, double* V, double* alpha_);

/* ---- a15: grow the factor at fixed theta (bordered update instead of refactorising) ------------- */
/* append_to_data(X, y, fit_gpr=False, fit_classifier=False) (gpry/gpr.py:577-753): k more training
 * points with the hyper-parameters AND the pre-processors frozen, so that the old rows of X_train_,
 * y_train_ and alpha are unchanged.  The reference rebuilds K and refactorises (gpry/gpr.py:1015-1017,
 * O(N^3)); this extends L, V = L^-1 and alpha_ by border rows, O(k N^2) (two N x N x k products).
 * Xnew_: k x d, ynew_: k, alphanew: k (noise_^2), all in the TRANSFORMED space as in gpry_set_train.
 * *info = 0 ok; > 0: the enlarged matrix is not positive definite at that (1-based) column -- the
 * model is then left without a valid factor (re-send the training set and call gpry_factorize).
 * Used by the "lies" of BatchOptimizer (gpry/gp_acquisition.py:488-491) and RankedPool.cache_model
 * (:1550-1553). */
int gpry_append_rows(gpry_ctx* ctx, const double* Xnew_, const double* ynew_, const double* alphanew,
                     int64_t k, int* info);

/* ---- a4/a5: log marginal likelihood (+ gradient) ---------------------------------- */
/* sklearn:_gpr.py:574-652 via gpry/gpr.py:876-881.  Non-PD: returns 0 with
 * *lml = -inf, grad = 0, *info > 0 (sklearn:_gpr.py:586-589).  Does not disturb the
 * factor held for prediction.  grad has 1+d entries (ignored if want_grad == 0). */
int gpry_lml(gpry_ctx* ctx, const double* theta, int want_grad, double* lml,
             double* grad, int* info);

/* B evaluations of the same objective in one call (thetas: B x (1 + d), lml: B, grad: B x (1 + d) or NULL, info: B or NULL,
 * each as gpry_lml).  The optimiser runs of a multi-restart fit (gpry/gpr.py:883-994, one after another there) stepped side
 * by side hand over one theta per run and round.  N <= 128, d <= 16: one launch, one workgroup per theta; larger training
 * sets up to option "lml_batch" padded rows (default 4096): ONE chain of launches in which every kernel (covariance build,
 * Cholesky panel steps, V = L^-1 levels, K^-1 = V^T V, traces) carries all thetas, each with its own scratch set; either way
 * every theta gets the arithmetic -- and the bits -- of a single gpry_lml call, and a theta whose matrix is not positive
 * definite returns (-inf, 0, info > 0) on its own.  Beyond that size the thetas are evaluated one after another. */
int gpry_lml_batch(gpry_ctx* ctx, const double* thetas, int64_t B, int want_grad, double* lml, double* grad, int* info);

/* ---- a8-a11: posterior mean / std (gpry/gpr.py:1022-1273, 1275-1352) -------------- */
/* X: M x d, raw if the affine map has_x_affine else already transformed.
 * mask (nullable): per-candidate GPRY_MASK_* bits.  mean/std in untransformed units,
 * clipped (clip_hi) and masked exactly as predict() does.  std may be NULL. */
int gpry_predict(gpry_ctx* ctx, const double* X, int64_t M, const uint8_t* mask,
                 double* mean, double* std);
/* Mean-only calls of <= 8 points -- the closures the nested samplers / MCMC call once per point
 * (gpry/gp_acquisition.py:766-771, 784-793; gpry/mc.py:387-391) -- do not launch a kernel: a resident kernel on
 * a stream of its own answers them through a mailbox in pinned host memory and leaves on its own when no request
 * arrives for "serve_idle_us".  Every other entry point stops it first (it holds the model of its launch), so the
 * caller sees nothing but the latency.  Same bits as the one-launch path ("predict_serve" = 0).
 * gpry_debug_serve_stats: launches of that kernel / requests it answered since the context was created. */
int gpry_debug_serve_stats(gpry_ctx* ctx, int64_t* launches, int64_t* requests);

/* ---- f3: x-gradients for one point (gpry/gpr.py:1236-1266) ------------------------- */
/* x: d doubles, raw/transformed as in gpry_predict.  With G[j][k] = d k(x, X_j) / d x_k in the
 * kernel's coordinates (kernel_.gradient_x: gpry/kernels.py:257-278 RBF, :326-432 Matern,
 * :687-699 product), returns
 *   kgrad      (nullable, N x d)  G
 *   mean_grad  (nullable, d)      G^T alpha_                  -> grad_mean = std_y * mean_grad
 *   kinvk_grad (nullable, d)      G^T K^-1 k*(x), if want_kinv -> grad_std = -std_y^2 * kinvk_grad / sigma_
 * in transformed units; the y-scalings (once for the mean, twice for the std, as the
 * reference does) are left to the caller. */
int gpry_predict_grad(gpry_ctx* ctx, const double* x, int want_kinv, double* kgrad,
                      double* mean_grad, double* kinvk_grad);

/* The same for m points in one call (m <= 4096): V is read once for all of them (two triangular
 * products on the MFMA engine) instead of twice per point.  What a batch of acquisition-optimiser
 * restarts evaluated side by side needs (gpry/gp_acquisition.py:270-389 runs them one after another,
 * each step = one predict with gradients).  mean / std (nullable, m each): as gpry_predict without a
 * mask; mean_grad, kinvk_grad: m x d, transformed units as above. */
int gpry_predict_grad_batch(gpry_ctx* ctx, const double* X, int64_t m, int want_kinv, double* mean,
                            double* std, double* mean_grad, double* kinvk_grad);

/* ONE point, everything in one call: what gpry/gpr.py:1022-1273 returns for predict(x, return_std=True,
 * return_mean_grad=True[, return_std_grad=True]) -- the call the acquisition optimiser makes once per L-BFGS step
 * (gpry/gp_acquisition.py:309-342, gpry/acquisition_functions.py:937-1009).  mean / std as gpry_predict finalises them
 * (y map, clipping; mask_bits: GPRY_MASK_* verdicts of the caller for this point: mean -inf, and std 0 for the
 * classifier bit; with option "predict_gates" the device's own verdict of gpry_set_gates is ORed in; *verdict (nullable)
 * returns the bits that were applied), mean_grad / kinvk_grad (d each; kinvk_grad only with want_kinv) as gpry_predict_grad.  Four launches
 * and one stream wait instead of gpry_predict + gpry_predict_grad (seven launches, two copies, two waits). */
int gpry_predict_point(gpry_ctx* ctx, const double* x, int mask_bits, int want_kinv, double* mean, double* std,
                       double* mean_grad, double* kinvk_grad, int* verdict);

/* ---- f4: gates of the sweep evaluated on the device --------------------------------- */
/* Replaces the host-side verdicts that gpry/gpr.py:1107-1112 (trust region, raw coordinates,
 * closed box) and :1145-1150 -> gpry/svm.py:308-347 (sklearn SVC, RBF kernel, two classes:
 * libsvm _dense_predict on the TRANSFORMED coordinates) compute per candidate.  Once set,
 * gpry_sweep_logexp ORs GPRY_MASK_OUTSIDE_TRUST / GPRY_MASK_CLASSIFIED_INF into the mask itself:
 *   finite  <=>  (sum_i coef[i] exp(-gamma |x_ - sv[i]|^2) + intercept > 0) == positive_is_finite
 * sv: n_sv x d support vectors, coef: dual_coef_[0], intercept: intercept_[0] of the fitted SVC;
 * trust_bounds: d x 2 (lo, hi) or NULL.  n_sv = 0 and trust_bounds = NULL switch the gates off.
 * gpry_predict keeps to the caller's mask unless the option "predict_gates" = 1 asks it to OR the same verdicts in
 * (what the host mirror does: the one-point calls of the samplers then cost no libsvm call each; the resident
 * kernel evaluates decision function and trust box next to the mean). */
int gpry_set_gates(gpry_ctx* ctx, const double* sv, const double* coef, int64_t n_sv, double gamma,
                   double intercept, int positive_is_finite, const double* trust_bounds);

/* ---- a8-a13: fused NORA sweep ----------------------------------------------------- */
/* For all M candidates: mean, std (as gpry_predict), acq = LogExp.f(mean, std,
 * baseline, sigma_n, zeta) (gpry/acquisition_functions.py:1068-1074), kept resident on
 * the device; y_all / sigma_all / acq_all (nullable) receive host copies.
 * n_nan receives the number of NaN acquisition values (the reference raises on those,
 * gpry/gp_acquisition.py:1453-1455). */
int gpry_sweep_logexp(gpry_ctx* ctx, const double* X, int64_t M, const uint8_t* mask,
                      double zeta, double baseline, double sigma_n,
                      double* y_all, double* sigma_all, double* acq_all,
                      int64_t* n_nan);
/* How the cross-kernel panel K(X*, X_train) (gpry/gpr.py:1179) of the context's last sweep / panel predict was built, and the
 * error estimates of the model that decided it.  *panel_form: 0 none yet, 1 distances from the matrix pipe (expanded form
 * |x|^2 + |y|^2 - 2 x.y on centred coordinates), 2 difference form (as scipy's cdist), 3 the small-batch kernel (difference
 * form), 4 hybrid: distances from the matrix pipe, every pair nearer than r^2 = 100 again from the coordinates (taken by a model
 * that fails the gate of form 1 through length scales far below the extent of its data; option "cross_hybrid").  est[0]: estimated error of the posterior mean in units of the normalised targets (entry error x ||alpha_||_2);
 * est[1]: its worst case (x ||alpha_||_1); est[2]: estimated error of the posterior variance relative to the prior
 * variance C (2 x entry error x the bound sqrt(C) / sigma_n,min of ||K^-1 k*||_2); est[3]: the gate both est[0] and est[2]
 * must stay below for form 1 (2.5e-7, a quarter of the 1e-6 the posterior is specified to).  Either pointer may be NULL. */
int gpry_sweep_info(gpry_ctx* ctx, int* panel_form, double* est);
/* Host copies of the arrays of the last gpry_sweep_logexp that are still resident on the device
 * (any pointer may be NULL; M must be the size of that sweep).  Lets a caller skip the copies in
 * gpry_sweep_logexp and fetch them only if somebody asks (NORA.last_MC_sample). */
int gpry_sweep_fetch(gpry_ctx* ctx, int64_t M, double* y_all, double* sigma_all, double* acq_all);
/* Shortlist of the last sweep: the Kp candidates with the largest acq in the total
 * order (acq desc, idx desc) -- the order of np.argsort(acq)[::-1] on distinct values
 * (gpry/gp_acquisition.py:1328-1329).  exclude (nullable, n_exclude sorted indices):
 * rows dropped as "already proposed" (gpry/gp_acquisition.py:1037-1047).
 * *n_out <= Kp records written, sorted; *bound = largest acq NOT returned (-inf if
 * none): every candidate outside the shortlist has acq <= *bound. */
int gpry_sweep_topk(gpry_ctx* ctx, int64_t Kp, const int64_t* exclude, int64_t n_exclude,
                    gpry_cand* top, int64_t* n_out, double* bound);

/* ---- a14/a15: Kriging-believer support (bordered factor instead of deepcopy+refit) - */
/* Start a session on the current factor; drops any registered candidates. */
int gpry_kb_reset(gpry_ctx* ctx);
/* Register m candidates (X: m x d, raw/transformed as in gpry_predict): computes and
 * keeps u(x) = V k*(x).  They get session indices [*first, *first + m).  var0 (nullable)
 * receives C - |u|^2 (transformed units, unclamped). */
int gpry_kb_register(gpry_ctx* ctx, const double* X, int64_t m, int64_t* first,
                     double* var0);
/* For registered candidate p: G[x] = u(p).u(x) and kvec[x] = kernel_(x_p, x) for every
 * registered x (n = number registered so far; both arrays length n). */
int gpry_kb_gram(gpry_ctx* ctx, int64_t p, double* G, double* kvec, int64_t* n);

/* ---- a16: multi-GPU (one process per GPU, RCCL over xGMI) -------------------------- */
/* 128-byte RCCL unique id, made on rank 0 and distributed by the caller. */
int gpry_comm_unique_id(uint8_t id[128]);
int gpry_comm_init(gpry_ctx* ctx, int world, int rank, const uint8_t id[128],
                   gpry_comm** out);
int gpry_comm_destroy(gpry_comm* comm);
/* what RCCL itself reports for this communicator: ncclCommCount / ncclCommUserRank / ncclCommCuDevice
 * (any pointer may be NULL).  bench.py prints the count as config.rccl_ranks. */
int gpry_comm_info(gpry_comm* comm, int* world, int* rank, int* device);
/* all-gather of `bytes` host bytes per rank (staged through device buffers, RCCL) */
int gpry_comm_allgather(gpry_comm* comm, const void* send, int64_t bytes, void* recv);
/* all-reduce(max) of n doubles */
int gpry_comm_allreduce_max(gpry_comm* comm, double* inout, int64_t n);
int gpry_comm_barrier(gpry_comm* comm);

/* ---- a16: single-process device group (k contexts behind one call) ------------------ */
/* What an unmodified gpry.Runner needs to use several GPUs: it is ONE Python process (mpi4py absent
 * => 1-rank dummy, gpry/mpi.py:18-28) that calls multi_add once per iteration (gpry/run.py:838-844).
 * A group holds n contexts on devices[0..n) (entries may repeat: several contexts on one GPU) and
 * drives them from n host threads inside each call.  Partitioning and merge are those of the
 * one-process-per-GPU path: contiguous candidate shards [i*ceil(M/n), ...), model replicated and
 * factorised on every member (gpry/mpi.py:105-131,182-218 shard X[rank::SIZE] and gather instead),
 * shortlists merged with the hold-back rule that keeps the descending stream exact
 * (gpry/gp_acquisition.py:1148-1191 merges per-rank pools).  adopt0 (nullable): an existing context
 * on devices[0] becomes member 0 and stays owned by the caller (it must hold the factorised model;
 * gpry_group_set_model leaves it alone).  The shortlist records (a few KB that the host ranks) are
 * copied out by every member and merged on the host (transport 0); with GPRY_GROUP_TRANSPORT=rccl and
 * all devices distinct they travel by an in-process RCCL all-gather instead (ncclCommInitAll;
 * transport 1; falls back to 0, with the reason in the error text, if RCCL declines). */
int gpry_group_create(int n, const int* devices, gpry_ctx* adopt0, gpry_group** out);
int gpry_group_destroy(gpry_group* group);
int gpry_group_size(gpry_group* group, int* n, int* transport);
gpry_ctx* gpry_group_member(gpry_group* group, int i);     /* for per-member options / timers */
const char* gpry_group_last_error(gpry_group* group);
/* gpry_set_train + gpry_set_theta + gpry_set_affine + gpry_factorize on every owned member,
 * concurrently; *info = first non-zero factorisation status. */
int gpry_group_set_model(gpry_group* group, const double* X_, const double* y_, const double* alpha,
                         int64_t N, int d, int kernel_id, const double* theta, const gpry_affine* tf,
                         int* info);
/* gpry_set_gates on every member (adopted one included) */
int gpry_group_set_gates(gpry_group* group, const double* sv, const double* coef, int64_t n_sv,
                         double gamma, double intercept, int positive_is_finite,
                         const double* trust_bounds);
/* gpry_sweep_logexp over the whole pool, member i on rows [lo_i, hi_i); arguments as there (host
 * arrays of M rows; X == NULL re-uses the resident shards).  gpry_group_sweep_fetch likewise. */
int gpry_group_sweep_logexp(gpry_group* group, const double* X, int64_t M, const uint8_t* mask,
                            double zeta, double baseline, double sigma_n, double* y_all,
                            double* sigma_all, double* acq_all, int64_t* n_nan);
int gpry_group_sweep_fetch(gpry_group* group, int64_t M, double* y_all, double* sigma_all,
                           double* acq_all);
/* Global shortlist: every member selects its Kp best (gpry_sweep_topk, exclusions = sorted GLOBAL
 * rows), the records are merged in the total order (acq desc, idx desc, idx global) and entries
 * with acq <= max_i bound_i are held back unless every member was exhausted.  top must hold
 * n * Kp records; *bound = largest acq that may be missing from the returned prefix (-inf if the
 * pool is exhausted), *exhausted (nullable) = 1 if every member returned fewer than Kp. */
int gpry_group_sweep_topk(gpry_group* group, int64_t Kp, const int64_t* exclude, int64_t n_exclude,
                          gpry_cand* top, int64_t* n_out, double* bound, int* exhausted);
/* n_theta independent LML (+gradient) evaluations, theta t on member t mod n, members concurrently
 * (restart farm, gpry/run.py:1252-1293: independent start points).  thetas: n_theta x (1+d);
 * grad: n_theta x (1+d); info: n_theta (nullable).  Every member must hold the training set. */
int gpry_group_lml_batch(gpry_group* group, const double* thetas, int n_theta, int want_grad,
                         double* lml, double* grad, int* info);

/* ---- measurement ------------------------------------------------------------------ */
/* Device-side timing of the last call's stages in milliseconds (HIP events on the
 * ctx stream).  Known names: "kernel_build", "potrf", "trtri", "lauum", "lml_traces",
 * "cross_build", "sweep_gemm", "sweep_finish", "topk".  Returns <0 if unknown.
 * *count = launches accumulated since gpry_timing_reset.  "lml_batch_shrinks" is a counter, not a timer: *count = the times a
 * batched objective of this context halved its chunk after an out-of-memory answer (any other failure is returned to the caller). */
int gpry_timing_reset(gpry_ctx* ctx);
int gpry_timing_get(gpry_ctx* ctx, const char* name, double* total_ms, int64_t* count);
/* Raw micro-benchmarks used by bench.py to quote measured peaks beside the spec:
 * kind 0: f64 MFMA 16x16x4 issue loop, compiler-allocated (AGPR) accumulators; `bytes` =
 *         workgroups per CU (1..8) -> *value = TFLOP/s  (measured 35-49: AGPR accumulators
 *         run the f64 MFMA well below rate, which is why the GEMM keeps them in VGPRs)
 * kind 2: the same loop with the accumulators pinned to VGPRs -> TFLOP/s (measured 77.1-77.7)
 * kind 1: HBM streaming copy of `bytes` -> *value = GB/s (read+write bytes / time)
 * kind 3: HBM streaming fill of `bytes`  -> *value = GB/s written */
int gpry_microbench(gpry_ctx* ctx, int kind, int64_t bytes, double* value);

/* ---- testing hook ------------------------------------------------------------------- */
/* Runs the FP64 MFMA GEMM engine on host matrices (unit test of the fragment layout).
 * C(MxN) op= A(MxK) B(KxN); a_trans: A given as K x M; b_trans: B given as N x K;
 * epi 0 store, 1 store -AB, 2 C -= AB, 3 per-128-row-tile column sums of squares
 * (C is then ceil(M/128) x N); kmode 0 full (see csrc/common.h for the others).
 * M, N, K must be multiples of 64.  tile_map: bits 0..7 the tile map, bits 8..11 a uniform split-K factor,
 * bits 16..27 > 0: a stream-K launch with segments of that many slab pairs (32 k; M, N multiples of 128). */
int gpry_debug_gemm(gpry_ctx* ctx, const double* A, const double* B, double* C, int M, int N,
                    int K, int a_trans, int b_trans, int epi, int kmode, int lower_only,
                    int tile_map);

/* The acquisition epilogue of the sweep on caller-given (mean, std) pairs, no model involved:
 * acq[i] = LogExp.f(mu[i], sigma[i], baseline, sigma_n, zeta) (gpry/acquisition_functions.py:1068-1074),
 * incl. the edge cases sigma <= sigma_n and mu = -inf (-> -inf). */
int gpry_debug_logexp(gpry_ctx* ctx, const double* mu, const double* sigma, int64_t n, double zeta,
                      double baseline, double sigma_n, double* acq);

#ifdef __cplusplus
}
#endif
#endif /* GPRY_HIP_H */
