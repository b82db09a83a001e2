/*
 * albatross_amd.h — C-ABI of the MI355X-native dense Gaussian-process engine.
 *
 * This is the drop-in boundary for the dense GP hot path of swift-nav/albatross
 * (Gram build -> LL^T factor -> solve -> log-marginal-likelihood -> predict).
 * The reference is a header-only C++ template library with no FFI of its own;
 * each entry point below names the reference function (file:line, relative to
 * the albatross checkout) whose work it replaces.  Plain pointers and sizes
 * only; opaque handles; int status returns; no exceptions cross this boundary.
 *
 * Matrix layout: column-major fp64, exactly like Eigen::MatrixXd.
 * Feature layout: row-major n x dim fp64 (array-of-structs, like
 * std::vector<Eigen::Vector3d> / std::vector<double>).
 */
#ifndef ALBATROSS_AMD_H
#define ALBATROSS_AMD_H

#include <stdint.h>

/* Every entry point below is exported with default visibility; the library is built with -fvisibility=hidden, so these
 * declarations ARE its export list (tests/test_capi_host.py compares them with `nm -D`). */
#if defined(__GNUC__) || defined(__clang__)
#define AGP_API __attribute__((visibility("default")))
#else
#define AGP_API
#endif

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status codes ------------------------------------------------------- */
typedef enum {
  AGP_OK = 0,
  AGP_ERR_INVALID_ARGUMENT = 1,
  /* the assembled covariance holds a NaN: ALBATROSS_ASSERT(!cov.hasNaN()),
   * include/albatross/src/models/gp.hpp:66 */
  AGP_ERR_NAN_INPUT = 2,
  /* un-pivoted LL^T met a pivot <= 0.  The reference's pivoted LDL^T tolerates semi-definite input: callers
   * that want that behaviour go on to the DEVICE implementation of that algorithm, agp_ldlt_* below (the
   * host mirrors do: gp.py `pivoted_fallback`, albatross.hpp `fit_pivoted`).  There is no CPU path. */
  AGP_ERR_NOT_POSITIVE_DEFINITE = 3,
  AGP_ERR_HIP = 4,
  AGP_ERR_COMM = 5,
  AGP_ERR_UNSUPPORTED = 6,
  AGP_ERR_NO_DEVICE = 7
} agp_status;

/* ---- covariance-function descriptor ------------------------------------- */
/* One node of a postfix (reverse-Polish) program describing a composed
 * CovarianceFunction.  Leaves push one value, SUM / PRODUCT pop two and push
 * one, MEASUREMENT_ONLY pops one and pushes one. */
typedef enum {
  /* sigma^2 exp(-(d/l)^2); params = {length_scale, sigma}
   * src/covariance_functions/radial.hpp:25-33,131-189 */
  AGP_OP_SQUARED_EXPONENTIAL = 1,
  /* sigma^2 exp(-|d/l|);   radial.hpp:191-198,239-287 */
  AGP_OP_EXPONENTIAL = 2,
  /* sigma^2 (1+q) exp(-q), q = sqrt(3) d/l; radial.hpp:289-297,421-459 */
  AGP_OP_MATERN32 = 3,
  /* sigma^2 (1+q+q^2/3) exp(-q), q = sqrt(5) d/l; radial.hpp:461-470,491-529 */
  AGP_OP_MATERN52 = 4,
  /* sigma^2; params = {sigma}; polynomials.hpp:31-61 */
  AGP_OP_CONSTANT = 5,
  /* sigma^2 iff x == y (by value / eq_id); params = {sigma}; noise.hpp:20-44 */
  AGP_OP_INDEPENDENT_NOISE = 6,
  /* sigma^2 iff x == y; params = {sigma}; nugget.hpp:32-49 */
  AGP_OP_NUGGET = 7,
  /* sum_p sigma_p^2 x^p y^p on 1-D features, p = 0..order (order <= 3);
   * params = {sigma_0..sigma_order}; polynomials.hpp:63-90 */
  AGP_OP_POLYNOMIAL = 8,
  /* f(x) f(y) with f precomputed per point in scale column `column`;
   * scaling_function.hpp:58-112 */
  AGP_OP_SCALING = 9,
  /* lhs + rhs; covariance_function.hpp:266-272 */
  AGP_OP_SUM = 10,
  /* lhs * rhs, rhs skipped when lhs == 0; covariance_function.hpp:357-367 */
  AGP_OP_PRODUCT = 11,
  /* sub-covariance iff BOTH arguments are Measurement<>, else 0;
   * measurement.hpp:70-106 */
  AGP_OP_MEASUREMENT_ONLY = 12,
  /* sub-covariance iff the two arguments hold the alternatives (a, b) of a
   * variant<> feature type, in either order, else 0: VariantForwarder
   * (covariance_functions/callers.hpp:419-544) returns 0 for every pair of
   * alternatives a covariance function defines no _call_impl for.  The
   * alternative index of each point travels as a value in scale column
   * `column`; params = {a, b}.  A gated-off term is UNDEFINED for the pair:
   * AGP_OP_SUM / AGP_OP_PRODUCT then keep their other operand alone
   * (covariance_function.hpp:266-294, 357-389); an undefined result is 0. */
  AGP_OP_TYPE_PAIR = 13
} agp_op;

typedef enum {
  AGP_METRIC_EUCLIDEAN = 0, /* distance_metrics.hpp:30-45 */
  AGP_METRIC_RADIAL = 1,    /* distance_metrics.hpp:47-62 */
  AGP_METRIC_ANGULAR = 2    /* distance_metrics.hpp:64-90 */
} agp_metric;

#define AGP_MAX_KERNEL_NODES 32
#define AGP_MAX_STACK 8
#define AGP_MAX_DIM 8
#define AGP_MAX_SCALE_COLUMNS 4

typedef struct {
  int32_t op;     /* agp_op */
  int32_t metric; /* agp_metric, radial leaves only */
  int32_t column; /* AGP_OP_SCALING / AGP_OP_TYPE_PAIR: which scale column */
  int32_t order;  /* AGP_OP_POLYNOMIAL: polynomial order */
  double params[4];
} agp_kernel_node;

typedef enum { AGP_HOST = 0, AGP_DEVICE = 1 } agp_location;

/* A vector of features, flattened to plain-old-data. */
typedef struct {
  int64_t n;
  int32_t dim;             /* 1..AGP_MAX_DIM */
  int32_t n_scale_columns; /* 0..AGP_MAX_SCALE_COLUMNS */
  const double *coords;    /* n x dim row-major */
  /* optional equality ids for IndependentNoise / Nugget (`x == y`,
   * noise.hpp:37-43): equal id <=> equal feature.  NULL: features are equal
   * iff all coords compare equal. */
  const int64_t *eq_id;
  const double *scales;    /* n x n_scale_columns column-major, or NULL */
  int32_t is_measurement;  /* 1: every feature is wrapped in Measurement<> */
  int32_t location;        /* agp_location of coords / eq_id / scales */
} agp_features;

typedef struct agp_context agp_context;
typedef struct agp_kernel agp_kernel;
typedef struct agp_fit agp_fit;
typedef struct agp_comm agp_comm; /* transport of the multi-GPU entry points, see "multi-GPU" below */

/* ---- context ------------------------------------------------------------- */
/* One context per host thread (or externally locked).  Owns the HIP streams
 * and scratch workspaces.  The reference's equivalent state is the model's
 * ThreadPool (src/core/model.hpp:30-36,133-135). */
AGP_API int agp_context_create(int device_id, agp_context **out);
AGP_API void agp_context_destroy(agp_context *ctx);
/* waits for everything queued on ANY of the context's streams (device-wide: hipDeviceSynchronize on its device) */
AGP_API int agp_context_synchronize(agp_context *ctx);
/* last HIP error text for AGP_ERR_HIP, "" otherwise */
AGP_API const char *agp_last_error(const agp_context *ctx);
AGP_API const char *agp_status_string(int status);
/* number of visible HIP devices (0 when no GPU / no driver) */
AGP_API int agp_device_count(void);

/* ---- device memory for AGP_DEVICE arguments ------------------------------ */
/* Every entry point that takes a `location` accepts buffers that already live in HBM (features, targets, outputs).  A host
 * program that links nothing but this library gets such buffers here - plain hipMalloc / hipMemcpy / hipFree on the
 * context's device, so that a caller needs no HIP headers and no second runtime in its process (bench.py allocates its
 * inputs and outputs with exactly these).  The reference keeps everything in host Eigen matrices; its equivalent is the
 * allocation inside Eigen::MatrixXd (models/gp.hpp:61-69 copies features and covariance into the fit).
 * agp_device_malloc: *out = `bytes` of device memory (bytes > 0).  agp_device_free(NULL) is a no-op.
 * agp_memcpy: `kind` = the agp_location of DST; the source is at the other location for AGP_HOST <-> AGP_DEVICE copies
 * (kind AGP_DEVICE: host -> device, kind AGP_HOST: device -> host); synchronous - it returns after the copy, and a
 * device -> host copy waits for the work queued on the context's streams first. */
AGP_API int agp_device_malloc(agp_context *ctx, int64_t bytes, void **out);
AGP_API int agp_device_free(agp_context *ctx, void *ptr);
AGP_API int agp_memcpy(agp_context *ctx, void *dst, const void *src, int64_t bytes, int kind);

/* ---- covariance function ------------------------------------------------- */
/* Flattened get_params() of a composed covariance function
 * (covariance_function.hpp:222-420). Immutable after creation. */
AGP_API int agp_kernel_create(const agp_kernel_node *postfix, int n_nodes,
                      agp_kernel **out);
AGP_API void agp_kernel_destroy(agp_kernel *k);

/* ---- Gram ---------------------------------------------------------------- */
/* compute_covariance_matrix (src/covariance_functions/callers.hpp:38-166).
 * y == NULL: symmetric n x n Gram of x (callers.hpp:107-166), full matrix
 * written. Else the n_x x n_y cross Gram (callers.hpp:38-102).
 * `out` is column-major with leading dimension ld, at out_location. */
AGP_API int agp_gram(agp_context *ctx, const agp_kernel *k, const agp_features *x,
             const agp_features *y, double *out, int64_t ld, int out_location);

/* Gram of LinearCombination<X> features: LinearCombinationCaller (covariance_functions/callers.hpp:321-396),
 *   out(a, b) = sum_{i in a} sum_{j in b} c_i c_j k(x_i, y_j).
 * x (y) holds the EXPANDED points, combination a = expanded points x_offsets[a] .. x_offsets[a + 1) with coefficients
 * x_coefficients[..] (nx + 1 offsets and one coefficient per expanded point, host arrays; offsets[0] = 0,
 * offsets[nx] = x->n).  A side with offsets == NULL is a vector of plain features (nx ignored).  y == NULL: the
 * symmetric Gram of x with itself (evaluated for a >= b and mirrored, callers.hpp:119-127).  The Measurement<> flag
 * of x / y applies to the expanded points (MeasurementForwarder sits outside LinearCombinationCaller).  Gram and
 * contraction both run on the device. */
AGP_API int agp_gram_combined(agp_context *ctx, const agp_kernel *k, const agp_features *x, int64_t nx, const int64_t *x_offsets,
                      const double *x_coefficients, const agp_features *y, int64_t ny, const int64_t *y_offsets,
                      const double *y_coefficients, double *out, int64_t ld, int out_location);

/* ---- fit ----------------------------------------------------------------- */
/* Fit<GPFit<...>>::Fit(features, train_cov, targets) (src/models/gp.hpp:61-69)
 * preceded by the Gram of GaussianProcessBase::_fit_impl (gp.hpp:281-294):
 *   K = k(x, x) + diag(y_var);  K = L L^T;  information = K^-1 y.
 * y, y_var (may be NULL = zeros) live at x->location; the factor stays on the
 * device inside *out.  information (n doubles, host) and log_det (host) may be
 * NULL.  The training features are copied into the fit (gp.hpp:63). */
/* A fit belongs to the context that created it and must be destroyed before
 * that context. */
AGP_API int agp_fit_create(agp_context *ctx, const agp_kernel *k, const agp_features *x,
                   const double *y, const double *y_var, agp_fit **out,
                   double *information, double *log_det);
/* Mixed-precision variant of agp_fit_create (BASELINE.json configs[3], SURVEY
 * section 8d config 4): the same Fit<GPFit> constructor (models/gp.hpp:61-69),
 * but the bulk trailing updates (and the next-block-column update) of the LL^T
 * form their products on the 16-bit matrix pipe: every row is scaled by a power
 * of two taken from the diagonal (|L[i, k]| <= sqrt(A[i, i])), every panel is split
 * ONCE into two fp16 planes (h1 + h2 = the scaled value to 22 bits), a product is
 * the four partial products (four v_mfma_f32_16x16x32_f16 per 16 x 16 x 32 block,
 * each exact in fp32; csrc/gemm_f16x2.hip), accumulated in fp32 inside one launch,
 * un-scaled exactly and subtracted from the fp64 matrix; the panel chain, the matrix
 * and every accumulation between outer steps stay fp64.  (AGP_MIXED_F16=0 selects
 * round 5's path, three bf16 planes and six products per block,
 * csrc/gemm_bf16x3.hip; AGP_MIXED_BF16=0 the oldest fallback: fp32-rounded panels on
 * v_mfma_f32_16x16x4_f32.)  The information
 * vector is then refined in fp64: conjugate gradients on the exact fp64 covariance,
 * preconditioned with that factor, until ||y - K a||_2 <= tolerance * ||y||_2,
 * `max_iterations` steps, or the fp64 floor of the system.  *iterations /
 * *residual (may be NULL) report the steps taken and the final relative residual.
 * What the result is good for (tests/test_full_size_configs_gpu.py and
 * tests/test_mixed_precision_gpu.py hold each line; all figures MEASURED,
 * profiles/r06/mixed_log_determinant_by_path.txt):
 *   information vector, predicted means: 1e-8 relative to the fp64 fit (refined) on every path;
 *   predicted variances: 1e-4 relative (they come from the mixed factor);
 *   log_det, default path: BASELINE config 4's covariance at N = 32768 0.017 absolute
 *     = 0.5e-6 N; config 3's kernel (SE(1,1) + noise(0.1)) 0.062 = 1.9e-6 N at N = 32768
 *     and 1.4e-6 N at N = 8192; Matern-5/2(2,1) + noise(0.1) at N = 5300 1.9e-6 N - all
 *     INSIDE the log-likelihood bar of the fp64 path (|nll error| <= 1e-6 N, i.e.
 *     |log_det error| <= 2e-6 N), the last three AT it: the bound is a property of the
 *     covariance function (how much of log|K| sits in the fp32 accumulation), not of N
 *     alone, and is not proven for an arbitrary one;
 *   log_det, bf16 x 3 path: 0.8e-6 N on config 4, 4.3e-6 N / 4.9e-6 N on the other two - outside;
 *   log_det, fp32 fallback path: 1.3e-5 relative (0.6 absolute) on config 4 - outside.
 * Use agp_nll / an fp64 fit where the likelihood must meet the bar for an arbitrary
 * covariance; the host mirrors make reading log_det of a mixed fit an explicit opt-in.
 * The reference has no reduced-precision path; this one exists for problems where one
 * fp64 factorisation is too slow (N >= 32768).  Device memory next to the factor (kept
 * in the context between mixed fits): the exact covariance (8 N^2 B), an fp32 copy of
 * the factor for the preconditioner (4 N^2 B), two panel copies (2 x 3 KB x N: room
 * for the bf16 planes; the fp16 planes take 2 x 2 KB x N of it) and 16 N B of row scales. */
AGP_API int agp_fit_create_mixed(agp_context *ctx, const agp_kernel *k, const agp_features *x,
                         const double *y, const double *y_var, int max_iterations,
                         double tolerance, agp_fit **out, double *information,
                         double *log_det, int *iterations, double *residual);
AGP_API void agp_fit_destroy(agp_fit *fit);
AGP_API int64_t agp_fit_size(const agp_fit *fit);
/* 0-based index of the first non-positive pivot of the last failed factor */
AGP_API int64_t agp_fit_failed_pivot(const agp_fit *fit);
/* sum(log D) of the reference's LDLT == 2 sum(log L_ii)
 * (src/eigen/serializable_ldlt.hpp:128-135) */
AGP_API int agp_fit_log_determinant(const agp_fit *fit, double *out);
/* lower-triangular factor, column-major n x n, strictly-upper part zeroed */
AGP_API int agp_fit_download_factor(agp_context *ctx, const agp_fit *fit, double *L,
                            int64_t ld);
AGP_API int agp_fit_download_information(agp_context *ctx, const agp_fit *fit,
                                 double *information);

/* negative_log_likelihood(deviation, covariance)
 * (src/evaluation/likelihood.hpp:38-66) on K = k(x,x) + diag(y_var):
 *   0.5 (log|K| + y^T K^-1 y + n log 2 pi).
 * Factor is not kept (GaussianProcessBase::log_likelihood, gp.hpp:442-451). */
AGP_API int agp_nll(agp_context *ctx, const agp_kernel *k, const agp_features *x,
            const double *y, const double *y_var, double *out);

/* Tuner objective batching: agp_nll for `count` parameter vectors of one model on one dataset in lock step
 * (batched Gram slabs + batched LL^T; blockIdx.y = parameter vector) — the evaluations that
 * compute_gradient (include/albatross/src/tune/finite_difference.hpp:20-94) and the ModelTuner objective
 * (include/albatross/src/tune/tune.hpp:151-161,276-290) perform one after the other.
 *   kernels[b]   the covariance function with parameter vector b
 *   features[b]  its feature view (normally all the same arrays; they differ only when a ScalingTerm
 *                parameter is tuned); all n equal, all at the same location
 *   y            n x count column-major with leading dimension ldy at that location (mean function removed
 *                per parameter vector); ldy = 0: one target vector shared by all
 *   out[b]       the negative log likelihood (host); NaN where the covariance is not positive definite or
 *                has NaN (the reference turns a NaN metric into +inf, tune.hpp:163-165) */
AGP_API int agp_nll_batch(agp_context *ctx, int count, const agp_kernel *const *kernels,
                  const agp_features *const *features, const double *y, int64_t ldy, const double *y_var,
                  double *out);

/* `count` INDEPENDENT fits of one shape in lock step: the Fit<GPFit> constructor (src/models/gp.hpp:61-69) for several
 * datasets / parameter vectors of n points each at once - the regime of the reference's own workloads
 * (benchmarks/bench_predict.cc:20-40: N = 512; the tuner, tune/tune.hpp:276-290), where one fit alone is bound by the
 * latency of its serial pivots.  The batch shares that latency and fills the GPU with the updates of all problems.
 *   kernels[b], features[b]   covariance function and features of problem b (all n equal, all at one location)
 *   y, ldy                    targets, n x count column-major at that location (ldy = 0: one vector shared by all)
 *   y_var, ldv                target variances likewise, or NULL
 *   out[b]                    an ordinary agp_fit per problem (agp_predict_*, agp_solve, agp_fit_destroy as usual); the fits of
 *                             a batch share one device allocation, released with the last of them
 *   information, ldi          n x count (host) or NULL; log_det[b] (host) or NULL
 *   status[b]                 AGP_OK / AGP_ERR_NAN_INPUT / AGP_ERR_NOT_POSITIVE_DEFINITE per problem (out[b] then reports
 *                             the pivot like agp_fit_create); the return value is about the call as a whole: when it is not
 *                             AGP_OK no handle has been published (every out[b] is NULL) and nothing is left to destroy
 * y and y_var live where features[0] lives (host or device); `information` and `log_det` are HOST arrays whatever the
 * location, and a failed problem leaves its column of `information` untouched.  No pivoted fall-back is applied to a
 * problem that is not positive definite (agp_fit_create's callers go on to agp_ldlt_*; a batch reports and moves on). */
AGP_API int agp_fit_create_batch(agp_context *ctx, int count, const agp_kernel *const *kernels, const agp_features *const *features,
                                 const double *y, int64_t ldy, const double *y_var, int64_t ldv, agp_fit **out, double *information,
                                 int64_t ldi, double *log_det, int *status);

/* ---- solve (CovarianceRepresentation::solve, gp.hpp:42-45,68,96,111) ----- */
/* out = K^-1 rhs; rhs/out column-major n x nrhs (ld = n), at `location`. */
AGP_API int agp_solve(agp_context *ctx, const agp_fit *fit, const double *rhs,
              int64_t nrhs, double *out, int location);

/* ---- dense-matrix factor (CovarianceRepresentation from a matrix) ----------- */
/* Eigen::SerializableLDLT(const MatrixXd &) (src/eigen/serializable_ldlt.hpp:27)
 * as used by update() (models/gp.hpp:393), BlockSymmetric (linalg/block_symmetric.hpp:
 * 46-60) and the dense negative_log_likelihood: factor a symmetric positive-definite
 * matrix given by ONE triangle (column-major, ld, at `location`): uplo = 0 reads the
 * lower triangle, uplo = 1 the upper one (a row-major array handed over as it is).  The handle
 * supports agp_solve, agp_fit_log_determinant, agp_fit_inverse_diagonal,
 * agp_fit_download_factor, agp_fit_size, agp_fit_failed_pivot; it has no training
 * features (agp_predict_* and agp_loo_marginal reject it). */
AGP_API int agp_factor_create(agp_context *ctx, const double *K, int64_t n, int64_t ld,
                      int uplo, int location, agp_fit **out);
/* negative_log_likelihood(deviation, covariance) (src/evaluation/likelihood.hpp:53-66):
 * 0.5 (log|K| + dev^T K^-1 dev + n log 2 pi); the univariate shortcut (:57-60) included. */
AGP_API int agp_nll_dense(agp_context *ctx, const double *deviation, const double *K,
                  int64_t n, int64_t ld, int uplo, int location, double *out);

/* ---- update: condition a fit on further observations without refitting ------------------------------------
 * FitModel::update -> GaussianProcessBase::_update_impl (src/models/gp.hpp:384-414) with BlockSymmetric
 * (src/linalg/block_symmetric.hpp:46-115).  The reference keeps the old solver plus Ai_B = A^-1 B and the factor of the
 * Schur complement S = C - B^T A^-1 B; here the same algebra extends the RESIDENT factor by one block row,
 *     [[A, B], [B^T, C]] = [[L, 0], [V^T, L_S]] [[L, 0], [V^T, L_S]]^T,   V = L^-1 B,  L_S L_S^T = S,
 * all on the device (triangular solve and SYRK on MFMA, LL^T of the m x m block, one back substitution).  As in the
 * reference: cross = k(train_features, features) and prior = k(features, features) on PLAIN features (gp.hpp:388-396:
 * no Measurement<> wrapper), targets.covariance on the diagonal of the new block, and
 * information = [old - Ai_B S^-1 delta; S^-1 delta].
 *   old      a fit made by agp_fit_create / agp_fit_update on this context (it stays valid)
 *   x_new    the m further features (same dim / scale columns / id convention as the training features)
 *   y_new    their targets with the mean function removed, y_var_new their variances or NULL, at x_new->location
 *   out      a NEW fit of old + m observations for agp_predict_*, agp_solve, agp_fit_download_*, agp_fit_update;
 *            cross-validation entry points reject it (AGP_ERR_UNSUPPORTED)
 *   information  (optional, host) all agp_fit_size(*out) entries: the old observations first, then the new ones */
AGP_API int agp_fit_update(agp_context *ctx, const agp_kernel *k, const agp_fit *old, const agp_features *x_new, const double *y_new,
                   const double *y_var_new, agp_fit **out, double *information, double *log_det);

/* ---- leave-one-out fast path (the tuner's LeaveOneOutLikelihood objective) --- */
/* diag(K^-1): SerializableLDLT::inverse_diagonal (src/eigen/serializable_ldlt.hpp:
 * 137-199: R = L^-1, then the squared column norms of R).  out: n doubles. */
AGP_API int agp_fit_inverse_diagonal(agp_context *ctx, const agp_fit *fit, double *out,
                             int out_location);
/* Leave-one-out predictive marginals, all n at once: held_out_predictions with
 * singleton groups (src/evaluation/cross_validation_utils.hpp:165-232) ==
 * leave_one_out_conditional (:138-163, GPML eq. 5.12):
 *   variance_i = 1 / (K^-1)_ii ,  mean_i = y_i - information_i / (K^-1)_ii.
 * y = the target means as passed to the fit's dataset (n doubles at `location`);
 * mean / variance: n doubles each at `location`. */
AGP_API int agp_loo_marginal(agp_context *ctx, const agp_fit *fit, const double *y,
                     double *mean, double *variance, int location);

/* Leave-one-GROUP-out.  Groups are index sets into the training data: group g is
 * indices[offsets[g] .. offsets[g + 1]) (offsets has n_groups + 1 entries, offsets[0] = 0; both
 * arrays live on the host).
 *
 * agp_fit_inverse_blocks = SerializableLDLT::inverse_blocks
 * (include/albatross/src/eigen/serializable_ldlt.hpp:137-179): blocks receives, concatenated, the
 * column-major |g| x |g| matrices (K^-1)[I_g, I_g].
 *
 * agp_held_out_predictions = details::held_out_predictions
 * (include/albatross/src/evaluation/cross_validation_utils.hpp:165-232; called by
 * gp_cross_validated_predictions, models/gp.hpp:465-482): for every group the prediction of its
 * own targets from all OTHER groups, without refitting:
 *   mean_g = y_g - B_g^-1 information_g,  marginal variance = diag(B_g^-1),  joint = B_g^-1,
 *   B_g = (K^-1)[I_g, I_g].
 * mean / variance (variance may be NULL) are written in the order of `indices`; joint (may be NULL)
 * receives the concatenated column-major blocks.  y, mean, variance, joint live at `location`. */
AGP_API int agp_fit_inverse_blocks(agp_context *ctx, const agp_fit *fit, int64_t n_groups, const int64_t *offsets,
                           const int64_t *indices, double *blocks, int out_location);
AGP_API int agp_held_out_predictions(agp_context *ctx, const agp_fit *fit, const double *y, int64_t n_groups,
                             const int64_t *offsets, const int64_t *indices, double *mean, double *variance,
                             double *joint, int location);

/* ---- pivoted L D L^T ------------------------------------------------------------------------
 * Eigen::LDLT<MatrixXd, Lower> as albatross uses it through SerializableLDLT
 * (include/albatross/src/eigen/serializable_ldlt.hpp:27; evaluation/likelihood.hpp:63,
 * covariance_functions/representations.hpp:64-96, models/gp.hpp:148,393): P A P^T = L D L^T with
 * diagonal pivoting, for symmetric matrices that are only SEMI-definite or too ill-conditioned for
 * the un-pivoted LL^T of agp_factor_create.  The factorisation follows the reference's unblocked
 * left-looking algorithm operation by operation (L, D and the transpositions are bit-identical to
 * the CPU restatement); it is a correctness path, level-2 bound, meant for moderate n.
 *   agp_ldlt_create   K as in agp_factor_create (one triangle, uplo); *success (optional) = Eigen's
 *                     info() == Success (0: a non-zero pivot followed a zero one, NumericalIssue)
 *   agp_ldlt_solve    LDLT::solve: P^T L^-T D^+ L^-1 P rhs, D^+ zeroing the pivots that are not above
 *                     the smallest normal number; rhs / out are n x nrhs column-major at `location`
 *   agp_ldlt_vector_d / _transpositions / _download   vectorD(), transpositionsP(), matrixLDLT() (host) */
typedef struct agp_ldlt agp_ldlt;
AGP_API int agp_ldlt_create(agp_context *ctx, const double *K, int64_t n, int64_t ld, int uplo, int location,
                    agp_ldlt **out, int *success);
AGP_API void agp_ldlt_destroy(agp_ldlt *ldlt);
AGP_API int64_t agp_ldlt_size(const agp_ldlt *ldlt);
AGP_API int agp_ldlt_solve(agp_context *ctx, const agp_ldlt *ldlt, const double *rhs, int64_t nrhs, double *out,
                   int location);
/* SerializableLDLT::sqrt_solve (serializable_ldlt.hpp:99-109): D^-1/2 L^-1 P rhs, D^-1/2 zero where D_i <= 0 (:58-69);
 * out^T out = rhs^T A^-1 rhs.  rhs / out n x nrhs column-major at `location`. */
AGP_API int agp_ldlt_sqrt_solve(agp_context *ctx, const agp_ldlt *ldlt, const double *rhs, int64_t nrhs, double *out,
                        int location);
AGP_API int agp_ldlt_vector_d(const agp_ldlt *ldlt, double *d);
AGP_API int agp_ldlt_transpositions(const agp_ldlt *ldlt, int64_t *tr);
AGP_API int agp_ldlt_download(agp_context *ctx, const agp_ldlt *ldlt, double *packed, int64_t ld);

/* ---- sparse Gaussian process (FITC / PITC) --------------------------------------------------
 * SparseGaussianProcessRegression, include/albatross/src/models/sparse_gp.hpp.
 *
 * agp_sparse_fit_create = _fit_impl (:354-381) on compute_internal_components (:631-706):
 *   x        the n training features ALREADY reordered group by group (reordered_inds, :645-662):
 *            independent group g = rows offsets[g] .. offsets[g + 1) (offsets: n_groups + 1 host
 *            entries, offsets[0] = 0, offsets[n_groups] = n); they are wrapped as measurements
 *            inside (as_measurements, :649-650)
 *   y, y_var target means / variances (or NULL) in the same order, at x->location; y is used as
 *            given (the reference copies y before it removes the mean function, :664-668)
 *   u        the m inducing features (the InducingPointStrategy's result, :358-360)
 *   nuggets  measurement_nugget on every diagonal of A (:692-696), inducing_nugget on K_uu (:676-677)
 * The handle keeps what prediction needs (Fit<SparseGPFit>: inducing features, the K_uu factor,
 * the factor of Sigma^-1 = K_uu + K_uf A^-1 K_fu, the information vector).  information (m doubles,
 * host) and nll (= -log_likelihood, :524-596, without parameter priors) are optional outputs.
 * out may be NULL (only nll / information wanted); agp_sparse_nll is that call.
 * Two algorithms: LL^T of K_uu and CholeskyQR2 of B (MFMA-bound, the fast path) first; where that finds K_uu or
 * B^T B not numerically positive definite - inducing points denser than the length scale, as in the reference's own
 * tests - the reference's algorithm itself runs on the device: pivoted L D L^T of K_uu (agp_ldlt_*) and the
 * column-pivoted Householder QR of B (level-2 bound, moderate m; AGP_SPARSE_PIVOTED=1 forces it; not with a
 * communicator).  Fits of the second kind are in "pivoted form" (R, P as the reference stores them) and stay so
 * under agp_sparse_fit_update.
 * Errors: AGP_ERR_NOT_POSITIVE_DEFINITE if a block of A is not numerically positive definite (the reference's
 * block LDLT would continue), AGP_ERR_NAN_INPUT. */
typedef struct agp_sparse_fit agp_sparse_fit;
AGP_API int agp_sparse_fit_create(agp_context *ctx, const agp_kernel *kernel, const agp_features *x, int64_t n_groups,
                          const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                          double measurement_nugget, double inducing_nugget, agp_sparse_fit **out,
                          double *information, double *nll);
/* The same fit with the observations split BY GROUP over the ranks of a communicator (BASELINE
 * configs[4]: "SparseGP (PITC) ... on 8 GPUs").  The groups of a PITC / FITC model are independent given the inducing
 * points (sparse_gp.hpp:129-243), so every rank builds K_uf, P, the blocks of A and W for ITS groups alone (x, offsets, y,
 * y_var: this rank's groups only, at least one) and the m x m sums over observations - W W^T and its CholeskyQR2
 * repair - plus four m-vectors are all-reduced (2 x 8 m^2 B + O(m) per fit: 2 x 32 MiB at m = 2048).  Every rank passes
 * the same inducing points u and receives the same fit (handle, information, nll of ALL observations); predictions then
 * need no further exchange.  Collective. */
AGP_API int agp_sparse_fit_create_sharded(agp_context *ctx, agp_comm *comm, const agp_kernel *kernel, const agp_features *x,
                                  int64_t n_groups, const int64_t *offsets, const double *y, const double *y_var,
                                  const agp_features *u, double measurement_nugget, double inducing_nugget,
                                  agp_sparse_fit **out, double *information, double *nll);
AGP_API void agp_sparse_fit_destroy(agp_sparse_fit *fit);
AGP_API int64_t agp_sparse_fit_size(const agp_sparse_fit *fit); /* number of inducing points */
AGP_API int agp_sparse_fit_information(agp_context *ctx, const agp_sparse_fit *fit, double *information);
/* FitModel::update for a sparse fit: _update_impl (sparse_gp.hpp:322-371).  Further observations (grouped
 * like those of agp_sparse_fit_create; wrapped as measurements inside) are folded into `old` through
 * B = [R_old P_old^T; A^-1/2 K_fu],  y_aug = [R_old P_old^T v_old; A^-1/2 y]; the inducing points and their
 * K_uu factor are shared with `old`, which stays valid.  Returns a NEW handle in *out; information
 * (m doubles, host) is optional.  An `old` made by agp_sparse_fit_from_prediction (or by an update of one) is
 * updated with the reference's own algorithm - the column-pivoted QR of B, v = B_qr.solve(y_aug), R's diagonal
 * inflated by 1e-10 when B is rank deficient (:353-365) - and the result stays in that pivoted form. */
AGP_API int agp_sparse_fit_update(agp_context *ctx, const agp_kernel *kernel, const agp_sparse_fit *old,
                          const agp_features *x, int64_t n_groups, const int64_t *offsets, const double *y,
                          const double *y_var, double measurement_nugget, agp_sparse_fit **out,
                          double *information);
/* SparseGaussianProcessRegression::fit_from_prediction (sparse_gp.hpp:406-461), the body of
 * rebase_inducing_points (:714-725): the fit on the m inducing features z that reproduces a joint prediction made
 * AT z - mean (m) and covariance (m x m column-major, leading dimension ldc, both triangles), at `location`.
 *   train_covariance = LDLT(K_zz) (no nugget), information = train_covariance.solve(mean),
 *   C = covariance + 1e-8 I (DEFAULT_NUGGET, :20), B_z = C_ldlt.sqrt_solve(K_zz), (R, P) = QR(B_z).
 * K_zz is singular to working precision whenever z is denser than the length scale, so this path runs the
 * reference's pivoted factorisations on the device (the pivoted L D L^T of agp_ldlt_*, a column-pivoted Householder
 * QR) instead of the LL^T / CholeskyQR2 of agp_sparse_fit_create; they are level-2 bound, meant for moderate m.
 * inducing_nugget is what a later agp_sparse_fit_update adds to K_zz for P = K_zz^-1/2 K_zf (:674-685).
 * Optional outputs: information (m doubles, host), numerical_rank (B_qr->rank(), :458). */
AGP_API int agp_sparse_fit_from_prediction(agp_context *ctx, const agp_kernel *kernel, const agp_features *z,
                                   const double *mean, const double *covariance, int64_t ldc, int location,
                                   double inducing_nugget, agp_sparse_fit **out, double *information,
                                   int64_t *numerical_rank);
/* Fit<SparseGPFit>::numerical_rank: the rank of the pivoted QR for fits in pivoted form, m otherwise. */
AGP_API int64_t agp_sparse_fit_numerical_rank(const agp_sparse_fit *fit);
AGP_API int agp_sparse_nll(agp_context *ctx, const agp_kernel *kernel, const agp_features *x, int64_t n_groups,
                   const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                   double measurement_nugget, double inducing_nugget, double *out);
/* _predict_impl (:447-521): mean = K_*u v; covariance = K_** - Q_** + K_*u Sigma K_u*.  The mean
 * function is the caller's (mean_function_.add_to). */
AGP_API int agp_sparse_predict_mean(agp_context *ctx, const agp_kernel *kernel, const agp_sparse_fit *fit,
                            const agp_features *xs, double *mean, int out_location);
AGP_API int agp_sparse_predict_marginal(agp_context *ctx, const agp_kernel *kernel, const agp_sparse_fit *fit,
                                const agp_features *xs, double *mean, double *variance, int out_location);
AGP_API int agp_sparse_predict_joint(agp_context *ctx, const agp_kernel *kernel, const agp_sparse_fit *fit,
                             const agp_features *xs, double *mean, double *covariance, int out_location);

/* ---- predict ------------------------------------------------------------- */
/* gp_mean_prediction (gp.hpp:82-85) via _predict_impl (gp.hpp:350-366):
 *   mean = k(train, xs)^T information.  mean: m doubles at out_location. */
AGP_API int agp_predict_mean(agp_context *ctx, const agp_kernel *k, const agp_fit *fit,
                     const agp_features *xs, double *mean, int out_location);
/* gp_marginal_prediction (gp.hpp:87-101) via _predict_impl (gp.hpp:326-348):
 *   var_j = k(xs_j, xs_j) - sum_i (K^-1 K*)_ij K*_ij.
 * Any number of test points: they pass in slices that keep the n x m workspace at 2 GiB (nothing couples the
 * columns of a marginal prediction). */
AGP_API int agp_predict_marginal(agp_context *ctx, const agp_kernel *k,
                         const agp_fit *fit, const agp_features *xs,
                         double *mean, double *variance, int out_location);
/* gp_joint_prediction (gp.hpp:103-113) via _predict_impl (gp.hpp:305-324):
 *   cov = k(xs, xs) - K*^T K^-1 K*.  cov: m x m column-major, ld = m. */
AGP_API int agp_predict_joint(agp_context *ctx, const agp_kernel *k, const agp_fit *fit,
                      const agp_features *xs, double *mean, double *cov,
                      int out_location);

/* ---- CovarianceRepresentation compositions on the device ----------------------------------------------------------
 * The solvers a fit can hold besides its own factor (src/models/gp.hpp:42-45 asks for `solve` and `rows`), and the generic
 * form of _predict_impl over any of them - everything stays in HBM, a solve is device solves + MFMA products:
 *   agp_solver_from_fit / _from_ldlt   the LL^T factor / the pivoted L D L^T as a solver (borrowed: the fit outlives it)
 *   agp_solver_block_symmetric         BlockSymmetric<Solver> (src/linalg/block_symmetric.hpp:46-133): solver of
 *                                      [[A, B], [B^T, C]] from a solver of A, B (rows(A) x rows(S), ldb, at `location`) and the
 *                                      solver S of the Schur complement C - B^T A^-1 B; Ai_B = A.solve(B) is kept on the device
 *   agp_solver_explained               ExplainedCovariance (src/covariance_functions/representations.hpp:64-96):
 *                                      S^-1 = outer^-1 inner outer^-1, inner n x n (ld, at `location`)
 *   agp_solver_solve                   rhs / out n x nrhs column-major (ld = n) at `location`
 *   agp_solver_predict                 gp.hpp:305-366 over a generic representation: mode 0 mean, 1 marginal (variance m),
 *                                      2 joint (covariance m x m, ld = m); train = the fit's training features as the
 *                                      covariance function sees them, information n doubles; all at `location`
 * Sub-solvers are borrowed and must outlive the composition. */
typedef struct agp_solver agp_solver;
AGP_API int agp_solver_from_fit(agp_context *ctx, const agp_fit *fit, agp_solver **out);
AGP_API int agp_solver_from_ldlt(agp_context *ctx, const agp_ldlt *ldlt, agp_solver **out);
AGP_API int agp_solver_block_symmetric(agp_context *ctx, const agp_solver *A, const double *B, int64_t ldb, int location,
                                       const agp_solver *S, agp_solver **out);
AGP_API int agp_solver_explained(agp_context *ctx, const agp_solver *outer, const double *inner, int64_t ld, int location,
                                 agp_solver **out);
AGP_API int64_t agp_solver_rows(const agp_solver *solver);
AGP_API int agp_solver_solve(agp_context *ctx, const agp_solver *solver, const double *rhs, int64_t nrhs, double *out, int location);
AGP_API int agp_solver_predict(agp_context *ctx, const agp_kernel *k, const agp_solver *solver, const agp_features *train,
                               const double *information, const agp_features *xs, double *mean, double *var_or_cov, int mode,
                               int location);
/* FitModel::update on a generic representation (src/models/gp.hpp:403-407): the information vector of the conditioned fit,
 * [information - Ai_B Si_delta ; Si_delta], from the Ai_B a BlockSymmetric solver holds in HBM (block_symmetric.hpp:51).
 * information: rows(A) doubles, si_delta: rows(S) doubles, out: rows(A) + rows(S) doubles, all at `location`. */
AGP_API int agp_solver_update_information(agp_context *ctx, const agp_solver *block_symmetric, const double *information,
                                          const double *si_delta, double *out, int location);
/* agp_solver_predict for LinearCombination<X> features on either side (covariance_functions/callers.hpp:321-396): train / xs
 * hold the EXPANDED points, combination a of a side = its expanded points offsets[a] .. offsets[a + 1) with
 * coefficients[..] (host arrays as in agp_gram_combined; offsets == NULL: plain features on that side).  The solver's
 * size is the number of training combinations.  Covariances, contraction, solve and products all run on the device. */
AGP_API int agp_solver_predict_combined(agp_context *ctx, const agp_kernel *k, const agp_solver *solver, const agp_features *train,
                                        int64_t n_train, const int64_t *train_offsets, const double *train_coefficients,
                                        const double *information, const agp_features *xs, int64_t n_xs, const int64_t *xs_offsets,
                                        const double *xs_coefficients, double *mean, double *var_or_cov, int mode, int location);
AGP_API void agp_solver_destroy(agp_solver *solver);

/* ---- multi-GPU: ONE fit sharded over the GPUs of a node ---------------------------------------
 * The work of the Fit<GPFit> constructor (include/albatross/src/models/gp.hpp:61-69: covariance +
 * diag(targets.covariance), SerializableLDLT, information = ldlt.solve(y)) and of _fit_impl's Gram
 * (gp.hpp:281-294) for one dataset, spread over `nranks` processes with one GPU each.
 *
 * Layout: ROW-block cyclic.  The training points are cut into row blocks of 512; row block b (the rows
 * 512 b .. 512 b + 511 of the lower triangle, columns 0 .. 512 (b + 1)) belongs to rank
 * snake(b) = 0..G-1, G-1..0, ... (equal work per pair of rounds); every rank stacks its row blocks into one
 * local column-major matrix and builds their Gram entries itself (no communication).  Per block column k:
 *   owner(k)   LL^T of the 512 x 512 diagonal block (+ the fused forward substitution), BROADCAST of
 *              L_kk / its tile images / z_k                                  (2.4 MB, RCCL broadcast)
 *   every rank X = A[own rows > k, k] L_kk^-T on its own rows, y_own -= X z_k
 *   all ranks  ALL-GATHER of the panel rows (each rank contributes its rows; every xGMI link of every GPU
 *              carries 1/G of the panel at the same time)                    (RCCL all-gather)
 *   every rank C[own rows, cols > k] -= X_own P^T                            (fp64 MFMA update kernels)
 * with one block column of look-ahead: the owner of block k + 1 updates and factors its diagonal block while the
 * panel of step k is still being gathered and applied.  information = L^-T z follows super-block by super-block (four
 * row blocks: every rank keeps the diagonal super-blocks and solves them redundantly) with ONE small all-reduce per
 * super-block.  Every rank receives the full information vector and log-determinant.
 *
 * agp_comm wraps the transport: RCCL (librccl of the ROCm installation, loaded at first use), or - for tests and
 * for boxes where RCCL cannot be used (it refuses two ranks on one device) - collectives supplied by the caller. */
#define AGP_COMM_ID_BYTES 128
/* ncclGetUniqueId: rank 0 calls it and hands the bytes to every rank by any means (a file, MPI, a TCP store). */
AGP_API int agp_comm_unique_id(void *id);
/* ncclCommInitRank on ctx's device; collective over all ranks.  AGP_ERR_COMM when RCCL is unavailable or fails. */
AGP_API int agp_comm_create(agp_context *ctx, int nranks, int rank, const void *id, agp_comm **out);
/* op for all_reduce: 0 = sum, 1 = max */
typedef struct {
  void *user;
  /* every callback works on HOST doubles and returns 0 on success; all ranks call them in the same order */
  int (*broadcast)(void *user, double *buf, int64_t count, int root);
  int (*all_gather)(void *user, const double *send, double *recv, int64_t count_per_rank);
  int (*all_reduce)(void *user, double *buf, int64_t count, int op);
} agp_comm_callbacks;
AGP_API int agp_comm_create_callbacks(int nranks, int rank, const agp_comm_callbacks *cb, agp_comm **out);
/* A DEVICE-ASYNCHRONOUS transport between processes that share ONE GPU (csrc/shard_ipc.hip): every collective is a few
 * kernels on the caller's stream - stores into the peers' mailboxes (hipIpcOpenMemHandle), stream-ordered flags, bounded
 * spins - and the host returns at once, as with RCCL.  For exercising the asynchronous multi-rank schedule on a one-GPU
 * box (RCCL refuses two ranks per device; the callback transport is host-synchronous); RCCL remains the transport of
 * real multi-GPU runs.  `bootstrap`: host collectives (all_gather, all_reduce) that carry the 64-byte IPC handles at
 * creation and serve agp_comm_all_reduce_host / agp_comm_barrier afterwards; it must outlive the communicator.
 * mailbox_doubles: capacity of one of the two mailbox slots (0: 8 Mi doubles); larger messages travel in pieces.
 * nranks <= 16; needs HSA_ENABLE_IPC_MODE_LEGACY=0 in the environment of every rank.  Collective (also the destroy). */
AGP_API int agp_comm_create_ipc(agp_context *ctx, int nranks, int rank, const agp_comm_callbacks *bootstrap, int64_t mailbox_doubles,
                                agp_comm **out);
AGP_API void agp_comm_destroy(agp_comm *comm);
AGP_API int agp_comm_size(const agp_comm *comm);
AGP_API int agp_comm_rank(const agp_comm *comm);
/* control-plane helpers for host code (bench.py's barrier and max-over-ranks timing): in-place on host doubles */
AGP_API int agp_comm_all_reduce_host(agp_comm *comm, double *buf, int64_t count, int op);
AGP_API int agp_comm_barrier(agp_comm *comm);

/* ownership arithmetic of the row-block-cyclic layout (pure host functions) */
AGP_API int64_t agp_shard_local_rows(int64_t n, int64_t block, int nranks, int rank);
/* global row of local row l of `rank` (l < agp_shard_local_rows) */
AGP_API int64_t agp_shard_global_row(int64_t n, int64_t block, int nranks, int rank, int64_t l);
AGP_API int agp_shard_owner(int64_t block_index, int nranks);

typedef struct agp_sharded_fit agp_sharded_fit;
/* One fit over all ranks of `comm`; collective.  Every rank passes the SAME full dataset (x, y, y_var as in
 * agp_fit_create; at x->location) and receives information (n doubles, host, may be NULL) and log_det.
 * A failure (NaN, non-positive pivot, transport error or timeout: AGP_COMM_TIMEOUT_S seconds, default 120) is
 * reported with the same status on every rank that can still be reached.  comm == NULL: one rank, no transport. */
AGP_API int agp_sharded_fit_create(agp_context *ctx, agp_comm *comm, const agp_kernel *k, const agp_features *x,
                           const double *y, const double *y_var, agp_sharded_fit **out, double *information,
                           double *log_det);
AGP_API void agp_sharded_fit_destroy(agp_sharded_fit *fit);
AGP_API int64_t agp_sharded_fit_failed_pivot(const agp_sharded_fit *fit);
/* Replicate the factor: all-gather of the row blocks, after which every rank holds an ordinary agp_fit of the whole
 * problem (the factor of gp.hpp:61-69) and predicts ITS share of the test points with agp_predict_* - predictions
 * are independent per test point (gp.hpp:82-113), so sharding M needs no further exchange.  Collective. */
AGP_API int agp_sharded_fit_replicate(agp_context *ctx, agp_sharded_fit *fit, agp_fit **out);
/* gp_marginal_prediction (src/models/gp.hpp:87-101) straight from the SHARDED factor, without replicating it: the
 * distributed forward substitution V = L^-1 K* by block rows (the owner of a block row solves it and broadcasts its
 * w x M rows of V, every rank updates its own later rows; N x M doubles of broadcasts per call) and one all-reduce of
 * the M column sums.  For factors too large to hold on every GPU; while it fits, agp_sharded_fit_replicate + the
 * ordinary agp_predict_* on each rank's share of the test points needs no exchange at all.  Collective: every rank
 * passes the same test points and receives all M means and variances (at `location`). */
AGP_API int agp_sharded_predict_marginal(agp_context *ctx, const agp_kernel *k, agp_sharded_fit *fit, const agp_features *xs,
                                         double *mean, double *variance, int location);
/* gp_joint_prediction (src/models/gp.hpp:103-113) from the sharded factor, likewise without replicating it: the same
 * distributed forward substitution, then cov = K** - V^T V where every rank multiplies its own rows of V = L^-1 K* and ONE
 * all-reduce of m x m doubles sums the products.  cov: m x m column-major, ld = m, on every rank.  Collective. */
AGP_API int agp_sharded_predict_joint(agp_context *ctx, const agp_kernel *k, agp_sharded_fit *fit, const agp_features *xs,
                                      double *mean, double *covariance, int location);
/* per-stage device time of the last sharded fit on this rank, ms: 0 gram, 1 factor, 2 back substitution,
 * 3 sum of the bulk update launches, 4 their count, 5 their algorithmic flop, 6 host time spent enqueueing the
 * schedule, 7 host time until the device had drained (6 ~ 7: the host is the bottleneck) */
AGP_API int agp_sharded_fit_stage(const agp_sharded_fit *fit, int stage, double *value);

/* ---- instrumentation (bench.py) ----------------------------------------- */
/* Per-stage device time of the LAST fit / nll on this context, measured with
 * HIP events on the stream the kernels were launched on.  Stages:
 * 0 gram, 1 factor (total), 2 solve, 3 trailing-update kernels only (sum),
 * 4 number of trailing-update launches.  Returns ms (or a count for 4). */
AGP_API int agp_last_stage_ms(const agp_context *ctx, int stage, double *ms);
/* enable (1) / disable (0) per-stage event timing (default off: events add
 * host overhead to the launch chain). */
AGP_API int agp_set_profiling(agp_context *ctx, int enabled);

/* ---- switches ------------------------------------------------------------
 * The library reads the following environment variables, ONCE per context, in agp_context_create (csrc/api.hip:
 * read_tuning); a later change of the environment does not affect an existing context.  Nothing else is switchable:
 * the schedule's other parameters are constants (csrc/chol.hip).
 *   AGP_PANEL_FUSED=0        POTRF and panel TRSM as two launches instead of the fused panel kernel
 *   AGP_STEP_BELOW=<rows>    remaining rows at or below which every panel is ONE step launch (default 4608; 0: off)
 *   AGP_MERGE_ABOVE=<rows>   trailing rows above which the update of the NEXT block column rides in the bulk launch (its
 *                            tiles first, counted; a one-wave gate kernel on the panel stream) instead of being a launch
 *                            of its own behind an event (default 8704; 0: the round-5 schedule)
 *   AGP_GRAM_SOP=0           covariance trees through the stack interpreter only (parity tests run both evaluators)
 *   AGP_MIXED_F16=0          agp_fit_create_mixed forms its products from three bf16 planes (round 5) instead of two fp16 planes
 *   AGP_F16X2_TERMS=3        ... from two fp16 planes WITHOUT the h2 h2 product (3 % faster, log_det outside the bar: gemm_f16x2.hip)
 *   AGP_F16X2_LDS_PAD=<b>    extra LDS bytes per workgroup of the fp16 x 2 kernel (default 8192: two per CU; 0: three)
 *   AGP_MIXED_BF16=0         agp_fit_create_mixed forms its fp32-accurate products on the fp32 MFMA instead of the 16-bit planes
 *   AGP_BF16X3_KERNEL=1      ... with the first bf16 x 3 tile kernel (one workgroup per CU) instead of the pair kernel
 *   AGP_BF16X3_LDS_PAD=<b>   extra LDS bytes per workgroup of the pair kernel (default 8192: two per CU; 0: three)
 *   AGP_MIXED_NBO=<w>        outer block width of the mixed factorisation while > 8192 rows remain (default 512)
 *   AGP_SOLVE_NBO=<w>        outer block width of the forward substitution with many right-hand sides (predictions; default 512;
 *                            measured flat from 256 to 1024: profiles/r06/time_predict_marginal_nbo.txt)
 *   AGP_FP64_NBO=<w>         the same for the fp64 factorisation (default 0 = 512)
 *   AGP_GEMM_SMALL_LIMIT=<t> 64 x 64 instead of 128 x 128 tiles for products of fewer than t 128-tiles (default 512)
 *   AGP_BACKSUB_COOP=0       the fit's back substitution as one launch per block (rounds 1-4) instead of ONE launch
 *   AGP_BACKSUB_COOP_MAX=<n> largest fit that uses the one-launch back substitution (default 2047)
 *   AGP_SPARSE_PIVOTED=1     the sparse GP always takes the literal (pivoted LDL^T + column-pivoted QR) path
 *   AGP_PREDICT_CHUNK=<m>    test points per slice of marginal predictions (default: by memory, 2 GiB per slice)
 *   AGP_SHARD_BLOCK=<b>      128 / 256 / 512 rows per row block of the sharded fit (default 512; tests)
 *   AGP_SHARD_FORCE_COMM=1   ONE rank runs the multi-rank schedule through its transport (RCCL group of one; tests)
 *   AGP_SHARD_HOST_PACING=1  the sharded schedule is paced by the host instead of device-side flags
 *   AGP_SHARD_MASK_GFLOP=<g> flop of a rank's bulk update per step (default 40e9) below which a sharded fit counts as
 *                            chain-bound: bulk updates on the CU-masked stream + device-side pacing from there on
 * and, process-wide, at first use:
 *   AGP_COMM_TIMEOUT_S=<s>   deadline of every wait that may hold a collective (default 120)
 *   AGP_RCCL_LIB=<path>      librccl to dlopen (default: the ROCm installation's)
 *   AGP_ROCTX=1              roctx ranges around the stages (rocprofv3 --marker-trace)
 * Test-only entry points (kernel probes, the schedule over caller-supplied block arithmetic, a transport that moves
 * nothing) are `agp_debug_*` in libalbatross_amd_debug.so and are not part of this interface. */

#ifdef __cplusplus
}
#endif
#endif /* ALBATROSS_AMD_H */
