// update_api.hip — FitModel::update on the device: agp_fit_update.
//
// Reference: GaussianProcessBase::_update_impl (include/albatross/src/models/gp.hpp:384-414) with BlockSymmetric
// (linalg/block_symmetric.hpp:46-115).  The reference keeps the old solver and adds Ai_B = A^-1 B and the factor of the
// Schur complement S = C - B^T A^-1 B; every solve then goes through the block-inverse formula.  The same linear
// algebra in factor form: the LL^T of the grown matrix is the old factor with a block row appended,
//
//     | A   B |   | L     0   | | L^T  V   |
//     | B^T C | = | V^T   L_S | | 0    L_S^T | ,   V = L^-1 B,   L_S L_S^T = C - V^T V = S ,
//
// so an update is: copy L, one multi-RHS triangular solve for V^T (MFMA), one SYRK for S (MFMA), the LL^T of the
// m x m block S, a forward substitution for the new targets and one backward substitution for the information vector
// ([information - Ai_B S^-1 delta; S^-1 delta] of gp.hpp:403-407 is exactly M^-1 [y_old; y_new]).  Predictions then
// run through the ordinary agp_predict_* on the grown factor - no host arithmetic anywhere.
#include <cstring>
#include <new>

#include "api_internal.h"

namespace agp {
void right_solve_lt(hipStream_t s, const double *A, long long n, long long lda, const double *invd, double *X, long long nrows,
                    long long ldx);

// rows [r0, r1) x columns [c0, c1) <- 0, then A[i][i] <- 1 for i in [r0, r1) when unit_diag
__global__ __launch_bounds__(256) void fill_block_kernel(double *A, long long ld, long long r0, long long r1, long long c0, long long c1,
                                                        int unit_diag) {
  const long long r = r0 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= r1) return;
  for (long long c = c0 + blockIdx.y; c < c1; c += gridDim.y) A[r + c * ld] = (unit_diag && r == c) ? 1. : 0.;
}

void launch_fill_block(hipStream_t s, double *A, long long ld, long long r0, long long r1, long long c0, long long c1, bool unit_diag) {
  if (r1 <= r0 || c1 <= c0) return;
  const long long cols = c1 - c0;
  hipLaunchKernelGGL(fill_block_kernel, dim3((unsigned)((r1 - r0 + 255) / 256), (unsigned)(cols < 1024 ? cols : 1024)), dim3(256), 0, s, A,
                     ld, r0, r1, c0, c1, unit_diag ? 1 : 0);
}

// feature rows [first, first + count) <- copies of row `src` (coords row-major n x dim, ids, scale columns with stride)
__global__ __launch_bounds__(256) void replicate_feature_kernel(double *coords, int dim, long long *ids, double *scales, long long sstride,
                                                               int nsc, long long first, long long count, long long src) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  for (int d = 0; d < dim; ++d) coords[(first + i) * dim + d] = coords[src * dim + d];
  if (ids) ids[first + i] = -2 - i;  // never equal to a caller's (non-negative) id, nor to each other
  for (int k = 0; k < nsc; ++k) scales[(long long)k * sstride + first + i] = scales[(long long)k * sstride + src];
}

// padded <-> real index maps of a fit with phantom rows
long long fit_real_rows(const agp_fit *f) { return f->n_real > 0 ? f->n_real : f->n; }

// calls fn(padded_start, real_start, length) for every maximal run of real rows
template <typename F>
static void for_each_real_run(const agp_fit *f, F fn) {
  long long pad = 0, real = 0;
  for (const auto &ph : f->phantom) {
    if (ph.first > pad) { fn(pad, real, ph.first - pad); real += ph.first - pad; }
    pad = ph.second;
  }
  if (f->n > pad) fn(pad, real, f->n - pad);
}

int fit_compact_vector(agp_context *ctx, const agp_fit *f, const double *padded_dev, double *real_out, int location) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  hipError_t e = hipSuccess;
  for_each_real_run(f, [&](long long pad, long long real, long long len) {
    if (e == hipSuccess) e = hipMemcpyAsync(real_out + real, padded_dev + pad, sizeof(double) * (size_t)len, kind, ctx->stream);
  });
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return AGP_ERR_HIP; }
  return AGP_OK;
}

// padded (n x nrhs, ldp, device; phantom rows zero) <- real rows (n_real x nrhs, ldr) at `location`
int fit_expand_matrix(agp_context *ctx, const agp_fit *f, const double *real_in, long long ldr, long long nrhs, double *padded_dev,
                      long long ldp, int location) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  hipError_t e = hipMemsetAsync(padded_dev, 0, sizeof(double) * (size_t)ldp * (size_t)nrhs, ctx->stream);
  for_each_real_run(f, [&](long long pad, long long real, long long len) {
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(padded_dev + pad, sizeof(double) * (size_t)ldp, real_in + real, sizeof(double) * (size_t)ldr,
                           sizeof(double) * (size_t)len, (size_t)nrhs, kind, ctx->stream);
  });
  if (e == hipSuccess && location == AGP_HOST) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return AGP_ERR_HIP; }
  return AGP_OK;
}

int fit_compact_matrix(agp_context *ctx, const agp_fit *f, const double *padded_dev, long long ldp, long long nrhs, double *real_out,
                       long long ldr, int location) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  hipError_t e = hipSuccess;
  for_each_real_run(f, [&](long long pad, long long real, long long len) {
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(real_out + real, sizeof(double) * (size_t)ldr, padded_dev + pad, sizeof(double) * (size_t)ldp,
                           sizeof(double) * (size_t)len, (size_t)nrhs, kind, ctx->stream);
  });
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return AGP_ERR_HIP; }
  return AGP_OK;
}

// the rows of a cross-covariance block that belong to phantom training rows <- 0 (a phantom carries a placeholder
// feature; its covariance with anything is zero by definition)
void fit_zero_phantom_rows(hipStream_t s, const agp_fit *f, double *V, long long ldv, long long cols) {
  for (const auto &ph : f->phantom) launch_fill_block(s, V, ldv, ph.first, ph.second, 0, cols, false);
}

}  // namespace agp

using namespace agp;

extern "C" {

int agp_fit_update(agp_context *c, const agp_kernel *k, const agp_fit *old, const agp_features *x_new, const double *y_new,
                   const double *y_var_new, agp_fit **out, double *information, double *log_det) {
  if (!c || !k || !old || !x_new || !y_new || !out) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  *out = nullptr;
  if (!old->A || !old->alpha || !old->invd || old->failed_pivot >= 0 || old->ctx != ctx) return AGP_ERR_INVALID_ARGUMENT;
  if (!old->z) return AGP_ERR_UNSUPPORTED;  // a factor without its forward-substituted targets (agp_factor_create, replicated fits)
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x_new);
  if (st != AGP_OK) return st;
  const FeatView &ov = old->train.v;
  const long long m = x_new->n;
  if (m <= 0 || x_new->dim != ov.dim || x_new->n_scale_columns != ov.nsc) return AGP_ERR_INVALID_ARGUMENT;
  if ((ov.ids != nullptr) != (x_new->eq_id != nullptr)) return AGP_ERR_INVALID_ARGUMENT;  // equality by id on one side only
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;

  const long long n0 = old->n, n_pad = round_up(n0, NB), n1 = n_pad + m;
  const long long nblk1 = (n1 + NB - 1) / NB, nblk0 = (n0 + NB - 1) / NB;
  const int dim = ov.dim, nsc = ov.nsc;
  hipStream_t s = ctx->stream;
  agp_fit *fit = new (std::nothrow) agp_fit();
  if (!fit) return AGP_ERR_INVALID_ARGUMENT;
  fit->ctx = ctx;
  fit->device = ctx->device;
  fit->n = n1;
  fit->lda = factor_ld(n1);
  fit->A_bytes = sizeof(double) * (size_t)fit->lda * (size_t)n1;
  fit->phantom = old->phantom;
  if (n_pad > n0) fit->phantom.emplace_back(n0, n_pad);
  fit->n_real = fit_real_rows(old) + m;
  const long long lda = fit->lda;
  double *ynew_d = nullptr, *yvar_d = nullptr;
#define UPD_CHECK(expr)                                                      \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);   \
      if (ynew_d) (void)hipFree(ynew_d);                                     \
      agp_fit_destroy(fit);                                                  \
      return AGP_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)
  if (ctx->pool_A && ctx->pool_A_bytes == fit->A_bytes) {
    fit->A = ctx->pool_A; ctx->pool_A = nullptr; ctx->pool_A_bytes = 0;
  } else {
    UPD_CHECK(hipMalloc(&fit->A, fit->A_bytes));
  }
  UPD_CHECK(hipMalloc(&fit->invd, sizeof(double) * (size_t)nblk1 * (36 * MB * MB)));
  UPD_CHECK(hipMalloc(&fit->winv, sizeof(double) * (size_t)nblk1 * NB * NB));
  UPD_CHECK(hipMalloc(&fit->alpha, sizeof(double) * (size_t)n1));
  UPD_CHECK(hipMalloc(&fit->z, sizeof(double) * (size_t)n1));
  UPD_CHECK(hipMalloc(&ynew_d, sizeof(double) * 2 * (size_t)round_up(m, 2)));
  yvar_d = y_var_new ? ynew_d + round_up(m, 2) : nullptr;

  // ---- train_features = concatenate(fit.train_features, features)   gp.hpp:387 ----
  const hipMemcpyKind kind = x_new->location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  DeviceFeatures &tr = fit->train;
  UPD_CHECK(hipMalloc(&tr.owned[0], sizeof(double) * (size_t)n1 * (size_t)dim));
  double *coords = static_cast<double *>(tr.owned[0]);
  UPD_CHECK(hipMemcpyAsync(coords, ov.coords, sizeof(double) * (size_t)n0 * (size_t)dim, hipMemcpyDeviceToDevice, s));
  UPD_CHECK(hipMemcpyAsync(coords + n_pad * dim, x_new->coords, sizeof(double) * (size_t)m * (size_t)dim, kind, s));
  long long *ids = nullptr;
  if (ov.ids) {
    UPD_CHECK(hipMalloc(&tr.owned[1], sizeof(long long) * (size_t)n1));
    ids = static_cast<long long *>(tr.owned[1]);
    UPD_CHECK(hipMemcpyAsync(ids, ov.ids, sizeof(long long) * (size_t)n0, hipMemcpyDeviceToDevice, s));
    UPD_CHECK(hipMemcpyAsync(ids + n_pad, x_new->eq_id, sizeof(long long) * (size_t)m, kind, s));
  }
  double *scales = nullptr;
  if (nsc > 0) {
    UPD_CHECK(hipMalloc(&tr.owned[2], sizeof(double) * (size_t)n1 * (size_t)nsc));
    scales = static_cast<double *>(tr.owned[2]);
    for (int kc = 0; kc < nsc; ++kc) {
      UPD_CHECK(hipMemcpyAsync(scales + (long long)kc * n1, ov.scales + (long long)kc * scale_stride(ov), sizeof(double) * (size_t)n0,
                               hipMemcpyDeviceToDevice, s));
      UPD_CHECK(hipMemcpyAsync(scales + (long long)kc * n1 + n_pad, x_new->scales + (long long)kc * m, sizeof(double) * (size_t)m, kind, s));
    }
  }
  UPD_CHECK(hipMemcpyAsync(ynew_d, y_new, sizeof(double) * (size_t)m, kind, s));
  if (yvar_d) UPD_CHECK(hipMemcpyAsync(yvar_d, y_var_new, sizeof(double) * (size_t)m, kind, s));
  if (x_new->location == AGP_HOST) UPD_CHECK(hipStreamSynchronize(s));
  if (n_pad > n0)
    hipLaunchKernelGGL(replicate_feature_kernel, dim3((unsigned)((n_pad - n0 + 255) / 256)), dim3(256), 0, s, coords, dim, ids, scales,
                       (long long)n1, nsc, n0, n_pad - n0, n_pad);
  tr.v.coords = coords; tr.v.ids = ids; tr.v.scales = scales;
  tr.v.n = n1; tr.v.dim = dim; tr.v.nsc = nsc; tr.v.meas = 0; tr.v.sstride = 0;

  // ---- the old factor, its tile images and forward-substituted targets; phantom rows = identity ----
  UPD_CHECK(hipMemcpy2DAsync(fit->A, sizeof(double) * (size_t)lda, old->A, sizeof(double) * (size_t)old->lda, sizeof(double) * (size_t)n0,
                             (size_t)n0, hipMemcpyDeviceToDevice, s));
  UPD_CHECK(hipMemcpyAsync(fit->invd, old->invd, sizeof(double) * (size_t)nblk0 * (36 * MB * MB), hipMemcpyDeviceToDevice, s));
  launch_fill_block(s, fit->A, lda, n0, n_pad, 0, n_pad, true);   // phantom rows: e_i^T
  UPD_CHECK(hipMemsetAsync(fit->z, 0, sizeof(double) * (size_t)n1, s));
  UPD_CHECK(hipMemcpyAsync(fit->z, old->z, sizeof(double) * (size_t)n0, hipMemcpyDeviceToDevice, s));
  UPD_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
  UPD_CHECK(hipMemsetAsync(ctx->d_scalars, 0, 4 * sizeof(double), s));

  // ---- cross = covariance_function_(fit_.train_features, features)  (gp.hpp:395-396; plain features on both sides),
  //      written transposed where V^T belongs: rows n_pad.., columns 0..n_pad ----
  FeatView vnew = tr.v, vold = tr.v;
  vnew.coords = coords + n_pad * dim; vnew.ids = ids ? ids + n_pad : nullptr; vnew.scales = scales ? scales + n_pad : nullptr;
  vnew.n = m; vnew.sstride = n1;
  vold.n = n_pad; vold.sstride = n1;
  double *X = fit->A + n_pad;  // m x n_pad, ld = lda
  launch_gram(s, dprog, vnew, vold, false, false, X, lda, nullptr, ctx->d_flags, &k->prog);
  for (const auto &ph : fit->phantom) launch_fill_block(s, X, lda, 0, m, ph.first, ph.second, false);  // phantom columns
  // ---- X <- X L^-T  (= (A^-1 B)^T L: the block row V^T of the grown factor; block_symmetric.hpp:51) ----
  right_solve_lt(s, fit->A, n_pad, lda, fit->invd, X, m, lda);
  // ---- S = prior(features) + targets.covariance - V^T V   (gp.hpp:389-392 folded into one Schur complement) ----
  double *S = fit->A + n_pad * (lda + 1);
  launch_gram(s, dprog, vnew, vnew, true, true, S, lda, yvar_d, ctx->d_flags, &k->prog);
  launch_gemm_nt_sub(s, S, lda, X, lda, false, X, lda, false, m, m, n_pad, true);
  // ---- forward substitution of the new targets: y_new - V^T z_old, then through L_S (fused into its factorisation) ----
  st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes,
                 sizeof(double) * (((size_t)(n_pad + 1023) / 1024) * (size_t)m + backsolve_ws_elems(n1) + 16));
  if (st != AGP_OK) { (void)hipFree(ynew_d); agp_fit_destroy(fit); return st; }
  launch_matvec(s, X, lda, m, n_pad, fit->z, ctx->ws_aux, -1.0, 1.0, ynew_d, fit->z + n_pad);
  // ---- LL^T of S in place (the sub-matrix is addressed as a matrix of its own: its 128-blocks start at n_pad) ----
  factor_lower(ctx, S, m, lda, fit->invd + (n_pad / NB) * (long long)(36 * MB * MB), fit->z + n_pad, nullptr);
  UPD_CHECK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  UPD_CHECK(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  UPD_CHECK(hipStreamSynchronize(s));
  UPD_CHECK(hipGetLastError());
  (void)hipFree(ynew_d);
  ynew_d = nullptr;
  fit->log_det = old->log_det + 2. * ctx->h_scalars[0];
  if (ctx->h_flags[0]) { agp_fit_destroy(fit); return AGP_ERR_NAN_INPUT; }
  if (ctx->h_flags[1]) {
    fit->failed_pivot = fit_real_rows(old) + (int64_t)ctx->h_flags[1] - 1;
    *out = fit;  // the handle only reports the pivot
    return AGP_ERR_NOT_POSITIVE_DEFINITE;
  }
  // ---- information = L^-T z over the whole grown factor ----
  UPD_CHECK(hipMemcpyAsync(fit->alpha, fit->z, sizeof(double) * (size_t)n1, hipMemcpyDeviceToDevice, s));
  backward_solve_vec_any(s, fit->A, n1, lda, fit->invd, fit->alpha, ctx->ws_aux);
  UPD_CHECK(hipStreamSynchronize(s));
  UPD_CHECK(hipGetLastError());
#undef UPD_CHECK
  if (information && (st = fit_compact_vector(ctx, fit, fit->alpha, information, AGP_HOST)) != AGP_OK) { agp_fit_destroy(fit); return st; }
  if (log_det) *log_det = fit->log_det;
  *out = fit;
  return AGP_OK;
}

}  // extern "C"
