// solver_api.hip — CovarianceRepresentation compositions ON THE DEVICE (C-ABI agp_solver_*): the solvers a fit can hold
// besides its own factor, and the generic form of _predict_impl over any of them.
//
// Reference work replaced:
//   BlockSymmetric<Solver>           include/albatross/src/linalg/block_symmetric.hpp:46-133  (FitModel::update, gp.hpp:384-414)
//   ExplainedCovariance              include/albatross/src/covariance_functions/representations.hpp:64-96
//                                    (fit_from_prediction, gp.hpp:139-153)
//   _predict_impl over a generic CovarianceRepresentation   include/albatross/src/models/gp.hpp:305-366
// Fits on the device LL^T are updated on the device (agp_fit_update) and predicted by agp_predict_*; the compositions
// here serve the solvers that are NOT a plain factor - updates of pivoted L D L^T fits, fit_from_prediction - whose
// block algebra used to be numpy between device solves.  Everything (the products with A^-1 B, with the inner matrix,
// the cross covariance, the explained covariance) now stays in HBM; a solve is device solves + MFMA products.
#include <algorithm>
#include <new>

#include "api_internal.h"

using namespace agp;

struct agp_solver {
  int kind = 0;  // 0: LL^T factor (agp_fit), 1: pivoted L D L^T (agp_ldlt), 2: BlockSymmetric, 3: ExplainedCovariance
  agp_context *ctx = nullptr;
  const agp_fit *fit = nullptr;    // borrowed
  const agp_ldlt *ldlt = nullptr;  // borrowed
  // BlockSymmetric: solver of [[A, B], [B^T, C]] from a solver of A, Ai_B = A^-1 B (na x nb, ld = na, owned) and the solver
  // of the Schur complement S = C - B^T A^-1 B (both sub-solvers borrowed: they must outlive this object)
  const agp_solver *A = nullptr, *S = nullptr;
  double *AiB = nullptr;
  // ExplainedCovariance: S^-1 = outer^-1 inner outer^-1 (outer borrowed; inner n x n, ld = n, owned)
  const agp_solver *outer = nullptr;
  double *inner = nullptr;
  long long n = 0, na = 0, nb = 0;
};

namespace {

#define SOL_HIP(expr)                                                        \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);   \
      return AGP_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)

struct DevBuf {  // scratch that goes with the scope (through the caching allocator of api.hip)
  double *p = nullptr;
  ~DevBuf() { if (p) (void)dev_free(p); }
  hipError_t get(size_t elems) { return dev_malloc(&p, sizeof(double) * std::max<size_t>(elems, 1)); }
};

// out (n x r, ld = n) = solver^-1 rhs (n x r, ld = n), both on the device
int solve_dev(agp_context *ctx, const agp_solver *sv, const double *rhs, long long r, double *out) {
  if (r <= 0) return AGP_OK;
  hipStream_t s = ctx->stream;
  switch (sv->kind) {
  case 0: return agp_solve(ctx, sv->fit, rhs, r, out, AGP_DEVICE);
  case 1: return agp_ldlt_solve(ctx, sv->ldlt, rhs, r, out, AGP_DEVICE);
  case 2: {  // block_symmetric.hpp:75-98
    const long long na = sv->na, nb = sv->nb, n = sv->n;
    DevBuf ra, rb, t1, si1, sib, xa;
    SOL_HIP(ra.get((size_t)na * r));
    SOL_HIP(rb.get((size_t)nb * r));
    SOL_HIP(t1.get((size_t)nb * r));
    SOL_HIP(si1.get((size_t)nb * r));
    SOL_HIP(sib.get((size_t)nb * r));
    SOL_HIP(xa.get((size_t)na * r));
    SOL_HIP(hipMemcpy2DAsync(ra.p, sizeof(double) * (size_t)na, rhs, sizeof(double) * (size_t)n, sizeof(double) * (size_t)na, (size_t)r,
                             hipMemcpyDeviceToDevice, s));
    SOL_HIP(hipMemcpy2DAsync(rb.p, sizeof(double) * (size_t)nb, rhs + na, sizeof(double) * (size_t)n, sizeof(double) * (size_t)nb,
                             (size_t)r, hipMemcpyDeviceToDevice, s));
    // Bt_Ai_rhs = Ai_B^T rhs_a   (nb x r):  t1 = 0 - Ai_B^T rhs_a, then negated
    SOL_HIP(hipMemsetAsync(t1.p, 0, sizeof(double) * (size_t)nb * (size_t)r, s));
    launch_gemm_nt_sub(s, t1.p, nb, sv->AiB, na, true, ra.p, na, true, nb, r, na, false);
    launch_axpby(s, nb * r, -1.0, t1.p, 0.0, t1.p, t1.p);
    int st = solve_dev(ctx, sv->S, t1.p, r, si1.p);              // Si_Bt_Ai_rhs
    if (st == AGP_OK) st = solve_dev(ctx, sv->S, rb.p, r, sib.p);  // Si_rhs_b
    if (st == AGP_OK) st = solve_dev(ctx, sv->A, ra.p, r, xa.p);   // Ai_rhs_a
    if (st != AGP_OK) return st;
    // d = Si_rhs_b - Si_Bt_Ai_rhs  = the b part of the answer;  a part = Ai_rhs_a - Ai_B d
    launch_axpby(s, nb * r, 1.0, sib.p, -1.0, si1.p, t1.p);
    launch_gemm_nt_sub(s, xa.p, na, sv->AiB, na, false, t1.p, nb, true, na, r, nb, false);
    SOL_HIP(hipMemcpy2DAsync(out, sizeof(double) * (size_t)n, xa.p, sizeof(double) * (size_t)na, sizeof(double) * (size_t)na, (size_t)r,
                             hipMemcpyDeviceToDevice, s));
    SOL_HIP(hipMemcpy2DAsync(out + na, sizeof(double) * (size_t)n, t1.p, sizeof(double) * (size_t)nb, sizeof(double) * (size_t)nb,
                             (size_t)r, hipMemcpyDeviceToDevice, s));
    SOL_HIP(hipStreamSynchronize(s));
    return AGP_OK;
  }
  case 3: {  // representations.hpp:80-82: outer^-1 (inner (outer^-1 rhs))
    const long long n = sv->n;
    DevBuf t, u;
    SOL_HIP(t.get((size_t)n * r));
    SOL_HIP(u.get((size_t)n * r));
    int st = solve_dev(ctx, sv->outer, rhs, r, t.p);
    if (st != AGP_OK) return st;
    SOL_HIP(hipMemsetAsync(u.p, 0, sizeof(double) * (size_t)n * (size_t)r, s));
    launch_gemm_nt_sub(s, u.p, n, sv->inner, n, false, t.p, n, true, n, r, n, false);  // u = -inner t
    launch_axpby(s, n * r, -1.0, u.p, 0.0, u.p, u.p);
    st = solve_dev(ctx, sv->outer, u.p, r, out);
    SOL_HIP(hipStreamSynchronize(s));
    return st;
  }
  default: return AGP_ERR_INVALID_ARGUMENT;
  }
}

int to_dev_matrix(agp_context *ctx, const double *src, long long rows, long long cols, long long ld, int location, double *dst) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  SOL_HIP(hipMemcpy2DAsync(dst, sizeof(double) * (size_t)rows, src, sizeof(double) * (size_t)ld, sizeof(double) * (size_t)rows, (size_t)cols,
                           kind, ctx->stream));
  SOL_HIP(hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

}  // namespace

extern "C" {

int agp_solver_from_fit(agp_context *ctx, const agp_fit *fit, agp_solver **out) {
  if (!ctx || !fit || !out || fit->failed_pivot >= 0 || !fit->A) return AGP_ERR_INVALID_ARGUMENT;
  agp_solver *s = new (std::nothrow) agp_solver();
  if (!s) return AGP_ERR_INVALID_ARGUMENT;
  s->kind = 0; s->ctx = ctx; s->fit = fit; s->n = fit_real_rows(fit);
  *out = s;
  return AGP_OK;
}

int agp_solver_from_ldlt(agp_context *ctx, const agp_ldlt *ldlt, agp_solver **out) {
  if (!ctx || !ldlt || !out) return AGP_ERR_INVALID_ARGUMENT;
  agp_solver *s = new (std::nothrow) agp_solver();
  if (!s) return AGP_ERR_INVALID_ARGUMENT;
  s->kind = 1; s->ctx = ctx; s->ldlt = ldlt; s->n = agp_ldlt_size(ldlt);
  *out = s;
  return AGP_OK;
}

int64_t agp_solver_rows(const agp_solver *s) { return s ? s->n : 0; }

void agp_solver_destroy(agp_solver *s) {
  if (!s) return;
  if (s->ctx) (void)hipSetDevice(s->ctx->device);
  if (s->AiB) (void)dev_free(s->AiB);
  if (s->inner) (void)dev_free(s->inner);
  delete s;
}

int agp_solver_block_symmetric(agp_context *ctx, const agp_solver *A, const double *B, int64_t ldb, int location, const agp_solver *S,
                               agp_solver **out) {
  if (!ctx || !A || !B || !S || !out || ldb < A->n || A->ctx != ctx || S->ctx != ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long na = A->n, nb = S->n;
  agp_solver *s = new (std::nothrow) agp_solver();
  if (!s) return AGP_ERR_INVALID_ARGUMENT;
  s->kind = 2; s->ctx = ctx; s->A = A; s->S = S; s->na = na; s->nb = nb; s->n = na + nb;
  DevBuf Bd;
  if (Bd.get((size_t)na * nb) != hipSuccess || dev_malloc(&s->AiB, sizeof(double) * (size_t)std::max<long long>(na * nb, 1)) != hipSuccess) {
    agp_solver_destroy(s);
    ctx->last_error = "agp_solver_block_symmetric: allocation";
    return AGP_ERR_HIP;
  }
  int st = to_dev_matrix(ctx, B, na, nb, ldb, location, Bd.p);
  if (st == AGP_OK) st = solve_dev(ctx, A, Bd.p, nb, s->AiB);  // Ai_B = A.solve(B), block_symmetric.hpp:51
  if (st != AGP_OK) { agp_solver_destroy(s); return st; }
  *out = s;
  return AGP_OK;
}

int agp_solver_explained(agp_context *ctx, const agp_solver *outer, const double *inner, int64_t ld, int location, agp_solver **out) {
  if (!ctx || !outer || !inner || !out || ld < outer->n || outer->ctx != ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = outer->n;
  agp_solver *s = new (std::nothrow) agp_solver();
  if (!s) return AGP_ERR_INVALID_ARGUMENT;
  s->kind = 3; s->ctx = ctx; s->outer = outer; s->n = n;
  if (dev_malloc(&s->inner, sizeof(double) * (size_t)std::max<long long>(n * n, 1)) != hipSuccess) {
    agp_solver_destroy(s);
    ctx->last_error = "agp_solver_explained: allocation";
    return AGP_ERR_HIP;
  }
  const int st = to_dev_matrix(ctx, inner, n, n, ld, location, s->inner);
  if (st != AGP_OK) { agp_solver_destroy(s); return st; }
  *out = s;
  return AGP_OK;
}

int agp_solver_solve(agp_context *ctx, const agp_solver *sv, const double *rhs, int64_t nrhs, double *out, int location) {
  if (!ctx || !sv || !rhs || !out || nrhs < 0 || sv->ctx != ctx) return AGP_ERR_INVALID_ARGUMENT;
  if (nrhs == 0 || sv->n == 0) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (location == AGP_DEVICE) return solve_dev(ctx, sv, rhs, nrhs, out);
  DevBuf r, o;
  if (r.get((size_t)sv->n * nrhs) != hipSuccess || o.get((size_t)sv->n * nrhs) != hipSuccess) { ctx->last_error = "agp_solver_solve: allocation"; return AGP_ERR_HIP; }
  int st = to_dev_matrix(ctx, rhs, sv->n, nrhs, sv->n, AGP_HOST, r.p);
  if (st == AGP_OK) st = solve_dev(ctx, sv, r.p, nrhs, o.p);
  if (st == AGP_OK) st = copy_out(ctx, o.p, sv->n * nrhs, out, AGP_HOST);
  return st;
}

// _predict_impl over a generic CovarianceRepresentation (gp.hpp:305-366): mode 0 mean (var_or_cov unused), 1 marginal,
// 2 joint (cov m x m, ld = m).  train: the fit's training features (as the covariance function sees them), information: n
// doubles (at `location`).  Mean functions are the caller's business (added on the host, like everywhere).
int agp_solver_predict(agp_context *ctx, const agp_kernel *k, const agp_solver *sv, const agp_features *train, const double *information,
                       const agp_features *xs, double *mean, double *var_or_cov, int mode, int location) {
  if (!ctx || !k || !sv || !train || !information || !xs || !mean || mode < 0 || mode > 2 || (mode > 0 && !var_or_cov))
    return AGP_ERR_INVALID_ARGUMENT;
  if (sv->ctx != ctx) return AGP_ERR_INVALID_ARGUMENT;  // a solver lives in the context (and on the device) that made it
  {
    int stv = validate_features(train);
    if (stv != AGP_OK || (stv = validate_features(xs)) != AGP_OK) return stv;
  }
  if (train->n != sv->n || xs->dim != train->dim) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = sv->n, m = xs->n;
  if (m == 0) return AGP_OK;
  const DevProgram *dprog = nullptr;
  int st = device_program(ctx, k, &dprog);
  if (st != AGP_OK) return st;
  DeviceFeatures dtr, dxs;
  if ((st = to_device(ctx, train, false, &dtr)) != AGP_OK) return st;
  if ((st = to_device(ctx, xs, false, &dxs)) != AGP_OK) return st;
  hipStream_t s = ctx->stream;
  DevBuf cross, expl, info, mean_d, prior;
  if (cross.get((size_t)n * m) != hipSuccess || info.get((size_t)n) != hipSuccess || mean_d.get((size_t)m) != hipSuccess ||
      (mode > 0 && expl.get((size_t)n * m) != hipSuccess) || (mode > 0 && prior.get(mode == 2 ? (size_t)m * m : (size_t)m) != hipSuccess)) {
    ctx->last_error = "agp_solver_predict: allocation";
    return AGP_ERR_HIP;
  }
  if ((st = vector_to_device(ctx, information, n, location, info.p)) != AGP_OK) return st;
  // cross_cov = cov(train_features, features); mean = cross_cov^T information   (gp.hpp:316,337,361-363)
  launch_gram(s, dprog, dtr.v, dxs.v, false, false, cross.p, n, nullptr, nullptr, &k->prog);
  launch_colvec_dot(s, cross.p, n, n, m, info.p, 1.0, 0.0, nullptr, mean_d.p);
  if ((st = copy_out(ctx, mean_d.p, m, mean, location)) != AGP_OK) return st;
  if (mode == 0) return AGP_OK;
  if ((st = solve_dev(ctx, sv, cross.p, m, expl.p)) != AGP_OK) return st;  // train_covariance.solve(cross_cov), gp.hpp:96,111
  if (mode == 1) {
    launch_gram_diagonal(s, dprog, dxs.v, prior.p);                          // gp.hpp:339-343
    launch_coldot(s, expl.p, n, cross.p, n, n, m, prior.p, 1.0, prior.p);    // prior - colsum(explained o cross), gp.hpp:97-99
    return copy_out(ctx, prior.p, m, var_or_cov, location);
  }
  launch_gram(s, dprog, dxs.v, dxs.v, true, false, prior.p, m, nullptr, nullptr, &k->prog);    // prior_cov, gp.hpp:317
  launch_gemm_nt_sub(s, prior.p, m, cross.p, n, true, expl.p, n, true, m, m, n, false);       // - cross^T explained, gp.hpp:111
  return copy_out(ctx, prior.p, m * m, var_or_cov, location);
}

// FitModel::update on a generic representation (gp.hpp:403-407): the information vector of the conditioned fit,
//   [ information - Ai_B Si_delta ; Si_delta ],
// with the Ai_B = A^-1 B the BlockSymmetric solver already holds in HBM (block_symmetric.hpp:51): one mat-vec on the device.
// information: na doubles, si_delta: nb doubles, out: na + nb doubles, all at `location`.
int agp_solver_update_information(agp_context *ctx, const agp_solver *bs, const double *information, const double *si_delta, double *out,
                                  int location) {
  if (!ctx || !bs || bs->kind != 2 || !bs->AiB || !information || !si_delta || !out) return AGP_ERR_INVALID_ARGUMENT;
  if (bs->ctx != ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long na = bs->na, nb = bs->nb;
  DevBuf v;
  if (v.get((size_t)(2 * na + nb)) != hipSuccess) { ctx->last_error = "agp_solver_update_information: allocation"; return AGP_ERR_HIP; }
  double *info = v.p, *res = v.p + na;  // res: na + nb
  int st = vector_to_device(ctx, information, na, location, info);
  if (st == AGP_OK) st = vector_to_device(ctx, si_delta, nb, location, res + na);
  if (st != AGP_OK) return st;
  // res[0 : na] = information - Ai_B (na x nb, ld = na) Si_delta
  launch_tall_matvec(ctx->stream, bs->AiB, na, na, nb, res + na, -1.0, 1.0, info, res);
  return copy_out(ctx, res, na + nb, out, location);
}

// _predict_impl over a generic representation (gp.hpp:305-366) for LinearCombination<X> features on either side
// (covariance_functions/callers.hpp:321-396): as agp_solver_predict, with every covariance matrix the contracted one
// of agp_gram_combined - train / xs hold the EXPANDED points, combination a of a side = its expanded points
// offsets[a] .. offsets[a + 1) with coefficients[..] (host arrays; offsets == NULL: plain features on that side).
// The solver's size must equal the number of training combinations.  Everything stays in HBM.
int agp_solver_predict_combined(agp_context *ctx, const agp_kernel *k, const agp_solver *sv, const agp_features *train, int64_t n_train,
                                const int64_t *train_offsets, const double *train_coefficients, const double *information,
                                const agp_features *xs, int64_t n_xs, const int64_t *xs_offsets, const double *xs_coefficients,
                                double *mean, double *var_or_cov, int mode, int location) {
  if (!ctx || !k || !sv || !train || !information || !xs || !mean || mode < 0 || mode > 2 || (mode > 0 && !var_or_cov))
    return AGP_ERR_INVALID_ARGUMENT;
  if (sv->ctx != ctx || xs->dim != train->dim) return AGP_ERR_INVALID_ARGUMENT;
  int st = validate_features(train);
  if (st != AGP_OK || (st = validate_features(xs)) != AGP_OK) return st;
  const long long n = train_offsets ? n_train : train->n, m = xs_offsets ? n_xs : xs->n;
  if (n != sv->n) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (m == 0) return AGP_OK;
  hipStream_t s = ctx->stream;
  DevBuf cross, expl, info, mean_d, prior, pdiag;
  if (cross.get((size_t)n * m) != hipSuccess || info.get((size_t)n) != hipSuccess || mean_d.get((size_t)m) != hipSuccess ||
      (mode > 0 && expl.get((size_t)n * m) != hipSuccess) || (mode > 0 && prior.get((size_t)m * m) != hipSuccess) ||
      (mode == 1 && pdiag.get((size_t)m) != hipSuccess)) {
    ctx->last_error = "agp_solver_predict_combined: allocation";
    return AGP_ERR_HIP;
  }
  if ((st = vector_to_device(ctx, information, n, location, info.p)) != AGP_OK) return st;
  // cross_cov = cov(train_features, features) through LinearCombinationCaller; mean = cross_cov^T information
  if ((st = agp_gram_combined(ctx, k, train, n_train, train_offsets, train_coefficients, xs, n_xs, xs_offsets, xs_coefficients, cross.p, n,
                              AGP_DEVICE)) != AGP_OK) return st;
  launch_colvec_dot(s, cross.p, n, n, m, info.p, 1.0, 0.0, nullptr, mean_d.p);
  if ((st = copy_out(ctx, mean_d.p, m, mean, location)) != AGP_OK) return st;
  if (mode == 0) return AGP_OK;
  if ((st = solve_dev(ctx, sv, cross.p, m, expl.p)) != AGP_OK) return st;  // train_covariance.solve(cross_cov), gp.hpp:96,111
  if ((st = agp_gram_combined(ctx, k, xs, n_xs, xs_offsets, xs_coefficients, nullptr, 0, nullptr, nullptr, prior.p, m, AGP_DEVICE)) != AGP_OK)
    return st;                                                              // prior_cov, gp.hpp:317,339-343
  if (mode == 1) {
    SOL_HIP(hipMemcpy2DAsync(pdiag.p, sizeof(double), prior.p, sizeof(double) * (size_t)(m + 1), sizeof(double), (size_t)m,
                             hipMemcpyDeviceToDevice, s));                  // its diagonal
    launch_coldot(s, expl.p, n, cross.p, n, n, m, pdiag.p, 1.0, pdiag.p);   // prior - colsum(explained o cross), gp.hpp:97-99
    return copy_out(ctx, pdiag.p, m, var_or_cov, location);
  }
  launch_gemm_nt_sub(s, prior.p, m, cross.p, n, true, expl.p, n, true, m, m, n, false);  // - cross^T explained, gp.hpp:111
  return copy_out(ctx, prior.p, m * m, var_or_cov, location);
}

}  // extern "C"
