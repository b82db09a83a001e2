// cv_api.hip — leave-one-GROUP-out entry points of the C-ABI (include/albatross_amd.h).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>

#include "api_internal.h"

using namespace agp;

extern "C" {

// ---- leave-one-GROUP-out -------------------------------------------------------
// SerializableLDLT::inverse_blocks (serializable_ldlt.hpp:137-179) and held_out_predictions
// (cross_validation_utils.hpp:165-232).  R = L^-1 is built once (N^3/3 flop on MFMA, the solve
// kernels on a triangular right-hand side); per group the columns I_g are gathered and
// B_g = G^T G = (K^-1)[I_g, I_g] is one MFMA product; the |g| x |g| system is then factored with
// the same LL^T kernels.
namespace {

struct GroupWork {
  agp_context *ctx = nullptr;
  double *R = nullptr, *G = nullptr, *B = nullptr, *tmp = nullptr;
  long long *idx = nullptr;
  long long n = 0, ldr = 0, ldg = 0, ldb = 0, mmax = 0;
  ~GroupWork() {
    (void)hipFree(R); (void)hipFree(G); (void)hipFree(B); (void)hipFree(tmp); (void)hipFree(idx);
  }
};

int group_work_init(agp_context *ctx, const agp_fit *fit, int64_t n_groups, const int64_t *offsets,
                    const int64_t *indices, GroupWork *w) {
  const long long n = fit->n;
  if (n_groups < 0 || !offsets || offsets[0] != 0) return AGP_ERR_INVALID_ARGUMENT;
  long long mmax = 0;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long m = offsets[g + 1] - offsets[g];
    if (m < 0) return AGP_ERR_INVALID_ARGUMENT;
    if (m > mmax) mmax = m;
  }
  const long long total = offsets[n_groups];
  if (total > 0 && !indices) return AGP_ERR_INVALID_ARGUMENT;
  for (long long i = 0; i < total; ++i)
    if (indices[i] < 0 || indices[i] >= n) return AGP_ERR_INVALID_ARGUMENT;
  w->ctx = ctx; w->n = n; w->mmax = mmax;
  if (total == 0) return AGP_OK;
  w->ldr = factor_ld(n); w->ldg = round_up(n, 2); w->ldb = factor_ld(mmax);
  AGP_HIP_CHECK(ctx, hipMalloc(&w->R, sizeof(double) * (size_t)w->ldr * (size_t)n));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->G, sizeof(double) * (size_t)w->ldg * (size_t)mmax));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->B, sizeof(double) * (size_t)w->ldb * (size_t)mmax));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->tmp, sizeof(double) * (size_t)(4 * round_up(mmax, 2) + 2 * round_up(n, 2))));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->idx, sizeof(long long) * (size_t)total));
  static_assert(sizeof(long long) == sizeof(int64_t), "index width");
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(w->idx, indices, sizeof(long long) * (size_t)total, hipMemcpyHostToDevice, ctx->stream));
  hipStream_t s = ctx->stream;
  launch_set_identity(s, w->R, w->ldr, n);
  forward_solve_mat(s, fit->A, n, fit->lda, fit->invd, w->R, n, w->ldr, /*rhs_lower=*/true);
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

// B (m x m, w->ldb) = (K^-1)[I_g, I_g] on the device
int group_inverse_block(GroupWork *w, const int64_t *indices, long long off, long long m) {
  agp_context *ctx = w->ctx;
  hipStream_t s = ctx->stream;
  long long row0 = w->n;
  for (long long a = 0; a < m; ++a)
    if (indices[off + a] < row0) row0 = indices[off + a];
  row0 &= ~1LL;  // column j of R is zero above row j: only rows >= min(I_g) contribute
  launch_gather_cols(s, w->R, w->ldr, w->idx + off, m, row0, w->n, w->G, w->ldg);
  AGP_HIP_CHECK(ctx, hipMemsetAsync(w->B, 0, sizeof(double) * (size_t)w->ldb * (size_t)m, s));
  // B -= G^T G (k-major operands), then negate
  launch_gemm_nt_sub(s, w->B, w->ldb, w->G + row0, w->ldg, true, w->G + row0, w->ldg, true, m, m, w->n - row0, false);
  launch_negate(s, w->B, w->ldb, m, nullptr);
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

}  // namespace

int agp_fit_inverse_blocks(agp_context *ctx, const agp_fit *fit, int64_t n_groups, const int64_t *offsets,
                           const int64_t *indices, double *blocks, int out_location) {
  if (!ctx || !fit || !fit->A || !blocks) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GroupWork w;
  int st = group_work_init(ctx, fit, n_groups, offsets, indices, &w);
  if (st != AGP_OK) return st;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long off = offsets[g], m = offsets[g + 1] - off;
    if (m == 0) continue;
    if ((st = group_inverse_block(&w, indices, off, m)) != AGP_OK) return st;
    if ((st = copy_out_2d(ctx, w.B, w.ldb, m, m, blocks, m, out_location)) != AGP_OK) return st;
    blocks += m * m;
  }
  return AGP_OK;
}

int agp_held_out_predictions(agp_context *ctx, const agp_fit *fit, const double *y, int64_t n_groups,
                             const int64_t *offsets, const int64_t *indices, double *mean, double *variance,
                             double *joint, int location) {
  if (!ctx || !fit || !fit->A || !fit->alpha || !y || !mean) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GroupWork w;
  int st = group_work_init(ctx, fit, n_groups, offsets, indices, &w);
  if (st != AGP_OK) return st;
  if (w.mmax == 0) return AGP_OK;
  const long long n = fit->n, mp = round_up(w.mmax, 2);
  double *v = w.tmp, *x = v + mp, *mu = x + mp, *var = mu + mp, *yd = var + mp;
  if ((st = vector_to_device(ctx, y, n, location, yd)) != AGP_OK) return st;
  hipStream_t s = ctx->stream;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long off = offsets[g], m = offsets[g + 1] - off;
    if (m == 0) continue;
    if ((st = group_inverse_block(&w, indices, off, m)) != AGP_OK) return st;
    // A_ldlt = SerializableLDLT(inverse_block)   (cross_validation_utils.hpp:181)
    agp_fit *fb = nullptr;
    st = agp_factor_create(ctx, w.B, m, w.ldb, 0, AGP_DEVICE, &fb);
    if (st != AGP_OK) { if (fb) agp_fit_destroy(fb); return st; }
    // mean = y - A_ldlt.solve(v), v = subset(information, indices)   (:175,182)
    launch_gather_vec(s, fit->alpha, w.idx + off, m, nullptr, v);
    st = agp_solve(ctx, fb, v, 1, x, AGP_DEVICE);
    if (st == AGP_OK) {
      launch_gather_vec(s, yd, w.idx + off, m, x, mu);
      st = copy_out(ctx, mu, m, mean + off, location);
    }
    if (st == AGP_OK && (variance || joint)) {
      // R_B = L_B^-1 ; inverse = R_B^T R_B  (inverse_diagonal :183 / inverse :192)
      const long long ldq = factor_ld(m);
      st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * 2 * (size_t)ldq * (size_t)m);
      if (st == AGP_OK) {
        double *Q = ctx->ws_aux, *J = Q + (size_t)ldq * (size_t)m;
        launch_set_identity(s, Q, ldq, m);
        forward_solve_mat(s, fb->A, m, fb->lda, fb->invd, Q, m, ldq, /*rhs_lower=*/true);
        if (joint) {
          (void)hipMemsetAsync(J, 0, sizeof(double) * (size_t)ldq * (size_t)m, s);
          launch_gemm_nt_sub(s, J, ldq, Q, ldq, true, Q, ldq, true, m, m, m, false);
          launch_negate(s, J, ldq, m, var);
          st = copy_out_2d(ctx, J, ldq, m, m, joint, m, location);
          joint += m * m;
        } else {
          launch_coldot(s, Q, ldq, Q, ldq, m, m, var, -1.0, nullptr);
        }
        if (st == AGP_OK && variance) st = copy_out(ctx, var, m, variance + off, location);
      }
    }
    agp_fit_destroy(fb);
    if (st != AGP_OK) return st;
  }
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

}  // extern "C"
