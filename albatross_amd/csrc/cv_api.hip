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

// ---- equal group sizes: every group in lock step through batched launches -------------------
struct UniformGroups {
  double *Gall = nullptr, *Ball = nullptr, *img = nullptr, *Q = nullptr, *vecs = nullptr;
  ~UniformGroups() { (void)hipFree(Gall); (void)hipFree(Ball); (void)hipFree(img); (void)hipFree(Q); (void)hipFree(vecs); }
};

bool uniform_groups(int64_t n_groups, const int64_t *offsets, long long *m_out) {
  if (n_groups < 2) return false;
  const long long m = offsets[1] - offsets[0];
  if (m <= 0) return false;
  for (int64_t g = 0; g < n_groups; ++g)
    if (offsets[g + 1] - offsets[g] != m) return false;
  *m_out = m;
  return true;
}

// Ball[g] (m x m slabs, ld ldb, stride ldb * m) = (K^-1)[I_g, I_g] for all groups: one gather, one batched product
int uniform_inverse_blocks(GroupWork *w, long long count, long long m, UniformGroups *u) {
  agp_context *ctx = w->ctx;
  hipStream_t s = ctx->stream;
  const long long total = count * m, ldb = factor_ld(m);
  AGP_HIP_CHECK(ctx, hipMalloc(&u->Gall, sizeof(double) * (size_t)w->ldg * (size_t)total));
  AGP_HIP_CHECK(ctx, hipMalloc(&u->Ball, sizeof(double) * (size_t)ldb * (size_t)total));
  launch_gather_cols(s, w->R, w->ldr, w->idx, total, 0, w->n, u->Gall, w->ldg);
  AGP_HIP_CHECK(ctx, hipMemsetAsync(u->Ball, 0, sizeof(double) * (size_t)ldb * (size_t)total, s));
  launch_gemm_nt_sub_batched(s, u->Ball, ldb, ldb * m, u->Gall, w->ldg, true, m * w->ldg, u->Gall, w->ldg, true, m * w->ldg, m,
                             m, w->n, false, count);
  launch_axpby(s, ldb * total, -1.0, u->Ball, 0.0, nullptr, u->Ball);
  (void)hipFree(u->Gall); u->Gall = nullptr;
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

// all columns of all slabs are ldb apart: the packed output is ONE pitched copy
int copy_out_slabs(agp_context *ctx, const double *slabs, long long ld, long long m, long long count, double *dst,
                   int location) {
  return copy_out_2d(ctx, slabs, ld, m, m * count, dst, m, location);
}

}  // namespace

int agp_fit_inverse_blocks(agp_context *ctx, const agp_fit *fit, int64_t n_groups, const int64_t *offsets,
                           const int64_t *indices, double *blocks, int out_location) {
  if (!ctx || !fit || !fit->A || !blocks) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GroupWork w;
  int st = group_work_init(ctx, fit, n_groups, offsets, indices, &w);
  if (st != AGP_OK) return st;
  long long mu = 0;
  if (uniform_groups(n_groups, offsets, &mu)) {
    UniformGroups u;
    if ((st = uniform_inverse_blocks(&w, n_groups, mu, &u)) != AGP_OK) return st;
    return copy_out_slabs(ctx, u.Ball, factor_ld(mu), mu, n_groups, blocks, out_location);
  }
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
  long long mu_sz = 0;
  if (uniform_groups(n_groups, offsets, &mu_sz)) {
    // equal group sizes: blocks, LL^T, inverses and solves of ALL groups in lock step (blockIdx.y = group)
    const long long m = mu_sz, count = n_groups, total = count * m;
    const long long ldb = factor_ld(m), nblk_b = (m + NB - 1) / NB, stride_B = ldb * m, stride_I = nblk_b * (36 * MB * MB);
    UniformGroups u;
    if ((st = uniform_inverse_blocks(&w, count, m, &u)) != AGP_OK) return st;
    AGP_HIP_CHECK(ctx, hipMalloc(&u.img, sizeof(double) * ((size_t)stride_I * (size_t)count + (size_t)round_up(count, 2))));
    AGP_HIP_CHECK(ctx, hipMalloc(&u.Q, sizeof(double) * (size_t)stride_B * (size_t)count));
    AGP_HIP_CHECK(ctx, hipMalloc(&u.vecs, sizeof(double) * 3 * (size_t)round_up(total, 2)));
    double *logsum = u.img + (size_t)stride_I * (size_t)count;
    double *vz = u.vecs, *xs = vz + round_up(total, 2), *outv = xs + round_up(total, 2);
    AGP_HIP_CHECK(ctx, hipMemsetAsync(logsum, 0, sizeof(double) * (size_t)round_up(count, 2), s));
    AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
    // v_g = subset(information, indices); z_g = L_g^-1 v_g rides along the factorisation   (:175,181-182)
    launch_gather_vec(s, fit->alpha, w.idx, total, nullptr, vz);
    factor_lower_batched(s, u.Ball, stride_B, m, ldb, u.img, stride_I, vz, m, count, ctx->d_flags, logsum);
    // R_g = L_g^-1 ;  A_g^-1 v_g = R_g^T z_g ;  inverse = R_g^T R_g
    launch_set_identity_batched(s, u.Q, ldb, stride_B, m, count);
    forward_solve_mat_batched(s, u.Ball, stride_B, m, ldb, u.img, stride_I, u.Q, stride_B, m, ldb, /*rhs_lower=*/true, count);
    launch_colvec_dot_batched(s, u.Q, ldb, stride_B, m, vz, m, count, xs);
    launch_gather_vec(s, yd, w.idx, total, xs, outv);  // mean = y - A^-1 v
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    if ((st = copy_out(ctx, outv, total, mean, location)) != AGP_OK) return st;
    if ((st = status_from_flags(ctx)) != AGP_OK) return st;
    if (variance) {  // diag(R^T R): the columns of all slabs are ldb apart
      launch_coldot(s, u.Q, ldb, u.Q, ldb, m, total, outv, -1.0, nullptr);
      if ((st = copy_out(ctx, outv, total, variance, location)) != AGP_OK) return st;
    }
    if (joint) {
      AGP_HIP_CHECK(ctx, hipMemsetAsync(u.Ball, 0, sizeof(double) * (size_t)stride_B * (size_t)count, s));
      launch_gemm_nt_sub_batched(s, u.Ball, ldb, stride_B, u.Q, ldb, true, stride_B, u.Q, ldb, true, stride_B, m, m, m, false,
                                 count);
      launch_axpby(s, stride_B * count, -1.0, u.Ball, 0.0, nullptr, u.Ball);
      if ((st = copy_out_slabs(ctx, u.Ball, ldb, m, count, joint, location)) != AGP_OK) return st;
    }
    AGP_HIP_CHECK(ctx, hipGetLastError());
    return AGP_OK;
  }
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
