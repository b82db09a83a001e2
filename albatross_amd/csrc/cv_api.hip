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
  if (!fit->phantom.empty()) return AGP_ERR_UNSUPPORTED;  // fits grown by agp_fit_update: cross-validate a fresh fit
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
  forward_solve_mat_lookahead(ctx, fit->A, n, fit->lda, fit->invd, w->R, n, w->ldr, /*rhs_lower=*/true);
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

// ---- lock-step path: every group advances through the same batched launches (blockIdx.y = group) ----
// Groups of one size use slabs of that size; ragged groups of comparable size are padded to the largest
// one (padding columns of G are zero, the padded diagonal of B_g is set to one: B_pad = [B 0; 0 I]).
struct LockStep {
  double *Gall = nullptr, *Ball = nullptr, *img = nullptr, *Q = nullptr, *vecs = nullptr, *packed = nullptr;
  long long *meta = nullptr;  // idx_pad (count * smax) | off (count + 1) | boff (count + 1)
  long long smax = 0, count = 0, total = 0, padded = 0, elems = 0;
  const long long *idx_pad = nullptr, *off_d = nullptr, *boff_d = nullptr;
  ~LockStep() {
    (void)hipFree(Gall); (void)hipFree(Ball); (void)hipFree(img); (void)hipFree(Q); (void)hipFree(vecs); (void)hipFree(packed);
    (void)hipFree(meta);
  }
};

// true when the lock-step path applies: >= 2 non-empty groups whose padded size stays within 3x
bool lock_step_plan(GroupWork *w, int64_t n_groups, const int64_t *offsets, const int64_t *indices, LockStep *u) {
  if (n_groups < 2) return false;
  long long smax = 0, total = offsets[n_groups];
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long m = offsets[g + 1] - offsets[g];
    if (m <= 0) return false;
    if (m > smax) smax = m;
  }
  if (smax * n_groups > 3 * total) return false;
  u->smax = smax; u->count = n_groups; u->total = total; u->padded = smax * n_groups;
  std::vector<long long> meta((size_t)u->padded + 2 * (size_t)(n_groups + 1), -1);
  long long *off = meta.data() + u->padded, *boff = off + (n_groups + 1);
  long long e = 0;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long m = offsets[g + 1] - offsets[g];
    for (long long a = 0; a < m; ++a) meta[(size_t)(g * smax + a)] = indices[offsets[g] + a];
    off[g] = offsets[g];
    boff[g] = e;
    e += m * m;
  }
  off[n_groups] = total;
  boff[n_groups] = e;
  u->elems = e;
  agp_context *ctx = w->ctx;
  if (hipMalloc(&u->meta, sizeof(long long) * meta.size()) != hipSuccess) return false;
  if (hipMemcpy(u->meta, meta.data(), sizeof(long long) * meta.size(), hipMemcpyHostToDevice) != hipSuccess) return false;
  (void)ctx;
  u->idx_pad = u->meta;
  u->off_d = u->meta + u->padded;
  u->boff_d = u->off_d + (n_groups + 1);
  return true;
}

// Ball[g] (smax x smax slabs, ld ldb, stride ldb * smax) = [(K^-1)[I_g, I_g] 0; 0 I]: one gather, one batched product
int lock_step_inverse_blocks(GroupWork *w, LockStep *u) {
  agp_context *ctx = w->ctx;
  hipStream_t s = ctx->stream;
  const long long m = u->smax, ldb = factor_ld(m);
  AGP_HIP_CHECK(ctx, hipMalloc(&u->Gall, sizeof(double) * (size_t)w->ldg * (size_t)u->padded));
  AGP_HIP_CHECK(ctx, hipMalloc(&u->Ball, sizeof(double) * (size_t)ldb * (size_t)u->padded));
  launch_gather_cols(s, w->R, w->ldr, u->idx_pad, u->padded, 0, w->n, u->Gall, w->ldg);
  AGP_HIP_CHECK(ctx, hipMemsetAsync(u->Ball, 0, sizeof(double) * (size_t)ldb * (size_t)u->padded, s));
  launch_gemm_nt_sub_batched(s, u->Ball, ldb, ldb * m, u->Gall, w->ldg, true, m * w->ldg, u->Gall, w->ldg, true, m * w->ldg, m,
                             m, w->n, false, u->count);
  launch_axpby(s, ldb * u->padded, -1.0, u->Ball, 0.0, nullptr, u->Ball);
  if (u->padded != u->total) launch_pad_identity(s, u->Ball, ldb, ldb * m, u->off_d, m, u->count);
  (void)hipFree(u->Gall); u->Gall = nullptr;
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

// the packed (ragged) blocks of all slabs to the caller
int copy_out_blocks(agp_context *ctx, LockStep *u, const double *slabs, long long ld, double *dst, int location) {
  const long long m = u->smax;
  if (u->padded == u->total)  // one size: all columns of all slabs are ld apart, ONE pitched copy
    return copy_out_2d(ctx, slabs, ld, m, m * u->count, dst, m, location);
  if (!u->packed) AGP_HIP_CHECK(ctx, hipMalloc(&u->packed, sizeof(double) * (size_t)u->elems));
  launch_compact_blocks(ctx->stream, slabs, ld, ld * m, u->off_d, u->boff_d, m, u->count, u->packed);
  return copy_out(ctx, u->packed, u->elems, dst, location);
}

// padded vector (count * smax) -> the caller's compact vector (total)
int copy_out_vector(agp_context *ctx, LockStep *u, const double *padded_vec, double *scratch, double *dst, int location) {
  if (u->padded == u->total) return copy_out(ctx, padded_vec, u->total, dst, location);
  launch_pad_columns(ctx->stream, padded_vec, 1, u->off_d, u->smax, u->count, 1, scratch, 1, 1);
  return copy_out(ctx, scratch, u->total, dst, location);
}

}  // namespace

int agp_fit_inverse_blocks(agp_context *ctx, const agp_fit *fit, int64_t n_groups, const int64_t *offsets,
                           const int64_t *indices, double *blocks, int out_location) {
  if (!ctx || !fit || !fit->A || !blocks) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GroupWork w;
  int st = group_work_init(ctx, fit, n_groups, offsets, indices, &w);
  if (st != AGP_OK) return st;
  {
    LockStep u;
    if (offsets[n_groups] > 0 && lock_step_plan(&w, n_groups, offsets, indices, &u)) {
      if ((st = lock_step_inverse_blocks(&w, &u)) != AGP_OK) return st;
      return copy_out_blocks(ctx, &u, u.Ball, factor_ld(u.smax), blocks, out_location);
    }
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
  LockStep u;
  if (lock_step_plan(&w, n_groups, offsets, indices, &u)) {
    // blocks, LL^T, inverses and solves of ALL groups in lock step (blockIdx.y = group)
    const long long m = u.smax, count = u.count, padded = u.padded;
    const long long ldb = factor_ld(m), nblk_b = (m + NB - 1) / NB, stride_B = ldb * m, stride_I = nblk_b * (36 * MB * MB);
    if ((st = lock_step_inverse_blocks(&w, &u)) != AGP_OK) return st;
    AGP_HIP_CHECK(ctx, hipMalloc(&u.img, sizeof(double) * ((size_t)stride_I * (size_t)count + (size_t)round_up(count, 2))));
    AGP_HIP_CHECK(ctx, hipMalloc(&u.Q, sizeof(double) * (size_t)stride_B * (size_t)count));
    AGP_HIP_CHECK(ctx, hipMalloc(&u.vecs, sizeof(double) * 4 * (size_t)round_up(padded, 2)));
    double *logsum = u.img + (size_t)stride_I * (size_t)count;
    double *vz = u.vecs, *xs = vz + round_up(padded, 2), *outv = xs + round_up(padded, 2), *compact = outv + round_up(padded, 2);
    AGP_HIP_CHECK(ctx, hipMemsetAsync(logsum, 0, sizeof(double) * (size_t)round_up(count, 2), s));
    AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
    // v_g = subset(information, indices); z_g = L_g^-1 v_g rides along the factorisation   (:175,181-182)
    launch_gather_vec(s, fit->alpha, u.idx_pad, padded, nullptr, vz);
    factor_lower_batched(s, u.Ball, stride_B, m, ldb, u.img, stride_I, vz, m, count, ctx->d_flags, logsum);
    // R_g = L_g^-1 ;  A_g^-1 v_g = R_g^T z_g ;  inverse = R_g^T R_g
    launch_set_identity_batched(s, u.Q, ldb, stride_B, m, count);
    forward_solve_mat_batched(s, u.Ball, stride_B, m, ldb, u.img, stride_I, u.Q, stride_B, m, ldb, /*rhs_lower=*/true, count);
    launch_colvec_dot_batched(s, u.Q, ldb, stride_B, m, vz, m, count, xs);
    launch_gather_vec(s, yd, u.idx_pad, padded, xs, outv);  // mean = y - A^-1 v
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    if ((st = copy_out_vector(ctx, &u, outv, compact, mean, location)) != AGP_OK) return st;
    if ((st = status_from_flags(ctx)) != AGP_OK) return st;
    if (variance) {  // diag(R^T R): the columns of all slabs are ldb apart
      launch_coldot(s, u.Q, ldb, u.Q, ldb, m, padded, outv, -1.0, nullptr);
      if ((st = copy_out_vector(ctx, &u, outv, compact, variance, location)) != AGP_OK) return st;
    }
    if (joint) {
      AGP_HIP_CHECK(ctx, hipMemsetAsync(u.Ball, 0, sizeof(double) * (size_t)stride_B * (size_t)count, s));
      launch_gemm_nt_sub_batched(s, u.Ball, ldb, stride_B, u.Q, ldb, true, stride_B, u.Q, ldb, true, stride_B, m, m, m, false,
                                 count);
      launch_axpby(s, stride_B * count, -1.0, u.Ball, 0.0, nullptr, u.Ball);
      if ((st = copy_out_blocks(ctx, &u, u.Ball, ldb, joint, location)) != AGP_OK) return st;
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
