// ldlt_api.hip — pivoted L D L^T entry points of the C-ABI (include/albatross_amd.h): the factorisation
// Eigen::SerializableLDLT performs in the reference, for matrices the un-pivoted LL^T path rejects.
#include <cmath>
#include <cstdlib>
#include <new>
#include <vector>

#include "api_internal.h"


using namespace agp;

extern "C" {

void agp_ldlt_destroy(agp_ldlt *f) {
  if (!f) return;
  if (f->ctx) (void)hipSetDevice(f->ctx->device);
  if (f->A) (void)hipFree(f->A);
  if (f->q_dev) (void)hipFree(f->q_dev);
  delete f;
}

int64_t agp_ldlt_size(const agp_ldlt *f) { return f ? f->n : 0; }

int agp_ldlt_create(agp_context *ctx, const double *K, int64_t n, int64_t ld, int uplo, int location, agp_ldlt **out,
                    int *success) {
  if (!ctx || !K || !out || n <= 0 || ld < n || (uplo != 0 && uplo != 1)) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  agp_ldlt *f = new (std::nothrow) agp_ldlt();
  if (!f) return AGP_ERR_INVALID_ARGUMENT;
  f->ctx = ctx; f->n = n; f->lda = round_up(n, 2);
#define LD_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      agp_ldlt_destroy(f);                                                               \
      return AGP_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
  LD_HIP(hipMalloc(&f->A, sizeof(double) * (size_t)f->lda * (size_t)n));
  LD_HIP(hipMalloc(&f->q_dev, sizeof(long long) * (size_t)n));
  // bring the triangle in (lower as given, or the transpose of the upper one)
  const double *src = K;
  if (location == AGP_HOST) {
    const size_t bytes = sizeof(double) * ((size_t)ld * (size_t)(n - 1) + (size_t)n);
    int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)ld * (size_t)n);
    if (st != AGP_OK) { agp_ldlt_destroy(f); return st; }
    LD_HIP(hipMemcpyAsync(ctx->ws_aux, K, bytes, hipMemcpyHostToDevice, s));
    src = ctx->ws_aux;
  }
  if (uplo == 0)
    LD_HIP(hipMemcpy2DAsync(f->A, sizeof(double) * (size_t)f->lda, src, sizeof(double) * (size_t)ld, sizeof(double) * (size_t)n,
                            (size_t)n, hipMemcpyDeviceToDevice, s));
  else
    launch_upper_to_lower(s, src, ld, f->A, f->lda, n);
  // the transposition sequence follows from the initial diagonal alone (see ldlt.hip)
  f->d.resize((size_t)n);
  LD_HIP(hipMemcpy2DAsync(f->d.data(), sizeof(double), f->A, sizeof(double) * (size_t)(f->lda + 1), sizeof(double), (size_t)n,
                          hipMemcpyDeviceToHost, s));
  LD_HIP(hipStreamSynchronize(s));
  f->tr.resize((size_t)n);
  {
    std::vector<double> d = f->d;
    for (long long k = 0; k < n; ++k) {
      long long big = k;
      double best = std::fabs(d[(size_t)k]);
      for (long long i = k + 1; i < n; ++i) {
        const double v = std::fabs(d[(size_t)i]);
        if (v > best) { best = v; big = i; }
      }
      f->tr[(size_t)k] = big;
      std::swap(d[(size_t)k], d[(size_t)big]);
    }
  }
  std::vector<long long> q((size_t)n);
  for (long long i = 0; i < n; ++i) q[(size_t)i] = i;
  for (long long k = 0; k < n; ++k) std::swap(q[(size_t)k], q[(size_t)f->tr[(size_t)k]]);
  LD_HIP(hipMemcpyAsync(f->q_dev, q.data(), sizeof(long long) * (size_t)n, hipMemcpyHostToDevice, s));
  LD_HIP(hipStreamSynchronize(s));
  // scratch: temp (n) | scal (2) | info (2 ints) | dotacc (n) | T panel (n x 32)
  double *scratch = nullptr;
  LD_HIP(hipMalloc(&scratch, sizeof(double) * (size_t)(n + 4 + n + 32 * n)));
  double *scal = scratch + n;
  int *info = reinterpret_cast<int *>(scal + 2);
  double *dotacc = scratch + n + 4, *Tpanel = dotacc + n;
  const int init_info[2] = {0, 1};
  hipError_t e = hipMemcpyAsync(info, init_info, sizeof(init_info), hipMemcpyHostToDevice, s);
  double *Ap = nullptr;
  if (e == hipSuccess && n >= 64) {
    // permute once (Ap = P A P^T), then the blocked factorisation: same arithmetic, ~n / 32 * 3 launches
    e = hipMalloc(&Ap, sizeof(double) * (size_t)f->lda * (size_t)n);
    if (e == hipSuccess) {
      launch_symmetrize(s, f->A, f->lda, n);
      ldlt_permute_sym(s, f->A, f->lda, f->q_dev, n, Ap, f->lda);
      ldlt_factor_blocked(s, Ap, f->lda, n, Tpanel, dotacc, info);
      e = hipStreamSynchronize(s);
    }
    if (e == hipSuccess) {
      (void)hipFree(f->A);
      f->A = Ap;
      Ap = nullptr;
    }
  } else if (e == hipSuccess) {
    ldlt_factor(s, f->A, f->lda, n, f->tr.data(), scratch, info, scal);
  }
  if (e == hipSuccess) {
    int h_info[2] = {0, 1};
    e = hipMemcpyAsync(h_info, info, sizeof(h_info), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess)
      e = hipMemcpy2DAsync(f->d.data(), sizeof(double), f->A, sizeof(double) * (size_t)(f->lda + 1), sizeof(double), (size_t)n,
                           hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    f->success = h_info[1];
  }
  if (Ap) (void)hipFree(Ap);
  (void)hipFree(scratch);
  if (e != hipSuccess) {
    ctx->last_error = hipGetErrorString(e);
    agp_ldlt_destroy(f);
    return AGP_ERR_HIP;
  }
#undef LD_HIP
  if (success) *success = f->success;
  *out = f;
  return AGP_OK;
}

int agp_ldlt_solve(agp_context *ctx, const agp_ldlt *f, const double *rhs, int64_t nrhs, double *out, int location) {
  if (!ctx || !f || !rhs || !out || nrhs < 0) return AGP_ERR_INVALID_ARGUMENT;
  if (nrhs == 0) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = f->n, ldw = round_up(n, 2);
  int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * 2 * (size_t)ldw * (size_t)nrhs);
  if (st != AGP_OK) return st;
  double *W = ctx->ws_aux, *R = W + (size_t)ldw * (size_t)nrhs;
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpy2DAsync(R, sizeof(double) * (size_t)ldw, rhs, sizeof(double) * (size_t)n,
                                      sizeof(double) * (size_t)n, (size_t)nrhs, kind, ctx->stream));
  if (location == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ldlt_solve(ctx->stream, f->A, f->lda, n, f->q_dev, W, R, ldw, nrhs);
  return copy_out_2d(ctx, R, ldw, n, nrhs, out, n, location);
}

int agp_ldlt_sqrt_solve(agp_context *ctx, const agp_ldlt *f, const double *rhs, int64_t nrhs, double *out, int location) {
  if (!ctx || !f || !rhs || !out || nrhs < 0) return AGP_ERR_INVALID_ARGUMENT;
  if (nrhs == 0) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = f->n, ldw = round_up(n, 2);
  int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * 2 * (size_t)ldw * (size_t)nrhs);
  if (st != AGP_OK) return st;
  double *W = ctx->ws_aux, *R = W + (size_t)ldw * (size_t)nrhs;
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpy2DAsync(R, sizeof(double) * (size_t)ldw, rhs, sizeof(double) * (size_t)n,
                                      sizeof(double) * (size_t)n, (size_t)nrhs, kind, ctx->stream));
  if (location == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  ldlt_sqrt_solve(ctx->stream, f->A, f->lda, n, f->q_dev, W, R, ldw, nrhs);
  return copy_out_2d(ctx, W, ldw, n, nrhs, out, n, location);
}

int agp_ldlt_vector_d(const agp_ldlt *f, double *d) {
  if (!f || !d) return AGP_ERR_INVALID_ARGUMENT;
  for (long long i = 0; i < f->n; ++i) d[i] = f->d[(size_t)i];
  return AGP_OK;
}

int agp_ldlt_transpositions(const agp_ldlt *f, int64_t *tr) {
  if (!f || !tr) return AGP_ERR_INVALID_ARGUMENT;
  for (long long i = 0; i < f->n; ++i) tr[i] = f->tr[(size_t)i];
  return AGP_OK;
}

int agp_ldlt_download(agp_context *ctx, const agp_ldlt *f, double *packed, int64_t ld) {
  if (!ctx || !f || !packed || ld < f->n) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return copy_out_2d(ctx, f->A, f->lda, f->n, f->n, packed, ld, AGP_HOST);
}

}  // extern "C"
