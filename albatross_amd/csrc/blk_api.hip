// blk_api.hip — block-level C-ABI entry points of the multi-GPU sharded fit
// (see include/albatross_amd.h and albatross_amd/distributed.py).  Thin host
// wrappers over the same kernels the single-GPU fit uses.
#include "common.h"

namespace agp {
int device_program_for(agp_context *ctx, const agp_kernel *k, const DevProgram **out);
int features_to_device(agp_context *ctx, const agp_features *f, bool copy, DeviceFeatures *out);
void panel_phase_public(agp_context *ctx, hipStream_t s, double *A, long long n, long long lda, double *img,
                        double *y, long long K0, long long kend);
void launch_back_update(hipStream_t s, const double *A, long long lda, long long k0, int nbk, long long ncols,
                        const double *x, double *z);
}  // namespace agp

using namespace agp;

extern "C" {

int agp_blk_gram(agp_context *ctx, const agp_kernel *k, const agp_features *rows, const agp_features *cols,
                 double *out, int64_t ld, const double *diag_add, int *nan_flag) {
  if (!ctx || !k || !rows || !cols || !out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (rows->dim != cols->dim || ld < rows->n) return AGP_ERR_INVALID_ARGUMENT;
  if (rows->n == 0 || cols->n == 0) return AGP_OK;
  const DevProgram *dprog = nullptr;
  int st = device_program_for(ctx, k, &dprog);
  if (st != AGP_OK) return st;
  DeviceFeatures dr, dc;
  if ((st = features_to_device(ctx, rows, false, &dr)) != AGP_OK) return st;
  if ((st = features_to_device(ctx, cols, false, &dc)) != AGP_OK) { dr.release(); return st; }
  hipError_t e = hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), ctx->stream);
  launch_gram(ctx->stream, dprog, dr.v, dc.v, /*symmetric=*/true, /*lower_only=*/true, out, ld, diag_add,
              ctx->d_flags, &k->prog);
  if (e == hipSuccess) e = hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, ctx->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
  dr.release();
  dc.release();
  if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return AGP_ERR_HIP; }
  if (nan_flag) *nan_flag = ctx->h_flags[0] ? 1 : 0;
  return AGP_OK;
}

int agp_blk_panel_factor(agp_context *ctx, double *A, int64_t m, int64_t lda, int64_t width, double *img,
                         double *y, int64_t *bad_pivot, double *log_sum) {
  if (!ctx || !A || !img || m <= 0 || width <= 0 || width > m || lda < m) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_scalars, 0, 4 * sizeof(double), s));
  panel_phase_public(ctx, s, A, m, lda, img, y, 0, width);
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  if (bad_pivot) *bad_pivot = ctx->h_flags[1] ? (int64_t)ctx->h_flags[1] - 1 : -1;
  if (log_sum) *log_sum = ctx->h_scalars[0];
  return AGP_OK;
}

int agp_blk_update(agp_context *ctx, double *C, int64_t ldc, const double *P, int64_t ldp, const double *Q,
                   int64_t ldq, int64_t M, int64_t N, int64_t K, int tri) {
  if (!ctx || !C || !P || !Q) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  launch_gemm_nt_sub(ctx->stream, C, ldc, P, ldp, false, Q, ldq, false, M, N, K, tri != 0);
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

int agp_blk_back_diag(agp_context *ctx, const double *A, int64_t lda, int64_t width, const double *img, double *z) {
  if (!ctx || !A || !img || !z || width <= 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long nblk = (width + NB - 1) / NB;
  const size_t need = sizeof(double) * ((size_t)nblk * NB * NB + (size_t)nblk * NB);
  if (ctx->ws_aux_bytes < need) {
    if (ctx->ws_aux) {
      AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
      AGP_HIP_CHECK(ctx, hipFree(ctx->ws_aux));
      ctx->ws_aux = nullptr;
      ctx->ws_aux_bytes = 0;
    }
    AGP_HIP_CHECK(ctx, hipMalloc(&ctx->ws_aux, need));
    ctx->ws_aux_bytes = need;
  }
  invert_diag_blocks(ctx->stream, A, width, lda, img, ctx->ws_aux);
  backward_solve_vec(ctx->stream, A, width, lda, ctx->ws_aux, z, ctx->ws_aux + (size_t)nblk * NB * NB);
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

int agp_blk_back_update(agp_context *ctx, const double *Arows, int64_t lda, int64_t nrows, int64_t ncols,
                        const double *x, double *z) {
  if (!ctx || !Arows || !x || !z || nrows < 0 || ncols < 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  for (long long r0 = 0; r0 < nrows; r0 += NB) {
    const int nbk = (int)((nrows - r0 < NB) ? nrows - r0 : NB);
    launch_back_update(ctx->stream, Arows, lda, r0, nbk, ncols, x + r0, z);
  }
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

}  // extern "C"
