// debug_api.hip — kernel-level entry points used only by tests/ to check the
// building blocks (MFMA lane map, update kernel, factorisation) in isolation.
// Not part of include/albatross_amd.h.
#include "common.h"
#include "mfma_f64.h"

namespace agp {

__global__ void mfma_tile_kernel(const double *A, const double *B, double *D) {
  const int l = threadIdx.x;
  // A is 16x4 row-major, B is 4x16 row-major, D 16x16 row-major
  v4d acc = v4zero();
  acc = mfma16(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

}  // namespace agp

using namespace agp;

extern "C" {

int agp_debug_mfma_tile(agp_context *ctx, const double *A, const double *B, double *D) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double *d = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d, sizeof(double) * (64 + 64 + 256)));
  AGP_HIP_CHECK(ctx, hipMemcpy(d, A, sizeof(double) * 64, hipMemcpyHostToDevice));
  AGP_HIP_CHECK(ctx, hipMemcpy(d + 64, B, sizeof(double) * 64, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(mfma_tile_kernel, dim3(1), dim3(64), 0, ctx->stream, d, d + 64, d + 128);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipMemcpy(D, d + 128, sizeof(double) * 256, hipMemcpyDeviceToHost));
  AGP_HIP_CHECK(ctx, hipFree(d));
  return AGP_OK;
}

// C (M x N, ldc) -= A * B^T on host arrays.
//   a_kmajor == 0: A is M x K column-major (lda >= M); else K x M column-major (lda >= K)
//   b_kmajor likewise with N.
int agp_debug_gemm(agp_context *ctx, double *C, int64_t ldc, const double *A, int64_t lda, int a_kmajor,
                   const double *B, int64_t ldb, int b_kmajor, int64_t M, int64_t N, int64_t K, int tri) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t cb = sizeof(double) * (size_t)ldc * (size_t)N;
  const size_t ab = sizeof(double) * (size_t)lda * (size_t)(a_kmajor ? M : K);
  const size_t bb = sizeof(double) * (size_t)ldb * (size_t)(b_kmajor ? N : K);
  double *dC = nullptr, *dA = nullptr, *dB = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dC, cb));
  AGP_HIP_CHECK(ctx, hipMalloc(&dA, ab));
  AGP_HIP_CHECK(ctx, hipMalloc(&dB, bb));
  AGP_HIP_CHECK(ctx, hipMemcpy(dC, C, cb, hipMemcpyHostToDevice));
  AGP_HIP_CHECK(ctx, hipMemcpy(dA, A, ab, hipMemcpyHostToDevice));
  AGP_HIP_CHECK(ctx, hipMemcpy(dB, B, bb, hipMemcpyHostToDevice));
  launch_gemm_nt_sub(ctx->stream, dC, ldc, dA, lda, a_kmajor != 0, dB, ldb, b_kmajor != 0, M, N, K, tri != 0);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  AGP_HIP_CHECK(ctx, hipMemcpy(C, dC, cb, hipMemcpyDeviceToHost));
  (void)hipFree(dC); (void)hipFree(dA); (void)hipFree(dB);
  return AGP_OK;
}

// In-place LL^T of the lower triangle of a host matrix; y (optional) -> L^-1 y.
int agp_debug_factor(agp_context *ctx, double *A, int64_t n, int64_t lda, double *y, double *log_det,
                     int64_t *bad_pivot) {
  if (!ctx || !A || n <= 0 || lda < n) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t ab = sizeof(double) * (size_t)lda * (size_t)n;
  const long long nblk = (n + NB - 1) / NB;
  double *dA = nullptr, *dI = nullptr, *dy = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dA, ab));
  AGP_HIP_CHECK(ctx, hipMalloc(&dI, sizeof(double) * (size_t)nblk * NMB * MB * MB));
  AGP_HIP_CHECK(ctx, hipMemcpy(dA, A, ab, hipMemcpyHostToDevice));
  if (y) {
    AGP_HIP_CHECK(ctx, hipMalloc(&dy, sizeof(double) * (size_t)n));
    AGP_HIP_CHECK(ctx, hipMemcpy(dy, y, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  }
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), ctx->stream));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_scalars, 0, 4 * sizeof(double), ctx->stream));
  factor_lower(ctx, dA, n, lda, dI, dy, nullptr);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  AGP_HIP_CHECK(ctx, hipMemcpy(A, dA, ab, hipMemcpyDeviceToHost));
  if (y) AGP_HIP_CHECK(ctx, hipMemcpy(y, dy, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  int flags[4];
  double scal[4];
  AGP_HIP_CHECK(ctx, hipMemcpy(flags, ctx->d_flags, sizeof(flags), hipMemcpyDeviceToHost));
  AGP_HIP_CHECK(ctx, hipMemcpy(scal, ctx->d_scalars, sizeof(scal), hipMemcpyDeviceToHost));
  if (log_det) *log_det = 2. * scal[0];
  if (bad_pivot) *bad_pivot = flags[1] ? flags[1] - 1 : -1;
  (void)hipFree(dA); (void)hipFree(dI);
  if (dy) (void)hipFree(dy);
  return AGP_OK;
}

}  // extern "C"
