// debug_api.hip — kernel-level entry points used only by tests/ and scripts/ to check and time the
// building blocks (MFMA lane map, update kernel, factorisation) in isolation.
// Not part of include/albatross_amd.h and NOT linked into libalbatross_amd.so: they live in
// libalbatross_amd_debug.so (the product objects + this file; csrc/Makefile), loaded by
// albatross_amd._capi.load_debug().
#include <algorithm>
#include <cstdio>
#include "common.h"
#include "api_internal.h"
#include "mfma_f64.h"
#include "shard_internal.h"
#include <cstdlib>
#include <cstring>
#include <new>

#define AGP_DEBUG_API __attribute__((visibility("default")))

namespace agp {

__global__ void mfma_tile_kernel(const double *A, const double *B, double *D) {
  const int l = threadIdx.x;
  // A is 16x4 row-major, B is 4x16 row-major, D 16x16 row-major
  v4d acc = v4zero();
  acc = mfma16(A[(l & 15) * 4 + (l >> 4)], B[(l >> 4) * 16 + (l & 15)], acc);
  for (int r = 0; r < 4; ++r) D[((l >> 4) + 4 * r) * 16 + (l & 15)] = acc[r];
}

// cycles (s_memtime) and 100 MHz ticks (s_memrealtime) around a loop of
// `iters` x NACC independent v_mfma_f64_16x16x4_f64 per wave
template <int NACC>
__global__ __launch_bounds__(256) void mfma_clock_kernel(unsigned long long *out, int iters, double a0, double b0) {
  v4d acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = v4zero();
  const double a = a0 * (1.0 + (threadIdx.x % 7) * 0.125), b = b0 * (1.0 - (threadIdx.x % 5) * 0.0625);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = mfma16(a, b, acc[i]);
  }
  double s = 0.;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3];
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (s == 1.2345e300) out[0] = 0;
}

// NM MFMA + NV independent v_fma_f64 per loop iteration, per wave
template <int NM, int NV>
__global__ __launch_bounds__(256) void mix_clock_kernel(unsigned long long *out, int iters, double a0, double b0) {
  v4d acc[NM > 0 ? NM : 1];
  double v[NV > 0 ? NV : 1];
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) acc[i] = v4zero();
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) v[i] = 0.001 * (i + 1) + threadIdx.x * 1e-6;
  const double a = a0 * (1.0 + (threadIdx.x % 7) * 0.125), b = b0 * (1.0 - (threadIdx.x % 5) * 0.0625);
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NM; ++i) acc[i] = mfma16(a, b, acc[i]);
#pragma unroll
    for (int i = 0; i < NV; ++i) v[i] = __builtin_fma(v[i], 0.999999, b);
  }
  double s = 0.;
#pragma unroll
  for (int i = 0; i < (NM > 0 ? NM : 1); ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
  for (int i = 0; i < (NV > 0 ? NV : 1); ++i) s += v[i];
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (s == 1.2345e300) out[0] = 0;
}

// 64 independent v_fmac_f64 per loop iteration: MODE 0 plain VGPR operands, 1 SGPR src0,
// 2 DPP row_newbcast src0.  out[0..63] (MODE 2 semantics probe): acc after ONE fmac with b = lane id, a = 1.
template <int MODE>
__global__ __launch_bounds__(256) void fmac_rate_kernel(unsigned long long *out, double *probe, int iters, double b0) {
  double acc[64];
#pragma unroll
  for (int i = 0; i < 64; ++i) acc[i] = 0.;
  double a[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) a[r] = 1.0 + 0.001 * r + threadIdx.x * 1e-6;
  double b = b0 * (1.0 + (threadIdx.x & 63) * 0.01);
  const double bs = __builtin_amdgcn_readfirstlane((int)(b0 * 1000.0)) * 0.001;
  if (probe && MODE == 2) {
    double pa = 1.0, pb = (double)(threadIdx.x & 63), pacc = 0.;
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(pacc) : "v"(pb), "v"(pa));
    if (blockIdx.x == 0 && threadIdx.x < 64) probe[threadIdx.x] = pacc;
  }
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 64; ++i) {
      if (MODE == 0) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "v"(b), "v"(a[i & 3]));
      else if (MODE == 1) asm volatile("v_fmac_f64_e32 %0, %1, %2" : "+v"(acc[i]) : "s"(bs), "v"(a[i & 3]));
      else asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:7 row_mask:0xf bank_mask:0xf" : "+v"(acc[i]) : "v"(b), "v"(a[i & 3]));
    }
  }
  const unsigned long long c1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0.;
#pragma unroll
  for (int i = 0; i < 64; ++i) s += acc[i];
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = c1 - c0;
    out[2 * blockIdx.x + 1] = r1 - r0;
  }
  if (s == 1.2345e300) out[0] = 0;
}

}  // namespace agp

using namespace agp;

// Bare issue loops of the two fp64 MFMA shapes: mode 0 = v_mfma_f64_16x16x4_f64 with NACC independent
// accumulators, mode 1 = v_mfma_f64_4x4x4_4b_f64 (four 4 x 4 x 4 blocks, 512 flop) with NACC accumulators.
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void mfma_shape_kernel(double *sink, int iters, double a0, double b0) {
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  double s = 0.;
  if (MODE == 0) {
    agp::v4d acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = agp::v4zero();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  } else if (MODE == 1) {
    double acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc[i], 0, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
  } else {
    // mode 2: as mode 1 but with 16 different A and 4 different B operand registers (the GEMM pattern)
    double acc[NACC], av[16], bv[4];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.;
    // a0 < 0 selects operands with random mantissas in [1, 2) x (+-1): data-dependent power
    unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    auto rnd = [&]() {
      h ^= h << 13; h ^= h >> 17; h ^= h << 5;
      const unsigned lo = h;
      h ^= h << 13; h ^= h >> 17; h ^= h << 5;
      const unsigned hi = (h & 0x800fffffu) | 0x3ff00000u;
      return __hiloint2double((int)hi, (int)lo);
    };
#pragma unroll
    for (int i = 0; i < 16; ++i) av[i] = a0 < 0. ? rnd() * 0.01 : a + i * 1e-3;
#pragma unroll
    for (int i = 0; i < 4; ++i) bv[i] = a0 < 0. ? rnd() * 0.01 : b - i * 1e-3;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(av[i & 15], bv[(i >> 4) & 3], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("" : "+v"(av[i]));
    }
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i];
  }
  if (s == 123.456) sink[0] = s;
}


// Lane-map probe of v_mfma_f64_4x4x4_4b_f64: for every pair (la, lb) the operand A is one-hot in lane la and
// B one-hot in lane lb; out[(la * 64 + lb)] = mask of result lanes that are non-zero.
template <int CBSZ, int ABID>
__global__ __launch_bounds__(64) void mfma44_probe_kernel(unsigned long long *out) {
  const int lane = threadIdx.x;
  for (int la = 0; la < 64; ++la)
    for (int lb = 0; lb < 64; ++lb) {
      const double a = lane == la ? 1. : 0., b = lane == lb ? 1. : 0.;
      const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0., CBSZ, ABID, 0);
      const unsigned long long m = __ballot(d != 0.);
      if (lane == 0) out[la * 64 + lb] = m;
    }
}

// exp_neg (cov_eval.h) on an array: accuracy test against the correctly rounded exp
__global__ void exp_neg_kernel(const double *t, double *out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = agp::exp_neg(t[i]);
}
extern "C" AGP_DEBUG_API int agp_debug_exp_neg(agp_context *ctx, const double *t, int64_t n, double *out) {
  if (!ctx || !t || !out || n <= 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double *d = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d, sizeof(double) * 2 * (size_t)n));
  AGP_HIP_CHECK(ctx, hipMemcpy(d, t, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(exp_neg_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, d + n, (long long)n);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipMemcpy(out, d + n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return AGP_OK;
}

// {sum of shader-clock cycles, sum of 100 MHz ticks, workgroups} of the trailing_update_kernel launches since the last
// reset (a library built with -DAGP_CLOCK_PROBE; zeros otherwise)
extern "C" AGP_DEBUG_API int agp_debug_symv_lower(agp_context *ctx, const double *K, int64_t n, int64_t ld, const double *p,
                                                  double alpha, double beta, const double *base, double *out) {
  if (!ctx || !K || !p || !out || n <= 0 || ld < n) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double *dK = nullptr, *dv = nullptr, *ws = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dK, sizeof(double) * (size_t)ld * (size_t)n));
  AGP_HIP_CHECK(ctx, hipMalloc(&dv, sizeof(double) * 3 * (size_t)n));
  AGP_HIP_CHECK(ctx, hipMalloc(&ws, sizeof(double) * symv_ws_elems(n)));
  AGP_HIP_CHECK(ctx, hipMemcpy(dK, K, sizeof(double) * (size_t)ld * (size_t)n, hipMemcpyHostToDevice));
  AGP_HIP_CHECK(ctx, hipMemcpy(dv, p, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  if (base) AGP_HIP_CHECK(ctx, hipMemcpy(dv + n, base, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  launch_symv_lower(ctx->stream, dK, ld, n, dv, alpha, beta, base ? dv + n : nullptr, dv + 2 * n, ws);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipMemcpy(out, dv + 2 * n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  (void)hipFree(dK); (void)hipFree(dv); (void)hipFree(ws);
  return AGP_OK;
}

// acos_fast (cov_eval.h) on an array: accuracy test against the correctly rounded acos
__global__ void acos_fast_kernel(const double *t, double *out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) out[i] = agp::acos_fast(t[i]);
}
extern "C" AGP_DEBUG_API int agp_debug_acos_fast(agp_context *ctx, const double *t, int64_t n, double *out) {
  if (!ctx || !t || !out || n <= 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double *d = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d, sizeof(double) * 2 * (size_t)n));
  AGP_HIP_CHECK(ctx, hipMemcpy(d, t, sizeof(double) * (size_t)n, hipMemcpyHostToDevice));
  hipLaunchKernelGGL(acos_fast_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream, d, d + n, (long long)n);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipMemcpy(out, d + n, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  (void)hipFree(d);
  return AGP_OK;
}

// ---- launch-chain latency probe ------------------------------------------------------------------------------
// Synthetic kernels with a chosen static LDS footprint: every workgroup writes a little, then spins `spin` shader
// clocks.  Used to find out what a dependent launch costs on this part when consecutive kernels of one stream
// differ (code, LDS size, grid): the panel chain of the factorisation is such a sequence of tiny launches.
template <int LDS_DOUBLES>
__global__ __launch_bounds__(256) void chain_probe_kernel(double *buf, long long stride, int spin, int touch) {
  __shared__ double lds[LDS_DOUBLES > 0 ? LDS_DOUBLES : 1];
  if (LDS_DOUBLES > 0) lds[threadIdx.x % LDS_DOUBLES] = threadIdx.x;
  __syncthreads();
  double v = (LDS_DOUBLES > 0) ? lds[(threadIdx.x + 1) % (LDS_DOUBLES > 0 ? LDS_DOUBLES : 1)] : 1.0;
  double *p = buf + (long long)blockIdx.x * stride;
  for (int t = 0; t < touch; ++t) p[threadIdx.x + 256 * t] = p[threadIdx.x + 256 * t] * 0.5 + v;
  const unsigned long long c0 = __builtin_amdgcn_s_memtime();
  while ((long long)(__builtin_amdgcn_s_memtime() - c0) < spin) {}
}

static void launch_probe(hipStream_t s, int kind, int wgs, double *buf, long long stride, int spin, int touch) {
  switch (kind) {
  case 0: hipLaunchKernelGGL(chain_probe_kernel<0>, dim3(wgs), dim3(256), 0, s, buf, stride, spin, touch); break;
  case 1: hipLaunchKernelGGL(chain_probe_kernel<5120>, dim3(wgs), dim3(256), 0, s, buf, stride, spin, touch); break;   // 40 KB
  case 2: hipLaunchKernelGGL(chain_probe_kernel<9344>, dim3(wgs), dim3(256), 0, s, buf, stride, spin, touch); break;   // 73 KB
  default: hipLaunchKernelGGL(chain_probe_kernel<9856>, dim3(wgs), dim3(256), 0, s, buf, stride, spin, touch); break;  // 77 KB
  }
}

extern "C" AGP_DEBUG_API int agp_debug_chain_probe(agp_context *ctx, const int *kinds, const int *wgs, int period, int reps, int spin,
                                     int touch, int priority_stream, double *us_per_launch) {
  if (!ctx || !kinds || !wgs || period <= 0 || reps <= 0 || !us_per_launch) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = priority_stream ? ctx->stream : ctx->stream2;
  const long long stride = 256LL * (touch > 0 ? touch : 1);
  double *buf = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&buf, sizeof(double) * (size_t)stride * 4096));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(buf, 0, sizeof(double) * (size_t)stride * 4096, s));
  hipEvent_t e0, e1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  for (int i = 0; i < period * 3; ++i) launch_probe(s, kinds[i % period], wgs[i % period], buf, stride, spin, touch);
  AGP_HIP_CHECK(ctx, hipEventRecord(e0, s));
  for (int i = 0; i < period * reps; ++i) launch_probe(s, kinds[i % period], wgs[i % period], buf, stride, spin, touch);
  AGP_HIP_CHECK(ctx, hipEventRecord(e1, s));
  AGP_HIP_CHECK(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_launch = 1e3 * ms / (double)(period * reps);
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(buf);
  return AGP_OK;
}

// the REAL chain: the panel phase (POTRF, TRSM, inner update per 128 columns) of an n x n block, `reps` times
namespace agp {
void panel_phase_public(agp_context *ctx, hipStream_t s, double *A, long long n, long long lda, double *img, double *y,
                        long long K0, long long kend);
}
// blocked: while the chain runs, `blocked` other streams sit at a hipStreamWaitEvent on an event that is recorded
// behind a long one-workgroup spinner on yet another stream (a queue whose head is an unsatisfied barrier packet)
extern "C" AGP_DEBUG_API int agp_debug_panel_chain(agp_context *ctx, int64_t n, int64_t width, int reps, int blocked, double *us_per_phase) {
  if (!ctx || n <= 0 || width <= 0 || width > n || reps <= 0 || !us_per_phase) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t s = ctx->stream;
  hipStream_t extra[3] = {nullptr, nullptr, nullptr};
  hipEvent_t gate = nullptr;
  double *spinbuf = nullptr;
  // blocked = -1: only the spinner (a long one-workgroup kernel on another stream), nobody waits on it
  // blocked = -2: a stream blocked in hipStreamWaitValue32 on host memory, NO kernel running elsewhere
  unsigned int *flag = nullptr;
  if (blocked == -2) {
    AGP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&extra[1], hipStreamNonBlocking));
    AGP_HIP_CHECK(ctx, hipExtMallocWithFlags((void **)&flag, 64, hipMallocSignalMemory));
    *flag = 0;
    AGP_HIP_CHECK(ctx, hipMalloc(&spinbuf, sizeof(double) * 4096));
    AGP_HIP_CHECK(ctx, hipStreamWaitValue32(extra[1], flag, 1, hipStreamWaitValueEq, 0xffffffffu));
    hipLaunchKernelGGL(chain_probe_kernel<0>, dim3(1), dim3(256), 0, extra[1], spinbuf, 256, 100, 1);
  }
  if (blocked == -1) {
    AGP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&extra[0], hipStreamNonBlocking));
    AGP_HIP_CHECK(ctx, hipMalloc(&spinbuf, sizeof(double) * 4096));
    hipLaunchKernelGGL(chain_probe_kernel<0>, dim3(1), dim3(256), 0, extra[0], spinbuf, 256, 144000000, 1);
  }
  if (blocked > 0) {
    if (blocked > 2) blocked = 2;
    for (int i = 0; i <= blocked; ++i) AGP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&extra[i], hipStreamNonBlocking));
    AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&gate, hipEventDisableTiming));
    AGP_HIP_CHECK(ctx, hipMalloc(&spinbuf, sizeof(double) * 4096));
    // ~60 ms spinner (2.4 GHz shader clock) on extra[0]; the gate event completes when it ends
    hipLaunchKernelGGL(chain_probe_kernel<0>, dim3(1), dim3(256), 0, extra[0], spinbuf, 256, 144000000, 1);
    AGP_HIP_CHECK(ctx, hipEventRecord(gate, extra[0]));
    for (int i = 1; i <= blocked; ++i) {
      AGP_HIP_CHECK(ctx, hipStreamWaitEvent(extra[i], gate, 0));
      hipLaunchKernelGGL(chain_probe_kernel<0>, dim3(1), dim3(256), 0, extra[i], spinbuf + 1024 * i, 256, 100, 1);
    }
  }
  const long long lda = n + 8;
  double *A = nullptr, *img = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&A, sizeof(double) * (size_t)lda * (size_t)width));
  AGP_HIP_CHECK(ctx, hipMalloc(&img, sizeof(double) * (size_t)((width + NB - 1) / NB) * 36 * 256));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(A, 0, sizeof(double) * (size_t)lda * (size_t)width, s));  // zeros: NaN results, same timing
  hipEvent_t e0, e1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  agp::panel_phase_public(ctx, s, A, n, lda, img, nullptr, 0, width);
  AGP_HIP_CHECK(ctx, hipEventRecord(e0, s));
  for (int r = 0; r < reps; ++r) agp::panel_phase_public(ctx, s, A, n, lda, img, nullptr, 0, width);
  AGP_HIP_CHECK(ctx, hipEventRecord(e1, s));
  AGP_HIP_CHECK(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  *us_per_phase = 1e3 * ms / reps;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(A);
  (void)hipFree(img);
  if (flag) *flag = 1;  // release the stream blocked on host memory
  if (blocked != 0) {
    (void)hipDeviceSynchronize();
    for (auto st : extra) if (st) (void)hipStreamDestroy(st);
    if (gate) (void)hipEventDestroy(gate);
    (void)hipFree(spinbuf);
    if (flag) (void)hipFree(flag);
  }
  return AGP_OK;
}

extern "C" {

AGP_DEBUG_API int agp_debug_mfma_tile(agp_context *ctx, const double *A, const double *B, double *D) {
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

// out[0] = median cycles per MFMA per wave, out[1] = effective clock (GHz),
// out[2] = chip TFLOP/s.  waves_per_simd in {1, 2}; nacc in {1, 4, 8}.
AGP_DEBUG_API int agp_debug_mfma_clock(agp_context *ctx, int blocks, int waves_per_simd, int nacc, int iters, double a0,
                         double b0, double *out) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int nblk = blocks * waves_per_simd;
  unsigned long long *d = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d, sizeof(unsigned long long) * 2 * nblk));
  hipEvent_t e0, e1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    AGP_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
    if (nacc == 1) hipLaunchKernelGGL(mfma_clock_kernel<1>, dim3(nblk), dim3(256), 0, ctx->stream, d, iters, a0, b0);
    else if (nacc == 4) hipLaunchKernelGGL(mfma_clock_kernel<4>, dim3(nblk), dim3(256), 0, ctx->stream, d, iters, a0, b0);
    else hipLaunchKernelGGL(mfma_clock_kernel<8>, dim3(nblk), dim3(256), 0, ctx->stream, d, iters, a0, b0);
    AGP_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
    AGP_HIP_CHECK(ctx, hipEventSynchronize(e1));
  }
  float ms = 0.f;
  AGP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * nblk);
  AGP_HIP_CHECK(ctx, hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * nblk, hipMemcpyDeviceToHost));
  std::vector<double> cyc(nblk), clk(nblk);
  const int na = nacc == 1 ? 1 : (nacc == 4 ? 4 : 8);
  for (int i = 0; i < nblk; ++i) {
    cyc[i] = (double)h[2 * i] / ((double)iters * na);
    clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0) ;  // cycles per 10 ns tick -> GHz
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  out[0] = cyc[nblk / 2];
  out[1] = clk[nblk / 2];
  out[2] = (double)nblk * 4.0 * iters * na * 2048.0 / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
  return AGP_OK;
}

// out[0] = cycles per loop iteration per wave, out[1] = clock GHz, out[2] = chip TFLOP/s (MFMA + VALU flops)
AGP_DEBUG_API int agp_debug_mix_clock(agp_context *ctx, int waves_per_simd, int variant, int iters, double *out) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int nblk = 256 * waves_per_simd;
  unsigned long long *d = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d, sizeof(unsigned long long) * 2 * nblk));
  hipEvent_t e0, e1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  int nm = 0, nv = 0;
  for (int rep = 0; rep < 2; ++rep) {
    AGP_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
#define MIX(NM_, NV_) { nm = NM_; nv = NV_; hipLaunchKernelGGL((mix_clock_kernel<NM_, NV_>), dim3(nblk), dim3(256), 0, ctx->stream, d, iters, 1.1, 0.9); }
    switch (variant) {
    case 0: MIX(0, 16) break;
    case 1: MIX(0, 32) break;
    case 2: MIX(4, 0) break;
    case 3: MIX(4, 16) break;
    case 4: MIX(4, 32) break;
    case 5: MIX(4, 64) break;
    case 6: MIX(4, 96) break;
    case 7: MIX(2, 48) break;
    default: MIX(4, 48) break;
    }
#undef MIX
    AGP_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
    AGP_HIP_CHECK(ctx, hipEventSynchronize(e1));
  }
  float ms = 0.f;
  AGP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * nblk);
  AGP_HIP_CHECK(ctx, hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * nblk, hipMemcpyDeviceToHost));
  std::vector<double> cyc(nblk), clk(nblk);
  for (int i = 0; i < nblk; ++i) {
    cyc[i] = (double)h[2 * i] / (double)iters;
    clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  out[0] = cyc[nblk / 2];
  out[1] = clk[nblk / 2];
  out[2] = (double)nblk * 4.0 * iters * (nm * 2048.0 + nv * 128.0) / (ms * 1e-3) / 1e12;
  out[3] = nm; out[4] = nv;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d);
  return AGP_OK;
}

// C (M x N, ldc) -= A * B^T on host arrays.
//   a_kmajor == 0: A is M x K column-major (lda >= M); else K x M column-major (lda >= K)
//   b_kmajor likewise with N.
AGP_DEBUG_API int agp_debug_gemm(agp_context *ctx, double *C, int64_t ldc, const double *A, int64_t lda, int a_kmajor,
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

// v_fmac_f64 issue rate with VGPR (mode 0), SGPR (1) or DPP row_newbcast (2) src0.
// out[0] = cycles per 64-FMA iteration per wave, out[1] = clock GHz, out[2] = chip TFLOP/s;
// probe[64] (mode 2): result of one fmac with b = lane id, a = 1, row_newbcast:5.
AGP_DEBUG_API int agp_debug_fmac_rate(agp_context *ctx, int waves_per_simd, int mode, int iters, double *out, double *probe) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const int nblk = 256 * waves_per_simd;
  unsigned long long *d = nullptr;
  double *dp = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d, sizeof(unsigned long long) * 2 * nblk));
  AGP_HIP_CHECK(ctx, hipMalloc(&dp, sizeof(double) * 64));
  hipEvent_t e0, e1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    AGP_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
    if (mode == 0) hipLaunchKernelGGL(fmac_rate_kernel<0>, dim3(nblk), dim3(256), 0, ctx->stream, d, dp, iters, 1.1);
    else if (mode == 1) hipLaunchKernelGGL(fmac_rate_kernel<1>, dim3(nblk), dim3(256), 0, ctx->stream, d, dp, iters, 1.1);
    else hipLaunchKernelGGL(fmac_rate_kernel<2>, dim3(nblk), dim3(256), 0, ctx->stream, d, dp, iters, 1.1);
    AGP_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
    AGP_HIP_CHECK(ctx, hipEventSynchronize(e1));
  }
  float ms = 0.f;
  AGP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(2 * nblk);
  AGP_HIP_CHECK(ctx, hipMemcpy(h.data(), d, sizeof(unsigned long long) * 2 * nblk, hipMemcpyDeviceToHost));
  if (probe) AGP_HIP_CHECK(ctx, hipMemcpy(probe, dp, sizeof(double) * 64, hipMemcpyDeviceToHost));
  std::vector<double> cyc(nblk), clk(nblk);
  for (int i = 0; i < nblk; ++i) {
    cyc[i] = (double)h[2 * i] / (double)iters;
    clk[i] = (double)h[2 * i] / ((double)h[2 * i + 1] * 10.0);
  }
  std::sort(cyc.begin(), cyc.end());
  std::sort(clk.begin(), clk.end());
  out[0] = cyc[nblk / 2];
  out[1] = clk[nblk / 2];
  out[2] = (double)nblk * 4.0 * iters * 64.0 * 128.0 / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(d); (void)hipFree(dp);
  return AGP_OK;
}

// mode 0: cbsz = 0; mode 1..4: cbsz = 2, abid = mode - 1.  out: 4096 masks.
AGP_DEBUG_API int agp_debug_mfma44_probe(agp_context *ctx, int mode, unsigned long long *out) {
  if (!ctx || !out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  unsigned long long *d = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d, sizeof(unsigned long long) * 4096));
  hipStream_t s = ctx->stream;
  switch (mode) {
    case 0: hipLaunchKernelGGL((mfma44_probe_kernel<0, 0>), dim3(1), dim3(64), 0, s, d); break;
    case 1: hipLaunchKernelGGL((mfma44_probe_kernel<2, 0>), dim3(1), dim3(64), 0, s, d); break;
    case 2: hipLaunchKernelGGL((mfma44_probe_kernel<2, 1>), dim3(1), dim3(64), 0, s, d); break;
    case 3: hipLaunchKernelGGL((mfma44_probe_kernel<2, 2>), dim3(1), dim3(64), 0, s, d); break;
    default: hipLaunchKernelGGL((mfma44_probe_kernel<2, 3>), dim3(1), dim3(64), 0, s, d); break;
  }
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(out, d, sizeof(unsigned long long) * 4096, hipMemcpyDeviceToHost, s));
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  (void)hipFree(d);
  return AGP_OK;
}


// out[0] = chip TFLOP/s, out[1] = ms
AGP_DEBUG_API int agp_debug_mfma_shape(agp_context *ctx, int mode, int nacc, int waves_per_simd, int iters, double *out) {
  const double a0 = iters < 0 ? -1.0 : 1.0;  // negative iteration count: random operands
  if (iters < 0) iters = -iters;
  if (!ctx || !out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double *sink = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&sink, 8));
  hipEvent_t e0, e1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  const int blocks = 256 * waves_per_simd;
  hipStream_t s = ctx->stream;
  for (int rep = 0; rep < 2; ++rep) {
    AGP_HIP_CHECK(ctx, hipEventRecord(e0, s));
#define SHAPE_LAUNCH(M_, N_) hipLaunchKernelGGL((mfma_shape_kernel<M_, N_>), dim3(blocks), dim3(256), 0, s, sink, iters, a0, 2.0)
    if (mode == 0 && nacc == 8) SHAPE_LAUNCH(0, 8);
    else if (mode == 0 && nacc == 16) SHAPE_LAUNCH(0, 16);
    else if (mode == 1 && nacc == 8) SHAPE_LAUNCH(1, 8);
    else if (mode == 1 && nacc == 16) SHAPE_LAUNCH(1, 16);
    else if (mode == 1 && nacc == 32) SHAPE_LAUNCH(1, 32);
    else if (mode == 1 && nacc == 64) SHAPE_LAUNCH(1, 64);
    else if (mode == 2 && nacc == 64) SHAPE_LAUNCH(2, 64);
    else if (mode == 2 && nacc == 32) SHAPE_LAUNCH(2, 32);
    else return AGP_ERR_INVALID_ARGUMENT;
#undef SHAPE_LAUNCH
    AGP_HIP_CHECK(ctx, hipEventRecord(e1, s));
    AGP_HIP_CHECK(ctx, hipEventSynchronize(e1));
  }
  float ms = 0.f;
  AGP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  const double flop_per_mfma = mode == 0 ? 2.0 * 16 * 16 * 4 : 2.0 * 4 * 4 * 4 * 4;
  out[0] = (double)blocks * 4.0 * (double)iters * nacc * flop_per_mfma / (ms * 1e-3) / 1e12;
  out[1] = ms;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(sink);
  return AGP_OK;
}

// fp16 x 2 planes of a host panel (gemm_f16x2.hip) with the row scales of a matrix whose diagonal is the squared row norm of
// the panel - the bound a Cholesky factor's rows obey; *scales_out = [r | 1 / r] (M doubles each), *planes_out the planes
static int make_f16x2_planes(agp_context *ctx, const double *hP, long long ldp, const double *dP, long long M, long long K,
                             double **scales_out, unsigned short **planes_out) {
  std::vector<double> diag((size_t)M, 0.);
  for (long long k = 0; k < K; ++k)
    for (long long i = 0; i < M; ++i) diag[(size_t)i] += hP[i + k * ldp] * hP[i + k * ldp];
  double *d_diag = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&d_diag, sizeof(double) * (size_t)M));
  AGP_HIP_CHECK(ctx, hipMalloc(scales_out, sizeof(double) * 2 * (size_t)M));
  AGP_HIP_CHECK(ctx, hipMalloc(planes_out, f16x2_bytes(M, K)));
  AGP_HIP_CHECK(ctx, hipMemcpy(d_diag, diag.data(), sizeof(double) * (size_t)M, hipMemcpyHostToDevice));
  launch_f16x2_row_scales(ctx->stream, d_diag, 0, M, *scales_out, *scales_out + M);
  launch_convert_panel_f16x2(ctx->stream, dP, ldp, M, K, *scales_out, *planes_out);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  (void)hipFree(d_diag);
  return AGP_OK;
}

// Bulk trailing update C (M x M, lower tiles) -= P P^T on host data.  variant 0: MFMA kernel,
// 2: DPP-broadcast VALU kernel.
AGP_DEBUG_API int agp_debug_trailing_update(agp_context *ctx, double *C, int64_t ldc, const double *P, int64_t ldp, int64_t M,
                              int64_t K, int variant) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t cb = sizeof(double) * (size_t)ldc * (size_t)M;
  const size_t pb = sizeof(double) * (size_t)ldp * (size_t)K;
  double *dC = nullptr, *dP = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dC, cb));
  AGP_HIP_CHECK(ctx, hipMalloc(&dP, pb));
  AGP_HIP_CHECK(ctx, hipMemcpy(dC, C, cb, hipMemcpyHostToDevice));
  AGP_HIP_CHECK(ctx, hipMemcpy(dP, P, pb, hipMemcpyHostToDevice));
  int st = AGP_OK;
  float *dP32 = nullptr;
  if (variant == 13) {  // the fp32-product kernel (3) reading an fp32 copy of the panel, as the mixed fit runs it
    const long long ld32 = (M + 7) / 8 * 8 + 8;
    AGP_HIP_CHECK(ctx, hipMalloc(&dP32, sizeof(float) * (size_t)ld32 * (size_t)K));
    launch_convert_panel_f32(ctx->stream, dP, ldp, M, K, dP32, ld32);
    launch_trailing_update_as(3, ctx->stream, dC, ldc, dP, dP, ldp, M, K, nullptr, dP32, dP32, ld32);
  } else if (variant == 14) {  // the bf16 x 3 kernel on the three bf16 planes of the panel, as the mixed fit runs it (gemm_bf16x3.hip)
    unsigned short *planes = nullptr;
    AGP_HIP_CHECK(ctx, hipMalloc(&planes, bf16x3_bytes(M, K)));
    launch_convert_panel_bf16x3(ctx->stream, dP, ldp, M, K, planes);
    const int ntr = (int)((M + 127) / 128);
    const long long tiles = (long long)ntr * (ntr + 1) / 2;
    long long olen = 0;
    const int *order = tiles >= 1024 ? bulk_tile_order(ntr, tiles, &olen) : nullptr;
    launch_update_bf16x3(ctx->stream, dC, ldc, planes, M, 0, 0, M, M, K, order, olen);
    AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(planes);
  } else if (variant == 15) {  // the fp16 x 2 kernel on the two fp16 planes of the row-scaled panel, as the mixed fit runs it (gemm_f16x2.hip)
    unsigned short *planes = nullptr;
    double *scales = nullptr;
    const int stp = make_f16x2_planes(ctx, P, ldp, dP, M, K, &scales, &planes);
    if (stp != AGP_OK) return stp;
    const int ntr = (int)((M + 127) / 128);
    const long long tiles = (long long)ntr * (ntr + 1) / 2;
    long long olen = 0;
    const int *order = tiles >= 1024 ? bulk_tile_order(ntr, tiles, &olen) : nullptr;
    launch_update_f16x2(ctx->stream, dC, ldc, planes, M, 0, 0, scales + M, M, M, K, order, olen);
    AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    (void)hipFree(planes); (void)hipFree(scales);
  } else if (variant >= 20 && variant < 30) {
    // the MERGED update of factor_lower with head_cols = variant - 20: the launch on the bulk stream, the gate kernel on
    // the chain stream (it must end - the head counted itself completely - although the launch it waits for is on another
    // stream), and the count it left must be exactly the number of head tiles
    unsigned long long *cnt = ctx->d_headcnt + 7, seen = ~0ull;
    AGP_HIP_CHECK(ctx, hipMemset(cnt, 0, sizeof(unsigned long long)));
    AGP_HIP_CHECK(ctx, hipMemset(ctx->d_flags, 0, 4 * sizeof(int)));
    long long head_tiles = 0;
    launch_trailing_update_merged(ctx->stream2, dC, ldc, dP, ldp, M, K, variant - 20, cnt, &head_tiles);
    launch_head_gate(ctx->stream, cnt, (unsigned long long)head_tiles, ctx->d_flags);
    AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream2));
    int flags[4] = {0, 0, 0, 0};
    AGP_HIP_CHECK(ctx, hipMemcpy(&seen, cnt, sizeof(seen), hipMemcpyDeviceToHost));
    AGP_HIP_CHECK(ctx, hipMemcpy(flags, ctx->d_flags, sizeof(flags), hipMemcpyDeviceToHost));
    if (seen != (unsigned long long)head_tiles || flags[2] != 0 || head_tiles <= 0) st = AGP_ERR_HIP;
  } else {
    launch_trailing_update_as(variant, ctx->stream, dC, ldc, dP, dP, ldp, M, K);
  }
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  AGP_HIP_CHECK(ctx, hipMemcpy(C, dC, cb, hipMemcpyDeviceToHost));
  (void)hipFree(dC); (void)hipFree(dP);
  if (dP32) (void)hipFree(dP32);
  return st;
}

// The same on a stream restricted to a CU mask (hipExtStreamCreateWithCUMask): mask_words 32-bit words, bit i = CU i.
// At the same time (optional, chain_reps > 0) the panel chain of an n = M block runs on the context's main stream:
// *chain_us = its time per 512-wide panel phase while the masked bulk updates are in flight.
AGP_DEBUG_API int agp_debug_time_masked_update(agp_context *ctx, int64_t M, int64_t K, int variant, int reps, const uint32_t *mask,
                                 int mask_words, int chain_reps, double *ms_out, double *chain_us) {
  if (!ctx || M <= 0 || K <= 0 || reps <= 0 || !ms_out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  hipStream_t sm = nullptr;
  if (mask && mask_words > 0) AGP_HIP_CHECK(ctx, hipExtStreamCreateWithCUMask(&sm, (uint32_t)mask_words, mask));
  else AGP_HIP_CHECK(ctx, hipStreamCreateWithFlags(&sm, hipStreamNonBlocking));
  const long long ld = M + 8;
  double *dC = nullptr, *dP = nullptr, *pA = nullptr, *pimg = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dC, sizeof(double) * (size_t)ld * (size_t)M));
  AGP_HIP_CHECK(ctx, hipMalloc(&dP, sizeof(double) * (size_t)ld * (size_t)K));
  AGP_HIP_CHECK(ctx, hipMemset(dC, 0, sizeof(double) * (size_t)ld * (size_t)M));
  AGP_HIP_CHECK(ctx, hipMemset(dP, 0, sizeof(double) * (size_t)ld * (size_t)K));
  const long long pw = 512, plda = M + 8;
  AGP_HIP_CHECK(ctx, hipMalloc(&pA, sizeof(double) * (size_t)plda * (size_t)pw));
  AGP_HIP_CHECK(ctx, hipMalloc(&pimg, sizeof(double) * 4 * 36 * 256));
  AGP_HIP_CHECK(ctx, hipMemset(pA, 0, sizeof(double) * (size_t)plda * (size_t)pw));
  hipEvent_t e0, e1, c0, c1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  AGP_HIP_CHECK(ctx, hipEventCreate(&c0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&c1));
  for (int r = -2; r < reps; ++r) {
    if (r == 0) AGP_HIP_CHECK(ctx, hipEventRecord(e0, sm));
    launch_trailing_update_as(variant, sm, dC, ld, dP, dP, ld, M, K);
    if (r == 0 && chain_reps > 0) {
      AGP_HIP_CHECK(ctx, hipEventRecord(c0, ctx->stream));
      for (int c = 0; c < chain_reps; ++c) agp::panel_phase_public(ctx, ctx->stream, pA, M, plda, pimg, nullptr, 0, pw);
      AGP_HIP_CHECK(ctx, hipEventRecord(c1, ctx->stream));
    }
  }
  AGP_HIP_CHECK(ctx, hipEventRecord(e1, sm));
  AGP_HIP_CHECK(ctx, hipDeviceSynchronize());
  float ms = 0.f;
  AGP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  *ms_out = (double)ms / reps;
  if (chain_us) {
    *chain_us = 0.;
    if (chain_reps > 0) {
      AGP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, c0, c1));
      *chain_us = 1e3 * (double)ms / chain_reps;
    }
  }
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(c0); (void)hipEventDestroy(c1);
  (void)hipFree(dC); (void)hipFree(dP); (void)hipFree(pA); (void)hipFree(pimg);
  (void)hipStreamDestroy(sm);
  return AGP_OK;
}

// Average milliseconds of `reps` bulk trailing updates of an M x M matrix (device-side random-ish data).
AGP_DEBUG_API int agp_debug_time_trailing_update(agp_context *ctx, int64_t M, int64_t K, int variant, int reps, double *ms_out) {
  if (!ctx || M <= 0 || K <= 0 || reps <= 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long ld = M + 8;
  double *dC = nullptr, *dP = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dC, sizeof(double) * (size_t)ld * (size_t)M));
  AGP_HIP_CHECK(ctx, hipMalloc(&dP, sizeof(double) * (size_t)ld * (size_t)K));
  std::vector<double> h((size_t)ld * (size_t)K);
  unsigned long long x = 88172645463325252ull;
  for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5; }
  // (AGP_DEBUG_ZERO_OPERANDS=1: how much of the kernel's time depends on the DATA - at K = 512 the update is 11 % faster on
  // zeros, at K = 2048 not at all: power, with the memory traffic as the swing term; profiles/r04/bulk_update_vs_k.txt)
  // (=1: panel and C zero, =2: the panel only, =3: C only)
  int zero_mode = getenv("AGP_DEBUG_ZERO_OPERANDS") ? atoi(getenv("AGP_DEBUG_ZERO_OPERANDS")) : 0;
  std::vector<double> hz;
  if (zero_mode) hz.assign(h.size(), 0.);
  if (zero_mode == 4) { zero_mode = 2; hz.assign(h.size(), 0.3); }                                         // constant panel
  if (zero_mode == 5) { zero_mode = 2; for (size_t i = 0; i < h.size(); ++i) hz[i] = h[i] < 0. ? -0.3 : 0.3; }  // random signs only
  if (zero_mode == 6) { zero_mode = 2; for (size_t i = 0; i < h.size(); ++i) hz[i] = (i % 7 == 0) ? h[i] : 0.; }  // 1 in 7 entries non-zero
  AGP_HIP_CHECK(ctx, hipMemcpy(dP, (zero_mode == 1 || zero_mode == 2) ? hz.data() : h.data(), sizeof(double) * h.size(), hipMemcpyHostToDevice));
  for (long long c = 0; c < M; c += K) {
    const long long w = (M - c < K) ? M - c : K;
    AGP_HIP_CHECK(ctx, hipMemcpy(dC + c * ld, (zero_mode == 1 || zero_mode == 3) ? hz.data() : h.data(), sizeof(double) * (size_t)ld * (size_t)w,
                            hipMemcpyHostToDevice));
  }
  hipEvent_t e0, e1;
  AGP_HIP_CHECK(ctx, hipEventCreate(&e0));
  AGP_HIP_CHECK(ctx, hipEventCreate(&e1));
  int st = AGP_OK;
  float *dP32 = nullptr;  // variant 4: the fp32-product kernel on an fp32 copy of the panel
  const long long ld32 = (M + 7) / 8 * 8 + 8;
  if (variant == 4) {
    AGP_HIP_CHECK(ctx, hipMalloc(&dP32, sizeof(float) * (size_t)ld32 * (size_t)K));
    launch_convert_panel_f32(ctx->stream, dP, ld, M, K, dP32, ld32);
  }
  unsigned short *planes = nullptr;  // variant 5: the bf16 x 3 kernel on the panel's planes
  const int ntr5 = (int)((M + 127) / 128);
  long long olen = 0;
  const int *order = nullptr;
  if (variant == 5) {
    AGP_HIP_CHECK(ctx, hipMalloc(&planes, bf16x3_bytes(M, K)));
    launch_convert_panel_bf16x3(ctx->stream, dP, ld, M, K, planes);
    const long long tiles = (long long)ntr5 * (ntr5 + 1) / 2;
    if (tiles >= 1024) order = bulk_tile_order(ntr5, tiles, &olen);
  }
  double *scales = nullptr;  // variant 6: the fp16 x 2 kernel on the row-scaled panel's planes
  if (variant == 6) {
    const int stp = make_f16x2_planes(ctx, (zero_mode == 1 || zero_mode == 2) ? hz.data() : h.data(), ld, dP, M, K, &scales, &planes);
    if (stp != AGP_OK) return stp;
    const long long tiles = (long long)ntr5 * (ntr5 + 1) / 2;
    if (tiles >= 1024) order = bulk_tile_order(ntr5, tiles, &olen);
  }
  for (int r = -2; r < reps && st == AGP_OK; ++r) {
    if (r == 0) AGP_HIP_CHECK(ctx, hipEventRecord(e0, ctx->stream));
    if (variant == 4) launch_trailing_update_as(3, ctx->stream, dC, ld, dP, dP, ld, M, K, nullptr, dP32, dP32, ld32);
    else if (variant == 5) launch_update_bf16x3(ctx->stream, dC, ld, planes, M, 0, 0, M, M, K, order, olen);
    else if (variant == 6) launch_update_f16x2(ctx->stream, dC, ld, planes, M, 0, 0, scales + M, M, M, K, order, olen);
    else launch_trailing_update_as(variant, ctx->stream, dC, ld, dP, dP, ld, M, K);
  }
  AGP_HIP_CHECK(ctx, hipEventRecord(e1, ctx->stream));
  AGP_HIP_CHECK(ctx, hipEventSynchronize(e1));
  float ms = 0.f;
  AGP_HIP_CHECK(ctx, hipEventElapsedTime(&ms, e0, e1));
  *ms_out = (double)ms / reps;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  (void)hipFree(dC); (void)hipFree(dP);
  if (dP32) (void)hipFree(dP32);
  if (planes) (void)hipFree(planes);
  if (scales) (void)hipFree(scales);
  return st;
}

// X (n x ncols, ld n) = L^-1 B for L = the LL^T factor of the host matrix K (n x n, lower triangle, ld n) through
// forward_solve_wide (solve.hip): out of place, explicitly inverted 512 x 512 diagonal blocks.  n a multiple of 512.
AGP_DEBUG_API int agp_debug_forward_solve_wide(agp_context *ctx, const double *K, int64_t n, const double *B, int64_t ncols, double *X) {
  if (!ctx || !K || !B || !X || n < 1024 || n % 512 != 0 || ncols <= 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  agp_fit *f = nullptr;
  int st = agp_factor_create(ctx, K, n, n, 0, AGP_HOST, &f);
  if (st != AGP_OK) return st;
  const long long ldb = n + 2;  // (a right-hand side whose leading dimension is not the matrix's)
  double *dB = nullptr, *dX = nullptr, *dW = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dB, sizeof(double) * (size_t)ldb * (size_t)ncols));
  AGP_HIP_CHECK(ctx, hipMalloc(&dX, sizeof(double) * (size_t)ldb * (size_t)ncols));
  AGP_HIP_CHECK(ctx, hipMalloc(&dW, sizeof(double) * (size_t)n * 512));
  AGP_HIP_CHECK(ctx, hipMemcpy2D(dB, sizeof(double) * (size_t)ldb, B, sizeof(double) * (size_t)n, sizeof(double) * (size_t)n, (size_t)ncols,
                                 hipMemcpyHostToDevice));
  AGP_HIP_CHECK(ctx, hipMemset(dX, 0, sizeof(double) * (size_t)ldb * (size_t)ncols));
  invert_wide_blocks(ctx->stream, f->A, n, f->lda, f->invd, WIDE_BW, dW);
  forward_solve_wide(ctx->stream, f->A, n, f->lda, dW, dB, ldb, dX, ldb, ncols);
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  AGP_HIP_CHECK(ctx, hipMemcpy2D(X, sizeof(double) * (size_t)n, dX, sizeof(double) * (size_t)ldb, sizeof(double) * (size_t)n, (size_t)ncols,
                                 hipMemcpyDeviceToHost));
  (void)hipFree(dB); (void)hipFree(dX); (void)hipFree(dW);
  agp_fit_destroy(f);
  return AGP_OK;
}

// In-place LL^T of the lower triangle of a host matrix; y (optional) -> L^-1 y.
AGP_DEBUG_API int agp_debug_factor(agp_context *ctx, double *A, int64_t n, int64_t lda, double *y, double *log_det,
                     int64_t *bad_pivot) {
  if (!ctx || !A || n <= 0 || lda < n) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const size_t ab = sizeof(double) * (size_t)lda * (size_t)n;
  const long long nblk = (n + NB - 1) / NB;
  double *dA = nullptr, *dI = nullptr, *dy = nullptr;
  AGP_HIP_CHECK(ctx, hipMalloc(&dA, ab));
  AGP_HIP_CHECK(ctx, hipMalloc(&dI, sizeof(double) * (size_t)nblk * (36 * MB * MB)));
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

// ---------------------------------------------------------------------------------------------------------------
// Test / measurement instrumentation that used to sit in the product library's ABI
// ---------------------------------------------------------------------------------------------------------------
namespace agp {
// ---- bare MFMA issue loop: measured fp64 matrix peak of this device ----------
__global__ __launch_bounds__(256) void mfma_peak_kernel(double *sink, int iters, double a0, double b0) {
  v4d acc[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc[i] = v4zero();
  double a = a0 + threadIdx.x * 1e-9, b = b0 - threadIdx.x * 1e-9;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = mfma16(a, b, acc[i]);
  }
  double s = 0.;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) sink[0] = s;  // keep the loop live
}

static int mfma_f64_peak(hipStream_t s, int iters, double *tflops) {
  double *sink = nullptr;
  if (hipMalloc(&sink, 8) != hipSuccess) return AGP_ERR_HIP;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  const int blocks = 256 * 2;  // 2 workgroups of 4 waves per CU: 2 waves per SIMD
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, sink, 16, 1.0, 2.0);
  (void)hipEventRecord(e0, s);
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(256), 0, s, sink, iters, 1.0, 2.0);
  (void)hipEventRecord(e1, s);
  hipError_t e = hipEventSynchronize(e1);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, e0, e1);
  const double flop = (double)blocks * 4.0 * (double)iters * 8.0 * 2.0 * 16 * 16 * 4;
  *tflops = flop / (ms * 1e-3) / 1e12;
  (void)hipEventDestroy(e0);
  (void)hipEventDestroy(e1);
  (void)hipFree(sink);
  return e == hipSuccess ? AGP_OK : AGP_ERR_HIP;
}

// A transport that moves nothing: times ONE rank's share of a G-rank sharded fit on a box with one GPU - same kernels,
// same shapes, same launch chain; the peers' data is whatever the buffers hold, so the numerical result is
// meaningless (scripts/time_sharded_rank.py).
struct NullComm : HostReducingComm {
  int broadcast(ShardOps &, int, double *, long long, int) override { return AGP_OK; }
  int all_gather(ShardOps &, int, const double *, double *, long long) override { return AGP_OK; }
  int all_reduce(ShardOps &, int, double *, long long, int) override { return AGP_OK; }
  int all_reduce_host(double *, long long, int) override { return AGP_OK; }
};
}  // namespace agp

extern "C" {

// the cycle stamps the factoring workgroup of the LAST potrf / panel launch left (a -DAGP_POTRF_TIMING build; zeros otherwise)
AGP_DEBUG_API int agp_debug_potrf_probe(agp_context *ctx, unsigned long long *out) {
  if (!ctx || !out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  AGP_HIP_CHECK(ctx, hipDeviceSynchronize());
  read_potrf_probe(out);
  return AGP_OK;
}

// cycles and 100 MHz ticks of one tile of the LAST fp64 bulk launch (a -DAGP_BULK_STAMPS build; zeros otherwise)
AGP_DEBUG_API int agp_debug_bulk_probe(agp_context *ctx, unsigned long long *out) {
  if (!ctx || !out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  AGP_HIP_CHECK(ctx, hipDeviceSynchronize());
  read_bulk_probe(out);
  return AGP_OK;
}

// the phase cycle sums one workgroup of the LAST bf16 x 3 bulk launch left (a -DAGP_BF16_STAMPS build; zeros otherwise)
AGP_DEBUG_API int agp_debug_bf16_probe(agp_context *ctx, unsigned long long *out) {
  if (!ctx || !out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  AGP_HIP_CHECK(ctx, hipDeviceSynchronize());
  read_bf16_probe(out);
  return AGP_OK;
}

AGP_DEBUG_API int agp_debug_mfma_f64_peak(agp_context *ctx, int iters, double *tflops) {
  if (!ctx || !tflops || iters <= 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return mfma_f64_peak(ctx->stream, iters, tflops);
}

AGP_DEBUG_API int agp_debug_comm_create_null(int nranks, int rank, agp_comm **out) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks) return AGP_ERR_INVALID_ARGUMENT;
  NullComm *c = new (std::nothrow) NullComm();
  agp_comm *h = new (std::nothrow) agp_comm();
  if (!c || !h) { delete c; delete h; return AGP_ERR_INVALID_ARGUMENT; }
  c->world = nranks;
  c->rank = rank;
  h->impl = c;
  *out = h;
  return AGP_OK;
}

}  // extern "C"
