// Issue rates of the fp64 VALU instructions the covariance kernels are made of (exp_neg: v_rndne_f64, v_cvt_i32_f64,
// v_ldexp_f64, v_fma_f64; acos_fast: v_sqrt / v_rsq / v_rcp), one wave per SIMD... and four: cycles per wave-instruction.
// Standalone:  hipcc --offload-arch=gfx950 -O3 valu_f64_rates.hip -o valu_f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>

#define OPK(NAME, ASM)                                                                        \
  __global__ __launch_bounds__(256) void NAME(double *sink, int iters, long long *cyc) {      \
    double r[8];                                                                              \
    for (int i = 0; i < 8; ++i) r[i] = 1.0 + 0.001 * (threadIdx.x + i);                        \
    const long long t0 = __builtin_amdgcn_s_memtime();                                        \
    for (int it = 0; it < iters; ++it) {                                                      \
      _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(r[i]));            \
    }                                                                                         \
    const long long t1 = __builtin_amdgcn_s_memtime();                                        \
    double s = 0.;                                                                            \
    for (int i = 0; i < 8; ++i) s += r[i];                                                    \
    if (s == 123.456) sink[0] = s;                                                            \
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;                                \
  }

OPK(k_fma, "v_fma_f64 %0, %0, 1.0, 0.5")
OPK(k_mul, "v_mul_f64 %0, %0, 0.5")
OPK(k_add, "v_add_f64 %0, %0, 1.0")
OPK(k_rndne, "v_rndne_f64 %0, %0")
OPK(k_ldexp, "v_ldexp_f64 %0, %0, 1")
OPK(k_rcp, "v_rcp_f64 %0, %0")
OPK(k_rsq, "v_rsq_f64 %0, %0")
OPK(k_sqrt, "v_sqrt_f64 %0, %0")
OPK(k_cmp, "v_cmp_eq_f64 vcc, %0, %0")
OPK(k_max, "v_max_f64 %0, %0, 1.0")

__global__ __launch_bounds__(256) void k_cvt(double *sink, int iters, long long *cyc) {
  double r[8];
  int q[8];
  for (int i = 0; i < 8; ++i) r[i] = 1.0 + 0.001 * (threadIdx.x + i);
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(q[i]) : "v"(r[i]));
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  int s = 0;
  for (int i = 0; i < 8; ++i) s += q[i];
  if (s == 123456) sink[0] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

int main() {
  double *sink; long long *cyc, h;
  hipMalloc(&sink, 8); hipMalloc(&cyc, 8);
  const int iters = 20000;
  struct { const char *name; void (*k)(double *, int, long long *); } ks[] = {
      {"v_fma_f64", k_fma}, {"v_mul_f64", k_mul}, {"v_add_f64", k_add}, {"v_max_f64", k_max}, {"v_cmp_eq_f64", k_cmp}, {"v_rndne_f64", k_rndne},
      {"v_ldexp_f64", k_ldexp}, {"v_cvt_i32_f64", k_cvt}, {"v_rcp_f64", k_rcp}, {"v_rsq_f64", k_rsq}, {"v_sqrt_f64", k_sqrt}};
  for (int waves = 1; waves <= 4; waves *= 4)
    for (auto &e : ks) {
      // `waves` waves per SIMD: a workgroup of 256 threads = one wave per SIMD of a CU
      hipLaunchKernelGGL(e.k, dim3(256 * waves), dim3(256), 0, 0, sink, 100, cyc);
      hipDeviceSynchronize();
      hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
      hipEventRecord(a);
      hipLaunchKernelGGL(e.k, dim3(256 * waves), dim3(256), 0, 0, sink, iters, cyc);
      hipEventRecord(b); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      // s_memtime ticks at 100 MHz; the kernel time gives cycles at the running clock: report ns per wave-instruction per SIMD
      const double per = (double)ms * 1e6 / ((double)iters * 8 * waves);
      printf("%-16s %d wave(s)/SIMD: %.2f ns per wave-instruction (= %.1f cycles at 2.4 GHz)\n", e.name, waves, per, per * 2.4);
    }
  return 0;
}
