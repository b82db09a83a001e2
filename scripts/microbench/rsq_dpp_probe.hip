// Does v_rsq_f64 honour a DPP row_newbcast operand on gfx950?  (round 6: it assembles; this prints what it computes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double *in, double *out, int nops) {
  double a = in[threadIdx.x], r, m;
  if (nops == 0) asm volatile("v_rsq_f64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(a));
  else asm volatile("s_nop 4\n\tv_rsq_f64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\ts_nop 4" : "=v"(r) : "v"(a));
  asm volatile("s_nop 1\n\tv_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(m) : "v"(a));
  out[threadIdx.x] = r;
  out[64 + threadIdx.x] = m;
}
int main() {
  double h[64], o[128], *d, *e;
  for (int i = 0; i < 64; ++i) h[i] = 1.0 + i;
  hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  for (int nops = 0; nops < 2; ++nops) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e, nops);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    printf("nops=%d: lane 0: rsq_dpp %.6f (own 1/sqrt(1)=1, bcast lane 3: 1/sqrt(4)=0.5), mov_dpp %.1f; lane 20: rsq_dpp %.6f (own %.6f, bcast lane 19: %.6f) mov %.1f\n",
           nops, o[0], o[64], o[20], 1 / sqrt(21.), 1 / sqrt(20.), o[84]);
  }
  return 0;
}
