// Which CU does a workgroup run on?  XCC_ID and HW_ID of every workgroup of a 1024-workgroup launch.
// hipcc --offload-arch=gfx950 -O2 -o hwid_probe hwid_probe.hip && ./hwid_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <map>
#include <set>
__global__ void probe(unsigned *out) {
  const unsigned xcc = __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20);
  const unsigned hw = __builtin_amdgcn_s_getreg((32 - 1) << 11 | 0 << 6 | 4);
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hw; }
  __builtin_amdgcn_s_sleep(100);
}
int main() {
  const int n = 1024;
  unsigned *d, h[2 * n];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(probe, dim3(n), dim3(256), 0, 0, d);
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  std::set<unsigned> xccs, cus;
  std::set<unsigned long long> pairs;
  for (int i = 0; i < n; ++i) {
    xccs.insert(h[2 * i] & 15);
    cus.insert((h[2 * i + 1] >> 8) & 255);
    pairs.insert((unsigned long long)(h[2 * i] & 15) << 8 | ((h[2 * i + 1] >> 8) & 255));
  }
  printf("distinct XCC ids %zu, distinct HW_ID[15:8] %zu, distinct (xcc, cu) %zu\n", xccs.size(), cus.size(), pairs.size());
  for (int i = 0; i < 16; ++i) printf("wg %d: xcc %u hw_id 0x%08x cu %u sh %u se %u\n", i, h[2 * i] & 15, h[2 * i + 1], (h[2 * i + 1] >> 8) & 15, (h[2 * i + 1] >> 12) & 1, (h[2 * i + 1] >> 13) & 7);
  return 0;
}
