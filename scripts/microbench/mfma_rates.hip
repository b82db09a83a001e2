// Sustained MFMA issue rates on random operands, all CUs busy, long enough for the clocks to settle:
// fp64 16x16x4, int8 32x32x32, int8 16x16x64.  Standalone:  hipcc --offload-arch=gfx950 -O3 mfma_rates.hip -o mfma_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

typedef double v4d __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));

__device__ inline unsigned hash32(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}

__global__ __launch_bounds__(256) void f64_kernel(double *sink, int iters) {
  v4d acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4d{0., 0., 0., 0.};
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    a[i] = (double)hash32(threadIdx.x * 8 + i + blockIdx.x * 977) / 4294967296.0 - 0.5;
    b[i] = (double)hash32(threadIdx.x * 8 + 4 + i + blockIdx.x * 131) / 4294967296.0 - 0.5;
  }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[(i + it) & 3], acc[i], 0, 0, 0);
  }
  double s = 0.;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123.456) sink[0] = s;
}

__global__ __launch_bounds__(256) void i8_32_kernel(int *sink, int iters) {
  v16i acc[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  v4i a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = (int)hash32(threadIdx.x * 64 + i * 4 + j + blockIdx.x * 977);
      b[i][j] = (int)hash32(threadIdx.x * 64 + 32 + i * 4 + j + blockIdx.x * 131);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i], b[(i + it) & 3], acc[i], 0, 0, 0);
  }
  int s = 0;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 16; ++j) s += acc[i][j];
  if (s == 123456789) sink[0] = s;
}

__global__ __launch_bounds__(256) void i8_16_kernel(int *sink, int iters) {
  v4i acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = v4i{0, 0, 0, 0};
  v4i a[4], b[4];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      a[i][j] = (int)hash32(threadIdx.x * 64 + i * 4 + j + blockIdx.x * 977);
      b[i][j] = (int)hash32(threadIdx.x * 64 + 32 + i * 4 + j + blockIdx.x * 131);
    }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[(i + it) & 3], acc[i], 0, 0, 0);
  }
  int s = 0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  if (s == 123456789) sink[0] = s;
}

template <typename K>
static double time_ms(K launch, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  launch();  // warm up
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) launch();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0); hipEventDestroy(e1);
  return ms / reps;
}

int main(int argc, char **argv) {
  const int wgs_per_cu = argc > 1 ? atoi(argv[1]) : 2;
  const int blocks = 256 * wgs_per_cu;
  void *sink = nullptr;
  hipMalloc(&sink, 64);
  const int reps = 6;
  {
    const int iters = 40000;
    const double ms = time_ms([&] { hipLaunchKernelGGL(f64_kernel, dim3(blocks), dim3(256), 0, 0, (double *)sink, iters); }, reps);
    const double ops = (double)blocks * 4 * iters * 8.0 * 2.0 * 16 * 16 * 4;
    printf("fp64 16x16x4   : %8.2f ms per launch, %8.1f TFLOP/s\n", ms, ops / ms / 1e9);
  }
  {
    const int iters = 80000;
    const double ms = time_ms([&] { hipLaunchKernelGGL(i8_32_kernel, dim3(blocks), dim3(256), 0, 0, (int *)sink, iters); }, reps);
    const double ops = (double)blocks * 4 * iters * 4.0 * 2.0 * 32 * 32 * 32;
    printf("int8 32x32x32  : %8.2f ms per launch, %8.1f TOP/s\n", ms, ops / ms / 1e9);
  }
  {
    const int iters = 80000;
    const double ms = time_ms([&] { hipLaunchKernelGGL(i8_16_kernel, dim3(blocks), dim3(256), 0, 0, (int *)sink, iters); }, reps);
    const double ops = (double)blocks * 4 * iters * 8.0 * 2.0 * 16 * 16 * 64;
    printf("int8 16x16x64  : %8.2f ms per launch, %8.1f TOP/s\n", ms, ops / ms / 1e9);
  }
  hipFree(sink);
  return 0;
}
