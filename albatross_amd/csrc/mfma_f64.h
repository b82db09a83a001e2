// mfma_f64.h — v_mfma_f64_16x16x4_f64 wrapper and its lane maps (gfx950).
//
//   D(16x16) = A(16x4) * B(4x16) + C, one wave:
//     A operand : lane l holds A[m = l & 15][k = l >> 4]
//     B operand : lane l holds B[k = l >> 4][n = l & 15]
//     C/D       : lane l, register r holds D[m = (l >> 4) + 4 r][n = l & 15]
//   (f64 differs from every other dtype's C/D map — see
//    /opt/skills/guides/cdna_hip_programming.md §3 "Fragment layout";
//    tests/test_kernels_gpu.py::test_mfma_f64_lane_map checks it on hardware.)
//
// Because a product sums over k in any order, the k index of step s can be
// permuted freely as long as A and B agree.  All kernels here use
//     k(step s, lane l) = (l >> 4) + 4 s
// so that register s of a C/D tile is directly the B operand of step s of a
// following product that contracts over that tile's m index: accumulators feed
// the next MFMA with no lane movement and no LDS round trip.
#pragma once
#include <hip/hip_runtime.h>

namespace agp {

typedef double v4d __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v4d mfma16(double a, double b, v4d c) {
  return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

// (The other fp64 shape, v_mfma_f64_4x4x4_4b_f64, issues faster in a bare loop but an update kernel built on it runs
// at the same speed at a lower clock: DESIGN.md section 8; scripts/microbench/ keeps the probes.)
__device__ __forceinline__ v4d v4zero() {
  v4d z = {0., 0., 0., 0.};
  return z;
}

}  // namespace agp
