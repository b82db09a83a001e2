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

// v_mfma_f64_4x4x4_4b_f64: four independent 4 x 4 x 4 products D_b = A_b * B_b + C_b in one instruction
// (lane maps measured with scripts/probe_mfma44.py; cbsz / abid have no effect on the f64 shapes):
//     A operand : lane l holds A_b[i = l & 3][k = l >> 4],  b = (l >> 2) & 3
//     B operand : lane l holds B_b[k = l >> 4][j = l & 3],  b = (l >> 2) & 3
//     C/D       : lane l holds D_b[i = l >> 4][j = l & 3],  b = (l >> 2) & 3
// It issues at 76 TFLOP/s chip-wide on MI355X where the 16x16x4 shape saturates at 47
// (scripts/diag_mfma_shapes.py), so the GEMM kernels build their 16 x 16 tiles from FOUR of these:
// call `rot` pairs B block b (4 values of the operand indexed by l & 15) with A block (b + rot) & 3, i.e.
// the A fragment is read from LDS four times with its 4-element groups rotated.  A 16 x 16 x 4 product
// D[m][n] (m: A index, n: B index) then lives in four accumulators:
//     acc[rot], lane l  =  D[m = 4 (((l >> 2) + rot) & 3) + (l >> 4)][n = l & 15]
__device__ __forceinline__ double mfma4(double a, double b, double c) {
  return __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c, 0, 0, 0);
}

// x of lane (row r, position p of 16) <- x of lane (r, (p + 4 groups) mod 16): the 4-element groups of a fragment
// rotated inside each 16-lane row by DPP row_ror, without another LDS read.
template <int GROUPS>
__device__ __forceinline__ double rotate_groups(double x) {
  // row_ror:n moves lane i's value to lane (i + n) mod 16, i.e. lane p receives lane (p - n) mod 16;
  // receiving from (p + 4 GROUPS) therefore is a right rotation by 16 - 4 GROUPS
  constexpr int ctrl = 0x120 + ((16 - 4 * GROUPS) & 15);
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(x), ctrl, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(x), ctrl, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

__device__ __forceinline__ v4d v4zero() {
  v4d z = {0., 0., 0., 0.};
  return z;
}

}  // namespace agp
