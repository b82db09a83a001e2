// trsm_kernel.h — the tile image of a factored 128 x 128 diagonal block and the MFMA substitution kernel against it
// (trsm_micro_kernel): shared by chol.hip (panel TRSM of the factorisation) and solve.hip (multi-RHS substitutions).
#pragma once
#include "common.h"
#include "mfma_f64.h"

namespace agp {

constexpr int NTILE = NMB * (NMB + 1) / 2;   // 36 lower 16x16 tiles
constexpr int IMG_DOUBLES = NTILE * MB * MB;  // 9216 doubles = 72 KiB per diagonal block

// LDS image: only the 36 lower micro tiles, each column-major 16x16
// (tile (ib, kb), ib >= kb, at index ib(ib+1)/2 + kb).  An MFMA operand
// fragment of a tile (element [k*16 + m], k = (lane >> 4) + 4 s, m = lane & 15)
// is 64 consecutive doubles per k-step: conflict-free ds_read_b64.
//
// The kernel also emits the "tile image" of the factored block to global
// memory: the same 36 tiles with the off-diagonal ones NEGATED and the diagonal
// ones replaced by their INVERSES.  That image is exactly the set of MFMA
// A-operand fragments the substitution kernels need, so they stage it with a
// straight coalesced copy.
__device__ __forceinline__ int tile_off(int ib, int kb) { return (ib * (ib + 1) / 2 + kb) * (MB * MB); }

// ---------------------------------------------------------------------------
// Substitution against one NB x NB diagonal block over its micro blocks.
//   Y (NB x 16 per wave, held as 8 C/D tiles) <- L11^-1 Y        (TRANS = false)
//   Y                                         <- L11^-T Y        (TRANS = true)
// Element (m, n) of Y lives at base[m * stride_m + n * stride_n]:
//   panel TRSM  X <- X L11^-T : Y = X^T, stride_m = lda, stride_n = 1
//   left  TRSM  V <- L11^-1 V : Y = V,   stride_m = 1,   stride_n = ldv
// LDS holds the 36 lower 16x16 tiles of L11 as ready-made MFMA A-operand
// fragments (negated off-diagonal tiles, inverted diagonal tiles).
// ---------------------------------------------------------------------------
constexpr int NFRAG_TILES = NTILE;

struct TrsmArgs {
  const double *img;  // tile image of the diagonal block (written by potrf_diag_kernel)
  int nbk;
  double *Y;  // element (0, 0) of the block to be solved
  long long stride_m, stride_n;
  long long ncols;  // number of n (panel rows / V columns)
  const double *z;  // z_b (nbk) or nullptr          (FUSE_Y only)
  double *yrest;    // y entries matching n = 0..ncols (FUSE_Y only)
  // batched launches (blockIdx.y = diagonal block index): element strides
  long long batch_img, batch_Y;
  long long n_total;  // matrix size, to derive nbk per batch entry (0: use nbk)
  long long batch_z = 0;  // FUSE_Y: offset of z / yrest per batch entry
};

// the fragment image of L11 into LDS (F: NFRAG_TILES * 256 doubles, zs: NB doubles); all 256 threads
// Image element [tile * 256 + k * 16 + m] is the A-operand value T[m][k] of
// the forward solve; the transposed solve needs T^T of every tile.
template <bool TRANS, bool FUSE_Y>
__device__ __forceinline__ void trsm_micro_stage(const TrsmArgs &p, double *F, double *zs) {
  const int tid = threadIdx.x;
  if (!TRANS) {
#pragma unroll
    for (int it = 0; it < IMG_DOUBLES / 2 / 256; ++it) {
      const int e = 2 * (tid + 256 * it);
      *reinterpret_cast<double2 *>(F + e) = *reinterpret_cast<const double2 *>(p.img + e);
    }
  } else {
#pragma unroll 4
    for (int e = tid; e < IMG_DOUBLES; e += 256) {
      const int m = e & 15, k = (e >> 4) & 15, t = e >> 8;
      F[e] = p.img[t * 256 + m * 16 + k];
    }
  }
  if (FUSE_Y && tid < NB) zs[tid] = (tid < p.nbk) ? p.z[tid] : 0.;
}

// one wave: the 16 columns n0 .. n0 + 15 of Y against the staged image (no barrier inside)
template <bool TRANS, bool FUSE_Y>
__device__ __forceinline__ void trsm_micro_solve(const TrsmArgs &p, const double *F, const double *zs, const long long n0) {
  const int lane = threadIdx.x & 63;
  const int ln = lane & 15, lg = lane >> 4;
  if (n0 >= p.ncols) return;
  const bool nok = n0 + ln < p.ncols;
  double *base = p.Y + (n0 + ln) * p.stride_n;

  v4d Y[NMB];
  // all 8 input tiles are requested up front: one HBM round trip, not eight
#pragma unroll
  for (int jb = 0; jb < NMB; ++jb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = jb * MB + lg + 4 * r;
      Y[jb][r] = (nok && m < p.nbk) ? base[m * p.stride_m] : 0.;
    }
  if (!TRANS) {
#pragma unroll
    for (int jb = 0; jb < NMB; ++jb) {
      // four independent accumulation chains (one per k-step) instead of one
      // chain of 4 jb dependent MFMAs: the dependent-issue latency of the f64
      // MFMA (~190 cycles) is what this kernel is bound by
      v4d pa[4] = {Y[jb], v4zero(), v4zero(), v4zero()};
#pragma unroll
      for (int ib = 0; ib < jb; ++ib) {
        const double *f = F + (jb * (jb + 1) / 2 + ib) * 256 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) pa[s] = mfma16(f[s * 64], Y[ib][s], pa[s]);
      }
      const v4d acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
      v4d po[4] = {v4zero(), v4zero(), v4zero(), v4zero()};
      const double *f = F + (jb * (jb + 1) / 2 + jb) * 256 + lane;
#pragma unroll
      for (int s = 0; s < 4; ++s) po[s] = mfma16(f[s * 64], acc[s], po[s]);
      const v4d out = (po[0] + po[1]) + (po[2] + po[3]);
      Y[jb] = out;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = jb * MB + lg + 4 * r;
        if (nok && m < p.nbk) base[m * p.stride_m] = out[r];
      }
    }
  } else {
#pragma unroll
    for (int jb = NMB - 1; jb >= 0; --jb) {
      v4d pa[4] = {Y[jb], v4zero(), v4zero(), v4zero()};
#pragma unroll
      for (int ib = NMB - 1; ib > jb; --ib) {
        // image tile index of the stored pair (row block ib, col block jb)
        const double *f = F + (ib * (ib + 1) / 2 + jb) * 256 + lane;
#pragma unroll
        for (int s = 0; s < 4; ++s) pa[s] = mfma16(f[s * 64], Y[ib][s], pa[s]);
      }
      const v4d acc = (pa[0] + pa[1]) + (pa[2] + pa[3]);
      v4d po[4] = {v4zero(), v4zero(), v4zero(), v4zero()};
      const double *f = F + (jb * (jb + 1) / 2 + jb) * 256 + lane;
#pragma unroll
      for (int s = 0; s < 4; ++s) po[s] = mfma16(f[s * 64], acc[s], po[s]);
      const v4d out = (po[0] + po[1]) + (po[2] + po[3]);
      Y[jb] = out;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int m = jb * MB + lg + 4 * r;
        if (nok && m < p.nbk) base[m * p.stride_m] = out[r];
      }
    }
  }

  if (FUSE_Y) {
    // y[n] -= sum_m X[n][m] z[m]   (forward substitution carried by the panel)
    double part = 0.;
#pragma unroll
    for (int jb = 0; jb < NMB; ++jb)
#pragma unroll
      for (int r = 0; r < 4; ++r) part += Y[jb][r] * zs[jb * MB + lg + 4 * r];
    part += __shfl_xor(part, 16, 64);
    part += __shfl_xor(part, 32, 64);
    if (lg == 0 && nok) p.yrest[n0 + ln] -= part;
  }
}


// Exchanges inside a row of 16 lanes on the DPP path (no LDS crossbar trip: __shfl_xor is a ds_bpermute, ~100 cycles of
// latency on a dependent chain): lane <-> lane ^ 1, ^ 2 by quad permutation, ^ 4 and ^ 8 by the mirrors
// (i ^ 7 then i ^ 3, i ^ 15 then i ^ 7).
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  int lo = __double2loint(v), hi = __double2hiint(v);
  lo = __builtin_amdgcn_mov_dpp(lo, CTRL, 0xF, 0xF, true);
  hi = __builtin_amdgcn_mov_dpp(hi, CTRL, 0xF, 0xF, true);
  return __hiloint2double(hi, lo);
}
template <int MASK>
__device__ __forceinline__ double row_xor(double v) {
  static_assert(MASK == 1 || MASK == 2 || MASK == 4 || MASK == 8, "inside a row of 16 lanes");
  if (MASK == 1) return dpp_move<0xB1>(v);                   // quad_perm [1, 0, 3, 2]
  if (MASK == 2) return dpp_move<0x4E>(v);                   // quad_perm [2, 3, 0, 1]
  if (MASK == 4) return dpp_move<0x1B>(dpp_move<0x141>(v));  // row_half_mirror, quad_perm [3, 2, 1, 0]
  return dpp_move<0x141>(dpp_move<0x140>(v));                // row_mirror, row_half_mirror
}

// x = L_bb^-T t for ONE vector by micro blocks, bottom-up, right-looking: one wave, values in registers.
// lane = (k = lane >> 2, mq = lane & 3): element k of every micro block, replicated over the four mq lanes, which
// split the sums over m.  Image tile (ib, kb) holds -L[16 ib + m][16 kb + k] at [k * 16 + m], the diagonal tile
// inv(L_ii)[m][k] at the same place: both substitution steps are out[k] += sum_m tile[k * 16 + m] in[m].
// F: the image in LDS (as stored), ts: t (NB doubles, LDS).  emit(index, value) receives x[16 jb + k] once, from the
// mq == 0 lanes, as soon as it is final (bottom micro block first).
template <class Emit>
__device__ __forceinline__ void micro_backsub_wave(const double *F, const double *ts, Emit emit) {
  const int lane = threadIdx.x & 63;
  const int k = lane >> 2, mq = lane & 3;
  double t[NMB];
#pragma unroll
  for (int jb = 0; jb < NMB; ++jb) t[jb] = ts[jb * MB + k];
#pragma unroll
  for (int jb = NMB - 1; jb >= 0; --jb) {
    const double *D = F + tile_off(jb, jb) + k * MB + 4 * mq;
    double part = 0.;
#pragma unroll
    for (int u = 0; u < 4; ++u) part += D[u] * __shfl(t[jb], (4 * mq + u) << 2, 64);
    part += row_xor<1>(part);
    part += row_xor<2>(part);
    const double xk = part;  // x[16 jb + k]
    if (mq == 0) emit(jb * MB + k, xk);
    if (jb == 0) break;
    double xm[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) xm[u] = __shfl(xk, (4 * mq + u) << 2, 64);
#pragma unroll
    for (int kb = 0; kb < jb; ++kb) {
      const double *T = F + tile_off(jb, kb) + k * MB + 4 * mq;
      double q = T[0] * xm[0] + T[1] * xm[1] + T[2] * xm[2] + T[3] * xm[3];
      q += row_xor<1>(q);
      q += row_xor<2>(q);
      t[kb] += q;
    }
  }
}

// 16 per-lane partial sums (one per column) -> their totals over the 64 lanes: a butterfly that halves the live values
// at every step (15 exchanges instead of 16 x 6), then the four 16-lane groups.  On return lanes 0 .. 15 hold the
// total of column colsum16_index(lane).
__device__ __forceinline__ double colsum16(double (&acc)[16]) {
  const int lane = threadIdx.x & 63;
#define AGP_COLSUM_STEP(BIT)                                                  \
  {                                                                           \
    constexpr int half = 8 >> BIT;                                            \
    const bool up = (lane >> BIT) & 1;                                        \
    _Pragma("unroll") for (int i = 0; i < half; ++i) {                        \
      const double keep = up ? acc[i + half] : acc[i];                        \
      const double send = up ? acc[i] : acc[i + half];                        \
      acc[i] = keep + row_xor<(1 << BIT)>(send);                              \
    }                                                                         \
  }
  AGP_COLSUM_STEP(0)
  AGP_COLSUM_STEP(1)
  AGP_COLSUM_STEP(2)
  AGP_COLSUM_STEP(3)
#undef AGP_COLSUM_STEP
  double sum = acc[0];
  sum += __shfl_xor(sum, 16, 64);
  sum += __shfl_xor(sum, 32, 64);
  return sum;
}
__device__ __forceinline__ int colsum16_index(int lane) {
  return ((lane & 1) << 3) | ((lane & 2) << 1) | ((lane & 4) >> 1) | ((lane & 8) >> 3);
}

template <bool TRANS, bool FUSE_Y>
__global__ __launch_bounds__(256, 2) void trsm_micro_kernel(TrsmArgs p) {
  __builtin_amdgcn_s_setprio(AGP_CHAIN_PRIO);  // panel chain (see potrf_diag_kernel)
  __shared__ double F[NFRAG_TILES * 4 * 64 + NB];
  double *zs = F + NFRAG_TILES * 4 * 64;
  if (blockIdx.y > 0 || p.n_total > 0) {
    const long long b = blockIdx.y;
    p.img += b * p.batch_img;
    p.Y += b * p.batch_Y;
    if (FUSE_Y) {
      p.z += b * p.batch_z;
      p.yrest += b * p.batch_z;
    }
    if (p.n_total > 0) {
      const long long left = p.n_total - b * NB;
      p.nbk = (int)(left < NB ? left : NB);
    }
  }
  trsm_micro_stage<TRANS, FUSE_Y>(p, F, zs);
  __syncthreads();
  trsm_micro_solve<TRANS, FUSE_Y>(p, F, zs, ((long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 16);
}

}  // namespace agp
