// gram.hip — K1/K2: pairwise covariance evaluation.
//
// Replaces compute_covariance_matrix (include/albatross/src/covariance_functions/
// callers.hpp:38-166): one workgroup produces a 128 x 32 tile of the
// column-major output.  The two coordinate panels (128 + 32 points) are staged
// once through LDS as structure-of-arrays; every lane owns two consecutive
// rows, so each wave store is 64 lanes x 16 B = 1 KiB of one output column
// (fully coalesced), and the y-point of a column is an LDS broadcast read.
// HBM-write bound: 8 B per entry out, 8*dim B per point in.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "common.h"

namespace agp {

constexpr int TM = 128;  // tile rows
constexpr int TN = 32;   // tile cols
constexpr int GRAM_THREADS = 256;

template <int DIMP>
struct TileLds {
  double c[DIMP][TM + TN];
  double norm[TM + TN];
  double s[AGP_MAX_SCALE_COLUMNS][TM + TN];
  long long id[TM + TN];
};

template <int DIMP>
__device__ __forceinline__ void stage_points(TileLds<DIMP> &L, int slot0, int count, const FeatView &F,
                                             long long first, bool need_norm) {
  for (int t = threadIdx.x; t < count; t += GRAM_THREADS) {
    const long long g = first + t;
    const bool ok = g < F.n;
    double nn = 0.;
#pragma unroll
    for (int d = 0; d < DIMP; ++d) {
      const double v = (ok && d < F.dim) ? F.coords[g * F.dim + d] : 0.;
      L.c[d][slot0 + t] = v;
      nn += v * v;
    }
    L.norm[slot0 + t] = need_norm ? sqrt(nn) : 0.;
#pragma unroll
    for (int k = 0; k < AGP_MAX_SCALE_COLUMNS; ++k)
      L.s[k][slot0 + t] = (ok && k < F.nsc) ? F.scales[(long long)k * scale_stride(F) + g] : 0.;
    L.id[slot0 + t] = (ok && F.ids) ? F.ids[g] : -1;
  }
}

template <int DIMP>
__device__ __forceinline__ Point<DIMP> read_point(const TileLds<DIMP> &L, int slot) {
  Point<DIMP> p;
#pragma unroll
  for (int d = 0; d < DIMP; ++d) p.c[d] = L.c[d][slot];
  p.norm = L.norm[slot];
#pragma unroll
  for (int k = 0; k < AGP_MAX_SCALE_COLUMNS; ++k) p.s[k] = L.s[k][slot];
  p.id = L.id[slot];
  return p;
}

// SOP: the covariance function is given in sum-of-products form (cov_eval.h: eval_sop), by value in the kernel
// arguments; otherwise P is the postfix program for the interpreter.
template <int DIMP, bool SOP>
__global__ __launch_bounds__(GRAM_THREADS) void gram_kernel(const DevProgram *__restrict__ P, SopProgram sop, FeatView X, FeatView Y,
                                                            int symmetric, int lower_only, double *out,
                                                            long long ld, const double *diag_add,
                                                            int *nan_flag) {
  __shared__ TileLds<DIMP> L;
  const long long row0 = (long long)blockIdx.x * TM;
  const long long col0 = (long long)blockIdx.y * TN;
  if (lower_only && col0 > row0 + TM - 1) return;  // tile strictly above the diagonal
  const int metric_mask = SOP ? sop.metric_mask : P->metric_mask;
  const bool need_norm = (metric_mask & ((1 << AGP_METRIC_RADIAL) | (1 << AGP_METRIC_ANGULAR))) != 0;
  stage_points<DIMP>(L, 0, TM, X, row0, need_norm);
  stage_points<DIMP>(L, TM, TN, Y, col0, need_norm);
  __syncthreads();

  const int lane_row = 2 * (threadIdx.x & 63);
  const int cgrp = threadIdx.x >> 6;
  const Point<DIMP> xa = read_point<DIMP>(L, lane_row);
  const Point<DIMP> xb = read_point<DIMP>(L, lane_row + 1);
  const long long ra = row0 + lane_row, rb = ra + 1;
  const bool have_ids = X.ids != nullptr && Y.ids != nullptr;
  const bool both_meas = X.meas && Y.meas;
  const bool wide = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  bool saw_nan = false;
  // one column: add the diagonal term, note NaNs, store the two rows
  auto finish = [&](long long col, double va, double vb) {
    if (diag_add) {
      if (ra == col) va += diag_add[col];
      if (rb == col) vb += diag_add[col];
    }
    // only entries that are stored count (padding rows of an edge tile are zero
    // vectors: the angular metric makes NaN out of them)
    saw_nan = saw_nan || (ra < X.n && va != va) || (rb < X.n && vb != vb);
    double *dst = out + col * ld + ra;
    if (rb < X.n) {
      if (wide) {
        *reinterpret_cast<double2 *>(dst) = make_double2(va, vb);
      } else {
        dst[0] = va;
        dst[1] = vb;
      }
    } else if (ra < X.n) {
      dst[0] = va;
    }
  };
  if (SOP) {
    // the two rows of a thread per walk of the program: the pairs share the scalar work (cov_eval.h: eval_sop_n)
#pragma unroll 1
    for (int jj = 0; jj < TN / 4; ++jj) {
      const int cslot = cgrp * (TN / 4) + jj;
      const long long col = col0 + cslot;
      if (col >= Y.n) break;
      const Point<DIMP> y = read_point<DIMP>(L, TM + cslot);
      // symmetric Gram: the reference evaluates caller(xs[i], xs[j]) with i >= j and mirrors (callers.hpp:119-127)
      const Point<DIMP> *const xs2[2] = {&xa, &xb};
      const Point<DIMP> *const ys2[2] = {&y, &y};
      const bool sw2[2] = {symmetric && ra < col, symmetric && rb < col};
      double v2[2];
      eval_sop_n<DIMP, 2>(sop, xs2, ys2, sw2, have_ids, both_meas, v2);
      finish(col, v2[0], v2[1]);
    }
  } else {
#pragma unroll 1
    for (int jj = 0; jj < TN / 4; ++jj) {
      const int cslot = cgrp * (TN / 4) + jj;
      const long long col = col0 + cslot;
      if (col >= Y.n) break;
      const Point<DIMP> y = read_point<DIMP>(L, TM + cslot);
      // symmetric Gram: the reference evaluates caller(xs[i], xs[j]) with i >= j
      // and mirrors (callers.hpp:119-127); keep the same argument order.
      const double va = eval_pair<DIMP>(P, xa, y, symmetric && ra < col, have_ids, both_meas);
      const double vb = eval_pair<DIMP>(P, xb, y, symmetric && rb < col, have_ids, both_meas);
      finish(col, va, vb);
    }
  }
  if (saw_nan && nan_flag) atomicOr(nan_flag, 1);
}

// ---------------------------------------------------------------------------
// Fast path for the commonest trees:  radial<Euclidean>  and
// radial<Euclidean> + [measurement_only](IndependentNoise | Nugget).
// Same tile shape and the same IEEE operation sequence per pair as the generic
// evaluator (distance -> q -> exp), but the tree is fixed at compile time: no
// program walk, no evaluation stack.  ~2.5x fewer VALU instructions per pair.
// ---------------------------------------------------------------------------
struct FastParams {
  double length_scale, sigma;
  double noise_var;   // sigma_noise^2 (0 when there is no noise term)
  int has_noise;      // tree is radial + noise
  int noise_meas_only;  // the noise term is wrapped in MeasurementOnly
  // host-side precomputation (match_fast): the pair loop multiplies instead of dividing
  double sigma2;      // sigma^2
  double inv_l2;      // SquaredExponential: 1 / l^2      (exp(-(d/l)^2) = exp(-d^2 / l^2): no sqrt, no divide per pair)
  double cq;          // Exponential: 1 / l, Matern-3/2: sqrt(3) / l, Matern-5/2: sqrt(5) / l
};

// k(x, y) from the SQUARED Euclidean distance s2 (radial.hpp:25-33,191-198,289-297,461-470).  Differs from the
// reference's operation sequence (sqrt, divide by l, square) by a few ulp of the exponent argument; the parity bar of
// tests/test_gram_gpu.py (4e-16 max|K| + 2e-14 |value|) holds with a margin of 10x.
template <int OP>
__device__ __forceinline__ double radial_fast(double s2, const FastParams &fp) {
  if (fp.length_scale <= 0.) return 0.;
  if (OP == AGP_OP_SQUARED_EXPONENTIAL) {
    return fp.sigma2 * exp_neg(s2 * fp.inv_l2);
  } else if (OP == AGP_OP_EXPONENTIAL) {
    return fp.sigma2 * exp_neg(sqrt(s2) * fp.cq);
  } else if (OP == AGP_OP_MATERN32) {
    const double q = sqrt(s2) * fp.cq;
    return fp.sigma2 * (1 + q) * exp_neg(q);
  } else {
    const double q = sqrt(s2) * fp.cq;
    return fp.sigma2 * (1 + q + q * q * (1. / 3.)) * exp_neg(q);
  }
}

// NT values at once, without the test of the length scale's sign (radial.hpp: 0 if l <= 0; the caller handles it) and
// with the exponentials in lock step (cov_eval.h: exp_neg_n).  Per value the operations of radial_fast.
template <int OP, int NT>
__device__ __forceinline__ void radial_fast_n(const double (&s2)[NT], const FastParams &fp, double (&v)[NT]) {
  double arg[NT], pre[NT], e[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) {
    if (OP == AGP_OP_SQUARED_EXPONENTIAL) {
      arg[i] = s2[i] * fp.inv_l2;
      pre[i] = fp.sigma2;
    } else if (OP == AGP_OP_EXPONENTIAL) {
      arg[i] = sqrt(s2[i]) * fp.cq;
      pre[i] = fp.sigma2;
    } else if (OP == AGP_OP_MATERN32) {
      const double q = sqrt(s2[i]) * fp.cq;
      arg[i] = q;
      pre[i] = fp.sigma2 * (1 + q);
    } else {
      const double q = sqrt(s2[i]) * fp.cq;
      arg[i] = q;
      pre[i] = fp.sigma2 * (1 + q + q * q * (1. / 3.));
    }
  }
  exp_neg_n<NT>(arg, e);
#pragma unroll
  for (int i = 0; i < NT; ++i) v[i] = pre[i] * e[i];
}

template <int DIMP, int OP>
__device__ __forceinline__ void gram_fast_body(const FastParams &fp, FeatView X, FeatView Y, int lower_only, double *out, long long ld,
                                               const double *diag_add, int *nan_flag, long long blk_rows, long long blk_stride,
                                               long long tile_r = -1, long long tile_c = -1) {
  __shared__ double xs[DIMP][TM], ys[DIMP][TN];
  __shared__ long long xid[TM], yid[TN];
  if (blk_rows > 0) {  // blockIdx.z = one diagonal block of a block-diagonal Gram matrix (launch_gram_blocks)
    const long long z = blockIdx.z;
    X.coords += z * blk_rows * X.dim; Y.coords += z * blk_rows * Y.dim;
    if (X.ids) X.ids += z * blk_rows;
    if (Y.ids) Y.ids += z * blk_rows;
    X.n = Y.n = blk_rows;
    out += z * blk_stride;
    if (diag_add) diag_add += z * blk_rows;
  }
  const long long row0 = (tile_r >= 0 ? tile_r : (long long)blockIdx.x) * TM;
  const long long col0 = (tile_c >= 0 ? tile_c : (long long)blockIdx.y) * TN;
  if (lower_only && col0 > row0 + TM - 1) return;
  const bool have_ids = X.ids != nullptr && Y.ids != nullptr;
  for (int t = threadIdx.x; t < TM + TN; t += GRAM_THREADS) {
    const bool isx = t < TM;
    const FeatView &F = isx ? X : Y;
    const long long g = isx ? row0 + t : col0 + (t - TM);
    const bool ok = g < F.n;
#pragma unroll
    for (int d = 0; d < DIMP; ++d) {
      const double v = (ok && d < F.dim) ? F.coords[g * F.dim + d] : 0.;
      if (isx) xs[d][t] = v; else ys[d][t - TM] = v;
    }
    const long long id = (ok && F.ids) ? F.ids[g] : -1;
    if (isx) xid[t] = id; else yid[t - TM] = id;
  }
  __syncthreads();
  const int lane_row = 2 * (threadIdx.x & 63);
  const int cgrp = threadIdx.x >> 6;
  double xa[DIMP], xb[DIMP];
#pragma unroll
  for (int d = 0; d < DIMP; ++d) { xa[d] = xs[d][lane_row]; xb[d] = xs[d][lane_row + 1]; }
  const long long ida = xid[lane_row], idb = xid[lane_row + 1];
  const long long ra = row0 + lane_row, rb = ra + 1;
  const bool wide = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const bool noise_on = fp.has_noise && (!fp.noise_meas_only || (X.meas && Y.meas));
  bool saw_nan = false;
  const bool dead = fp.length_scale <= 0.;  // radial.hpp: the covariance is 0 for a non-positive length scale (wave-uniform)
  // Two columns (four entries) per trip: their exponentials run in lock step (cov_eval.h: exp_neg_n) - round 5's loop made
  // two separate calls per trip, which the compiler left one behind the other -, and the noise term's equality test
  // (noise.hpp:37-43) runs only in a wave that holds an equal pair: equal coordinates give a squared distance of exactly 0.
  // Same operations per entry as before: bit-identical matrices.
  for (int jj = 0; jj < TN / 4; jj += 2) {
    const int cslot = cgrp * (TN / 4) + jj;
    const long long col = col0 + cslot;
    if (col >= Y.n) break;
    const bool two = col + 1 < Y.n;  // (TN / 4 is even: the second column is this wave group's too)
    double s2[4] = {0., 0., 0., 0.};  // (row a, col 0), (row b, col 0), (row a, col 1), (row b, col 1)
    double yv[2][DIMP];
#pragma unroll
    for (int d = 0; d < DIMP; ++d) {
      yv[0][d] = ys[d][cslot];
      yv[1][d] = ys[d][cslot + 1];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int d = 0; d < DIMP; ++d) {
        const double ta = xa[d] - yv[c][d], tb = xb[d] - yv[c][d];
        s2[2 * c] += ta * ta;
        s2[2 * c + 1] += tb * tb;
      }
    double v[4];
    if (dead) {
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = 0.;
    } else {
      radial_fast_n<OP, 4>(s2, fp, v);
    }
    if (fp.has_noise) {  // lhs + rhs with rhs = noise (0 when not measurements / not equal)
      bool e[4] = {false, false, false, false};
      bool maybe = false;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        const long long yi = yid[cslot + c];
        maybe = maybe || (have_ids ? (ida == yi || idb == yi) : (s2[2 * c] == 0. || s2[2 * c + 1] == 0.));
      }
      if (noise_on && __any(maybe)) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          bool ea = true, eb = true;
#pragma unroll
          for (int d = 0; d < DIMP; ++d) {
            ea = ea && (xa[d] == yv[c][d]);
            eb = eb && (xb[d] == yv[c][d]);
          }
          if (have_ids) {
            ea = ida == yid[cslot + c];
            eb = idb == yid[cslot + c];
          }
          e[2 * c] = ea;
          e[2 * c + 1] = eb;
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = v[q] + (e[q] ? fp.noise_var : 0.);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      if (c == 1 && !two) break;
      const long long cc = col + c;
      double va = v[2 * c], vb = v[2 * c + 1];
      if (diag_add) {
        if (ra == cc) va += diag_add[cc];
        if (rb == cc) vb += diag_add[cc];
      }
      // only entries that are stored count (padding rows of an edge tile are zero
      // vectors: the angular metric makes NaN out of them)
      saw_nan = saw_nan || (ra < X.n && va != va) || (rb < X.n && vb != vb);
      double *dst = out + cc * ld + ra;
      if (rb < X.n) {
        if (wide) *reinterpret_cast<double2 *>(dst) = make_double2(va, vb);  // (non-temporal stores measured in round 6: no difference)
        else { dst[0] = va; dst[1] = vb; }
      } else if (ra < X.n) {
        dst[0] = va;
      }
    }
  }
  if (saw_nan && nan_flag) atomicOr(nan_flag, 1);
}

template <int DIMP, int OP>
__global__ __launch_bounds__(GRAM_THREADS) void gram_fast_kernel(FastParams fp, FeatView X, FeatView Y, int lower_only,
                                                                 double *out, long long ld, const double *diag_add,
                                                                 int *nan_flag, long long blk_rows, long long blk_stride) {
  gram_fast_body<DIMP, OP>(fp, X, Y, lower_only, out, ld, diag_add, nan_flag, blk_rows, blk_stride);
}

// The lower triangle of ONE symmetric Gram matrix with workgroups for the tiles on or below the diagonal only (blockIdx.x =
// index of such a tile, row tile by row tile as in gram_fast_batch_kernel below): at N = 16384 the rectangular grid starts
// 65536 workgroups of which 32256 return at once.
template <int DIMP, int OP>
__global__ __launch_bounds__(GRAM_THREADS) void gram_fast_tri_kernel(FastParams fp, FeatView X, double *out, long long ld,
                                                                     const double *diag_add, int *nan_flag) {
  constexpr long long PER = TM / TN;
  const long long idx = blockIdx.x;
  // tiles before row tile r: PER r (r + 1) / 2
  long long r = (long long)((sqrt(1. + 8. * (double)idx / (double)PER) - 1.) * 0.5);
  while (r > 0 && PER * r * (r + 1) / 2 > idx) --r;
  while (PER * (r + 1) * (r + 2) / 2 <= idx) ++r;
  const long long c = idx - PER * r * (r + 1) / 2;
  if (c * TN >= X.n) return;
  gram_fast_body<DIMP, OP>(fp, X, X, 1, out, ld, diag_add, nan_flag, 0, 0, r, c);
}

// `count` symmetric Gram matrices of one SHAPE (same fast-path operator, same DIMP) but their own parameters and
// features in ONE launch: blockIdx.z = problem, described by a table in device memory (agp_fit_create_batch and
// agp_nll_batch built them with one launch per problem: 256 launches are 1.2 ms of host enqueue time).
struct GramBatchItem {
  FastParams fp;
  FeatView X;
  double *out;
  const double *diag_add;
  int *nan_flag;
};

template <int DIMP, int OP>
__global__ __launch_bounds__(GRAM_THREADS) void gram_fast_batch_kernel(const GramBatchItem *__restrict__ items, int lower_only, long long ld) {
  const GramBatchItem it = items[blockIdx.z];
  // blockIdx.x = index of a tile ON OR BELOW the diagonal, row tile by row tile (row tile r holds the column tiles
  // 0 .. (r + 1) TM / TN - 1): the launch has no workgroups for the upper triangle - at N = 512 a batch of 256 problems
  // started 16384 workgroups of which 6144 returned at once, and the dispatcher's ~20 ns per workgroup showed in the launch
  constexpr long long PER = TM / TN;
  long long r = 0, left = blockIdx.x;
  while (left >= (r + 1) * PER) { left -= (r + 1) * PER; ++r; }
  if (r * TM >= it.X.n || left * TN >= it.X.n) return;
  gram_fast_body<DIMP, OP>(it.fp, it.X, it.X, lower_only, it.out, ld, it.diag_add, it.nan_flag, 0, 0, r, left);
}

// Does the program have one of the fast-path shapes?
static bool match_fast(const DevProgram &H, FastParams *fp, int *op) {
  const agp_kernel_node *n = H.nodes;
  if (H.n_nodes < 1 || n[0].op > AGP_OP_MATERN52 || n[0].metric != AGP_METRIC_EUCLIDEAN) return false;
  *op = n[0].op;
  fp->length_scale = n[0].params[0];
  fp->sigma = n[0].params[1];
  fp->sigma2 = fp->sigma * fp->sigma;
  fp->inv_l2 = fp->length_scale > 0. ? 1. / (fp->length_scale * fp->length_scale) : 0.;
  fp->cq = fp->length_scale > 0. ? (n[0].op == AGP_OP_MATERN32 ? sqrt(3.) : (n[0].op == AGP_OP_MATERN52 ? sqrt(5.) : 1.)) / fp->length_scale : 0.;
  fp->noise_var = 0.; fp->has_noise = 0; fp->noise_meas_only = 0;
  if (H.n_nodes == 1) return true;
  const bool is_noise = n[1].op == AGP_OP_INDEPENDENT_NOISE || n[1].op == AGP_OP_NUGGET;
  if (!is_noise) return false;
  if (H.n_nodes == 3 && n[2].op == AGP_OP_SUM) {
    fp->has_noise = 1; fp->noise_var = n[1].params[0] * n[1].params[0];
    return true;
  }
  if (H.n_nodes == 4 && n[2].op == AGP_OP_MEASUREMENT_ONLY && n[3].op == AGP_OP_SUM) {
    fp->has_noise = 1; fp->noise_meas_only = 1; fp->noise_var = n[1].params[0] * n[1].params[0];
    return true;
  }
  return false;
}

template <int DIMP>
static bool launch_gram_fast_t(hipStream_t s, const FastParams &fp, int op, const FeatView &X, const FeatView &Y,
                               bool lower_only, double *out, long long ld, const double *diag_add, int *nan_flag,
                               long long blk_rows = 0, long long blk_stride = 0, long long blk_count = 1) {
  const long long xn = blk_rows > 0 ? blk_rows : X.n, yn = blk_rows > 0 ? blk_rows : Y.n;
  dim3 grid((unsigned)((xn + TM - 1) / TM), (unsigned)((yn + TN - 1) / TN), (unsigned)blk_count), block(GRAM_THREADS);
  const int lo = lower_only ? 1 : 0;
  if (lower_only && blk_rows == 0 && blk_count == 1 && X.coords == Y.coords && X.ids == Y.ids && X.n == Y.n && X.meas == Y.meas) {
    const long long rt = (X.n + TM - 1) / TM;
    dim3 tri((unsigned)(rt * (rt + 1) / 2 * (TM / TN)));
    switch (op) {
    case AGP_OP_SQUARED_EXPONENTIAL:
      hipLaunchKernelGGL((gram_fast_tri_kernel<DIMP, AGP_OP_SQUARED_EXPONENTIAL>), tri, block, 0, s, fp, X, out, ld, diag_add, nan_flag);
      return true;
    case AGP_OP_EXPONENTIAL:
      hipLaunchKernelGGL((gram_fast_tri_kernel<DIMP, AGP_OP_EXPONENTIAL>), tri, block, 0, s, fp, X, out, ld, diag_add, nan_flag);
      return true;
    case AGP_OP_MATERN32:
      hipLaunchKernelGGL((gram_fast_tri_kernel<DIMP, AGP_OP_MATERN32>), tri, block, 0, s, fp, X, out, ld, diag_add, nan_flag);
      return true;
    case AGP_OP_MATERN52:
      hipLaunchKernelGGL((gram_fast_tri_kernel<DIMP, AGP_OP_MATERN52>), tri, block, 0, s, fp, X, out, ld, diag_add, nan_flag);
      return true;
    default: return false;
    }
  }
  switch (op) {
  case AGP_OP_SQUARED_EXPONENTIAL:
    hipLaunchKernelGGL((gram_fast_kernel<DIMP, AGP_OP_SQUARED_EXPONENTIAL>), grid, block, 0, s, fp, X, Y, lo, out, ld, diag_add, nan_flag, blk_rows, blk_stride);
    return true;
  case AGP_OP_EXPONENTIAL:
    hipLaunchKernelGGL((gram_fast_kernel<DIMP, AGP_OP_EXPONENTIAL>), grid, block, 0, s, fp, X, Y, lo, out, ld, diag_add, nan_flag, blk_rows, blk_stride);
    return true;
  case AGP_OP_MATERN32:
    hipLaunchKernelGGL((gram_fast_kernel<DIMP, AGP_OP_MATERN32>), grid, block, 0, s, fp, X, Y, lo, out, ld, diag_add, nan_flag, blk_rows, blk_stride);
    return true;
  case AGP_OP_MATERN52:
    hipLaunchKernelGGL((gram_fast_kernel<DIMP, AGP_OP_MATERN52>), grid, block, 0, s, fp, X, Y, lo, out, ld, diag_add, nan_flag, blk_rows, blk_stride);
    return true;
  default: return false;
  }
}

// Postfix program -> sum-of-products form, when the tree is one (cov_eval.h).  A host-side symbolic evaluation of the
// postfix program: every stack entry is a list of terms; SUM concatenates, PRODUCT joins two single-term operands,
// MEASUREMENT_ONLY flags the terms of its operand.
static bool build_sop(const DevProgram &H, SopProgram *out) {
  struct Expr { int n_terms; SopTerm t[SOP_MAX_TERMS]; };
  static thread_local Expr stack[AGP_MAX_STACK];
  int sp = 0;
  for (int i = 0; i < H.n_nodes; ++i) {
    const agp_kernel_node &nd = H.nodes[i];
    if (nd.op <= AGP_OP_SCALING) {
      if (sp >= AGP_MAX_STACK) return false;
      Expr &e = stack[sp++];
      e.n_terms = 1;
      SopTerm &t = e.t[0];
      t.n_factors = 1;
      t.measurement_only = 0;
      SopFactor &f = t.f[0];
      f.packed = SopFactor::pack(nd.op, nd.metric, nd.column, nd.order); f.pad = 0;
      f.a = f.b = f.c = f.d = f.e = 0.;
      if (nd.op <= AGP_OP_MATERN52) {
        const double l = nd.params[0], sg = nd.params[1];
        f.a = sg * sg;
        const double scale = nd.op == AGP_OP_MATERN32 ? sqrt(3.) : (nd.op == AGP_OP_MATERN52 ? sqrt(5.) : 1.);
        f.b = l > 0. ? scale / l : 0.;
      } else if (nd.op == AGP_OP_CONSTANT || nd.op == AGP_OP_INDEPENDENT_NOISE || nd.op == AGP_OP_NUGGET) {
        f.a = nd.params[0] * nd.params[0];
      } else if (nd.op == AGP_OP_POLYNOMIAL) {
        f.a = nd.params[0]; f.c = nd.params[1]; f.d = nd.params[2]; f.e = nd.params[3];
      }
    } else if (nd.op == AGP_OP_SUM) {
      if (sp < 2) return false;
      Expr &l = stack[sp - 2], &r = stack[sp - 1];
      if (l.n_terms + r.n_terms > SOP_MAX_TERMS) return false;
      for (int k = 0; k < r.n_terms; ++k) l.t[l.n_terms + k] = r.t[k];
      l.n_terms += r.n_terms;
      --sp;
    } else if (nd.op == AGP_OP_PRODUCT) {
      if (sp < 2) return false;
      Expr &l = stack[sp - 2], &r = stack[sp - 1];
      if (l.n_terms != 1 || r.n_terms != 1) return false;  // a sum inside a product: not a sum of products
      SopTerm &lt = l.t[0];
      const SopTerm &rt = r.t[0];
      if (lt.n_factors + rt.n_factors > SOP_MAX_FACTORS) return false;
      for (int k = 0; k < rt.n_factors; ++k) lt.f[lt.n_factors + k] = rt.f[k];
      lt.n_factors += rt.n_factors;
      lt.measurement_only = lt.measurement_only || rt.measurement_only;  // a zero factor zeroes the product
      --sp;
    } else if (nd.op == AGP_OP_MEASUREMENT_ONLY) {
      if (sp < 1) return false;
      Expr &e = stack[sp - 1];
      for (int k = 0; k < e.n_terms; ++k) e.t[k].measurement_only = 1;
    } else {
      return false;  // AGP_OP_TYPE_PAIR: defined / undefined bookkeeping stays with the interpreter
    }
  }
  if (sp != 1) return false;
  std::memset(out, 0, sizeof(*out));
  out->n_terms = stack[0].n_terms;
  out->metric_mask = H.metric_mask;
  out->uses_equality = H.uses_equality;
  for (int k = 0; k < stack[0].n_terms; ++k) out->t[k] = stack[0].t[k];
  return true;
}

bool gram_sop_enabled();  // api.hip: AGP_GRAM_SOP of the context created last (0: always the postfix interpreter)
static bool sop_enabled() { return gram_sop_enabled(); }

// ---------------------------------------------------------------------------------------------------------------
// Second fast path: sums of AT MOST three terms of fixed kinds,
//     [ScalingTerm * Constant]  +  [IndependentNoise | Nugget]  +  radial<metric A> [* radial<metric B>]
// with any of the four radial kernels over any of the three metrics, each term optionally measurement-only - the
// temperature example's tree (ScalingTerm<Elevation> * Constant + IndependentNoise + Exponential<Angular> *
// SquaredExponential<Radial>, examples/temperature_example/temperature_example.cc:34-85), or one radial leaf over a
// non-Euclidean metric.  Same arithmetic as the sum-of-products evaluator (cov_eval.h: eval_sop_n - distances, factor
// order, one exp per product, the `lhs != 0` short circuit, the order of the sum), but the shape is fixed: no term /
// factor loop, no scalar loads of the program per pair, and only the point data the tree needs goes through LDS.
// ---------------------------------------------------------------------------------------------------------------
struct Pair2Params {
  int n_rad;                 // radial leaves in the product (1 or 2)
  int op[2], metric[2];
  double a[2], b[2];         // sigma^2 and scale / l of each leaf (b = 0: the leaf is 0, radial.hpp:26-28)
  int rad_meas_only;
  int has_rank1;             // a term made of ScalingTerm and / or Constant factors
  int rank1_scaling, rank1_const, rank1_const_first, rank1_col, rank1_meas_only;
  double rank1_c2;
  int has_noise, noise_meas_only;
  double noise_var;
  int order[3];              // kind of the 1st, 2nd, 3rd term of the sum: 0 radial, 1 rank-1, 2 noise, -1 none
  int metric_mask;
};

__device__ __forceinline__ void pair2_leaf(int op, double q, double a, double b, double &v, double &expo) {
  double coef = a;
  if (op == AGP_OP_SQUARED_EXPONENTIAL) expo += q * q;
  else if (op == AGP_OP_EXPONENTIAL) expo += fabs(q);
  else if (op == AGP_OP_MATERN32) { coef = coef * (1 + q); expo += q; }
  else { coef = coef * (1 + q + q * q * (1. / 3.)); expo += q; }
  const double f = (b > 0.) ? coef : 0.;
  v = (v != 0.) ? v * f : v;
}

template <int DIMP, bool EUCLID, bool ANGULAR>
__global__ __launch_bounds__(GRAM_THREADS) void gram_pair2_kernel(Pair2Params pp, FeatView X, FeatView Y, int lower_only,
                                                                  double *out, long long ld, const double *diag_add,
                                                                  int *nan_flag) {
  __shared__ double xs[DIMP][TM], ys[DIMP][TN], xnorm[TM], ynorm[TN], xsc[TM], ysc[TN];
  __shared__ long long xid[TM], yid[TN];
  const long long row0 = (long long)blockIdx.x * TM;
  const long long col0 = (long long)blockIdx.y * TN;
  if (lower_only && col0 > row0 + TM - 1) return;
  const bool have_ids = X.ids != nullptr && Y.ids != nullptr;
  const bool need_norm = (pp.metric_mask & ((1 << AGP_METRIC_RADIAL) | (1 << AGP_METRIC_ANGULAR))) != 0;
  for (int t = threadIdx.x; t < TM + TN; t += GRAM_THREADS) {
    const bool isx = t < TM;
    const FeatView &F = isx ? X : Y;
    const long long g = isx ? row0 + t : col0 + (t - TM);
    const bool ok = g < F.n;
    double nn = 0.;
#pragma unroll
    for (int d = 0; d < DIMP; ++d) {
      const double v = (ok && d < F.dim) ? F.coords[g * F.dim + d] : 0.;
      nn += v * v;
      if (isx) xs[d][t] = v; else ys[d][t - TM] = v;
    }
    const double nrm = need_norm ? sqrt(nn) : 0.;
    const double sc = (ok && pp.rank1_scaling && pp.rank1_col < F.nsc) ? F.scales[(long long)pp.rank1_col * scale_stride(F) + g] : 0.;
    const long long id = (ok && F.ids) ? F.ids[g] : -1;
    if (isx) { xnorm[t] = nrm; xsc[t] = sc; xid[t] = id; }
    else { ynorm[t - TM] = nrm; ysc[t - TM] = sc; yid[t - TM] = id; }
  }
  __syncthreads();
  const int lane_row = 2 * (threadIdx.x & 63);
  const int cgrp = threadIdx.x >> 6;
  double xa[DIMP], xb[DIMP];
#pragma unroll
  for (int d = 0; d < DIMP; ++d) { xa[d] = xs[d][lane_row]; xb[d] = xs[d][lane_row + 1]; }
  const double na = xnorm[lane_row], nb = xnorm[lane_row + 1];
  const double sca = xsc[lane_row], scb = xsc[lane_row + 1];
  const long long ida = xid[lane_row], idb = xid[lane_row + 1];
  const long long ra = row0 + lane_row, rb = ra + 1;
  const bool wide = ((ld & 1) == 0) && ((reinterpret_cast<uintptr_t>(out) & 15) == 0);
  const bool both_meas = X.meas && Y.meas;
  const bool rad_on = !pp.rad_meas_only || both_meas;
  const bool rank1_on = pp.has_rank1 && (!pp.rank1_meas_only || both_meas);
  const bool noise_on = pp.has_noise && (!pp.noise_meas_only || both_meas);
  const bool use_radial = (pp.metric_mask & (1 << AGP_METRIC_RADIAL)) != 0;
  const bool need_eq = pp.has_noise && !have_ids;  // equality of all coordinates by value (noise.hpp:37-43)
  bool saw_nan = false;
  for (int jj = 0; jj < TN / 4; ++jj) {
    const int cslot = cgrp * (TN / 4) + jj;
    const long long col = col0 + cslot;
    if (col >= Y.n) break;
    const double ny = ynorm[cslot];
    double val[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const double *xp = h ? xb : xa;
      const double nx = h ? nb : na;
      double d_e = 0., d_r = 0., d_a = 0.;
      bool eq = true;
      double ssq = 0., dot = 0.;
#pragma unroll
      for (int d = 0; d < DIMP; ++d) {
        const double yd = ys[d][cslot];
        if (EUCLID) {
          const double t = xp[d] - yd;
          ssq += t * t;
        }
        if (ANGULAR) dot += xp[d] * yd;
        if (need_eq) eq = eq && (xp[d] == yd);
      }
      if (EUCLID) d_e = (DIMP == 1) ? fabs(xp[0] - ys[0][cslot]) : sqrt(ssq);
      if (use_radial) d_r = fabs(nx - ny);
      if (ANGULAR) {
        const double c = dot / (nx * ny);
        const double eps = 1e-16;  // EPSILON, distance_metrics.hpp:18
        d_a = (c > 1. - eps) ? 0. : ((c < -1. + eps) ? M_PI : acos_fast(c));
      }
      if (have_ids) eq = (h ? idb : ida) == yid[cslot];
      // the terms, each exactly as eval_sop_n forms it
      double t_rad = 0., t_rank1 = 0., t_noise = 0.;
      if (rad_on) {
        double v = 1., expo = 0.;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i < pp.n_rad) {
            const double dist = pp.metric[i] == AGP_METRIC_EUCLIDEAN ? d_e : (pp.metric[i] == AGP_METRIC_RADIAL ? d_r : d_a);
            pair2_leaf(pp.op[i], dist * pp.b[i], pp.a[i], pp.b[i], v, expo);
          }
        }
        t_rad = (v != 0.) ? v * exp_neg(expo) : v;
      }
      if (rank1_on) {
        double v = 1.;
        const double ff = (h ? scb : sca) * ysc[cslot];
        if (pp.rank1_const && pp.rank1_const_first) v = v * pp.rank1_c2;
        if (pp.rank1_scaling) v = (v != 0.) ? v * ff : v;
        if (pp.rank1_const && !pp.rank1_const_first) v = (v != 0.) ? v * pp.rank1_c2 : v;
        t_rank1 = v;
      }
      if (noise_on) t_noise = eq ? pp.noise_var : 0.;
      double acc = 0.;
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int kind = pp.order[k];
        // (a measurement-only term that is off is skipped by eval_sop_n: adding its 0 here gives the same sum)
        if (kind == 0) acc += t_rad;
        else if (kind == 1) acc += t_rank1;
        else if (kind == 2) acc += t_noise;
      }
      val[h] = acc;
    }
    double va = val[0], vb = val[1];
    if (diag_add) {
      if (ra == col) va += diag_add[col];
      if (rb == col) vb += diag_add[col];
    }
    saw_nan = saw_nan || (ra < X.n && va != va) || (rb < X.n && vb != vb);
    double *dst = out + col * ld + ra;
    if (rb < X.n) {
      if (wide) *reinterpret_cast<double2 *>(dst) = make_double2(va, vb);
      else { dst[0] = va; dst[1] = vb; }
    } else if (ra < X.n) {
      dst[0] = va;
    }
  }
  if (saw_nan && nan_flag) atomicOr(nan_flag, 1);
}

// Is the sum-of-products form of the tree one of the shapes above?
static bool match_pair2(const SopProgram &S, Pair2Params *pp) {
  std::memset(pp, 0, sizeof(*pp));
  pp->order[0] = pp->order[1] = pp->order[2] = -1;
  if (S.n_terms < 1 || S.n_terms > 3) return false;
  bool have_rad = false;
  for (int ti = 0; ti < S.n_terms; ++ti) {
    const SopTerm &T = S.t[ti];
    const int op0 = T.f[0].packed & 0xff;
    if (op0 <= AGP_OP_MATERN52) {
      if (have_rad || T.n_factors > 2) return false;
      for (int fi = 0; fi < T.n_factors; ++fi) {
        const SopFactor &F = T.f[fi];
        const int op = F.packed & 0xff;
        if (op > AGP_OP_MATERN52) return false;
        pp->op[fi] = op;
        pp->metric[fi] = (F.packed >> 8) & 0xff;
        pp->a[fi] = F.a;
        pp->b[fi] = F.b;
      }
      pp->n_rad = T.n_factors;
      pp->rad_meas_only = T.measurement_only;
      have_rad = true;
      pp->order[ti] = 0;
    } else if (op0 == AGP_OP_INDEPENDENT_NOISE || op0 == AGP_OP_NUGGET) {
      if (pp->has_noise || T.n_factors != 1) return false;
      pp->has_noise = 1;
      pp->noise_var = T.f[0].a;
      pp->noise_meas_only = T.measurement_only;
      pp->order[ti] = 2;
    } else if (op0 == AGP_OP_SCALING || op0 == AGP_OP_CONSTANT) {
      if (pp->has_rank1 || T.n_factors > 2) return false;
      for (int fi = 0; fi < T.n_factors; ++fi) {
        const SopFactor &F = T.f[fi];
        const int op = F.packed & 0xff;
        if (op == AGP_OP_SCALING) {
          if (pp->rank1_scaling) return false;
          pp->rank1_scaling = 1;
          pp->rank1_col = (F.packed >> 16) & 0xff;
        } else if (op == AGP_OP_CONSTANT) {
          if (pp->rank1_const) return false;
          pp->rank1_const = 1;
          pp->rank1_c2 = F.a;
          pp->rank1_const_first = fi == 0;
        } else {
          return false;
        }
      }
      pp->has_rank1 = 1;
      pp->rank1_meas_only = T.measurement_only;
      pp->order[ti] = 1;
    } else {
      return false;
    }
  }
  if (!have_rad) return false;
  pp->metric_mask = 0;
  for (int i = 0; i < pp->n_rad; ++i) pp->metric_mask |= 1 << pp->metric[i];
  return true;
}

template <int DIMP>
static void launch_gram_t(hipStream_t s, const DevProgram *P, const FeatView &X, const FeatView &Y,
                          bool symmetric, bool lower_only, double *out, long long ld,
                          const double *diag_add, int *nan_flag, const DevProgram *host_program) {
  dim3 grid((unsigned)((X.n + TM - 1) / TM), (unsigned)((Y.n + TN - 1) / TN));
  SopProgram sop;
  if (host_program && sop_enabled() && build_sop(*host_program, &sop)) {
    hipLaunchKernelGGL((gram_kernel<DIMP, true>), grid, dim3(GRAM_THREADS), 0, s, P, sop, X, Y, symmetric ? 1 : 0,
                       lower_only ? 1 : 0, out, ld, diag_add, nan_flag);
    return;
  }
  std::memset(&sop, 0, sizeof(sop));
  hipLaunchKernelGGL((gram_kernel<DIMP, false>), grid, dim3(GRAM_THREADS), 0, s, P, sop, X, Y, symmetric ? 1 : 0,
                     lower_only ? 1 : 0, out, ld, diag_add, nan_flag);
}

// `count` diagonal blocks of `rows` consecutive features each - block g = the symmetric Gram matrix of features
// [g rows, (g + 1) rows), lower tiles, written to out + g * stride (leading dimension ld), diag_add likewise - in ONE launch
// (the A blocks of a sparse fit with equal groups: 512 launches of a few microseconds each were host-bound).  Only for the
// radial fast-path shapes; returns false otherwise (the caller launches block by block).
bool launch_gram_blocks(hipStream_t s, const DevProgram *host_program, const FeatView &X, long long rows, long long count,
                        double *out, long long ld, long long stride, const double *diag_add, int *nan_flag) {
  if (!host_program || X.dim > 3 || rows <= 0 || count <= 0 || count > 65535 || !sop_enabled()) return false;
  FastParams fp;
  int op = 0;
  if (!match_fast(*host_program, &fp, &op)) return false;
  if (X.dim == 1) return launch_gram_fast_t<1>(s, fp, op, X, X, true, out, ld, diag_add, nan_flag, rows, stride, count);
  if (X.dim == 2) return launch_gram_fast_t<2>(s, fp, op, X, X, true, out, ld, diag_add, nan_flag, rows, stride, count);
  return launch_gram_fast_t<3>(s, fp, op, X, X, true, out, ld, diag_add, nan_flag, rows, stride, count);
}

size_t gram_batch_table_bytes(long long count) { return sizeof(GramBatchItem) * (size_t)(count > 0 ? count : 0); }

// Symmetric (lower-only) Gram matrices of `count` problems in one launch when all of them take the same fast path
// (radial<Euclidean> [+ noise], dim <= 3, same operator): false = not applicable, nothing launched.
// table_dev: gram_batch_table_bytes(count) bytes of device scratch the launch reads (must stay valid until it has run).
bool launch_gram_batch(hipStream_t s, long long count, const DevProgram *const *host_programs, const FeatView *Xs, double *const *outs,
                       long long ld, const double *const *diag_adds, int *const *nan_flags, void *table_dev, void *host_stage) {
  if (count <= 0 || count > 65535 || !table_dev || !sop_enabled()) return false;
  std::vector<GramBatchItem> pageable;
  GramBatchItem *items = static_cast<GramBatchItem *>(host_stage);
  if (!items) {
    pageable.resize((size_t)count);
    items = pageable.data();
  }
  int op0 = 0, dim0 = 0;
  long long nmax = 0;
  for (long long b = 0; b < count; ++b) {
    GramBatchItem &it = items[(size_t)b];
    int op = 0;
    if (!host_programs[b] || Xs[b].dim > 3 || !match_fast(*host_programs[b], &it.fp, &op)) return false;
    if (b == 0) { op0 = op; dim0 = Xs[b].dim; }
    else if (op != op0 || Xs[b].dim != dim0) return false;
    it.X = Xs[b];
    it.out = outs[b];
    it.diag_add = diag_adds ? diag_adds[b] : nullptr;
    it.nan_flag = nan_flags ? nan_flags[b] : nullptr;
    if (Xs[b].n > nmax) nmax = Xs[b].n;
  }
  if (nmax <= 0) return true;
  if (hipMemcpyAsync(table_dev, items, sizeof(GramBatchItem) * (size_t)count, hipMemcpyHostToDevice, s) != hipSuccess) {
    (void)hipGetLastError();
    return false;
  }
  if (!host_stage) (void)hipStreamSynchronize(s);  // (pageable source: the vector goes out of scope; a pinned stage outlives the launch)
  const long long rt = (nmax + TM - 1) / TM;  // row tiles; the lower tiles of row tile r: (r + 1) TM / TN
  dim3 grid((unsigned)(rt * (rt + 1) / 2 * (TM / TN)), 1u, (unsigned)count), block(GRAM_THREADS);
  auto *tab = static_cast<const GramBatchItem *>(table_dev);
#define AGP_GB(D, O) hipLaunchKernelGGL((gram_fast_batch_kernel<D, O>), grid, block, 0, s, tab, 1, ld)
#define AGP_GB_DIM(D)                                                        \
  switch (op0) {                                                             \
  case AGP_OP_SQUARED_EXPONENTIAL: AGP_GB(D, AGP_OP_SQUARED_EXPONENTIAL); break; \
  case AGP_OP_EXPONENTIAL: AGP_GB(D, AGP_OP_EXPONENTIAL); break;             \
  case AGP_OP_MATERN32: AGP_GB(D, AGP_OP_MATERN32); break;                   \
  default: AGP_GB(D, AGP_OP_MATERN52); break;                                \
  }
  if (dim0 == 1) { AGP_GB_DIM(1) }
  else if (dim0 == 2) { AGP_GB_DIM(2) }
  else { AGP_GB_DIM(3) }
#undef AGP_GB_DIM
#undef AGP_GB
  return true;
}

void launch_gram(hipStream_t s, const DevProgram *P, const FeatView &X, const FeatView &Y, bool symmetric,
                 bool lower_only, double *out, long long ld, const double *diag_add, int *nan_flag,
                 const DevProgram *host_program) {
  if (X.n == 0 || Y.n == 0) return;
  const int dim = X.dim;
  if (host_program && dim <= 3) {
    // (the fast shapes are bitwise symmetric in their arguments, so `symmetric`
    // needs no special handling)
    FastParams fp;
    int op = 0;
    if (sop_enabled() && match_fast(*host_program, &fp, &op)) {
      bool done = false;
      if (dim == 1) done = launch_gram_fast_t<1>(s, fp, op, X, Y, lower_only, out, ld, diag_add, nan_flag);
      else if (dim == 2) done = launch_gram_fast_t<2>(s, fp, op, X, Y, lower_only, out, ld, diag_add, nan_flag);
      else done = launch_gram_fast_t<3>(s, fp, op, X, Y, lower_only, out, ld, diag_add, nan_flag);
      if (done) return;
    }
  }
  if (host_program && dim <= 3 && sop_enabled()) {
    // (every term of these shapes is bitwise symmetric in its arguments too)
    SopProgram sop;
    Pair2Params pp;
    if (build_sop(*host_program, &sop) && match_pair2(sop, &pp)) {
      dim3 grid((unsigned)((X.n + TM - 1) / TM), (unsigned)((Y.n + TN - 1) / TN)), block(GRAM_THREADS);
      const int lo = lower_only ? 1 : 0;
      const bool eu = (pp.metric_mask & (1 << AGP_METRIC_EUCLIDEAN)) != 0, an = (pp.metric_mask & (1 << AGP_METRIC_ANGULAR)) != 0;
#define AGP_P2(D, E, A) hipLaunchKernelGGL((gram_pair2_kernel<D, E, A>), grid, block, 0, s, pp, X, Y, lo, out, ld, diag_add, nan_flag)
#define AGP_P2_DIM(D)                                   \
  do {                                                  \
    if (eu && an) AGP_P2(D, true, true);                \
    else if (eu) AGP_P2(D, true, false);                \
    else if (an) AGP_P2(D, false, true);                \
    else AGP_P2(D, false, false);                       \
  } while (0)
      if (dim == 1) AGP_P2_DIM(1);
      else if (dim == 2) AGP_P2_DIM(2);
      else AGP_P2_DIM(3);
#undef AGP_P2_DIM
#undef AGP_P2
      return;
    }
  }
  if (dim == 1) launch_gram_t<1>(s, P, X, Y, symmetric, lower_only, out, ld, diag_add, nan_flag, host_program);
  else if (dim == 2) launch_gram_t<2>(s, P, X, Y, symmetric, lower_only, out, ld, diag_add, nan_flag, host_program);
  else if (dim == 3) launch_gram_t<3>(s, P, X, Y, symmetric, lower_only, out, ld, diag_add, nan_flag, host_program);
  else if (dim == 4) launch_gram_t<4>(s, P, X, Y, symmetric, lower_only, out, ld, diag_add, nan_flag, host_program);
  else launch_gram_t<8>(s, P, X, Y, symmetric, lower_only, out, ld, diag_add, nan_flag, host_program);
}

// ---- diagonal: prior_variance[i] = cov(f_i, f_i)  (gp.hpp:339-343) -----------
template <int DIMP>
__global__ __launch_bounds__(256) void gram_diag_kernel(const DevProgram *__restrict__ P, FeatView X, double *out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= X.n) return;
  Point<DIMP> p;
  double nn = 0.;
#pragma unroll
  for (int d = 0; d < DIMP; ++d) {
    p.c[d] = d < X.dim ? X.coords[i * X.dim + d] : 0.;
    nn += p.c[d] * p.c[d];
  }
  p.norm = sqrt(nn);
#pragma unroll
  for (int k = 0; k < AGP_MAX_SCALE_COLUMNS; ++k) p.s[k] = k < X.nsc ? X.scales[(long long)k * scale_stride(X) + i] : 0.;
  p.id = X.ids ? X.ids[i] : -1;
  out[i] = eval_pair<DIMP>(P, p, p, false, X.ids != nullptr, X.meas != 0);
}

void launch_gram_diagonal(hipStream_t s, const DevProgram *P, const FeatView &X, double *out) {
  if (X.n == 0) return;
  dim3 grid((unsigned)((X.n + 255) / 256));
  const int dim = X.dim;
  if (dim == 1) hipLaunchKernelGGL(gram_diag_kernel<1>, grid, dim3(256), 0, s, P, X, out);
  else if (dim == 2) hipLaunchKernelGGL(gram_diag_kernel<2>, grid, dim3(256), 0, s, P, X, out);
  else if (dim == 3) hipLaunchKernelGGL(gram_diag_kernel<3>, grid, dim3(256), 0, s, P, X, out);
  else if (dim == 4) hipLaunchKernelGGL(gram_diag_kernel<4>, grid, dim3(256), 0, s, P, X, out);
  else hipLaunchKernelGGL(gram_diag_kernel<8>, grid, dim3(256), 0, s, P, X, out);
}

// ---- fused predictive mean: mean_j = sum_i k(x_i, xs_j) alpha_i ----------------
// gp_mean_prediction (gp.hpp:82-85) without materialising the N x M cross
// Gram: one workgroup per 4 test points... each wave owns one test point and
// strides over the training points; wave reduction at the end.
constexpr int PM_WAVES = 4;

template <int DIMP, bool SOP>
__global__ __launch_bounds__(64 * PM_WAVES) void predict_mean_kernel(const DevProgram *__restrict__ P, SopProgram sop, FeatView X,
                                                                      FeatView XS, const double *alpha, double *mean) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long j = (long long)blockIdx.x * PM_WAVES + wave;
  if (j >= XS.n) return;
  const int metric_mask = SOP ? sop.metric_mask : P->metric_mask;
  const bool need_norm = (metric_mask & ((1 << AGP_METRIC_RADIAL) | (1 << AGP_METRIC_ANGULAR))) != 0;
  Point<DIMP> y;
  double nn = 0.;
#pragma unroll
  for (int d = 0; d < DIMP; ++d) {
    y.c[d] = d < XS.dim ? XS.coords[j * XS.dim + d] : 0.;
    nn += y.c[d] * y.c[d];
  }
  y.norm = need_norm ? sqrt(nn) : 0.;
#pragma unroll
  for (int k = 0; k < AGP_MAX_SCALE_COLUMNS; ++k) y.s[k] = k < XS.nsc ? XS.scales[(long long)k * scale_stride(XS) + j] : 0.;
  y.id = XS.ids ? XS.ids[j] : -1;
  const bool have_ids = X.ids != nullptr && XS.ids != nullptr;
  const bool both_meas = X.meas && XS.meas;
  double acc = 0.;
  auto load_x = [&](long long i) {
    Point<DIMP> x;
    double xn = 0.;
#pragma unroll
    for (int d = 0; d < DIMP; ++d) {
      x.c[d] = d < X.dim ? X.coords[i * X.dim + d] : 0.;
      xn += x.c[d] * x.c[d];
    }
    x.norm = need_norm ? sqrt(xn) : 0.;
#pragma unroll
    for (int k = 0; k < AGP_MAX_SCALE_COLUMNS; ++k) x.s[k] = k < X.nsc ? X.scales[(long long)k * scale_stride(X) + i] : 0.;
    x.id = X.ids ? X.ids[i] : -1;
    return x;
  };
  long long i = lane;
  if (SOP) {  // two training points per walk of the program (cov_eval.h: eval_sop_n)
    for (; i + 64 < X.n; i += 128) {
      const Point<DIMP> x0 = load_x(i), x1 = load_x(i + 64);
      const Point<DIMP> *const xs2[2] = {&x0, &x1};
      const Point<DIMP> *const ys2[2] = {&y, &y};
      const bool sw2[2] = {false, false};
      double v2[2];
      eval_sop_n<DIMP, 2>(sop, xs2, ys2, sw2, have_ids, both_meas, v2);
      acc += v2[0] * alpha[i];
      acc += v2[1] * alpha[i + 64];
    }
  }
  for (; i < X.n; i += 64) {
    const Point<DIMP> x = load_x(i);
    acc += (SOP ? eval_sop<DIMP>(sop, x, y, false, have_ids, both_meas) : eval_pair<DIMP>(P, x, y, false, have_ids, both_meas)) * alpha[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) mean[j] = acc;
}

// fast-path twin of predict_mean_kernel for radial<Euclidean> [+ noise] trees:
// 8 test points per workgroup... one wave per test point, training points
// strided over the lanes, coordinates read as row-major triples.
template <int DIMP, int OP>
__global__ __launch_bounds__(64 * PM_WAVES) void predict_mean_fast_kernel(FastParams fp, FeatView X, FeatView XS,
                                                                           const double *alpha, double *mean) {
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const long long j = (long long)blockIdx.x * PM_WAVES + wave;
  if (j >= XS.n) return;
  double y[DIMP];
#pragma unroll
  for (int d = 0; d < DIMP; ++d) y[d] = XS.coords[j * XS.dim + d];
  const bool have_ids = X.ids != nullptr && XS.ids != nullptr;
  const long long yid = have_ids ? XS.ids[j] : -1;
  const bool noise_on = fp.has_noise && (!fp.noise_meas_only || (X.meas && XS.meas));
  double acc = 0.;
  for (long long i = lane; i < X.n; i += 64) {
    bool eq = true;
    double s = 0.;
#pragma unroll
    for (int d = 0; d < DIMP; ++d) {
      const double xd = X.coords[i * DIMP + d];
      const double t = xd - y[d];
      s += t * t;
      eq = eq && (xd == y[d]);
    }
    if (have_ids) eq = X.ids[i] == yid;
    double v = radial_fast<OP>(s, fp);
    if (fp.has_noise) v = v + ((noise_on && eq) ? fp.noise_var : 0.);
    acc += v * alpha[i];
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if (lane == 0) mean[j] = acc;
}

// The same for MANY test points (M >= 1024): a workgroup of 256 threads owns TJ test points - their coordinates are
// wave-uniform, i.e. scalar registers - and every thread walks the training points tid, tid + 256, ... loading each one
// (and its information entry) ONCE for TJ kernel evaluations; TJ independent exp_neg chains per iteration fill
// the latency of one another.  One shuffle reduction per test point at the end.  (The one-wave-per-test-point kernel
// above loads three coordinates and one information entry per single evaluation: 0.145 ms for M = 4096 at N = 16384,
// a quarter of the VALU roofline.)  Deterministic; the summation order differs from the kernel above (both hold the
// 1e-8 parity bar with orders of magnitude to spare).
// TJ = 4 test points per workgroup, 8 from 16384 test points on (half the training loads per evaluation; below that the
// launch has too few workgroups for it: 0.185 against 0.166 ms at M = 4096).  The next training point's coordinates are
// requested before the current one is evaluated.
template <int DIMP, int OP, int TJ>
__global__ __launch_bounds__(256) void predict_mean_tiled_kernel(FastParams fp, FeatView X, FeatView XS, const double *alpha,
                                                                double *mean) {
  const long long j0 = (long long)blockIdx.x * TJ;
  double y[TJ][DIMP];
  long long yid[TJ];
  const bool have_ids = X.ids != nullptr && XS.ids != nullptr;
#pragma unroll
  for (int t = 0; t < TJ; ++t) {
    const long long j = j0 + t < XS.n ? j0 + t : XS.n - 1;  // (a partial last tile repeats the last point; not stored)
#pragma unroll
    for (int d = 0; d < DIMP; ++d) y[t][d] = XS.coords[j * XS.dim + d];
    yid[t] = have_ids ? XS.ids[j] : -1;
  }
  const bool noise_on = fp.has_noise && (!fp.noise_meas_only || (X.meas && XS.meas));
  double acc[TJ];
#pragma unroll
  for (int t = 0; t < TJ; ++t) acc[t] = 0.;
  double xn[DIMP], an = 0.;
  long long idn = 0;
  {
    const long long i = threadIdx.x < X.n ? threadIdx.x : X.n - 1;
#pragma unroll
    for (int d = 0; d < DIMP; ++d) xn[d] = X.coords[i * DIMP + d];
    an = alpha[i];
    idn = have_ids ? X.ids[i] : 0;
  }
  for (long long i = threadIdx.x; i < X.n; i += 256) {
    double x[DIMP];
#pragma unroll
    for (int d = 0; d < DIMP; ++d) x[d] = xn[d];
    const double a = an;
    const long long xid = idn;
    {
      const long long in = i + 256 < X.n ? i + 256 : i;  // (the last trip re-reads its own point)
#pragma unroll
      for (int d = 0; d < DIMP; ++d) xn[d] = X.coords[in * DIMP + d];
      an = alpha[in];
      idn = have_ids ? X.ids[in] : 0;
    }
    // All TJ evaluations in ONE basic block: the exp_neg chains (13 dependent v_fma_f64 each) interleave, and a wave hides
    // its own latencies instead of leaning on the other three of its SIMD.  (Round 5's loop body branched per evaluation -
    // on the length scale's sign and, since round 6, on the noise term - which kept the chains one behind the other: the
    // kernel sat at 60 % of the fp64 issue rate.)  Same operations per evaluation: bit-identical results.
    double sq[TJ], c[TJ];
#pragma unroll
    for (int t = 0; t < TJ; ++t) {
      sq[t] = 0.;
#pragma unroll
      for (int d = 0; d < DIMP; ++d) {
        const double dd = x[d] - y[t][d];
        sq[t] += dd * dd;
      }
    }
    double v[TJ];
    radial_fast_n<OP, TJ>(sq, fp, v);  // (the launcher sends a non-positive length scale to the other kernel)
#pragma unroll
    for (int t = 0; t < TJ; ++t) c[t] = v[t] * a;
    // the noise term (noise.hpp:37-43: sigma^2 iff x == y) only where a training point IS a test point: equal coordinates
    // give sq == 0 exactly, so the per-coordinate comparison (three v_cmp_f64 + the selects, a sixth of the instructions of
    // an evaluation) runs only in a wave that holds such a pair
    if (noise_on) {
      bool maybe = false;
#pragma unroll
      for (int t = 0; t < TJ; ++t) maybe = maybe || (have_ids ? (xid == yid[t]) : (sq[t] == 0.));
      if (__any(maybe)) {
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
          bool eq = true;
#pragma unroll
          for (int d = 0; d < DIMP; ++d) eq = eq && (x[d] == y[t][d]);
          if (have_ids) eq = xid == yid[t];
          if (eq) c[t] = (v[t] + fp.noise_var) * a;  // (lhs + rhs first, like the reference's SumOfCovarianceFunctions: the value round 5 computed)
        }
      }
    }
#pragma unroll
    for (int t = 0; t < TJ; ++t) acc[t] += c[t];
  }
  __shared__ double red[4][TJ];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int t = 0; t < TJ; ++t) {
    double r = acc[t];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) r += __shfl_down(r, off, 64);
    if (lane == 0) red[wave][t] = r;
  }
  __syncthreads();
  if (threadIdx.x < TJ && j0 + threadIdx.x < XS.n)
    mean[j0 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

template <int DIMP, int TJ>
static bool launch_predict_mean_tiled_t(hipStream_t s, const FastParams &fp, int op, const FeatView &X, const FeatView &XS,
                                        const double *alpha, double *mean) {
  dim3 tgrid((unsigned)((XS.n + TJ - 1) / TJ)), tblock(256);
  switch (op) {
  case AGP_OP_SQUARED_EXPONENTIAL:
    hipLaunchKernelGGL((predict_mean_tiled_kernel<DIMP, AGP_OP_SQUARED_EXPONENTIAL, TJ>), tgrid, tblock, 0, s, fp, X, XS, alpha, mean);
    return true;
  case AGP_OP_EXPONENTIAL:
    hipLaunchKernelGGL((predict_mean_tiled_kernel<DIMP, AGP_OP_EXPONENTIAL, TJ>), tgrid, tblock, 0, s, fp, X, XS, alpha, mean);
    return true;
  case AGP_OP_MATERN32:
    hipLaunchKernelGGL((predict_mean_tiled_kernel<DIMP, AGP_OP_MATERN32, TJ>), tgrid, tblock, 0, s, fp, X, XS, alpha, mean);
    return true;
  case AGP_OP_MATERN52:
    hipLaunchKernelGGL((predict_mean_tiled_kernel<DIMP, AGP_OP_MATERN52, TJ>), tgrid, tblock, 0, s, fp, X, XS, alpha, mean);
    return true;
  default: return false;
  }
}

template <int DIMP>
static bool launch_predict_mean_fast_t(hipStream_t s, const FastParams &fp, int op, const FeatView &X,
                                       const FeatView &XS, const double *alpha, double *mean) {
  // (radial.hpp: the covariance is 0 for a non-positive length scale - the tiled kernel has no test for it in its loop)
  if (XS.n >= 1024 && XS.dim == DIMP && fp.length_scale > 0.) {  // enough test points for 256 workgroups of four (16384: of eight)
    // (TJ = 2 below 8192 test points - twice the workgroups - measured in round 6: +6 % at M = 2048, 0 at 4096, -4 % at 8192)
    const bool done = XS.n >= 16384 ? launch_predict_mean_tiled_t<DIMP, 8>(s, fp, op, X, XS, alpha, mean)
                                    : launch_predict_mean_tiled_t<DIMP, 4>(s, fp, op, X, XS, alpha, mean);
    if (done) return true;
  }
  dim3 grid((unsigned)((XS.n + PM_WAVES - 1) / PM_WAVES)), block(64 * PM_WAVES);
  switch (op) {
  case AGP_OP_SQUARED_EXPONENTIAL:
    hipLaunchKernelGGL((predict_mean_fast_kernel<DIMP, AGP_OP_SQUARED_EXPONENTIAL>), grid, block, 0, s, fp, X, XS, alpha, mean);
    return true;
  case AGP_OP_EXPONENTIAL:
    hipLaunchKernelGGL((predict_mean_fast_kernel<DIMP, AGP_OP_EXPONENTIAL>), grid, block, 0, s, fp, X, XS, alpha, mean);
    return true;
  case AGP_OP_MATERN32:
    hipLaunchKernelGGL((predict_mean_fast_kernel<DIMP, AGP_OP_MATERN32>), grid, block, 0, s, fp, X, XS, alpha, mean);
    return true;
  case AGP_OP_MATERN52:
    hipLaunchKernelGGL((predict_mean_fast_kernel<DIMP, AGP_OP_MATERN52>), grid, block, 0, s, fp, X, XS, alpha, mean);
    return true;
  default: return false;
  }
}

void launch_predict_mean(hipStream_t s, const DevProgram *P, const FeatView &X, const FeatView &XS,
                         const double *alpha, double *mean, const DevProgram *host_program) {
  if (XS.n == 0) return;
  if (host_program && X.dim <= 3 && X.dim == XS.dim) {
    FastParams fp;
    int op = 0;
    if (sop_enabled() && match_fast(*host_program, &fp, &op)) {
      bool done = false;
      if (X.dim == 1) done = launch_predict_mean_fast_t<1>(s, fp, op, X, XS, alpha, mean);
      else if (X.dim == 2) done = launch_predict_mean_fast_t<2>(s, fp, op, X, XS, alpha, mean);
      else done = launch_predict_mean_fast_t<3>(s, fp, op, X, XS, alpha, mean);
      if (done) return;
    }
  }
  dim3 grid((unsigned)((XS.n + PM_WAVES - 1) / PM_WAVES)), block(64 * PM_WAVES);
  const int dim = X.dim;
  SopProgram sop;
  const bool use_sop = host_program && sop_enabled() && build_sop(*host_program, &sop);
  if (!use_sop) std::memset(&sop, 0, sizeof(sop));
#define AGP_PM_LAUNCH(D)                                                                                                      \
  do {                                                                                                                       \
    if (use_sop) hipLaunchKernelGGL((predict_mean_kernel<D, true>), grid, block, 0, s, P, sop, X, XS, alpha, mean);           \
    else hipLaunchKernelGGL((predict_mean_kernel<D, false>), grid, block, 0, s, P, sop, X, XS, alpha, mean);                  \
  } while (0)
  if (dim == 1) AGP_PM_LAUNCH(1);
  else if (dim == 2) AGP_PM_LAUNCH(2);
  else if (dim == 3) AGP_PM_LAUNCH(3);
  else if (dim == 4) AGP_PM_LAUNCH(4);
  else AGP_PM_LAUNCH(8);
#undef AGP_PM_LAUNCH
}

}  // namespace agp
