// solve.hip — K4: the substitutions against a factor held as L (lower triangle) + the tile images of its diagonal blocks:
// multi-RHS forward / backward / right solves on the MFMA (trsm_micro_kernel + update launches), their batched forms,
// the explicit inverses of the diagonal blocks, and the one-vector substitution chains.
//
// Replaces LDLT::solve (include/albatross/src/eigen/serializable_ldlt.hpp, used at models/gp.hpp:68,96,111) and
// matrixL().solveInPlace (serializable_ldlt.hpp:105,160).
#include "common.h"
#include "mfma_f64.h"
#include "gemm_tiles.h"
#include "trsm_kernel.h"
#include "pub.h"

namespace agp {
void launch_set_identity_batched(hipStream_t s, double *B, long long ld, long long stride, long long m, long long count);  // reduce.hip
void launch_colvec_dot_strided(hipStream_t s, const double *W, long long ld, long long stride_W, long long m, long long n,
                               const double *v, long long stride_v, double alpha, double beta, const double *base, double *out,
                               long long count);  // reduce.hip

// ---------------------------------------------------------------------------
// multi-RHS triangular solves (K4): B <- L^-1 B and B <- L^-T B
// ---------------------------------------------------------------------------
void forward_solve_mat(hipStream_t s, const double *A, long long n, long long lda, const double *invd,
                       double *B, long long m, long long ldb, bool rhs_lower) {
  // rhs_lower: column j of B is zero above row j (e.g. the identity): block row
  // k then only has work in its first k + nbk columns (N^3/3 instead of N^3 flop)
  if (m <= 0) return;
  for (long long K0 = 0; K0 < n; K0 += NBO) {
    const long long kend = (K0 + NBO < n) ? K0 + NBO : n;
    for (long long k = K0; k < kend; k += NB) {
      const int nbk = (int)((n - k < NB) ? n - k : NB);
      TrsmArgs t;
      t.img = invd + (k / NB) * (long long)IMG_DOUBLES;
      t.nbk = nbk;
      t.Y = B + k;
      t.stride_m = 1; t.stride_n = ldb;
      const long long m_act = (rhs_lower && k + nbk < m) ? k + nbk : m;
      t.ncols = m_act;
      t.z = nullptr; t.yrest = nullptr;
      t.batch_img = t.batch_Y = 0; t.n_total = 0;
      hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3((unsigned)((m_act + 63) / 64)), dim3(256), 0, s, t);
      const long long rows = kend - (k + nbk);
      if (rows > 0)  // B[k+nbk : kend] -= L[k+nbk : kend, k : k+nbk] B[k : k+nbk]
        launch_gemm_nt_sub(s, B + k + nbk, ldb, A + k * lda + (k + nbk), lda, false, B + k, ldb, true, rows, m_act,
                           nbk, false);
    }
    if (kend < n) {  // B[kend :] -= L[kend :, K0 : kend] B[K0 : kend]
      const long long m_act = (rhs_lower && kend < m) ? kend : m;
      launch_gemm_nt_sub(s, B + kend, ldb, A + K0 * lda + kend, lda, false, B + K0, ldb, true, n - kend, m_act,
                         kend - K0, false);
    }
  }
}

// The same substitution with one outer block of look-ahead on the context's two streams (like
// factor_lower): the solve of block rows j + 1 (small, latency-bound launches) runs on the main
// stream while the update of everything below with block j's solution (the MFMA-bound bulk) runs on
// the second one.  Returns with the main stream ordered after all work.
void forward_solve_mat_lookahead(agp_context *ctx, const double *A, long long n, long long lda, const double *invd,
                                 double *B, long long m, long long ldb, bool rhs_lower) {
  if (m <= 0) return;
  hipStream_t sa = ctx->stream, sb = ctx->stream2;
  if (n <= 2 * NBO || !sb || m < 64) {  // too small for the second stream to pay for its events
    forward_solve_mat(sa, A, n, lda, invd, B, m, ldb, rhs_lower);
    return;
  }
  // (AGP_SOLVE_NBO: measurement switch - outer block width of this substitution, a multiple of 128)
  static const long long nbo_env = [] { const char *e = getenv("AGP_SOLVE_NBO"); const long long v = e && e[0] ? atoll(e) : 0; return (v >= 128 && v % 128 == 0) ? v : (long long)NBO; }();
  const long long NBO = nbo_env;
  bool have_u2 = false;
  for (long long K0 = 0; K0 < n; K0 += NBO) {
    const long long kend = (K0 + NBO < n) ? K0 + NBO : n;
    for (long long k = K0; k < kend; k += NB) {
      const int nbk = (int)((n - k < NB) ? n - k : NB);
      TrsmArgs t;
      t.img = invd + (k / NB) * (long long)IMG_DOUBLES;
      t.nbk = nbk;
      t.Y = B + k;
      t.stride_m = 1; t.stride_n = ldb;
      const long long m_act = (rhs_lower && k + nbk < m) ? k + nbk : m;
      t.ncols = m_act;
      t.z = nullptr; t.yrest = nullptr;
      t.batch_img = t.batch_Y = 0; t.n_total = 0;
      hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3((unsigned)((m_act + 63) / 64)), dim3(256), 0, sa, t);
      const long long rows = kend - (k + nbk);
      if (rows > 0)
        launch_gemm_nt_sub(sa, B + k + nbk, ldb, A + k * lda + (k + nbk), lda, false, B + k, ldb, true, rows, m_act,
                           nbk, false);
    }
    if (kend >= n) break;
    const long long next_end = (kend + NBO < n) ? kend + NBO : n;
    const long long m_act = (rhs_lower && kend < m) ? kend : m;
    (void)hipEventRecord(ctx->ev_a, sa);                      // block j solved
    if (have_u2) (void)hipStreamWaitEvent(sa, ctx->ev_b, 0);  // U2(j - 1) done: it wrote the rows U1(j) writes
    // U1(j): the next block's rows
    launch_gemm_nt_sub(sa, B + kend, ldb, A + K0 * lda + kend, lda, false, B + K0, ldb, true, next_end - kend, m_act,
                       kend - K0, false);
    if (next_end < n) {
      (void)hipStreamWaitEvent(sb, ctx->ev_a, 0);
      launch_gemm_nt_sub(sb, B + next_end, ldb, A + K0 * lda + next_end, lda, false, B + K0, ldb, true, n - next_end,
                         m_act, kend - K0, false);
      (void)hipEventRecord(ctx->ev_b, sb);
      have_u2 = true;
    } else {
      have_u2 = false;
    }
  }
  if (have_u2) (void)hipStreamWaitEvent(sa, ctx->ev_b, 0);
}

// B_b (n x m, ldb) <- L_b^-1 B_b for `count` problems (B_b = B + b * stride_B); rhs_lower as in forward_solve_mat
void forward_solve_mat_batched(hipStream_t s, const double *A, long long stride_A, long long n, long long lda,
                               const double *invd, long long stride_invd, double *B, long long stride_B, long long m,
                               long long ldb, bool rhs_lower, long long count) {
  if (m <= 0 || count <= 0) return;
  for (long long k = 0; k < n; k += NB) {
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    const long long m_act = (rhs_lower && k + nbk < m) ? k + nbk : m;
    TrsmArgs t;
    t.img = invd + (k / NB) * (long long)IMG_DOUBLES;
    t.nbk = nbk;
    t.Y = B + k;
    t.stride_m = 1; t.stride_n = ldb;
    t.ncols = m_act;
    t.z = nullptr; t.yrest = nullptr;
    t.batch_img = stride_invd; t.batch_Y = stride_B; t.n_total = 0;
    hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3((unsigned)((m_act + 63) / 64), (unsigned)count), dim3(256), 0,
                       s, t);
    const long long rows = n - (k + nbk);
    if (rows > 0)  // B[k + nbk :] -= L[k + nbk :, k : k + nbk] B[k : k + nbk]
      launch_gemm_nt_sub_batched(s, B + k + nbk, ldb, stride_B, A + k * lda + (k + nbk), lda, false, stride_A, B + k, ldb,
                                 true, stride_B, rows, m_act, nbk, false, count);
  }
}

// X_b (nrows x n, ldx) <- X_b L_b^-T for `count` problems: X_b = X + b * stride_X, L_b = A + b * stride_A
void right_solve_lt_batched(hipStream_t s, const double *A, long long stride_A, long long n, long long lda,
                            const double *invd, long long stride_invd, double *X, long long stride_X, long long nrows,
                            long long ldx, long long count) {
  if (nrows <= 0 || count <= 0) return;
  for (long long k = 0; k < n; k += NB) {
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    TrsmArgs t;
    t.img = invd + (k / NB) * (long long)IMG_DOUBLES;
    t.nbk = nbk;
    t.Y = X + k * ldx;
    t.stride_m = ldx; t.stride_n = 1;
    t.ncols = nrows;
    t.z = nullptr; t.yrest = nullptr;
    t.batch_img = stride_invd; t.batch_Y = stride_X; t.n_total = 0;
    hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3((unsigned)((nrows + 63) / 64), (unsigned)count), dim3(256), 0,
                       s, t);
    const long long rest = n - (k + nbk);
    if (rest > 0)
      launch_gemm_nt_sub_batched(s, X + (k + nbk) * ldx, ldx, stride_X, X + k * ldx, ldx, false, stride_X,
                                 A + k * lda + (k + nbk), lda, false, stride_A, nrows, rest, nbk, false, count);
  }
}

// X (nrows x n, ldx) <- X L^-T : the panel TRSM of the factorisation applied to a free-standing
// matrix (sparse GP: K_uf[:, group] L_A^-T = (A^-1/2 K_fu)^T, models/sparse_gp.hpp:347-349).
void right_solve_lt(hipStream_t s, const double *A, long long n, long long lda, const double *invd, double *X,
                    long long nrows, long long ldx) {
  if (nrows <= 0) return;
  for (long long k = 0; k < n; k += NB) {
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    TrsmArgs t;
    t.img = invd + (k / NB) * (long long)IMG_DOUBLES;
    t.nbk = nbk;
    t.Y = X + k * ldx;           // Y = X[:, k : k + nbk]^T : element (m, n) at Y[m * ldx + n]
    t.stride_m = ldx; t.stride_n = 1;
    t.ncols = nrows;
    t.z = nullptr; t.yrest = nullptr;
    t.batch_img = t.batch_Y = 0; t.n_total = 0;
    hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3((unsigned)((nrows + 63) / 64)), dim3(256), 0, s, t);
    const long long rest = n - (k + nbk);
    if (rest > 0)  // X[:, k + nbk :] -= X[:, k : k + nbk] L[k + nbk :, k : k + nbk]^T
      launch_gemm_nt_sub(s, X + (k + nbk) * ldx, ldx, X + k * ldx, ldx, false, A + k * lda + (k + nbk), lda, false,
                         nrows, rest, nbk, false);
  }
}

void backward_solve_mat(hipStream_t s, const double *A, long long n, long long lda, const double *invd,
                        double *B, long long m, long long ldb) {
  if (m <= 0 || n <= 0) return;
  const long long nblk = (n + NB - 1) / NB;
  for (long long b = nblk - 1; b >= 0; --b) {
    const long long k = b * NB;
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    TrsmArgs t;
    t.img = invd + b * (long long)IMG_DOUBLES;
    t.nbk = nbk;
    t.Y = B + k;
    t.stride_m = 1; t.stride_n = ldb;
    t.ncols = m;
    t.z = nullptr; t.yrest = nullptr;
    t.batch_img = t.batch_Y = 0; t.n_total = 0;
    hipLaunchKernelGGL((trsm_micro_kernel<true, false>), dim3((unsigned)((m + 63) / 64)), dim3(256), 0, s, t);
    if (k > 0)  // B[0 : k] -= L[k : k+nbk, 0 : k]^T B[k : k+nbk]
      launch_gemm_nt_sub(s, B, ldb, A + k, lda, true, B + k, ldb, true, k, m, nbk, false);
  }
}

// ---------------------------------------------------------------------------
// one right-hand side: x = L^-T z  (second half of K^-1 y, gp.hpp:68)
//
// Right-looking over NB blocks from the bottom.  The diagonal blocks are
// inverted beforehand by ONE batched launch (all blocks in parallel, off the
// serial chain), so a step is two short kernels:
//   x_b = inv(L_bb)^T z_b                      (128 x 128 mat-vec, one workgroup)
//   z[0:k] -= L[k:k+nb, 0:k]^T x_b             (one wave per 8 columns, coalesced
//                                               1-KiB column segments)
// Bandwidth: L is read exactly once (8 N^2 / 2 bytes).  (Tried: two blocks per launch with the
// three 128 x 128 mat-vecs recomputed in every workgroup - 44 us per launch instead of 2 x 11.)
// ---------------------------------------------------------------------------
__global__ __launch_bounds__(256) void set_identity_blocks_kernel(double *W, long long count) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= count) return;
  const int within = (int)(i & (NB * NB - 1));
  W[i] = ((within >> 7) == (within & (NB - 1))) ? 1. : 0.;
}

// Winv[b] = inv(L_bb)^T as a column-major NB x NB array (i.e. inv(L_bb) row-major), for every diagonal block.
void invert_diag_blocks(hipStream_t s, const double *A, long long n, long long lda, const double *invd,
                        double *Winv) {
  const long long nblk = (n + NB - 1) / NB;
  const long long count = nblk * NB * NB;
  hipLaunchKernelGGL(set_identity_blocks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, Winv, count);
  TrsmArgs t;
  (void)A; (void)lda;
  t.img = invd; t.nbk = NB;
  // element (m, n) of inv(L_bb) goes to Winv[m * NB + n]: the blocks are stored TRANSPOSED
  // (row-major), so that the mat-vec x = inv(L_bb)^T z reads them coalesced
  t.Y = Winv; t.stride_m = NB; t.stride_n = 1; t.ncols = NB;
  t.z = nullptr; t.yrest = nullptr;
  t.batch_img = IMG_DOUBLES; t.batch_Y = NB * NB; t.n_total = n;
  hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3(2, (unsigned)nblk), dim3(256), 0, s, t);
}

// Wfwd[b] = inv(L_bb) column-major (element (m, n) at [n * NB + m]): what the FORWARD vector
// substitution reads coalesced.
void invert_diag_blocks_forward(hipStream_t s, long long n, const double *invd, double *Wfwd) {
  const long long nblk = (n + NB - 1) / NB;
  const long long count = nblk * NB * NB;
  hipLaunchKernelGGL(set_identity_blocks_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, Wfwd, count);
  TrsmArgs t;
  t.img = invd; t.nbk = NB;
  t.Y = Wfwd; t.stride_m = 1; t.stride_n = NB; t.ncols = NB;
  t.z = nullptr; t.yrest = nullptr;
  t.batch_img = IMG_DOUBLES; t.batch_Y = NB * NB; t.n_total = n;
  hipLaunchKernelGGL((trsm_micro_kernel<false, false>), dim3(2, (unsigned)nblk), dim3(256), 0, s, t);
}

// One step of the right-looking FORWARD substitution on a vector, ONE launch:
//   x_b = inv(L_bb) z_b                         (recomputed by every workgroup, as in back_step_kernel)
//   z[i] -= sum_c L[i][k0 + c] x_b[c]           for this workgroup's 64 rows i >= k0 + nbk
// (wave w sums columns 32 w .. 32 w + 31, lane = row: coalesced 512-B column segments)
__global__ __launch_bounds__(256) void fwd_step_kernel(const double *__restrict__ A, long long lda, long long k0,
                                                       int nbk, long long n, const double *__restrict__ Wfwd,
                                                       double *__restrict__ z, double *__restrict__ x_out) {
  __shared__ double xs[NB], part[NB], zs[NB], red[4][64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < NB) zs[tid] = (tid < nbk) ? z[k0 + tid] : 0.;
  __syncthreads();
  {
    // x[c] = sum_r inv(L)[c][r] z[r];  Wfwd holds inv(L)[c][r] at [r * NB + c]
    const int c = tid & (NB - 1), half = tid >> 7;
    double acc = 0.;
#pragma unroll 8
    for (int r = half * 64; r < half * 64 + 64; ++r) acc += Wfwd[r * NB + c] * zs[r];
    if (half == 1) part[c] = acc;
    __syncthreads();
    if (half == 0) {
      const double v = (c < nbk) ? acc + part[c] : 0.;
      xs[c] = v;
      if (blockIdx.x == 0 && c < nbk) x_out[k0 + c] = v;
    }
    __syncthreads();
  }
  const long long i = k0 + nbk + (long long)blockIdx.x * 64 + lane;
  double acc = 0.;
  if (i < n) {
    const double *p = A + (k0 + 32 * wave) * lda + i;
#pragma unroll 8
    for (int c = 0; c < 32; ++c)
      if (32 * wave + c < nbk) acc += p[(long long)c * lda] * xs[32 * wave + c];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && i < n) z[i] -= (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// z <- L^-1 z for one vector (the fused substitution of the factorisation covers the fit's own y; this
// one serves the refinement steps of the mixed-precision fit).  xstage: n doubles.
void forward_solve_vec(hipStream_t s, const double *A, long long n, long long lda, const double *Wfwd, double *z,
                       double *xstage) {
  const long long nblk = (n + NB - 1) / NB;
  for (long long b = 0; b < nblk; ++b) {
    const long long k = b * NB;
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    const long long below = n - k - nbk;
    const unsigned grid = (unsigned)(below > 0 ? (below + 63) / 64 : 1);
    hipLaunchKernelGGL(fwd_step_kernel, dim3(grid), dim3(256), 0, s, A, lda, k, nbk, n,
                       Wfwd + b * (long long)(NB * NB), z, xstage);
  }
  (void)hipMemcpyAsync(z, xstage, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
}

// One step of the right-looking back substitution, ONE launch:
//   x_b = inv(L_bb)^T z_b            (every workgroup recomputes this 128 x 128 mat-vec from the
//                                     L2-resident transposed inverse: 128 KB, coalesced rows)
//   z[c] -= sum_r L[k0 + r][c] x_b[r]  for this workgroup's 32 columns c < k0
// Workgroup 0 also publishes x_b into `x_out` (z_b itself stays untouched: other workgroups may
// still be reading it).
__global__ __launch_bounds__(256) void back_step_kernel(const double *__restrict__ A, long long lda, long long k0,
                                                        int nbk, const double *__restrict__ WinvT,
                                                        double *__restrict__ z, double *__restrict__ x_out) {
  __shared__ double xs[NB], part[NB], zs[NB];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < NB) zs[tid] = (tid < nbk) ? z[k0 + tid] : 0.;
  __syncthreads();
  {
    // x[c] = sum_r inv(L)[r][c] z[r];  WinvT holds inv(L)[r][c] at [r * NB + c]
    const int c = tid & (NB - 1), half = tid >> 7;
    double acc = 0.;
#pragma unroll 8
    for (int r = half * 64; r < half * 64 + 64; ++r) acc += WinvT[r * NB + c] * zs[r];
    if (half == 1) part[c] = acc;
    __syncthreads();
    if (half == 0) {
      const double v = (c < nbk) ? acc + part[c] : 0.;
      xs[c] = v;
      if (blockIdx.x == 0 && c < nbk) x_out[k0 + c] = v;
    }
    __syncthreads();
  }
  const long long c0 = ((long long)blockIdx.x * 4 + wave) * 8;
  if (c0 >= k0) return;
  const int r = 2 * lane;
  const double x0 = xs[r], x1 = xs[r + 1];
  const bool vec = ((lda & 1) == 0) && ((k0 & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  double acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const long long c = c0 + q;
    double a0 = 0., a1 = 0.;
    if (c < k0) {
      const double *p = A + c * lda + k0 + r;
      if (vec && r + 1 < nbk) {
        const double2 v = *reinterpret_cast<const double2 *>(p);
        a0 = v.x; a1 = v.y;
      } else {
        a0 = r < nbk ? p[0] : 0.;
        a1 = r + 1 < nbk ? p[1] : 0.;
      }
    }
    acc[q] = a0 * x0 + a1 * x1;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (c0 + q < k0) z[c0 + q] -= acc[q];
  }
}

// z[c] -= sum_r L[k0 + r][c] x[r]  for c < ncols ; r < nbk.   8 columns per wave.
__global__ __launch_bounds__(256) void back_update_kernel(const double *__restrict__ A, long long lda,
                                                          long long k0, int nbk, long long ncols,
                                                          const double *__restrict__ x, double *__restrict__ z) {
  // rows k0 .. k0 + nbk of A, columns 0 .. ncols
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long c0 = ((long long)blockIdx.x * 4 + wave) * 8;
  if (c0 >= ncols) return;
  const int r = 2 * lane;
  const double x0 = r < nbk ? x[r] : 0., x1 = r + 1 < nbk ? x[r + 1] : 0.;
  const bool vec = ((lda & 1) == 0) && ((k0 & 1) == 0) && ((reinterpret_cast<uintptr_t>(A) & 15) == 0);
  double acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    const long long c = c0 + q;
    double a0 = 0., a1 = 0.;
    if (c < ncols) {
      const double *p = A + c * lda + k0 + r;
      if (vec && r + 1 < nbk) {
        const double2 v = *reinterpret_cast<const double2 *>(p);
        a0 = v.x; a1 = v.y;
      } else {
        a0 = r < nbk ? p[0] : 0.;
        a1 = r + 1 < nbk ? p[1] : 0.;
      }
    }
    acc[q] = a0 * x0 + a1 * x1;
  }
#pragma unroll
  for (int q = 0; q < 8; ++q) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) acc[q] += __shfl_down(acc[q], off, 64);
  }
  if (lane == 0) {
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (c0 + q < ncols) z[c0 + q] -= acc[q];
  }
}

void launch_back_update(hipStream_t s, const double *A, long long lda, long long k0, int nbk, long long ncols,
                        const double *x, double *z) {
  if (ncols <= 0 || nbk <= 0) return;
  hipLaunchKernelGGL(back_update_kernel, dim3((unsigned)((ncols + 31) / 32)), dim3(256), 0, s, A, lda, k0, nbk,
                     ncols, x, z);
}

void backward_solve_vec(hipStream_t s, const double *A, long long n, long long lda, const double *Winv,
                        double *z, double *xstage) {
  // z is consumed; the solution is produced in `xstage` (n doubles) block by block and copied
  // back at the end: x of block b may not overwrite z_b while other workgroups of the same
  // launch still read z_b
  const long long nblk = (n + NB - 1) / NB;
  for (long long b = nblk - 1; b >= 0; --b) {
    const long long k = b * NB;
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    const unsigned grid = (unsigned)(k > 0 ? (k + 31) / 32 : 1);
    hipLaunchKernelGGL(back_step_kernel, dim3(grid), dim3(256), 0, s, A, lda, k, nbk,
                       Winv + b * (long long)(NB * NB), z, xstage);
  }
  (void)hipMemcpyAsync(z, xstage, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
}

// z_b <- L_b^-T z_b for `count` problems with ONE right-hand side each (the information vectors of
// agp_fit_create_batch): per 128 rows from the bottom one MFMA substitution against the tile image (blockIdx.y =
// problem) and one batched column-dot launch for the rows above.
void backward_solve_vec_batched(hipStream_t s, const double *A, long long stride_A, long long n, long long lda,
                                const double *invd, long long stride_invd, double *z, long long stride_z, long long count) {
  if (n <= 0 || count <= 0) return;
  const long long nblk = (n + NB - 1) / NB;
  for (long long b = nblk - 1; b >= 0; --b) {
    const long long k = b * NB;
    const int nbk = (int)((n - k < NB) ? n - k : NB);
    TrsmArgs t;
    t.img = invd + b * (long long)IMG_DOUBLES;
    t.nbk = nbk;
    t.Y = z + k;
    t.stride_m = 1; t.stride_n = 1;
    t.ncols = 1;
    t.z = nullptr; t.yrest = nullptr;
    t.batch_img = stride_invd; t.batch_Y = stride_z; t.n_total = 0;
    hipLaunchKernelGGL((trsm_micro_kernel<true, false>), dim3(1, (unsigned)count), dim3(256), 0, s, t);
    if (k > 0)  // z[0 : k] -= L[k : k + nbk, 0 : k]^T x
      launch_colvec_dot_strided(s, A + k, lda, stride_A, nbk, k, z + k, stride_z, -1.0, 1.0, z, z, count);
  }
}


// ---------------------------------------------------------------------------------------------------------------
// x = L^-T z for ONE vector in ONE launch (round 5).
// The launch-per-block chains above cost a launch (4-8 us of stream time) per 128 or 512 rows - 45 us at N = 512,
// 190 us at N = 4096 (of which 108 us to invert the wide diagonal blocks), 0.6 ms at N = 16384 - for a computation
// whose serial part is a 128 x 128 substitution per block.  Here workgroup b owns the 128-column block column b:
//   acc_b = sum_{j > b} L[block row j, block column b]^T x_j   accumulated block row by block row as the x_j appear
//   x_b   = L_bb^-T (z_b - acc_b)                              micro-block substitution against the tile image in LDS
// and PUBLISHES x_b in the output vector itself: `x` is sentinel-filled before the launch, written with device-scope
// stores and polled with device-scope loads (pub.h) - no flags, no atomics.  Every workgroup but the one that owns
// block row j is AHEAD of its next dependency (it consumes x_j while the owner of block j - 1 is still substituting),
// so the chain per block is: one hand-over (~1 us), 32 FMAs per lane, one butterfly reduction, the substitution.
// Liveness does not need co-residency: blockIdx.x = 0 is the LAST block, workgroups are dispatched in order, and a
// workgroup only ever waits for blocks dispatched before it.  blockIdx.y = problem of a batch.
// Layout of the work inside a workgroup (512 threads): wave w owns the columns 16 w .. 16 w + 15 of the block column,
// lane l the rows l and l + 64 of the current block row - coalesced 512-B column segments, 32 values in flight per lane,
// requested right after the previous block row has been consumed (the next x is ~3 us away).
// ---------------------------------------------------------------------------------------------------------------
struct BackCoopArgs {
  const double *A;
  long long lda, n;
  const double *img;  // tile images of the diagonal blocks (IMG_DOUBLES each)
  const double *z;    // right-hand side
  double *x;          // solution
  int *flags;         // flags[2]: a hand-over timed out
  unsigned long long *done;  // one word per block (zeroed by the caller): set when the block's x is in memory;
                             // nullptr: x itself is the hand-over (SENTINEL-filled by the caller, polled per value)
  long long batch_A = 0, batch_img = 0, batch_z = 0, batch_x = 0, batch_flags = 0, batch_done = 0;
  // the fit's status block (flags + scalars, device memory) and its pinned host mirror as the device sees it: the
  // workgroup that finishes LAST (block 0) forwards it - the two 4-5 us copy launches behind a small fit
  const unsigned long long *status_src = nullptr;
  unsigned long long *status_dst = nullptr;
  int status_words = 0;
};

__global__ __launch_bounds__(512) void backsub_coop_kernel(BackCoopArgs p) {
  __shared__ double F[IMG_DOUBLES];
  __shared__ double ts[NB];
  {
    const long long pb = blockIdx.y;
    p.A += pb * p.batch_A; p.img += pb * p.batch_img; p.z += pb * p.batch_z; p.x += pb * p.batch_x;
    if (p.flags) p.flags += pb * p.batch_flags;
    if (p.done) p.done += pb * p.batch_done;
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long nb = (p.n + NB - 1) / NB;
  const long long b = nb - 1 - (long long)blockIdx.x;
  const long long k0 = b * NB;
  const int nbk = (int)((p.n - k0 < NB) ? p.n - k0 : NB);
  // the image of this block: needed only at the very end, requested first
  {
    const double *src = p.img + b * (long long)IMG_DOUBLES;
#pragma unroll
    for (int it = 0; it < IMG_DOUBLES / 2 / 512; ++it) {
      const int e = 2 * (tid + 512 * it);
      *reinterpret_cast<double2 *>(F + e) = *reinterpret_cast<const double2 *>(src + e);
    }
  }
  double acc[16];
#pragma unroll
  for (int q = 0; q < 16; ++q) acc[q] = 0.;
  const long long c0 = k0 + 16 * wave;  // first column of this wave
  double a0[16], a1[16];
  auto load_rows = [&](long long j) {
    const long long r0 = j * NB + lane, r1 = r0 + 64;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const long long c = c0 + q;
      const double *col = p.A + c * p.lda;
      a0[q] = (c < p.n && r0 < p.n) ? col[r0] : 0.;
      a1[q] = (c < p.n && r1 < p.n) ? col[r1] : 0.;
    }
  };
  if (nb - 1 > b) load_rows(nb - 1);
  for (long long j = nb - 1; j > b; --j) {
    const long long r0 = j * NB + lane, r1 = r0 + 64;
    double x0 = 0., x1 = 0.;
    if (p.done) {
      // many blocks: ONE lane per wave polls the block's flag (every lane polling the values themselves is 131 k threads
      // hammering the memory side at N = 4096), then every lane reads its two values once
      if (lane == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
        for (int spin = 1; __hip_atomic_load(p.done + j, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0; ++spin) {
          if ((spin & 63) == 0 && poll_expired(t0, p.flags)) break;
          __builtin_amdgcn_s_sleep(4);
        }
      }
      if (r0 < p.n) x0 = load_pub(p.x + r0);
      if (r1 < p.n) x1 = load_pub(p.x + r1);
    } else {
      // few blocks (<= BACKSUB_DIRECT_BLOCKS x 512 threads): the values themselves are the hand-over - x enters
      // sentinel-filled and every lane polls its own two: ONE memory-side trip per block instead of flag, then values
      const bool h0 = r0 < p.n, h1 = r1 < p.n;
      const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
      for (int spin = 1;; ++spin) {
        if (h0) x0 = load_pub(p.x + r0);
        if (h1) x1 = load_pub(p.x + r1);
        if (!((h0 && is_unpublished(x0)) || (h1 && is_unpublished(x1)))) break;
        if ((spin & 63) == 0 && poll_expired(t0, p.flags)) break;
        __builtin_amdgcn_s_sleep(2);
      }
    }
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] += a0[q] * x0 + a1[q] * x1;
    if (j - 1 > b) load_rows(j - 1);
  }
  const double sum = colsum16(acc);  // (trsm_kernel.h)
  if (lane < 16) {
    const int c = 16 * wave + colsum16_index(lane);
    ts[c] = (c < nbk) ? p.z[k0 + c] - sum : 0.;
  }
  __syncthreads();
  if (wave != 0) return;
  // x_b = L_bb^-T t by micro blocks (trsm_kernel.h), every value published the moment it is final
  micro_backsub_wave(F, ts, [&](int c, double v) { if (c < nbk) store_pub(p.x + k0 + c, v); });
  if (p.done && lane == 0) __hip_atomic_store(p.done + b, 1ull, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);  // (behind the wave's stores)
  // block 0 consumed every other block's x: whatever those raised in the status block is in memory by now
  if (b == 0 && p.status_dst && lane < p.status_words)
    p.status_dst[lane] = __hip_atomic_load(p.status_src + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// x = L^-T z; count problems (strides 0 for one).  done: one zeroed word per 128-row block and problem (backsub_done_words)
long long backsub_done_words(long long n, long long count) { return (n + NB - 1) / NB * (count > 0 ? count : 1); }
void backward_solve_coop(hipStream_t s, const double *A, long long n, long long lda, const double *invd, const double *z,
                         double *x, int *flags, unsigned long long *done, long long count, long long stride_A, long long stride_invd,
                         long long stride_z, long long stride_x, long long stride_flags, const void *status_src, void *status_dst,
                         int status_words) {
  if (n <= 0 || count <= 0) return;
  BackCoopArgs p;
  if (count == 1 && status_src && status_dst && status_words > 0 && status_words <= 64) {
    p.status_src = static_cast<const unsigned long long *>(status_src);
    p.status_dst = static_cast<unsigned long long *>(status_dst);
    p.status_words = status_words;
  }
  p.done = done; p.batch_done = (n + NB - 1) / NB;
  p.A = A; p.lda = lda; p.n = n; p.img = invd; p.z = z; p.x = x; p.flags = flags;
  p.batch_A = stride_A; p.batch_img = stride_invd; p.batch_z = stride_z; p.batch_x = stride_x; p.batch_flags = stride_flags;
  const long long nb = (n + NB - 1) / NB;
  hipLaunchKernelGGL(backsub_coop_kernel, dim3((unsigned)nb, (unsigned)count), dim3(512), 0, s, p);
}

void launch_fill_sentinel(hipStream_t s, double *p, long long count) {
  PrepArgs a;
  a.sentinel(p, count);
  launch_prep(s, a);
}


// ---- X = L^-1 B out of place for a right-hand side MUCH wider than L (the sparse GP's m x n matrices K_uf and W, n in the
// hundred thousands) --------------------------------------------------------------------------------------------------
// forward_solve_mat walks such a right-hand side 16 times per 2048 rows with K = 128 products; here the 512 x 512
// diagonal blocks are inverted explicitly (Winv, invert_wide_blocks) and a block row is two steps:
//   X_i = B_i - L[i, 0:i] X[0:i]      one product of depth 512 i over all columns (B is read once, nothing is copied)
//   X_i = inv(L_ii) X_i               four products of depth <= 512 with the inverse's 128-row tile rows, bottom-up in
//                                     place (tile row r reads rows <= r of the same column strip before it stores)
// n a multiple of 512.  B and X may not overlap.
void launch_set_identity_batched(hipStream_t s, double *B, long long ld, long long stride, long long m, long long count);  // reduce.hip

bool forward_solve_wide_ok(long long n, long long ncols) { return n >= 2 * WIDE_BW && n % WIDE_BW == 0 && ncols >= 8 * n; }

void invert_wide_blocks(hipStream_t s, const double *A, long long n, long long lda, const double *invd, long long BW, double *W) {
  const long long nb = n / BW;
  launch_set_identity_batched(s, W, BW, BW * BW, BW, nb);
  forward_solve_mat_batched(s, A, BW * (lda + 1), BW, lda, invd, (BW / NB) * (long long)IMG_DOUBLES, W, BW * BW, BW, BW,
                            /*rhs_lower=*/true, nb);
}

void forward_solve_wide(hipStream_t s, const double *A, long long n, long long lda, const double *Winv, const double *B,
                        long long ldb, double *X, long long ldx, long long ncols) {
  constexpr long long BW = WIDE_BW;
  if (ncols <= 0) return;
  for (long long i = 0; i < n / BW; ++i) {
    const long long k0 = i * BW;
    const double *Wi = Winv + i * BW * BW;
    if (i > 0) launch_gemm_nt_ext(s, X + k0, ldx, B + k0, ldb, A + k0, lda, X, ldx, BW, ncols, k0);
    const double *src = i > 0 ? X + k0 : B;
    const long long lds = i > 0 ? ldx : ldb;
    for (long long r = BW / NB - 1; r >= 0; --r)
      launch_gemm_nt_ext(s, X + k0 + r * NB, ldx, nullptr, 0, Wi + r * NB, BW, src, lds, NB, ncols, (r + 1) * NB);
  }
}

}  // namespace agp
