// ldlt.hip — pivoted L D L^T for symmetric (semi-)definite matrices: the fallback for inputs the
// un-pivoted LL^T of chol.hip rejects.
//
// Replaces Eigen::LDLT<MatrixXd, Lower>::compute + solve as albatross uses it through
// SerializableLDLT (include/albatross/src/eigen/serializable_ldlt.hpp:27; call sites
// evaluation/likelihood.hpp:63, covariance_functions/representations.hpp:64-96, models/gp.hpp:148,393).
// Eigen 3.3's LDLT is an UNBLOCKED left-looking factorisation with diagonal pivoting:
//   for k: p = first argmax_{i >= k} |A_ii|, symmetric swap k <-> p,
//          temp = D[:k] .* A[k, :k],  A_kk -= A[k, :k] . temp,  A[k+1:, k] -= A[k+1:, :k] temp,
//          A[k+1:, k] /= A_kk  (if the pivot is non-zero)
// The diagonal entries i > k are untouched until they become the pivot, so the transposition
// sequence follows from the INITIAL diagonal alone: the host derives it from one n-double download
// and the device runs the n column steps (two launches each) in exactly the reference's operation
// order — this file is compiled with -ffp-contract=off and every row accumulates left to right, so
// L, D and P are bit-identical to the CPU restatement.  Cost: n^3/3 flop at level-2 intensity
// (8 n^3 / 6 bytes of reads): a correctness path for moderate n, not a fast one.
//
// The solve (P^T L^-T D^+ L^-1 P b, D^+ zeroing the numerically zero pivots like Eigen) is blocked:
// 64 x 64 unit-triangular diagonal blocks by substitution, everything else on the fp64 MFMA update
// kernel of gemm.hip.
#include "common.h"

namespace agp {

constexpr int LB = 64;  // diagonal block of the triangular solves

// ---- factorisation ---------------------------------------------------------------------------
// one workgroup: symmetric swap k <-> p, temp = D .* A[k, :k], pivot update, bookkeeping
// info[0] = found_zero_pivot, info[1] = ok (Eigen's Success), scal[0] = A_kk after the update,
// scal[1] = pivot_is_valid
__global__ __launch_bounds__(1024) void ldlt_pivot_kernel(double *A, long long lda, long long n, long long k, long long p,
                                                          double *temp, int *info, double *scal) {
  __shared__ double prod[2048];
  __shared__ double dot_s;
  const int tid = threadIdx.x;
  if (p != k) {
    for (long long c = tid; c < k; c += 1024) {  // rows k, p in the finished columns
      const double t = A[k + c * lda];
      A[k + c * lda] = A[p + c * lda];
      A[p + c * lda] = t;
    }
    for (long long r = p + 1 + tid; r < n; r += 1024) {  // columns k, p below p
      const double t = A[r + k * lda];
      A[r + k * lda] = A[r + p * lda];
      A[r + p * lda] = t;
    }
    for (long long i = k + 1 + tid; i < p; i += 1024) {  // the part between: column k <-> row p
      const double t = A[i + k * lda];
      A[i + k * lda] = A[p + i * lda];
      A[p + i * lda] = t;
    }
    if (tid == 0) {
      const double t = A[k + k * lda];
      A[k + k * lda] = A[p + p * lda];
      A[p + p * lda] = t;
    }
    __threadfence_block();
    __syncthreads();
  }
  if (tid == 0) dot_s = 0.;
  __syncthreads();
  for (long long c0 = 0; c0 < k; c0 += 2048) {
    const long long cnt = (k - c0 < 2048) ? k - c0 : 2048;
    for (long long c = tid; c < cnt; c += 1024) {
      const double akc = A[k + (c0 + c) * lda];
      const double t = A[(c0 + c) + (c0 + c) * lda] * akc;  // temp = D .* A10^T
      temp[c0 + c] = t;
      prod[c] = akc * t;
    }
    __syncthreads();
    if (tid == 0) {  // left-to-right sum, as the reference accumulates it
      double d = dot_s;
      long long c = 0;
      for (; c + 8 <= cnt; c += 8) {
        double q[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) q[e] = prod[c + e];
#pragma unroll
        for (int e = 0; e < 8; ++e) d += q[e];
      }
      for (; c < cnt; ++c) d += prod[c];
      dot_s = d;
    }
    __syncthreads();
  }
  if (tid == 0) {
    double akk = A[k + k * lda];
    if (k > 0) {
      akk -= dot_s;
      A[k + k * lda] = akk;
    }
    const int valid = fabs(akk) > 0.;
    scal[0] = akk;
    scal[1] = valid ? 1. : 0.;
    if (info[0] && valid) info[1] = 0;  // a non-zero pivot after a zero one: NumericalIssue
    else if (!valid) info[0] = 1;
  }
}

// rows r > k: A_rk -= sum_c A_rc temp_c (left to right), then / A_kk; a zero pivot requires a zero column
__global__ __launch_bounds__(256) void ldlt_column_kernel(double *A, long long lda, long long n, long long k,
                                                          const double *__restrict__ temp, int *info,
                                                          const double *__restrict__ scal) {
  const long long r = k + 1 + (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double v = A[r + k * lda];
  // left-to-right like the reference, with 2 x 32 loads in flight (few rows -> few waves: the latency of a
  // dependent load per term would otherwise be fully exposed); the chain is only the subtractions
  long long c = 0;
  if (k >= 32) {
    double a[32], b[32];
#pragma unroll
    for (int q = 0; q < 32; ++q) a[q] = A[r + q * lda];
    for (; c + 64 <= k; c += 32) {
#pragma unroll
      for (int q = 0; q < 32; ++q) b[q] = A[r + (c + 32 + q) * lda];
#pragma unroll
      for (int q = 0; q < 32; ++q) v -= a[q] * temp[c + q];
#pragma unroll
      for (int q = 0; q < 32; ++q) a[q] = b[q];
    }
#pragma unroll
    for (int q = 0; q < 32; ++q) v -= a[q] * temp[c + q];
    c += 32;
  }
  for (; c < k; ++c) v -= A[r + c * lda] * temp[c];
  if (scal[1] != 0.) v /= scal[0];
  else if (v != 0.) info[1] = 0;
  A[r + k * lda] = v;
}

void ldlt_factor(hipStream_t s, double *A, long long lda, long long n, const long long *tr_host, double *temp, int *info,
                 double *scal) {
  for (long long k = 0; k < n; ++k) {
    hipLaunchKernelGGL(ldlt_pivot_kernel, dim3(1), dim3(1024), 0, s, A, lda, n, k, tr_host[k], temp, info, scal);
    const long long rs = n - k - 1;
    if (rs > 0)
      hipLaunchKernelGGL(ldlt_column_kernel, dim3((unsigned)((rs + 255) / 256)), dim3(256), 0, s, A, lda, n, k, temp, info,
                         scal);
  }
}

// ---- blocked variant (same arithmetic, GPU-sized launches) ------------------------------------------
// The transpositions are known up front, so the matrix can be permuted ONCE (Ap = P A P^T) and factored
// without further swaps.  Every entry then still receives exactly the reference's sequence of operations:
//   off-diagonal (r, k):  v = A_rk; for c = 0 .. k-1 ascending: v -= L_rc * (D_c L_kc);  v /= D_k
//   diagonal k:           A_kk -= sum_{c<k} L_kc * (D_c L_kc)   (summed left to right from zero, then subtracted)
// only the ORDER OF LAUNCHES changes: columns are finished 32 at a time (diagonal block by one workgroup, the
// rows below by a row-per-thread kernel), and the terms of those 32 columns are applied to every later entry
// by a tiled kernel that walks c in ascending order with separate multiply and subtract.  The diagonal sums
// are carried in `dotacc`.  Results are bit-identical to the unblocked kernels above.
constexpr int PB = 32;  // columns finished per block

// Ap lower triangle (i >= j) <- S[q[i], q[j]] of the symmetric S
__global__ __launch_bounds__(256) void ldlt_permute_sym_kernel(const double *__restrict__ S, long long lds,
                                                               const long long *__restrict__ q, long long n,
                                                               double *__restrict__ Ap, long long lda) {
  const long long j = blockIdx.y;
  const long long qj = q[j];
  for (long long i = j + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    Ap[i + j * lda] = S[q[i] + qj * lds];
}

// diagonal block [c0, c0 + nbk): one wave, lane t = row c0 + t
__global__ __launch_bounds__(64) void ldlt_diag_block_kernel(double *A, long long lda, long long c0, int nbk, double *T,
                                                             long long ldt, double *dotacc, int *info) {
  __shared__ double S[PB][PB + 1], Tl[PB][PB + 1];
  __shared__ double dk_s;
  __shared__ int valid_s;
  const int t = threadIdx.x;
  if (t < nbk)
    for (int c = 0; c <= t; ++c) S[t][c] = A[(c0 + t) + (c0 + c) * lda];
  double dacc = (t < nbk) ? dotacc[c0 + t] : 0.;
  __syncthreads();
  for (int k = 0; k < nbk; ++k) {
    if (t == k) {
      double akk = S[k][k];
      if (c0 + k > 0) akk -= dacc;
      S[k][k] = akk;
      const int valid = fabs(akk) > 0.;
      dk_s = akk;
      valid_s = valid;
      if (info[0] && valid) info[1] = 0;
      else if (!valid) info[0] = 1;
    }
    __syncthreads();
    if (t > k && t < nbk) {
      double v = S[t][k];
      for (int c = 0; c < k; ++c) v -= S[t][c] * Tl[k][c];
      if (valid_s) v /= dk_s;
      else if (v != 0.) info[1] = 0;
      S[t][k] = v;
      const double tt = dk_s * v;  // temp entry D_k * L_tk
      Tl[t][k] = tt;
      dacc += v * tt;
    }
    __syncthreads();
  }
  if (t < nbk) {
    for (int c = 0; c <= t; ++c) A[(c0 + t) + (c0 + c) * lda] = S[t][c];
    for (int c = 0; c < t; ++c) T[(c0 + t) + c * ldt] = Tl[t][c];
  }
}

// rows below the diagonal block: thread = row; finishes its 32 entries, their temp values and its diagonal sum
__global__ __launch_bounds__(256) void ldlt_panel_kernel(double *A, long long lda, long long n, long long c0, int nbk,
                                                         double *T, long long ldt, double *dotacc, int *info) {
  __shared__ double Tl[PB][PB + 1];
  __shared__ double dks[PB];
  for (int e = threadIdx.x; e < PB * PB; e += 256) {
    const int k = e / PB, c = e % PB;
    Tl[k][c] = (k < nbk && c < k) ? T[(c0 + k) + c * ldt] : 0.;
  }
  if (threadIdx.x < PB) dks[threadIdx.x] = threadIdx.x < nbk ? A[(c0 + threadIdx.x) + (c0 + threadIdx.x) * lda] : 1.;
  __syncthreads();
  const long long r = c0 + nbk + (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= n) return;
  double x[PB];
#pragma unroll
  for (int k = 0; k < PB; ++k) x[k] = k < nbk ? A[r + (c0 + k) * lda] : 0.;
  double dacc = dotacc[r];
#pragma unroll
  for (int k = 0; k < PB; ++k) {
    if (k < nbk) {
      double v = x[k];
#pragma unroll
      for (int c = 0; c < k; ++c) v -= x[c] * Tl[k][c];
      const double dk = dks[k];
      if (fabs(dk) > 0.) v /= dk;
      else if (v != 0.) info[1] = 0;
      x[k] = v;
      const double tt = dk * v;
      T[r + k * ldt] = tt;
      dacc += v * tt;
      A[r + (c0 + k) * lda] = v;
    }
  }
  dotacc[r] = dacc;
}

// entries (r, k), r > k >= c1: the nbk terms of the finished block, ascending, multiply then subtract
__global__ __launch_bounds__(256) void ldlt_trailing_kernel(double *A, long long lda, long long n, long long c0, int nbk,
                                                            long long c1, const double *__restrict__ T, long long ldt,
                                                            int ntile) {
  __shared__ double Lr[64][PB + 1], Tk[64][PB + 1];
  // lower tiles of the (n - c1)^2 trailing block, column by column
  int bj = 0;
  long long id = blockIdx.x;
  while (id >= ntile - bj) { id -= ntile - bj; ++bj; }
  const int bi = bj + (int)id;
  const long long r0 = c1 + 64LL * bi, k0 = c1 + 64LL * bj;
  for (int e = threadIdx.x; e < 64 * PB; e += 256) {
    const int i = e & 63, c = e >> 6;
    Lr[i][c] = (r0 + i < n && c < nbk) ? A[(r0 + i) + (c0 + c) * lda] : 0.;
    Tk[i][c] = (k0 + i < n && c < nbk) ? T[(k0 + i) + c * ldt] : 0.;
  }
  __syncthreads();
  const int tx = threadIdx.x & 15, ty = threadIdx.x >> 4;  // rows 4 tx .. 4 tx + 3, columns 4 ty .. 4 ty + 3
  double v[4][4];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const long long r = r0 + 4 * tx + a, k = k0 + 4 * ty + b;
      v[b][a] = (r < n && k < n && r > k) ? A[r + k * lda] : 0.;
    }
  for (int c = 0; c < nbk; ++c) {
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double tk = Tk[4 * ty + b][c];
#pragma unroll
      for (int a = 0; a < 4; ++a) v[b][a] -= Lr[4 * tx + a][c] * tk;
    }
  }
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const long long r = r0 + 4 * tx + a, k = k0 + 4 * ty + b;
      if (r < n && k < n && r > k) A[r + k * lda] = v[b][a];
    }
}

// Ap (lower, already permuted) -> L, D in place; T: n x PB scratch, dotacc: n doubles (zeroed here)
void ldlt_factor_blocked(hipStream_t s, double *Ap, long long lda, long long n, double *T, double *dotacc, int *info) {
  (void)hipMemsetAsync(dotacc, 0, sizeof(double) * (size_t)n, s);
  for (long long c0 = 0; c0 < n; c0 += PB) {
    const int nbk = (int)((n - c0 < PB) ? n - c0 : PB);
    hipLaunchKernelGGL(ldlt_diag_block_kernel, dim3(1), dim3(64), 0, s, Ap, lda, c0, nbk, T, n, dotacc, info);
    const long long c1 = c0 + nbk, below = n - c1;
    if (below <= 0) continue;
    hipLaunchKernelGGL(ldlt_panel_kernel, dim3((unsigned)((below + 255) / 256)), dim3(256), 0, s, Ap, lda, n, c0, nbk, T, n,
                       dotacc, info);
    const int ntile = (int)((below + 63) / 64);
    const long long tiles = (long long)ntile * (ntile + 1) / 2;
    hipLaunchKernelGGL(ldlt_trailing_kernel, dim3((unsigned)tiles), dim3(256), 0, s, Ap, lda, n, c0, nbk, c1, T, n, ntile);
  }
}

void ldlt_permute_sym(hipStream_t s, const double *S, long long lds, const long long *q_dev, long long n, double *Ap,
                      long long lda) {
  long long chunks = (n + 255) / 256;
  if (chunks > 32) chunks = 32;
  hipLaunchKernelGGL(ldlt_permute_sym_kernel, dim3((unsigned)chunks, (unsigned)n), dim3(256), 0, s, S, lds, q_dev, n, Ap, lda);
}

// ---- solve -------------------------------------------------------------------------------------
// P b and P^T b through the permutation q the transpositions compose to (position i of P b holds b[q[i]]):
// forward: W[i, j] = R[q[i], j];  backward: R[q[i], j] = W[i, j]
__global__ __launch_bounds__(256) void ldlt_permute_kernel(double *W, const double *R_in, double *R_out, long long ld,
                                                           long long n, const long long *__restrict__ q, int backward, long long nrhs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const long long qi = q[i];
  for (long long j = blockIdx.y; j < nrhs; j += gridDim.y) {  // (the y extent of a grid ends at 65535)
    if (!backward) W[i + j * ld] = R_in[qi + j * ld];
    else R_out[qi + j * ld] = W[i + j * ld];
  }
}

// unit-lower (TRANS = false) / unit-upper L^T (TRANS = true) substitution against one LB x LB diagonal
// block held in LDS; one WAVE per right-hand side, lane = row.  Both directions run column by column:
// once x_j is final its contribution L[:, j] x_j (forward) or L[j, :]^T x_j (backward) leaves all other rows.
template <bool TRANS>
__global__ __launch_bounds__(256) void ldlt_diag_solve_kernel(const double *__restrict__ A, long long lda, long long k0,
                                                              int nb, double *W, long long ldw, long long nrhs) {
  __shared__ double Lc[LB * LB];  // Lc[j * LB + i] = L[i][j] (i > j), zero elsewhere: column j contiguous over lanes
  for (int e = threadIdx.x; e < LB * LB; e += 256) {
    const int i = e % LB, j = e / LB;
    Lc[e] = (i > j && i < nb && j < nb) ? A[(k0 + i) + (k0 + j) * lda] : 0.;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long col = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= nrhs) return;
  double *b = W + col * ldw + k0;
  double x = lane < nb ? b[lane] : 0.;
  if (!TRANS) {
    for (int j = 0; j < nb; ++j) {
      const double xj = __shfl(x, j, 64);
      x -= Lc[j * LB + lane] * xj;  // zero for lane <= j
    }
  } else {
    // x_j -= sum_{i > j} L[i][j] x_i : when x_i is final (i descending), row i of L leaves every j < i
    for (int i = nb - 1; i > 0; --i) {
      const double xi = __shfl(x, i, 64);
      const double lij = lane < i ? Lc[lane * LB + i] : 0.;  // L[i][lane]
      x -= lij * xi;
    }
  }
  if (lane < nb) b[lane] = x;
}

// D^+ : rows whose |D| is not above the smallest normal number become zero (Eigen's solve)
__global__ __launch_bounds__(256) void ldlt_dscale_kernel(const double *__restrict__ A, long long lda, long long n,
                                                          double *W, long long ldw, long long nrhs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double d = A[i + i * lda];
  const bool keep = fabs(d) > 2.2250738585072014e-308;
  for (long long j = blockIdx.y; j < nrhs; j += gridDim.y) {
    double *w = W + j * ldw + i;
    *w = keep ? *w / d : 0.;
  }
}

// D^-1/2 : 1 / sqrt(D_i) where D_i > 0, else 0 (diagonal_sqrt_inverse, serializable_ldlt.hpp:58-69)
__global__ __launch_bounds__(256) void ldlt_dsqrt_scale_kernel(const double *__restrict__ A, long long lda, long long n,
                                                               double *W, long long ldw, long long nrhs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double d = A[i + i * lda];
  const double f = d > 0. ? 1. / sqrt(d) : 0.;
  for (long long j = blockIdx.y; j < nrhs; j += gridDim.y) {
    double *w = W + j * ldw + i;
    *w = *w * f;
  }
}

// sqrt_solve (serializable_ldlt.hpp:99-109): W <- D^-1/2 L^-1 P R.  R (n x nrhs, ldw) is only read.
void ldlt_sqrt_solve(hipStream_t s, const double *A, long long lda, long long n, const long long *q_dev, double *W,
                     const double *R, long long ldw, long long nrhs) {
  if (n <= 0 || nrhs <= 0) return;
  const unsigned cgrid = (unsigned)((nrhs + 3) / 4);
  const dim3 pgrid((unsigned)((n + 255) / 256), (unsigned)(nrhs < 4096 ? nrhs : 4096));
  hipLaunchKernelGGL(ldlt_permute_kernel, pgrid, dim3(256), 0, s, W, R, (double *)nullptr, ldw, n, q_dev, 0, nrhs);
  for (long long k = 0; k < n; k += LB) {
    const int nb = (int)((n - k < LB) ? n - k : LB);
    hipLaunchKernelGGL((ldlt_diag_solve_kernel<false>), dim3(cgrid), dim3(256), 0, s, A, lda, k, nb, W, ldw, nrhs);
    const long long rows = n - (k + nb);
    if (rows > 0)
      launch_gemm_nt_sub(s, W + k + nb, ldw, A + k * lda + (k + nb), lda, false, W + k, ldw, true, rows, nrhs, nb, false);
  }
  const unsigned gy = (unsigned)(nrhs < 64 ? nrhs : 64);
  hipLaunchKernelGGL(ldlt_dsqrt_scale_kernel, dim3((unsigned)((n + 255) / 256), gy), dim3(256), 0, s, A, lda, n, W, ldw, nrhs);
}

// R (n x nrhs, ldw) holds the right-hand sides on entry and the solution on return; W is scratch of the same shape
void ldlt_solve(hipStream_t s, const double *A, long long lda, long long n, const long long *q_dev, double *W,
                double *R, long long ldw, long long nrhs) {
  if (n <= 0 || nrhs <= 0) return;
  const unsigned cgrid = (unsigned)((nrhs + 3) / 4);  // one wave per right-hand side
  const dim3 pgrid((unsigned)((n + 255) / 256), (unsigned)(nrhs < 4096 ? nrhs : 4096));
  hipLaunchKernelGGL(ldlt_permute_kernel, pgrid, dim3(256), 0, s, W, R, R, ldw, n, q_dev, 0, nrhs);
  for (long long k = 0; k < n; k += LB) {  // L^-1
    const int nb = (int)((n - k < LB) ? n - k : LB);
    hipLaunchKernelGGL((ldlt_diag_solve_kernel<false>), dim3(cgrid), dim3(256), 0, s, A, lda, k, nb, W, ldw, nrhs);
    const long long rows = n - (k + nb);
    if (rows > 0)  // W[k + nb :] -= L[k + nb :, k : k + nb] W[k : k + nb]
      launch_gemm_nt_sub(s, W + k + nb, ldw, A + k * lda + (k + nb), lda, false, W + k, ldw, true, rows, nrhs, nb, false);
  }
  {
    unsigned gy = (unsigned)(nrhs < 64 ? nrhs : 64);
    hipLaunchKernelGGL(ldlt_dscale_kernel, dim3((unsigned)((n + 255) / 256), gy), dim3(256), 0, s, A, lda, n, W, ldw, nrhs);
  }
  const long long nblk = (n + LB - 1) / LB;
  for (long long b = nblk - 1; b >= 0; --b) {  // L^-T, right-looking: a finished block leaves all rows above it
    const long long k = b * LB;
    const int nb = (int)((n - k < LB) ? n - k : LB);
    hipLaunchKernelGGL((ldlt_diag_solve_kernel<true>), dim3(cgrid), dim3(256), 0, s, A, lda, k, nb, W, ldw, nrhs);
    if (k > 0)  // W[0 : k] -= L[k : k + nb, 0 : k]^T W[k : k + nb]   (short K = nb, many tiles)
      launch_gemm_nt_sub(s, W, ldw, A + k, lda, true, W + k, ldw, true, k, nrhs, nb, false);
  }
  hipLaunchKernelGGL(ldlt_permute_kernel, pgrid, dim3(256), 0, s, W, R, R, ldw, n, q_dev, 1, nrhs);
}

}  // namespace agp
