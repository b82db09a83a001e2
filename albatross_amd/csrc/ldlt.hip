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

// ---- solve -------------------------------------------------------------------------------------
// rows of every column swapped in the order of the transpositions (forward: P b, backward: P^T b)
__global__ __launch_bounds__(256) void ldlt_permute_kernel(double *W, long long ldw, long long n, long long nrhs,
                                                           const long long *__restrict__ tr, int backward) {
  const long long j = (long long)blockIdx.x * 256 + threadIdx.x;
  if (j >= nrhs) return;
  double *b = W + j * ldw;
  if (!backward) {
    for (long long k = 0; k < n; ++k) {
      const long long p = tr[k];
      if (p != k) { const double t = b[k]; b[k] = b[p]; b[p] = t; }
    }
  } else {
    for (long long k = n - 1; k >= 0; --k) {
      const long long p = tr[k];
      if (p != k) { const double t = b[k]; b[k] = b[p]; b[p] = t; }
    }
  }
}

// unit-lower (TRANS = false) / unit-upper L^T (TRANS = true) substitution against one LB x LB diagonal
// block held in LDS; one thread per right-hand side
template <bool TRANS>
__global__ __launch_bounds__(256) void ldlt_diag_solve_kernel(const double *__restrict__ A, long long lda, long long k0,
                                                              int nb, double *W, long long ldw, long long nrhs) {
  __shared__ double L[LB * LB];
  for (int e = threadIdx.x; e < LB * LB; e += 256) {  // the whole LDS block: rows / columns beyond nb are zero
    const int i = e % LB, j = e / LB;
    L[e] = (i > j && i < nb && j < nb) ? A[(k0 + i) + (k0 + j) * lda] : 0.;
  }
  __syncthreads();
  const long long col = (long long)blockIdx.x * 256 + threadIdx.x;
  if (col >= nrhs) return;
  double *b = W + col * ldw + k0;
  double x[LB];
#pragma unroll
  for (int i = 0; i < LB; ++i) x[i] = i < nb ? b[i] : 0.;
  if (!TRANS) {
#pragma unroll
    for (int j = 0; j < LB; ++j) {
      const double xj = x[j];
#pragma unroll
      for (int i = j + 1; i < LB; ++i) x[i] -= L[i + j * LB] * xj;
    }
  } else {
#pragma unroll
    for (int j = LB - 1; j >= 0; --j) {
      double sacc = x[j];
#pragma unroll
      for (int i = j + 1; i < LB; ++i) sacc -= L[i + j * LB] * x[i];
      x[j] = sacc;
    }
  }
#pragma unroll
  for (int i = 0; i < LB; ++i)
    if (i < nb) b[i] = x[i];
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

void ldlt_solve(hipStream_t s, const double *A, long long lda, long long n, const long long *tr_dev, double *W,
                long long ldw, long long nrhs) {
  if (n <= 0 || nrhs <= 0) return;
  const unsigned cgrid = (unsigned)((nrhs + 255) / 256);
  hipLaunchKernelGGL(ldlt_permute_kernel, dim3(cgrid), dim3(256), 0, s, W, ldw, n, nrhs, tr_dev, 0);
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
  for (long long b = nblk - 1; b >= 0; --b) {  // L^-T
    const long long k = b * LB;
    const int nb = (int)((n - k < LB) ? n - k : LB);
    const long long rows = n - (k + nb);
    if (rows > 0)  // W[k : k + nb] -= L[k + nb :, k : k + nb]^T W[k + nb :]
      launch_gemm_nt_sub(s, W + k, ldw, A + k * lda + (k + nb), lda, true, W + k + nb, ldw, true, nb, nrhs, rows, false);
    hipLaunchKernelGGL((ldlt_diag_solve_kernel<true>), dim3(cgrid), dim3(256), 0, s, A, lda, k, nb, W, ldw, nrhs);
  }
  hipLaunchKernelGGL(ldlt_permute_kernel, dim3(cgrid), dim3(256), 0, s, W, ldw, n, nrhs, tr_dev, 1);
}

}  // namespace agp
