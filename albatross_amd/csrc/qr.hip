// qr.hip — Householder QR with column pivoting, and the substitutions against its R factor.
//
// Replaces Eigen::ColPivHouseholderQR as the reference's DenseQRImplementation uses it
// (include/albatross/src/models/sparse_gp.hpp:80-88; get_R / get_P, linalg/qr_utils.hpp:18-53) on the paths where the
// matrix is rank deficient by construction and the un-pivoted CholeskyQR2 of sparse_api.hip cannot be used:
// fit_from_prediction / rebase_inducing_points (sparse_gp.hpp:406-461, 714-725) and updates of such fits (:322-371).
//   for k: bring the remaining column of largest norm (rows k..) forward, Householder vector of column k,
//          apply I - tau v v^T to the columns behind it.
// Eigen down-dates the column norms; here the apply kernel recomputes them over rows k+1.. while it has the column in
// hand (same pivots except on ties at rounding level).
// `extra` columns behind the `cols` pivoted ones are carried along un-pivoted (a right-hand side: its first rows end up
// as Q^T y).  Two launches per column, level-2 intensity: a correctness path for moderate sizes, like ldlt.hip.
#include "common.h"

namespace agp {

namespace {

constexpr double QR_EPS = 2.220446049250313e-16;
constexpr double QR_DBL_MIN = 2.2250738585072014e-308;

// state[0] = threshold_helper, state[1] = maxpivot, state[2] = nonzero_pivots, state[3] = rank (as doubles)
__global__ __launch_bounds__(256) void qr_norms_kernel(const double *__restrict__ A, long long lda, long long rows, long long row0,
                                                       double *norms) {
  __shared__ double red[256];
  const double *c = A + (long long)blockIdx.x * lda;
  double s = 0.;
  for (long long i = row0 + threadIdx.x; i < rows; i += 256) s += c[i] * c[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) norms[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void qr_init_kernel(const double *__restrict__ norms, long long rows, long long cols, double *state) {
  __shared__ double red[256];
  double mx = 0.;
  for (long long j = threadIdx.x; j < cols; j += 256) mx = fmax(mx, sqrt(norms[j]));
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + w]);
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const double t = red[0] * QR_EPS / (double)rows;
    state[0] = t * t;
    state[1] = 0.;
    state[2] = (double)cols;
    state[3] = 0.;
  }
}

// one workgroup: pivot search, column swap, Householder vector of column k
__global__ __launch_bounds__(1024) void qr_pivot_kernel(double *A, long long lda, long long rows, long long cols, long long extra,
                                                        long long k, double *norms, double *tau, long long *perm, double *state) {
  __shared__ double rv[1024];
  __shared__ long long ri[1024];
  __shared__ double sh[4];
  const int tid = threadIdx.x;
  // first index of the largest remaining squared norm
  double bv = -1.;
  long long bi = cols;
  for (long long j = k + tid; j < cols; j += 1024) {
    const double v = norms[j];
    if (v > bv) { bv = v; bi = j; }
  }
  rv[tid] = bv; ri[tid] = bi;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (tid < w) {
      const double v = rv[tid + w];
      const long long i2 = ri[tid + w];
      if (v > rv[tid] || (v == rv[tid] && i2 < ri[tid])) { rv[tid] = v; ri[tid] = i2; }
    }
    __syncthreads();
  }
  const long long best = ri[0];
  const double best_sq = rv[0];
  if (tid == 0) {
    if (state[2] == (double)cols && best_sq < state[0] * (double)(rows - k)) state[2] = (double)k;
    if (best != k) {
      const long long t = perm[k]; perm[k] = perm[best]; perm[best] = t;
      norms[best] = norms[k];
    }
  }
  double *ck = A + k * lda;
  if (best != k) {
    double *cb = A + best * lda;
    for (long long i = tid; i < rows; i += 1024) {
      const double t = ck[i]; ck[i] = cb[i]; cb[i] = t;
    }
  }
  __syncthreads();
  // makeHouseholderInPlace on x = A[k:, k]
  double *x = ck + k;
  const long long len = rows - k;
  double s = 0.;
  for (long long i = 1 + tid; i < len; i += 1024) s += x[i] * x[i];
  rv[tid] = s;
  __syncthreads();
  for (int w = 512; w > 0; w >>= 1) {
    if (tid < w) rv[tid] += rv[tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    const double tail = rv[0], x0 = x[0];
    double beta, t;
    if (tail <= QR_DBL_MIN) {
      t = 0.; beta = x0;
      sh[2] = 0.;  // the essential part becomes zero
    } else {
      beta = sqrt(x0 * x0 + tail);
      if (x0 >= 0.) beta = -beta;
      t = (beta - x0) / beta;
      sh[2] = 1.;
    }
    sh[0] = beta; sh[1] = x0;
    tau[k] = t;
    if (fabs(beta) > state[1]) state[1] = fabs(beta);
  }
  __syncthreads();
  const double beta = sh[0], x0 = sh[1];
  if (sh[2] == 0.) {
    for (long long i = 1 + tid; i < len; i += 1024) x[i] = 0.;
  } else {
    const double d = x0 - beta;
    for (long long i = 1 + tid; i < len; i += 1024) x[i] /= d;
  }
  if (tid == 0) x[0] = beta;
}

// one workgroup per trailing column j = k + 1 + blockIdx.x: c <- (I - tau v v^T) c, then the squared norm of c[k+1:]
__global__ __launch_bounds__(256) void qr_apply_kernel(double *A, long long lda, long long rows, long long cols, long long k,
                                                       const double *__restrict__ tau, double *norms) {
  __shared__ double red[256];
  const long long j = k + 1 + blockIdx.x;
  const double *v = A + k * lda + k;
  double *c = A + j * lda + k;
  const long long len = rows - k;
  const int tid = threadIdx.x;
  double s = 0.;
  for (long long i = 1 + tid; i < len; i += 256) s += v[i] * c[i];
  red[tid] = s;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  const double dot = (c[0] + red[0]) * tau[k];
  __syncthreads();
  double nn = 0.;
  for (long long i = 1 + tid; i < len; i += 256) {
    const double t = c[i] - dot * v[i];
    c[i] = t;
    nn += t * t;
  }
  red[tid] = nn;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  if (tid == 0) {
    c[0] -= dot;
    if (j < cols) norms[j] = red[0];
  }
}

// rank(): |R_ii| > maxpivot * eps * diagSize
__global__ __launch_bounds__(256) void qr_rank_kernel(const double *__restrict__ A, long long lda, long long cols, long long diag_size,
                                                      double *state) {
  __shared__ int cnt[256];
  const double thr = state[1] * QR_EPS * (double)diag_size;
  int c = 0;
  for (long long i = threadIdx.x; i < cols; i += 256) c += fabs(A[i + i * lda]) > thr ? 1 : 0;
  cnt[threadIdx.x] = c;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if ((int)threadIdx.x < w) cnt[threadIdx.x] += cnt[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) state[3] = (double)cnt[0];
}

__global__ __launch_bounds__(256) void qr_iota_kernel(long long *perm, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) perm[i] = i;
}

// R (m x ldr, zero below the diagonal) <- upper triangle of the factored matrix; inflate: added to the diagonal
__global__ __launch_bounds__(256) void qr_extract_r_kernel(const double *__restrict__ A, long long lda, long long m, double *R,
                                                           long long ldr, double inflate) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x, j = blockIdx.y;
  if (i >= m) return;
  R[i + j * ldr] = i < j ? A[i + j * lda] : (i == j ? A[i + j * lda] + inflate : 0.);
}

// T = P R^T (m x ldt): T[perm[i], r] = R[r, i]  -  the square root of Sigma^-1 = P R^T R P^T the update stacks on
__global__ __launch_bounds__(256) void qr_root_kernel(const double *__restrict__ R, long long ldr, const long long *__restrict__ perm,
                                                      long long m, double *T, long long ldt) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x, i = blockIdx.y;
  if (r >= m) return;
  T[perm[i] + r * ldt] = R[r + i * ldr];
}

// W[i, c] = X[perm[i], c]   (P^T X)
__global__ __launch_bounds__(256) void qr_permute_rows_kernel(const double *__restrict__ X, long long ldx, const long long *__restrict__ perm,
                                                              long long m, double *W, long long ldw, long long nrhs) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  const long long src = perm[i];
  for (long long c = blockIdx.y; c < nrhs; c += gridDim.y) W[i + c * ldw] = X[src + c * ldx];
}

// forward substitution against one QB x QB diagonal block of the lower-triangular R^T (L[i][j] = R[j][i], non-unit),
// one wave per right-hand side, lane = row (the scheme of ldlt_diag_solve_kernel)
constexpr int QB = 64;
__global__ __launch_bounds__(256) void qr_rt_diag_solve_kernel(const double *__restrict__ R, long long ldr, long long k0, int nb,
                                                               double *W, long long ldw, long long nrhs) {
  __shared__ double Lc[QB * QB];  // Lc[j * QB + i] = L[i][j] = R[k0 + j][k0 + i] for i >= j
  for (int e = threadIdx.x; e < QB * QB; e += 256) {
    const int i = e % QB, j = e / QB;
    Lc[e] = (i >= j && i < nb && j < nb) ? R[(k0 + j) + (k0 + i) * ldr] : 0.;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const long long col = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (col >= nrhs) return;
  double *b = W + col * ldw + k0;
  double x = lane < nb ? b[lane] : 0.;
  for (int j = 0; j < nb; ++j) {
    const double d = Lc[j * QB + j];
    const double xj = __shfl(x, j, 64) / d;
    if (lane == j) x = xj;
    else if (lane > j) x -= Lc[j * QB + lane] * xj;
  }
  if (lane < nb) b[lane] = x;
}

// z = P [R11^-1 c[0:np]; 0] for ONE vector (ColPivHouseholderQR::solve after Q^T has been applied): one workgroup,
// column-oriented back substitution
__global__ __launch_bounds__(1024) void qr_back_solve_kernel(const double *__restrict__ R, long long ldr, const long long *__restrict__ perm,
                                                             long long m, long long np, double *c, double *out) {
  __shared__ double xi;
  const int tid = threadIdx.x;
  for (long long i = np - 1; i >= 0; --i) {
    if (tid == 0) {
      xi = c[i] / R[i + i * ldr];
      c[i] = xi;
    }
    __syncthreads();
    const double x = xi;
    for (long long r = tid; r < i; r += 1024) c[r] -= R[r + i * ldr] * x;
    __syncthreads();
  }
  for (long long i = tid; i < m; i += 1024) out[perm[i]] = i < np ? c[i] : 0.;
}

}  // namespace

// A (rows x (cols + extra), lda) is factored in place: R in the upper triangle of the first `cols` columns, the
// Householder vectors below; perm[k] = original index of the column at position k; tau (cols); norms (cols) scratch;
// state (4 doubles, device): threshold helper, max |pivot|, nonzero_pivots, rank
void colpiv_qr(hipStream_t s, double *A, long long lda, long long rows, long long cols, long long extra, double *tau,
               long long *perm, double *norms, double *state) {
  if (rows <= 0 || cols <= 0) return;
  hipLaunchKernelGGL(qr_iota_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, s, perm, cols);
  hipLaunchKernelGGL(qr_norms_kernel, dim3((unsigned)cols), dim3(256), 0, s, A, lda, rows, 0ll, norms);
  hipLaunchKernelGGL(qr_init_kernel, dim3(1), dim3(256), 0, s, norms, rows, cols, state);
  const long long steps = cols < rows ? cols : rows;
  for (long long k = 0; k < steps; ++k) {
    hipLaunchKernelGGL(qr_pivot_kernel, dim3(1), dim3(1024), 0, s, A, lda, rows, cols, extra, k, norms, tau, perm, state);
    const long long trailing = cols + extra - (k + 1);
    if (trailing > 0)
      hipLaunchKernelGGL(qr_apply_kernel, dim3((unsigned)trailing), dim3(256), 0, s, A, lda, rows, cols, k, tau, norms);
  }
  hipLaunchKernelGGL(qr_rank_kernel, dim3(1), dim3(256), 0, s, A, lda, steps, steps, state);
}

void qr_extract_r(hipStream_t s, const double *A, long long lda, long long m, double *R, long long ldr, double inflate) {
  hipLaunchKernelGGL(qr_extract_r_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, s, A, lda, m, R, ldr,
                     inflate);
}

void qr_root(hipStream_t s, const double *R, long long ldr, const long long *perm, long long m, double *T, long long ldt) {
  (void)hipMemsetAsync(T, 0, sizeof(double) * (size_t)ldt * (size_t)m, s);
  hipLaunchKernelGGL(qr_root_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, s, R, ldr, perm, m, T, ldt);
}

// sqrt_solve(R, P, X) = R^-T P^T X (linalg/qr_utils.hpp:37-45): X (m x nrhs, ldx) -> W (m x nrhs, ldw)
void qr_sqrt_solve(hipStream_t s, const double *R, long long ldr, const long long *perm, long long m, const double *X,
                   long long ldx, double *W, long long ldw, long long nrhs) {
  if (m <= 0 || nrhs <= 0) return;
  hipLaunchKernelGGL(qr_permute_rows_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)(nrhs < 4096 ? nrhs : 4096)), dim3(256), 0,
                     s, X, ldx, perm, m, W, ldw, nrhs);
  const unsigned cgrid = (unsigned)((nrhs + 3) / 4);
  for (long long k = 0; k < m; k += QB) {
    const int nb = (int)((m - k < QB) ? m - k : QB);
    hipLaunchKernelGGL(qr_rt_diag_solve_kernel, dim3(cgrid), dim3(256), 0, s, R, ldr, k, nb, W, ldw, nrhs);
    const long long below = m - (k + nb);
    if (below > 0)  // W[k + nb :] -= R[k : k + nb, k + nb :]^T W[k : k + nb]
      launch_gemm_nt_sub(s, W + k + nb, ldw, R + k + (k + nb) * ldr, ldr, true, W + k, ldw, true, below, nrhs, nb, false);
  }
}

void qr_back_solve(hipStream_t s, const double *R, long long ldr, const long long *perm, long long m, long long np, double *c,
                   double *out) {
  hipLaunchKernelGGL(qr_back_solve_kernel, dim3(1), dim3(1024), 0, s, R, ldr, perm, m, np, c, out);
}

}  // namespace agp
