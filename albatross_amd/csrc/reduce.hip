// reduce.hip — K5: small bandwidth-bound reductions of the predict / NLL path.
#include "common.h"

namespace agp {

// out[j] = base[j] - scale * sum_i A[i,j] * B[i,j]
// gp_marginal_prediction's explained variance (models/gp.hpp:96-99); with
// A = B = V = L^-1 K* it is k** - colsum(V o V).
__global__ __launch_bounds__(256) void coldot_kernel(const double *__restrict__ A, long long lda,
                                                     const double *__restrict__ B, long long ldb, long long n,
                                                     double *__restrict__ out, double scale,
                                                     const double *__restrict__ base) {
  __shared__ double red[4];
  const long long j = blockIdx.x;
  const double *a = A + j * lda, *b = B + j * ldb;
  double acc = 0.;
  for (long long i = threadIdx.x; i < n; i += 256) acc += a[i] * b[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    const double s = (red[0] + red[1]) + (red[2] + red[3]);
    out[j] = (base ? base[j] : 0.) - scale * s;
  }
}

void launch_coldot(hipStream_t s, const double *A, long long lda, const double *B, long long ldb, long long n,
                   long long m, double *out, double scale, const double *base) {
  if (m <= 0) return;
  hipLaunchKernelGGL(coldot_kernel, dim3((unsigned)m), dim3(256), 0, s, A, lda, B, ldb, n, out, scale, base);
}

// out[0] = sum_i a_i b_i   (single workgroup, deterministic order)
__global__ __launch_bounds__(1024) void dot_kernel(const double *__restrict__ a, const double *__restrict__ b,
                                                   long long n, double *__restrict__ out) {
  __shared__ double red[16];
  double acc = 0.;
  for (long long i = threadIdx.x; i < n; i += 1024) acc += a[i] * b[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double s = 0.;
    for (int w = 0; w < 16; ++w) s += red[w];
    out[0] = s;
  }
}

void launch_dot(hipStream_t s, const double *a, const double *b, long long n, double *out) {
  hipLaunchKernelGGL(dot_kernel, dim3(1), dim3(1024), 0, s, a, b, n, out);
}

// Mirror the lower triangle of a column-major n x n matrix into the upper one.
__global__ __launch_bounds__(256) void symmetrize_kernel(double *A, long long ld, long long n) {
  __shared__ double tile[32][33];
  const long long bi = blockIdx.x, bj = blockIdx.y;
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const long long row = bi * 32 + tx, col = bj * 32 + r;
    tile[r][tx] = (row < n && col < n) ? A[col * ld + row] : 0.;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    // write A[bj*32 + tx, bi*32 + r] = tile element (row = bi*32 + r, col = bj*32 + tx)
    const long long row = bj * 32 + tx, col = bi * 32 + r;
    if (row < n && col < n && row < col) A[col * ld + row] = tile[tx][r];
  }
}

void launch_symmetrize(hipStream_t s, double *A, long long ld, long long n) {
  if (n <= 0) return;
  const unsigned nb = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(symmetrize_kernel, dim3(nb, nb), dim3(256), 0, s, A, ld, n);
}

// zero the strictly-upper triangle (factor download)
__global__ __launch_bounds__(256) void zero_upper_kernel(double *A, long long ld, long long n) {
  const long long col = blockIdx.x;
  for (long long r = threadIdx.x; r < col && r < n; r += 256) A[col * ld + r] = 0.;
}

void launch_zero_upper(hipStream_t s, double *A, long long ld, long long n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(zero_upper_kernel, dim3((unsigned)n), dim3(256), 0, s, A, ld, n);
}

// dst lower triangle (ld_dst) <- transpose of the upper triangle of src (ld_src)
__global__ __launch_bounds__(256) void upper_to_lower_kernel(const double *__restrict__ src, long long ld_src,
                                                             double *__restrict__ dst, long long ld_dst, long long n) {
  __shared__ double tile[32][33];
  const long long bi = blockIdx.x, bj = blockIdx.y;  // destination tile (rows bi, cols bj), bi >= bj
  if (bj > bi) return;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    // source element (row = bj*32 + tx, col = bi*32 + r): upper part since bj <= bi
    const long long row = bj * 32 + tx, col = bi * 32 + r;
    tile[r][tx] = (row < n && col < n) ? src[col * ld_src + row] : 0.;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    // destination element (row = bi*32 + tx, col = bj*32 + r) = source (col', row') transposed
    const long long row = bi * 32 + tx, col = bj * 32 + r;
    if (row < n && col < n && row >= col) dst[col * ld_dst + row] = tile[tx][r];
  }
}

void launch_upper_to_lower(hipStream_t s, const double *src, long long ld_src, double *dst, long long ld_dst,
                           long long n) {
  if (n <= 0) return;
  const unsigned nb = (unsigned)((n + 31) / 32);
  hipLaunchKernelGGL(upper_to_lower_kernel, dim3(nb, nb), dim3(256), 0, s, src, ld_src, dst, ld_dst, n);
}

// NaN scan of the lower triangle (ALBATROSS_ASSERT(!cov.hasNaN()), gp.hpp:66)
__global__ __launch_bounds__(256) void nan_scan_lower_kernel(const double *A, long long ld, long long n, int *flag) {
  const long long col = blockIdx.x;
  bool bad = false;
  for (long long r = col + threadIdx.x; r < n; r += 256) {
    const double v = A[col * ld + r];
    bad = bad || (v != v);
  }
  if (bad) atomicOr(flag, 1);
}

void launch_nan_scan_lower(hipStream_t s, const double *A, long long ld, long long n, int *flag) {
  if (n <= 0) return;
  hipLaunchKernelGGL(nan_scan_lower_kernel, dim3((unsigned)n), dim3(256), 0, s, A, ld, n, flag);
}

// B (n x n, ld) <- identity
__global__ __launch_bounds__(256) void set_identity_kernel(double *B, long long ld, long long n) {
  const long long col = blockIdx.x;
  for (long long r = threadIdx.x; r < n; r += 256) B[col * ld + r] = (r == col) ? 1. : 0.;
}

void launch_set_identity(hipStream_t s, double *B, long long ld, long long n) {
  if (n <= 0) return;
  hipLaunchKernelGGL(set_identity_kernel, dim3((unsigned)n), dim3(256), 0, s, B, ld, n);
}

// leave-one-out marginals from diag(K^-1): cross_validation_utils.hpp:138-163,171-186
// (variance may alias kinv_diag: every thread reads its element before writing it)
__global__ __launch_bounds__(256) void loo_kernel(const double *kinv_diag, const double *__restrict__ y,
                                                  const double *__restrict__ information, long long n,
                                                  double *__restrict__ mean, double *variance) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const double d = kinv_diag[i];
  mean[i] = y[i] - information[i] / d;  // y - A_ldlt.solve(v) for a 1 x 1 block
  variance[i] = 1. / d;                 // leave_one_out_conditional_variance
}

void launch_loo(hipStream_t s, const double *kinv_diag, const double *y, const double *information, long long n,
                double *mean, double *variance) {
  if (n <= 0) return;
  hipLaunchKernelGGL(loo_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, kinv_diag, y, information, n, mean,
                     variance);
}

// G[i, a] = R[i, idx[a]] for i >= row0 (columns of the inverse Cholesky factor of one index group)
__global__ __launch_bounds__(256) void gather_cols_kernel(const double *__restrict__ R, long long ldr,
                                                          const long long *__restrict__ idx, long long row0,
                                                          long long n, double *__restrict__ G, long long ldg) {
  const long long a = blockIdx.y;
  const long long j = idx[a];  // < 0: padding column, zero-filled
  const double *src = R + (j < 0 ? 0 : j) * ldr;
  double *dst = G + a * ldg;
  for (long long i = row0 + (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
    dst[i] = j < 0 ? 0. : src[i];
}

void launch_gather_cols(hipStream_t s, const double *R, long long ldr, const long long *idx, long long m,
                        long long row0, long long n, double *G, long long ldg) {
  if (m <= 0 || n <= row0) return;
  long long chunks = (n - row0 + 255) / 256;
  if (chunks > 64) chunks = 64;
  hipLaunchKernelGGL(gather_cols_kernel, dim3((unsigned)chunks, (unsigned)m), dim3(256), 0, s, R, ldr, idx, row0, n, G, ldg);
}

// out[a] = base ? base[idx[a]] - sub[a] : src[idx[a]]     (subset(v, indices); y - solve(v))
__global__ __launch_bounds__(256) void gather_vec_kernel(const double *__restrict__ src, const long long *__restrict__ idx,
                                                         long long m, const double *__restrict__ sub,
                                                         double *__restrict__ out) {
  const long long a = (long long)blockIdx.x * 256 + threadIdx.x;
  if (a >= m) return;
  const long long j = idx[a];  // < 0: padding entry
  const double v = j < 0 ? 0. : src[j];
  out[a] = sub ? v - sub[a] : v;
}

void launch_gather_vec(hipStream_t s, const double *src, const long long *idx, long long m, const double *sub,
                       double *out) {
  if (m <= 0) return;
  hipLaunchKernelGGL(gather_vec_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, src, idx, m, sub, out);
}

// A (m x m, ld) <- -A ; optionally diag_out[i] = new A[i, i]
__global__ __launch_bounds__(256) void negate_kernel(double *A, long long ld, long long m, double *diag_out) {
  const long long col = blockIdx.x;
  for (long long r = threadIdx.x; r < m; r += 256) {
    const double v = -A[col * ld + r];
    A[col * ld + r] = v;
    if (diag_out && r == col) diag_out[col] = v;
  }
}

void launch_negate(hipStream_t s, double *A, long long ld, long long m, double *diag_out) {
  if (m <= 0) return;
  hipLaunchKernelGGL(negate_kernel, dim3((unsigned)m), dim3(256), 0, s, A, ld, m, diag_out);
}

// ---- small dense helpers of the sparse-GP path (W is m x n column-major, m << n) ---------------
// partial[chunk * m + i] = sum_{j in chunk} W[i, j] x[j]   (rows across threads: coalesced)
__global__ __launch_bounds__(256) void matvec_partial_kernel(const double *__restrict__ W, long long ld, long long m,
                                                             long long n, long long chunk,
                                                             const double *__restrict__ x,
                                                             double *__restrict__ partial) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  const long long j0 = (long long)blockIdx.y * chunk;
  long long j1 = j0 + chunk;
  if (j1 > n) j1 = n;
  if (i >= m) return;
  double a0 = 0., a1 = 0.;
  long long j = j0;
  for (; j + 1 < j1; j += 2) {
    a0 += W[i + j * ld] * x[j];
    a1 += W[i + (j + 1) * ld] * x[j + 1];
  }
  if (j < j1) a0 += W[i + j * ld] * x[j];
  partial[(long long)blockIdx.y * m + i] = a0 + a1;
}

// out[i] = alpha * sum_c partial[c * m + i] + beta * base[i]   (fixed order: deterministic)
__global__ __launch_bounds__(256) void matvec_reduce_kernel(const double *__restrict__ partial, long long m,
                                                            long long chunks, double alpha, double beta,
                                                            const double *base, double *out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= m) return;
  double s = 0.;
  for (long long c = 0; c < chunks; ++c) s += partial[c * m + i];
  out[i] = alpha * s + (base ? beta * base[i] : 0.);
}

// out = alpha * W x + beta * base ; `partial` needs ceil(n / 1024) * m doubles
void launch_matvec(hipStream_t s, const double *W, long long ld, long long m, long long n, const double *x,
                   double *partial, double alpha, double beta, const double *base, double *out) {
  if (m <= 0) return;
  const long long chunk = 1024, chunks = n > 0 ? (n + chunk - 1) / chunk : 0;
  if (chunks > 0)
    hipLaunchKernelGGL(matvec_partial_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)chunks), dim3(256), 0, s, W, ld,
                       m, n, chunk, x, partial);
  hipLaunchKernelGGL(matvec_reduce_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, partial, m, chunks, alpha,
                     beta, base, out);
}

// LinearCombinationCaller (covariance_functions/callers.hpp:336-347): out(a, b) = c_a^T K[members(a), members(b)] c_b,
// evaluated like xs.coefficients.dot(mat * ys.coefficients): inner sum over the members of b, outer over those of a.
// members(a) = expanded points xoff[a] .. xoff[a + 1] (xoff == nullptr: a itself, coefficient 1).  symmetric: only a >= b
// is evaluated and mirrored, like the symmetric compute_covariance_matrix (callers.hpp:119-127).
__global__ __launch_bounds__(256) void contract_combinations_kernel(const double *__restrict__ K, long long ldk,
                                                                    const long long *xoff, const double *xc, long long na,
                                                                    const long long *yoff, const double *yc, long long nb,
                                                                    int symmetric, double *out, long long ldo) {
  const long long a = (long long)blockIdx.x * 16 + (threadIdx.x & 15), b = (long long)blockIdx.y * 16 + (threadIdx.x >> 4);
  if (a >= na || b >= nb || (symmetric && a < b)) return;
  const long long i0 = xoff ? xoff[a] : a, i1 = xoff ? xoff[a + 1] : a + 1;
  const long long j0 = yoff ? yoff[b] : b, j1 = yoff ? yoff[b + 1] : b + 1;
  double acc = 0.;
  for (long long i = i0; i < i1; ++i) {
    double t = 0.;
    for (long long j = j0; j < j1; ++j) t += K[i + j * ldk] * (yc ? yc[j] : 1.);
    acc += (xc ? xc[i] : 1.) * t;
  }
  out[a + b * ldo] = acc;
  if (symmetric && a != b) out[b + a * ldo] = acc;
}

void launch_contract_combinations(hipStream_t s, const double *K, long long ldk, const long long *xoff, const double *xc, long long na,
                                  const long long *yoff, const double *yc, long long nb, bool symmetric, double *out, long long ldo) {
  if (na <= 0 || nb <= 0) return;
  hipLaunchKernelGGL(contract_combinations_kernel, dim3((unsigned)((na + 15) / 16), (unsigned)((nb + 15) / 16)), dim3(256), 0, s, K, ldk,
                     xoff, xc, na, yoff, yc, nb, symmetric ? 1 : 0, out, ldo);
}

// out[i] = alpha * sum_{c < ncols} W[i, c] x[c] + beta * base[i] for a TALL matrix (rows >> ncols <= 2048): a workgroup
// owns 64 rows (lane = row: every load is a 512-B segment of one column), its 4 waves split the columns and keep 8
// loads in flight each; fixed-order reduction through LDS.  The right-looking vector substitutions are sequences of
// such products (rows below x 512).
template <typename T>  // T: the matrix's element type (double; float: the fp32 copy of a factor as a preconditioner)
__global__ __launch_bounds__(256) void tall_matvec_kernel(const T *__restrict__ W, long long ld, long long rows, int ncols,
                                                          const double *__restrict__ x, double alpha, double beta,
                                                          const double *base, double *out) {
  __shared__ double xs[2048], red[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < ncols; c += 256) xs[c] = x[c];
  __syncthreads();
  const long long i = (long long)blockIdx.x * 64 + lane;
  const int per = (ncols + 3) / 4, c0 = wave * per, c1 = (c0 + per < ncols) ? c0 + per : ncols;
  double acc = 0.;
  if (i < rows) {
    const T *p = W + i + (long long)c0 * ld;
    int c = c0;
    // (sixteen loads in flight per lane: a wave's 256 columns were 32 dependent round trips with eight - 25 us per launch
    // at 2.7 TB/s in the preconditioner sweeps of the mixed fit)
    for (; c + 16 <= c1; c += 16, p += 16 * ld) {
      T w[16];
#pragma unroll
      for (int u = 0; u < 16; ++u) w[u] = p[u * ld];
      double lo = 0., hi = 0.;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        lo += (double)w[u] * xs[c + u];
        hi += (double)w[8 + u] * xs[c + 8 + u];
      }
      acc += lo + hi;
    }
    for (; c + 8 <= c1; c += 8, p += 8 * ld) {
      const double v0 = p[0], v1 = p[ld], v2 = p[2 * ld], v3 = p[3 * ld], v4 = p[4 * ld], v5 = p[5 * ld], v6 = p[6 * ld], v7 = p[7 * ld];
      acc += ((v0 * xs[c] + v1 * xs[c + 1]) + (v2 * xs[c + 2] + v3 * xs[c + 3])) +
             ((v4 * xs[c + 4] + v5 * xs[c + 5]) + (v6 * xs[c + 6] + v7 * xs[c + 7]));
    }
    for (; c < c1; ++c, p += ld) acc += p[0] * xs[c];
  }
  red[wave][lane] = acc;
  __syncthreads();
  if (wave == 0 && i < rows) {
    const double sum = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
    out[i] = alpha * sum + (base ? beta * base[i] : 0.);
  }
}

void launch_tall_matvec(hipStream_t s, const double *W, long long ld, long long rows, long long ncols, const double *x, double alpha,
                        double beta, const double *base, double *out) {
  if (rows <= 0 || ncols <= 0 || ncols > 2048) return;
  hipLaunchKernelGGL(tall_matvec_kernel<double>, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, s, W, ld, rows, (int)ncols, x, alpha, beta,
                     base, out);
}
void launch_tall_matvec_f32(hipStream_t s, const float *W, long long ld, long long rows, long long ncols, const double *x, double alpha,
                            double beta, const double *base, double *out) {
  if (rows <= 0 || ncols <= 0 || ncols > 2048) return;
  hipLaunchKernelGGL(tall_matvec_kernel<float>, dim3((unsigned)((rows + 63) / 64)), dim3(256), 0, s, W, ld, rows, (int)ncols, x, alpha, beta,
                     base, out);
}

// out[j] = alpha * sum_i W[i, j] v[i] + beta * base[j]   (one workgroup per column)
// (blockIdx.y = batch entry; the strides are 0 for a single problem)
template <typename T>
__global__ __launch_bounds__(256) void colvec_dot_kernel(const T *__restrict__ W, long long ld, long long m,
                                                         const double *__restrict__ v, double alpha, double beta,
                                                         const double *base, double *out, long long stride_W,
                                                         long long stride_v) {
  __shared__ double red[4];
  const long long j = blockIdx.x;
  W += (long long)blockIdx.y * stride_W;
  v += (long long)blockIdx.y * stride_v;
  if (base) base += (long long)blockIdx.y * stride_v;
  out += (long long)blockIdx.y * stride_v;
  const T *w = W + j * ld;
  double acc = 0.;
  for (long long i = threadIdx.x; i < m; i += 256) acc += (double)w[i] * v[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[j] = alpha * ((red[0] + red[1]) + (red[2] + red[3])) + (base ? beta * base[j] : 0.);
}

void launch_colvec_dot(hipStream_t s, const double *W, long long ld, long long m, long long n, const double *v,
                       double alpha, double beta, const double *base, double *out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(colvec_dot_kernel<double>, dim3((unsigned)n), dim3(256), 0, s, W, ld, m, v, alpha, beta, base, out, 0LL, 0LL);
}
// The same with EIGHT columns per workgroup (a wave owns two, sixteen matrix loads in flight per lane, no barrier): one
// column of a 1024-row fp32 block is 4 KB - a workgroup per column spent its time starting and ending (27 us per launch
// at 2.5 TB/s in the backward sweep of the mixed fit's preconditioner).  Summation order differs from colvec_dot_kernel.
template <typename T>
__global__ __launch_bounds__(256) void colvec_dot8_kernel(const T *__restrict__ W, long long ld, long long m, long long n,
                                                          const double *__restrict__ v, double alpha, double beta,
                                                          const double *base, double *out) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long j0 = (long long)blockIdx.x * 8 + 2 * wave;
  if (j0 >= n) return;
  const bool two = j0 + 1 < n;
  const T *w0 = W + j0 * ld, *w1 = two ? w0 + ld : w0;
  double a0 = 0., a1 = 0.;
  for (long long i0 = lane; i0 < m; i0 += 512) {
    double x[8];
    T p[8], q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const long long i = i0 + 64 * u;
      const bool ok = i < m;
      x[u] = ok ? v[i] : 0.;
      p[u] = ok ? w0[i] : (T)0;
      q[u] = ok ? w1[i] : (T)0;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a0 += (double)p[u] * x[u];
      a1 += (double)q[u] * x[u];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    a0 += __shfl_down(a0, off, 64);
    a1 += __shfl_down(a1, off, 64);
  }
  if (lane == 0) {
    out[j0] = alpha * a0 + (base ? beta * base[j0] : 0.);
    if (two) out[j0 + 1] = alpha * a1 + (base ? beta * base[j0 + 1] : 0.);
  }
}

void launch_colvec_dot_f32(hipStream_t s, const float *W, long long ld, long long m, long long n, const double *v,
                           double alpha, double beta, const double *base, double *out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(colvec_dot8_kernel<float>, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, s, W, ld, m, n, v, alpha, beta, base, out);
}

// L32 (lower triangle incl. diagonal, ld) = (float) L: the fp32 copy of a factor that preconditions the refinement of a
// mixed-precision fit (api.hip: refine_information)
__global__ __launch_bounds__(256) void convert_lower_f32_kernel(const double *__restrict__ L, long long ld, long long n, float *__restrict__ L32) {
  const long long j = blockIdx.y;
  const long long i = j / 2 * 2 + ((long long)blockIdx.x * 256 + threadIdx.x) * 2;  // pairs from the even row at or above the diagonal
  if (i + 1 < n) {
    const double2 v = *reinterpret_cast<const double2 *>(L + i + j * ld);
    *reinterpret_cast<float2 *>(L32 + i + j * ld) = make_float2((float)v.x, (float)v.y);
  } else if (i < n) {
    L32[i + j * ld] = (float)L[i + j * ld];
  }
}
void launch_convert_lower_f32(hipStream_t s, const double *L, long long ld, long long n, float *L32) {
  if (n <= 0) return;
  // (one launch per 2048 columns keeps the idle blocks above the diagonal few)
  for (long long j0 = 0; j0 < n; j0 += 2048) {
    const long long cols = (n - j0 < 2048) ? n - j0 : 2048, rows = n - j0 / 2 * 2;
    hipLaunchKernelGGL(convert_lower_f32_kernel, dim3((unsigned)((rows + 511) / 512), (unsigned)cols), dim3(256), 0, s, L + j0 * ld + j0 / 2 * 2,
                       ld, n - j0 / 2 * 2, L32 + j0 * ld + j0 / 2 * 2);
  }
}

// dst (lower triangle incl. diagonal, ld) = src: the factor's working copy of a covariance matrix that is kept as well (the
// mixed-precision fit: api.hip) - a copy is 2 x 8 B per entry, evaluating the covariance function a second time was
// 3.6 ms at N = 32768 for the temperature example's tree.  A NaN among the copied entries raises *nan_flag like the
// Gram kernels do (the flags are zeroed between the Gram launch and this copy).
__global__ __launch_bounds__(256) void copy_lower_kernel(const double *__restrict__ src, long long ld, long long n, double *__restrict__ dst,
                                                        int *__restrict__ nan_flag) {
  const long long j = blockIdx.y;
  const long long i = j / 2 * 2 + ((long long)blockIdx.x * 256 + threadIdx.x) * 2;  // pairs from the even row at or above the diagonal
  bool bad = false;
  if (i + 1 < n) {
    const double2 v = *reinterpret_cast<const double2 *>(src + i + j * ld);
    *reinterpret_cast<double2 *>(dst + i + j * ld) = v;
    bad = (i >= j && v.x != v.x) || v.y != v.y;
  } else if (i < n) {
    const double v = src[i + j * ld];
    dst[i + j * ld] = v;
    bad = v != v;
  }
  if (bad && nan_flag) atomicOr(nan_flag, 1);
}
void launch_copy_lower(hipStream_t s, const double *src, long long ld, long long n, double *dst, int *nan_flag) {
  if (n <= 0) return;
  for (long long j0 = 0; j0 < n; j0 += 2048) {
    const long long cols = (n - j0 < 2048) ? n - j0 : 2048, rows = n - j0 / 2 * 2;
    hipLaunchKernelGGL(copy_lower_kernel, dim3((unsigned)((rows + 511) / 512), (unsigned)cols), dim3(256), 0, s, src + j0 * ld + j0 / 2 * 2, ld,
                       n - j0 / 2 * 2, dst + j0 * ld + j0 / 2 * 2, nan_flag);
  }
}

// dst_b = src_b^T for `count` contiguous m x m blocks (leading dimension m): the transposed copies of the wide inverted
// diagonal blocks, so that BOTH sweeps of the mixed fit's preconditioner apply them with the one-workgroup-per-output
// colvec_dot (a 1024 x 1024 row-wise product on 16 workgroups took 21 us, the column-wise one on 1024 takes 8)
__global__ __launch_bounds__(256) void transpose_blocks_kernel(const double *__restrict__ src, double *__restrict__ dst, long long m) {
  __shared__ double tile[32][33];
  const double *S = src + (long long)blockIdx.z * m * m;
  double *D = dst + (long long)blockIdx.z * m * m;
  const long long bi = blockIdx.x, bj = blockIdx.y;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    const long long row = bi * 32 + tx, col = bj * 32 + r;
    tile[r][tx] = (row < m && col < m) ? S[col * m + row] : 0.;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const long long row = bj * 32 + tx, col = bi * 32 + r;  // D[row, col] = S[col, row]
    if (row < m && col < m) D[col * m + row] = tile[tx][r];
  }
}
void launch_transpose_blocks(hipStream_t s, const double *src, double *dst, long long m, long long count) {
  if (m <= 0 || count <= 0) return;
  const unsigned nb = (unsigned)((m + 31) / 32);
  hipLaunchKernelGGL(transpose_blocks_kernel, dim3(nb, nb, (unsigned)count), dim3(256), 0, s, src, dst, m);
}

// the same for `count` problems: W_b = W + b * stride_W; v, base and out are slices of vectors stride_v apart
void launch_colvec_dot_strided(hipStream_t s, const double *W, long long ld, long long stride_W, long long m, long long n,
                               const double *v, long long stride_v, double alpha, double beta, const double *base, double *out,
                               long long count) {
  if (n <= 0 || count <= 0) return;
  hipLaunchKernelGGL(colvec_dot_kernel<double>, dim3((unsigned)n, (unsigned)count), dim3(256), 0, s, W, ld, m, v, alpha, beta, base, out,
                     stride_W, stride_v);
}

// ---------------------------------------------------------------------------------------------------------------
// out = alpha * K p + beta * base for a SYMMETRIC K of which only the lower triangle is stored (column-major, ld): every
// stored entry is read once and used twice - half the traffic of a product with the full matrix, which is what bounds
// the K p of the conjugate-gradient steps of the mixed-precision fit (api.hip: refine_information).  Deterministic (no
// atomics):
//   symv_lower_kernel   block (strip c of 32 columns, segment s of 4096 rows): for its rows r and columns j <= r
//                       rowpart[c][r] = sum_j K[r][j] p[j]            (the product with the stored triangle)
//                       colpart[s][j] = sum_{r > j} K[r][j] p[r]      (the product with its mirror image)
//   symv_reduce_kernel  out[r] = alpha (sum_{c <= r / 32} rowpart[c][r] + sum_s colpart[s][r]) + beta base[r], fixed order
// ---------------------------------------------------------------------------------------------------------------
constexpr int SYMV_W = 32;
constexpr int SYMV_SEG = 4096;

__global__ __launch_bounds__(256) void symv_lower_kernel(const double *__restrict__ K, long long ld, long long n,
                                                         const double *__restrict__ p, double *__restrict__ rowpart,
                                                         long long ldp, double *__restrict__ colpart) {
  const long long j0 = (long long)blockIdx.x * SYMV_W, seg0 = (long long)blockIdx.y * SYMV_SEG;
  const long long r_begin = seg0 > j0 ? seg0 : j0, r_end = seg0 + SYMV_SEG < n ? seg0 + SYMV_SEG : n;
  if (r_begin >= r_end) return;  // the segment lies above the strip's diagonal block (symv_reduce_kernel knows which do)
  __shared__ double pj[SYMV_W];
  __shared__ double red[4][SYMV_W];
  const int tid = threadIdx.x;
  if (tid < SYMV_W) pj[tid] = (j0 + tid < n) ? p[j0 + tid] : 0.;
  __syncthreads();
  double colacc[SYMV_W];
#pragma unroll
  for (int jj = 0; jj < SYMV_W; ++jj) colacc[jj] = 0.;
  const double *Kc = K + j0 * ld;
  for (long long r = r_begin + tid; r < r_end; r += 256) {
    const double pr = p[r];
    double racc = 0.;
    if (r >= j0 + SYMV_W) {  // below the diagonal block: all 32 columns (j < j0 + 32 <= r < n)
      double v[SYMV_W];
#pragma unroll
      for (int jj = 0; jj < SYMV_W; ++jj) v[jj] = Kc[r + jj * ld];
#pragma unroll
      for (int jj = 0; jj < SYMV_W; ++jj) {
        racc += v[jj] * pj[jj];
        colacc[jj] += v[jj] * pr;
      }
    } else {  // a row of the diagonal block: columns j <= r, the diagonal entry counted once
#pragma unroll
      for (int jj = 0; jj < SYMV_W; ++jj) {
        const long long j = j0 + jj;
        if (j <= r) {
          const double v = Kc[r + jj * ld];
          racc += v * pj[jj];
          if (j < r) colacc[jj] += v * pr;
        }
      }
    }
    rowpart[(long long)blockIdx.x * ldp + r] = racc;
  }
#pragma unroll
  for (int jj = 0; jj < SYMV_W; ++jj) {
    double a = colacc[jj];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) a += __shfl_down(a, off, 64);
    if ((tid & 63) == 0) red[tid >> 6][jj] = a;
  }
  __syncthreads();
  if (tid < SYMV_W && j0 + tid < n) colpart[(long long)blockIdx.y * ldp + j0 + tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}

// (256 rows per block, four threads per row: thread group g sums the strips c = g, g + 4, ... - one thread per row walked
// up to n / 32 strided loads alone and the launch took 0.37 ms at N = 32768, a third of the product itself)
__global__ __launch_bounds__(1024) void symv_reduce_kernel(const double *__restrict__ rowpart, const double *__restrict__ colpart,
                                                           long long ldp, long long n, double alpha, double beta,
                                                           const double *base, double *out) {
  __shared__ double part[3][256];
  const int lane = threadIdx.x & 255, g = threadIdx.x >> 8;
  const long long r = (long long)blockIdx.x * 256 + lane;
  double acc = 0.;
  if (r < n) {
    const long long cmax = r / SYMV_W;
    for (long long c = g; c <= cmax; c += 4) acc += rowpart[c * ldp + r];
    if (g == 0) {
      const long long nseg = (n + SYMV_SEG - 1) / SYMV_SEG;
      for (long long sgm = (cmax * SYMV_W) / SYMV_SEG; sgm < nseg; ++sgm) acc += colpart[sgm * ldp + r];
    }
  }
  if (g > 0) part[g - 1][lane] = acc;
  __syncthreads();
  if (g == 0 && r < n) {
    acc = (acc + part[0][lane]) + (part[1][lane] + part[2][lane]);  // fixed order
    out[r] = alpha * acc + (base ? beta * base[r] : 0.);
  }
}

// ws: symv_ws_elems(n) doubles of scratch
size_t symv_ws_elems(long long n) {
  const long long ldp = (n + 7) / 8 * 8;
  return (size_t)ldp * (size_t)((n + SYMV_W - 1) / SYMV_W + (n + SYMV_SEG - 1) / SYMV_SEG);
}

void launch_symv_lower(hipStream_t s, const double *K, long long ld, long long n, const double *p, double alpha, double beta,
                       const double *base, double *out, double *ws) {
  if (n <= 0) return;
  const long long ldp = (n + 7) / 8 * 8, nstrip = (n + SYMV_W - 1) / SYMV_W, nseg = (n + SYMV_SEG - 1) / SYMV_SEG;
  double *rowpart = ws, *colpart = ws + ldp * nstrip;
  hipLaunchKernelGGL(symv_lower_kernel, dim3((unsigned)nstrip, (unsigned)nseg), dim3(256), 0, s, K, ld, n, p, rowpart, ldp, colpart);
  hipLaunchKernelGGL(symv_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(1024), 0, s, rowpart, colpart, ldp, n, alpha, beta, base, out);
}

// batched: out[b * m + j] = sum_i Q_b[i, j] z_b[i], Q_b = Q + b * stride_Q (m x m, ld), z_b = z + b * stride_z
__global__ __launch_bounds__(256) void colvec_dot_batched_kernel(const double *__restrict__ Q, long long ld,
                                                                 long long stride_Q, long long m,
                                                                 const double *__restrict__ z, long long stride_z,
                                                                 double *__restrict__ out) {
  __shared__ double red[4];
  const long long j = blockIdx.x, b = blockIdx.y;
  const double *q = Q + b * stride_Q + j * ld;
  const double *zb = z + b * stride_z;
  double acc = 0.;
  for (long long i = threadIdx.x; i < m; i += 256) acc += q[i] * zb[i];
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, 64);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) out[b * m + j] = (red[0] + red[1]) + (red[2] + red[3]);
}

void launch_colvec_dot_batched(hipStream_t s, const double *Q, long long ld, long long stride_Q, long long m,
                               const double *z, long long stride_z, long long count, double *out) {
  if (m <= 0 || count <= 0) return;
  hipLaunchKernelGGL(colvec_dot_batched_kernel, dim3((unsigned)m, (unsigned)count), dim3(256), 0, s, Q, ld, stride_Q, m, z,
                     stride_z, out);
}

// identity in every m x m slab (ld, stride) of `count`
__global__ __launch_bounds__(256) void set_identity_batched_kernel(double *B, long long ld, long long stride, long long m) {
  double *b = B + (long long)blockIdx.y * stride;
  const long long col = blockIdx.x;
  for (long long r = threadIdx.x; r < m; r += 256) b[col * ld + r] = (r == col) ? 1. : 0.;
}

void launch_set_identity_batched(hipStream_t s, double *B, long long ld, long long stride, long long m, long long count) {
  if (m <= 0 || count <= 0) return;
  hipLaunchKernelGGL(set_identity_batched_kernel, dim3((unsigned)m, (unsigned)count), dim3(256), 0, s, B, ld, stride, m);
}

// ---- ragged groups <-> slabs of one width (sparse GP blocks) --------------------------------
// group g owns the source columns off[g] .. off[g + 1); in the padded layout it owns the columns
// g * smax .. (g + 1) * smax, the ones beyond its size zero-filled.
// dir = 0: dst(padded) <- src(compact); dir = 1: dst(compact) <- src(padded), padding dropped.
__global__ __launch_bounds__(256) void pad_columns_kernel(const double *__restrict__ src, long long ld_src,
                                                          const long long *__restrict__ off, long long smax,
                                                          long long rows, double *__restrict__ dst, long long ld_dst,
                                                          int dir) {
  const long long pc = blockIdx.x;  // padded column
  const long long g = pc / smax, a = pc % smax;
  const long long sg = off[g + 1] - off[g];
  const bool real = a < sg;
  const long long cc = off[g] + a;  // compact column
  if (dir == 0) {
    double *d = dst + pc * ld_dst;
    const double *sp = src + cc * ld_src;
    for (long long i = threadIdx.x; i < rows; i += 256) d[i] = real ? sp[i] : 0.;
  } else if (real) {
    double *d = dst + cc * ld_dst;
    const double *sp = src + pc * ld_src;
    for (long long i = threadIdx.x; i < rows; i += 256) d[i] = sp[i];
  }
}

void launch_pad_columns(hipStream_t s, const double *src, long long ld_src, const long long *off, long long smax,
                        long long n_groups, long long rows, double *dst, long long ld_dst, int dir) {
  if (n_groups <= 0 || smax <= 0 || rows <= 0) return;
  hipLaunchKernelGGL(pad_columns_kernel, dim3((unsigned)(smax * n_groups)), dim3(256), 0, s, src, ld_src, off, smax, rows, dst,
                     ld_dst, dir);
}

// every smax x smax slab (ld, stride): rows / columns beyond the group's size become identity (lower part)
__global__ __launch_bounds__(256) void pad_identity_kernel(double *A, long long ld, long long stride,
                                                           const long long *__restrict__ off, long long smax) {
  const long long g = blockIdx.y, col = blockIdx.x;
  const long long sg = off[g + 1] - off[g];
  double *a = A + g * stride + col * ld;
  for (long long r = col + threadIdx.x; r < smax; r += 256)
    if (r >= sg || col >= sg) a[r] = (r == col) ? 1. : 0.;
}

void launch_pad_identity(hipStream_t s, double *A, long long ld, long long stride, const long long *off, long long smax,
                         long long n_groups) {
  if (n_groups <= 0 || smax <= 0) return;
  hipLaunchKernelGGL(pad_identity_kernel, dim3((unsigned)smax, (unsigned)n_groups), dim3(256), 0, s, A, ld, stride, off, smax);
}

// packed ragged blocks <- slabs: block g (sg x sg, ld sg) at out + boff[g] from the top-left of slab g
__global__ __launch_bounds__(256) void compact_blocks_kernel(const double *__restrict__ slabs, long long ld, long long stride,
                                                             const long long *__restrict__ off,
                                                             const long long *__restrict__ boff, double *__restrict__ out) {
  const long long g = blockIdx.y, c = blockIdx.x;
  const long long sg = off[g + 1] - off[g];
  if (c >= sg) return;
  const double *sp = slabs + g * stride + c * ld;
  double *d = out + boff[g] + c * sg;
  for (long long r = threadIdx.x; r < sg; r += 256) d[r] = sp[r];
}

void launch_compact_blocks(hipStream_t s, const double *slabs, long long ld, long long stride, const long long *off,
                           const long long *boff, long long smax, long long n_groups, double *out) {
  if (n_groups <= 0 || smax <= 0) return;
  hipLaunchKernelGGL(compact_blocks_kernel, dim3((unsigned)smax, (unsigned)n_groups), dim3(256), 0, s, slabs, ld, stride, off,
                     boff, out);
}

// out[i] = a * x[i] + b * (y ? y[i] : 1)
__global__ __launch_bounds__(256) void axpby_kernel(long long n, double a, const double *x, double b, const double *y,
                                                    double *out) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  out[i] = (x ? a * x[i] : 0.) + b * (y ? y[i] : 1.);
}

void launch_axpby(hipStream_t s, long long n, double a, const double *x, double b, const double *y, double *out) {
  if (n <= 0) return;
  hipLaunchKernelGGL(axpby_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, n, a, x, b, y, out);
}

}  // namespace agp
