/*
 * oracle.c — CPU restatement of albatross's dense-GP hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under albatross_amd/ may include, link or
 * call this file; only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg use it, as the checker / the timed CPU baseline.
 *
 * Parity status: PINNED against the reference's own golden vectors
 * (tests/golden/*.json, taken from /root/reference/tests: Matern 15x15 oracle
 * matrices, MVN NLL known answer, radial edge cases, distance-metric values,
 * toy-linear-data GP).  The pivoted LDL^T below restates the PUBLISHED
 * algorithm of Eigen 3.3 `LDLT<MatrixXd, Lower>` (third-party dependency
 * `eigen` 3.3.swiftnav.1, MODULE.bazel:38-42 — its source is NOT in the
 * reference checkout); the factor itself (L, D, P) is parity-unpinned, results
 * are pinned at the solve / log-det / prediction level where every reference
 * test checks them.  The reference itself cannot be compiled here (every
 * header needs Eigen), so there is no oracle/_ref.
 *
 * All file:line citations are relative to the albatross checkout,
 * include/albatross/src/...
 */
#include <math.h>
#include <pthread.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/albatross_amd.h"

#define ORC_API __attribute__((visibility("default")))

/* ---------------------------------------------------------------------- */
/* distance metrics: covariance_functions/distance_metrics.hpp:30-90       */
/* ---------------------------------------------------------------------- */
static double norm_of(const double *x, int dim) {
  double s = 0.;
  for (int d = 0; d < dim; ++d) s += x[d] * x[d];
  return sqrt(s);
}

static double dist_euclidean(const double *x, const double *y, int dim) {
  if (dim == 1) return fabs(x[0] - y[0]); /* :34-36 */
  double s = 0.;                          /* (x - y).norm()  :38-42 */
  for (int d = 0; d < dim; ++d) {
    const double t = x[d] - y[d];
    s += t * t;
  }
  return sqrt(s);
}

static double dist_radial(const double *x, const double *y, int dim) {
  return fabs(norm_of(x, dim) - norm_of(y, dim)); /* :47-51 */
}

static double dist_angular(const double *x, const double *y, int dim) {
  const double eps = 1e-16; /* EPSILON, :18 */
  double dot = 0.;
  for (int d = 0; d < dim; ++d) dot += x[d] * y[d];
  const double c = dot / (norm_of(x, dim) * norm_of(y, dim)); /* :70 */
  if (c > 1. - eps) return 0.;                               /* :71-72 */
  if (c < -1. + eps) return M_PI;                            /* :73-74 */
  return acos(c);
}

static double distance(int metric, const double *x, const double *y, int dim) {
  switch (metric) {
  case AGP_METRIC_RADIAL: return dist_radial(x, y, dim);
  case AGP_METRIC_ANGULAR: return dist_angular(x, y, dim);
  default: return dist_euclidean(x, y, dim);
  }
}

/* ---------------------------------------------------------------------- */
/* radial kernels: covariance_functions/radial.hpp                         */
/* ---------------------------------------------------------------------- */
static double squared_exponential(double d, double l, double sigma) { /* :25-33 */
  if (l <= 0.) return 0.;
  return sigma * sigma * exp(-pow(d / l, 2));
}
static double exponential(double d, double l, double sigma) { /* :191-198 */
  if (l <= 0.) return 0.;
  return sigma * sigma * exp(-fabs(d / l));
}
static double matern32(double d, double l, double sigma) { /* :289-297 */
  if (l <= 0.) return 0.;
  const double q = sqrt(3.) * d / l;
  return sigma * sigma * (1 + q) * exp(-q);
}
static double matern52(double d, double l, double sigma) { /* :461-470 */
  if (l <= 0.) return 0.;
  const double q = sqrt(5.) * d / l;
  return sigma * sigma * (1 + q + q * q / 3.) * exp(-q);
}

/* ---------------------------------------------------------------------- */
/* one k(x, y): postfix walk of the composed covariance function           */
/* ---------------------------------------------------------------------- */
typedef struct {
  const double *coords;
  const int64_t *eq_id;
  const double *scales;
  int64_t n;
  int dim;
  int is_measurement;
} orc_set;

static int features_equal(const orc_set *X, int64_t i, const orc_set *Y,
                          int64_t j) {
  if (X->eq_id && Y->eq_id) return X->eq_id[i] == Y->eq_id[j];
  for (int d = 0; d < X->dim; ++d)
    if (!(X->coords[i * X->dim + d] == Y->coords[j * Y->dim + d])) return 0;
  return 1;
}

static double eval_pair(const agp_kernel_node *prog, int n_nodes,
                        const orc_set *X, int64_t i, const orc_set *Y,
                        int64_t j) {
  double st[AGP_MAX_STACK];
  int def[AGP_MAX_STACK]; /* slot holds the value of a term that HAS a caller for this pair of feature types */
  int sp = 0;
  const double *x = X->coords + i * X->dim;
  const double *y = Y->coords + j * Y->dim;
  const int dim = X->dim;
  for (int t = 0; t < n_nodes; ++t) {
    const agp_kernel_node *nd = &prog[t];
    const double *p = nd->params;
    if (nd->op <= AGP_OP_SCALING) def[sp] = 1; /* every leaf is defined for every pair */
    switch (nd->op) {
    case AGP_OP_SQUARED_EXPONENTIAL:
      st[sp++] = squared_exponential(distance(nd->metric, x, y, dim), p[0], p[1]);
      break;
    case AGP_OP_EXPONENTIAL:
      st[sp++] = exponential(distance(nd->metric, x, y, dim), p[0], p[1]);
      break;
    case AGP_OP_MATERN32:
      st[sp++] = matern32(distance(nd->metric, x, y, dim), p[0], p[1]);
      break;
    case AGP_OP_MATERN52:
      st[sp++] = matern52(distance(nd->metric, x, y, dim), p[0], p[1]);
      break;
    case AGP_OP_CONSTANT: /* polynomials.hpp:56-60 */
      st[sp++] = p[0] * p[0];
      break;
    case AGP_OP_INDEPENDENT_NOISE: /* noise.hpp:37-43 */
    case AGP_OP_NUGGET:            /* nugget.hpp:40-48 */
      st[sp++] = features_equal(X, i, Y, j) ? p[0] * p[0] : 0.;
      break;
    case AGP_OP_POLYNOMIAL: { /* polynomials.hpp:78-86 */
      double cov = 0.;
      for (int q = 0; q < nd->order + 1; ++q) {
        const double s = p[q];
        cov += s * s * pow(x[0], (double)q) * pow(y[0], (double)q);
      }
      st[sp++] = cov;
    } break;
    case AGP_OP_SCALING: /* scaling_function.hpp:79-83: f(x) * f(y) */
      st[sp++] = X->scales[nd->column * X->n + i] * Y->scales[nd->column * Y->n + j];
      break;
    case AGP_OP_SUM: /* covariance_function.hpp:266-294: both sides, or the one side that has a caller */
      st[sp - 2] = st[sp - 2] + st[sp - 1]; /* a side without a caller holds 0 */
      def[sp - 2] = def[sp - 2] || def[sp - 1];
      --sp;
      break;
    case AGP_OP_PRODUCT: { /* covariance_function.hpp:357-389 */
      double out;
      if (def[sp - 2] && def[sp - 1]) { /* :357-367 */
        out = st[sp - 2];
        if (out != 0.) out *= st[sp - 1];
      } else if (def[sp - 2]) { /* :372-378: only LHS has a caller, R is ignored */
        out = st[sp - 2];
      } else if (def[sp - 1]) { /* :383-389 */
        out = st[sp - 1];
      } else {
        out = 0.;
      }
      st[sp - 2] = out;
      def[sp - 2] = def[sp - 2] || def[sp - 1];
      --sp;
    } break;
    case AGP_OP_MEASUREMENT_ONLY: /* measurement.hpp:87-102 */
      if (!(X->is_measurement && Y->is_measurement)) st[sp - 1] = 0.;
      break;
    case AGP_OP_TYPE_PAIR: { /* VariantForwarder, callers.hpp:419-544: no _call_impl for the pair of alternatives -> 0 */
      const double tx = X->scales[nd->column * X->n + i], ty = Y->scales[nd->column * Y->n + j];
      if (!((tx == p[0] && ty == p[1]) || (tx == p[1] && ty == p[0]))) {
        st[sp - 1] = 0.;
        def[sp - 1] = 0; /* no _call_impl for this pair of alternatives */
      }
    } break;
    default:
      st[sp++] = NAN;
    }
  }
  return st[0];
}

static orc_set as_set(const agp_features *f) {
  orc_set s;
  s.coords = f->coords;
  s.eq_id = f->eq_id;
  s.scales = f->scales;
  s.n = f->n;
  s.dim = f->dim;
  s.is_measurement = f->is_measurement;
  return s;
}

ORC_API double orc_eval(const agp_kernel_node *prog, int n_nodes,
                        const agp_features *x, int64_t i,
                        const agp_features *y, int64_t j) {
  orc_set X = as_set(x), Y = as_set(y);
  return eval_pair(prog, n_nodes, &X, i, &Y, j);
}

/* ---------------------------------------------------------------------- */
/* Gram: covariance_functions/callers.hpp                                  */
/* ---------------------------------------------------------------------- */
/* cross, serial: callers.hpp:38-60 (row-major walk of a col-major matrix) */
ORC_API void orc_gram_cross(const agp_kernel_node *prog, int n_nodes,
                            const agp_features *x, const agp_features *y,
                            double *C, int64_t ld) {
  orc_set X = as_set(x), Y = as_set(y);
  for (int64_t i = 0; i < X.n; ++i)
    for (int64_t j = 0; j < Y.n; ++j)
      C[i + j * ld] = eval_pair(prog, n_nodes, &X, i, &Y, j);
}

/* symmetric, serial: callers.hpp:107-129 (lower triangle, mirrored) */
ORC_API void orc_gram_sym(const agp_kernel_node *prog, int n_nodes,
                          const agp_features *x, double *C, int64_t ld) {
  orc_set X = as_set(x);
  for (int64_t i = 0; i < X.n; ++i)
    for (int64_t j = 0; j <= i; ++j) {
      C[i + j * ld] = eval_pair(prog, n_nodes, &X, i, &X, j);
      C[j + i * ld] = C[i + j * ld];
    }
}

/* partition_triangular: indexing/block.hpp:25-44 — column blocks of
 * approximately equal triangle area: end_fraction_{b+1} = sqrt(1/k + area_b),
 * end index = rint(size * end_fraction), last block clamped to size. */
static int partition_triangular(int64_t size, int64_t block_count,
                                int64_t *starts, int64_t *ends) {
  double area = 0.;
  int64_t start = 0;
  for (int64_t b = 0; b < block_count; ++b) {
    const double end_fraction = sqrt(1. / (double)block_count + area);
    area = end_fraction * end_fraction;
    const int64_t end = (int64_t)rint((double)size * end_fraction);
    starts[b] = start;
    ends[b] = end;
    start = end;
  }
  if (ends[block_count - 1] > size) ends[block_count - 1] = size;
  return (int)block_count;
}

typedef struct {
  const agp_kernel_node *prog;
  int n_nodes;
  const orc_set *X;
  const orc_set *Y;
  double *C;
  int64_t ld;
  int64_t c0, c1;
  int symmetric;
} gram_job;

static void *gram_worker(void *arg) {
  gram_job *jb = (gram_job *)arg;
  for (int64_t col = jb->c0; col < jb->c1; ++col) {
    /* symmetric: rows 0..col of column col (upper triangle, callers.hpp:149-157)
     * cross:     every row of column col (callers.hpp:86-96) */
    const int64_t rows = jb->symmetric ? col + 1 : jb->X->n;
    for (int64_t row = 0; row < rows; ++row)
      jb->C[row + col * jb->ld] =
          eval_pair(jb->prog, jb->n_nodes, jb->X, row, jb->Y, col);
  }
  return NULL;
}

/* symmetric, pooled: callers.hpp:134-166 */
ORC_API void orc_gram_sym_pooled(const agp_kernel_node *prog, int n_nodes,
                                 const agp_features *x, double *C, int64_t ld,
                                 int threads) {
  orc_set X = as_set(x);
  if (threads <= 1) { /* should_serial_apply */
    orc_gram_sym(prog, n_nodes, x, C, ld);
    return;
  }
  int64_t *starts = malloc(sizeof(int64_t) * (size_t)threads);
  int64_t *ends = malloc(sizeof(int64_t) * (size_t)threads);
  const int nb = partition_triangular(X.n, threads, starts, ends);
  pthread_t *tid = malloc(sizeof(pthread_t) * (size_t)nb);
  gram_job *jobs = malloc(sizeof(gram_job) * (size_t)nb);
  for (int b = 0; b < nb; ++b) {
    gram_job jb = {prog, n_nodes, &X, &X, C, ld, starts[b], ends[b], 1};
    jobs[b] = jb;
    pthread_create(&tid[b], NULL, gram_worker, &jobs[b]);
  }
  for (int b = 0; b < nb; ++b) pthread_join(tid[b], NULL);
  /* output.triangularView<Lower>() = output.transpose()  (:163-164) */
  for (int64_t j = 0; j < X.n; ++j)
    for (int64_t i = j + 1; i < X.n; ++i) C[i + j * ld] = C[j + i * ld];
  free(starts); free(ends); free(tid); free(jobs);
}

/* cross, pooled: callers.hpp:66-102 (ceil(n_cols / threads) column blocks) */
ORC_API void orc_gram_cross_pooled(const agp_kernel_node *prog, int n_nodes,
                                   const agp_features *x, const agp_features *y,
                                   double *C, int64_t ld, int threads) {
  orc_set X = as_set(x), Y = as_set(y);
  if (threads <= 1) {
    orc_gram_cross(prog, n_nodes, x, y, C, ld);
    return;
  }
  const int64_t block = (int64_t)ceil((double)Y.n / (double)threads);
  pthread_t *tid = malloc(sizeof(pthread_t) * (size_t)threads);
  gram_job *jobs = malloc(sizeof(gram_job) * (size_t)threads);
  int nb = 0;
  for (int b = 0; b < threads; ++b) {
    const int64_t c0 = b * block;
    const int64_t c1 = (b + 1) * block < Y.n ? (b + 1) * block : Y.n;
    if (c0 >= c1) continue;
    gram_job jb = {prog, n_nodes, &X, &Y, C, ld, c0, c1, 0};
    jobs[nb] = jb;
    pthread_create(&tid[nb], NULL, gram_worker, &jobs[nb]);
    ++nb;
  }
  for (int b = 0; b < nb; ++b) pthread_join(tid[b], NULL);
  free(tid); free(jobs);
}

/* ---------------------------------------------------------------------- */
/* Eigen 3.3 LDLT<MatrixXd, Lower>, unblocked, in place (published         */
/* algorithm; call sites eigen/serializable_ldlt.hpp:27, likelihood.hpp:63)*/
/*   P A P^T = L D L^T ; L unit lower in the strict lower triangle of A,   */
/*   D on the diagonal, transpositions in tr[].                            */
/* ---------------------------------------------------------------------- */
ORC_API int orc_ldlt(double *A, int64_t n, int64_t ld, int64_t *tr) {
  int ok = 1, found_zero_pivot = 0;
  if (n <= 1) {
    if (n == 1) tr[0] = 0;
    return 1;
  }
  double *temp = malloc(sizeof(double) * (size_t)n);
  for (int64_t k = 0; k < n; ++k) {
    /* largest |diagonal| of the trailing block */
    int64_t big = k;
    double best = fabs(A[k + k * ld]);
    for (int64_t i = k + 1; i < n; ++i) {
      const double v = fabs(A[i + i * ld]);
      if (v > best) { best = v; big = i; }
    }
    tr[k] = big;
    if (big != k) {
      /* symmetric row/column swap touching only the lower triangle */
      for (int64_t c = 0; c < k; ++c) {
        const double t = A[k + c * ld];
        A[k + c * ld] = A[big + c * ld];
        A[big + c * ld] = t;
      }
      for (int64_t r = big + 1; r < n; ++r) {
        const double t = A[r + k * ld];
        A[r + k * ld] = A[r + big * ld];
        A[r + big * ld] = t;
      }
      {
        const double t = A[k + k * ld];
        A[k + k * ld] = A[big + big * ld];
        A[big + big * ld] = t;
      }
      for (int64_t i = k + 1; i < big; ++i) {
        const double t = A[i + k * ld];
        A[i + k * ld] = A[big + i * ld];
        A[big + i * ld] = t;
      }
    }
    const int64_t rs = n - k - 1;
    if (k > 0) {
      /* temp = D[:k] .* A10^T ; A_kk -= A10 temp ; A21 -= A20 temp */
      double dot = 0.;
      for (int64_t c = 0; c < k; ++c) {
        temp[c] = A[c + c * ld] * A[k + c * ld];
        dot += A[k + c * ld] * temp[c];
      }
      A[k + k * ld] -= dot;
      for (int64_t c = 0; c < k; ++c) {
        const double t = temp[c];
        const double *col = A + c * ld;
        double *dst = A + k * ld;
        for (int64_t r = k + 1; r < n; ++r) dst[r] -= col[r] * t;
      }
    }
    const double akk = A[k + k * ld];
    const int pivot_is_valid = fabs(akk) > 0.;
    if (k == 0 && !pivot_is_valid) {
      for (int64_t j = 0; j < n; ++j) {
        tr[j] = j;
        for (int64_t r = j + 1; r < n; ++r) ok = ok && (A[r + j * ld] == 0.);
      }
      free(temp);
      return ok;
    }
    if (rs > 0 && pivot_is_valid) {
      for (int64_t r = k + 1; r < n; ++r) A[r + k * ld] /= akk;
    } else if (rs > 0) {
      for (int64_t r = k + 1; r < n; ++r) ok = ok && (A[r + k * ld] == 0.);
    }
    if (found_zero_pivot && pivot_is_valid) ok = 0;
    else if (!pivot_is_valid) found_zero_pivot = 1;
  }
  free(temp);
  return ok;
}

/* LDLT::solve: P^T L^-T D^+ L^-1 P rhs, D^+ zeroes rows whose |D| is not
 * above numeric_limits<double>::min().  In place on B (n x nrhs, ld ldb). */
ORC_API void orc_ldlt_solve(const double *A, int64_t n, int64_t ld,
                            const int64_t *tr, double *B, int64_t nrhs,
                            int64_t ldb) {
  const double tol = 2.2250738585072014e-308;
  for (int64_t c = 0; c < nrhs; ++c) {
    double *b = B + c * ldb;
    for (int64_t k = 0; k < n; ++k)
      if (tr[k] != k) { const double t = b[k]; b[k] = b[tr[k]]; b[tr[k]] = t; }
    for (int64_t j = 0; j < n; ++j) { /* unit-lower forward substitution */
      const double bj = b[j];
      const double *col = A + j * ld;
      for (int64_t i = j + 1; i < n; ++i) b[i] -= col[i] * bj;
    }
    for (int64_t i = 0; i < n; ++i) {
      const double d = A[i + i * ld];
      if (fabs(d) > tol) b[i] /= d; else b[i] = 0.;
    }
    for (int64_t j = n - 1; j >= 0; --j) { /* unit-upper (L^T) back substitution */
      const double *col = A + j * ld;
      double s = b[j];
      for (int64_t i = j + 1; i < n; ++i) s -= col[i] * b[i];
      b[j] = s;
    }
    for (int64_t k = n - 1; k >= 0; --k)
      if (tr[k] != k) { const double t = b[k]; b[k] = b[tr[k]]; b[tr[k]] = t; }
  }
}

/* log_sum(vectorD): likelihood.hpp:26-32 / serializable_ldlt.hpp:128-135 */
ORC_API double orc_ldlt_logdet(const double *A, int64_t n, int64_t ld) {
  double s = 0.;
  for (int64_t i = 0; i < n; ++i) s += log(A[i + i * ld]);
  return s;
}

/* ---------------------------------------------------------------------- */
/* un-pivoted LL^T (what the device computes) — left-looking, column form  */
/* returns 0 on success, k+1 if pivot k is not positive                    */
/* ---------------------------------------------------------------------- */
ORC_API int64_t orc_llt(double *A, int64_t n, int64_t ld) {
  for (int64_t j = 0; j < n; ++j) {
    double *cj = A + j * ld;
    for (int64_t c = 0; c < j; ++c) {
      const double *cc = A + c * ld;
      const double t = cc[j];
      for (int64_t r = j; r < n; ++r) cj[r] -= cc[r] * t;
    }
    const double d = cj[j];
    if (!(d > 0.)) return j + 1;
    const double s = sqrt(d);
    cj[j] = s;
    for (int64_t r = j + 1; r < n; ++r) cj[r] /= s;
  }
  return 0;
}

ORC_API void orc_llt_solve(const double *L, int64_t n, int64_t ld, double *B,
                           int64_t nrhs, int64_t ldb) {
  for (int64_t c = 0; c < nrhs; ++c) {
    double *b = B + c * ldb;
    for (int64_t j = 0; j < n; ++j) {
      const double *col = L + j * ld;
      b[j] /= col[j];
      const double bj = b[j];
      for (int64_t i = j + 1; i < n; ++i) b[i] -= col[i] * bj;
    }
    for (int64_t j = n - 1; j >= 0; --j) {
      const double *col = L + j * ld;
      double s = b[j];
      for (int64_t i = j + 1; i < n; ++i) s -= col[i] * b[i];
      b[j] = s / col[j];
    }
  }
}

ORC_API double orc_llt_logdet(const double *L, int64_t n, int64_t ld) {
  double s = 0.;
  for (int64_t i = 0; i < n; ++i) s += log(L[i + i * ld]);
  return 2. * s;
}

/* ---------------------------------------------------------------------- */
/* fit / nll / predict: models/gp.hpp, evaluation/likelihood.hpp           */
/* ---------------------------------------------------------------------- */
typedef struct {
  int64_t n;
  double *ldlt;   /* n x n packed factor */
  int64_t *tr;
  double *information;
  agp_features train; /* un-wrapped training features (gp.hpp:293) */
  double *coords_copy;
  int64_t *eq_copy;
  double *scales_copy;
  int use_llt;
  /* Fit<GPFit<BlockSymmetric<Solver>, F>> made by update (gp.hpp:384-414): the old fit as solver A, Ai_B = A^-1 B
   * (n_A x m) and the pivoted LDL^T of S (m x m): linalg/block_symmetric.hpp:46-60.  base == NULL: plain factor. */
  const void *base;  /* const orc_fit *: the fit that was updated (must outlive this one) */
  double *Ai_B;
  double *S_ldlt;
  int64_t *S_tr;
  int64_t m;         /* rows of S */
} orc_fit;

static int has_nan(const double *A, int64_t n, int64_t ld) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i)
      if (isnan(A[i + j * ld])) return 1;
  return 0;
}

/* gp.hpp:281-294 (_fit_impl) + gp.hpp:61-69 (Fit ctor).
 * threads <= 1: serial Gram (the reference default, core/model.hpp:20).
 * use_llt: factor with un-pivoted LL^T instead of Eigen's pivoted LDL^T.
 * Returns NULL with *status = AGP_ERR_NAN_INPUT when the covariance has NaN. */
ORC_API orc_fit *orc_fit_create(const agp_kernel_node *prog, int n_nodes,
                                const agp_features *x, const double *y,
                                const double *y_var, int threads, int use_llt,
                                int *status) {
  const int64_t n = x->n;
  agp_features meas = *x;
  meas.is_measurement = 1; /* as_measurements(features), gp.hpp:288 */
  double *K = malloc(sizeof(double) * (size_t)(n * n));
  orc_gram_sym_pooled(prog, n_nodes, &meas, K, n, threads);
  if (y_var)
    for (int64_t i = 0; i < n; ++i) K[i + i * n] += y_var[i]; /* gp.hpp:65 */
  if (has_nan(K, n, n)) { /* gp.hpp:66 */
    free(K);
    if (status) *status = AGP_ERR_NAN_INPUT;
    return NULL;
  }
  orc_fit *f = calloc(1, sizeof(orc_fit));
  f->n = n;
  f->ldlt = K;
  f->tr = malloc(sizeof(int64_t) * (size_t)n);
  f->use_llt = use_llt;
  f->information = malloc(sizeof(double) * (size_t)n);
  memcpy(f->information, y, sizeof(double) * (size_t)n);
  if (use_llt) {
    const int64_t bad = orc_llt(K, n, n);
    if (bad) {
      if (status) *status = AGP_ERR_NOT_POSITIVE_DEFINITE;
      free(K); free(f->tr); free(f->information); free(f);
      return NULL;
    }
    orc_llt_solve(K, n, n, f->information, 1, n);
  } else {
    orc_ldlt(K, n, n, f->tr);                                 /* gp.hpp:67 */
    orc_ldlt_solve(K, n, n, f->tr, f->information, 1, n);     /* gp.hpp:68 */
  }
  /* train_features = features (un-wrapped), gp.hpp:63 */
  f->train = *x;
  f->train.is_measurement = 0;
  f->coords_copy = malloc(sizeof(double) * (size_t)(n * x->dim));
  memcpy(f->coords_copy, x->coords, sizeof(double) * (size_t)(n * x->dim));
  f->train.coords = f->coords_copy;
  if (x->eq_id) {
    f->eq_copy = malloc(sizeof(int64_t) * (size_t)n);
    memcpy(f->eq_copy, x->eq_id, sizeof(int64_t) * (size_t)n);
    f->train.eq_id = f->eq_copy;
  }
  if (x->scales && x->n_scale_columns > 0) {
    const size_t cnt = (size_t)(n * x->n_scale_columns);
    f->scales_copy = malloc(sizeof(double) * cnt);
    memcpy(f->scales_copy, x->scales, sizeof(double) * cnt);
    f->train.scales = f->scales_copy;
  }
  if (status) *status = AGP_OK;
  return f;
}

ORC_API void orc_fit_destroy(orc_fit *f) {
  if (!f) return;
  free(f->ldlt); free(f->tr); free(f->information);
  free(f->coords_copy); free(f->eq_copy); free(f->scales_copy);
  free(f->Ai_B); free(f->S_ldlt); free(f->S_tr);
  free(f);
}

ORC_API void orc_fit_information(const orc_fit *f, double *out) {
  memcpy(out, f->information, sizeof(double) * (size_t)f->n);
}

ORC_API double orc_fit_logdet(const orc_fit *f) {
  return f->use_llt ? orc_llt_logdet(f->ldlt, f->n, f->n)
                    : orc_ldlt_logdet(f->ldlt, f->n, f->n);
}

ORC_API void orc_fit_solve(const orc_fit *f, double *B, int64_t nrhs);

/* BlockSymmetric<Solver>::solve, linalg/block_symmetric.hpp:75-98 (block-matrix inversion with the pre-computed
 * Ai_B = A^-1 B and the factor of S = C - B^T A^-1 B); B is n x nrhs with leading dimension n, overwritten. */
static void block_symmetric_solve(const orc_fit *f, double *B, int64_t nrhs) {
  const orc_fit *A = (const orc_fit *)f->base;
  const int64_t na = A->n, m = f->m, n = f->n;
  double *rhs_a = malloc(sizeof(double) * (size_t)(na * nrhs));
  double *rhs_b = malloc(sizeof(double) * (size_t)(m * nrhs));
  double *t = malloc(sizeof(double) * (size_t)(m * nrhs));
  for (int64_t j = 0; j < nrhs; ++j) {
    memcpy(rhs_a + j * na, B + j * n, sizeof(double) * (size_t)na);      /* rhs.topRows */
    memcpy(rhs_b + j * m, B + j * n + na, sizeof(double) * (size_t)m);   /* rhs.bottomRows */
  }
  for (int64_t j = 0; j < nrhs; ++j)                                     /* Bt_Ai_rhs = Ai_B^T rhs_a   :86 */
    for (int64_t c = 0; c < m; ++c) {
      double acc = 0.;
      for (int64_t i = 0; i < na; ++i) acc += f->Ai_B[i + c * na] * rhs_a[i + j * na];
      t[c + j * m] = acc;
    }
  orc_ldlt_solve(f->S_ldlt, m, m, f->S_tr, t, nrhs, m);                  /* Si_Bt_Ai_rhs               :87 */
  orc_ldlt_solve(f->S_ldlt, m, m, f->S_tr, rhs_b, nrhs, m);              /* Si_rhs_b                   :88 */
  orc_fit_solve(A, rhs_a, nrhs);                                         /* A.solve(rhs_a)             :91 */
  for (int64_t j = 0; j < nrhs; ++j) {
    for (int64_t i = 0; i < na; ++i) {                                   /* + Ai_B (Si_Bt_Ai_rhs - Si_rhs_b) */
      double acc = 0.;
      for (int64_t c = 0; c < m; ++c) acc += f->Ai_B[i + c * na] * (t[c + j * m] - rhs_b[c + j * m]);
      B[i + j * n] = rhs_a[i + j * na] + acc;
    }
    for (int64_t c = 0; c < m; ++c) B[na + c + j * n] = rhs_b[c + j * m] - t[c + j * m];  /* :92 */
  }
  free(rhs_a); free(rhs_b); free(t);
}

ORC_API void orc_fit_solve(const orc_fit *f, double *B, int64_t nrhs) {
  if (f->base) block_symmetric_solve(f, B, nrhs);
  else if (f->use_llt) orc_llt_solve(f->ldlt, f->n, f->n, B, nrhs, f->n);
  else orc_ldlt_solve(f->ldlt, f->n, f->n, f->tr, B, nrhs, f->n);
}

ORC_API void orc_predict_joint(const orc_fit *f, const agp_kernel_node *prog, int n_nodes, const agp_features *xs,
                               double *mean, double *cov);
ORC_API void orc_gram_cross(const agp_kernel_node *prog, int n_nodes, const agp_features *x, const agp_features *y,
                            double *out, int64_t ld);

/* GaussianProcessBase::_update_impl, models/gp.hpp:384-414: condition `old` on further observations.
 * y = targets.mean with the mean function removed by the caller (FitModel::update -> ModelBase::update).
 * `old` must outlive the returned fit (it is the `A` solver of the BlockSymmetric). */
ORC_API orc_fit *orc_fit_update(const orc_fit *old, const agp_kernel_node *prog, int n_nodes, const agp_features *x,
                                const double *y, const double *y_var) {
  const int64_t na = old->n, m = x->n, n = na + m;
  const int dim = x->dim, nsc = x->n_scale_columns;
  orc_fit *f = calloc(1, sizeof(orc_fit));
  f->n = n; f->m = m; f->base = old;
  /* new_features = concatenate(fit_.train_features, features)                                     :387 */
  f->train = old->train;
  f->train.n = n;
  f->coords_copy = malloc(sizeof(double) * (size_t)(n * dim));
  memcpy(f->coords_copy, old->train.coords, sizeof(double) * (size_t)(na * dim));
  memcpy(f->coords_copy + na * dim, x->coords, sizeof(double) * (size_t)(m * dim));
  f->train.coords = f->coords_copy;
  if (old->train.eq_id && x->eq_id) {
    f->eq_copy = malloc(sizeof(int64_t) * (size_t)n);
    memcpy(f->eq_copy, old->train.eq_id, sizeof(int64_t) * (size_t)na);
    memcpy(f->eq_copy + na, x->eq_id, sizeof(int64_t) * (size_t)m);
    f->train.eq_id = f->eq_copy;
  } else {
    f->train.eq_id = NULL;
  }
  if (nsc > 0) {
    f->scales_copy = malloc(sizeof(double) * (size_t)(n * nsc));
    for (int kc = 0; kc < nsc; ++kc) {
      memcpy(f->scales_copy + (int64_t)kc * n, old->train.scales + (int64_t)kc * na, sizeof(double) * (size_t)na);
      memcpy(f->scales_copy + (int64_t)kc * n + na, x->scales + (int64_t)kc * m, sizeof(double) * (size_t)m);
    }
    f->train.scales = f->scales_copy;
  }
  /* pred = _predict_impl(features, fit_, JointDistribution)                                       :389-390 */
  agp_features plain = *x;
  plain.is_measurement = 0;
  double *pmean = malloc(sizeof(double) * (size_t)m);
  double *S = malloc(sizeof(double) * (size_t)(m * m));
  orc_predict_joint(old, prog, n_nodes, &plain, pmean, S);
  /* delta = targets.mean - pred.mean; pred.covariance += targets.covariance; S_ldlt = ...ldlt()    :392-394 */
  double *delta = malloc(sizeof(double) * (size_t)m);
  for (int64_t i = 0; i < m; ++i) delta[i] = y[i] - pmean[i];
  if (y_var)
    for (int64_t i = 0; i < m; ++i) S[i + i * m] += y_var[i];
  f->S_ldlt = S;
  f->S_tr = malloc(sizeof(int64_t) * (size_t)m);
  orc_ldlt(f->S_ldlt, m, m, f->S_tr);
  /* cross = covariance_function_(fit_.train_features, features); BlockSymmetric(A, cross, S_ldlt)  :396-400 */
  f->Ai_B = malloc(sizeof(double) * (size_t)(na * m));
  orc_gram_cross(prog, n_nodes, &old->train, &plain, f->Ai_B, na);
  orc_fit_solve(old, f->Ai_B, m);                             /* Ai_B(A_.solve(B_)), block_symmetric.hpp:51 */
  /* Si_delta = S_ldlt.solve(delta); new_information = [information - Ai_B Si_delta; Si_delta]       :402-407 */
  orc_ldlt_solve(f->S_ldlt, m, m, f->S_tr, delta, 1, m);
  f->information = malloc(sizeof(double) * (size_t)n);
  for (int64_t i = 0; i < na; ++i) {
    double acc = 0.;
    for (int64_t c = 0; c < m; ++c) acc += f->Ai_B[i + c * na] * delta[c];
    f->information[i] = old->information[i] - acc;
  }
  memcpy(f->information + na, delta, sizeof(double) * (size_t)m);
  free(pmean); free(delta);
  return f;
}

/* negative_log_likelihood(deviation, covariance): likelihood.hpp:38-66 */
ORC_API double orc_nll_dense(const double *dev, const double *cov, int64_t n,
                             int64_t ld) {
  if (n == 1) { /* univariate shortcut, likelihood.hpp:57-60 */
    const double v = cov[0];
    return 0.5 * (log(2 * M_PI * v) + dev[0] * dev[0] / v);
  }
  double *A = malloc(sizeof(double) * (size_t)(n * n));
  for (int64_t j = 0; j < n; ++j)
    memcpy(A + j * n, cov + j * ld, sizeof(double) * (size_t)n);
  int64_t *tr = malloc(sizeof(int64_t) * (size_t)n);
  double *s = malloc(sizeof(double) * (size_t)n);
  memcpy(s, dev, sizeof(double) * (size_t)n);
  orc_ldlt(A, n, n, tr);
  orc_ldlt_solve(A, n, n, tr, s, 1, n);
  double maha = 0.;
  for (int64_t i = 0; i < n; ++i) maha += dev[i] * s[i];
  const double log_det = orc_ldlt_logdet(A, n, n);
  free(A); free(tr); free(s);
  return 0.5 * (log_det + maha + (double)n * log(2 * M_PI));
}

/* negative_log_likelihood(y, k(x, x) + diag(y_var)) with the features wrapped
 * as measurements; y_var may be NULL.  NOT a reference entry point by itself:
 * the checker of the C-ABI's agp_nll, which takes an optional variance. */
ORC_API double orc_nll_with_variance(const agp_kernel_node *prog, int n_nodes,
                                     const agp_features *x, const double *y,
                                     const double *y_var) {
  const int64_t n = x->n;
  agp_features meas = *x;
  meas.is_measurement = 1; /* as_measurements(dataset.features), gp.hpp:445 */
  double *K = malloc(sizeof(double) * (size_t)(n * n));
  orc_gram_sym(prog, n_nodes, &meas, K, n); /* gp.hpp:447 */
  if (y_var)
    for (int64_t i = 0; i < n; ++i) K[i + i * n] += y_var[i];
  const double out = orc_nll_dense(y, K, n, n); /* gp.hpp:448 */
  free(K);
  return out;
}

/* -log_likelihood without the prior term: GaussianProcessBase::log_likelihood,
 * gp.hpp:442-451.  y = targets.mean with the mean function already removed
 * (:446).  The covariance is covariance_function_(measurement_features) ALONE:
 * the reference does not add dataset.targets.covariance here. */
ORC_API double orc_nll(const agp_kernel_node *prog, int n_nodes,
                       const agp_features *x, const double *y) {
  return orc_nll_with_variance(prog, n_nodes, x, y, NULL);
}

/* ---------------------------------------------------------------------- */
/* mean functions: covariance_functions/mean_function.hpp,                 */
/* LinearMean polynomials.hpp:92-106                                       */
/* ---------------------------------------------------------------------- */
/* A mean function flattened to a postfix program like the covariance
 * functions: op 0 = ZeroMean (mean_function.hpp:274-276), 1 = LinearMean
 * {slope, offset} on the first coordinate (polynomials.hpp:103-105),
 * 2 = constant {value} (a user mean function with _call_impl = value),
 * 10 = SumOfMeanFunctions (:150-155), 11 = ProductOfMeanFunctions with the
 * `output != 0` short circuit (:221-227). */
typedef struct { int32_t op; int32_t pad; double params[2]; } orc_mean_node;

static double mean_eval(const orc_mean_node *prog, int n_nodes, const double *x) {
  double stack[16];
  int sp = 0;
  for (int t = 0; t < n_nodes; ++t) {
    const orc_mean_node *nd = &prog[t];
    switch (nd->op) {
    case 0: stack[sp++] = 0.; break;                                        /* :275 */
    case 1: stack[sp++] = nd->params[0] * x[0] + nd->params[1]; break;      /* polynomials.hpp:104 */
    case 2: stack[sp++] = nd->params[0]; break;
    case 10: { const double r = stack[--sp]; stack[sp - 1] = stack[sp - 1] + r; break; } /* :154 */
    case 11: { const double r = stack[--sp]; double o = stack[sp - 1];      /* :222-226 */
               if (o != 0.) o *= r;
               stack[sp - 1] = o; break; }
    default: return NAN;
    }
  }
  return stack[0];
}

/* MeanFunction::operator()(std::vector<X>) -> compute_mean_vector, mean_function.hpp:73-84
 * (Measurement<X> features are unwrapped by DefaultCaller, callers.hpp) */
ORC_API void orc_mean_vector(const orc_mean_node *prog, int n_nodes, const agp_features *x, double *out) {
  for (int64_t i = 0; i < x->n; ++i) out[i] = mean_eval(prog, n_nodes, x->coords + i * x->dim);
}

/* remove_from (:98-107) / add_to (:86-95): target -= / += mean(features); sign = -1 / +1.
 * A pure ZeroMean program returns without touching the target (:90-92, :101-103). */
ORC_API void orc_mean_apply(const orc_mean_node *prog, int n_nodes, const agp_features *x, double sign, double *target) {
  if (n_nodes == 1 && prog[0].op == 0) return;
  for (int64_t i = 0; i < x->n; ++i) target[i] += sign * mean_eval(prog, n_nodes, x->coords + i * x->dim);
}

/* gp.hpp:350-366 + 82-85 */
ORC_API void orc_predict_mean(const orc_fit *f, const agp_kernel_node *prog,
                              int n_nodes, const agp_features *xs,
                              double *mean) {
  const int64_t n = f->n, m = xs->n;
  double *Ks = malloc(sizeof(double) * (size_t)(n * m));
  orc_gram_cross(prog, n_nodes, &f->train, xs, Ks, n);
  for (int64_t j = 0; j < m; ++j) {
    double s = 0.;
    for (int64_t i = 0; i < n; ++i) s += Ks[i + j * n] * f->information[i];
    mean[j] = s;
  }
  free(Ks);
}

/* gp.hpp:326-348 + 87-101 */
ORC_API void orc_predict_marginal(const orc_fit *f, const agp_kernel_node *prog,
                                  int n_nodes, const agp_features *xs,
                                  double *mean, double *variance) {
  const int64_t n = f->n, m = xs->n;
  double *Ks = malloc(sizeof(double) * (size_t)(n * m));
  double *E = malloc(sizeof(double) * (size_t)(n * m));
  orc_gram_cross(prog, n_nodes, &f->train, xs, Ks, n);
  memcpy(E, Ks, sizeof(double) * (size_t)(n * m));
  orc_fit_solve(f, E, m); /* explained = train_covariance.solve(cross_cov) */
  orc_set XS = as_set(xs);
  for (int64_t j = 0; j < m; ++j) {
    double mu = 0., ex = 0.;
    for (int64_t i = 0; i < n; ++i) {
      mu += Ks[i + j * n] * f->information[i];
      ex += E[i + j * n] * Ks[i + j * n];
    }
    mean[j] = mu;
    variance[j] = eval_pair(prog, n_nodes, &XS, j, &XS, j) - ex; /* :339-343,99 */
  }
  free(Ks); free(E);
}

/* gp.hpp:305-324 + 103-113 */
ORC_API void orc_predict_joint(const orc_fit *f, const agp_kernel_node *prog,
                               int n_nodes, const agp_features *xs,
                               double *mean, double *cov) {
  const int64_t n = f->n, m = xs->n;
  double *Ks = malloc(sizeof(double) * (size_t)(n * m));
  double *E = malloc(sizeof(double) * (size_t)(n * m));
  orc_gram_cross(prog, n_nodes, &f->train, xs, Ks, n);
  memcpy(E, Ks, sizeof(double) * (size_t)(n * m));
  orc_fit_solve(f, E, m);
  orc_gram_sym(prog, n_nodes, xs, cov, m); /* prior_cov */
  for (int64_t j = 0; j < m; ++j) {
    double mu = 0.;
    for (int64_t i = 0; i < n; ++i) mu += Ks[i + j * n] * f->information[i];
    mean[j] = mu;
  }
  for (int64_t b = 0; b < m; ++b)
    for (int64_t a = 0; a < m; ++a) {
      double s = 0.;
      for (int64_t i = 0; i < n; ++i) s += Ks[i + a * n] * E[i + b * n];
      cov[a + b * m] -= s;
    }
  free(Ks); free(E);
}

/* ---------------------------------------------------------------------- */
/* leave-one-out fast path: eigen/serializable_ldlt.hpp:137-199,           */
/* evaluation/cross_validation_utils.hpp:132-163,165-232                   */
/* ---------------------------------------------------------------------- */
/* inverse_cholesky = D^-1/2 L^-1 P  (serializable_ldlt.hpp:154-163), column-major n x n into R */
static void ldlt_inverse_cholesky(const double *A, int64_t n, int64_t ld, const int64_t *tr, double *R) {
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i) R[i + j * n] = (i == j) ? 1. : 0.;
  /* P * I : apply the transpositions to the rows */
  for (int64_t k = 0; k < n; ++k)
    if (tr[k] != k)
      for (int64_t c = 0; c < n; ++c) {
        const double t = R[k + c * n];
        R[k + c * n] = R[tr[k] + c * n];
        R[tr[k] + c * n] = t;
      }
  for (int64_t c = 0; c < n; ++c) { /* unit-lower solve, column by column */
    double *b = R + c * n;
    for (int64_t j = 0; j < n; ++j) {
      const double bj = b[j];
      if (bj == 0.) continue;
      const double *col = A + j * ld;
      for (int64_t i = j + 1; i < n; ++i) b[i] -= col[i] * bj;
    }
  }
  for (int64_t i = 0; i < n; ++i) { /* diagonal_sqrt_inverse, :50-68 */
    const double d = A[i + i * ld];
    const double sc = d > 0. ? 1. / sqrt(d) : 0.;
    for (int64_t c = 0; c < n; ++c) R[i + c * n] *= sc;
  }
}

/* SerializableLDLT::inverse_diagonal (:181-199): blocks of one index each */
ORC_API void orc_fit_inverse_diagonal(const orc_fit *f, double *out) {
  const int64_t n = f->n;
  double *R = malloc(sizeof(double) * (size_t)(n * n));
  if (f->use_llt) {
    /* R = L^-1 */
    for (int64_t j = 0; j < n; ++j)
      for (int64_t i = 0; i < n; ++i) R[i + j * n] = (i == j) ? 1. : 0.;
    for (int64_t c = 0; c < n; ++c) {
      double *b = R + c * n;
      for (int64_t j = c; j < n; ++j) {
        b[j] /= f->ldlt[j + j * n];
        const double bj = b[j];
        const double *col = f->ldlt + j * n;
        for (int64_t i = j + 1; i < n; ++i) b[i] -= col[i] * bj;
      }
    }
  } else {
    ldlt_inverse_cholesky(f->ldlt, n, n, f->tr, R);
  }
  for (int64_t c = 0; c < n; ++c) { /* sub_matrix^T * sub_matrix for a single column */
    double s = 0.;
    for (int64_t i = 0; i < n; ++i) s += R[i + c * n] * R[i + c * n];
    out[c] = s;
  }
  free(R);
}

/* held_out_predictions with singleton groups (cross_validation_utils.hpp:199-232,
 * held_out_prediction :171-186): mean_i = y_i - v_i / Ainv_ii, var_i = 1 / Ainv_ii,
 * == leave_one_out_conditional (:138-163, GPML eq. 5.12) */
ORC_API void orc_fit_loo_marginal(const orc_fit *f, const double *y, double *mean, double *variance) {
  const int64_t n = f->n;
  double *d = malloc(sizeof(double) * (size_t)n);
  orc_fit_inverse_diagonal(f, d);
  for (int64_t i = 0; i < n; ++i) {
    variance[i] = 1. / d[i];
    mean[i] = y[i] - f->information[i] / d[i];
  }
  free(d);
}

/* R^-1 = D^-1/2 L^-1 P (pivoted LDLT) or L^-1 (LLT) of the fit's covariance, n x n column-major
 * (serializable_ldlt.hpp:154-165) */
static double *fit_inverse_cholesky(const orc_fit *f) {
  const int64_t n = f->n;
  double *R = malloc(sizeof(double) * (size_t)(n * n));
  if (f->use_llt) {
    for (int64_t j = 0; j < n; ++j)
      for (int64_t i = 0; i < n; ++i) R[i + j * n] = (i == j) ? 1. : 0.;
    for (int64_t c = 0; c < n; ++c) {
      double *b = R + c * n;
      for (int64_t j = c; j < n; ++j) {
        b[j] /= f->ldlt[j + j * n];
        const double bj = b[j];
        const double *col = f->ldlt + j * n;
        for (int64_t i = j + 1; i < n; ++i) b[i] -= col[i] * bj;
      }
    }
  } else {
    ldlt_inverse_cholesky(f->ldlt, n, n, f->tr, R);
  }
  return R;
}

/* SerializableLDLT::inverse_blocks (serializable_ldlt.hpp:137-179): for every index group the
 * block (K^-1)[I_g, I_g] = sub_matrix^T sub_matrix of the columns I_g of R^-1.
 * offsets has n_groups + 1 entries into indices; blocks are written column-major, concatenated. */
ORC_API void orc_fit_inverse_blocks(const orc_fit *f, int64_t n_groups, const int64_t *offsets,
                                    const int64_t *indices, double *out) {
  const int64_t n = f->n;
  double *R = fit_inverse_cholesky(f);
  for (int64_t g = 0; g < n_groups; ++g) {
    const int64_t m = offsets[g + 1] - offsets[g];
    const int64_t *idx = indices + offsets[g];
    for (int64_t b = 0; b < m; ++b)
      for (int64_t a = 0; a < m; ++a) {
        const double *ca = R + idx[a] * n, *cb = R + idx[b] * n;
        double s = 0.;
        for (int64_t i = 0; i < n; ++i) s += ca[i] * cb[i];
        out[a + b * m] = s;
      }
    out += m * m;
  }
  free(R);
}

/* held_out_predictions for arbitrary groups (cross_validation_utils.hpp:165-232):
 *   A = inverse block of the group, v = information[I_g], y = targets[I_g]
 *   mean = y - A.ldlt().solve(v)                       (:171-176)
 *   marginal variance = SerializableLDLT(A).inverse_diagonal()   (:178-186)
 *   joint covariance  = A.inverse()                    (:188-197)
 * mean / variance are written in the order of `indices`; joint (may be NULL) as concatenated
 * column-major blocks.  Returns 0 if a block is not invertible. */
ORC_API int orc_fit_held_out(const orc_fit *f, const double *y, int64_t n_groups, const int64_t *offsets,
                             const int64_t *indices, double *mean, double *variance, double *joint) {
  const int64_t total = offsets[n_groups];
  int64_t blk_elems = 0;
  for (int64_t g = 0; g < n_groups; ++g) {
    const int64_t m = offsets[g + 1] - offsets[g];
    blk_elems += m * m;
  }
  double *blocks = malloc(sizeof(double) * (size_t)(blk_elems > 0 ? blk_elems : 1));
  orc_fit_inverse_blocks(f, n_groups, offsets, indices, blocks);
  int ok = 1;
  const double *A = blocks;
  (void)total;
  for (int64_t g = 0; g < n_groups; ++g) {
    const int64_t m = offsets[g + 1] - offsets[g];
    const int64_t *idx = indices + offsets[g];
    double *P = malloc(sizeof(double) * (size_t)(m * m));
    int64_t *tr = malloc(sizeof(int64_t) * (size_t)m);
    double *v = malloc(sizeof(double) * (size_t)m);
    double *Rg = malloc(sizeof(double) * (size_t)(m * m));
    memcpy(P, A, sizeof(double) * (size_t)(m * m));
    if (!orc_ldlt(P, m, m, tr)) ok = 0;
    for (int64_t a = 0; a < m; ++a) v[a] = f->information[idx[a]];
    orc_ldlt_solve(P, m, m, tr, v, 1, m);
    for (int64_t a = 0; a < m; ++a) mean[offsets[g] + a] = y[idx[a]] - v[a];
    ldlt_inverse_cholesky(P, m, m, tr, Rg); /* columns of R^-1 of the block */
    if (variance)
      for (int64_t a = 0; a < m; ++a) {
        double s = 0.;
        for (int64_t i = 0; i < m; ++i) s += Rg[i + a * m] * Rg[i + a * m];
        variance[offsets[g] + a] = s;
      }
    if (joint) {
      for (int64_t b = 0; b < m; ++b)
        for (int64_t a = 0; a < m; ++a) {
          double s = 0.;
          for (int64_t i = 0; i < m; ++i) s += Rg[i + a * m] * Rg[i + b * m];
          joint[a + b * m] = s;
        }
      joint += m * m;
    }
    free(P); free(tr); free(v); free(Rg);
    A += m * m;
  }
  free(blocks);
  return ok;
}

/* ====================================================================== */
/* Sparse Gaussian process (FITC / PITC): models/sparse_gp.hpp             */
/* ====================================================================== */
/* SerializableLDLT::sqrt_solve (serializable_ldlt.hpp:99-109): B <- D^-1/2 L^-1 P B, in place */
static void ldlt_sqrt_solve(const double *A, int64_t n, int64_t ld, const int64_t *tr, double *B, int64_t nrhs,
                            int64_t ldb) {
  for (int64_t c = 0; c < nrhs; ++c) {
    double *b = B + c * ldb;
    for (int64_t k = 0; k < n; ++k)
      if (tr[k] != k) { const double t = b[k]; b[k] = b[tr[k]]; b[tr[k]] = t; }
    for (int64_t j = 0; j < n; ++j) {
      const double bj = b[j];
      const double *col = A + j * ld;
      for (int64_t i = j + 1; i < n; ++i) b[i] -= col[i] * bj;
    }
    for (int64_t i = 0; i < n; ++i) { /* diagonal_sqrt_inverse (:58-69): 1 / sqrt(D_i) where D_i > 0, else 0 */
      const double d = A[i + i * ld];
      b[i] = d > 0. ? b[i] * (1. / sqrt(d)) : 0.;
    }
  }
}

ORC_API void orc_ldlt_sqrt_solve(const double *A, int64_t n, int64_t ld, const int64_t *tr, double *B, int64_t nrhs,
                                 int64_t ldb) {
  ldlt_sqrt_solve(A, n, ld, tr, B, nrhs, ldb);
}

/* SerializableLDLT::sqrt_transpose (:111-115): out = D^1/2 (P^T L)^T, n x n column-major */
static void ldlt_sqrt_transpose(const double *A, int64_t n, int64_t ld, const int64_t *tr, double *out) {
  double *M = malloc(sizeof(double) * (size_t)(n * n));
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i) M[i + j * n] = (i == j) ? 1. : (i > j ? A[i + j * ld] : 0.);
  for (int64_t k = n - 1; k >= 0; --k) /* P^T: the transpositions in reverse order, on the rows */
    if (tr[k] != k)
      for (int64_t c = 0; c < n; ++c) {
        const double t = M[k + c * n];
        M[k + c * n] = M[tr[k] + c * n];
        M[tr[k] + c * n] = t;
      }
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < n; ++i) out[i + j * n] = sqrt(A[i + i * ld]) * M[j + i * n];
  free(M);
}

/* Householder QR with column pivoting (the published algorithm behind Eigen::ColPivHouseholderQR,
 * which DenseQRImplementation uses, sparse_gp.hpp:80-88): at every step the remaining column of
 * largest norm is brought forward.  Eigen down-dates the column norms; here they are recomputed,
 * which picks the same pivots except on ties at rounding level.
 * B (rows x cols, ld) is overwritten: R in the upper triangle, Householder vectors below;
 * perm[k] = original index of the column at position k. */
typedef struct {
  int64_t rows, cols;
  double *qr;   /* rows x cols */
  double *tau;  /* cols */
  int64_t *perm;
  int64_t nonzero_pivots, rank;
} orc_qr;

static orc_qr *colpiv_qr(const double *B, int64_t rows, int64_t cols) {
  orc_qr *q = calloc(1, sizeof(orc_qr));
  q->rows = rows; q->cols = cols;
  q->qr = malloc(sizeof(double) * (size_t)(rows * cols));
  memcpy(q->qr, B, sizeof(double) * (size_t)(rows * cols));
  q->tau = calloc((size_t)cols, sizeof(double));
  q->perm = malloc(sizeof(int64_t) * (size_t)cols);
  for (int64_t j = 0; j < cols; ++j) q->perm[j] = j;
  double *A = q->qr;
  double max_norm = 0.;
  for (int64_t j = 0; j < cols; ++j) {
    double s = 0.;
    for (int64_t i = 0; i < rows; ++i) s += A[i + j * rows] * A[i + j * rows];
    if (sqrt(s) > max_norm) max_norm = sqrt(s);
  }
  const double eps = 2.220446049250313e-16;
  const double threshold_helper = (max_norm * eps / (double)rows) * (max_norm * eps / (double)rows);
  q->nonzero_pivots = cols;
  double maxpivot = 0.;
  for (int64_t k = 0; k < cols; ++k) {
    int64_t best = k;
    double best_sq = -1.;
    for (int64_t j = k; j < cols; ++j) {
      double s = 0.;
      for (int64_t i = k; i < rows; ++i) s += A[i + j * rows] * A[i + j * rows];
      if (s > best_sq) { best_sq = s; best = j; }
    }
    if (q->nonzero_pivots == cols && best_sq < threshold_helper * (double)(rows - k)) q->nonzero_pivots = k;
    if (best != k) {
      for (int64_t i = 0; i < rows; ++i) {
        const double t = A[i + k * rows]; A[i + k * rows] = A[i + best * rows]; A[i + best * rows] = t;
      }
      const int64_t t = q->perm[k]; q->perm[k] = q->perm[best]; q->perm[best] = t;
    }
    /* makeHouseholderInPlace */
    double *x = A + k + k * rows;
    const int64_t len = rows - k;
    double tail = 0.;
    for (int64_t i = 1; i < len; ++i) tail += x[i] * x[i];
    double beta, tau;
    if (tail <= 2.2250738585072014e-308) {
      tau = 0.; beta = x[0];
      for (int64_t i = 1; i < len; ++i) x[i] = 0.;
    } else {
      beta = sqrt(x[0] * x[0] + tail);
      if (x[0] >= 0.) beta = -beta;
      for (int64_t i = 1; i < len; ++i) x[i] /= (x[0] - beta);
      tau = (beta - x[0]) / beta;
    }
    x[0] = beta;
    q->tau[k] = tau;
    /* apply H = I - tau v v^T (v = [1; essential]) to the remaining columns */
    for (int64_t j = k + 1; j < cols; ++j) {
      double *c = A + k + j * rows;
      double s = c[0];
      for (int64_t i = 1; i < len; ++i) s += x[i] * c[i];
      s *= tau;
      c[0] -= s;
      for (int64_t i = 1; i < len; ++i) c[i] -= s * x[i];
    }
    if (fabs(beta) > maxpivot) maxpivot = fabs(beta);
  }
  const double thr = maxpivot * eps * (double)cols; /* rank(): |R_ii| > maxpivot * eps * diagSize */
  q->rank = 0;
  for (int64_t k = 0; k < cols; ++k)
    if (fabs(A[k + k * rows]) > thr) ++q->rank;
  return q;
}

static void qr_free(orc_qr *q) {
  if (!q) return;
  free(q->qr); free(q->tau); free(q->perm); free(q);
}

/* ColPivHouseholderQR::solve: dst = P [R11^-1 (Q^T rhs)_{1:np}; 0] */
static void qr_solve(const orc_qr *q, const double *rhs, double *dst) {
  const int64_t rows = q->rows, cols = q->cols, np = q->nonzero_pivots;
  double *c = malloc(sizeof(double) * (size_t)rows);
  memcpy(c, rhs, sizeof(double) * (size_t)rows);
  for (int64_t k = 0; k < np; ++k) {
    const double *v = q->qr + k + k * rows;
    double s = c[k];
    for (int64_t i = 1; i < rows - k; ++i) s += v[i] * c[k + i];
    s *= q->tau[k];
    c[k] -= s;
    for (int64_t i = 1; i < rows - k; ++i) c[k + i] -= s * v[i];
  }
  for (int64_t i = np - 1; i >= 0; --i) {
    double s = c[i];
    for (int64_t j = i + 1; j < np; ++j) s -= q->qr[i + j * rows] * c[j];
    c[i] = s / q->qr[i + i * rows];
  }
  for (int64_t i = 0; i < cols; ++i) dst[q->perm[i]] = (i < np) ? c[i] : 0.;
  free(c);
}

/* sqrt_solve(R, P, rhs) = R^-T P^T rhs (linalg/qr_utils.hpp:37-45), rhs m x nrhs in place */
static void qr_sqrt_solve(const double *R, const int64_t *perm, int64_t m, double *rhs, int64_t nrhs, int64_t ld) {
  double *t = malloc(sizeof(double) * (size_t)m);
  for (int64_t c = 0; c < nrhs; ++c) {
    double *b = rhs + c * ld;
    for (int64_t i = 0; i < m; ++i) t[i] = b[perm[i]]; /* P^T rhs */
    for (int64_t i = 0; i < m; ++i) {                  /* R^T lower-triangular solve */
      double s = t[i];
      for (int64_t j = 0; j < i; ++j) s -= R[j + i * m] * t[j];
      t[i] = s / R[i + i * m];
    }
    memcpy(b, t, sizeof(double) * (size_t)m);
  }
  free(t);
}

typedef struct {
  int64_t m;             /* inducing points */
  agp_features u;        /* train_features = inducing points (owned copies) */
  double *coords_copy; int64_t *eq_copy; double *scales_copy;
  double *kuu_ldlt; int64_t *kuu_tr;   /* train_covariance */
  double *R; int64_t *perm;            /* R (m x m upper), P */
  double *information;
  int64_t numerical_rank;
  double nll;                          /* of the data the fit was made from */
} orc_sparse_fit;

typedef struct {
  int64_t n, m, n_groups;
  int64_t *order, *offsets;      /* reordered_inds; group g = order[offsets[g] .. offsets[g+1]) */
  double **a_ldlt; int64_t **a_tr; /* BlockDiagonalLDLT A */
  double *kuu_ldlt; int64_t *kuu_tr;
  double *K_fu;                  /* n x m, rows in reordered order */
  double *y;                     /* reordered targets */
} sparse_parts;

static int cmp_key_index(const void *a, const void *b) {
  const int64_t *x = a, *y = b;
  if (x[0] != y[0]) return x[0] < y[0] ? -1 : 1;
  return x[1] < y[1] ? -1 : (x[1] > y[1]);
}

static void sparse_parts_free(sparse_parts *p) {
  if (!p) return;
  for (int64_t g = 0; g < p->n_groups; ++g) { free(p->a_ldlt[g]); free(p->a_tr[g]); }
  free(p->a_ldlt); free(p->a_tr); free(p->order); free(p->offsets);
  free(p->kuu_ldlt); free(p->kuu_tr); free(p->K_fu); free(p->y);
  free(p);
}

static agp_features subset_features(const agp_features *x, const int64_t *idx, int64_t cnt, int is_measurement,
                                    double **coords, int64_t **eq, double **scales) {
  agp_features s = *x;
  s.n = cnt; s.is_measurement = is_measurement;
  *coords = malloc(sizeof(double) * (size_t)(cnt * x->dim + 1));
  for (int64_t a = 0; a < cnt; ++a)
    memcpy(*coords + a * x->dim, x->coords + idx[a] * x->dim, sizeof(double) * (size_t)x->dim);
  s.coords = *coords;
  *eq = NULL; *scales = NULL;
  if (x->eq_id) {
    *eq = malloc(sizeof(int64_t) * (size_t)(cnt + 1));
    for (int64_t a = 0; a < cnt; ++a) (*eq)[a] = x->eq_id[idx[a]];
    s.eq_id = *eq;
  }
  if (x->scales && x->n_scale_columns > 0) {
    const int c = x->n_scale_columns;
    *scales = malloc(sizeof(double) * (size_t)(cnt * c + 1));
    for (int64_t a = 0; a < cnt; ++a) memcpy(*scales + a * c, x->scales + idx[a] * c, sizeof(double) * (size_t)c);
    s.scales = *scales;
  }
  return s;
}

/* compute_internal_components (sparse_gp.hpp:631-706).  group_key[i] is the grouper's value for
 * feature i; groups are visited in ascending key order (std::map), members in ascending index
 * order.  y is copied BEFORE the mean function is removed (:664-668), so the caller passes the
 * raw target means. */
static sparse_parts *sparse_components(const agp_kernel_node *prog, int n_nodes, const agp_features *x,
                                       const int64_t *group_key, const double *y, const double *y_var,
                                       const agp_features *u, double measurement_nugget, double inducing_nugget) {
  const int64_t n = x->n, m = u->n;
  sparse_parts *p = calloc(1, sizeof(sparse_parts));
  p->n = n; p->m = m;
  int64_t *ki = malloc(sizeof(int64_t) * 2 * (size_t)(n + 1));
  for (int64_t i = 0; i < n; ++i) { ki[2 * i] = group_key[i]; ki[2 * i + 1] = i; }
  qsort(ki, (size_t)n, 2 * sizeof(int64_t), cmp_key_index);
  p->order = malloc(sizeof(int64_t) * (size_t)(n + 1));
  p->offsets = malloc(sizeof(int64_t) * (size_t)(n + 2));
  p->n_groups = 0;
  for (int64_t i = 0; i < n; ++i) {
    p->order[i] = ki[2 * i + 1];
    if (i == 0 || ki[2 * i] != ki[2 * (i - 1)]) p->offsets[p->n_groups++] = i;
  }
  p->offsets[p->n_groups] = n;
  free(ki);
  /* reordered measurement features, K_fu, K_uu + nugget, P = K_uu^-1/2 K_uf */
  double *cc; int64_t *ee; double *ss;
  agp_features xr = subset_features(x, p->order, n, 1, &cc, &ee, &ss);
  p->y = malloc(sizeof(double) * (size_t)(n + 1));
  for (int64_t i = 0; i < n; ++i) p->y[i] = y[p->order[i]];
  p->K_fu = malloc(sizeof(double) * (size_t)(n * m + 1));
  orc_gram_cross(prog, n_nodes, &xr, u, p->K_fu, n);
  p->kuu_ldlt = malloc(sizeof(double) * (size_t)(m * m));
  p->kuu_tr = malloc(sizeof(int64_t) * (size_t)m);
  orc_gram_sym(prog, n_nodes, u, p->kuu_ldlt, m);
  for (int64_t i = 0; i < m; ++i) p->kuu_ldlt[i + i * m] += inducing_nugget;
  orc_ldlt(p->kuu_ldlt, m, m, p->kuu_tr);
  double *P = malloc(sizeof(double) * (size_t)(m * n + 1)); /* m x n */
  for (int64_t j = 0; j < n; ++j)
    for (int64_t i = 0; i < m; ++i) P[i + j * m] = p->K_fu[j + i * n];
  ldlt_sqrt_solve(p->kuu_ldlt, m, m, p->kuu_tr, P, n, m);
  /* A = K_ff (block diagonal, + target variance) - diag blocks of P^T P + measurement nugget */
  p->a_ldlt = calloc((size_t)p->n_groups, sizeof(double *));
  p->a_tr = calloc((size_t)p->n_groups, sizeof(int64_t *));
  for (int64_t g = 0; g < p->n_groups; ++g) {
    const int64_t o = p->offsets[g], s = p->offsets[g + 1] - o;
    double *c2; int64_t *e2; double *s2;
    agp_features xg = subset_features(x, p->order + o, s, 1, &c2, &e2, &s2);
    double *A = malloc(sizeof(double) * (size_t)(s * s));
    orc_gram_sym(prog, n_nodes, &xg, A, s);
    for (int64_t a = 0; a < s; ++a) A[a + a * s] += y_var ? y_var[p->order[o + a]] : 0.;
    for (int64_t b = 0; b < s; ++b)
      for (int64_t a = 0; a < s; ++a) {
        double q = 0.;
        for (int64_t k = 0; k < m; ++k) q += P[k + (o + a) * m] * P[k + (o + b) * m];
        A[a + b * s] -= q;
      }
    for (int64_t a = 0; a < s; ++a) A[a + a * s] += measurement_nugget;
    p->a_tr[g] = malloc(sizeof(int64_t) * (size_t)s);
    orc_ldlt(A, s, s, p->a_tr[g]);
    p->a_ldlt[g] = A;
    free(c2); free(e2); free(s2);
  }
  free(P); free(cc); free(ee); free(ss);
  return p;
}

/* compute_sigma_qr (sparse_gp.hpp:343-352): B = [A^-1/2 K_fu; K_uu^T/2] */
static orc_qr *sparse_sigma_qr(const sparse_parts *p) {
  const int64_t n = p->n, m = p->m, rows = n + m;
  double *B = malloc(sizeof(double) * (size_t)(rows * m));
  for (int64_t g = 0; g < p->n_groups; ++g) {
    const int64_t o = p->offsets[g], s = p->offsets[g + 1] - o;
    double *blk = malloc(sizeof(double) * (size_t)(s * m));
    for (int64_t c = 0; c < m; ++c)
      for (int64_t a = 0; a < s; ++a) blk[a + c * s] = p->K_fu[(o + a) + c * n];
    ldlt_sqrt_solve(p->a_ldlt[g], s, s, p->a_tr[g], blk, m, s);
    for (int64_t c = 0; c < m; ++c)
      for (int64_t a = 0; a < s; ++a) B[(o + a) + c * rows] = blk[a + c * s];
    free(blk);
  }
  double *st = malloc(sizeof(double) * (size_t)(m * m));
  ldlt_sqrt_transpose(p->kuu_ldlt, m, m, p->kuu_tr, st);
  for (int64_t c = 0; c < m; ++c)
    for (int64_t a = 0; a < m; ++a) B[(n + a) + c * rows] = st[a + c * m];
  free(st);
  orc_qr *q = colpiv_qr(B, rows, m);
  free(B);
  return q;
}

/* A_ldlt.sqrt_solve(y) / A_ldlt.solve(y), block by block */
static void sparse_a_apply(const sparse_parts *p, const double *y, double *out, int full_solve) {
  memcpy(out, y, sizeof(double) * (size_t)p->n);
  for (int64_t g = 0; g < p->n_groups; ++g) {
    const int64_t o = p->offsets[g], s = p->offsets[g + 1] - o;
    if (full_solve) orc_ldlt_solve(p->a_ldlt[g], s, s, p->a_tr[g], out + o, 1, s);
    else ldlt_sqrt_solve(p->a_ldlt[g], s, s, p->a_tr[g], out + o, 1, s);
  }
}

/* log_likelihood (sparse_gp.hpp:524-596), returned as the NEGATIVE log likelihood, without priors */
static double sparse_nll_from(const sparse_parts *p, const orc_qr *q) {
  const int64_t n = p->n, m = p->m, rows = q->rows;
  double log_det_a = 0.;
  for (int64_t g = 0; g < p->n_groups; ++g) {
    const int64_t s = p->offsets[g + 1] - p->offsets[g];
    log_det_a += orc_ldlt_logdet(p->a_ldlt[g], s, s);
  }
  double log_det_r = 0.;
  for (int64_t i = 0; i < m; ++i) log_det_r += log(fabs(q->qr[i + i * rows]));
  const double log_det = log_det_a + 2. * log_det_r - orc_ldlt_logdet(p->kuu_ldlt, m, m);
  double *y_a = malloc(sizeof(double) * (size_t)n), *y_b = calloc((size_t)m, sizeof(double));
  sparse_a_apply(p, p->y, y_a, 1);
  for (int64_t c = 0; c < m; ++c)
    for (int64_t i = 0; i < n; ++i) y_b[c] += p->K_fu[i + c * n] * y_a[i];
  double *R = malloc(sizeof(double) * (size_t)(m * m));
  for (int64_t j = 0; j < m; ++j)
    for (int64_t i = 0; i < m; ++i) R[i + j * m] = (i <= j) ? q->qr[i + j * rows] : 0.;
  qr_sqrt_solve(R, q->perm, m, y_b, 1, m);
  double quad = 0.;
  for (int64_t i = 0; i < n; ++i) quad += p->y[i] * y_a[i];
  for (int64_t i = 0; i < m; ++i) quad -= y_b[i] * y_b[i];
  free(y_a); free(y_b); free(R);
  return 0.5 * (log_det + quad + (double)n * log(2. * M_PI));
}

/* _fit_impl (sparse_gp.hpp:354-381) */
ORC_API orc_sparse_fit *orc_sparse_fit_create(const agp_kernel_node *prog, int n_nodes, const agp_features *x,
                                              const int64_t *group_key, const double *y, const double *y_var,
                                              const agp_features *u, double measurement_nugget,
                                              double inducing_nugget) {
  const int64_t n = x->n, m = u->n;
  sparse_parts *p = sparse_components(prog, n_nodes, x, group_key, y, y_var, u, measurement_nugget, inducing_nugget);
  orc_qr *q = sparse_sigma_qr(p);
  double *y_aug = calloc((size_t)(n + m), sizeof(double));
  sparse_a_apply(p, p->y, y_aug, 0);
  orc_sparse_fit *f = calloc(1, sizeof(orc_sparse_fit));
  f->m = m;
  f->information = malloc(sizeof(double) * (size_t)m);
  qr_solve(q, y_aug, f->information);
  f->R = malloc(sizeof(double) * (size_t)(m * m));
  for (int64_t j = 0; j < m; ++j)
    for (int64_t i = 0; i < m; ++i) f->R[i + j * m] = (i <= j) ? q->qr[i + j * q->rows] : 0.;
  f->perm = malloc(sizeof(int64_t) * (size_t)m);
  memcpy(f->perm, q->perm, sizeof(int64_t) * (size_t)m);
  f->numerical_rank = q->rank;
  f->nll = sparse_nll_from(p, q);
  f->kuu_ldlt = p->kuu_ldlt; p->kuu_ldlt = NULL;
  f->kuu_tr = p->kuu_tr; p->kuu_tr = NULL;
  int64_t *all = malloc(sizeof(int64_t) * (size_t)(m + 1));
  for (int64_t i = 0; i < m; ++i) all[i] = i;
  f->u = subset_features(u, all, m, u->is_measurement, &f->coords_copy, &f->eq_copy, &f->scales_copy);
  free(all); free(y_aug);
  qr_free(q);
  sparse_parts_free(p);
  return f;
}

/* SparseGaussianProcessRegression::fit_from_prediction (sparse_gp.hpp:406-461), which rebase_inducing_points
 * (:714-725) calls with the old fit's joint prediction at the new inducing points z:
 *   train_covariance = LDLT(K_zz)  (no nugget),  information = train_covariance.solve(mean),
 *   C = covariance + DEFAULT_NUGGET I,  B_z = C^-1/2 K_zz = C_ldlt.sqrt_solve(K_zz),  (R, P) = QR(B_z).
 * cov is m x m column-major (both triangles). */
ORC_API orc_sparse_fit *orc_sparse_fit_from_prediction(const agp_kernel_node *prog, int n_nodes, const agp_features *z,
                                                       const double *mean, const double *cov) {
  const int64_t m = z->n;
  orc_sparse_fit *f = calloc(1, sizeof(orc_sparse_fit));
  f->m = m;
  double *K = malloc(sizeof(double) * (size_t)(m * m + 1));
  orc_gram_sym(prog, n_nodes, z, K, m);                                 /* K_zz                  :416-417 */
  f->kuu_ldlt = malloc(sizeof(double) * (size_t)(m * m + 1));
  memcpy(f->kuu_ldlt, K, sizeof(double) * (size_t)(m * m));
  f->kuu_tr = malloc(sizeof(int64_t) * (size_t)(m + 1));
  orc_ldlt(f->kuu_ldlt, m, m, f->kuu_tr);                               /* train_covariance      :418 */
  f->information = malloc(sizeof(double) * (size_t)(m + 1));
  memcpy(f->information, mean, sizeof(double) * (size_t)m);
  orc_ldlt_solve(f->kuu_ldlt, m, m, f->kuu_tr, f->information, 1, m);   /* information           :426 */
  double *C = malloc(sizeof(double) * (size_t)(m * m + 1));
  memcpy(C, cov, sizeof(double) * (size_t)(m * m));
  for (int64_t i = 0; i < m; ++i) C[i + i * m] += 1e-8;                 /* DEFAULT_NUGGET        :20, 423-425 */
  int64_t *c_tr = malloc(sizeof(int64_t) * (size_t)(m + 1));
  orc_ldlt(C, m, m, c_tr);                                              /* C_ldlt                :452 */
  ldlt_sqrt_solve(C, m, m, c_tr, K, m, m);                              /* sigma_inv_sqrt        :453 */
  orc_qr *q = colpiv_qr(K, m, m);                                       /* B_qr                  :454 */
  f->R = malloc(sizeof(double) * (size_t)(m * m + 1));
  for (int64_t j = 0; j < m; ++j)
    for (int64_t i = 0; i < m; ++i) f->R[i + j * m] = (i <= j) ? q->qr[i + j * m] : 0.;
  f->perm = malloc(sizeof(int64_t) * (size_t)(m + 1));
  memcpy(f->perm, q->perm, sizeof(int64_t) * (size_t)m);
  f->numerical_rank = q->rank;
  f->nll = NAN;
  int64_t *all = malloc(sizeof(int64_t) * (size_t)(m + 1));
  for (int64_t i = 0; i < m; ++i) all[i] = i;
  f->u = subset_features(z, all, m, z->is_measurement, &f->coords_copy, &f->eq_copy, &f->scales_copy);
  free(all); free(c_tr); free(C); free(K);
  qr_free(q);
  return f;
}

ORC_API void orc_sparse_fit_destroy(orc_sparse_fit *f) {
  if (!f) return;
  free(f->coords_copy); free(f->eq_copy); free(f->scales_copy);
  free(f->kuu_ldlt); free(f->kuu_tr); free(f->R); free(f->perm); free(f->information);
  free(f);
}

ORC_API void orc_sparse_fit_information(const orc_sparse_fit *f, double *out) {
  memcpy(out, f->information, sizeof(double) * (size_t)f->m);
}
ORC_API int64_t orc_sparse_fit_rank(const orc_sparse_fit *f) { return f->numerical_rank; }
ORC_API double orc_sparse_fit_nll(const orc_sparse_fit *f) { return f->nll; }

/* _predict_impl x 3 (sparse_gp.hpp:447-521); mean function handled by the caller.
 * variance / cov may be NULL (mean only); joint != 0 fills cov (M x M), else variance (M). */
ORC_API void orc_sparse_predict(const orc_sparse_fit *f, const agp_kernel_node *prog, int n_nodes,
                                const agp_features *xs, double *mean, double *variance, double *cov) {
  const int64_t m = f->m, M = xs->n;
  double *cross = malloc(sizeof(double) * (size_t)(m * M + 1));
  orc_gram_cross(prog, n_nodes, &f->u, xs, cross, m);
  for (int64_t j = 0; j < M; ++j) {
    double s = 0.;
    for (int64_t i = 0; i < m; ++i) s += cross[i + j * m] * f->information[i];
    mean[j] = s;
  }
  if (variance || cov) {
    double *Q = malloc(sizeof(double) * (size_t)(m * M + 1)), *S = malloc(sizeof(double) * (size_t)(m * M + 1));
    memcpy(Q, cross, sizeof(double) * (size_t)(m * M));
    memcpy(S, cross, sizeof(double) * (size_t)(m * M));
    ldlt_sqrt_solve(f->kuu_ldlt, m, m, f->kuu_tr, Q, M, m); /* Q_sqrt */
    qr_sqrt_solve(f->R, f->perm, m, S, M, m);                /* S_sqrt */
    orc_set XS = as_set(xs);
    if (variance)
      for (int64_t j = 0; j < M; ++j) {
        double qd = 0., sd = 0.;
        for (int64_t i = 0; i < m; ++i) { qd += Q[i + j * m] * Q[i + j * m]; sd += S[i + j * m] * S[i + j * m]; }
        variance[j] = eval_pair(prog, n_nodes, &XS, j, &XS, j) - qd + sd;
      }
    if (cov) {
      orc_gram_sym(prog, n_nodes, xs, cov, M);
      for (int64_t b = 0; b < M; ++b)
        for (int64_t a = 0; a < M; ++a) {
          double qd = 0., sd = 0.;
          for (int64_t i = 0; i < m; ++i) { qd += Q[i + a * m] * Q[i + b * m]; sd += S[i + a * m] * S[i + b * m]; }
          cov[a + b * M] += sd - qd;
        }
    }
    free(Q); free(S);
  }
  free(cross);
}

/* _update_impl (sparse_gp.hpp:322-371): B = [R_old P_old^T; A^-1/2 K_fu], y_aug = [R_old P_old^T v_old; A^-1/2 y];
 * the new fit keeps the old inducing points and train_covariance.  The inducing nugget passed here is the
 * model's current one (compute_internal_components recomputes K_uu_ldlt for P, :674-685). */
ORC_API orc_sparse_fit *orc_sparse_fit_update(const orc_sparse_fit *old, const agp_kernel_node *prog, int n_nodes,
                                              const agp_features *x, const int64_t *group_key, const double *y,
                                              const double *y_var, double measurement_nugget,
                                              double inducing_nugget) {
  const int64_t n = x->n, m = old->m;
  sparse_parts *p = sparse_components(prog, n_nodes, x, group_key, y, y_var, &old->u, measurement_nugget,
                                      inducing_nugget);
  const int64_t rows = m + n;
  double *B = calloc((size_t)(rows * m), sizeof(double));
  /* top: R_old P_old^T   (P^T permutes the columns: column perm[i] of the product is column i of R) */
  for (int64_t i = 0; i < m; ++i)
    for (int64_t r = 0; r < m; ++r) B[r + old->perm[i] * rows] = old->R[r + i * m];
  for (int64_t g = 0; g < p->n_groups; ++g) {
    const int64_t o = p->offsets[g], s = p->offsets[g + 1] - o;
    double *blk = malloc(sizeof(double) * (size_t)(s * m));
    for (int64_t c = 0; c < m; ++c)
      for (int64_t a = 0; a < s; ++a) blk[a + c * s] = p->K_fu[(o + a) + c * n];
    ldlt_sqrt_solve(p->a_ldlt[g], s, s, p->a_tr[g], blk, m, s);
    for (int64_t c = 0; c < m; ++c)
      for (int64_t a = 0; a < s; ++a) B[(m + o + a) + c * rows] = blk[a + c * s];
    free(blk);
  }
  orc_qr *q = colpiv_qr(B, rows, m);
  double *y_aug = calloc((size_t)rows, sizeof(double));
  /* y_aug.top = R_old (P_old^T v_old) */
  double *pv = malloc(sizeof(double) * (size_t)m);
  for (int64_t i = 0; i < m; ++i) pv[i] = old->information[old->perm[i]];
  for (int64_t r = 0; r < m; ++r) {
    double s = 0.;
    for (int64_t i = r; i < m; ++i) s += old->R[r + i * m] * pv[i];
    y_aug[r] = s;
  }
  sparse_a_apply(p, p->y, y_aug + m, 0);
  orc_sparse_fit *f = calloc(1, sizeof(orc_sparse_fit));
  f->m = m;
  f->information = malloc(sizeof(double) * (size_t)m);
  qr_solve(q, y_aug, f->information);
  f->R = malloc(sizeof(double) * (size_t)(m * m));
  for (int64_t j = 0; j < m; ++j)
    for (int64_t i = 0; i < m; ++i) f->R[i + j * m] = (i <= j) ? q->qr[i + j * rows] : 0.;
  if (q->rank < m) /* "Inflate the diagonal of R in an attempt to avoid singularity" (:361-365) */
    for (int64_t i = 0; i < m; ++i) f->R[i + i * m] += 1.e-10;
  f->perm = malloc(sizeof(int64_t) * (size_t)m);
  memcpy(f->perm, q->perm, sizeof(int64_t) * (size_t)m);
  f->numerical_rank = q->rank;
  f->nll = NAN;
  f->kuu_ldlt = malloc(sizeof(double) * (size_t)(m * m));
  memcpy(f->kuu_ldlt, old->kuu_ldlt, sizeof(double) * (size_t)(m * m));
  f->kuu_tr = malloc(sizeof(int64_t) * (size_t)m);
  memcpy(f->kuu_tr, old->kuu_tr, sizeof(int64_t) * (size_t)m);
  int64_t *all = malloc(sizeof(int64_t) * (size_t)(m + 1));
  for (int64_t i = 0; i < m; ++i) all[i] = i;
  f->u = subset_features(&old->u, all, m, old->u.is_measurement, &f->coords_copy, &f->eq_copy, &f->scales_copy);
  free(all); free(pv); free(y_aug); free(B);
  qr_free(q);
  sparse_parts_free(p);
  return f;
}
