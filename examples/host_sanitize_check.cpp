// host_sanitize_check.cpp — the host-side C++ of the project under AddressSanitizer + UndefinedBehaviorSanitizer, on
// a machine WITHOUT a GPU (examples/Makefile: `make asan`).  Two parts:
//
//  1. the sharded-fit schedule (albatross_amd/csrc/shard_sched.hip is plain C++: compiled INTO this binary with the
//     sanitizers on) driven through agp_debug_shard_factor_custom (csrc/shard_custom.hip) with naive block operations, one rank and - through
//     in-process "collectives" - the forced multi-rank path; checked against a naive dense solve;
//  2. the header-only host layer (include/albatross_amd/albatross.hpp): covariance-function programs, parameter
//     handling, feature flattening (Measurement<>, scale columns), grouping - everything that runs before the
//     first device call.
//
// Prints "host_sanitize_check ok" and exits 0; any sanitizer report aborts with a non-zero status.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "albatross_amd/albatross.hpp"
#include "shard_internal.h"  // agp_shard_ops_callbacks (csrc/, test-only)

extern "C" {
int64_t agp_debug_shard_work_doubles(int64_t n, int64_t block, int nranks, int rank);
int agp_debug_shard_factor_custom(const agp_shard_ops_callbacks *ops, agp_comm *comm, int64_t n, int64_t block, double *A, int64_t ld,
                                  double *y, double *work, double *information, double *log_det, int64_t *bad_pivot);
}

using namespace albatross;

namespace {
double *at(double *p, std::int64_t ld, std::int64_t r, std::int64_t c) { return p + r + c * ld; }

std::int64_t cb_factor_diag(void *, double *D, std::int64_t ld, std::int64_t w, double *, double *z, double *logsum) {
  std::int64_t bad = 0;
  double ls = 0.;
  for (std::int64_t j = 0; j < w; ++j) {
    double d = *at(D, ld, j, j);
    for (std::int64_t k = 0; k < j; ++k) d -= *at(D, ld, j, k) * *at(D, ld, j, k);
    if (!(d > 0.) && bad == 0) bad = j + 1;
    const double l = std::sqrt(d);
    *at(D, ld, j, j) = l;
    ls += std::log(l);
    for (std::int64_t i = j + 1; i < w; ++i) {
      double v = *at(D, ld, i, j);
      for (std::int64_t k = 0; k < j; ++k) v -= *at(D, ld, i, k) * *at(D, ld, j, k);
      *at(D, ld, i, j) = v / l;
    }
  }
  for (std::int64_t i = 0; i < w; ++i) {
    double v = z[i];
    for (std::int64_t k = 0; k < i; ++k) v -= *at(D, ld, i, k) * z[k];
    z[i] = v / *at(D, ld, i, i);
  }
  *logsum = ls;
  return bad;
}
void cb_trsm_rows(void *, double *X, std::int64_t ld, std::int64_t nrows, std::int64_t w, const double *L, const double *, const double *z,
                  double *y) {
  for (std::int64_t r = 0; r < nrows; ++r) {
    for (std::int64_t c = 0; c < w; ++c) {
      double v = *at(X, ld, r, c);
      for (std::int64_t k = 0; k < c; ++k) v -= *at(X, ld, r, k) * L[c + k * w];
      *at(X, ld, r, c) = v / L[c + c * w];
    }
    for (std::int64_t c = 0; c < w; ++c) y[r] -= *at(X, ld, r, c) * z[c];
  }
}
void cb_gemm(void *, double *C, std::int64_t ldc, const double *P, std::int64_t ldp, const double *Q, std::int64_t ldq, std::int64_t M,
             std::int64_t N, std::int64_t K, int tri) {
  for (std::int64_t j = 0; j < N; ++j)
    for (std::int64_t i = tri ? j : 0; i < M; ++i) {
      double s = 0.;
      for (std::int64_t k = 0; k < K; ++k) s += P[i + k * ldp] * Q[j + k * ldq];
      C[i + j * ldc] -= s;
    }
}
void cb_copy2d(void *, double *dst, std::int64_t ldd, const double *src, std::int64_t lds, std::int64_t rows, std::int64_t cols) {
  for (std::int64_t c = 0; c < cols; ++c)
    for (std::int64_t r = 0; r < rows; ++r) dst[r + c * ldd] = src[r + c * lds];
}
void cb_invert_diag(void *, const double *D, std::int64_t ld, std::int64_t w, const double *, double *W) {
  for (std::int64_t c = 0; c < w; ++c)
    for (std::int64_t r = 0; r < w; ++r) {
      double v = r == c ? 1. : 0.;
      for (std::int64_t k = 0; k < r; ++k) v -= D[r + k * ld] * W[k + c * w];
      W[r + c * w] = v / D[r + r * ld];
    }
}
void cb_colvec_dot(void *, const double *W, std::int64_t ld, std::int64_t m, std::int64_t n, const double *v, double alpha, double beta,
                   const double *base, double *out) {
  for (std::int64_t c = 0; c < n; ++c) {
    double s = 0.;
    for (std::int64_t r = 0; r < m; ++r) s += W[r + c * ld] * v[r];
    out[c] = alpha * s + (base ? beta * base[c] : 0.);
  }
}
void cb_axpby(void *, std::int64_t n, double a, const double *x, double b, const double *y, double *out) {
  for (std::int64_t i = 0; i < n; ++i) out[i] = a * x[i] + b * y[i];
}
void cb_fill_zero(void *, double *p, std::int64_t count) {
  for (std::int64_t i = 0; i < count; ++i) p[i] = 0.;
}
int cb_broadcast(void *, double *, std::int64_t, int root) { return root == 0 ? 0 : 1; }
int cb_all_gather(void *, const double *send, double *recv, std::int64_t count) {
  for (std::int64_t i = 0; i < count; ++i) recv[i] = send[i];
  return 0;
}
int cb_all_reduce(void *, double *, std::int64_t, int) { return 0; }

void require(bool ok, const char *what) {
  if (!ok) {
    std::fprintf(stderr, "host_sanitize_check: FAILED: %s\n", what);
    std::exit(1);
  }
}

void schedule_under_sanitizers(bool force_comm) {
  const std::int64_t n = 333, block = 128;
  std::vector<double> K(n * n), y(n), ref(n);
  unsigned long long s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) * (1.0 / 9007199254740992.0); };
  std::vector<double> pts(n);
  for (auto &p : pts) p = 10. * rnd();
  for (std::int64_t i = 0; i < n; ++i) {
    y[i] = std::sin(pts[i]);
    for (std::int64_t j = 0; j < n; ++j) K[i + j * n] = std::exp(-(pts[i] - pts[j]) * (pts[i] - pts[j])) + (i == j ? 0.01 : 0.);
  }
  // reference: naive dense LL^T solve
  {
    std::vector<double> L(K), z(y);
    double ls = 0.;
    require(cb_factor_diag(nullptr, L.data(), n, n, nullptr, z.data(), &ls) == 0, "reference factorisation");
    for (std::int64_t i = n - 1; i >= 0; --i) {
      double v = z[i];
      for (std::int64_t k = i + 1; k < n; ++k) v -= L[k + i * n] * ref[k];
      ref[i] = v / L[i + i * n];
    }
  }
  if (force_comm) setenv("AGP_SHARD_FORCE_COMM", "1", 1);
  else unsetenv("AGP_SHARD_FORCE_COMM");
  agp_comm *comm = nullptr;
  agp_comm_callbacks ccb{nullptr, cb_broadcast, cb_all_gather, cb_all_reduce};
  if (force_comm) require(agp_comm_create_callbacks(1, 0, &ccb, &comm) == AGP_OK, "agp_comm_create_callbacks");
  const std::int64_t n_loc = agp_shard_local_rows(n, block, 1, 0);
  require(n_loc == n, "one rank owns every row");
  const std::int64_t ld = n_loc + 1;
  std::vector<double> A((size_t)ld * n, std::nan("")), yl(y), work((size_t)agp_debug_shard_work_doubles(n, block, 1, 0), std::nan("")), info(n);
  for (std::int64_t l = 0; l < n_loc; ++l) {
    const std::int64_t g = agp_shard_global_row(n, block, 1, 0, l);
    const std::int64_t end = std::min<std::int64_t>(n, (g / block + 1) * block);
    for (std::int64_t c = 0; c < end; ++c) A[l + c * ld] = K[g + c * n];
  }
  agp_shard_ops_callbacks ops{nullptr, cb_factor_diag, cb_trsm_rows, cb_gemm, cb_copy2d, cb_invert_diag, cb_colvec_dot, cb_axpby, cb_fill_zero};
  double logdet = 0.;
  std::int64_t bad = -1;
  const int st = agp_debug_shard_factor_custom(&ops, comm, n, block, A.data(), ld, yl.data(), work.data(), info.data(), &logdet, &bad);
  require(st == AGP_OK && bad == -1, "agp_debug_shard_factor_custom");
  double worst = 0., scale = 0.;
  for (std::int64_t i = 0; i < n; ++i) { worst = std::fmax(worst, std::fabs(info[i] - ref[i])); scale = std::fmax(scale, std::fabs(ref[i])); }
  require(worst <= 1e-9 * scale, "sharded schedule == dense solve");
  agp_comm_destroy(comm);
}

void host_layer_under_sanitizers() {
  auto cov = SquaredExponential<EuclideanDistance>(2.0, 1.5) * Matern52<EuclideanDistance>(3.0, 0.7) + measurement_only(IndependentNoise<double>(0.1)) +
             Constant(2.0);
  const auto prog = cov.program();
  require(!prog.empty() && prog.size() <= AGP_MAX_KERNEL_NODES, "program size");
  int depth = 0;
  for (const auto &nd : prog) {  // a well-formed postfix program ends with exactly one value on the stack
    if (nd.op == AGP_OP_SUM || nd.op == AGP_OP_PRODUCT) --depth;
    else if (nd.op != AGP_OP_MEASUREMENT_ONLY && nd.op != AGP_OP_TYPE_PAIR) ++depth;
    require(depth >= 1, "stack underflow");
  }
  require(depth == 1, "one value left");
  agp_kernel *k = nullptr;
  require(agp_kernel_create(prog.data(), (int)prog.size(), &k) == AGP_OK, "agp_kernel_create");
  agp_kernel_destroy(k);
  agp_kernel_node bad_prog[2] = {prog[0], prog[0]};  // two leaves, no operator
  require(agp_kernel_create(bad_prog, 2, &k) == AGP_ERR_INVALID_ARGUMENT, "malformed program rejected");
  auto params = cov.get_params();
  require(!params.empty(), "parameters");
  for (const auto &kv : params) cov.set_param(kv.first, kv.second * 1.5);
  for (const auto &kv : cov.get_params()) require(std::fabs(kv.second - 1.5 * params.at(kv.first)) < 1e-15, "set_param round trip");
  std::vector<double> xs = {0.5, 1.5, 2.5, 1.5};
  auto flat = detail::flatten(cov, xs);
  require(flat.view.n == 4 && flat.view.dim == 1 && flat.view.is_measurement == 0 && flat.coords[2] == 2.5, "flatten");
  auto flat_m = detail::flatten(cov, as_measurements(xs));
  require(flat_m.view.is_measurement == 1 && flat_m.coords.size() == 4, "flatten Measurement<>");
  auto groups = group_indexer(xs, [](const double &x) { return (int)std::floor(x); });
  require(groups.size() == 3 && groups.at(1).size() == 2, "group_indexer");
  auto loo = group_indexer(xs, LeaveOneOutGrouper{});
  require(loo.size() == 4, "leave-one-out grouper");
  RegressionDataset<double> ds(xs, MarginalDistribution(Vector{1., 2., 3., 4.}, Vector{0.1, 0.1, 0.1, 0.1}));
  require(ds.size() == 4, "dataset");
}
}  // namespace

int main() {
  schedule_under_sanitizers(false);
  schedule_under_sanitizers(true);
  host_layer_under_sanitizers();
  std::printf("host_sanitize_check ok\n");
  return 0;
}
