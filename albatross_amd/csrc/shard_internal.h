// shard_internal.h — the handles behind agp_comm (shared by shard_sched.hip and shard_hip.hip)
#pragma once
#include <vector>

#include "shard.h"

// Block arithmetic supplied by the caller on host memory (agp_debug_shard_factor_custom, debug_api.hip): lets the schedule
// of shard_sched.hip run on machines without a GPU (tests/test_distributed_cpu.py: numpy + gloo).  TEST-ONLY: the
// entry point lives in libalbatross_amd_debug.so, not in the product library.
typedef struct {
  void *user;
  /* LL^T of the w x w block D (ld) in place, z <- L^-1 z on the w entries at zblk; img: scratch the other callbacks
   * get back (4 * 9216 doubles); returns 0, or 1 + index of the first non-positive pivot */
  int64_t (*factor_diag)(void *user, double *D, int64_t ld, int64_t w, double *img, double *zblk, double *logsum);
  /* X (nrows x w, ld) <- X L^-T with L the w x w lower triangle at Lkk (leading dimension w);
   * yrows[r] -= sum_c X[r][c] z[c] */
  void (*trsm_rows)(void *user, double *X, int64_t ld, int64_t nrows, int64_t w, const double *Lkk,
                    const double *img, const double *z, double *yrows);
  /* C (M x N, ldc) -= P (M x K, ldp) Q (N x K, ldq)^T; tri: only entries on / below the diagonal of C are needed */
  void (*gemm)(void *user, double *C, int64_t ldc, const double *P, int64_t ldp, const double *Q, int64_t ldq,
               int64_t M, int64_t N, int64_t K, int tri);
  void (*copy2d)(void *user, double *dst, int64_t ldd, const double *src, int64_t lds, int64_t rows, int64_t cols);
  /* W (w x w, ld = w) <- inverse of the lower-triangular w x w block at D (ld) */
  void (*invert_diag)(void *user, const double *D, int64_t ld, int64_t w, const double *img, double *W);
  /* out[j] = alpha * sum_i W[i + j * ld] v[i] + beta * base[j]   (i < m, j < n; base may be NULL) */
  void (*colvec_dot)(void *user, const double *W, int64_t ld, int64_t m, int64_t n, const double *v, double alpha,
                     double beta, const double *base, double *out);
  void (*axpby)(void *user, int64_t n, double a, const double *x, double b, const double *y, double *out);
  void (*fill_zero)(void *user, double *p, int64_t count);
} agp_shard_ops_callbacks;

namespace agp {

// a transport that can also serve the host-side control plane (agp_comm_all_reduce_host / agp_comm_barrier)
struct HostReducingComm : ShardComm {
  virtual int all_reduce_host(double *buf, long long count, int op) = 0;
};

// collectives supplied by the caller, on host memory (agp_comm_create_callbacks)
struct CallbackComm : HostReducingComm {
  agp_comm_callbacks cb;
  std::vector<double> stage;
  explicit CallbackComm(const agp_comm_callbacks &c) : cb(c) {}
  int broadcast(ShardOps &ops, int q, double *buf, long long count, int root) override;
  int all_gather(ShardOps &ops, int q, const double *send, double *recv, long long count) override;
  int all_reduce(ShardOps &ops, int q, double *buf, long long count, int op) override;
  int all_reduce_host(double *buf, long long count, int op) override;

 private:
  int staged(ShardOps &ops, int q, double *buf, long long count, int kind, int arg, const double *send, long long send_count);
};


struct CallbackShardOps : ShardOps {
  agp_shard_ops_callbacks cb;
  double logsum = 0.;
  long long bad = 0;
  explicit CallbackShardOps(const agp_shard_ops_callbacks &c) : cb(c) {}
  void factor_diag(int, double *D, long long ld, long long w, long long pivot_base, double *img, double *zblk) override {
    double ls = 0.;
    const long long b = cb.factor_diag(cb.user, D, ld, w, img, zblk, &ls);
    logsum += ls;
    if (b > 0 && bad == 0) bad = pivot_base + b;
  }
  void trsm_rows(int, double *X, long long ld, long long nrows, long long w, const double *Lkk, const double *img,
                 const double *z, double *yrows) override {
    cb.trsm_rows(cb.user, X, ld, nrows, w, Lkk, img, z, yrows);
  }
  void gemm(int, double *C, long long ldc, const double *P, long long ldp, const double *Q, long long ldq, long long M,
            long long N, long long K, bool tri, int) override {
    cb.gemm(cb.user, C, ldc, P, ldp, Q, ldq, M, N, K, tri ? 1 : 0);
  }
  void copy2d(int, double *dst, long long ldd, const double *src, long long lds, long long rows, long long cols) override {
    cb.copy2d(cb.user, dst, ldd, src, lds, rows, cols);
  }
  void invert_diag(int, const double *D, long long ld, long long w, const double *img, double *W) override {
    cb.invert_diag(cb.user, D, ld, w, img, W);
  }
  void colvec_dot(int, const double *W, long long ld, long long m, long long n, const double *v, double alpha, double beta,
                  const double *base, double *out) override {
    cb.colvec_dot(cb.user, W, ld, m, n, v, alpha, beta, base, out);
  }
  void axpby(int, long long n, double a, const double *x, double b, const double *y, double *out) override {
    cb.axpby(cb.user, n, a, x, b, y, out);
  }
  void fill_zero(int, double *p, long long count) override { cb.fill_zero(cb.user, p, count); }
  void status(double out[2]) override { out[0] = logsum; out[1] = (double)bad; }
};

}  // namespace agp

struct agp_comm {
  agp::HostReducingComm *impl = nullptr;
};
