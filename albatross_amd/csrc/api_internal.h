// api_internal.h — host-side helpers shared by the translation units that implement the C-ABI
// (api.hip: context / Gram / dense fit / predict; cv_api.hip: leave-one-group-out;
// sparse_api.hip: sparse GP).  Not part of the public interface.
#pragma once
#include <atomic>
#include <string>
#include <vector>

#include "common.h"

namespace agp {
// Device allocations of the entry points that build several medium-sized objects per call (sparse GP fits, dense factors,
// cross validation): a hipMalloc / hipFree of tens of MB is 1-6 ms on this runtime, a sparse fit did ~40 of them (20 ms in
// its first stage alone).  dev_free parks the block (after the device-wide synchronisation hipFree implies) and
// dev_malloc hands out a parked block of exactly the requested size; at most DEV_CACHE_BYTES are kept, the oldest
// blocks go first; agp_context_destroy empties the cache.  Pointers that did not come from dev_malloc are hipFree'd.
// A block that leaves through any other door (a context pool that is closed, a hand-over between owners) goes through
// dev_release, so that the table of live blocks never holds an address the runtime may hand out again.
hipError_t dev_malloc_bytes(void **p, size_t bytes);
template <class T>
inline hipError_t dev_malloc(T **p, size_t bytes) { return dev_malloc_bytes(reinterpret_cast<void **>(p), bytes); }
hipError_t dev_free(void *p);
// plain hipFree of a block that may have come from dev_malloc (forgets it first; never parks)
hipError_t dev_release(void *p);
void dev_cache_trim();
void launch_symmetrize(hipStream_t s, double *A, long long ld, long long n);
void launch_transpose_blocks(hipStream_t s, const double *src, double *dst, long long m, long long count);  // reduce.hip
void launch_copy_lower(hipStream_t s, const double *src, long long ld, long long n, double *dst, int *nan_flag);  // reduce.hip
void launch_zero_upper(hipStream_t s, double *A, long long ld, long long n);
void launch_set_identity(hipStream_t s, double *B, long long ld, long long n);
void launch_nan_scan_lower(hipStream_t s, const double *A, long long ld, long long n, int *flag);
void launch_upper_to_lower(hipStream_t s, const double *src, long long ld_src, double *dst, long long ld_dst,
                           long long n);
void launch_gather_cols(hipStream_t s, const double *R, long long ldr, const long long *idx, long long m,
                        long long row0, long long n, double *G, long long ldg);
void launch_gather_vec(hipStream_t s, const double *src, const long long *idx, long long m, const double *sub,
                       double *out);
void launch_negate(hipStream_t s, double *A, long long ld, long long m, double *diag_out);
void launch_matvec(hipStream_t s, const double *W, long long ld, long long m, long long n, const double *x,
                   double *partial, double alpha, double beta, const double *base, double *out);
void launch_contract_combinations(hipStream_t s, const double *K, long long ldk, const long long *xoff, const double *xc, long long na,
                                  const long long *yoff, const double *yc, long long nb, bool symmetric, double *out, long long ldo);
void launch_tall_matvec(hipStream_t s, const double *W, long long ld, long long rows, long long ncols, const double *x, double alpha,
                        double beta, const double *base, double *out);
void launch_tall_matvec_f32(hipStream_t s, const float *W, long long ld, long long rows, long long ncols, const double *x, double alpha,
                            double beta, const double *base, double *out);
void launch_colvec_dot_f32(hipStream_t s, const float *W, long long ld, long long m, long long n, const double *v,
                           double alpha, double beta, const double *base, double *out);
void launch_convert_lower_f32(hipStream_t s, const double *L, long long ld, long long n, float *L32);
void launch_colvec_dot(hipStream_t s, const double *W, long long ld, long long m, long long n, const double *v,
                       double alpha, double beta, const double *base, double *out);
void launch_colvec_dot_strided(hipStream_t s, const double *W, long long ld, long long stride_W, long long m, long long n,
                               const double *v, long long stride_v, double alpha, double beta, const double *base, double *out,
                               long long count);
// out = alpha K p + beta base for a symmetric K given by its LOWER triangle only (reduce.hip); ws: symv_ws_elems(n) doubles
size_t symv_ws_elems(long long n);
void launch_symv_lower(hipStream_t s, const double *K, long long ld, long long n, const double *p, double alpha, double beta,
                       const double *base, double *out, double *ws);
void launch_axpby(hipStream_t s, long long n, double a, const double *x, double b, const double *y, double *out);
void launch_loo(hipStream_t s, const double *kinv_diag, const double *y, const double *information, long long n,
                double *mean, double *variance);
void launch_colvec_dot_batched(hipStream_t s, const double *Q, long long ld, long long stride_Q, long long m,
                               const double *z, long long stride_z, long long count, double *out);
void launch_set_identity_batched(hipStream_t s, double *B, long long ld, long long stride, long long m, long long count);
void launch_pad_columns(hipStream_t s, const double *src, long long ld_src, const long long *off, long long smax,
                        long long n_groups, long long rows, double *dst, long long ld_dst, int dir);
void launch_pad_identity(hipStream_t s, double *A, long long ld, long long stride, const long long *off, long long smax,
                         long long n_groups);
void launch_compact_blocks(hipStream_t s, const double *slabs, long long ld, long long stride, const long long *off,
                           const long long *boff, long long smax, long long n_groups, double *out);
long long round_up(long long x, long long m);
long long factor_ld(long long n);
// fits grown by agp_fit_update carry phantom rows (common.h: agp_fit::phantom); update_api.hip
long long fit_real_rows(const agp_fit *f);
int fit_compact_vector(agp_context *ctx, const agp_fit *f, const double *padded_dev, double *real_out, int location);
int fit_expand_matrix(agp_context *ctx, const agp_fit *f, const double *real_in, long long ldr, long long nrhs, double *padded_dev,
                      long long ldp, int location);
int fit_compact_matrix(agp_context *ctx, const agp_fit *f, const double *padded_dev, long long ldp, long long nrhs, double *real_out,
                       long long ldr, int location);
void fit_zero_phantom_rows(hipStream_t s, const agp_fit *f, double *V, long long ldv, long long cols);
// pivoted L D L^T (ldlt.hip)
void ldlt_factor(hipStream_t s, double *A, long long lda, long long n, const long long *tr_host, double *temp, int *info,
                 double *scal);
void ldlt_factor_blocked(hipStream_t s, double *Ap, long long lda, long long n, double *T, double *dotacc, int *info);
void ldlt_permute_sym(hipStream_t s, const double *S, long long lds, const long long *q_dev, long long n, double *Ap,
                      long long lda);
void ldlt_solve(hipStream_t s, const double *A, long long lda, long long n, const long long *q_dev, double *W,
                double *R, long long ldw, long long nrhs);
void ldlt_sqrt_solve(hipStream_t s, const double *A, long long lda, long long n, const long long *q_dev, double *W,
                     const double *R, long long ldw, long long nrhs);
// column-pivoted Householder QR and the substitutions against R (qr.hip)
void colpiv_qr(hipStream_t s, double *A, long long lda, long long rows, long long cols, long long extra, double *tau,
               long long *perm, double *norms, double *state);
void qr_extract_r(hipStream_t s, const double *A, long long lda, long long m, double *R, long long ldr, double inflate);
void qr_root(hipStream_t s, const double *R, long long ldr, const long long *perm, long long m, double *T, long long ldt);
void qr_sqrt_solve(hipStream_t s, const double *R, long long ldr, const long long *perm, long long m, const double *X,
                   long long ldx, double *W, long long ldw, long long nrhs);
void qr_back_solve(hipStream_t s, const double *R, long long ldr, const long long *perm, long long m, long long np, double *c,
                   double *out);
}  // namespace agp

struct agp_ldlt {
  agp_context *ctx = nullptr;
  long long n = 0, lda = 0;
  double *A = nullptr;          // matrixLDLT: L strictly below the diagonal (unit diagonal implied), D on it
  long long *q_dev = nullptr;   // the permutation the transpositions compose to: (P b)[i] = b[q[i]]
  std::vector<long long> tr;
  std::vector<double> d;        // vectorD (host copy)
  int success = 1;              // Eigen's info() == Success
};

struct ProgSlot {
  unsigned long long uid = 0;
  agp::DevProgram *dev = nullptr;
};

struct agp_context_ext {
  ProgSlot slots[8];
  int next = 0;
};

extern std::atomic<unsigned long long> g_kernel_uid;

struct agp_kernel_full : agp_kernel {
  unsigned long long uid;
};


struct agp_context_impl : agp_context {
  agp_context_ext ext;
  std::vector<hipEvent_t> gemm_events;
  std::vector<double> gemm_flops;
  hipEvent_t stage_ev[6] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  double *partial_ws = nullptr;
  size_t partial_bytes = 0;
  double gemm_ms_sum = 0., gemm_flop_sum = 0.;
  int gemm_launches = 0;
  // helper contexts (own streams / workspaces) for host threads that work through independent
  // small problems concurrently (the blocks of a sparse GP); created on first use
  std::vector<agp_context *> helpers;
};

// device copy of a kernel program (small LRU ring per context)
int device_program(agp_context *ctx, const agp_kernel *k, const agp::DevProgram **out);
int validate_features(const agp_features *f);
// device view of a feature vector (uploads host data; `copy` forces an owned device copy)
int to_device(agp_context *ctx, const agp_features *f, bool copy, agp::DeviceFeatures *out);
int ensure_ws(agp_context *ctx, double **ws, size_t *have, size_t need);
int vector_to_device(agp_context *ctx, const double *src, long long n, int location, double *dst);
int copy_out(agp_context *ctx, const double *dev, long long count, double *dst, int location);
int copy_out_2d(agp_context *ctx, const double *dev, long long ld_dev, long long rows, long long cols, double *dst,
                long long ld_dst, int location);
int status_from_flags(const agp_context *ctx);
// x = L^-T z for ONE vector (api.hip): z is overwritten with x; ws: backsolve_ws_elems(n) doubles of scratch
extern "C" {  // (defined inside api.hip's extern "C" block)
size_t backsolve_ws_elems(long long n);
void backward_solve_vec_any(hipStream_t s, const double *A, long long n, long long lda, const double *invd, double *z,
                            double *ws, long long first_done = 0, hipEvent_t ev_done = nullptr);
}
