// common.h — internal types shared by the HIP translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <string>
#include <utility>
#include <vector>

#include "../../include/albatross_amd.h"
#include "cov_eval.h"

// Wave priority of the panel-chain kernels (s_setprio); the bulk update stays at 0 (the four combinations were measured:
// DESIGN.md section 8, profiles/r03/sweep_prio.txt)
#define AGP_CHAIN_PRIO 3

namespace agp {

// Blocking factors of the LL^T factorisation:
//   NB    diagonal-block / panel width handled by one potrf + one trsm launch
//   NBO   outer block: trailing updates accumulate NBO columns of panel (K=NBO)
//   MB    micro-block inside a diagonal block (one f64 MFMA tile)
constexpr int NB = 128;
constexpr int NBO = 512;
constexpr int MB = 16;
constexpr int NMB = NB / MB;  // 8 micro blocks per diagonal block

struct FeatView {
  const double *coords;  // n x dim row-major (device)
  const long long *ids;  // n or nullptr
  const double *scales;  // n x nsc column-major or nullptr
  long long n;
  int dim;
  int nsc;
  int meas;
  long long sstride = 0;  // distance between scale columns (0: n) - lets a view cover a sub-range of a larger vector
};

__host__ __device__ inline long long scale_stride(const FeatView &f) { return f.sstride ? f.sstride : f.n; }

struct DeviceFeatures {
  FeatView v{};
  void *owned[3] = {nullptr, nullptr, nullptr};
  DeviceFeatures() = default;
  DeviceFeatures(const DeviceFeatures &) = delete;
  DeviceFeatures &operator=(const DeviceFeatures &) = delete;
  ~DeviceFeatures() { release(); }  // error paths return early: the owned device copies go with the object
  void release();
};

}  // namespace agp

struct agp_kernel {
  agp::DevProgram prog;
};

struct agp_context {
  int device = 0;
  hipStream_t stream = nullptr;   // main chain
  hipStream_t stream2 = nullptr;  // look-ahead / side chain
  hipStream_t stream3 = nullptr;  // host-side control-plane collectives of the RCCL transport (shard_hip.hip)
  // bulk stream restricted to a CU mask (hipExtStreamCreateWithCUMask): in the chain-bound end phase of the
  // factorisation the bulk updates run here and leave a few CUs per XCD to the panel chain (chol.hip: factor_lower)
  hipStream_t stream_masked = nullptr;
  hipEvent_t ev_c = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  std::vector<hipEvent_t> ev_pool;
  std::string last_error;
  int *d_flags = nullptr;  // [0] nan flag, [1] first bad pivot + 1
  int *h_flags = nullptr;  // pinned mirror
  double *d_scalars = nullptr;  // [0] sum log L_ii, [1] z^T z
  double *h_scalars = nullptr;  // pinned mirror
  void *h_status_dev = nullptr;  // h_flags (the whole status block) as the device addresses it, or nullptr
  bool profiling = false;
  double stage_ms[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  // reusable factor workspace (agp_nll re-uses it between tuner steps)
  double *ws_A = nullptr;
  size_t ws_A_bytes = 0;
  double *ws_aux = nullptr;
  size_t ws_aux_bytes = 0;
  double *ws_refine = nullptr;  // scratch of the mixed fit's refinement (api.hip: refine_information), kept between fits
  size_t ws_refine_bytes = 0;
  // one cached factor allocation (agp_fit_destroy parks its N x N buffer here;
  // the next agp_fit_create of the same size takes it instead of hipMalloc)
  double *pool_A = nullptr;
  size_t pool_A_bytes = 0;
  // ... and one cached exact-covariance buffer of the mixed-precision fit (its CG refinement multiplies with K itself)
  double *pool_K = nullptr;
  size_t pool_K_bytes = 0;
  // mixed-precision fits: two alternating fp32 copies of the current panel (rows x 512 each)
  float *p32 = nullptr;
  size_t p32_bytes = 0;
  double *f16_scales = nullptr;  // fp16 x 2 products (gemm_f16x2.hip): row scales r and 1 / r, f16_scales_n doubles each
  long long f16_scales_n = 0;
  // ... and the fp32 copy of the factor that preconditions their refinement
  float *pool_L32 = nullptr;
  size_t pool_L32_bytes = 0;
  // ... and one cached block of a fit's small buffers (agp_fit::aux_base)
  double *pool_aux = nullptr;
  size_t pool_aux_bytes = 0;
  // bulk-update kernel of the next factorisation: -1 = default (fp64 MFMA), 3 = fp32 products, 4 = bf16 x 3 products (fp32 accuracy on the BF16 pipe), 5 = fp16 x 2 products of scaled rows
  // (agp_fit_create_mixed sets and resets it around its factor_lower call)
  int update_variant = -1;
  long long nbo_override = 0;  // outer block width of the next factorisation (0 = default schedule)
  long long nbo_wide = 0;      // mixed precision (bf16 x 3): outer block width while many rows remain (0 / 512 = default schedule)
  // scratch of the sharded fit (agp_sharded_fit_destroy parks it here, like pool_A)
  double *pool_shard = nullptr;
  size_t pool_shard_bytes = 0;
  // the n-sized buffers of a sparse GP fit (K_uf / W, P / Q1^T, split-K slabs): grow-only, kept between fits - a tuner
  // re-fits the same shapes, and hipMalloc / hipFree of several GB per call costs more than some of the stages
  double *pool_sparse = nullptr;
  size_t pool_sparse_bytes = 0;
  // ... and one cached slab of agp_fit_create_batch (released by the last fit of a batch)
  double *pool_batch = nullptr;
  size_t pool_batch_bytes = 0;
  // fused panel kernel (chol.hip: panel_fused_kernel): the slots through which a diagonal block's z_b reaches the
  // workgroups solving the rows below in the same launch (one per matrix row, sentinel-filled per factorisation), and
  // which tile-image buffer has been sentinel-filled for the factorisation in progress
  double *d_zpub = nullptr;
  long long zpub_cap = 0, zpub_ready_n = 0;
  // ... and the slots through which an UPDATED diagonal block reaches the workgroup that factors it (one tile image
  // per diagonal block, like invd; only allocated when the update-ahead panel kernel is in use)
  double *d_dpub = nullptr;
  long long dpub_cap = 0;  // diagonal blocks
  int cus = 256;  // CUs of the device (hipDeviceAttributeMultiprocessorCount)
  long long step_slots = 0;  // workgroups of the panel step kernel the device holds at once (chol.hip: step_slots)
  // The switches a caller can set through the environment, read ONCE at agp_context_create (api.hip: read_tuning; the
  // table is in include/albatross_amd.h).  Everything else about the schedule is a constant of the library.
  struct Tuning {
    bool panel_fused = true;       // AGP_PANEL_FUSED=0: POTRF and panel TRSM as two launches
    long long step_below = 4608;   // AGP_STEP_BELOW: remaining rows at or below which every panel is ONE step launch (0: off)
    bool gram_sop = true;          // AGP_GRAM_SOP=0: covariance trees through the stack interpreter only
    long long fp64_nbo = 0;        // AGP_FP64_NBO: outer block width of the fp64 factorisation while > 8192 rows remain (0: 512)
    long long mixed_nbo = 512;     // AGP_MIXED_NBO: outer block width of the bf16 x 3 factorisation while > 8192 rows remain
    bool mixed_f16 = true;         // AGP_MIXED_F16=0: the mixed-precision fit's products from three bf16 planes (round 5) instead of two fp16 planes
    bool mixed_bf16 = true;        // AGP_MIXED_BF16=0: the mixed-precision fit's products on the fp32 MFMA (rounds 1-4) instead of bf16 x 3
    long long backsub_coop_max = 2047;  // AGP_BACKSUB_COOP_MAX: largest n whose fit uses it (measurement switch)
    bool backsub_coop = true;      // AGP_BACKSUB_COOP=0: the fit's back substitution as a launch per block (rounds 1-4) instead of ONE launch
    bool sparse_pivoted = false;   // AGP_SPARSE_PIVOTED=1: the sparse GP's literal (pivoted LDL^T + QR) path always
    long long predict_chunk = 0;   // AGP_PREDICT_CHUNK: test points per slice of the marginal / joint predictions (0: by memory)
    long long shard_block = 0;     // AGP_SHARD_BLOCK: 128 / 256 / 512 rows per row block of the sharded fit (0: 512)
    bool shard_force_comm = false; // AGP_SHARD_FORCE_COMM=1: ONE rank runs the multi-rank schedule through its transport
    bool shard_host_pacing = false;  // AGP_SHARD_HOST_PACING=1: the sharded schedule is paced by the host (round-3 scheme)
    double shard_mask_gflop = 40.;   // AGP_SHARD_MASK_GFLOP: bulk update per step below which a sharded fit is chain-bound
    long long merge_above = 8704;  // AGP_MERGE_ABOVE: trailing rows above which the next-block-column update rides in the bulk launch (0: never)
  } tune;
  unsigned long long *d_rowcnt = nullptr;  // one counter per 64 rows (tail of the d_dpub allocation): hand-over of the step launches' row updates
  // merged bulk updates (chol.hip: factor_lower): one counter per outer step - the tiles of the next block column count
  // themselves, the chain stream's gate kernel waits for all of them; zeroed by panel_fused_plan (headcnt_ready)
  // pinned host memory for the small tables a batched entry point uploads (descriptors of its problems): a copy from
  // pinned memory is really asynchronous, so the call needs no synchronisation between building a table and the launch
  // that reads it (api.hip: host_stage; valid until the call's final synchronisation)
  void *h_stage = nullptr;
  size_t h_stage_bytes = 0;
  static constexpr long long HEADCNT_WORDS = 256;
  unsigned long long *d_headcnt = nullptr;
  bool headcnt_ready = false;
  // Early inversion of the wide diagonal blocks for the backward substitution of a fit (api.hip: backward_solve_vec_any):
  // set by the caller of factor_lower (bs_W = where the inverses go, bs_BW = their width); factor_lower inverts the
  // blocks that are final when it enters its single-stream tail on the (then idle) second stream, records ev_inv and
  // reports how many it did in bs_done.  bs_W == nullptr: not requested.
  double *bs_W = nullptr;
  long long bs_BW = 0, bs_done = 0;
  hipEvent_t ev_inv = nullptr;
  const double *img_ready = nullptr;
  bool prep_external = false;  // the caller of factor_lower has made the fills of panel_fused_plan itself (api.hip: fit_create_impl)
  // sharded fit, device-side pacing (shard_hip.hip: HipShardOps): one flag per schedule event + probe flags in device
  // memory, their sequence numbers (monotonic over the life of the context), and the pacing mode (-1: not decided yet)
  hipStream_t stream_comm = nullptr;  // queue of the collectives (created by the first sharded call)
  unsigned long long *shard_flags = nullptr;
  unsigned long long shard_seq[16] = {};
  unsigned long long shard_probe_seq = 0;
  int shard_host_pacing = -1;
  bool shard_probe_ok = false;
};

// agp_fit_create_batch: the fits of one batch are slices of ONE device allocation (the batched kernels need constant
// strides between the problems); it goes when the last of them is destroyed
struct agp_fit_slab {
  double *base = nullptr;
  size_t bytes = 0;
  int refs = 0;
};

struct agp_fit {
  agp_context *ctx = nullptr;  // owner; a fit must not outlive its context
  agp_fit_slab *slab = nullptr;  // non-null: A / invd / alpha / z are slices of slab->base (winv: not kept)
  size_t A_bytes = 0;
  int device = 0;
  int64_t n = 0;
  int64_t lda = 0;
  double *A = nullptr;      // n x n lower factor, column-major, ld = lda
  double *invd = nullptr;   // tile image of every NB x NB diagonal block (72 KiB each):
                            // negated off-diagonal micro tiles + inverted diagonal micro tiles
  double *winv = nullptr;   // (n/NB) inverted NB x NB diagonal blocks
  double *alpha = nullptr;  // information vector K^-1 y
  double *z = nullptr;      // L^-1 y
  // agp_fit_create makes invd / winv / alpha / z slices of ONE allocation (aux_base, aux_bytes), which agp_fit_destroy
  // parks in the context like the factor's buffer: four hipMalloc + four hipFree per fit less.  nullptr: the four are
  // allocations of their own (every other maker of an agp_fit).
  double *aux_base = nullptr;
  size_t aux_bytes = 0;
  agp::DeviceFeatures train;
  double log_det = 0.;
  int64_t failed_pivot = -1;
  // Fits grown by agp_fit_update: the appended block starts at a multiple of 128, so when the size before the update
  // was not one, the rows in between are PHANTOM rows (identity rows of the factor, zero information, decoupled from
  // everything).  `n` counts them - every kernel works on the padded factor -, `n_real` does not (0: no phantoms) and
  // the C-ABI only ever shows real rows.  phantom: sorted half-open ranges [first, last) of padded row indices.
  int64_t n_real = 0;
  std::vector<std::pair<long long, long long>> phantom;
};

#define AGP_HIP_CHECK(ctx, expr)                                              \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      if (ctx) {                                                              \
        (ctx)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e); \
      }                                                                       \
      return AGP_ERR_HIP;                                                     \
    }                                                                         \
  } while (0)

namespace agp {

// ---- launchers implemented in the .hip files ----
bool launch_gram_blocks(hipStream_t s, const DevProgram *host_program, const FeatView &X, long long rows, long long count,
                        double *out, long long ld, long long stride, const double *diag_add, int *nan_flag);
void launch_gram(hipStream_t s, const DevProgram *P, const FeatView &X, const FeatView &Y,
                 bool symmetric, bool lower_only, double *out, long long ld,
                 const double *diag_add, int *nan_flag, const DevProgram *host_program = nullptr);
// `count` symmetric lower-only Gram matrices in ONE launch when every problem takes the same fast path (gram.hip); false:
// not applicable, nothing launched.  table_dev: gram_batch_table_bytes(count) bytes of device scratch.
size_t gram_batch_table_bytes(long long count);
// host_stage (optional): gram_batch_table_bytes(count) bytes of PINNED host memory that stay untouched until the launch has
// run - the table is built there and uploaded without a synchronisation; nullptr: pageable staging + one stream synchronisation
bool launch_gram_batch(hipStream_t s, long long count, const DevProgram *const *host_programs, const FeatView *Xs, double *const *outs,
                       long long ld, const double *const *diag_adds, int *const *nan_flags, void *table_dev, void *host_stage = nullptr);
void launch_gram_diagonal(hipStream_t s, const DevProgram *P, const FeatView &X, double *out);
// mean_j = sum_i k(x_i, xs_j) alpha_i without materialising the cross Gram
void launch_predict_mean(hipStream_t s, const DevProgram *P, const FeatView &X, const FeatView &XS,
                         const double *alpha, double *mean, const DevProgram *host_program = nullptr);

// LL^T of the n x n lower triangle of A (ld = lda), in place.  y (n) is
// overwritten with z = L^-1 y when non-null.  invd receives the inverted
// diagonal micro blocks.  flags[1] = first non-positive pivot + 1.
// scalars[0] += sum log L_ii.
struct FactorTimers {
  hipEvent_t *ev = nullptr;  // optional pool: one (start, stop) pair per trailing-update launch
  double *flops = nullptr;   // per pair: flop of that launch
  int n_ev = 0;
  int used = 0;
};
void factor_lower(agp_context *ctx, double *A, long long n, long long lda, double *invd, double *y,
                  FactorTimers *timers);
struct PrepArgs;  // pub.h
bool panel_fused_plan(agp_context *ctx, double *invd, long long k_begin, long long k_end, bool want_step, PrepArgs *prep);

// Winv[b] = inv(L_bb) for every NB x NB diagonal block (batched, one launch)
void invert_diag_blocks(hipStream_t s, const double *A, long long n, long long lda, const double *invd,
                        double *Winv);
// x = L^-T z (one right-hand side), z overwritten.
void invert_diag_blocks_forward(hipStream_t s, long long n, const double *invd, double *Wfwd);
void forward_solve_vec(hipStream_t s, const double *A, long long n, long long lda, const double *Wfwd, double *z,
                       double *xstage);
void backward_solve_vec(hipStream_t s, const double *A, long long n, long long lda, const double *Winv,
                        double *z, double *xstage);
// B (n x m, ldb) <- L^-1 B
void forward_solve_mat(hipStream_t s, const double *A, long long n, long long lda, const double *invd,
                       double *B, long long m, long long ldb, bool rhs_lower = false);
// B (n x m, ldb) <- L^-T B
void factor_lower_batched(hipStream_t s, double *A, long long stride_A, long long n, long long lda, double *invd,
                          long long stride_invd, double *y, long long stride_y, long long count, int *flags,
                          double *logsum, long long stride_flags = 0, double *zpub = nullptr, long long stride_zpub = 0);
// may a batch of `count` n x n problems use the fused panel launches of factor_lower_batched (zpub != nullptr)?
bool batched_fused_fits(agp_context *ctx, long long n, long long count);
void factor_lower_batched_lookahead(agp_context *ctx, double *A, long long stride_A, long long n, long long lda, double *invd,
                                    long long stride_invd, double *y, long long stride_y, long long count, int *flags,
                                    double *logsum, long long stride_flags = 0);
// z_b <- L_b^-T z_b for `count` problems, one vector each (solve.hip)
void backward_solve_vec_batched(hipStream_t s, const double *A, long long stride_A, long long n, long long lda,
                                const double *invd, long long stride_invd, double *z, long long stride_z, long long count);
// x = L^-T z in ONE launch (solve.hip: backsub_coop_kernel); done: backsub_done_words(n, count) ZEROED words (PrepArgs::fill),
// or nullptr with x SENTINEL-filled (PrepArgs::sentinel) for at most BACKSUB_DIRECT_BLOCKS 128-row blocks
constexpr long long BACKSUB_DIRECT_BLOCKS = 16;
long long backsub_done_words(long long n, long long count);
void backward_solve_coop(hipStream_t s, const double *A, long long n, long long lda, const double *invd, const double *z,
                         double *x, int *flags, unsigned long long *done, long long count = 1, long long stride_A = 0,
                         long long stride_invd = 0, long long stride_z = 0, long long stride_x = 0, long long stride_flags = 0,
                         const void *status_src = nullptr, void *status_dst = nullptr, int status_words = 0);
void launch_fill_sentinel(hipStream_t s, double *p, long long count);
void forward_solve_mat_batched(hipStream_t s, const double *A, long long stride_A, long long n, long long lda,
                               const double *invd, long long stride_invd, double *B, long long stride_B, long long m,
                               long long ldb, bool rhs_lower, long long count);
void right_solve_lt_batched(hipStream_t s, const double *A, long long stride_A, long long n, long long lda,
                            const double *invd, long long stride_invd, double *X, long long stride_X, long long nrows,
                            long long ldx, long long count);
void launch_gemm_nt_sub_batched(hipStream_t s, double *C, long long ldc, long long batch_C, const double *A,
                                long long lda, bool a_kmajor, long long batch_A, const double *B, long long ldb,
                                bool b_kmajor, long long batch_B, long long M, long long N, long long K, bool tri,
                                long long count);
void right_solve_lt(hipStream_t s, const double *A, long long n, long long lda, const double *invd, double *X,
                    long long nrows, long long ldx);
void forward_solve_mat_lookahead(agp_context *ctx, const double *A, long long n, long long lda, const double *invd,
                                 double *B, long long m, long long ldb, bool rhs_lower = false);
void backward_solve_mat(hipStream_t s, const double *A, long long n, long long lda, const double *invd,
                        double *B, long long m, long long ldb);
// X = L^-1 B OUT OF PLACE for a right-hand side much wider than L (the sparse GP's m x n matrices): see solve.hip
constexpr long long WIDE_BW = 512;
bool forward_solve_wide_ok(long long n, long long ncols);
void invert_wide_blocks(hipStream_t s, const double *A, long long n, long long lda, const double *invd, long long BW, double *W);
void forward_solve_wide(hipStream_t s, const double *A, long long n, long long lda, const double *Winv, const double *B,
                        long long ldb, double *X, long long ldx, long long ncols);

// out[j] = sum_i A[i,j] * B[i,j]   (column-wise dot of two n x m matrices)
void launch_coldot(hipStream_t s, const double *A, long long lda, const double *B, long long ldb,
                   long long n, long long m, double *out, double scale, const double *base);
// C (m x m, ldc) = base - V^T W  with V, W n x m
void launch_gemm_tn_sub(hipStream_t s, const double *V, long long ldv, const double *W, long long ldw,
                        long long n, long long m, double *C, long long ldc);
// y = A^T x for A n x m
void launch_gemv_t(hipStream_t s, const double *A, long long lda, long long n, long long m,
                   const double *x, double *y);
void launch_dot(hipStream_t s, const double *a, const double *b, long long n, double *out);

}  // namespace agp

namespace agp {
// C(M x N) -= A(M x K) * B(N x K)^T  (fp64 MFMA).  a_kmajor / b_kmajor select
// transposed operand storage; tri keeps only tiles on/below C's diagonal.
void launch_gemm_nt_sub(hipStream_t s, double *C, long long ldc, const double *A, long long lda,
                        bool a_kmajor, const double *B, long long ldb, bool b_kmajor, long long M,
                        long long N, long long K, bool tri);
void launch_gemm_nt_ext(hipStream_t s, double *C, long long ldc, const double *Cin, long long ldcin, const double *A, long long lda,
                        const double *B, long long ldb, long long M, long long N, long long K);
void launch_gemm_nt_sub_stair(hipStream_t s, double *C, long long ldc, const double *A, long long lda, const double *B,
                              long long ldb, long long M, long long N, long long K, int world, int rank, long long lb0,
                              long long block, long long c0);
// bulk trailing update of the factorisation: C(M x M, lower tiles) -= P Q^T
// `timing` (optional): an event pair recorded around the trailing_update_kernel launch of this update
// (not around the 64-tile tail launch) and the algorithmic flop of exactly the tiles that launch covers;
// flops == 0 on return means no such launch was made and the events were not recorded.
struct BulkTiming {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  double flops = 0.;
};
void launch_trailing_update_as(int variant, hipStream_t s, double *C, long long ldc, const double *P,
                               const double *Q, long long ldp, long long M, long long K,
                               BulkTiming *timing = nullptr, const float *P32 = nullptr, const float *Q32 = nullptr,
                               long long ld32 = 0);
void launch_trailing_update(hipStream_t s, double *C, long long ldc, const double *P, const double *Q,
                            long long ldp, long long M, long long K, BulkTiming *timing = nullptr);
void read_bulk_probe(unsigned long long *out);   // gemm.hip: cycles and 100 MHz ticks of one tile of a -DAGP_BULK_STAMPS build (zeros otherwise)
void read_bf16_probe(unsigned long long *out);   // gemm_bf16x3.hip: 8 words of a -DAGP_BF16_STAMPS build (zeros otherwise)
void read_potrf_probe(unsigned long long *out);  // chol.hip: 4 x 32 cycle stamps of a -DAGP_POTRF_TIMING build (zeros otherwise)
void launch_head_gate(hipStream_t s, const unsigned long long *done, unsigned long long expect, int *flags);  // chol.hip
// the whole trailing matrix of an outer step in ONE launch, the next block column's tiles first and counted (gemm.hip)
void launch_trailing_update_merged(hipStream_t s, double *C, long long ldc, const double *P, long long ldp, long long M, long long K,
                                   int head_cols, unsigned long long *head_done, long long *head_tiles, BulkTiming *timing = nullptr);
void launch_update_f32(hipStream_t s, double *C, long long ldc, const double *P, const double *Q, long long ldp, long long M,
                       long long N, long long K, const float *P32 = nullptr, const float *Q32 = nullptr, long long ld32 = 0);
// The bf16 x 3 path of the mixed-precision factorisation (gemm_bf16x3.hip): the panel of one outer step as three bf16
// planes (hi + mid + lo = the value to fp32 accuracy), and C -= P[row_a ..] P[row_b ..]^T from them on the BF16 pipe
size_t bf16x3_bytes(long long rows, long long K);
void launch_convert_panel_bf16x3(hipStream_t s, const double *P, long long ldp, long long rows, long long K, unsigned short *planes);
void set_bf16x3_kernel(int choice, int lds_pad);  // 1: one workgroup per CU (first version), 2: two per CU (AGP_BF16X3_KERNEL)
void launch_update_bf16x3(hipStream_t s, double *C, long long ldc, const unsigned short *planes, long long panel_rows, long long row_a,
                          long long row_b, long long M, long long N, long long K, const int *order = nullptr, long long order_len = 0);
// The fp16 x 2 path (gemm_f16x2.hip): power-of-two row scales from the diagonal (rs, irs = 1 / rs: n doubles each), the
// panel of one outer step as two fp16 planes of the scaled rows, and C -= P[row_a ..] P[row_b ..]^T from them (three
// v_mfma_f32_16x16x32_f16 per block; irs belongs to the panel's row 0)
size_t f16x2_bytes(long long rows, long long K);
void launch_f16x2_row_scales(hipStream_t s, const double *A, long long lda, long long n, double *rs, double *irs);
void launch_convert_panel_f16x2(hipStream_t s, const double *P, long long ldp, long long rows, long long K, const double *rs,
                                unsigned short *planes);
void set_f16x2_kernel(int lds_pad, int terms, int chunk);  // AGP_F16X2_LDS_PAD, AGP_F16X2_TERMS (3, or 4: with h2 h2), AGP_F16X2_CHUNK (32 / 64)
bool f16x2_depth_ok(long long K);  // K a whole number of the kernel's K chunks
void launch_update_f16x2(hipStream_t s, double *C, long long ldc, const unsigned short *planes, long long panel_rows, long long row_a,
                         long long row_b, const double *irs, long long M, long long N, long long K, const int *order = nullptr,
                         long long order_len = 0);
// the XCD-aware order of the `tiles` first lower 128 x 128 tiles of a grid with ntr tile rows (gemm.hip; cached): nullptr = none
const int *bulk_tile_order(int ntr, long long tiles, long long *len);
// P32 (rows x K, ld32) = (float) P: the fp32 copy of one outer step's panel for the fp32-product kernels
void launch_convert_panel_f32(hipStream_t s, const double *P, long long ldp, long long rows, long long K, float *P32, long long ld32);
}  // namespace agp
