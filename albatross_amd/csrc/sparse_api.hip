// sparse_api.hip — sparse Gaussian process (FITC / PITC) entry points of the C-ABI (include/albatross_amd.h).
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <thread>

#include "api_internal.h"

using namespace agp;

// ---- sparse Gaussian process (FITC / PITC) -----------------------------------------
// SparseGaussianProcessRegression (include/albatross/src/models/sparse_gp.hpp).  The reference
// stores Sigma = (K_uu + K_uf A^-1 K_fu)^-1 through the pivoted Householder QR of
// B = [A^-1/2 K_fu; K_uu^T/2] (:343-352); only R^T R = B^T B enters any result.  The device path
// works on B^T = [T | W] (m x (m + n), column-major: every product is a panel-major MFMA GEMM),
// T = L_u the root of K_uu, W = K_uf A^-T/2, forms M = B^T B = T T^T + W W^T (MFMA SYRK) and factors
// it with the LL^T kernels (L1).  Forming B^T B squares the condition number, which the reference's
// QR avoids, so the factor is repaired the CholeskyQR2 way: Q1^T = L1^-1 B^T is formed explicitly
// (one more triangular solve over all columns), Q1^T Q1 = I + O(eps cond) is factored again (L2),
// and B^T B = (L1 L2)(L1 L2)^T holds to working accuracy: log|R|, R^-T x and the information
// vector (plus two refinement steps against B itself) then agree with the QR-based reference
// algorithm to ~1e-8 even with cond(K_uu) ~ 1e7.  K_uu and every block of A use LL^T as well.
// L_acc = L1 L2 (the transpose of the reference's R P^T, up to an orthogonal factor) is kept for
// FitModel::update.
struct agp_sparse_fit {
  agp_context *ctx = nullptr;
  long long m = 0;
  std::shared_ptr<DeviceFeatures> u;   // train_features = inducing points (shared with updated fits)
  std::shared_ptr<agp_fit> kuu;        // train_covariance = factor of K_uu + inducing_nugget I
  agp_fit *sigma = nullptr;            // L1
  agp_fit *sigma2 = nullptr;           // L2
  double *Lacc = nullptr;              // L1 L2, m x ldm, zero above the diagonal
  double *v = nullptr;                 // information (m)
  double nll = 0.;
  // "pivoted form" of a fit made by agp_sparse_fit_from_prediction (rebase_inducing_points) or by an update of one:
  // the reference's own representation, for covariances that are singular to working precision.
  std::shared_ptr<agp_ldlt> kz;        // train_covariance as a pivoted L D L^T: K_zz WITHOUT nugget after fit_from_prediction
                                       // (:416-418), K_uu + inducing nugget after a pivoted fit (:676-679); else kuu
  std::shared_ptr<agp_ldlt> kp;        // pivoted L D L^T of K_uu + inducing nugget for P = K_uu^-1/2 K_uf of an update, when
                                       // the LL^T of that matrix (kuu) does not exist
  double *R = nullptr;                 // m x round_up(m, 2), upper triangular: Sigma^-1 = P R^T R P^T; else sigma/sigma2
  long long *perm = nullptr;           // P: perm[i] = original index of the column at position i
  long long rank = -1;                 // numerical_rank of the QR (-1: not a pivoted fit)
  double inducing_nugget = 0.;         // the nugget an update adds to K_uu for P = K_uu^-1/2 K_uf (:674-685)
};

namespace agp {
int comm_all_reduce_device(agp_context *ctx, agp_comm *comm, double *dev, long long count, int op);  // shard_hip.hip
int comm_wait_stream(agp_context *ctx, hipStream_t s);                                               // shard_hip.hip
}

namespace {

struct SparseScratch {
  double *Kuf = nullptr, *Pbuf = nullptr, *M0 = nullptr, *T = nullptr, *vecs = nullptr, *partial = nullptr,
         *Ag = nullptr, *Pimg = nullptr, *Q1T = nullptr, *Winv = nullptr;
  std::vector<agp_fit *> blocks;
  DeviceFeatures dx;
  // Kuf, Pbuf / Q1T (one region: P is dead before Q1^T is formed) and the split-K slabs live in ctx->pool_sparse
  double *slabs = nullptr, *pads = nullptr;
  long long slab_count = 0;
  ~SparseScratch() {
    (void)dev_free(M0); (void)dev_free(T); (void)dev_free(vecs);
    (void)dev_free(partial); (void)dev_free(Ag); (void)dev_free(Pimg); (void)dev_free(Winv);
    for (agp_fit *b : blocks) agp_fit_destroy(b);
    dx.release();
  }
};

FeatView feature_rows(const FeatView &v, long long o, long long cnt) {
  FeatView r = v;
  r.n = cnt;
  r.coords = v.coords + o * v.dim;
  if (v.ids) r.ids = v.ids + o;
  if (v.scales) r.scales = v.scales + o * v.nsc;
  return r;
}

// AGP_SPARSE_TIMING=1: wall time of every stage (with a stream synchronisation at each boundary) on stderr
struct StageTimer {
  hipStream_t s;
  bool on;
  std::chrono::steady_clock::time_point last;
  explicit StageTimer(hipStream_t st, bool enabled) : s(st), on(enabled), last(std::chrono::steady_clock::now()) {}
  void operator()(const char *name) {
    if (!on) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  [sparse fit] %-28s %8.2f ms\n", name, std::chrono::duration<double, std::milli>(now - last).count());
    last = now;
  }
};

#define SPX_HIP(expr)                                                                    \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      return AGP_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)

// C (m x m, lower tiles) -= W W^T for W m x n with n >> m: 136 tiles of 128 x 128 at m = 2048 would leave most of the
// chip idle behind one very long K loop, so the columns are cut into `slab_count` equal slices that run as one batched
// launch into separate m x m slabs, summed in a fixed order afterwards (deterministic; no atomics).
// The slice count is the one (up to 32, slices of at least 4096 columns) whose tiles fill whole rounds of the chip's 512
// workgroup slots best: 6 slices of the 136 tiles at m = 2048 were 1.6 rounds (49 TFLOP/s), 15 are 3.98.
static long long syrk_slabs(long long m, long long n) {
  const long long tr = (m + 127) / 128, tiles = tr * (tr + 1) / 2;
  constexpr long long slots = 512;
  long long best = 1;
  double best_fill = 0.;
  for (long long s = 1; s <= 32; ++s) {
    if (s > 1 && n / s < 4096) break;
    const long long t = tiles * s, rounds = (t + slots - 1) / slots;
    // (a launch of less than one round is as long as one tile: prefer more slices until a round is full)
    const double fill = (double)t / (double)(rounds * slots) * (rounds >= 2 ? 1. : 0.5);
    if (fill > best_fill + 1e-9) { best_fill = fill; best = s; }
  }
  return best;
}

__global__ __launch_bounds__(256) void sum_slabs_kernel(double *C, const double *__restrict__ slabs, long long elems, long long count) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= elems) return;
  double acc = 0.;
  for (long long b = 0; b < count; ++b) acc += slabs[b * elems + i];
  C[i] += acc;
}

static void syrk_over_observations(hipStream_t s, SparseScratch &w, double *C, long long ldc, const double *W, long long ldw,
                                   long long m, long long n) {
  const long long S = w.slab_count;
  const long long per = S > 1 ? n / S : 0;
  if (S <= 1 || !w.slabs || per <= 0) {
    launch_gemm_nt_sub(s, C, ldc, W, ldw, false, W, ldw, false, m, m, n, true);
    return;
  }
  const long long elems = ldc * m;
  (void)hipMemsetAsync(w.slabs, 0, sizeof(double) * (size_t)elems * (size_t)S, s);
  launch_gemm_nt_sub_batched(s, w.slabs, ldc, elems, W, ldw, false, per * ldw, W, ldw, false, per * ldw, m, m, per, true, S);
  const long long rest = n - per * S;
  if (rest > 0) launch_gemm_nt_sub(s, C, ldc, W + (size_t)(per * S) * (size_t)ldw, ldw, false, W + (size_t)(per * S) * (size_t)ldw, ldw, false, m, m, rest, true);
  hipLaunchKernelGGL(sum_slabs_kernel, dim3((unsigned)((elems + 255) / 256)), dim3(256), 0, s, C, w.slabs, elems, S);
}

// The data-dependent half of compute_internal_components (sparse_gp.hpp:642-704) for one set of
// observations: uploads x / y / y_var, builds K_uf, P = L_u^-1 K_uf, the blocks of
// A = K_ff + target variance - P_g^T P_g + measurement nugget with their LL^T, and returns
//   w.Kuf = W = K_uf A^-T/2 (m x n, ld round_up(m, 2)),   yw = A^-1/2 y (inside w.vecs),   log|A|.
int sparse_observations(agp_context *ctx, const agp_kernel *k, const DevProgram *dprog, const agp_features *x,
                        int64_t n_groups, const int64_t *offsets, const double *y, const double *y_var,
                        double measurement_nugget, const FeatView &uv, const agp_fit *kuu, SparseScratch &w,
                        double **yw_out, double *log_det_a_out, StageTimer &stage, const agp_ldlt *kp = nullptr) {
  const long long n = x->n, m = uv.n;
  hipStream_t s = ctx->stream;
  int st = AGP_OK;
  long long smax = 0;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long sg = offsets[g + 1] - offsets[g];
    if (sg <= 0) return AGP_ERR_INVALID_ARGUMENT;
    if (sg > smax) smax = sg;
  }
  if ((st = to_device(ctx, x, false, &w.dx)) != AGP_OK) return st;
  FeatView xm = w.dx.v;
  xm.meas = 1;  // as_measurements(out_of_order_features), sparse_gp.hpp:649-650
  const long long ldk = round_up(m, 2), np2 = round_up(n, 2);
  // vectors: dvar (n) | yw (n) | t (n)
  SPX_HIP(dev_malloc(&w.vecs, sizeof(double) * (size_t)(3 * np2)));
  double *dvar = w.vecs, *yw = dvar + np2, *tvec = yw + np2;
  const hipMemcpyKind kind = x->location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  SPX_HIP(hipMemcpyAsync(yw, y, sizeof(double) * (size_t)n, kind, s));
  if (y_var) {
    SPX_HIP(hipMemcpyAsync(tvec, y_var, sizeof(double) * (size_t)n, kind, s));
    if (x->location == AGP_HOST) SPX_HIP(hipStreamSynchronize(s));
    launch_axpby(s, n, 1.0, tvec, measurement_nugget, nullptr, dvar);  // target variance + measurement nugget, :692-696
  } else {
    if (x->location == AGP_HOST) SPX_HIP(hipStreamSynchronize(s));
    launch_axpby(s, n, 0.0, nullptr, measurement_nugget, nullptr, dvar);
  }
  stage("upload");
  // K_uf (m x n) and P = K_uu^-1/2 K_uf = L_u^-1 K_uf  (:669-685)
  {
    // pool: K_uf | P, later Q1^T (ldk x (n + m)) | split-K slabs of the m x m products over the observations
    const long long ldm_p = factor_ld(m);
    w.slab_count = syrk_slabs(m, n);
    // the padded lock-step path of ragged groups (below) adds two zero-padded copies of P and K_uf
    bool all_equal = true;
    for (int64_t g = 0; g < n_groups; ++g) all_equal = all_equal && (offsets[g + 1] - offsets[g] == smax);
    const bool padded_path = !all_equal && n_groups >= 4 && smax * n_groups <= 3 * n;
    const size_t kuf_e = (size_t)ldk * (size_t)n, pq_e = (size_t)ldk * (size_t)(n + m),
                 slab_e = (size_t)w.slab_count * (size_t)ldm_p * (size_t)m,
                 pad_e = padded_path ? (size_t)ldk * (size_t)(smax * n_groups) : 0;
    if ((st = ensure_ws(ctx, &ctx->pool_sparse, &ctx->pool_sparse_bytes,
                        sizeof(double) * (kuf_e + pq_e + slab_e + 2 * pad_e))) != AGP_OK)
      return st;
    w.Kuf = ctx->pool_sparse;
    w.Pbuf = w.Kuf + kuf_e;
    w.Q1T = w.Pbuf;
    w.slabs = w.Pbuf + pq_e;
    w.pads = w.slabs + slab_e;
  }
  launch_gram(s, dprog, uv, xm, false, false, w.Kuf, ldk, nullptr, nullptr, &k->prog);
  if (kp) {  // P = K_uu_ldlt.sqrt_solve(K_uf) with the pivoted L D L^T (:680-685)
    ldlt_sqrt_solve(s, kp->A, kp->lda, m, kp->q_dev, w.Pbuf, w.Kuf, ldk, n);
  } else if (forward_solve_wide_ok(m, n)) {
    // many more observations than inducing points: out of place through the inverted 512 x 512 diagonal blocks
    if (!w.Winv) SPX_HIP(dev_malloc(&w.Winv, sizeof(double) * (size_t)m * (size_t)WIDE_BW));
    invert_wide_blocks(s, kuu->A, m, kuu->lda, kuu->invd, WIDE_BW, w.Winv);
    forward_solve_wide(s, kuu->A, m, kuu->lda, w.Winv, w.Kuf, ldk, w.Pbuf, ldk, n);
  } else {
    SPX_HIP(hipMemcpyAsync(w.Pbuf, w.Kuf, sizeof(double) * (size_t)ldk * (size_t)n, hipMemcpyDeviceToDevice, s));
    forward_solve_mat(s, kuu->A, m, kuu->lda, kuu->invd, w.Pbuf, n, ldk);
  }
  SPX_HIP(hipStreamSynchronize(s));
  stage("K_uf, P = L_u^-1 K_uf");
  // A block by block, then W = K_uf A^-T/2 (in place in K_uf) and y_w = A^-1/2 y  (:652-704; B's top block
  // transposed, :347-349; :372)
  bool uniform = true;
  for (int64_t g = 0; g < n_groups; ++g) uniform = uniform && (offsets[g + 1] - offsets[g] == smax);
  double log_det_a = 0.;
  if (uniform) {
    // All groups have the same size: the blocks advance in LOCK STEP through batched launches
    // (blockIdx.y = group) - a dozen launches for the whole of A instead of ~40 per block, which
    // is what the per-block path below is bound by (the HIP launch path is serial per process).
    const long long sb = smax, lda_b = factor_ld(sb), nblk_b = (sb + NB - 1) / NB;
    const long long stride_A = lda_b * sb, stride_I = nblk_b * (36 * MB * MB);
    SPX_HIP(dev_malloc(&w.Ag, sizeof(double) * (size_t)stride_A * (size_t)n_groups));
    SPX_HIP(dev_malloc(&w.Pimg, sizeof(double) * ((size_t)stride_I + 1) * (size_t)n_groups));
    double *logsum = w.Pimg + (size_t)stride_I * (size_t)n_groups;
    SPX_HIP(hipMemsetAsync(logsum, 0, sizeof(double) * (size_t)n_groups, s));
    SPX_HIP(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
    // K_gg + target variance, all groups: one launch for the radial fast-path kernels, else one per group
    if (!launch_gram_blocks(s, &k->prog, xm, sb, n_groups, w.Ag, lda_b, stride_A, dvar, ctx->d_flags))
      for (int64_t g = 0; g < n_groups; ++g) {
        const FeatView xg = feature_rows(xm, g * sb, sb);
        launch_gram(s, dprog, xg, xg, true, true, w.Ag + g * stride_A, lda_b, dvar + g * sb, ctx->d_flags, &k->prog);
      }
    // A_g -= P_g^T P_g, all groups
    launch_gemm_nt_sub_batched(s, w.Ag, lda_b, stride_A, w.Pbuf, ldk, true, sb * ldk, w.Pbuf, ldk, true, sb * ldk, sb, sb, m,
                               true, n_groups);
    // block LL^T with y_w = A^-1/2 y carried along (fused forward substitution), then W = K_uf A^-T/2 in place
    factor_lower_batched(s, w.Ag, stride_A, sb, lda_b, w.Pimg, stride_I, yw, sb, n_groups, ctx->d_flags, logsum);
    right_solve_lt_batched(s, w.Ag, stride_A, sb, lda_b, w.Pimg, stride_I, w.Kuf, sb * ldk, m, ldk, n_groups);
    std::vector<double> hl((size_t)n_groups);
    SPX_HIP(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    SPX_HIP(hipMemcpyAsync(hl.data(), logsum, sizeof(double) * (size_t)n_groups, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipStreamSynchronize(s));
    SPX_HIP(hipGetLastError());
    if ((st = status_from_flags(ctx)) != AGP_OK) return st;
    for (int64_t g = 0; g < n_groups; ++g) log_det_a += 2. * hl[(size_t)g];  // fixed order
  } else if (n_groups >= 4 && smax * n_groups <= 3 * n) {
    // Ragged groups of comparable size: the same lock-step path on slabs of ONE width smax, every group
    // padded with an identity block (A_pad = [A 0; 0 I]); the operands P_g, K_uf[:, g], y_g are copied into
    // zero-padded slabs and W / y_w copied back.  At most 3x the compact memory.
    const long long sb = smax, lda_b = factor_ld(sb), nblk_b = (sb + NB - 1) / NB, G = n_groups;
    const long long stride_A = lda_b * sb, stride_I = nblk_b * (36 * MB * MB), padded = sb * G;
    double *Ppad = nullptr, *Kpad = nullptr, *ypad = nullptr;
    long long *off_d = nullptr;
    auto free_pads = [&]() { (void)dev_free(ypad); (void)dev_free(off_d); };
#define SPX_HIP2(expr)                                                                   \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      free_pads();                                                                       \
      return AGP_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
    SPX_HIP2(dev_malloc(&off_d, sizeof(long long) * (size_t)(G + 1)));
    SPX_HIP2(hipMemcpyAsync(off_d, offsets, sizeof(long long) * (size_t)(G + 1), hipMemcpyHostToDevice, s));
    Ppad = w.pads;  // in the pool (sized above)
    Kpad = Ppad + (size_t)ldk * (size_t)padded;
    SPX_HIP2(dev_malloc(&ypad, sizeof(double) * (size_t)round_up(padded, 2)));
    SPX_HIP2(dev_malloc(&w.Ag, sizeof(double) * (size_t)stride_A * (size_t)G));
    SPX_HIP2(dev_malloc(&w.Pimg, sizeof(double) * ((size_t)stride_I + 1) * (size_t)G));
    double *logsum = w.Pimg + (size_t)stride_I * (size_t)G;
    SPX_HIP2(hipMemsetAsync(logsum, 0, sizeof(double) * (size_t)G, s));
    SPX_HIP2(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
    launch_pad_columns(s, w.Pbuf, ldk, off_d, sb, G, m, Ppad, ldk, 0);
    launch_pad_columns(s, w.Kuf, ldk, off_d, sb, G, m, Kpad, ldk, 0);
    launch_pad_columns(s, yw, 1, off_d, sb, G, 1, ypad, 1, 0);
    for (int64_t g = 0; g < G; ++g) {
      const long long o = offsets[g], sg = offsets[g + 1] - o;
      const FeatView xg = feature_rows(xm, o, sg);
      launch_gram(s, dprog, xg, xg, true, true, w.Ag + g * stride_A, lda_b, dvar + o, ctx->d_flags, &k->prog);
    }
    launch_pad_identity(s, w.Ag, lda_b, stride_A, off_d, sb, G);
    launch_gemm_nt_sub_batched(s, w.Ag, lda_b, stride_A, Ppad, ldk, true, sb * ldk, Ppad, ldk, true, sb * ldk, sb, sb, m, true, G);
    factor_lower_batched(s, w.Ag, stride_A, sb, lda_b, w.Pimg, stride_I, ypad, sb, G, ctx->d_flags, logsum);
    right_solve_lt_batched(s, w.Ag, stride_A, sb, lda_b, w.Pimg, stride_I, Kpad, sb * ldk, m, ldk, G);
    launch_pad_columns(s, Kpad, ldk, off_d, sb, G, m, w.Kuf, ldk, 1);
    launch_pad_columns(s, ypad, 1, off_d, sb, G, 1, yw, 1, 1);
    std::vector<double> hl((size_t)G);
    SPX_HIP2(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    SPX_HIP2(hipMemcpyAsync(hl.data(), logsum, sizeof(double) * (size_t)G, hipMemcpyDeviceToHost, s));
    SPX_HIP2(hipStreamSynchronize(s));
    SPX_HIP2(hipGetLastError());
#undef SPX_HIP2
    free_pads();
    if ((st = status_from_flags(ctx)) != AGP_OK) return st;
    for (int64_t g = 0; g < G; ++g) log_det_a += 2. * hl[(size_t)g];  // fixed order
  } else {
    // few or very uneven groups: one block at a time, T host threads on T helper contexts (own streams)
    agp_context_impl *ci = static_cast<agp_context_impl *>(ctx);
    constexpr int want_threads = 4;  // measured: 4 threads 590 ms, 16: 650 ms, 32: 740 ms at 512 blocks of 512
    const int T = (int)std::min<long long>(want_threads, n_groups);
    while ((int)ci->helpers.size() < T) {
      agp_context *h = nullptr;
      if ((st = agp_context_create(ctx->device, &h)) != AGP_OK) return st;
      ci->helpers.push_back(h);
    }
    w.blocks.assign((size_t)n_groups, nullptr);
    std::vector<int> status((size_t)T, AGP_OK);
    std::vector<std::string> errors((size_t)T);
    const long long lda_g = factor_ld(smax);
    double *Kuf = w.Kuf, *Pbuf = w.Pbuf;
    auto worker = [&](int t) {
      agp_context *h = ci->helpers[(size_t)t];
      int &stt = status[(size_t)t];
      if (hipSetDevice(ctx->device) != hipSuccess) { stt = AGP_ERR_HIP; return; }
      const DevProgram *hprog = nullptr;
      if ((stt = device_program(h, k, &hprog)) != AGP_OK) return;
      double *Ag = nullptr;
      if (dev_malloc(&Ag, sizeof(double) * (size_t)lda_g * (size_t)smax) != hipSuccess) { stt = AGP_ERR_HIP; return; }
      hipStream_t hs = h->stream;
      for (int64_t g = t; g < n_groups && stt == AGP_OK; g += T) {
        const long long o = offsets[g], sg = offsets[g + 1] - o;
        const FeatView xg = feature_rows(xm, o, sg);
        launch_gram(hs, hprog, xg, xg, true, true, Ag, lda_g, dvar + o, nullptr, &k->prog);
        launch_gemm_nt_sub(hs, Ag, lda_g, Pbuf + o * ldk, ldk, true, Pbuf + o * ldk, ldk, true, sg, sg, m, true);
        agp_fit *blk = nullptr;
        stt = agp_factor_create(h, Ag, sg, lda_g, 0, AGP_DEVICE, &blk);
        w.blocks[(size_t)g] = blk;
        if (stt != AGP_OK) break;
        right_solve_lt(hs, blk->A, sg, blk->lda, blk->invd, Kuf + o * ldk, m, ldk);
        forward_solve_mat(hs, blk->A, sg, blk->lda, blk->invd, yw + o, 1, sg);
      }
      if (hipStreamSynchronize(hs) != hipSuccess && stt == AGP_OK) stt = AGP_ERR_HIP;
      if (stt != AGP_OK) errors[(size_t)t] = h->last_error;
      (void)dev_free(Ag);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto &th : pool) th.join();
    for (int t = 0; t < T; ++t)
      if (status[(size_t)t] != AGP_OK) {
        ctx->last_error = errors[(size_t)t];
        return status[(size_t)t];
      }
    for (int64_t g = 0; g < n_groups; ++g) log_det_a += w.blocks[(size_t)g]->log_det;  // fixed order
  }
  w.Pbuf = nullptr;  // dead: its region of the pool becomes Q1^T
  stage("blocks of A, W, y_w");
  *yw_out = yw;
  *log_det_a_out = log_det_a;
  return AGP_OK;
}

// Sigma^-1 = B^T B for B^T = [T | W] and the information vector v = (B^T B)^-1 (T y_t + W y_w):
//   T   m x m lower-triangular root of the "prior" part (zero above the diagonal, ld ldt):
//       L_u for a fit (compute_sigma_qr, :343-352), the old L_acc for an update (:336-339)
//   W   m x n (ld ldk), y_w (n);  y_t (m) or nullptr (zero)
// Fills f->sigma (L1), f->sigma2 (L2), f->Lacc, f->v.  bq (optional, device, 1 double) receives
// || L2^-1 L1^-1 (T y_t + W y_w) ||^2 for the likelihood.
// comm (optional): the observations are split over the ranks of a communicator BY GROUP (every rank holds the W and y_w
// of its own groups, all hold the same T): the m x m sums over observations - W W^T, Q1_W Q1_W^T - and the m-vectors
// W y_w, W (y_w - W^T v) are all-reduced, everything else is replicated arithmetic; every rank ends with the same fit.
int sparse_sigma(agp_context *ctx, agp_sparse_fit *f, const double *T, long long ldt, const double *W, long long ldk,
                 long long n, const double *yw, const double *yt, SparseScratch &w, double *bq, StageTimer &stage,
                 agp_comm *comm = nullptr) {
  const long long m = f->m, ldm = factor_ld(m), mp2 = round_up(m, 2), np2 = round_up(std::max<long long>(n, 1), 2);
  hipStream_t s = ctx->stream;
  int st = AGP_OK;
  // M = T T^T + W W^T   (the W part summed over the ranks)
  SPX_HIP(dev_malloc(&w.M0, sizeof(double) * (size_t)ldm * (size_t)m));
  SPX_HIP(hipMemsetAsync(w.M0, 0, sizeof(double) * (size_t)ldm * (size_t)m, s));
  if (n > 0) syrk_over_observations(s, w, w.M0, ldm, W, ldk, m, n);
  if ((st = comm_all_reduce_device(ctx, comm, w.M0, ldm * m, 0)) != AGP_OK) return st;
  launch_gemm_nt_sub(s, w.M0, ldm, T, ldt, false, T, ldt, false, m, m, m, true);
  launch_negate(s, w.M0, ldm, m, nullptr);
  stage("M = T T^T + W W^T");
  if ((st = agp_factor_create(ctx, w.M0, m, ldm, 0, AGP_DEVICE, &f->sigma)) != AGP_OK) return st;
  stage("factor M");
  // CholeskyQR2: Q1^T = L1^-1 [T | W]  (m x (m + n)), G = Q1^T Q1 = L2 L2^T
  SPX_HIP(hipMemcpy2DAsync(w.Q1T, sizeof(double) * (size_t)ldk, T, sizeof(double) * (size_t)ldt, sizeof(double) * (size_t)m,
                           (size_t)m, hipMemcpyDeviceToDevice, s));
  if (n > 0 && forward_solve_wide_ok(m, n)) {  // the W columns out of place (no copy), the T columns in place
    if (!w.Winv) SPX_HIP(dev_malloc(&w.Winv, sizeof(double) * (size_t)m * (size_t)WIDE_BW));
    invert_wide_blocks(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, WIDE_BW, w.Winv);
    forward_solve_wide(s, f->sigma->A, m, f->sigma->lda, w.Winv, W, ldk, w.Q1T + (size_t)ldk * (size_t)m, ldk, n);
    forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, w.Q1T, m, ldk, /*rhs_lower=*/true);
  } else {
    if (n > 0)
      SPX_HIP(hipMemcpyAsync(w.Q1T + (size_t)ldk * (size_t)m, W, sizeof(double) * (size_t)ldk * (size_t)n,
                             hipMemcpyDeviceToDevice, s));
    forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, w.Q1T, n + m, ldk);
  }
  SPX_HIP(hipMemsetAsync(w.M0, 0, sizeof(double) * (size_t)ldm * (size_t)m, s));
  if (n > 0)  // the W columns of Q1 (summed over the ranks), then the T columns
    syrk_over_observations(s, w, w.M0, ldm, w.Q1T + (size_t)ldk * (size_t)m, ldk, m, n);
  if ((st = comm_all_reduce_device(ctx, comm, w.M0, ldm * m, 0)) != AGP_OK) return st;
  launch_gemm_nt_sub(s, w.M0, ldm, w.Q1T, ldk, false, w.Q1T, ldk, false, m, m, m, true);
  launch_negate(s, w.M0, ldm, m, nullptr);
  st = agp_factor_create(ctx, w.M0, m, ldm, 0, AGP_DEVICE, &f->sigma2);
  if (st != AGP_OK) return st;
  stage("CholeskyQR2 (Q1, L2)");

  // x <- (B^T B)^-1 x = L1^-T (G^-1 (L1^-1 x))
  auto sigma_solve = [&](double *xv) -> int {
    forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, xv, 1, m);
    const int e = agp_solve(ctx, f->sigma2, xv, 1, xv, AGP_DEVICE);
    if (e != AGP_OK) return e;
    backward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, xv, 1, m);
    return AGP_OK;
  };
  // vectors: b | v | r | dv | tt (m each) | t (n) ; partial sums of the mat-vecs
  double *mv = nullptr;
  SPX_HIP(dev_malloc(&mv, sizeof(double) * (size_t)(5 * mp2 + np2 + 2)));
  std::unique_ptr<double, void (*)(double *)> mv_guard(mv, [](double *p) { (void)dev_free(p); });
  double *bvec = mv, *vvec = bvec + mp2, *rvec = vvec + mp2, *dv = rvec + mp2, *tt = dv + mp2, *tvec = tt + mp2;
  const long long cols = std::max(n, m), chunks = (cols + 1023) / 1024;
  SPX_HIP(dev_malloc(&w.partial, sizeof(double) * (size_t)chunks * (size_t)m));
  // b = T y_t + W y_w   (B^T y_aug: :370-372 for a fit, :344-350 for an update)
  if (n > 0) launch_matvec(s, W, ldk, m, n, yw, w.partial, 1.0, 0.0, nullptr, bvec);
  else launch_axpby(s, m, 0.0, nullptr, 0.0, nullptr, bvec);
  if ((st = comm_all_reduce_device(ctx, comm, bvec, m, 0)) != AGP_OK) return st;
  if (yt) launch_matvec(s, T, ldt, m, m, yt, w.partial, 1.0, 1.0, bvec, bvec);
  SPX_HIP(hipMemcpyAsync(vvec, bvec, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s));
  if ((st = sigma_solve(vvec)) != AGP_OK) return st;
  // two refinement steps against B itself: r = W (y_w - W^T v) + T (y_t - T^T v), v += (B^T B)^-1 r
  for (int it = 0; it < 2; ++it) {
    // the observations' part first (summed over the ranks), then the prior part
    if (n > 0) {
      launch_colvec_dot(s, W, ldk, m, n, vvec, -1.0, 1.0, yw, tvec);
      launch_matvec(s, W, ldk, m, n, tvec, w.partial, 1.0, 0.0, nullptr, rvec);
    } else {
      launch_axpby(s, m, 0.0, nullptr, 0.0, nullptr, rvec);
    }
    if ((st = comm_all_reduce_device(ctx, comm, rvec, m, 0)) != AGP_OK) return st;
    launch_colvec_dot(s, T, ldt, m, m, vvec, -1.0, 1.0, yt, tt);
    launch_matvec(s, T, ldt, m, m, tt, w.partial, 1.0, 1.0, rvec, rvec);
    SPX_HIP(hipMemcpyAsync(dv, rvec, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s));
    if ((st = sigma_solve(dv)) != AGP_OK) return st;
    launch_axpby(s, m, 1.0, vvec, 1.0, dv, vvec);
    if (it == 0) {
      // a first correction below 1e-10 |v| leaves nothing for a second one (corrections shrink by the same factor again):
      // one small read-back instead of two more passes over W (every rank holds the same v and takes the same branch)
      double *nrm = tvec + np2, h[2] = {1., 0.};
      launch_dot(s, dv, dv, m, nrm);
      launch_dot(s, vvec, vvec, m, nrm + 1);
      SPX_HIP(hipMemcpyAsync(h, nrm, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
      SPX_HIP(hipStreamSynchronize(s));
      if (h[0] <= 1e-20 * h[1]) break;
    }
  }
  SPX_HIP(dev_malloc(&f->v, sizeof(double) * (size_t)m));
  SPX_HIP(hipMemcpyAsync(f->v, vvec, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s));
  if (bq) {  // y_b = R^-T P^T B^T y_aug = L2^-1 L1^-1 b  (:590)
    forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, bvec, 1, m);
    forward_solve_mat(s, f->sigma2->A, m, f->sigma2->lda, f->sigma2->invd, bvec, 1, m);
    launch_dot(s, bvec, bvec, m, bq);
  }
  stage("information + refinement");
  // L_acc = L1 L2 (for update): C = 0 - L1z (L2z^T)^T, negated
  SPX_HIP(dev_malloc(&f->Lacc, sizeof(double) * (size_t)ldm * (size_t)m));
  {
    double *L1z = w.M0;  // m x ldm scratch, no longer needed
    double *L2z = nullptr;
    SPX_HIP(dev_malloc(&L2z, sizeof(double) * (size_t)ldm * (size_t)m));
    std::unique_ptr<double, void (*)(double *)> g2(L2z, [](double *p) { (void)dev_free(p); });
    SPX_HIP(hipMemcpy2DAsync(L1z, sizeof(double) * (size_t)ldm, f->sigma->A, sizeof(double) * (size_t)f->sigma->lda,
                             sizeof(double) * (size_t)m, (size_t)m, hipMemcpyDeviceToDevice, s));
    SPX_HIP(hipMemcpy2DAsync(L2z, sizeof(double) * (size_t)ldm, f->sigma2->A, sizeof(double) * (size_t)f->sigma2->lda,
                             sizeof(double) * (size_t)m, (size_t)m, hipMemcpyDeviceToDevice, s));
    launch_zero_upper(s, L1z, ldm, m);
    launch_zero_upper(s, L2z, ldm, m);
    SPX_HIP(hipMemsetAsync(f->Lacc, 0, sizeof(double) * (size_t)ldm * (size_t)m, s));
    launch_gemm_nt_sub(s, f->Lacc, ldm, L1z, ldm, false, L2z, ldm, true, m, m, m, false);
    launch_negate(s, f->Lacc, ldm, m, nullptr);
    SPX_HIP(hipStreamSynchronize(s));
  }
  SPX_HIP(hipGetLastError());
  return AGP_OK;
}

}  // namespace

extern "C" {

void agp_sparse_fit_destroy(agp_sparse_fit *f) {
  if (!f) return;
  if (f->ctx) (void)hipSetDevice(f->ctx->device);
  f->u.reset();
  f->kuu.reset();
  if (f->sigma) agp_fit_destroy(f->sigma);
  if (f->sigma2) agp_fit_destroy(f->sigma2);
  if (f->Lacc) (void)dev_free(f->Lacc);
  if (f->v) (void)dev_free(f->v);
  f->kz.reset();
  f->kp.reset();
  if (f->R) (void)dev_free(f->R);
  if (f->perm) (void)dev_free(f->perm);
  delete f;
}

int64_t agp_sparse_fit_size(const agp_sparse_fit *f) { return f ? f->m : 0; }

// B[r0 + a, c] = W[c, a]: the observations' rows of B = [R P^T; A^-1/2 K_fu] from W = K_uf A^-T/2 (m x n)
__global__ __launch_bounds__(256) void transpose_into_kernel(const double *__restrict__ W, long long ldw, long long m, long long n,
                                                             double *B, long long ldb, long long r0) {
  __shared__ double tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const long long c0 = (long long)blockIdx.x * 32, a0 = (long long)blockIdx.y * 32;
  for (int q = ty; q < 32; q += 8) {
    const long long c = c0 + tx, a = a0 + q;
    tile[q][tx] = (c < m && a < n) ? W[c + a * ldw] : 0.;
  }
  __syncthreads();
  for (int q = ty; q < 32; q += 8) {
    const long long a = a0 + tx, c = c0 + q;
    if (a < n && c < m) B[r0 + a + c * ldb] = tile[tx][q];
  }
}

__global__ __launch_bounds__(256) void add_to_diagonal_kernel(double *A, long long ld, long long n, double value) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) A[i + i * ld] += value;
}

// LL^T of K_uu + nugget I  (compute_internal_components, sparse_gp.hpp:674-679)
static int factor_kuu(agp_context *ctx, const agp_kernel *k, const DevProgram *dprog, const FeatView &uv, double nugget,
                      std::shared_ptr<agp_fit> *out, double **T_out) {
  const long long m = uv.n, ldm = factor_ld(m);
  hipStream_t s = ctx->stream;
  double *nug = nullptr, *T = nullptr;
  SPX_HIP(dev_malloc(&nug, sizeof(double) * (size_t)round_up(m, 2)));
  std::unique_ptr<double, void (*)(double *)> nug_guard(nug, [](double *p) { (void)dev_free(p); });
  launch_axpby(s, m, 0.0, nullptr, nugget, nullptr, nug);
  SPX_HIP(dev_malloc(&T, sizeof(double) * (size_t)ldm * (size_t)m));
  std::unique_ptr<double, void (*)(double *)> t_guard(T, [](double *p) { (void)dev_free(p); });
  launch_gram(s, dprog, uv, uv, true, true, T, ldm, nug, nullptr, &k->prog);
  agp_fit *kuu = nullptr;
  const int st = agp_factor_create(ctx, T, m, ldm, 0, AGP_DEVICE, &kuu);
  *out = std::shared_ptr<agp_fit>(kuu, [](agp_fit *p) { agp_fit_destroy(p); });
  if (st != AGP_OK) return st;
  if (T_out) *T_out = t_guard.release();
  return AGP_OK;
}

// T = P^T L D^1/2 (m x ldt) from a pivoted L D L^T: T T^T = P^T L D L^T P is the factored matrix; its transpose is the
// reference's sqrt_transpose() = D^1/2 (P^T L)^T (serializable_ldlt.hpp:111-115), the prior rows of B (:349)
__global__ __launch_bounds__(256) void ldlt_root_kernel(const double *__restrict__ A, long long lda, const long long *__restrict__ q,
                                                        long long m, double *T, long long ldt) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x, c = blockIdx.y;
  if (i >= m) return;
  const double d = A[c + c * lda];
  const double sd = d > 0. ? sqrt(d) : 0.;  // diagonal_sqrt (:74-84)
  const double l = i == c ? 1. : (i > c ? A[i + c * lda] : 0.);
  T[q[i] + c * ldt] = l * sd;
}

// pivoted L D L^T of K_uu + nugget I  (compute_internal_components, sparse_gp.hpp:674-679, as the reference factors it)
static int factor_kuu_pivoted(agp_context *ctx, const agp_kernel *k, const DevProgram *dprog, const FeatView &uv, double nugget,
                              std::shared_ptr<agp_ldlt> *out) {
  const long long m = uv.n, ldq = round_up(m, 2);
  hipStream_t s = ctx->stream;
  double *K = nullptr;
  SPX_HIP(dev_malloc(&K, sizeof(double) * (size_t)ldq * (size_t)m));
  std::unique_ptr<double, void (*)(double *)> guard(K, [](double *p) { (void)dev_free(p); });
  launch_gram(s, dprog, uv, uv, true, false, K, ldq, nullptr, nullptr, &k->prog);
  hipLaunchKernelGGL(add_to_diagonal_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, K, ldq, m, nugget);
  agp_ldlt *kz = nullptr;
  const int st = agp_ldlt_create(ctx, K, m, ldq, 0, AGP_DEVICE, &kz, nullptr);
  *out = std::shared_ptr<agp_ldlt>(kz, [](agp_ldlt *p) { agp_ldlt_destroy(p); });
  return st;
}

static int sparse_pivoted_from_rows(agp_context *ctx, agp_sparse_fit *f, const double *T, long long ldt, const double *W,
                                    long long ldw, long long n, const double *yt, const double *yw, bool inflate);

// _fit_impl (:354-381) with the reference's own factorisations - pivoted L D L^T of K_uu + nugget, column-pivoted QR of
// B = [A^-1/2 K_fu; K_uu^T/2] - for the cases the LL^T / CholeskyQR2 path below rejects (K_uu or B^T B singular to
// working precision: inducing points denser than the length scale).  Level-2 bound: moderate m.
static int sparse_fit_create_pivoted(agp_context *ctx, const agp_kernel *k, const agp_features *x, int64_t n_groups,
                                     const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                                     double measurement_nugget, double inducing_nugget, agp_sparse_fit **out,
                                     double *information, double *nll_out) {
  if (!ctx || !k || !x || !u || !y || !offsets || n_groups <= 0) return AGP_ERR_INVALID_ARGUMENT;
  if (out) *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st == AGP_OK) st = validate_features(u);
  if (st != AGP_OK) return st;
  const long long n = x->n, m = u->n, ldm = factor_ld(m), ldk = round_up(m, 2);
  if (n <= 0 || m <= 0 || u->dim != x->dim || offsets[0] != 0 || offsets[n_groups] != n) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  hipStream_t s = ctx->stream;
  StageTimer stage(s, getenv("AGP_SPARSE_TIMING") != nullptr);  // (development aid: prints the stage times of one fit)
  std::unique_ptr<agp_sparse_fit, void (*)(agp_sparse_fit *)> f(new (std::nothrow) agp_sparse_fit(), agp_sparse_fit_destroy);
  if (!f) return AGP_ERR_INVALID_ARGUMENT;
  f->ctx = ctx; f->m = m; f->inducing_nugget = inducing_nugget;
  SparseScratch w;
  f->u = std::shared_ptr<DeviceFeatures>(new DeviceFeatures(), [](DeviceFeatures *d) { d->release(); delete d; });
  if ((st = to_device(ctx, u, true, f->u.get())) != AGP_OK) return st;
  if ((st = factor_kuu_pivoted(ctx, k, dprog, f->u->v, inducing_nugget, &f->kz)) != AGP_OK) return st;
  f->kp = f->kz;
  SPX_HIP(dev_malloc(&w.T, sizeof(double) * (size_t)ldm * (size_t)m));
  SPX_HIP(hipMemsetAsync(w.T, 0, sizeof(double) * (size_t)ldm * (size_t)m, s));
  hipLaunchKernelGGL(ldlt_root_kernel, dim3((unsigned)((m + 255) / 256), (unsigned)m), dim3(256), 0, s, f->kz->A, f->kz->lda,
                     f->kz->q_dev, m, w.T, ldm);
  stage("K_uu + pivoted factor");
  double *yw = nullptr, log_det_a = 0.;
  if ((st = sparse_observations(ctx, k, dprog, x, n_groups, offsets, y, y_var, measurement_nugget, f->u->v, nullptr, w, &yw,
                                &log_det_a, stage, f->kp.get())) != AGP_OK)
    return st;
  if ((st = sparse_pivoted_from_rows(ctx, f.get(), w.T, ldm, w.Kuf, ldk, n, nullptr, yw, false)) != AGP_OK) return st;
  stage("pivoted QR of B");
  // negative log likelihood (:524-596): log|K| = log|A| + 2 log|R| - log|K_uu|; q = y^T A^-1 y - y_b^T y_b, y_b = R^-T P^T K_uf A^-1 y
  const long long chunks = (std::max(n, m) + 1023) / 1024;
  double *vb = nullptr;
  SPX_HIP(dev_malloc(&vb, sizeof(double) * ((size_t)chunks * (size_t)m + (size_t)3 * (size_t)ldk)));
  std::unique_ptr<double, void (*)(double *)> vb_guard(vb, [](double *p) { (void)dev_free(p); });
  double *partial = vb, *bvec = vb + (size_t)chunks * (size_t)m, *yb = bvec + ldk, *rdiag = yb + ldk;
  launch_matvec(s, w.Kuf, ldk, m, n, yw, partial, 1.0, 0.0, nullptr, bvec);
  qr_sqrt_solve(s, f->R, ldk, f->perm, m, bvec, ldk, yb, ldk, 1);
  launch_dot(s, yw, yw, n, ctx->d_scalars + 1);
  launch_dot(s, yb, yb, m, ctx->d_scalars + 2);
  SPX_HIP(hipMemcpy2DAsync(rdiag, sizeof(double), f->R, sizeof(double) * (size_t)(ldk + 1), sizeof(double), (size_t)m,
                           hipMemcpyDeviceToDevice, s));
  std::vector<double> hr((size_t)m);
  SPX_HIP(hipMemcpyAsync(hr.data(), rdiag, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
  SPX_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (information) SPX_HIP(hipMemcpyAsync(information, f->v, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
  SPX_HIP(hipStreamSynchronize(s));
  SPX_HIP(hipGetLastError());
  double log_det_r = 0., log_det_kuu = 0.;
  for (long long i = 0; i < m; ++i) {
    log_det_r += std::log(std::fabs(hr[(size_t)i]));     // matrixR().diagonal().array().cwiseAbs().log().sum() (:549-550)
    log_det_kuu += std::log(f->kz->d[(size_t)i]);       // vectorD().array().log().sum() (serializable_ldlt.hpp:128-135)
  }
  const double log_det = log_det_a + 2. * log_det_r - log_det_kuu;
  f->nll = 0.5 * (log_det + (ctx->h_scalars[1] - ctx->h_scalars[2]) + (double)n * std::log(2 * M_PI));
  if (nll_out) *nll_out = f->nll;
  if (out) *out = f.release();
  return AGP_OK;
}

// One status for all ranks: the largest code any rank holds (AGP_OK = 0).  Every exit of the sharded fit that only ONE
// rank may take - bad arguments, a non-positive-definite or NaN block of its own groups, an allocation failure - goes
// through here BEFORE the next device collective, so that no rank is left waiting inside an all-reduce for a peer that
// has already returned.
static int agree_status(agp_comm *comm, int st) {
  if (!comm) return st;
  double v = (double)st;
  if (agp_comm_all_reduce_host(comm, &v, 1, 1) != AGP_OK) return AGP_ERR_COMM;
  return (int)v;
}

// 52-bit checksum of a feature set's coordinates (host or device), exact in a double
static int features_checksum(agp_context *ctx, const agp_features *u, double *out) {
  const size_t cnt = (size_t)u->n * (size_t)u->dim;
  std::vector<double> host;
  const double *p = u->coords;
  if (u->location != AGP_HOST) {
    host.resize(cnt);
    SPX_HIP(hipMemcpy(host.data(), u->coords, sizeof(double) * cnt, hipMemcpyDeviceToHost));
    p = host.data();
  }
  unsigned long long h = 1469598103934665603ull;
  for (size_t i = 0; i < cnt; ++i) {
    unsigned long long b;
    std::memcpy(&b, p + i, sizeof(b));
    h = (h ^ b) * 1099511628211ull;
    h ^= h >> 29;
  }
  *out = (double)(h & ((1ull << 52) - 1));
  return AGP_OK;
}

static int sparse_fit_create_fast(agp_context *ctx, agp_comm *comm, const agp_kernel *k, const agp_features *x, int64_t n_groups,
                                  const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                                  double measurement_nugget, double inducing_nugget, agp_sparse_fit **out,
                                  double *information, double *nll_out) {
  if (out) *out = nullptr;
  // ---- argument checks: rank-local, agreed before anything else when the fit is sharded ----
  int st = AGP_OK;
  if (!ctx || !k || !x || !u || !y || !offsets || n_groups <= 0) st = AGP_ERR_INVALID_ARGUMENT;
  if (st == AGP_OK && hipSetDevice(ctx->device) != hipSuccess) st = AGP_ERR_HIP;
  if (st == AGP_OK) st = validate_features(x);
  if (st == AGP_OK) st = validate_features(u);
  if (st == AGP_OK && (x->n <= 0 || u->n <= 0 || u->dim != x->dim || offsets[0] != 0 || offsets[n_groups] != x->n))
    st = AGP_ERR_INVALID_ARGUMENT;
  if (comm) {
    // every rank must pass the SAME inducing points (the m x m sums below add matrices built against them): compare a
    // checksum of u over the ranks - max and min in one all-reduce - together with the status
    double cks = 0.;
    if (st == AGP_OK && ctx) st = features_checksum(ctx, u, &cks);
    double v[3] = {(double)st, cks, -cks};
    if (agp_comm_all_reduce_host(comm, v, 3, 1) != AGP_OK) return AGP_ERR_COMM;
    if (v[0] != 0.) return st != AGP_OK ? st : (int)v[0];
    if (v[1] != -v[2]) {
      ctx->last_error = "agp_sparse_fit_create_sharded: the inducing points differ between the ranks";
      return AGP_ERR_INVALID_ARGUMENT;
    }
  } else if (st != AGP_OK) {
    return st;
  }
  const long long n = x->n, m = u->n;
  hipStream_t s = ctx->stream;
  StageTimer stage(s, getenv("AGP_SPARSE_TIMING") != nullptr);  // (development aid: prints the stage times of one fit)
  std::unique_ptr<agp_sparse_fit, void (*)(agp_sparse_fit *)> f(new (std::nothrow) agp_sparse_fit(), agp_sparse_fit_destroy);
  SparseScratch w;
  const long long ldm = factor_ld(m);
  double *yw = nullptr, log_det_a = 0.;
  // ---- everything up to the first device collective: K_uu (replicated arithmetic) and this rank's own groups ----
  auto local_part = [&]() -> int {
    const DevProgram *dprog = nullptr;
    int e = device_program(ctx, k, &dprog);
    if (e != AGP_OK) return e;
    if (!f) return AGP_ERR_INVALID_ARGUMENT;
    f->ctx = ctx; f->m = m; f->inducing_nugget = inducing_nugget;
    f->u = std::shared_ptr<DeviceFeatures>(new DeviceFeatures(), [](DeviceFeatures *d) { d->release(); delete d; });
    if ((e = to_device(ctx, u, true, f->u.get())) != AGP_OK) return e;

    // K_uu + inducing_nugget I  (:674-679) -> LL^T ; T = L_u with explicit zeros above the diagonal
    if ((e = factor_kuu(ctx, k, dprog, f->u->v, inducing_nugget, &f->kuu, &w.T)) != AGP_OK) return e;
    SPX_HIP(hipMemcpy2DAsync(w.T, sizeof(double) * (size_t)ldm, f->kuu->A, sizeof(double) * (size_t)f->kuu->lda,
                             sizeof(double) * (size_t)m, (size_t)m, hipMemcpyDeviceToDevice, s));
    launch_zero_upper(s, w.T, ldm, m);  // K_uu^T/2 = L_u^T (sqrt_transpose, :349): its transpose L_u
    stage("K_uu + factor");
    return sparse_observations(ctx, k, dprog, x, n_groups, offsets, y, y_var, measurement_nugget, f->u->v, f->kuu.get(), w, &yw,
                               &log_det_a, stage);
  };
  st = local_part();
  {
    const int agreed = agree_status(comm, st);
    if (agreed != AGP_OK) return st != AGP_OK ? st : agreed;  // a peer failed: this rank reports the peer's code
  }
  if ((st = sparse_sigma(ctx, f.get(), w.T, ldm, w.Kuf, round_up(m, 2), n, yw, nullptr, w, ctx->d_scalars + 2, stage, comm)) != AGP_OK)
    return st;
  launch_dot(s, yw, yw, n, ctx->d_scalars + 1);  // y^T A^-1 y = y_w^T y_w  (:583-592); after the factor calls, which reset the scalars
  // negative log likelihood (:524-596): log|K| = log|A| + log|B^T B| - log|K_uu'|
  SPX_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (information) SPX_HIP(hipMemcpyAsync(information, f->v, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
  if (comm) {
    if ((st = comm_wait_stream(ctx, s)) != AGP_OK) return st;  // bounded: a dead peer is AGP_ERR_COMM, not a hang
  } else {
    SPX_HIP(hipStreamSynchronize(s));
  }
  SPX_HIP(hipGetLastError());
  // sums over the observations of all ranks: log|A|, y^T A^-1 y, n
  double sums[3] = {log_det_a, ctx->h_scalars[1], (double)n};
  if (comm && (st = agp_comm_all_reduce_host(comm, sums, 3, 0)) != AGP_OK) return st;
  const double log_det = sums[0] + (f->sigma->log_det + f->sigma2->log_det) - f->kuu->log_det;
  f->nll = 0.5 * (log_det + (sums[1] - ctx->h_scalars[2]) + sums[2] * std::log(2 * M_PI));
  if (nll_out) *nll_out = f->nll;
  if (out) *out = f.release();
  return AGP_OK;
}

// The LL^T / CholeskyQR2 path first; where it finds K_uu or B^T B not numerically positive definite the reference's
// pivoted algorithm takes over (one process only).  AGP_SPARSE_PIVOTED=1 forces the pivoted path.
static int sparse_fit_create_impl(agp_context *ctx, agp_comm *comm, const agp_kernel *k, const agp_features *x, int64_t n_groups,
                                  const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                                  double measurement_nugget, double inducing_nugget, agp_sparse_fit **out,
                                  double *information, double *nll_out) {
  int st = AGP_ERR_NOT_POSITIVE_DEFINITE;
  if (comm || !ctx->tune.sparse_pivoted)
    st = sparse_fit_create_fast(ctx, comm, k, x, n_groups, offsets, y, y_var, u, measurement_nugget, inducing_nugget, out,
                                information, nll_out);
  if (st == AGP_ERR_NOT_POSITIVE_DEFINITE && !comm)
    st = sparse_fit_create_pivoted(ctx, k, x, n_groups, offsets, y, y_var, u, measurement_nugget, inducing_nugget, out,
                                   information, nll_out);
  return st;
}

int agp_sparse_fit_create(agp_context *ctx, const agp_kernel *k, const agp_features *x, int64_t n_groups,
                          const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                          double measurement_nugget, double inducing_nugget, agp_sparse_fit **out,
                          double *information, double *nll_out) {
  return sparse_fit_create_impl(ctx, nullptr, k, x, n_groups, offsets, y, y_var, u, measurement_nugget, inducing_nugget, out,
                                information, nll_out);
}

int agp_sparse_fit_create_sharded(agp_context *ctx, agp_comm *comm, const agp_kernel *k, const agp_features *x, int64_t n_groups,
                                  const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                                  double measurement_nugget, double inducing_nugget, agp_sparse_fit **out,
                                  double *information, double *nll_out) {
  if (!comm) return AGP_ERR_INVALID_ARGUMENT;
  return sparse_fit_create_impl(ctx, comm, k, x, n_groups, offsets, y, y_var, u, measurement_nugget, inducing_nugget, out,
                                information, nll_out);
}

// Pivoted form of Sigma from the rows of B: B = [T^T; W^T] ((m + n) x m; T m x m with ld ldt, W m x n with ld ldw, either
// part may be absent) and, optionally, y_aug = [y_t; y_w] carried through the QR as one more column.  Fills f->R, f->perm,
// f->rank, f->Lacc = P R^T and - with a right-hand side - f->v = B_qr.solve(y_aug).  inflate: the update's
// "inflate the diagonal of R" when the QR is rank deficient (:361-365).
static int sparse_pivoted_from_rows(agp_context *ctx, agp_sparse_fit *f, const double *T, long long ldt, const double *W,
                                    long long ldw, long long n, const double *yt, const double *yw, bool inflate) {
  const long long m = f->m, rows = (T ? m : 0) + n, ldb = round_up(rows, 2), ldr = round_up(m, 2), ldm = factor_ld(m);
  const long long extra = (yt || yw) ? 1 : 0;
  hipStream_t s = ctx->stream;
  double *B = nullptr, *aux = nullptr;
  SPX_HIP(dev_malloc(&B, sizeof(double) * (size_t)ldb * (size_t)(m + extra)));
  std::unique_ptr<double, void (*)(double *)> b_guard(B, [](double *p) { (void)dev_free(p); });
  SPX_HIP(dev_malloc(&aux, sizeof(double) * (size_t)(2 * ldr + 8)));  // tau | norms | state
  std::unique_ptr<double, void (*)(double *)> a_guard(aux, [](double *p) { (void)dev_free(p); });
  double *tau = aux, *norms = aux + ldr, *state = norms + ldr;
  SPX_HIP(hipMemsetAsync(B, 0, sizeof(double) * (size_t)ldb * (size_t)(m + extra), s));
  long long r0 = 0;
  const dim3 tb(256);
  if (T) {
    hipLaunchKernelGGL(transpose_into_kernel, dim3((unsigned)((m + 31) / 32), (unsigned)((m + 31) / 32)), tb, 0, s, T, ldt, m, m, B,
                       ldb, 0ll);
    if (yt) SPX_HIP(hipMemcpyAsync(B + (size_t)ldb * (size_t)m, yt, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s));
    r0 = m;
  }
  if (n > 0) {
    hipLaunchKernelGGL(transpose_into_kernel, dim3((unsigned)((m + 31) / 32), (unsigned)((n + 31) / 32)), tb, 0, s, W, ldw, m, n, B,
                       ldb, r0);
    if (yw) SPX_HIP(hipMemcpyAsync(B + (size_t)ldb * (size_t)m + r0, yw, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
  }
  if (!f->perm) SPX_HIP(dev_malloc(&f->perm, sizeof(long long) * (size_t)m));
  colpiv_qr(s, B, ldb, rows, m, extra, tau, f->perm, norms, state);
  double hstate[4] = {0., 0., 0., 0.};
  SPX_HIP(hipMemcpyAsync(hstate, state, sizeof(hstate), hipMemcpyDeviceToHost, s));
  SPX_HIP(hipStreamSynchronize(s));
  SPX_HIP(hipGetLastError());
  f->rank = (long long)hstate[3];
  const long long nonzero_pivots = (long long)hstate[2];
  if (!f->R) SPX_HIP(dev_malloc(&f->R, sizeof(double) * (size_t)ldr * (size_t)m));
  qr_extract_r(s, B, ldb, m, f->R, ldr, 0.0);
  if (extra) {
    if (!f->v) SPX_HIP(dev_malloc(&f->v, sizeof(double) * (size_t)m));
    qr_back_solve(s, f->R, ldr, f->perm, m, nonzero_pivots, B + (size_t)ldb * (size_t)m, f->v);  // the last column is Q^T y_aug
  }
  if (inflate && f->rank < m) qr_extract_r(s, B, ldb, m, f->R, ldr, 1e-10);
  if (!f->Lacc) SPX_HIP(dev_malloc(&f->Lacc, sizeof(double) * (size_t)ldm * (size_t)m));
  qr_root(s, f->R, ldr, f->perm, m, f->Lacc, ldm);
  SPX_HIP(hipStreamSynchronize(s));
  SPX_HIP(hipGetLastError());
  return AGP_OK;
}

// FitModel::update for the sparse GP: _update_impl (sparse_gp.hpp:322-371).  B = [R_old P_old^T; A^-1/2 K_fu],
// y_aug = [R_old P_old^T v_old; A^-1/2 y]: with the kept root L_acc (L_acc L_acc^T = Sigma_old^-1) that is
// B^T = [L_acc | W_new], y_t = L_acc^T v_old.  The inducing points and their K_uu factor are shared with the old fit.
int agp_sparse_fit_update(agp_context *ctx, const agp_kernel *k, const agp_sparse_fit *old, const agp_features *x,
                          int64_t n_groups, const int64_t *offsets, const double *y, const double *y_var,
                          double measurement_nugget, agp_sparse_fit **out, double *information) {
  if (!ctx || !k || !old || !x || !y || !offsets || !out || n_groups <= 0) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st != AGP_OK) return st;
  const long long n = x->n, m = old->m;
  if (n <= 0 || x->dim != old->u->v.dim || offsets[0] != 0 || offsets[n_groups] != n) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  hipStream_t s = ctx->stream;
  StageTimer stage(s, getenv("AGP_SPARSE_TIMING") != nullptr);  // (development aid: prints the stage times of one fit)
  std::unique_ptr<agp_sparse_fit, void (*)(agp_sparse_fit *)> f(new (std::nothrow) agp_sparse_fit(), agp_sparse_fit_destroy);
  if (!f) return AGP_ERR_INVALID_ARGUMENT;
  f->ctx = ctx; f->m = m;
  f->u = old->u;
  f->kuu = old->kuu;
  f->kz = old->kz;
  f->kp = old->kp;
  f->inducing_nugget = old->inducing_nugget;
  if (!f->kuu && !f->kp) {  // a rebased fit's first update: K_uu + inducing nugget (:674-679), LL^T where that exists
    st = factor_kuu(ctx, k, dprog, f->u->v, old->inducing_nugget, &f->kuu, nullptr);
    if (st == AGP_ERR_NOT_POSITIVE_DEFINITE) {
      f->kuu.reset();
      st = factor_kuu_pivoted(ctx, k, dprog, f->u->v, old->inducing_nugget, &f->kp);
    }
    if (st != AGP_OK) return st;
  }
  SparseScratch w;
  double *yw = nullptr, log_det_a = 0.;
  if ((st = sparse_observations(ctx, k, dprog, x, n_groups, offsets, y, y_var, measurement_nugget, f->u->v, f->kuu.get(), w,
                                &yw, &log_det_a, stage, f->kuu ? nullptr : f->kp.get())) != AGP_OK)
    return st;
  const long long ldm = factor_ld(m);
  double *yt = nullptr;
  SPX_HIP(dev_malloc(&yt, sizeof(double) * (size_t)round_up(m, 2)));
  std::unique_ptr<double, void (*)(double *)> yt_guard(yt, [](double *p) { (void)dev_free(p); });
  launch_colvec_dot(s, old->Lacc, ldm, m, m, old->v, 1.0, 0.0, nullptr, yt);  // y_t = L_acc^T v_old  (:344-347)
  if (old->R) {
    // pivoted form: the reference's own algorithm (:336-371) - column-pivoted QR of B = [R_old P_old^T; A^-1/2 K_fu]
    // with y_aug = [R_old P_old^T v_old; A^-1/2 y] carried as one more column, v = B_qr.solve(y_aug)
    if ((st = sparse_pivoted_from_rows(ctx, f.get(), old->Lacc, ldm, w.Kuf, round_up(m, 2), n, yt, yw, true)) != AGP_OK) return st;
    if (information) SPX_HIP(hipMemcpyAsync(information, f->v, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
    SPX_HIP(hipStreamSynchronize(s));
    f->nll = std::nan("");
    *out = f.release();
    return AGP_OK;
  }
  st = sparse_sigma(ctx, f.get(), old->Lacc, ldm, w.Kuf, round_up(m, 2), n, yw, yt, w, nullptr, stage);
  if (st == AGP_ERR_NOT_POSITIVE_DEFINITE) {  // B^T B singular to working precision: the pivoted QR of B instead
    if (f->sigma) { agp_fit_destroy(f->sigma); f->sigma = nullptr; }
    if (f->sigma2) { agp_fit_destroy(f->sigma2); f->sigma2 = nullptr; }
    st = sparse_pivoted_from_rows(ctx, f.get(), old->Lacc, ldm, w.Kuf, round_up(m, 2), n, yt, yw, true);
  }
  if (st != AGP_OK) return st;
  if (information) SPX_HIP(hipMemcpyAsync(information, f->v, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
  SPX_HIP(hipStreamSynchronize(s));
  f->nll = std::nan("");  // the likelihood of an updated fit is not defined by the reference
  *out = f.release();
  return AGP_OK;
}

// SparseGaussianProcessRegression::fit_from_prediction (sparse_gp.hpp:406-461), the body of rebase_inducing_points
// (:714-725): the fit on the inducing points z that reproduces a joint prediction (mean, covariance) made AT z.
//   train_covariance = LDLT(K_zz), information = train_covariance.solve(mean),
//   C = covariance + DEFAULT_NUGGET I, B_z = C^-1/2 K_zz = C_ldlt.sqrt_solve(K_zz), (R, P) = QR(B_z)
// K_zz carries no nugget and is singular to working precision whenever z is denser than the length scale, so this path
// keeps the reference's pivoted factorisations (ldlt.hip, qr.hip) instead of the LL^T / CholeskyQR2 of a fit.
int agp_sparse_fit_from_prediction(agp_context *ctx, const agp_kernel *k, const agp_features *z, const double *mean,
                                   const double *covariance, int64_t ldc, int location, double inducing_nugget,
                                   agp_sparse_fit **out, double *information, int64_t *numerical_rank) {
  if (!ctx || !k || !z || !mean || !covariance || !out) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(z);
  if (st != AGP_OK) return st;
  const long long m = z->n, ldq = round_up(m, 2);
  if (m <= 0 || ldc < m) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  hipStream_t s = ctx->stream;
  std::unique_ptr<agp_sparse_fit, void (*)(agp_sparse_fit *)> f(new (std::nothrow) agp_sparse_fit(), agp_sparse_fit_destroy);
  if (!f) return AGP_ERR_INVALID_ARGUMENT;
  f->ctx = ctx; f->m = m; f->inducing_nugget = inducing_nugget;
  f->u = std::shared_ptr<DeviceFeatures>(new DeviceFeatures(), [](DeviceFeatures *d) { d->release(); delete d; });
  if ((st = to_device(ctx, z, true, f->u.get())) != AGP_OK) return st;
  // buffers: K_zz | C (later B_z) | W (m x m each, ld ldq) | mean, scratch (2 ldq)
  double *buf = nullptr;
  SPX_HIP(dev_malloc(&buf, sizeof(double) * ((size_t)3 * (size_t)ldq * (size_t)m + (size_t)2 * (size_t)ldq)));
  std::unique_ptr<double, void (*)(double *)> guard(buf, [](double *p) { (void)dev_free(p); });
  double *Kzz = buf, *C = Kzz + (size_t)ldq * (size_t)m, *Bz = C + (size_t)ldq * (size_t)m, *mv = Bz + (size_t)ldq * (size_t)m,
         *mw = mv + ldq;
  launch_gram(s, dprog, f->u->v, f->u->v, true, false, Kzz, ldq, nullptr, nullptr, &k->prog);  // K_zz, both triangles (:416-417)
  {
    agp_ldlt *kz = nullptr;
    st = agp_ldlt_create(ctx, Kzz, m, ldq, 0, AGP_DEVICE, &kz, nullptr);                      // train_covariance (:418)
    f->kz = std::shared_ptr<agp_ldlt>(kz, [](agp_ldlt *p) { agp_ldlt_destroy(p); });
    if (st != AGP_OK) return st;
  }
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  SPX_HIP(hipMemcpyAsync(mv, mean, sizeof(double) * (size_t)m, kind, s));
  SPX_HIP(hipMemcpy2DAsync(C, sizeof(double) * (size_t)ldq, covariance, sizeof(double) * (size_t)ldc, sizeof(double) * (size_t)m,
                           (size_t)m, kind, s));
  if (location == AGP_HOST) SPX_HIP(hipStreamSynchronize(s));
  ldlt_solve(s, f->kz->A, f->kz->lda, m, f->kz->q_dev, mw, mv, ldq, 1);                        // information (:426)
  SPX_HIP(dev_malloc(&f->v, sizeof(double) * (size_t)m));
  SPX_HIP(hipMemcpyAsync(f->v, mv, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s));
  {
    // DEFAULT_NUGGET on the diagonal of the predictive covariance (:20, 423-425)
    hipLaunchKernelGGL(add_to_diagonal_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, C, ldq, m, 1e-8);
    agp_ldlt *cl = nullptr;
    st = agp_ldlt_create(ctx, C, m, ldq, 0, AGP_DEVICE, &cl, nullptr);                        // C_ldlt (:452)
    std::unique_ptr<agp_ldlt, void (*)(agp_ldlt *)> cl_guard(cl, agp_ldlt_destroy);
    if (st != AGP_OK) return st;
    ldlt_sqrt_solve(s, cl->A, cl->lda, m, cl->q_dev, Bz, Kzz, ldq, m);                        // sigma_inv_sqrt (:453)
    SPX_HIP(hipStreamSynchronize(s));
  }
  // (R, P) = QR(B_z) (:454-458): B_z is handed over as W^T, i.e. W = B_z^T ... the helper transposes, so pass B_z^T's
  // transpose: rows of B are the rows of B_z
  {
    double *Wt = C;  // C is free now: W = B_z^T (m x m)
    hipLaunchKernelGGL(transpose_into_kernel, dim3((unsigned)((m + 31) / 32), (unsigned)((m + 31) / 32)), dim3(256), 0, s, Bz, ldq, m,
                       m, Wt, ldq, 0ll);
    if ((st = sparse_pivoted_from_rows(ctx, f.get(), nullptr, 0, Wt, ldq, m, nullptr, nullptr, false)) != AGP_OK) return st;
  }
  if (information) SPX_HIP(hipMemcpyAsync(information, f->v, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
  SPX_HIP(hipStreamSynchronize(s));
  SPX_HIP(hipGetLastError());
  if (numerical_rank) *numerical_rank = f->rank;
  f->nll = std::nan("");
  *out = f.release();
  return AGP_OK;
}

int64_t agp_sparse_fit_numerical_rank(const agp_sparse_fit *f) { return f ? (f->rank >= 0 ? f->rank : f->m) : 0; }

int agp_sparse_nll(agp_context *ctx, const agp_kernel *k, const agp_features *x, int64_t n_groups,
                   const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                   double measurement_nugget, double inducing_nugget, double *out) {
  if (!out) return AGP_ERR_INVALID_ARGUMENT;
  return agp_sparse_fit_create(ctx, k, x, n_groups, offsets, y, y_var, u, measurement_nugget, inducing_nugget, nullptr,
                               nullptr, out);
}

int agp_sparse_fit_information(agp_context *ctx, const agp_sparse_fit *f, double *information) {
  if (!ctx || !f || !information) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return copy_out(ctx, f->v, f->m, information, AGP_HOST);
}

// _predict_impl x 3 (sparse_gp.hpp:447-521): mean = K_*u v ; C = K_** - Q_sqrt^T Q_sqrt + S_sqrt^T S_sqrt with
// Q_sqrt = K_uu^-1/2 K_u* = L_u^-1 K_u* and S_sqrt = R^-T P^T K_u* == L2^-1 L1^-1 K_u*
static int sparse_predict_common(agp_context *ctx, const agp_kernel *k, const agp_sparse_fit *f, const agp_features *xs,
                                 double *mean, double *var_or_cov, int mode, int out_location) {
  if (!ctx || !k || !f || !xs || !mean || (mode > 0 && !var_or_cov)) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(xs);
  if (st != AGP_OK) return st;
  if (xs->dim != f->u->v.dim) return AGP_ERR_INVALID_ARGUMENT;
  const long long M_all = xs->n, m = f->m;
  if (M_all == 0) return AGP_OK;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dxs;
  if ((st = to_device(ctx, xs, false, &dxs)) != AGP_OK) return st;
  const long long ldq = round_up(m, 2);
  // mean and marginal predictions pass over the test points in slices that keep the m x M buffers at 2 GiB each
  // (AGP_PREDICT_CHUNK=<points> overrides); a joint prediction needs all of them at once
  long long chunk = M_all;
  if (mode != 2) {
    const long long forced = ctx->tune.predict_chunk;
    const long long c = forced > 0 ? forced : std::min<long long>(1LL << 20, std::max<long long>(1024, (1LL << 28) / ldq));
    chunk = std::min(M_all, c);
  }
  const long long ldc = round_up(chunk, 2);
  const size_t q_elems = (size_t)ldq * (size_t)chunk;
  const size_t p_elems = mode == 2 ? (size_t)ldc * (size_t)chunk : (size_t)ldc;
  const bool pivoted = f->kz || f->R;  // a third m x M buffer: the pivoted substitutions are out of place
  st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes,
                 sizeof(double) * ((pivoted ? 3 : 2) * q_elems + (size_t)ldc + p_elems));
  if (st != AGP_OK) { dxs.release(); return st; }
  double *Q = ctx->ws_aux, *S = Q + q_elems, *mean_d = S + q_elems, *prior = mean_d + ldc, *X = prior + p_elems;
  hipStream_t s = ctx->stream;
  for (long long o = 0; o < M_all && st == AGP_OK; o += chunk) {
    const long long M = std::min(chunk, M_all - o);
    FeatView xv = dxs.v;
    if (!(o == 0 && M == M_all)) {
      xv.sstride = scale_stride(dxs.v);
      xv.n = M;
      xv.coords = dxs.v.coords + o * dxs.v.dim;
      xv.ids = dxs.v.ids ? dxs.v.ids + o : nullptr;
      xv.scales = dxs.v.scales ? dxs.v.scales + o : nullptr;
    }
    const size_t q_used = (size_t)ldq * (size_t)M;
    launch_predict_mean(s, dprog, f->u->v, xv, f->v, mean_d, &k->prog);
    if (mode > 0 && !pivoted) {
      launch_gram(s, dprog, f->u->v, xv, false, false, Q, ldq, nullptr, nullptr, &k->prog);
      (void)hipMemcpyAsync(S, Q, sizeof(double) * q_used, hipMemcpyDeviceToDevice, s);
      forward_solve_mat(s, f->kuu->A, m, f->kuu->lda, f->kuu->invd, Q, M, ldq);
      forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, S, M, ldq);
      forward_solve_mat(s, f->sigma2->A, m, f->sigma2->lda, f->sigma2->invd, S, M, ldq);
    } else if (mode > 0) {
      launch_gram(s, dprog, f->u->v, xv, false, false, X, ldq, nullptr, nullptr, &k->prog);
      if (f->kz) {  // Q_sqrt = train_covariance.sqrt_solve(cross_cov) with the pivoted L D L^T (:497-498)
        ldlt_sqrt_solve(s, f->kz->A, f->kz->lda, m, f->kz->q_dev, Q, X, ldq, M);
      } else {
        (void)hipMemcpyAsync(Q, X, sizeof(double) * q_used, hipMemcpyDeviceToDevice, s);
        forward_solve_mat(s, f->kuu->A, m, f->kuu->lda, f->kuu->invd, Q, M, ldq);
      }
      if (f->R) {   // S_sqrt = sqrt_solve(R, P, cross_cov) = R^-T P^T cross_cov (:503-504)
        qr_sqrt_solve(s, f->R, ldq, f->perm, m, X, ldq, S, ldq, M);
      } else {
        (void)hipMemcpyAsync(S, X, sizeof(double) * q_used, hipMemcpyDeviceToDevice, s);
        forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, S, M, ldq);
        forward_solve_mat(s, f->sigma2->A, m, f->sigma2->lda, f->sigma2->invd, S, M, ldq);
      }
    }
    if (mode == 1) {
      launch_gram_diagonal(s, dprog, xv, prior);
      launch_coldot(s, Q, ldq, Q, ldq, m, M, prior, 1.0, prior);   // - Q_diag
      launch_coldot(s, S, ldq, S, ldq, m, M, prior, -1.0, prior);  // + S_diag
    } else if (mode == 2) {
      launch_gram(s, dprog, xv, xv, true, false, prior, ldc, nullptr, nullptr, &k->prog);
      launch_gemm_nt_sub(s, prior, ldc, Q, ldq, true, Q, ldq, true, M, M, m, true);  // - max_explained
      launch_axpby(s, (long long)q_used, -1.0, S, 0.0, nullptr, Q);  // Q <- -S
      launch_gemm_nt_sub(s, prior, ldc, Q, ldq, true, S, ldq, true, M, M, m, true);  // + unexplained
      launch_symmetrize(s, prior, ldc, M);
    }
    st = copy_out(ctx, mean_d, M, mean + o, out_location);
    if (st == AGP_OK && mode == 1) st = copy_out(ctx, prior, M, var_or_cov + o, out_location);
    if (st == AGP_OK && mode == 2) st = copy_out_2d(ctx, prior, ldc, M, M, var_or_cov, M, out_location);
  }
  dxs.release();
  return st;
}

int agp_sparse_predict_mean(agp_context *ctx, const agp_kernel *k, const agp_sparse_fit *f, const agp_features *xs,
                            double *mean, int out_location) {
  return sparse_predict_common(ctx, k, f, xs, mean, nullptr, 0, out_location);
}
int agp_sparse_predict_marginal(agp_context *ctx, const agp_kernel *k, const agp_sparse_fit *f,
                                const agp_features *xs, double *mean, double *variance, int out_location) {
  return sparse_predict_common(ctx, k, f, xs, mean, variance, 1, out_location);
}
int agp_sparse_predict_joint(agp_context *ctx, const agp_kernel *k, const agp_sparse_fit *f, const agp_features *xs,
                             double *mean, double *covariance, int out_location) {
  return sparse_predict_common(ctx, k, f, xs, mean, covariance, 2, out_location);
}

}  // extern "C"
