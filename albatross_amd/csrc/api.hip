// api.hip — the C-ABI of include/albatross_amd.h.
//
// Host-side orchestration only: uploads POD feature vectors, enqueues the HIP
// kernels of gram.hip / chol.hip / gemm.hip / reduce.hip on the context's
// stream, reads back the small results.  No CPU arithmetic fallback exists.
#include <atomic>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <new>
#include <thread>

#include "common.h"

namespace agp {
void launch_symmetrize(hipStream_t s, double *A, long long ld, long long n);
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
void launch_colvec_dot(hipStream_t s, const double *W, long long ld, long long m, long long n, const double *v,
                       double alpha, double beta, const double *base, double *out);
void launch_axpby(hipStream_t s, long long n, double a, const double *x, double b, const double *y, double *out);
void launch_loo(hipStream_t s, const double *kinv_diag, const double *y, const double *information, long long n,
                double *mean, double *variance);

void DeviceFeatures::release() {
  for (void *&p : owned) {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  v = FeatView{};
}

static long long round_up(long long x, long long m) { return (x + m - 1) / m * m; }

// leading dimension of the factor: even (16-B aligned columns) and not a
// multiple of 256 doubles, so that consecutive columns do not alias the same
// HBM channel / L2 set pattern
static long long factor_ld(long long n) {
  long long ld = round_up(n, 8);
  if (ld % 256 == 0) ld += 8;
  return ld;
}

}  // namespace agp

using namespace agp;

struct ProgSlot {
  unsigned long long uid = 0;
  DevProgram *dev = nullptr;
};

struct agp_context_ext {
  ProgSlot slots[8];
  int next = 0;
};

static std::atomic<unsigned long long> g_kernel_uid{1};

struct agp_kernel_full : agp_kernel {
  unsigned long long uid;
};

// context extension kept out of common.h (host-only bookkeeping)
static agp_context_ext *ext_of(agp_context *ctx);

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

static agp_context_ext *ext_of(agp_context *ctx) { return &static_cast<agp_context_impl *>(ctx)->ext; }

extern "C" {

const char *agp_status_string(int status) {
  switch (status) {
  case AGP_OK: return "ok";
  case AGP_ERR_INVALID_ARGUMENT: return "invalid argument";
  case AGP_ERR_NAN_INPUT: return "covariance matrix contains NaN";
  case AGP_ERR_NOT_POSITIVE_DEFINITE: return "covariance matrix is not positive definite";
  case AGP_ERR_HIP: return "HIP runtime error";
  case AGP_ERR_COMM: return "communication error";
  case AGP_ERR_UNSUPPORTED: return "unsupported";
  case AGP_ERR_NO_DEVICE: return "no HIP device";
  default: return "unknown status";
  }
}

int agp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

int agp_context_create(int device_id, agp_context **out) {
  if (!out) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return AGP_ERR_NO_DEVICE;
  if (device_id < 0 || device_id >= n) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = new (std::nothrow) agp_context_impl();
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  ctx->device = device_id;
  AGP_HIP_CHECK(ctx, hipSetDevice(device_id));
  {
    // main / panel stream at the highest priority: its short kernels must not
    // queue behind the bulk update running on stream2
    int lo = 0, hi = 0;
    AGP_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
    AGP_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, hi));
    AGP_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, lo));
    AGP_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, lo));
    AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_c, hipEventDisableTiming));
  }
  AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
  AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
  AGP_HIP_CHECK(ctx, hipMalloc(&ctx->d_flags, 4 * sizeof(int)));
  AGP_HIP_CHECK(ctx, hipMalloc(&ctx->d_scalars, 4 * sizeof(double)));
  AGP_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_flags, 4 * sizeof(int)));
  AGP_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_scalars, 4 * sizeof(double)));
  for (auto &e : ctx->stage_ev) AGP_HIP_CHECK(ctx, hipEventCreate(&e));
  for (auto &sl : ctx->ext.slots) AGP_HIP_CHECK(ctx, hipMalloc(&sl.dev, sizeof(DevProgram)));
  *out = ctx;
  return AGP_OK;
}

void agp_context_destroy(agp_context *c) {
  if (!c) return;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  (void)hipSetDevice(ctx->device);
  (void)hipDeviceSynchronize();
  for (agp_context *h : ctx->helpers) agp_context_destroy(h);
  ctx->helpers.clear();
  for (auto &sl : ctx->ext.slots)
    if (sl.dev) (void)hipFree(sl.dev);
  for (auto e : ctx->gemm_events) (void)hipEventDestroy(e);
  for (auto e : ctx->stage_ev)
    if (e) (void)hipEventDestroy(e);
  if (ctx->partial_ws) (void)hipFree(ctx->partial_ws);
  if (ctx->ws_A) (void)hipFree(ctx->ws_A);
  if (ctx->pool_A) (void)hipFree(ctx->pool_A);
  if (ctx->ws_aux) (void)hipFree(ctx->ws_aux);
  if (ctx->d_flags) (void)hipFree(ctx->d_flags);
  if (ctx->d_scalars) (void)hipFree(ctx->d_scalars);
  if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
  if (ctx->h_scalars) (void)hipHostFree(ctx->h_scalars);
  if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
  if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
  if (ctx->ev_c) (void)hipEventDestroy(ctx->ev_c);
  delete ctx;
}

int agp_context_synchronize(agp_context *ctx) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

const char *agp_last_error(const agp_context *ctx) { return ctx ? ctx->last_error.c_str() : ""; }

int agp_set_profiling(agp_context *ctx, int enabled) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  ctx->profiling = enabled != 0;
  return AGP_OK;
}

int agp_last_stage_ms(const agp_context *c, int stage, double *ms) {
  if (!c || !ms || stage < 0 || stage > 7) return AGP_ERR_INVALID_ARGUMENT;
  *ms = c->stage_ms[stage];
  return AGP_OK;
}

int agp_mfma_f64_peak(agp_context *ctx, int iters, double *tflops) {
  if (!ctx || !tflops || iters <= 0) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return mfma_f64_peak(ctx->stream, iters, tflops);
}

// ---- covariance function ---------------------------------------------------
int agp_kernel_create(const agp_kernel_node *postfix, int n_nodes, agp_kernel **out) {
  if (!postfix || !out || n_nodes <= 0 || n_nodes > AGP_MAX_KERNEL_NODES) return AGP_ERR_INVALID_ARGUMENT;
  int depth = 0, mask = 0, uses_eq = 0;
  for (int t = 0; t < n_nodes; ++t) {
    const agp_kernel_node &nd = postfix[t];
    switch (nd.op) {
    case AGP_OP_SQUARED_EXPONENTIAL:
    case AGP_OP_EXPONENTIAL:
    case AGP_OP_MATERN32:
    case AGP_OP_MATERN52:
      if (nd.metric < 0 || nd.metric > AGP_METRIC_ANGULAR) return AGP_ERR_INVALID_ARGUMENT;
      mask |= 1 << nd.metric;
      ++depth;
      break;
    case AGP_OP_CONSTANT: ++depth; break;
    case AGP_OP_INDEPENDENT_NOISE:
    case AGP_OP_NUGGET: uses_eq = 1; ++depth; break;
    case AGP_OP_POLYNOMIAL:
      if (nd.order < 0 || nd.order > 3) return AGP_ERR_UNSUPPORTED;
      ++depth;
      break;
    case AGP_OP_SCALING:
      if (nd.column < 0 || nd.column >= AGP_MAX_SCALE_COLUMNS) return AGP_ERR_INVALID_ARGUMENT;
      ++depth;
      break;
    case AGP_OP_SUM:
    case AGP_OP_PRODUCT:
      if (depth < 2) return AGP_ERR_INVALID_ARGUMENT;
      --depth;
      break;
    case AGP_OP_MEASUREMENT_ONLY:
      if (depth < 1) return AGP_ERR_INVALID_ARGUMENT;
      break;
    default: return AGP_ERR_INVALID_ARGUMENT;
    }
    if (depth > AGP_MAX_STACK) return AGP_ERR_UNSUPPORTED;
  }
  if (depth != 1) return AGP_ERR_INVALID_ARGUMENT;
  agp_kernel_full *k = new (std::nothrow) agp_kernel_full();
  if (!k) return AGP_ERR_INVALID_ARGUMENT;
  std::memset(&k->prog, 0, sizeof(k->prog));
  k->prog.n_nodes = n_nodes;
  k->prog.metric_mask = mask;
  k->prog.uses_equality = uses_eq;
  std::memcpy(k->prog.nodes, postfix, sizeof(agp_kernel_node) * (size_t)n_nodes);
  k->uid = g_kernel_uid.fetch_add(1);
  *out = k;
  return AGP_OK;
}

void agp_kernel_destroy(agp_kernel *k) { delete static_cast<agp_kernel_full *>(k); }

}  // extern "C"

// device copy of a kernel program (small LRU ring per context)
static int device_program(agp_context *ctx, const agp_kernel *k, const DevProgram **out) {
  agp_context_ext *x = ext_of(ctx);
  const unsigned long long uid = static_cast<const agp_kernel_full *>(k)->uid;
  for (auto &sl : x->slots)
    if (sl.uid == uid) {
      *out = sl.dev;
      return AGP_OK;
    }
  ProgSlot &sl = x->slots[x->next];
  x->next = (x->next + 1) % 8;
  // the slot may still be read by kernels in flight on the stream
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  AGP_HIP_CHECK(ctx, hipMemcpy(sl.dev, &k->prog, sizeof(DevProgram), hipMemcpyHostToDevice));
  sl.uid = uid;
  *out = sl.dev;
  return AGP_OK;
}

namespace agp {
int device_program_for(agp_context *ctx, const agp_kernel *k, const DevProgram **out) {
  return device_program(ctx, k, out);
}
}  // namespace agp

static int validate_features(const agp_features *f) {
  if (!f || f->n < 0 || f->dim < 1 || f->dim > AGP_MAX_DIM) return AGP_ERR_INVALID_ARGUMENT;
  if (f->n_scale_columns < 0 || f->n_scale_columns > AGP_MAX_SCALE_COLUMNS) return AGP_ERR_INVALID_ARGUMENT;
  if (f->n > 0 && !f->coords) return AGP_ERR_INVALID_ARGUMENT;
  if (f->n_scale_columns > 0 && f->n > 0 && !f->scales) return AGP_ERR_INVALID_ARGUMENT;
  return AGP_OK;
}

// Make a device view of a feature vector (uploads host data; `copy` forces an
// owned device copy of device-resident data as well).
static int to_device(agp_context *ctx, const agp_features *f, bool copy, DeviceFeatures *out) {
  const int st = validate_features(f);
  if (st != AGP_OK) return st;
  out->release();
  FeatView v;
  v.n = f->n; v.dim = f->dim; v.nsc = f->n_scale_columns; v.meas = f->is_measurement;
  v.coords = nullptr; v.ids = nullptr; v.scales = nullptr;
  const bool on_host = f->location == AGP_HOST;
  const hipMemcpyKind kind = on_host ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  if (f->n > 0) {
    if (on_host || copy) {
      const size_t cb = sizeof(double) * (size_t)f->n * (size_t)f->dim;
      AGP_HIP_CHECK(ctx, hipMalloc(&out->owned[0], cb));
      AGP_HIP_CHECK(ctx, hipMemcpyAsync(out->owned[0], f->coords, cb, kind, ctx->stream));
      v.coords = static_cast<const double *>(out->owned[0]);
      if (f->eq_id) {
        const size_t ib = sizeof(long long) * (size_t)f->n;
        AGP_HIP_CHECK(ctx, hipMalloc(&out->owned[1], ib));
        AGP_HIP_CHECK(ctx, hipMemcpyAsync(out->owned[1], f->eq_id, ib, kind, ctx->stream));
        v.ids = static_cast<const long long *>(out->owned[1]);
      }
      if (f->n_scale_columns > 0) {
        const size_t sb = sizeof(double) * (size_t)f->n * (size_t)f->n_scale_columns;
        AGP_HIP_CHECK(ctx, hipMalloc(&out->owned[2], sb));
        AGP_HIP_CHECK(ctx, hipMemcpyAsync(out->owned[2], f->scales, sb, kind, ctx->stream));
        v.scales = static_cast<const double *>(out->owned[2]);
      }
      if (on_host) AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));  // pageable source
    } else {
      v.coords = f->coords;
      v.ids = reinterpret_cast<const long long *>(f->eq_id);
      v.scales = f->n_scale_columns > 0 ? f->scales : nullptr;
    }
  }
  out->v = v;
  return AGP_OK;
}

namespace agp {
int features_to_device(agp_context *ctx, const agp_features *f, bool copy, DeviceFeatures *out) {
  return to_device(ctx, f, copy, out);
}
}  // namespace agp

static int ensure_ws(agp_context *ctx, double **ws, size_t *have, size_t need) {
  if (*have >= need) return AGP_OK;
  if (*ws) {
    AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    AGP_HIP_CHECK(ctx, hipFree(*ws));
    *ws = nullptr;
    *have = 0;
  }
  AGP_HIP_CHECK(ctx, hipMalloc(ws, need));
  *have = need;
  return AGP_OK;
}

// a device staging copy of an n-vector living at `location`
static int vector_to_device(agp_context *ctx, const double *src, long long n, int location, double *dst) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, kind, ctx->stream));
  if (location == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

static int copy_out(agp_context *ctx, const double *dev, long long count, double *dst, int location) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(dst, dev, sizeof(double) * (size_t)count, kind, ctx->stream));
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

static int copy_out_2d(agp_context *ctx, const double *dev, long long ld_dev, long long rows, long long cols,
                       double *dst, long long ld_dst, int location) {
  if (location == AGP_HOST && ld_dev == ld_dst) {  // one contiguous transfer instead of one per column
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(dst, dev, sizeof(double) * ((size_t)ld_dev * (size_t)(cols - 1) + (size_t)rows),
                                      hipMemcpyDeviceToHost, ctx->stream));
    AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
    return AGP_OK;
  }
  if (location == AGP_HOST && (size_t)rows * (size_t)cols >= (1u << 16)) {
    // re-pitch on the device, then one contiguous transfer
    double *tmp = nullptr;
    const size_t elems = (size_t)ld_dst * (size_t)(cols - 1) + (size_t)rows;
    AGP_HIP_CHECK(ctx, hipMalloc(&tmp, sizeof(double) * elems));
    hipError_t e = hipMemcpy2DAsync(tmp, sizeof(double) * (size_t)ld_dst, dev, sizeof(double) * (size_t)ld_dev,
                                    sizeof(double) * (size_t)rows, (size_t)cols, hipMemcpyDeviceToDevice, ctx->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(dst, tmp, sizeof(double) * elems, hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    (void)hipFree(tmp);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return AGP_ERR_HIP; }
    return AGP_OK;
  }
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpy2DAsync(dst, sizeof(double) * (size_t)ld_dst, dev, sizeof(double) * (size_t)ld_dev,
                                      sizeof(double) * (size_t)rows, (size_t)cols, kind, ctx->stream));
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

// Gram + diag add + LL^T (+ fused forward substitution) on A / y.  On return
// the stream has been synchronised and ctx->h_flags / h_scalars are valid.
static int build_and_factor(agp_context *c, const DevProgram *dprog, const DevProgram *hprog, const FeatView &xm,
                            double *A, long long lda, double *invd, double *y, const double *yvar) {
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  const long long n = xm.n;
  hipStream_t s = ctx->stream;
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_scalars, 0, 4 * sizeof(double), s));
  const bool prof = ctx->profiling;
  if (prof) AGP_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[0], s));
  // as_measurements(features) -> covariance_function_(measurement_features)   gp.hpp:288-290
  launch_gram(s, dprog, xm, xm, /*symmetric=*/true, /*lower_only=*/true, A, lda, yvar, ctx->d_flags, hprog);
  if (prof) AGP_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[1], s));
  FactorTimers timers;
  if (prof) {
    const size_t want = (size_t)(2 * (2 * ((n + NB - 1) / NB) + 4));
    while (ctx->gemm_events.size() < want) {
      hipEvent_t e;
      AGP_HIP_CHECK(ctx, hipEventCreate(&e));
      ctx->gemm_events.push_back(e);
    }
    ctx->gemm_flops.assign(want / 2, 0.);
    timers.ev = ctx->gemm_events.data();
    timers.flops = ctx->gemm_flops.data();
    timers.n_ev = (int)want;
  }
  factor_lower(ctx, A, n, lda, invd, y, prof ? &timers : nullptr);
  if (prof) AGP_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[2], s));
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  if (prof) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->stage_ev[0], ctx->stage_ev[1]);
    ctx->stage_ms[0] = ms;
    (void)hipEventElapsedTime(&ms, ctx->stage_ev[1], ctx->stage_ev[2]);
    ctx->stage_ms[1] = ms;
    double sum = 0., flop = 0.;
    for (int i = 0; i + 1 < timers.used; i += 2) {
      (void)hipEventElapsedTime(&ms, timers.ev[i], timers.ev[i + 1]);
      sum += ms;
      flop += timers.flops[i / 2];
    }
    ctx->stage_ms[3] = sum;
    ctx->stage_ms[4] = timers.used / 2;
    ctx->stage_ms[5] = flop;  // flop of the trailing updates (not ms)
  }
  return AGP_OK;
}

static int status_from_flags(const agp_context *ctx) {
  if (ctx->h_flags[0]) return AGP_ERR_NAN_INPUT;
  if (ctx->h_flags[1]) return AGP_ERR_NOT_POSITIVE_DEFINITE;
  return AGP_OK;
}

extern "C" {

// ---- Gram ------------------------------------------------------------------
int agp_gram(agp_context *ctx, const agp_kernel *k, const agp_features *x, const agp_features *y, double *out,
             int64_t ld, int out_location) {
  if (!ctx || !k || !x || !out) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st != AGP_OK) return st;
  if (y && (st = validate_features(y)) != AGP_OK) return st;
  if (y && y->dim != x->dim) return AGP_ERR_INVALID_ARGUMENT;
  const long long rows = x->n, cols = y ? y->n : x->n;
  if (rows == 0 || cols == 0) return AGP_OK;
  if (ld < rows) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dx, dy;
  if ((st = to_device(ctx, x, false, &dx)) != AGP_OK) return st;
  if (y && (st = to_device(ctx, y, false, &dy)) != AGP_OK) { dx.release(); return st; }
  const FeatView &vy = y ? dy.v : dx.v;
  if (out_location == AGP_DEVICE) {
    launch_gram(ctx->stream, dprog, dx.v, vy, y == nullptr, false, out, ld, nullptr, nullptr, &k->prog);
    st = AGP_OK;
    hipError_t e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); st = AGP_ERR_HIP; }
  } else {
    const long long ldd = round_up(rows, 2);
    st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)ldd * (size_t)cols);
    if (st == AGP_OK) {
      launch_gram(ctx->stream, dprog, dx.v, vy, y == nullptr, false, ctx->ws_aux, ldd, nullptr, nullptr, &k->prog);
      st = copy_out_2d(ctx, ctx->ws_aux, ldd, rows, cols, out, ld, AGP_HOST);
    }
  }
  dx.release();
  dy.release();
  return st;
}

// ---- fit -------------------------------------------------------------------
void agp_fit_destroy(agp_fit *fit) {
  if (!fit) return;
  (void)hipSetDevice(fit->device);
  if (fit->A) {
    agp_context *ctx = fit->ctx;
    if (ctx && !ctx->pool_A) {
      // kernels reading the factor were enqueued on the context's streams; the
      // next user of the buffer is enqueued on the same streams, after them
      ctx->pool_A = fit->A;
      ctx->pool_A_bytes = fit->A_bytes;
    } else {
      (void)hipFree(fit->A);
    }
  }
  if (fit->invd) (void)hipFree(fit->invd);
  if (fit->winv) (void)hipFree(fit->winv);
  if (fit->alpha) (void)hipFree(fit->alpha);
  if (fit->z) (void)hipFree(fit->z);
  fit->train.release();
  delete fit;
}

int agp_fit_create(agp_context *c, const agp_kernel *k, const agp_features *x, const double *y,
                   const double *y_var, agp_fit **out, double *information, double *log_det) {
  if (!c || !k || !x || !y || !out) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st != AGP_OK) return st;
  const long long n = x->n;
  if (n <= 0) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;

  agp_fit *fit = new (std::nothrow) agp_fit();
  if (!fit) return AGP_ERR_INVALID_ARGUMENT;
  fit->device = ctx->device;
  fit->n = n;
  fit->lda = factor_ld(n);
  const long long nblk = (n + NB - 1) / NB;
  hipStream_t s = ctx->stream;
  double *yvar_d = nullptr;
#define FIT_CHECK(expr)                                                                  \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      if (yvar_d) (void)hipFree(yvar_d);                                                 \
      agp_fit_destroy(fit);                                                              \
      return AGP_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
  // train_features = features (un-wrapped; gp.hpp:63,293): always an owned copy
  if ((st = to_device(ctx, x, true, &fit->train)) != AGP_OK) { agp_fit_destroy(fit); return st; }
  fit->train.v.meas = 0;
  fit->ctx = ctx;
  fit->A_bytes = sizeof(double) * (size_t)fit->lda * (size_t)n;
  if (ctx->pool_A && ctx->pool_A_bytes == fit->A_bytes) {
    fit->A = ctx->pool_A;
    ctx->pool_A = nullptr;
    ctx->pool_A_bytes = 0;
  } else {
    FIT_CHECK(hipMalloc(&fit->A, fit->A_bytes));
  }
  FIT_CHECK(hipMalloc(&fit->invd, sizeof(double) * (size_t)nblk * (36 * MB * MB)));
  FIT_CHECK(hipMalloc(&fit->winv, sizeof(double) * (size_t)nblk * NB * NB));
  FIT_CHECK(hipMalloc(&fit->alpha, sizeof(double) * (size_t)n));
  FIT_CHECK(hipMalloc(&fit->z, sizeof(double) * (size_t)n));
  const hipMemcpyKind kind = x->location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  FIT_CHECK(hipMemcpyAsync(fit->z, y, sizeof(double) * (size_t)n, kind, s));
  if (y_var) {
    FIT_CHECK(hipMalloc(&yvar_d, sizeof(double) * (size_t)n));
    FIT_CHECK(hipMemcpyAsync(yvar_d, y_var, sizeof(double) * (size_t)n, kind, s));
  }
  if (x->location == AGP_HOST) FIT_CHECK(hipStreamSynchronize(s));
  FeatView xm = fit->train.v;
  xm.meas = 1;  // as_measurements(features), gp.hpp:288
  st = build_and_factor(ctx, dprog, &k->prog, xm, fit->A, fit->lda, fit->invd, fit->z, yvar_d);
  if (yvar_d) { (void)hipFree(yvar_d); yvar_d = nullptr; }
  if (st != AGP_OK) { agp_fit_destroy(fit); return st; }
  st = status_from_flags(ctx);
  fit->failed_pivot = ctx->h_flags[1] ? (int64_t)ctx->h_flags[1] - 1 : -1;
  fit->log_det = 2. * ctx->h_scalars[0];
  if (st != AGP_OK) {
    // keep a handle so the caller can query the failed pivot, but no factor
    *out = fit;
    return st;
  }
  // information = L^-T (L^-1 y)
  if (ctx->profiling) FIT_CHECK(hipEventRecord(ctx->stage_ev[3], s));
  FIT_CHECK(hipMemcpyAsync(fit->alpha, fit->z, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
  invert_diag_blocks(s, fit->A, n, fit->lda, fit->invd, fit->winv);
  {
    const int st2 = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)round_up(n, 2));
    if (st2 != AGP_OK) { agp_fit_destroy(fit); return st2; }
  }
  backward_solve_vec(s, fit->A, n, fit->lda, fit->winv, fit->alpha, ctx->ws_aux);
  if (ctx->profiling) FIT_CHECK(hipEventRecord(ctx->stage_ev[4], s));
  if (information) FIT_CHECK(hipMemcpyAsync(information, fit->alpha, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  FIT_CHECK(hipStreamSynchronize(s));
  FIT_CHECK(hipGetLastError());
  if (ctx->profiling) {
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, ctx->stage_ev[3], ctx->stage_ev[4]);
    ctx->stage_ms[2] = ms;
  }
  if (log_det) *log_det = fit->log_det;
  *out = fit;
#undef FIT_CHECK
  return AGP_OK;
}

int64_t agp_fit_size(const agp_fit *fit) { return fit ? fit->n : 0; }
int64_t agp_fit_failed_pivot(const agp_fit *fit) { return fit ? fit->failed_pivot : -1; }

int agp_fit_log_determinant(const agp_fit *fit, double *out) {
  if (!fit || !out) return AGP_ERR_INVALID_ARGUMENT;
  *out = fit->log_det;
  return AGP_OK;
}

int agp_fit_download_information(agp_context *ctx, const agp_fit *fit, double *information) {
  if (!ctx || !fit || !information || !fit->alpha) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  return copy_out(ctx, fit->alpha, fit->n, information, AGP_HOST);
}

int agp_fit_download_factor(agp_context *ctx, const agp_fit *fit, double *L, int64_t ld) {
  if (!ctx || !fit || !L || ld < fit->n) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = fit->n;
  int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)fit->lda * (size_t)n);
  if (st != AGP_OK) return st;
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ws_aux, fit->A, sizeof(double) * (size_t)fit->lda * (size_t)n,
                                    hipMemcpyDeviceToDevice, ctx->stream));
  launch_zero_upper(ctx->stream, ctx->ws_aux, fit->lda, n);
  return copy_out_2d(ctx, ctx->ws_aux, fit->lda, n, n, L, ld, AGP_HOST);
}

// ---- nll -------------------------------------------------------------------
int agp_nll(agp_context *c, const agp_kernel *k, const agp_features *x, const double *y, const double *y_var,
            double *out) {
  if (!c || !k || !x || !y || !out) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st != AGP_OK) return st;
  const long long n = x->n;
  if (n <= 0) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  const long long lda = factor_ld(n);
  const long long nblk = (n + NB - 1) / NB;
  // workspace: [A | invd | z | yvar]
  const size_t a_bytes = sizeof(double) * (size_t)lda * (size_t)n;
  const size_t aux = sizeof(double) * ((size_t)nblk * (36 * MB * MB) + 2 * (size_t)round_up(n, 2));
  if ((st = ensure_ws(ctx, &ctx->ws_A, &ctx->ws_A_bytes, a_bytes + aux)) != AGP_OK) return st;
  double *A = ctx->ws_A;
  double *invd = A + (size_t)lda * (size_t)n;
  double *z = invd + (size_t)nblk * (36 * MB * MB);
  double *yvar_d = y_var ? z + round_up(n, 2) : nullptr;
  DeviceFeatures dx;
  if ((st = to_device(ctx, x, false, &dx)) != AGP_OK) return st;
  if ((st = vector_to_device(ctx, y, n, x->location, z)) != AGP_OK) { dx.release(); return st; }
  if (y_var && (st = vector_to_device(ctx, y_var, n, x->location, yvar_d)) != AGP_OK) { dx.release(); return st; }
  FeatView xm = dx.v;
  xm.meas = 1;
  st = build_and_factor(ctx, dprog, &k->prog, xm, A, lda, invd, z, yvar_d);
  if (st == AGP_OK) st = status_from_flags(ctx);
  if (st == AGP_OK) {
    // mahalanobis = y^T K^-1 y = z^T z with z = L^-1 y   (likelihood.hpp:44)
    launch_dot(ctx->stream, z, z, n, ctx->d_scalars + 1);
    hipError_t e = hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); st = AGP_ERR_HIP; }
    else {
      const double log_det = 2. * ctx->h_scalars[0];
      *out = 0.5 * (log_det + ctx->h_scalars[1] + (double)n * std::log(2 * M_PI));  // likelihood.hpp:46
    }
  }
  dx.release();
  return st;
}

// ---- tuner objective batching ---------------------------------------------------------
// The negative log likelihoods of `count` parameter vectors of one model on one dataset, in lock step:
// what compute_gradient (tune/finite_difference.hpp:20-94) and the ModelTuner objective
// (tune/tune.hpp:151-161,276-290) evaluate one after the other.  `count` Gram matrices are built into
// slabs and factored by the batched kernels (blockIdx.y = parameter vector), so small and medium N pay
// the launch chain once instead of `count` times.  A parameter vector whose covariance is not positive
// definite (or has NaN) yields NaN in its slot, like the reference's NaN metric (tune.hpp:163-165).
int agp_nll_batch(agp_context *c, int count, const agp_kernel *const *kernels, const agp_features *const *features,
                  const double *y, int64_t ldy, const double *y_var, double *out) {
  if (!c || count <= 0 || !kernels || !features || !y || !out) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = features[0] ? features[0]->n : 0;
  if (n <= 0 || (ldy != 0 && ldy < n)) return AGP_ERR_INVALID_ARGUMENT;
  int st = AGP_OK;
  for (int b = 0; b < count; ++b) {
    if (!kernels[b] || !features[b] || features[b]->n != n || features[b]->location != features[0]->location)
      return AGP_ERR_INVALID_ARGUMENT;
    if ((st = validate_features(features[b])) != AGP_OK) return st;
  }
  const long long lda = factor_ld(n), nblk = (n + NB - 1) / NB, np2 = round_up(n, 2);
  const long long stride_A = lda * n, stride_I = nblk * (36 * MB * MB);
  hipStream_t s = ctx->stream;
  // workspace: [A slabs | tile images | y slabs | yvar | logsum | quad]
  const size_t elems = (size_t)count * ((size_t)stride_A + (size_t)stride_I + (size_t)np2) + (size_t)np2 +
                       2 * (size_t)round_up(count, 2);
  if ((st = ensure_ws(ctx, &ctx->ws_A, &ctx->ws_A_bytes, sizeof(double) * elems)) != AGP_OK) return st;
  double *A = ctx->ws_A, *invd = A + (size_t)count * (size_t)stride_A, *ys = invd + (size_t)count * (size_t)stride_I;
  double *yvar_d = ys + (size_t)count * (size_t)np2, *logsum = yvar_d + np2, *quad = logsum + round_up(count, 2);
  const int loc = features[0]->location;
  const hipMemcpyKind kind = loc == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  for (int b = 0; b < count; ++b)
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(ys + (size_t)b * (size_t)np2, y + (size_t)b * (size_t)ldy, sizeof(double) * (size_t)n,
                                      kind, s));
  if (y_var) AGP_HIP_CHECK(ctx, hipMemcpyAsync(yvar_d, y_var, sizeof(double) * (size_t)n, kind, s));
  if (loc == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(logsum, 0, sizeof(double) * (size_t)round_up(count, 2), s));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
  std::vector<DeviceFeatures> dxs((size_t)count);
  const agp_features *last = nullptr;
  int last_b = -1;
  for (int b = 0; b < count && st == AGP_OK; ++b) {
    const DevProgram *dprog = nullptr;
    if ((st = device_program(ctx, kernels[b], &dprog)) != AGP_OK) break;
    // parameter vectors usually share one feature array: upload it once
    const bool same = last && features[b]->coords == last->coords && features[b]->scales == last->scales &&
                      features[b]->eq_id == last->eq_id;
    if (!same) {
      if ((st = to_device(ctx, features[b], false, &dxs[(size_t)b])) != AGP_OK) break;
      last = features[b];
      last_b = b;
    }
    FeatView xm = dxs[(size_t)(same ? last_b : b)].v;
    xm.meas = 1;  // as_measurements(features), gp.hpp:288
    launch_gram(s, dprog, xm, xm, true, true, A + (size_t)b * (size_t)stride_A, lda, y_var ? yvar_d : nullptr, nullptr,
                &kernels[b]->prog);
  }
  if (st == AGP_OK) {
    factor_lower_batched(s, A, stride_A, n, lda, invd, stride_I, ys, np2, count, ctx->d_flags, logsum);
    launch_coldot(s, ys, np2, ys, np2, n, count, quad, -1.0, nullptr);  // z_b^T z_b, z_b = L_b^-1 y_b
    std::vector<double> h(2 * (size_t)round_up(count, 2));
    hipError_t e = hipMemcpyAsync(h.data(), logsum, sizeof(double) * h.size(), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e == hipSuccess) e = hipGetLastError();
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); st = AGP_ERR_HIP; }
    else
      for (int b = 0; b < count; ++b)  // likelihood.hpp:38-47
        out[b] = 0.5 * (2. * h[(size_t)b] + h[(size_t)round_up(count, 2) + (size_t)b] + (double)n * std::log(2 * M_PI));
  }
  for (auto &d : dxs) d.release();
  return st;
}

// ---- solve -----------------------------------------------------------------
int agp_solve(agp_context *ctx, const agp_fit *fit, const double *rhs, int64_t nrhs, double *out, int location) {
  if (!ctx || !fit || !rhs || !out || nrhs < 0) return AGP_ERR_INVALID_ARGUMENT;
  if (nrhs == 0) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = fit->n, ldb = round_up(n, 2);
  int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)ldb * (size_t)nrhs);
  if (st != AGP_OK) return st;
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpy2DAsync(ctx->ws_aux, sizeof(double) * (size_t)ldb, rhs, sizeof(double) * (size_t)n,
                                      sizeof(double) * (size_t)n, (size_t)nrhs, kind, ctx->stream));
  if (location == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  forward_solve_mat(ctx->stream, fit->A, n, fit->lda, fit->invd, ctx->ws_aux, nrhs, ldb);
  backward_solve_mat(ctx->stream, fit->A, n, fit->lda, fit->invd, ctx->ws_aux, nrhs, ldb);
  return copy_out_2d(ctx, ctx->ws_aux, ldb, n, nrhs, out, n, location);
}

// ---- dense-matrix factor ---------------------------------------------------------
// copies the lower triangle of K into a fresh factor buffer and runs the LL^T;
// y (device, optional) receives the fused forward substitution
static int factor_dense(agp_context *c, const double *K, long long n, long long ld, int uplo, int location,
                        agp_fit *fit, double *y) {
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  hipStream_t s = ctx->stream;
  const long long nblk = (n + NB - 1) / NB;
  fit->ctx = ctx;
  fit->device = ctx->device;
  fit->n = n;
  fit->lda = factor_ld(n);
  fit->A_bytes = sizeof(double) * (size_t)fit->lda * (size_t)n;
  if (ctx->pool_A && ctx->pool_A_bytes == fit->A_bytes) {
    fit->A = ctx->pool_A;
    ctx->pool_A = nullptr;
    ctx->pool_A_bytes = 0;
  } else {
    AGP_HIP_CHECK(ctx, hipMalloc(&fit->A, fit->A_bytes));
  }
  AGP_HIP_CHECK(ctx, hipMalloc(&fit->invd, sizeof(double) * (size_t)nblk * (36 * MB * MB)));
  const double *src = K;  // device-resident source of the triangle
  if (location == AGP_HOST) {
    // a pitched copy from pageable host memory degenerates into one small copy per
    // column; upload the matrix in one piece and re-pitch it on the device
    const size_t bytes = sizeof(double) * (size_t)ld * (size_t)n;
    int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, bytes);
    if (st != AGP_OK) return st;
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->ws_aux, K, bytes - sizeof(double) * (size_t)(ld - n), hipMemcpyHostToDevice, s));
    src = ctx->ws_aux;
  }
  if (uplo == 0)
    AGP_HIP_CHECK(ctx, hipMemcpy2DAsync(fit->A, sizeof(double) * (size_t)fit->lda, src, sizeof(double) * (size_t)ld,
                                        sizeof(double) * (size_t)n, (size_t)n, hipMemcpyDeviceToDevice, s));
  else
    launch_upper_to_lower(s, src, ld, fit->A, fit->lda, n);
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
  AGP_HIP_CHECK(ctx, hipMemsetAsync(ctx->d_scalars, 0, 4 * sizeof(double), s));
  launch_nan_scan_lower(s, fit->A, fit->lda, n, ctx->d_flags);
  factor_lower(ctx, fit->A, n, fit->lda, fit->invd, y, nullptr);
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  fit->failed_pivot = ctx->h_flags[1] ? (int64_t)ctx->h_flags[1] - 1 : -1;
  fit->log_det = 2. * ctx->h_scalars[0];
  return status_from_flags(ctx);
}

int agp_factor_create(agp_context *ctx, const double *K, int64_t n, int64_t ld, int uplo, int location,
                      agp_fit **out) {
  if (!ctx || !K || !out || n <= 0 || ld < n || (uplo != 0 && uplo != 1)) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  agp_fit *fit = new (std::nothrow) agp_fit();
  if (!fit) return AGP_ERR_INVALID_ARGUMENT;
  const int st = factor_dense(ctx, K, n, ld, uplo, location, fit, nullptr);
  if (st == AGP_ERR_HIP) { agp_fit_destroy(fit); return st; }
  *out = fit;  // on NOT_POSITIVE_DEFINITE / NAN the handle only carries the failed pivot
  return st;
}

int agp_nll_dense(agp_context *ctx, const double *deviation, const double *K, int64_t n, int64_t ld, int uplo,
                  int location, double *out) {
  if (!ctx || !deviation || !K || !out || n <= 0 || ld < n || (uplo != 0 && uplo != 1)) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (n == 1) {  // univariate shortcut, likelihood.hpp:57-60 -> -gaussian::log_pdf(deviation, variance)
    double d = 0., v = 0.;
    const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToHost : hipMemcpyDeviceToHost;
    AGP_HIP_CHECK(ctx, hipMemcpy(&d, deviation, sizeof(double), kind));
    AGP_HIP_CHECK(ctx, hipMemcpy(&v, K, sizeof(double), kind));
    *out = 0.5 * (std::log(2 * M_PI * v) + d * d / v);
    return AGP_OK;
  }
  agp_fit *fit = new (std::nothrow) agp_fit();
  if (!fit) return AGP_ERR_INVALID_ARGUMENT;
  double *z = nullptr;
  hipError_t e = hipMalloc(&z, sizeof(double) * (size_t)n);
  if (e != hipSuccess) { delete fit; ctx->last_error = hipGetErrorString(e); return AGP_ERR_HIP; }
  int st = vector_to_device(ctx, deviation, n, location, z);
  if (st == AGP_OK) st = factor_dense(ctx, K, n, ld, uplo, location, fit, z);
  if (st == AGP_OK) {
    launch_dot(ctx->stream, z, z, n, ctx->d_scalars + 1);  // dev^T K^-1 dev = z^T z
    e = hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); st = AGP_ERR_HIP; }
    else *out = 0.5 * (fit->log_det + ctx->h_scalars[1] + (double)n * std::log(2 * M_PI));
  }
  (void)hipFree(z);
  agp_fit_destroy(fit);
  return st;
}

// ---- leave-one-out fast path ---------------------------------------------------
// diag(K^-1) into ws_aux[ldr * n ...]; returns the device pointer of the n results
static int inverse_diagonal_device(agp_context *ctx, const agp_fit *fit, double **diag_out) {
  const long long n = fit->n, ldr = factor_ld(n);
  const size_t r_elems = (size_t)ldr * (size_t)n;
  int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (r_elems + 3 * (size_t)round_up(n, 2)));
  if (st != AGP_OK) return st;
  double *R = ctx->ws_aux, *diag = R + r_elems;
  hipStream_t s = ctx->stream;
  launch_set_identity(s, R, ldr, n);
  // R = L^-1 (serializable_ldlt.hpp:154-160), exploiting the triangular right-hand side
  forward_solve_mat(s, fit->A, n, fit->lda, fit->invd, R, n, ldr, /*rhs_lower=*/true);
  // (K^-1)_ii = || R[:, i] ||^2   (sub_matrix^T * sub_matrix, :171-172)
  launch_coldot(s, R, ldr, R, ldr, n, n, diag, -1.0, nullptr);
  AGP_HIP_CHECK(ctx, hipGetLastError());
  *diag_out = diag;
  return AGP_OK;
}

int agp_fit_inverse_diagonal(agp_context *ctx, const agp_fit *fit, double *out, int out_location) {
  if (!ctx || !fit || !out || !fit->A) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  double *diag = nullptr;
  int st = inverse_diagonal_device(ctx, fit, &diag);
  if (st != AGP_OK) return st;
  return copy_out(ctx, diag, fit->n, out, out_location);
}

int agp_loo_marginal(agp_context *ctx, const agp_fit *fit, const double *y, double *mean, double *variance,
                     int location) {
  if (!ctx || !fit || !y || !mean || !variance || !fit->A || !fit->alpha) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = fit->n;
  double *diag = nullptr;
  int st = inverse_diagonal_device(ctx, fit, &diag);
  if (st != AGP_OK) return st;
  double *yd = diag + round_up(n, 2), *md = yd + round_up(n, 2);  // mean overwrites... separate slots below
  if ((st = vector_to_device(ctx, y, n, location, yd)) != AGP_OK) return st;
  // variance is written over the diag slot's successor: reuse `diag` for the variance after reading it
  launch_loo(ctx->stream, diag, yd, fit->alpha, n, md, diag);
  if ((st = copy_out(ctx, md, n, mean, location)) != AGP_OK) return st;
  return copy_out(ctx, diag, n, variance, location);
}

// ---- leave-one-GROUP-out -------------------------------------------------------
// SerializableLDLT::inverse_blocks (serializable_ldlt.hpp:137-179) and held_out_predictions
// (cross_validation_utils.hpp:165-232).  R = L^-1 is built once (N^3/3 flop on MFMA, the solve
// kernels on a triangular right-hand side); per group the columns I_g are gathered and
// B_g = G^T G = (K^-1)[I_g, I_g] is one MFMA product; the |g| x |g| system is then factored with
// the same LL^T kernels.
namespace {

struct GroupWork {
  agp_context *ctx = nullptr;
  double *R = nullptr, *G = nullptr, *B = nullptr, *tmp = nullptr;
  long long *idx = nullptr;
  long long n = 0, ldr = 0, ldg = 0, ldb = 0, mmax = 0;
  ~GroupWork() {
    (void)hipFree(R); (void)hipFree(G); (void)hipFree(B); (void)hipFree(tmp); (void)hipFree(idx);
  }
};

int group_work_init(agp_context *ctx, const agp_fit *fit, int64_t n_groups, const int64_t *offsets,
                    const int64_t *indices, GroupWork *w) {
  const long long n = fit->n;
  if (n_groups < 0 || !offsets || offsets[0] != 0) return AGP_ERR_INVALID_ARGUMENT;
  long long mmax = 0;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long m = offsets[g + 1] - offsets[g];
    if (m < 0) return AGP_ERR_INVALID_ARGUMENT;
    if (m > mmax) mmax = m;
  }
  const long long total = offsets[n_groups];
  if (total > 0 && !indices) return AGP_ERR_INVALID_ARGUMENT;
  for (long long i = 0; i < total; ++i)
    if (indices[i] < 0 || indices[i] >= n) return AGP_ERR_INVALID_ARGUMENT;
  w->ctx = ctx; w->n = n; w->mmax = mmax;
  if (total == 0) return AGP_OK;
  w->ldr = factor_ld(n); w->ldg = round_up(n, 2); w->ldb = factor_ld(mmax);
  AGP_HIP_CHECK(ctx, hipMalloc(&w->R, sizeof(double) * (size_t)w->ldr * (size_t)n));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->G, sizeof(double) * (size_t)w->ldg * (size_t)mmax));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->B, sizeof(double) * (size_t)w->ldb * (size_t)mmax));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->tmp, sizeof(double) * (size_t)(4 * round_up(mmax, 2) + 2 * round_up(n, 2))));
  AGP_HIP_CHECK(ctx, hipMalloc(&w->idx, sizeof(long long) * (size_t)total));
  static_assert(sizeof(long long) == sizeof(int64_t), "index width");
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(w->idx, indices, sizeof(long long) * (size_t)total, hipMemcpyHostToDevice, ctx->stream));
  hipStream_t s = ctx->stream;
  launch_set_identity(s, w->R, w->ldr, n);
  forward_solve_mat(s, fit->A, n, fit->lda, fit->invd, w->R, n, w->ldr, /*rhs_lower=*/true);
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

// B (m x m, w->ldb) = (K^-1)[I_g, I_g] on the device
int group_inverse_block(GroupWork *w, const int64_t *indices, long long off, long long m) {
  agp_context *ctx = w->ctx;
  hipStream_t s = ctx->stream;
  long long row0 = w->n;
  for (long long a = 0; a < m; ++a)
    if (indices[off + a] < row0) row0 = indices[off + a];
  row0 &= ~1LL;  // column j of R is zero above row j: only rows >= min(I_g) contribute
  launch_gather_cols(s, w->R, w->ldr, w->idx + off, m, row0, w->n, w->G, w->ldg);
  AGP_HIP_CHECK(ctx, hipMemsetAsync(w->B, 0, sizeof(double) * (size_t)w->ldb * (size_t)m, s));
  // B -= G^T G (k-major operands), then negate
  launch_gemm_nt_sub(s, w->B, w->ldb, w->G + row0, w->ldg, true, w->G + row0, w->ldg, true, m, m, w->n - row0, false);
  launch_negate(s, w->B, w->ldb, m, nullptr);
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

}  // namespace

int agp_fit_inverse_blocks(agp_context *ctx, const agp_fit *fit, int64_t n_groups, const int64_t *offsets,
                           const int64_t *indices, double *blocks, int out_location) {
  if (!ctx || !fit || !fit->A || !blocks) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GroupWork w;
  int st = group_work_init(ctx, fit, n_groups, offsets, indices, &w);
  if (st != AGP_OK) return st;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long off = offsets[g], m = offsets[g + 1] - off;
    if (m == 0) continue;
    if ((st = group_inverse_block(&w, indices, off, m)) != AGP_OK) return st;
    if ((st = copy_out_2d(ctx, w.B, w.ldb, m, m, blocks, m, out_location)) != AGP_OK) return st;
    blocks += m * m;
  }
  return AGP_OK;
}

int agp_held_out_predictions(agp_context *ctx, const agp_fit *fit, const double *y, int64_t n_groups,
                             const int64_t *offsets, const int64_t *indices, double *mean, double *variance,
                             double *joint, int location) {
  if (!ctx || !fit || !fit->A || !fit->alpha || !y || !mean) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  GroupWork w;
  int st = group_work_init(ctx, fit, n_groups, offsets, indices, &w);
  if (st != AGP_OK) return st;
  if (w.mmax == 0) return AGP_OK;
  const long long n = fit->n, mp = round_up(w.mmax, 2);
  double *v = w.tmp, *x = v + mp, *mu = x + mp, *var = mu + mp, *yd = var + mp;
  if ((st = vector_to_device(ctx, y, n, location, yd)) != AGP_OK) return st;
  hipStream_t s = ctx->stream;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long off = offsets[g], m = offsets[g + 1] - off;
    if (m == 0) continue;
    if ((st = group_inverse_block(&w, indices, off, m)) != AGP_OK) return st;
    // A_ldlt = SerializableLDLT(inverse_block)   (cross_validation_utils.hpp:181)
    agp_fit *fb = nullptr;
    st = agp_factor_create(ctx, w.B, m, w.ldb, 0, AGP_DEVICE, &fb);
    if (st != AGP_OK) { if (fb) agp_fit_destroy(fb); return st; }
    // mean = y - A_ldlt.solve(v), v = subset(information, indices)   (:175,182)
    launch_gather_vec(s, fit->alpha, w.idx + off, m, nullptr, v);
    st = agp_solve(ctx, fb, v, 1, x, AGP_DEVICE);
    if (st == AGP_OK) {
      launch_gather_vec(s, yd, w.idx + off, m, x, mu);
      st = copy_out(ctx, mu, m, mean + off, location);
    }
    if (st == AGP_OK && (variance || joint)) {
      // R_B = L_B^-1 ; inverse = R_B^T R_B  (inverse_diagonal :183 / inverse :192)
      const long long ldq = factor_ld(m);
      st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * 2 * (size_t)ldq * (size_t)m);
      if (st == AGP_OK) {
        double *Q = ctx->ws_aux, *J = Q + (size_t)ldq * (size_t)m;
        launch_set_identity(s, Q, ldq, m);
        forward_solve_mat(s, fb->A, m, fb->lda, fb->invd, Q, m, ldq, /*rhs_lower=*/true);
        if (joint) {
          (void)hipMemsetAsync(J, 0, sizeof(double) * (size_t)ldq * (size_t)m, s);
          launch_gemm_nt_sub(s, J, ldq, Q, ldq, true, Q, ldq, true, m, m, m, false);
          launch_negate(s, J, ldq, m, var);
          st = copy_out_2d(ctx, J, ldq, m, m, joint, m, location);
          joint += m * m;
        } else {
          launch_coldot(s, Q, ldq, Q, ldq, m, m, var, -1.0, nullptr);
        }
        if (st == AGP_OK && variance) st = copy_out(ctx, var, m, variance + off, location);
      }
    }
    agp_fit_destroy(fb);
    if (st != AGP_OK) return st;
  }
  AGP_HIP_CHECK(ctx, hipGetLastError());
  return AGP_OK;
}

// ---- sparse Gaussian process (FITC / PITC) -----------------------------------------
// SparseGaussianProcessRegression (include/albatross/src/models/sparse_gp.hpp).  The reference
// stores Sigma = (K_uu + K_uf A^-1 K_fu)^-1 through the pivoted Householder QR of
// B = [A^-1/2 K_fu; K_uu^T/2] (:343-352); only R^T R = B^T B enters any result, so the device path
// forms  M = B^T B = K_uu + W W^T  (W = K_uf A^-T/2, one MFMA SYRK), factors it with the same
// LL^T kernels (M = L1 L1^T, i.e. R = L1^T up to the column permutation) and removes the squared
// LL^T kernels.  Forming B^T B squares the condition number, which the reference's QR avoids, so
// the factor is repaired the CholeskyQR2 way: Q1^T = L1^-1 B^T is formed explicitly (one more
// triangular solve over all n columns), Q1^T Q1 = I + O(eps cond) is factored again (L2), and
// B^T B = (L1 L2)(L1 L2)^T holds to working accuracy: log|R|, R^-T x and the information vector
// (plus two refinement steps against B itself) then agree with the QR-based reference algorithm to ~1e-9 even
// with cond(K_uu) ~ 1e7.  K_uu and every block of A use LL^T as well.
struct agp_sparse_fit {
  agp_context *ctx = nullptr;
  long long m = 0;
  DeviceFeatures u;            // train_features = inducing points
  agp_fit *kuu = nullptr;      // train_covariance = factor of K_uu + inducing_nugget I
  agp_fit *sigma = nullptr;    // L1: LL^T of M = B^T B as formed in floating point
  agp_fit *sigma2 = nullptr;   // L2: LL^T of Q1^T Q1, Q1 = B L1^-T (CholeskyQR2: B^T B = L1 L2 L2^T L1^T to working accuracy)
  double *v = nullptr;         // information (m)
  double nll = 0.;
};

namespace {

struct SparseScratch {
  double *Kuf = nullptr, *Pbuf = nullptr, *M0 = nullptr, *Ksym = nullptr, *vecs = nullptr, *partial = nullptr,
         *Ag = nullptr, *Pimg = nullptr, *Q1T = nullptr;
  std::vector<agp_fit *> blocks;
  ~SparseScratch() {
    (void)hipFree(Kuf); (void)hipFree(Pbuf); (void)hipFree(M0); (void)hipFree(Ksym); (void)hipFree(vecs);
    (void)hipFree(partial); (void)hipFree(Ag); (void)hipFree(Pimg); (void)hipFree(Q1T);
    for (agp_fit *b : blocks) agp_fit_destroy(b);
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

}  // namespace

void agp_sparse_fit_destroy(agp_sparse_fit *f) {
  if (!f) return;
  if (f->ctx) (void)hipSetDevice(f->ctx->device);
  f->u.release();
  if (f->kuu) agp_fit_destroy(f->kuu);
  if (f->sigma) agp_fit_destroy(f->sigma);
  if (f->sigma2) agp_fit_destroy(f->sigma2);
  if (f->v) (void)hipFree(f->v);
  delete f;
}

int64_t agp_sparse_fit_size(const agp_sparse_fit *f) { return f ? f->m : 0; }

int agp_sparse_fit_create(agp_context *ctx, const agp_kernel *k, const agp_features *x, int64_t n_groups,
                          const int64_t *offsets, const double *y, const double *y_var, const agp_features *u,
                          double measurement_nugget, double inducing_nugget, agp_sparse_fit **out,
                          double *information, double *nll_out) {
  if (!ctx || !k || !x || !u || !y || !offsets || n_groups <= 0) return AGP_ERR_INVALID_ARGUMENT;
  if (out) *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st == AGP_OK) st = validate_features(u);
  if (st != AGP_OK) return st;
  const long long n = x->n, m = u->n;
  if (n <= 0 || m <= 0 || u->dim != x->dim || offsets[0] != 0 || offsets[n_groups] != n) return AGP_ERR_INVALID_ARGUMENT;
  long long smax = 0;
  for (int64_t g = 0; g < n_groups; ++g) {
    const long long sg = offsets[g + 1] - offsets[g];
    if (sg <= 0) return AGP_ERR_INVALID_ARGUMENT;
    if (sg > smax) smax = sg;
  }
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  hipStream_t s = ctx->stream;

  agp_sparse_fit *f = new (std::nothrow) agp_sparse_fit();
  if (!f) return AGP_ERR_INVALID_ARGUMENT;
  f->ctx = ctx; f->m = m;
  SparseScratch w;
  // AGP_SPARSE_TIMING=1: wall time of every stage (with a stream synchronisation at each boundary) on stderr
  static const bool timing = getenv("AGP_SPARSE_TIMING") != nullptr;
  auto t_last = std::chrono::steady_clock::now();
  auto stage = [&](const char *name) {
    if (!timing) return;
    (void)hipStreamSynchronize(s);
    const auto now = std::chrono::steady_clock::now();
    fprintf(stderr, "  [sparse fit] %-28s %8.2f ms\n", name, std::chrono::duration<double, std::milli>(now - t_last).count());
    t_last = now;
  };
  DeviceFeatures dx;
#define SP_FAIL(code) do { dx.release(); agp_sparse_fit_destroy(f); return (code); } while (0)
#define SP_HIP(expr)                                                                     \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      SP_FAIL(AGP_ERR_HIP);                                                              \
    }                                                                                    \
  } while (0)
  if ((st = to_device(ctx, u, true, &f->u)) != AGP_OK) SP_FAIL(st);
  if ((st = to_device(ctx, x, false, &dx)) != AGP_OK) SP_FAIL(st);
  FeatView xm = dx.v;
  xm.meas = 1;  // as_measurements(out_of_order_features), sparse_gp.hpp:649-650

  const long long ldm = factor_ld(m), ldk = round_up(m, 2), np2 = round_up(n, 2), mp2 = round_up(m, 2);
  const long long chunks = (n + 1023) / 1024;
  // vectors: dvar (n) | yw (n) | t (n) | nug (m) | b (m) | v (m) | r (m) | dv (m)
  SP_HIP(hipMalloc(&w.vecs, sizeof(double) * (size_t)(3 * np2 + 5 * mp2)));
  double *dvar = w.vecs, *yw = dvar + np2, *tvec = yw + np2, *nug = tvec + np2, *bvec = nug + mp2, *vvec = bvec + mp2,
         *rvec = vvec + mp2, *dv = rvec + mp2;
  SP_HIP(hipMalloc(&w.partial, sizeof(double) * (size_t)(chunks > 0 ? chunks : 1) * (size_t)m));
  const hipMemcpyKind kind = x->location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  SP_HIP(hipMemcpyAsync(yw, y, sizeof(double) * (size_t)n, kind, s));
  if (y_var) {
    SP_HIP(hipMemcpyAsync(tvec, y_var, sizeof(double) * (size_t)n, kind, s));
    if (x->location == AGP_HOST) SP_HIP(hipStreamSynchronize(s));
    launch_axpby(s, n, 1.0, tvec, measurement_nugget, nullptr, dvar);   // target variance + measurement nugget, :692-696
  } else {
    if (x->location == AGP_HOST) SP_HIP(hipStreamSynchronize(s));
    launch_axpby(s, n, 0.0, nullptr, measurement_nugget, nullptr, dvar);
  }
  launch_axpby(s, m, 0.0, nullptr, inducing_nugget, nullptr, nug);

  stage("upload");
  // K_uu + inducing_nugget I  (:674-679) -> LL^T
  SP_HIP(hipMalloc(&w.M0, sizeof(double) * (size_t)ldm * (size_t)m));
  SP_HIP(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
  launch_gram(s, dprog, f->u.v, f->u.v, true, true, w.M0, ldm, nug, ctx->d_flags, &k->prog);
  st = agp_factor_create(ctx, w.M0, m, ldm, 0, AGP_DEVICE, &f->kuu);
  if (st != AGP_OK) SP_FAIL(st);

  stage("K_uu + factor");
  // K_uf (m x n) and P = K_uu^-1/2 K_uf = L_u^-1 K_uf  (:669-685)
  SP_HIP(hipMalloc(&w.Kuf, sizeof(double) * (size_t)ldk * (size_t)n));
  SP_HIP(hipMalloc(&w.Pbuf, sizeof(double) * (size_t)ldk * (size_t)n));
  launch_gram(s, dprog, f->u.v, xm, false, false, w.Kuf, ldk, nullptr, nullptr, &k->prog);
  SP_HIP(hipMemcpyAsync(w.Pbuf, w.Kuf, sizeof(double) * (size_t)ldk * (size_t)n, hipMemcpyDeviceToDevice, s));
  forward_solve_mat(s, f->kuu->A, m, f->kuu->lda, f->kuu->invd, w.Pbuf, n, ldk);

  // A = K_ff (block diagonal) + target variance - diag blocks of P^T P + measurement nugget, block LL^T
  // (:652-704); then W = K_uf A^-T/2 (in place in K_uf) and y_w = A^-1/2 y, block by block (B's top block
  // transposed, :347-349; :372).
  SP_HIP(hipStreamSynchronize(s));
  stage("K_uf, P = L_u^-1 K_uf");
  bool uniform = true;
  for (int64_t g = 0; g < n_groups; ++g) uniform = uniform && (offsets[g + 1] - offsets[g] == smax);
  double log_det_a = 0.;
  if (uniform) {
    // All groups have the same size: the blocks advance in LOCK STEP through batched launches
    // (blockIdx.y = group) - a dozen launches for the whole of A instead of ~40 per block, which
    // is what the per-block path below is bound by (the HIP launch path is serial per process).
    const long long sb = smax, lda_b = factor_ld(sb), nblk_b = (sb + NB - 1) / NB;
    const long long stride_A = lda_b * sb, stride_I = nblk_b * (36 * MB * MB);
    SP_HIP(hipMalloc(&w.Ag, sizeof(double) * (size_t)stride_A * (size_t)n_groups));
    SP_HIP(hipMalloc(&w.Pimg, sizeof(double) * ((size_t)stride_I + 1) * (size_t)n_groups));
    double *logsum = w.Pimg + (size_t)stride_I * (size_t)n_groups;
    SP_HIP(hipMemsetAsync(logsum, 0, sizeof(double) * (size_t)n_groups, s));
    SP_HIP(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
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
    SP_HIP(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    SP_HIP(hipMemcpyAsync(hl.data(), logsum, sizeof(double) * (size_t)n_groups, hipMemcpyDeviceToHost, s));
    SP_HIP(hipStreamSynchronize(s));
    SP_HIP(hipGetLastError());
    if ((st = status_from_flags(ctx)) != AGP_OK) SP_FAIL(st);
    for (int64_t g = 0; g < n_groups; ++g) log_det_a += 2. * hl[(size_t)g];  // fixed order
  } else {
    // ragged groups: one block at a time, T host threads on T helper contexts (own streams)
    agp_context_impl *ci = static_cast<agp_context_impl *>(ctx);
    static int want_threads = -1;
    if (want_threads < 0) {
      const char *e = getenv("AGP_SPARSE_THREADS");
      want_threads = e ? atoi(e) : 4;  // measured: 4 threads 590 ms, 16: 650 ms, 32: 740 ms at 512 blocks of 512
      if (want_threads < 1) want_threads = 1;
    }
    const int T = (int)std::min<long long>(want_threads, n_groups);
    while ((int)ci->helpers.size() < T) {
      agp_context *h = nullptr;
      if ((st = agp_context_create(ctx->device, &h)) != AGP_OK) SP_FAIL(st);
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
      if (hipMalloc(&Ag, sizeof(double) * (size_t)lda_g * (size_t)smax) != hipSuccess) { stt = AGP_ERR_HIP; return; }
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
      (void)hipFree(Ag);
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < T; ++t) pool.emplace_back(worker, t);
    worker(0);
    for (auto &th : pool) th.join();
    for (int t = 0; t < T; ++t)
      if (status[(size_t)t] != AGP_OK) {
        ctx->last_error = errors[(size_t)t];
        SP_FAIL(status[(size_t)t]);
      }
    for (int64_t g = 0; g < n_groups; ++g) log_det_a += w.blocks[(size_t)g]->log_det;  // fixed order
  }
  (void)hipFree(w.Pbuf); w.Pbuf = nullptr;
  stage("blocks of A, W, y_w");
  double *W = w.Kuf;

  // M = B^T B = (K_uu + nugget I) + W W^T ; keep a symmetric copy of K_uu' for the refinement
  SP_HIP(hipMalloc(&w.Ksym, sizeof(double) * (size_t)ldm * (size_t)m));
  SP_HIP(hipMemcpyAsync(w.Ksym, w.M0, sizeof(double) * (size_t)ldm * (size_t)m, hipMemcpyDeviceToDevice, s));
  launch_symmetrize(s, w.Ksym, ldm, m);
  launch_negate(s, w.M0, ldm, m, nullptr);
  launch_gemm_nt_sub(s, w.M0, ldm, W, ldk, false, W, ldk, false, m, m, n, true);
  launch_negate(s, w.M0, ldm, m, nullptr);
  stage("M = K_uu + W W^T");
  st = agp_factor_create(ctx, w.M0, m, ldm, 0, AGP_DEVICE, &f->sigma);
  if (st != AGP_OK) SP_FAIL(st);
  stage("factor M");

  // CholeskyQR2: Q1^T = L1^-1 [W | L_u]  (m x (n + m)), G = Q1^T Q1 = L2 L2^T
  {
    double *Q1T = nullptr;
    SP_HIP(hipMalloc(&Q1T, sizeof(double) * (size_t)ldk * (size_t)(n + m)));
    w.Q1T = Q1T;
    SP_HIP(hipMemcpyAsync(Q1T, W, sizeof(double) * (size_t)ldk * (size_t)n, hipMemcpyDeviceToDevice, s));
    SP_HIP(hipMemcpy2DAsync(Q1T + (size_t)ldk * (size_t)n, sizeof(double) * (size_t)ldk, f->kuu->A,
                            sizeof(double) * (size_t)f->kuu->lda, sizeof(double) * (size_t)m, (size_t)m,
                            hipMemcpyDeviceToDevice, s));
    launch_zero_upper(s, Q1T + (size_t)ldk * (size_t)n, ldk, m);  // K_uu^T/2 = L_u^T: its transpose L_u, lower
    forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, Q1T, n + m, ldk);
    SP_HIP(hipMemsetAsync(w.M0, 0, sizeof(double) * (size_t)ldm * (size_t)m, s));
    launch_gemm_nt_sub(s, w.M0, ldm, Q1T, ldk, false, Q1T, ldk, false, m, m, n + m, true);
    launch_negate(s, w.M0, ldm, m, nullptr);
    st = agp_factor_create(ctx, w.M0, m, ldm, 0, AGP_DEVICE, &f->sigma2);
    (void)hipFree(Q1T); w.Q1T = nullptr;
    if (st != AGP_OK) SP_FAIL(st);
  }
  stage("CholeskyQR2 (Q1, L2)");

  // x <- (B^T B)^-1 x = L1^-T (G^-1 (L1^-1 x))
  auto sigma_solve = [&](double *xv) -> int {
    forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, xv, 1, m);
    const int e = agp_solve(ctx, f->sigma2, xv, 1, xv, AGP_DEVICE);
    if (e != AGP_OK) return e;
    backward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, xv, 1, m);
    return AGP_OK;
  };
  // information v = (B^T B)^-1 B^T [y_w; 0]  (:370-373), then two refinement steps against B:
  //   r = W (y_w - W^T v) - K_uu' v ,  v += (B^T B)^-1 r
  launch_matvec(s, W, ldk, m, n, yw, w.partial, 1.0, 0.0, nullptr, bvec);
  SP_HIP(hipMemcpyAsync(vvec, bvec, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s));
  if ((st = sigma_solve(vvec)) != AGP_OK) SP_FAIL(st);
  for (int it = 0; it < 2; ++it) {
    launch_colvec_dot(s, W, ldk, m, n, vvec, -1.0, 1.0, yw, tvec);                  // t = y_w - W^T v
    launch_matvec(s, w.Ksym, ldm, m, m, vvec, w.partial, 1.0, 0.0, nullptr, rvec);  // K_uu' v
    launch_matvec(s, W, ldk, m, n, tvec, w.partial, 1.0, -1.0, rvec, dv);           // r = W t - K_uu' v
    if ((st = sigma_solve(dv)) != AGP_OK) SP_FAIL(st);
    launch_axpby(s, m, 1.0, vvec, 1.0, dv, vvec);
  }
  stage("information + refinement");
  SP_HIP(hipMalloc(&f->v, sizeof(double) * (size_t)m));
  SP_HIP(hipMemcpyAsync(f->v, vvec, sizeof(double) * (size_t)m, hipMemcpyDeviceToDevice, s));

  // negative log likelihood (:524-596): log|K| = log|A| + log|B^T B| - log|K_uu'| ,
  // y^T K^-1 y = y_w^T y_w - || L1^-1 W y_w ||^2
  forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, bvec, 1, m);
  forward_solve_mat(s, f->sigma2->A, m, f->sigma2->lda, f->sigma2->invd, bvec, 1, m);  // y_b = L2^-1 L1^-1 W y_w
  launch_dot(s, yw, yw, n, ctx->d_scalars + 1);
  launch_dot(s, bvec, bvec, m, ctx->d_scalars + 2);
  SP_HIP(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
  if (information) SP_HIP(hipMemcpyAsync(information, f->v, sizeof(double) * (size_t)m, hipMemcpyDeviceToHost, s));
  SP_HIP(hipStreamSynchronize(s));
  SP_HIP(hipGetLastError());
  const double log_det = log_det_a + (f->sigma->log_det + f->sigma2->log_det) - f->kuu->log_det;
  f->nll = 0.5 * (log_det + (ctx->h_scalars[1] - ctx->h_scalars[2]) + (double)n * std::log(2 * M_PI));
  if (nll_out) *nll_out = f->nll;
  dx.release();
  if (out) *out = f;
  else agp_sparse_fit_destroy(f);
#undef SP_HIP
#undef SP_FAIL
  return AGP_OK;
}

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
  if (xs->dim != f->u.v.dim) return AGP_ERR_INVALID_ARGUMENT;
  const long long M = xs->n, m = f->m;
  if (M == 0) return AGP_OK;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dxs;
  if ((st = to_device(ctx, xs, false, &dxs)) != AGP_OK) return st;
  const long long ldq = round_up(m, 2), ldc = round_up(M, 2);
  const size_t q_elems = (size_t)ldq * (size_t)M;
  const size_t p_elems = mode == 2 ? (size_t)ldc * (size_t)M : (size_t)ldc;
  st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (2 * q_elems + (size_t)ldc + p_elems));
  if (st != AGP_OK) { dxs.release(); return st; }
  double *Q = ctx->ws_aux, *S = Q + q_elems, *mean_d = S + q_elems, *prior = mean_d + ldc;
  hipStream_t s = ctx->stream;
  launch_predict_mean(s, dprog, f->u.v, dxs.v, f->v, mean_d, &k->prog);
  if (mode > 0) {
    launch_gram(s, dprog, f->u.v, dxs.v, false, false, Q, ldq, nullptr, nullptr, &k->prog);
    (void)hipMemcpyAsync(S, Q, sizeof(double) * q_elems, hipMemcpyDeviceToDevice, s);
    forward_solve_mat(s, f->kuu->A, m, f->kuu->lda, f->kuu->invd, Q, M, ldq);
    forward_solve_mat(s, f->sigma->A, m, f->sigma->lda, f->sigma->invd, S, M, ldq);
    forward_solve_mat(s, f->sigma2->A, m, f->sigma2->lda, f->sigma2->invd, S, M, ldq);
  }
  if (mode == 1) {
    launch_gram_diagonal(s, dprog, dxs.v, prior);
    launch_coldot(s, Q, ldq, Q, ldq, m, M, prior, 1.0, prior);   // - Q_diag
    launch_coldot(s, S, ldq, S, ldq, m, M, prior, -1.0, prior);  // + S_diag
  } else if (mode == 2) {
    launch_gram(s, dprog, dxs.v, dxs.v, true, false, prior, ldc, nullptr, nullptr, &k->prog);
    launch_gemm_nt_sub(s, prior, ldc, Q, ldq, true, Q, ldq, true, M, M, m, true);  // - max_explained
    launch_axpby(s, (long long)q_elems, -1.0, S, 0.0, nullptr, Q);  // Q <- -S
    launch_gemm_nt_sub(s, prior, ldc, Q, ldq, true, S, ldq, true, M, M, m, true);  // + unexplained
    launch_symmetrize(s, prior, ldc, M);
  }
  st = copy_out(ctx, mean_d, M, mean, out_location);
  if (st == AGP_OK && mode == 1) st = copy_out(ctx, prior, M, var_or_cov, out_location);
  if (st == AGP_OK && mode == 2) st = copy_out_2d(ctx, prior, ldc, M, M, var_or_cov, M, out_location);
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

// ---- predict ---------------------------------------------------------------
int agp_predict_mean(agp_context *ctx, const agp_kernel *k, const agp_fit *fit, const agp_features *xs,
                     double *mean, int out_location) {
  if (!ctx || !k || !fit || !xs || !mean) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(xs);
  if (st != AGP_OK) return st;
  if (!fit->alpha || xs->dim != fit->train.v.dim) return AGP_ERR_INVALID_ARGUMENT;
  const long long m = xs->n;
  if (m == 0) return AGP_OK;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dxs;
  if ((st = to_device(ctx, xs, false, &dxs)) != AGP_OK) return st;
  st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)m);
  if (st == AGP_OK) {
    // cross_cov = cov(train_features, features); mean = cross_cov^T information  (gp.hpp:361-363)
    launch_predict_mean(ctx->stream, dprog, fit->train.v, dxs.v, fit->alpha, ctx->ws_aux, &k->prog);
    st = copy_out(ctx, ctx->ws_aux, m, mean, out_location);
  }
  dxs.release();
  return st;
}

static int predict_common(agp_context *ctx, const agp_kernel *k, const agp_fit *fit, const agp_features *xs,
                          double *mean, double *var_or_cov, bool joint, int out_location) {
  if (!ctx || !k || !fit || !xs || !mean || !var_or_cov) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(xs);
  if (st != AGP_OK) return st;
  if (!fit->alpha || xs->dim != fit->train.v.dim) return AGP_ERR_INVALID_ARGUMENT;
  const long long m = xs->n, n = fit->n;
  if (m == 0) return AGP_OK;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dxs;
  if ((st = to_device(ctx, xs, false, &dxs)) != AGP_OK) return st;
  const long long ldv = round_up(n, 2), ldc = round_up(m, 2);
  // workspace: V (n x m) | mean (m) | prior (m or m x m)
  const size_t v_elems = (size_t)ldv * (size_t)m;
  const size_t p_elems = joint ? (size_t)ldc * (size_t)m : (size_t)round_up(m, 2);
  st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (v_elems + (size_t)round_up(m, 2) + p_elems));
  if (st != AGP_OK) { dxs.release(); return st; }
  double *V = ctx->ws_aux, *mean_d = V + v_elems, *prior = mean_d + round_up(m, 2);
  hipStream_t s = ctx->stream;
  // mean (gp.hpp:82-85)
  launch_predict_mean(s, dprog, fit->train.v, dxs.v, fit->alpha, mean_d, &k->prog);
  // cross_cov = cov(train_features, features)   (gp.hpp:316,337)
  launch_gram(s, dprog, fit->train.v, dxs.v, false, false, V, ldv, nullptr, nullptr, &k->prog);
  // V = L^-1 K*  ;  explained = V^T V  (== K*^T K^-1 K*, gp.hpp:96,111)
  forward_solve_mat(s, fit->A, n, fit->lda, fit->invd, V, m, ldv);
  if (!joint) {
    launch_gram_diagonal(s, dprog, dxs.v, prior);                   // gp.hpp:339-343
    launch_coldot(s, V, ldv, V, ldv, n, m, prior, 1.0, prior);      // gp.hpp:97-99
    st = copy_out(ctx, mean_d, m, mean, out_location);
    if (st == AGP_OK) st = copy_out(ctx, prior, m, var_or_cov, out_location);
  } else {
    launch_gram(s, dprog, dxs.v, dxs.v, true, false, prior, ldc, nullptr, nullptr, &k->prog);  // prior_cov, gp.hpp:317
    launch_gemm_nt_sub(s, prior, ldc, V, ldv, true, V, ldv, true, m, m, n, true);   // lower tiles
    launch_symmetrize(s, prior, ldc, m);
    st = copy_out(ctx, mean_d, m, mean, out_location);
    if (st == AGP_OK) st = copy_out_2d(ctx, prior, ldc, m, m, var_or_cov, m, out_location);
  }
  dxs.release();
  return st;
}

int agp_predict_marginal(agp_context *ctx, const agp_kernel *k, const agp_fit *fit, const agp_features *xs,
                         double *mean, double *variance, int out_location) {
  return predict_common(ctx, k, fit, xs, mean, variance, false, out_location);
}

int agp_predict_joint(agp_context *ctx, const agp_kernel *k, const agp_fit *fit, const agp_features *xs,
                      double *mean, double *cov, int out_location) {
  return predict_common(ctx, k, fit, xs, mean, cov, true, out_location);
}

}  // extern "C"
