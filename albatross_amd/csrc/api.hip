// api.hip — the C-ABI of include/albatross_amd.h.
//
// Host-side orchestration only: uploads POD feature vectors, enqueues the HIP
// kernels of gram.hip / chol.hip / gemm.hip / reduce.hip on the context's
// stream, reads back the small results.  No CPU arithmetic fallback exists.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <chrono>
#include <cstdio>
#include <new>
#include <thread>

#include "api_internal.h"
#include <deque>
#include <mutex>
#include <tuple>
#include <unordered_map>
#include "trace.h"
#include "pub.h"

namespace agp {

void DeviceFeatures::release() {
  for (void *&p : owned) {
    if (p) (void)hipFree(p);
    p = nullptr;
  }
  v = FeatView{};
}

long long round_up(long long x, long long m) { return (x + m - 1) / m * m; }

// leading dimension of the factor: even (16-B aligned columns) and not a
// multiple of 256 doubles, so that consecutive columns do not alias the same
// HBM channel / L2 set pattern
long long factor_ld(long long n) {
  long long ld = round_up(n, 8);
  if (ld % 256 == 0) ld += 8;
  return ld;
}

}  // namespace agp

using namespace agp;

static agp_context_ext *ext_of(agp_context *ctx) { return &static_cast<agp_context_impl *>(ctx)->ext; }
std::atomic<unsigned long long> g_kernel_uid{1};

namespace agp {
namespace {
constexpr size_t DEV_CACHE_BYTES = 8ull << 30;
struct DevCache {
  std::mutex mu;
  std::unordered_map<void *, std::pair<int, size_t>> live;   // blocks handed out: device, size
  std::deque<std::tuple<int, size_t, void *>> parked;        // oldest first
  size_t held = 0;
};
DevCache &dev_cache() {
  // never destroyed: contexts held in a host program's statics are closed after this library's statics would be gone
  static DevCache *c = new DevCache();
  return *c;
}
}  // namespace

hipError_t dev_malloc_bytes(void **p, size_t bytes) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  DevCache &c = dev_cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto it = c.parked.begin(); it != c.parked.end(); ++it)
      if (std::get<0>(*it) == dev && std::get<1>(*it) == bytes) {
        *p = std::get<2>(*it);
        c.held -= bytes;
        c.parked.erase(it);
        c.live[*p] = {dev, bytes};
        return hipSuccess;
      }
  }
  const hipError_t e = hipMalloc(p, bytes);
  if (e == hipSuccess) {
    std::lock_guard<std::mutex> lock(c.mu);
    c.live[*p] = {dev, bytes};
  } else {
    dev_cache_trim();  // out of memory with blocks parked: give them back and try once more
    const hipError_t e2 = hipMalloc(p, bytes);
    if (e2 == hipSuccess) {
      std::lock_guard<std::mutex> lock(c.mu);
      c.live[*p] = {dev, bytes};
    }
    return e2;
  }
  return e;
}

hipError_t dev_free(void *p) {
  if (!p) return hipSuccess;
  DevCache &c = dev_cache();
  std::pair<int, size_t> info{-1, 0};
  {
    std::lock_guard<std::mutex> lock(c.mu);
    auto it = c.live.find(p);
    if (it != c.live.end()) { info = it->second; c.live.erase(it); }
  }
  if (info.first < 0 || info.second > DEV_CACHE_BYTES / 4) return hipFree(p);
  (void)hipDeviceSynchronize();  // what hipFree implies: nothing in flight uses the block any more
  std::vector<void *> evict;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    c.parked.emplace_back(info.first, info.second, p);
    c.held += info.second;
    while (c.held > DEV_CACHE_BYTES || c.parked.size() > 96) {
      evict.push_back(std::get<2>(c.parked.front()));
      c.held -= std::get<1>(c.parked.front());
      c.parked.pop_front();
    }
  }
  for (void *q : evict) (void)hipFree(q);
  return hipSuccess;
}

// hipFree of a block that MAY have come from dev_malloc (a factor parked in a context pool, a buffer handed from one
// owner to another): the `live` entry goes with it - a stale entry would later match an unrelated hipMalloc that
// happens to return the same address, and dev_free would park that block under the old (larger) size.
hipError_t dev_release(void *p) {
  if (!p) return hipSuccess;
  DevCache &c = dev_cache();
  {
    std::lock_guard<std::mutex> lock(c.mu);
    c.live.erase(p);
  }
  return hipFree(p);
}

void dev_cache_trim() {
  DevCache &c = dev_cache();
  std::vector<void *> all;
  {
    std::lock_guard<std::mutex> lock(c.mu);
    for (auto &t : c.parked) all.push_back(std::get<2>(t));
    c.parked.clear();
    c.held = 0;
  }
  for (void *q : all) (void)hipFree(q);
}
}  // namespace agp

namespace agp {
// the Gram launchers have no context: they follow the switches of the context created last
static bool g_gram_sop = true;
bool gram_sop_enabled() { return g_gram_sop; }
}  // namespace agp

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

static constexpr long long BACKSUB_COOP_MAX_N = 2047;
// the switches of include/albatross_amd.h ("switches"): read here, once per context, and nowhere else
static agp_context::Tuning read_tuning() {
  agp_context::Tuning t;
  auto flag = [](const char *name, bool dflt) {
    const char *e = getenv(name);
    return e && e[0] ? e[0] == '1' : dflt;
  };
  auto number = [](const char *name, long long dflt) {
    const char *e = getenv(name);
    return e && e[0] ? atoll(e) : dflt;
  };
  t.panel_fused = flag("AGP_PANEL_FUSED", true);
  t.step_below = number("AGP_STEP_BELOW", 4608);
  t.gram_sop = flag("AGP_GRAM_SOP", true);
  t.backsub_coop = flag("AGP_BACKSUB_COOP", true);
  t.backsub_coop_max = number("AGP_BACKSUB_COOP_MAX", BACKSUB_COOP_MAX_N);
  t.mixed_bf16 = flag("AGP_MIXED_BF16", true);
  t.mixed_f16 = flag("AGP_MIXED_F16", true);
  set_f16x2_kernel((int)number("AGP_F16X2_LDS_PAD", 8192), (int)number("AGP_F16X2_TERMS", 4), (int)number("AGP_F16X2_CHUNK", 32));
  t.mixed_nbo = number("AGP_MIXED_NBO", 512);
  t.fp64_nbo = number("AGP_FP64_NBO", 0);
  set_bf16x3_kernel((int)number("AGP_BF16X3_KERNEL", 2), (int)number("AGP_BF16X3_LDS_PAD", 8192));
  t.sparse_pivoted = flag("AGP_SPARSE_PIVOTED", false);
  t.predict_chunk = number("AGP_PREDICT_CHUNK", 0);
  t.shard_block = number("AGP_SHARD_BLOCK", 0);
  t.shard_force_comm = flag("AGP_SHARD_FORCE_COMM", false);
  t.shard_host_pacing = flag("AGP_SHARD_HOST_PACING", false);
  if (const char *e = getenv("AGP_SHARD_MASK_GFLOP")) t.shard_mask_gflop = atof(e);
  t.merge_above = number("AGP_MERGE_ABOVE", 8704);
  return t;
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
  ctx->tune = read_tuning();
  agp::g_gram_sop = ctx->tune.gram_sop;
  AGP_HIP_CHECK(ctx, hipSetDevice(device_id));
  {
    // main / panel stream at the highest priority: its short kernels must not
    // queue behind the bulk update running on stream2
    int lo = 0, hi = 0;
    AGP_HIP_CHECK(ctx, hipDeviceGetStreamPriorityRange(&lo, &hi));
    AGP_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->stream, hipStreamNonBlocking, hi));
    AGP_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->stream2, hipStreamNonBlocking, lo));
    // (no further streams in the context: one more high-priority stream changed how the runtime maps streams onto its few
    // hardware queues and cost 3.5 ms per N = 16384 fit before it was used at all)
    AGP_HIP_CHECK(ctx, hipStreamCreateWithPriority(&ctx->stream3, hipStreamNonBlocking, lo));
    AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_inv, hipEventDisableTiming));
    AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_c, hipEventDisableTiming));
    // CU mask of the end-phase bulk stream: bit i = CU i, and CU i sits on XCD i % 8 (measured with
    // scripts/probe_cumask.py: dropping the LAST indices keeps the XCDs balanced, dropping i % 32 >= 28 does not).
    // AGP_MASK_CUS = CUs the bulk stream keeps (multiple of 8; 0 = no masked stream)
    int cus = 256;
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device_id);
    ctx->cus = cus;
    // 224 of 256 CUs (28 per XCD) from 8704 remaining rows on is the best pair at N = 16384 (profiles/r02/sweep_mask.txt)
    const int keep = cus / 8 * 7 / 8 * 8;
    if (keep > 0 && keep < cus) {
      uint32_t mask[16] = {0};
      for (int i = 0; i < keep && i < 512; ++i) mask[i / 32] |= 1u << (i % 32);
      if (hipExtStreamCreateWithCUMask(&ctx->stream_masked, (uint32_t)((cus + 31) / 32), mask) != hipSuccess) {
        (void)hipGetLastError();
        ctx->stream_masked = nullptr;
      }
    }
  }
  AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_a, hipEventDisableTiming));
  AGP_HIP_CHECK(ctx, hipEventCreateWithFlags(&ctx->ev_b, hipEventDisableTiming));
  // flags and scalars of a factorisation: ONE block on either side ([4 ints | 4 doubles]) so that one 48-byte copy brings both back
  AGP_HIP_CHECK(ctx, hipMalloc(&ctx->d_headcnt, sizeof(unsigned long long) * agp_context::HEADCNT_WORDS));
  AGP_HIP_CHECK(ctx, hipMalloc(&ctx->d_flags, 4 * sizeof(int) + 4 * sizeof(double)));
  ctx->d_scalars = reinterpret_cast<double *>(ctx->d_flags + 4);
  AGP_HIP_CHECK(ctx, hipHostMalloc(&ctx->h_flags, 4 * sizeof(int) + 4 * sizeof(double)));
  ctx->h_scalars = reinterpret_cast<double *>(ctx->h_flags + 4);
  if (hipHostGetDevicePointer(&ctx->h_status_dev, ctx->h_flags, 0) != hipSuccess) { (void)hipGetLastError(); ctx->h_status_dev = nullptr; }
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
  if (ctx->ws_A) (void)agp::dev_release(ctx->ws_A);
  if (ctx->pool_A) (void)agp::dev_release(ctx->pool_A);
  if (ctx->pool_K) (void)agp::dev_release(ctx->pool_K);
  if (ctx->ws_refine) (void)hipFree(ctx->ws_refine);
  if (ctx->p32) (void)hipFree(ctx->p32);
  if (ctx->f16_scales) (void)hipFree(ctx->f16_scales);
  if (ctx->pool_L32) (void)hipFree(ctx->pool_L32);
  if (ctx->pool_aux) (void)agp::dev_release(ctx->pool_aux);
  if (ctx->pool_shard) (void)agp::dev_release(ctx->pool_shard);
  if (ctx->pool_sparse) (void)agp::dev_release(ctx->pool_sparse);
  if (ctx->pool_batch) (void)agp::dev_release(ctx->pool_batch);
  if (ctx->ws_aux) (void)hipFree(ctx->ws_aux);
  if (ctx->d_zpub) (void)hipFree(ctx->d_zpub);
  if (ctx->d_dpub) (void)hipFree(ctx->d_dpub);
  if (ctx->shard_flags) (void)hipFree(ctx->shard_flags);
  if (ctx->d_headcnt) (void)hipFree(ctx->d_headcnt);
  if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
  if (ctx->d_flags) (void)hipFree(ctx->d_flags);  // (d_scalars / h_scalars are the tails of these blocks)
  if (ctx->h_flags) (void)hipHostFree(ctx->h_flags);
  if (ctx->ev_a) (void)hipEventDestroy(ctx->ev_a);
  if (ctx->ev_b) (void)hipEventDestroy(ctx->ev_b);
  if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
  if (ctx->stream2) (void)hipStreamDestroy(ctx->stream2);
  if (ctx->stream3) (void)hipStreamDestroy(ctx->stream3);
  if (ctx->ev_inv) (void)hipEventDestroy(ctx->ev_inv);
  if (ctx->stream_masked) (void)hipStreamDestroy(ctx->stream_masked);
  if (ctx->stream_comm) (void)hipStreamDestroy(ctx->stream_comm);
  if (ctx->ev_c) (void)hipEventDestroy(ctx->ev_c);
  delete ctx;
  agp::dev_cache_trim();  // (the parked blocks of dev_free: a process that closes its contexts gives its device memory back)
}

int agp_context_synchronize(agp_context *ctx) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  AGP_HIP_CHECK(ctx, hipDeviceSynchronize());
  return AGP_OK;
}

// device buffers for AGP_DEVICE arguments (include/albatross_amd.h, "device memory"): plain runtime calls on the context's
// device, so that a host program needs neither HIP headers nor a second runtime in its process
int agp_device_malloc(agp_context *ctx, int64_t bytes, void **out) {
  if (!ctx || !out || bytes <= 0) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  AGP_HIP_CHECK(ctx, hipMalloc(out, (size_t)bytes));
  return AGP_OK;
}

int agp_device_free(agp_context *ctx, void *ptr) {
  if (!ctx) return AGP_ERR_INVALID_ARGUMENT;
  if (!ptr) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  AGP_HIP_CHECK(ctx, hipFree(ptr));
  return AGP_OK;
}

int agp_memcpy(agp_context *ctx, void *dst, const void *src, int64_t bytes, int kind) {
  if (!ctx || bytes < 0 || (kind != AGP_HOST && kind != AGP_DEVICE) || (bytes > 0 && (!dst || !src))) return AGP_ERR_INVALID_ARGUMENT;
  if (bytes == 0) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (kind == AGP_HOST) AGP_HIP_CHECK(ctx, hipDeviceSynchronize());  // the context's streams are non-blocking: the copy would not wait for them
  AGP_HIP_CHECK(ctx, hipMemcpy(dst, src, (size_t)bytes, kind == AGP_DEVICE ? hipMemcpyHostToDevice : hipMemcpyDeviceToHost));
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
    case AGP_OP_TYPE_PAIR:
      if (depth < 1 || nd.column < 0 || nd.column >= AGP_MAX_SCALE_COLUMNS) return AGP_ERR_INVALID_ARGUMENT;
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
int device_program(agp_context *ctx, const agp_kernel *k, const DevProgram **out) {
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

int validate_features(const agp_features *f) {
  if (!f || f->n < 0 || f->dim < 1 || f->dim > AGP_MAX_DIM) return AGP_ERR_INVALID_ARGUMENT;
  if (f->n_scale_columns < 0 || f->n_scale_columns > AGP_MAX_SCALE_COLUMNS) return AGP_ERR_INVALID_ARGUMENT;
  if (f->n > 0 && !f->coords) return AGP_ERR_INVALID_ARGUMENT;
  if (f->n_scale_columns > 0 && f->n > 0 && !f->scales) return AGP_ERR_INVALID_ARGUMENT;
  return AGP_OK;
}

// Make a device view of a feature vector (uploads host data; `copy` forces an
// owned device copy of device-resident data as well).
int to_device(agp_context *ctx, const agp_features *f, bool copy, DeviceFeatures *out) {
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

int ensure_ws(agp_context *ctx, double **ws, size_t *have, size_t need) {
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
int vector_to_device(agp_context *ctx, const double *src, long long n, int location, double *dst) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(dst, src, sizeof(double) * (size_t)n, kind, ctx->stream));
  if (location == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

int copy_out(agp_context *ctx, const double *dev, long long count, double *dst, int location) {
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyDeviceToHost : hipMemcpyDeviceToDevice;
  AGP_HIP_CHECK(ctx, hipMemcpyAsync(dst, dev, sizeof(double) * (size_t)count, kind, ctx->stream));
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  return AGP_OK;
}

int copy_out_2d(agp_context *ctx, const double *dev, long long ld_dev, long long rows, long long cols,
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
// finish = false: everything is enqueued, nothing is waited for - the caller goes on enqueueing (the backward
// substitution of a fit) and calls finish_factor() after its own synchronisation.
static void finish_factor(agp_context_impl *ctx, const FactorTimers &timers) {
  if (!ctx->profiling) return;
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

static constexpr size_t STATUS_BYTES = 4 * sizeof(int) + 4 * sizeof(double);

// pre (optional): fills and copies the caller wants made BEFORE the Gram matrix is built, in the same launch as the
// zeroing of the flags and the sentinel fills of the panel kernels (pub.h: a fit of a few hundred points used to
// spend ten launches on these)
static int build_and_factor(agp_context *c, const DevProgram *dprog, const DevProgram *hprog, const FeatView &xm,
                            double *A, long long lda, double *invd, double *y, const double *yvar, bool finish = true,
                            FactorTimers *timers_out = nullptr, PrepArgs *pre = nullptr, bool copy_status = true,
                            const double *gram_copy = nullptr) {
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  const long long n = xm.n;
  hipStream_t s = ctx->stream;
  {
    PrepArgs local;
    PrepArgs *prep = pre ? pre : &local;
    prep->fill(ctx->d_flags, 0ull, (long long)(STATUS_BYTES / 8));
    if (prep->n + 5 <= PREP_MAX) ctx->prep_external = panel_fused_plan(ctx, invd, 0, n, true, prep);
    launch_prep(s, *prep);
  }
  const bool prof = ctx->profiling;
  if (prof) AGP_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[0], s));
  // as_measurements(features) -> covariance_function_(measurement_features)   gp.hpp:288-290
  {
    TraceRange tr("agp: gram (compute_covariance_matrix, callers.hpp:107-166)");
    // (gram_copy: the caller holds this very matrix - same leading dimension, measurement variances included - already)
    if (gram_copy) launch_copy_lower(s, gram_copy, lda, n, A, ctx->d_flags);
    else launch_gram(s, dprog, xm, xm, /*symmetric=*/true, /*lower_only=*/true, A, lda, yvar, ctx->d_flags, hprog);
  }
  if (prof) AGP_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[1], s));
  TraceRange tr_factor("agp: factor LL^T + forward substitution (SerializableLDLT, serializable_ldlt.hpp:27)");
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
  ctx->prep_external = false;
  if (prof) AGP_HIP_CHECK(ctx, hipEventRecord(ctx->stage_ev[2], s));
  // (copy_status false: the caller's next launch forwards the status block itself - backsub_coop_kernel)
  if (copy_status || finish) AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_flags, ctx->d_flags, STATUS_BYTES, hipMemcpyDeviceToHost, s));
  if (timers_out) *timers_out = timers;
  if (!finish) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  AGP_HIP_CHECK(ctx, hipGetLastError());
  finish_factor(ctx, timers);
  return AGP_OK;
}

int status_from_flags(const agp_context *ctx) {
  if (ctx->h_flags[2]) {  // a consumer of the fused panel kernel gave up waiting for its producer (chol.hip)
    const_cast<agp_context *>(ctx)->last_error = "panel kernel: hand-over of a diagonal block timed out";
    return AGP_ERR_HIP;
  }
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

// ---- Gram of LinearCombination features ------------------------------------------------------------------
int agp_gram_combined(agp_context *ctx, const agp_kernel *k, const agp_features *x, int64_t nx, const int64_t *x_offsets,
                      const double *x_coefficients, const agp_features *y, int64_t ny, const int64_t *y_offsets,
                      const double *y_coefficients, double *out, int64_t ld, int out_location) {
  if (!ctx || !k || !x || !out || nx < 0 || ny < 0) return AGP_ERR_INVALID_ARGUMENT;
  if ((x_offsets == nullptr) != (x_coefficients == nullptr) || (y_offsets == nullptr) != (y_coefficients == nullptr))
    return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st != AGP_OK) return st;
  if (y && (st = validate_features(y)) != AGP_OK) return st;
  if (y && y->dim != x->dim) return AGP_ERR_INVALID_ARGUMENT;
  const bool symmetric = y == nullptr;
  const long long ex = x->n, ey = y ? y->n : x->n;
  const long long na = x_offsets ? nx : ex, nb = symmetric ? na : (y_offsets ? ny : ey);
  if (na == 0 || nb == 0) return AGP_OK;
  if (ld < na) return AGP_ERR_INVALID_ARGUMENT;
  if (x_offsets && (x_offsets[0] != 0 || x_offsets[nx] != ex)) return AGP_ERR_INVALID_ARGUMENT;
  if (!symmetric && y_offsets && (y_offsets[0] != 0 || y_offsets[ny] != ey)) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dx, dy;
  if ((st = to_device(ctx, x, false, &dx)) != AGP_OK) return st;
  if (y && (st = to_device(ctx, y, false, &dy)) != AGP_OK) return st;
  const FeatView &vy = y ? dy.v : dx.v;
  // workspace: expanded Gram | contracted result | offsets / coefficients of both sides
  const long long ldk = round_up(ex, 2), ldo = round_up(na, 2);
  const size_t k_elems = (size_t)ldk * (size_t)ey, o_elems = (size_t)ldo * (size_t)nb;
  const size_t meta = (size_t)(x_offsets ? 2 * (nx + 1) + ex : 0) + (size_t)((!symmetric && y_offsets) ? 2 * (ny + 1) + ey : 0) + 8;
  if ((st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (k_elems + o_elems + meta))) != AGP_OK) return st;
  double *Kd = ctx->ws_aux, *Od = Kd + k_elems, *m = Od + o_elems;
  hipStream_t s = ctx->stream;
  const long long *xoff_d = nullptr, *yoff_d = nullptr;
  const double *xc_d = nullptr, *yc_d = nullptr;
  if (x_offsets) {
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(m, x_offsets, sizeof(int64_t) * (size_t)(nx + 1), hipMemcpyHostToDevice, s));
    xoff_d = reinterpret_cast<const long long *>(m);
    m += nx + 1;
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(m, x_coefficients, sizeof(double) * (size_t)ex, hipMemcpyHostToDevice, s));
    xc_d = m;
    m += ex;
  }
  if (symmetric) { yoff_d = xoff_d; yc_d = xc_d; }
  else if (y_offsets) {
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(m, y_offsets, sizeof(int64_t) * (size_t)(ny + 1), hipMemcpyHostToDevice, s));
    yoff_d = reinterpret_cast<const long long *>(m);
    m += ny + 1;
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(m, y_coefficients, sizeof(double) * (size_t)ey, hipMemcpyHostToDevice, s));
    yc_d = m;
  }
  AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));  // pageable sources
  launch_gram(s, dprog, dx.v, vy, symmetric, false, Kd, ldk, nullptr, nullptr, &k->prog);
  launch_contract_combinations(s, Kd, ldk, xoff_d, xc_d, na, yoff_d, yc_d, nb, symmetric, Od, ldo);
  return copy_out_2d(ctx, Od, ldo, na, nb, out, ld, out_location);
}

// ---- fit -------------------------------------------------------------------
void agp_fit_destroy(agp_fit *fit) {
  if (!fit) return;
  (void)hipSetDevice(fit->device);
  if (fit->slab) {
    if (--fit->slab->refs == 0) {
      agp_context *ctx = fit->ctx;
      if (ctx && ctx->pool_batch && ctx->pool_batch_bytes != fit->slab->bytes) {  // (the most recent size wins the slot, as pool_A)
        (void)dev_release(ctx->pool_batch);
        ctx->pool_batch = nullptr;
        ctx->pool_batch_bytes = 0;
      }
      if (ctx && !ctx->pool_batch) { ctx->pool_batch = fit->slab->base; ctx->pool_batch_bytes = fit->slab->bytes; }  // (like pool_A)
      else (void)dev_release(fit->slab->base);
      delete fit->slab;
    }
    fit->train.release();
    delete fit;
    return;
  }
  if (fit->A) {
    agp_context *ctx = fit->ctx;
    // (the most recent size wins the slot: a buffer of another size parked there - the headline's 2 GiB factor, say, in a
    // process that goes on to fit 512 points - would cost every later fit a hipMalloc and a hipFree: +0.2 ms per N = 512
    // fit, the "0.21 vs 0.41 ms" of the round-4 review)
    if (ctx && ctx->pool_A && ctx->pool_A_bytes != fit->A_bytes) {
      (void)dev_release(ctx->pool_A);
      ctx->pool_A = nullptr;
      ctx->pool_A_bytes = 0;
    }
    if (ctx && !ctx->pool_A) {
      // kernels reading the factor were enqueued on the context's streams; the
      // next user of the buffer is enqueued on the same streams, after them
      ctx->pool_A = fit->A;
      ctx->pool_A_bytes = fit->A_bytes;
    } else {
      (void)dev_free(fit->A);
    }
  }
  if (fit->aux_base) {
    agp_context *ctx = fit->ctx;
    if (ctx && ctx->pool_aux && ctx->pool_aux_bytes != fit->aux_bytes) {
      (void)dev_release(ctx->pool_aux);
      ctx->pool_aux = nullptr;
      ctx->pool_aux_bytes = 0;
    }
    if (ctx && !ctx->pool_aux) {
      ctx->pool_aux = fit->aux_base;
      ctx->pool_aux_bytes = fit->aux_bytes;
    } else {
      (void)dev_release(fit->aux_base);
    }
  } else {
    if (fit->invd) (void)dev_free(fit->invd);
    if (fit->winv) (void)dev_free(fit->winv);
    if (fit->alpha) (void)dev_free(fit->alpha);
    if (fit->z) (void)dev_free(fit->z);
  }
  fit->train.release();
  delete fit;
}

// ---- x = L^-T z for ONE vector ------------------------------------------------------------------------------
// Wide path (n a multiple of 512, n >= 2048): the 512 x 512 diagonal blocks are inverted explicitly (one batched
// triangular solve against the identity), after which a step is two column-dot launches per 512 rows instead of
// four fused launches per 128 rows: the chain is launch-latency-bound (AGP_WIDE_BACKSOLVE=0: off, =<width>: other
// block width).  Otherwise the 128-row chain on 128 x 128 inverses.  ws: backsolve_ws_elems(n) doubles of scratch.
long long backsolve_width(long long n) {
  constexpr long long BW = 512;
  return (n >= 4 * BW && n % BW == 0) ? BW : 0;
}

size_t backsolve_ws_elems(long long n) {
  const long long BW = backsolve_width(n);
  const size_t blocks = BW ? (size_t)(n / BW) * (size_t)BW * (size_t)BW : (size_t)((n + NB - 1) / NB) * NB * NB;
  return (size_t)round_up(n, 2) + blocks;
}

// first_done / ev_done: the inverses of the first `first_done` blocks are already being computed on another stream
// (factor_lower's early inversion); ev_done completes when they are there.
void backward_solve_vec_any(hipStream_t s, const double *A, long long n, long long lda, const double *invd,
                            double *z, double *ws, long long first_done, hipEvent_t ev_done) {
  double *xs = ws, *W = ws + round_up(n, 2);
  const long long BW = backsolve_width(n);
  if (!BW) {
    invert_diag_blocks(s, A, n, lda, invd, W);
    backward_solve_vec(s, A, n, lda, W, z, xs);
    return;
  }
  const long long nb = n / BW;
  if (first_done < 0 || first_done > nb) first_done = 0;
  // (first_done == nb - 1 - every inverse but the last block's under way - is backward_solve_vec_from's fast path below)
  if (first_done < nb) {
    const long long cnt = nb - first_done;
    launch_set_identity_batched(s, W + first_done * BW * BW, BW, BW * BW, BW, cnt);
    forward_solve_mat_batched(s, A + first_done * BW * (lda + 1), BW * (lda + 1), BW, lda,
                              invd + first_done * (BW / NB) * (long long)(36 * MB * MB), (BW / NB) * (long long)(36 * MB * MB),
                              W + first_done * BW * BW, BW * BW, BW, BW, /*rhs_lower=*/true, cnt);
  }
  if (first_done > 0 && ev_done) (void)hipStreamWaitEvent(s, ev_done, 0);
  for (long long b = nb - 1; b >= 0; --b) {
    const long long k0 = b * BW;
    launch_colvec_dot(s, W + b * BW * BW, BW, BW, BW, z + k0, 1.0, 0.0, nullptr, xs + k0);  // x_B = inv(L_BB)^T z_B
    if (k0 > 0) launch_colvec_dot(s, A + k0, lda, BW, k0, xs + k0, -1.0, 1.0, z, z);         // z[0:k0] -= L[B, 0:k0]^T x_B
  }
  (void)hipMemcpyAsync(z, xs, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
}

// ---- one right-hand side through explicitly inverted BW x BW diagonal blocks (n a multiple of BW): invert_wide_blocks
// (solve.hip) ----------------
// z <- L^-1 z, right-looking: x_B = W_B z_B (in place), z[below] -= L[below, B] x_B: two mat-vec launches per BW rows
// instead of one fused launch per 128 (the chain is launch-latency-bound).  partial: n doubles of scratch.
// A32 (optional): an fp32 copy of the factor's lower triangle (same leading dimension) for the products with the rows
// below / the columns left of a diagonal block - half the bytes of sweeps that are bound by them; for a PRECONDITIONER
static void forward_solve_vec_wide(hipStream_t s, const double *A, long long n, long long lda, const double *W, long long BW,
                                   double *z, double *partial, const float *A32 = nullptr) {
  const long long nb = n / BW;
  for (long long b = 0; b < nb; ++b) {
    const long long k0 = b * BW, below = n - k0 - BW;
    // (x_B goes through `partial` first: the kernel may not overwrite z_B while other workgroups still read it)
    launch_tall_matvec(s, W + b * BW * BW, BW, BW, BW, z + k0, 1.0, 0.0, nullptr, partial);
    (void)hipMemcpyAsync(z + k0, partial, sizeof(double) * (size_t)BW, hipMemcpyDeviceToDevice, s);
    if (below > 0 && A32) launch_tall_matvec_f32(s, A32 + k0 * lda + k0 + BW, lda, below, BW, z + k0, -1.0, 1.0, z + k0 + BW, z + k0 + BW);
    else if (below > 0) launch_tall_matvec(s, A + k0 * lda + k0 + BW, lda, below, BW, z + k0, -1.0, 1.0, z + k0 + BW, z + k0 + BW);
  }
}

// x = L^-T z_in OUT of place (the fit: z = L^-1 y stays, the information vector is written where it lives).  With the
// early inversion under way this is the loop above without its two copy launches: the first update reads z_in and
// writes the work vector (ws[0:n]), every x_B goes straight into x.
void backward_solve_vec_from(hipStream_t s, const double *A, long long n, long long lda, const double *invd, const double *z_in,
                             double *x, double *ws, long long first_done, hipEvent_t ev_done, int *flags) {
  const long long BW = backsolve_width(n), nb = BW ? n / BW : 0;
  // (flags == nullptr: the one-launch substitution of the last block is off - AGP_BACKSUB_COOP=0, or a hand-over timed out
  // earlier on this context)
  if (BW && nb >= 2 && first_done == nb - 1 && flags) {
    double *work = ws, *W = ws + round_up(n, 2);
    const long long k0 = (nb - 1) * BW;
    launch_fill_sentinel(s, x + k0, BW);
    backward_solve_coop(s, A + k0 * (lda + 1), BW, lda, invd + (k0 / NB) * (long long)(36 * MB * MB), z_in + k0, x + k0, flags, nullptr);
    launch_colvec_dot(s, A + k0, lda, BW, k0, x + k0, -1.0, 1.0, z_in, work);  // work[0:k0] = z[0:k0] - L[B, 0:k0]^T x_B
    if (ev_done) (void)hipStreamWaitEvent(s, ev_done, 0);
    for (long long b = nb - 2; b >= 0; --b) {
      const long long c0 = b * BW;
      launch_colvec_dot(s, W + b * BW * BW, BW, BW, BW, work + c0, 1.0, 0.0, nullptr, x + c0);
      if (c0 > 0) launch_colvec_dot(s, A + c0, lda, BW, c0, x + c0, -1.0, 1.0, work, work);
    }
    return;
  }
  (void)hipMemcpyAsync(x, z_in, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
  backward_solve_vec_any(s, A, n, lda, invd, x, ws, first_done, ev_done);
}

// z <- L^-T z with the same inverses (the loop of backward_solve_vec_any); xs: n doubles of scratch
static void backward_solve_vec_wide(hipStream_t s, const double *A, long long n, long long lda, const double *W, long long BW,
                                    double *z, double *xs, const float *A32 = nullptr) {
  const long long nb = n / BW;
  for (long long b = nb - 1; b >= 0; --b) {
    const long long k0 = b * BW;
    launch_colvec_dot(s, W + b * BW * BW, BW, BW, BW, z + k0, 1.0, 0.0, nullptr, xs + k0);
    if (k0 > 0 && A32) launch_colvec_dot_f32(s, A32 + k0, lda, BW, k0, xs + k0, -1.0, 1.0, z, z);
    else if (k0 > 0) launch_colvec_dot(s, A + k0, lda, BW, k0, xs + k0, -1.0, 1.0, z, z);
  }
  (void)hipMemcpyAsync(z, xs, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
}

namespace {
struct MixedRequest {
  int max_iterations = 0;
  double tolerance = 0.;
  int iterations = 0;       // out
  double residual = 0.;     // out: ||y - K a||_2 / ||y||_2 of the returned information vector
};
}  // namespace

static int refine_information(agp_context_impl *ctx, agp_fit *fit, const double *Kfull, const double *Wfwd,
                              double *vec, MixedRequest *mixed);

static int fit_create_impl(agp_context *c, const agp_kernel *k, const agp_features *x, const double *y,
                           const double *y_var, agp_fit **out, double *information, double *log_det,
                           MixedRequest *mixed) {
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
  double *Kfull = nullptr, *Wfwd = nullptr, *vec = nullptr;  // mixed precision only
  size_t Kfull_bytes = 0;
  auto drop_mixed = [&]() {
    if (Kfull) {  // parked for the next mixed fit of the same size, like the factor's own buffer (pool_A)
      if (!ctx->pool_K) { ctx->pool_K = Kfull; ctx->pool_K_bytes = Kfull_bytes; }
      else (void)hipFree(Kfull);
    }
    if (Wfwd) (void)hipFree(Wfwd);
    if (vec) (void)hipFree(vec);
    Kfull = Wfwd = vec = nullptr;
  };
#define FIT_CHECK(expr)                                                                  \
  do {                                                                                   \
    hipError_t _e = (expr);                                                              \
    if (_e != hipSuccess) {                                                              \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);               \
      if (yvar_d) (void)hipFree(yvar_d);                                                 \
      drop_mixed();                                                                      \
      agp_fit_destroy(fit);                                                              \
      return AGP_ERR_HIP;                                                                \
    }                                                                                    \
  } while (0)
  fit->ctx = ctx;
  fit->A_bytes = sizeof(double) * (size_t)fit->lda * (size_t)n;
  if (ctx->pool_A && ctx->pool_A_bytes == fit->A_bytes) {
    fit->A = ctx->pool_A;
    ctx->pool_A = nullptr;
    ctx->pool_A_bytes = 0;
  } else {
    FIT_CHECK(hipMalloc(&fit->A, fit->A_bytes));
  }
  {
    const size_t n_invd = (size_t)nblk * (36 * MB * MB), n_winv = (size_t)nblk * NB * NB, n_vec = (size_t)round_up(n, 2);
    // (+ train_features = features, un-wrapped; gp.hpp:63,293: always a copy, inside this pooled block - a hipMalloc and a
    // hipFree per fit are ~0.1 ms of a 2 ms fit)
    const size_t n_feat = (size_t)n * ((size_t)x->dim + (x->eq_id ? 1 : 0) + (size_t)x->n_scale_columns);
    fit->aux_bytes = sizeof(double) * (n_invd + n_winv + 2 * n_vec + n_feat);
    if (ctx->pool_aux && ctx->pool_aux_bytes == fit->aux_bytes) {
      fit->aux_base = ctx->pool_aux;
      ctx->pool_aux = nullptr;
      ctx->pool_aux_bytes = 0;
    } else {
      FIT_CHECK(hipMalloc(&fit->aux_base, fit->aux_bytes));
    }
    fit->invd = fit->aux_base;
    fit->winv = fit->invd + n_invd;
    fit->alpha = fit->winv + n_winv;
    fit->z = fit->alpha + n_vec;
  }
  const hipMemcpyKind kind = x->location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  // device-resident inputs: the copies of the training features and targets travel in the ONE preparation launch of
  // build_and_factor (pub.h: PrepArgs) instead of a copy kernel each; host inputs are uploads and stay what they were
  PrepArgs pre;
  const bool dev_in = x->location != AGP_HOST;
  auto stage = [&](void *dst, const void *src, long long words) -> hipError_t {
    if (dev_in) { pre.copy(dst, src, words); return hipSuccess; }
    return hipMemcpyAsync(dst, src, sizeof(double) * (size_t)words, kind, s);
  };
  {
    double *fcur = fit->z + round_up(n, 2);
    FeatView v;
    v.n = n; v.dim = x->dim; v.nsc = x->n_scale_columns; v.meas = 0;
    v.coords = fcur; v.ids = nullptr; v.scales = nullptr;
    FIT_CHECK(stage(fcur, x->coords, n * (long long)x->dim));
    fcur += (size_t)n * (size_t)x->dim;
    if (x->eq_id) {
      v.ids = reinterpret_cast<const long long *>(fcur);
      FIT_CHECK(stage(fcur, x->eq_id, n));
      fcur += n;
    }
    if (x->n_scale_columns > 0) {
      v.scales = fcur;
      FIT_CHECK(stage(fcur, x->scales, n * (long long)x->n_scale_columns));
    }
    fit->train.v = v;  // (a view into aux_base: DeviceFeatures::release has nothing to free)
  }
  FIT_CHECK(stage(fit->z, y, n));
  if (y_var) {
    FIT_CHECK(hipMalloc(&yvar_d, sizeof(double) * (size_t)n));
    FIT_CHECK(stage(yvar_d, y_var, n));
  }
  if (x->location == AGP_HOST) FIT_CHECK(hipStreamSynchronize(s));
  if (mixed && pre.n > 0) {  // (the mixed fit reads the staged inputs before build_and_factor: its exact covariance, the copy of y)
    launch_prep(s, pre);
    pre = PrepArgs();
  }
  FeatView xm = fit->train.v;
  xm.meas = 1;  // as_measurements(features), gp.hpp:288
  if (mixed) {
    // the exact fp64 covariance (lower triangle: the refinement multiplies with launch_symv_lower) for the residuals, the targets,
    // and the work vectors r, z, p, q
    Kfull_bytes = fit->A_bytes;
    if (ctx->pool_K && ctx->pool_K_bytes == Kfull_bytes) {
      Kfull = ctx->pool_K;
      ctx->pool_K = nullptr;
      ctx->pool_K_bytes = 0;
    } else {
      if (ctx->pool_K) { (void)dev_release(ctx->pool_K); ctx->pool_K = nullptr; ctx->pool_K_bytes = 0; }
      FIT_CHECK(hipMalloc(&Kfull, Kfull_bytes));
    }
    FIT_CHECK(hipMalloc(&Wfwd, sizeof(double) * (size_t)nblk * NB * NB));
    FIT_CHECK(hipMalloc(&vec, sizeof(double) * (size_t)n * 5));
    FIT_CHECK(hipMemcpyAsync(vec, fit->z, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
    launch_gram(s, dprog, xm, xm, /*symmetric=*/true, /*lower_only=*/true, Kfull, fit->lda, yvar_d, ctx->d_flags,
                &k->prog);
    // fp32-accurate products of the bulk updates: on the 16-bit matrix pipe from split planes of the panel, or
    // - AGP_MIXED_BF16=0 - on the fp32 MFMA as in rounds 1-4
    // default (round 6): from two fp16 planes of power-of-two-scaled rows (gemm_f16x2.hip); AGP_MIXED_F16=0: three bf16 planes
    const bool planes16 = ctx->tune.mixed_bf16;
    const bool f16 = planes16 && ctx->tune.mixed_f16;
    ctx->update_variant = f16 ? 5 : (planes16 ? 4 : 3);
    ctx->nbo_wide = (planes16 && ctx->tune.mixed_nbo > 512 && ctx->tune.mixed_nbo % 128 == 0) ? ctx->tune.mixed_nbo : 0;
    if (f16 && ctx->f16_scales_n < n) {
      if (ctx->f16_scales) (void)hipFree(ctx->f16_scales);
      ctx->f16_scales = nullptr; ctx->f16_scales_n = 0;
      const long long cap = round_up(n, 2);
      if (hipMalloc(&ctx->f16_scales, sizeof(double) * 2 * (size_t)cap) == hipSuccess) ctx->f16_scales_n = cap;
      else (void)hipGetLastError();  // (factor_lower falls back to the bf16 planes)
    }
    {  // two panel copies of (n rows + padding) x 512 (chol.hip, factor_lower): fp32, or three bf16 planes (two fp16 planes fit in the
       // same space: the fall-back from fp16 to bf16 planes needs no second allocation); kept in the context
      const size_t want = planes16 ? 2 * bf16x3_bytes(n, ctx->nbo_wide > 512 ? ctx->nbo_wide : 512)
                                   : sizeof(float) * 2 * ((size_t)n + 16) * 512;
      if (ctx->p32_bytes < want) {
        if (ctx->p32) (void)hipFree(ctx->p32);
        ctx->p32 = nullptr; ctx->p32_bytes = 0;
        if (hipMalloc(&ctx->p32, want) == hipSuccess) ctx->p32_bytes = want;
        else (void)hipGetLastError();  // (the kernels round the operands themselves then)
      }
    }
  }
  // The fp64 fit does not wait for the factorisation before it enqueues the backward substitution: one host round trip
  // (~0.1 ms) less; the status is read after the single synchronisation at the end, and a substitution through a factor
  // that turns out not to be positive definite was wasted work on garbage, nothing more.  (The flags and the
  // log-determinant are copied to the host right behind the factorisation, before the substitution touches d_scalars.)
  const bool deferred = !mixed && !yvar_d;
  // information = L^-T z in ONE launch (solve.hip: backsub_coop_kernel) for the fp64 fit: its output vector is the
  // hand-over buffer and is sentinel-filled by the preparation launch.  (The refinement of a mixed fit needs the
  // inverted diagonal blocks anyway and keeps the launch-per-block substitution.)
  // Measured (profiles/r05): one hand-over + substitution per 128-column block is ~9 us - 37 us at N = 512 against 45 us
  // for the launch chain, 286 us at N = 4096 against 190 us through the 512-wide inverted blocks: small fits only.
  const bool coop = !mixed && ctx->tune.backsub_coop && n <= ctx->tune.backsub_coop_max;
  // (hand-over: the sentinel-filled output itself up to BACKSUB_DIRECT_BLOCKS blocks, per-block flags in the unused
  // block-inverse buffer beyond)
  const bool coop_direct = coop && (n + NB - 1) / NB <= BACKSUB_DIRECT_BLOCKS;
  if (coop && coop_direct) pre.sentinel(fit->alpha, n);
  else if (coop) pre.fill(fit->winv, 0ull, backsub_done_words(n, 1));
  FactorTimers ftimers;
  long long bs_done = 0;
  if (!coop && deferred && backsolve_width(n)) {
    // the inverses of the wide diagonal blocks for the backward substitution: computed by factor_lower on its idle
    // second stream while the chain-bound tail of the factorisation runs (common.h: bs_W)
    if (ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * backsolve_ws_elems(n)) == AGP_OK) {
      ctx->bs_W = ctx->ws_aux + round_up(n, 2);
      ctx->bs_BW = backsolve_width(n);
      ctx->bs_done = 0;
    }
  }
  // (a deferred fit with the one-launch substitution: that launch's last workgroup writes the status block into the
  // pinned mirror - no copy launch behind the factorisation, none behind the substitution)
  const bool status_in_kernel = coop && deferred && ctx->h_status_dev != nullptr;
  st = build_and_factor(ctx, dprog, &k->prog, xm, fit->A, fit->lda, fit->invd, fit->z, yvar_d, !deferred, &ftimers, &pre,
                        !status_in_kernel, mixed ? Kfull : nullptr);
  ctx->update_variant = -1;
  ctx->nbo_override = 0;
  ctx->nbo_wide = 0;
  bs_done = ctx->bs_W ? ctx->bs_done : 0;
  ctx->bs_W = nullptr;
  ctx->bs_done = 0;
  if (yvar_d) { (void)hipFree(yvar_d); yvar_d = nullptr; }
  if (st != AGP_OK) {
    // (the early inversion may still be writing ws_aux on the second stream: whatever uses it next is ordered behind it)
    if (bs_done > 0) (void)hipStreamWaitEvent(s, ctx->ev_inv, 0);
    drop_mixed();
    agp_fit_destroy(fit);
    return st;
  }
  if (!deferred) {
    st = status_from_flags(ctx);
    fit->failed_pivot = ctx->h_flags[1] ? (int64_t)ctx->h_flags[1] - 1 : -1;
    fit->log_det = 2. * ctx->h_scalars[0];
    if (st != AGP_OK) {
      // keep a handle so the caller can query the failed pivot, but no factor
      drop_mixed();
      *out = fit;
      return st;
    }
  }
  // information = L^-T (L^-1 y)
  if (ctx->profiling) FIT_CHECK(hipEventRecord(ctx->stage_ev[3], s));
  if (coop) {
    TraceRange tr("agp: backward substitution (information = ldlt.solve(y), gp.hpp:68)");
    backward_solve_coop(s, fit->A, n, fit->lda, fit->invd, fit->z, fit->alpha, ctx->d_flags,
                        coop_direct ? nullptr : reinterpret_cast<unsigned long long *>(fit->winv), 1, 0, 0, 0, 0, 0,
                        status_in_kernel ? ctx->d_flags : nullptr, ctx->h_status_dev, (int)(STATUS_BYTES / 8));
    // (the hand-over flag of the substitution: the 48-byte status copy left before it ran)
    if (!status_in_kernel) FIT_CHECK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  } else {
    const int st2 = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * backsolve_ws_elems(n));
    if (st2 != AGP_OK) { drop_mixed(); agp_fit_destroy(fit); return st2; }
    {
      TraceRange tr("agp: backward substitution (information = ldlt.solve(y), gp.hpp:68)");
      backward_solve_vec_from(s, fit->A, n, fit->lda, fit->invd, fit->z, fit->alpha, ctx->ws_aux, bs_done, ctx->ev_inv,
                              ctx->tune.backsub_coop ? ctx->d_flags : nullptr);
      // (its last wide block goes through the one-launch substitution, which records a timed-out hand-over in flags[2]: the
      // status block left for the host BEFORE the substitution ran - the flags once more behind it)
      if (deferred) FIT_CHECK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    }
    // the refinement steps of the mixed-precision fit use the 128-row chain on fit->winv
    if (mixed) invert_diag_blocks(s, fit->A, n, fit->lda, fit->invd, fit->winv);
  }
  if (mixed) {
    invert_diag_blocks_forward(s, n, fit->invd, Wfwd);
    const int st3 = refine_information(ctx, fit, Kfull, Wfwd, vec, mixed);
    drop_mixed();
    if (st3 != AGP_OK) { agp_fit_destroy(fit); return st3; }
  }
  if (ctx->profiling) FIT_CHECK(hipEventRecord(ctx->stage_ev[4], s));
  // (deferred status: the caller's buffer is written only once the factor is known to be good - a NaN, a non-positive
  // pivot or a timed-out hand-over leaves it untouched)
  if (information && !deferred) FIT_CHECK(hipMemcpyAsync(information, fit->alpha, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost, s));
  FIT_CHECK(hipStreamSynchronize(s));
  FIT_CHECK(hipGetLastError());
  if (deferred) {
    finish_factor(ctx, ftimers);
    st = status_from_flags(ctx);
    fit->failed_pivot = ctx->h_flags[1] ? (int64_t)ctx->h_flags[1] - 1 : -1;
    fit->log_det = 2. * ctx->h_scalars[0];
    if (st != AGP_OK) {  // keep a handle so the caller can query the failed pivot, but no factor
      *out = fit;
      return st;
    }
    if (information) FIT_CHECK(hipMemcpy(information, fit->alpha, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  } else if (coop && ctx->h_flags[2]) {  // the one-launch substitution gave up on a hand-over (its producer died)
    ctx->last_error = "back substitution: hand-over timed out";
    agp_fit_destroy(fit);
    return AGP_ERR_HIP;
  }
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

// A hand-over of the step launches that timed out (flags[2]: fewer workgroup slots than the launch layout assumed - a CU
// mask, a partition - so that waiters sat in front of their producers) is not the caller's problem: the fit is repeated
// once on the two-launch schedule, which waits for nothing inside a launch, and the context stays on it.
static int fit_create_retrying(agp_context *c, const agp_kernel *k, const agp_features *x, const double *y, const double *y_var,
                               agp_fit **out, double *information, double *log_det, MixedRequest *mixed) {
  int st = fit_create_impl(c, k, x, y, y_var, out, information, log_det, mixed);
  if (st == AGP_ERR_HIP && c && c->h_flags && c->h_flags[2] && (c->tune.step_below > 0 || c->tune.panel_fused || c->tune.merge_above > 0)) {
    if (out && *out) { agp_fit_destroy(*out); *out = nullptr; }
    c->tune.step_below = 0;
    c->tune.panel_fused = false;
    c->tune.backsub_coop = false;  // (the one-launch substitution hands over inside a launch too)
    c->tune.merge_above = 0;       // (... and the gate of a merged bulk update waits for another stream's launch)
    st = fit_create_impl(c, k, x, y, y_var, out, information, log_det, mixed);
  }
  return st;
}

int agp_fit_create(agp_context *c, const agp_kernel *k, const agp_features *x, const double *y,
                   const double *y_var, agp_fit **out, double *information, double *log_det) {
  return fit_create_retrying(c, k, x, y, y_var, out, information, log_det, nullptr);
}

int agp_fit_create_mixed(agp_context *c, const agp_kernel *k, const agp_features *x, const double *y,
                         const double *y_var, int max_iterations, double tolerance, agp_fit **out,
                         double *information, double *log_det, int *iterations, double *residual) {
  if (max_iterations < 0 || !(tolerance >= 0.)) return AGP_ERR_INVALID_ARGUMENT;
  MixedRequest m;
  m.max_iterations = max_iterations;
  m.tolerance = tolerance;
  const int st = fit_create_retrying(c, k, x, y, y_var, out, information, log_det, &m);
  if (iterations) *iterations = m.iterations;
  if (residual) *residual = m.residual;
  return st;
}

// Conjugate gradients on K a = y in fp64, preconditioned with the mixed-precision factor (M = L L^T,
// ||I - M^-1 K|| ~ cond(K) * 2^-24): fit->alpha enters as M^-1 y and leaves as K^-1 y to the requested
// relative residual.  vec: 5 n doubles, vec[0:n] = y on entry.
static int refine_information(agp_context_impl *ctx, agp_fit *fit, const double *Kfull, const double *Wfwd,
                              double *vec, MixedRequest *mixed) {
  hipStream_t s = ctx->stream;
  const long long n = fit->n, lda = fit->lda;
  double *yv = vec, *r = vec + n, *z = vec + 2 * n, *p = vec + 3 * n, *q = vec + 4 * n;
  double *xa = fit->alpha;
  double *dots = ctx->d_scalars;  // 4 device doubles (log-det already read back)
  auto read_dots = [&](int count, double *host) -> int {
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(ctx->h_scalars, dots, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
    AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
    for (int i = 0; i < count; ++i) host[i] = ctx->h_scalars[i];
    return AGP_OK;
  };
  // the substitutions of every step: through 512-wide inverted diagonal blocks when the size allows (2 x n / 512
  // mat-vec launches per direction instead of n / 128 fused steps: the chains are launch-latency-bound)
  // (1024-wide inverted blocks here: the eight-odd preconditioner applications share one inversion, and half as many
  // launch-bound block steps per sweep are worth 5 ms at N = 32768; a single substitution is better off with 512)
  long long BW = backsolve_width(n);
  if (BW == 512 && n % 1024 == 0 && n >= 8192) BW = 1024;
  // the mat-vec's workspace and the inverted blocks (268 MB at N = 32768): kept in the context between mixed fits - a
  // hipMalloc + hipFree pair of that size per fit was 2-3 ms of a 120 ms fit
  const size_t symv_elems = (symv_ws_elems(n) + 1) / 2 * 2;
  const size_t wide_elems = BW ? (size_t)(n / BW) * (size_t)BW * (size_t)BW : 0;
  const size_t np2 = (size_t)round_up(n, 2);
  {
    const int st_ws = ensure_ws(ctx, &ctx->ws_refine, &ctx->ws_refine_bytes, sizeof(double) * (symv_elems + 2 * wide_elems + 2 * np2));
    if (st_ws != AGP_OK) return st_ws;
  }
  double *symv_ws = ctx->ws_refine;
  double *Wwide = BW ? ctx->ws_refine + symv_elems : nullptr;
  // ... and their transposes: the forward sweep applies inv(L_BB) as a column-wise product with the transposed copy
  // (one workgroup per output, like the backward sweep) instead of a row-wise one on BW / 64 workgroups
  double *WwideT = BW ? Wwide + wide_elems : nullptr;
  double *t1 = ctx->ws_refine + symv_elems + 2 * wide_elems, *t2 = t1 + np2;  // the sweeps' work vectors
  if (BW) {
    invert_wide_blocks(s, fit->A, n, lda, fit->invd, BW, Wwide);
    launch_transpose_blocks(s, Wwide, WwideT, BW, n / BW);
  }
  // The preconditioner M = L L^T is applied eight-odd times and each application streams L twice; it does not have to
  // be exact - L itself comes from fp32-rounded products -, so the sweeps read an fp32 COPY of L's off-diagonal blocks
  // (the inverted diagonal blocks stay fp64): half the bytes, kept in the context between fits.
  float *L32 = nullptr;
  if (BW) {
    const size_t want = sizeof(float) * (size_t)lda * (size_t)n;
    if (ctx->pool_L32 && ctx->pool_L32_bytes != want) { (void)hipFree(ctx->pool_L32); ctx->pool_L32 = nullptr; ctx->pool_L32_bytes = 0; }
    if (!ctx->pool_L32) {
      if (hipMalloc(&ctx->pool_L32, want) == hipSuccess) ctx->pool_L32_bytes = want;
      else { (void)hipGetLastError(); ctx->pool_L32 = nullptr; }
    }
    L32 = ctx->pool_L32;
    if (L32) launch_convert_lower_f32(s, fit->A, lda, n, L32);
  }
  auto precondition = [&](const double *in, double *outv) {
    if (BW && L32) {
      // both sweeps out of place - no staging copy per block, none at the end: t1 = in is consumed by the forward sweep
      // (x_B into t2, the rows below updated in t1), t2 by the backward sweep (x_B into outv, the rows above updated in t2)
      const long long nbw = n / BW;
      (void)hipMemcpyAsync(t1, in, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
      for (long long b = 0; b < nbw; ++b) {
        const long long k0 = b * BW, below = n - k0 - BW;
        launch_colvec_dot(s, WwideT + b * BW * BW, BW, BW, BW, t1 + k0, 1.0, 0.0, nullptr, t2 + k0);  // x_B = inv(L_BB) z_B
        if (below > 0) launch_tall_matvec_f32(s, L32 + k0 * lda + k0 + BW, lda, below, BW, t2 + k0, -1.0, 1.0, t1 + k0 + BW, t1 + k0 + BW);
      }
      for (long long b = nbw - 1; b >= 0; --b) {
        const long long k0 = b * BW;
        launch_colvec_dot(s, Wwide + b * BW * BW, BW, BW, BW, t2 + k0, 1.0, 0.0, nullptr, outv + k0);  // x_B = inv(L_BB)^T z_B
        if (k0 > 0) launch_colvec_dot_f32(s, L32 + k0, lda, BW, k0, outv + k0, -1.0, 1.0, t2, t2);
      }
      return;
    }
    (void)hipMemcpyAsync(outv, in, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
    if (BW) {
      forward_solve_vec_wide(s, fit->A, n, lda, Wwide, BW, outv, ctx->ws_aux, L32);
      backward_solve_vec_wide(s, fit->A, n, lda, Wwide, BW, outv, ctx->ws_aux, L32);
    } else {
      forward_solve_vec(s, fit->A, n, lda, Wfwd, outv, ctx->ws_aux);
      backward_solve_vec(s, fit->A, n, lda, fit->winv, outv, ctx->ws_aux);
    }
  };
  double h[4];
  int st;
  // r = y - K a
  launch_symv_lower(s, Kfull, lda, n, xa, -1., 1., yv, r, symv_ws);
  launch_dot(s, yv, yv, n, dots + 0);
  launch_dot(s, r, r, n, dots + 1);
  if ((st = read_dots(2, h)) != AGP_OK) return st;
  const double ynorm = std::sqrt(h[0]);
  double rnorm = std::sqrt(h[1]);
  const double target = mixed->tolerance * ynorm;
  int it = 0;
  if (rnorm > target && mixed->max_iterations > 0) {
    precondition(r, z);
    (void)hipMemcpyAsync(p, z, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
    launch_dot(s, r, z, n, dots + 0);
    if ((st = read_dots(1, h)) != AGP_OK) return st;
    double rz = h[0];
    double best = rnorm;
    int stalled = 0;
    while (it < mixed->max_iterations && rnorm > target) {
      launch_symv_lower(s, Kfull, lda, n, p, 1., 0., nullptr, q, symv_ws);
      launch_dot(s, p, q, n, dots + 0);
      if ((st = read_dots(1, h)) != AGP_OK) return st;
      if (!(h[0] > 0.) || !(rz > 0.)) break;  // breakdown: K or the preconditioner lost definiteness
      const double a = rz / h[0];
      launch_axpby(s, n, a, p, 1., xa, xa);
      launch_axpby(s, n, -a, q, 1., r, r);
      ++it;
      precondition(r, z);
      launch_dot(s, r, r, n, dots + 0);
      launch_dot(s, r, z, n, dots + 1);
      if ((st = read_dots(2, h)) != AGP_OK) return st;
      rnorm = std::sqrt(h[0]);
      if (rnorm < 0.9 * best) { best = rnorm; stalled = 0; }
      else if (++stalled >= 3) break;  // at the fp64 floor of this system
      const double beta = h[1] / rz;
      rz = h[1];
      launch_axpby(s, n, 1., z, beta, p, p);
    }
    // report the TRUE residual of what is returned (the recurrence drifts)
    launch_symv_lower(s, Kfull, lda, n, xa, -1., 1., yv, r, symv_ws);
    launch_dot(s, r, r, n, dots + 0);
    if ((st = read_dots(1, h)) != AGP_OK) return st;
    rnorm = std::sqrt(h[0]);
  }
  mixed->iterations = it;
  mixed->residual = ynorm > 0. ? rnorm / ynorm : rnorm;
  return AGP_OK;
}

int64_t agp_fit_size(const agp_fit *fit) { return fit ? fit_real_rows(fit) : 0; }
int64_t agp_fit_failed_pivot(const agp_fit *fit) { return fit ? fit->failed_pivot : -1; }

int agp_fit_log_determinant(const agp_fit *fit, double *out) {
  if (!fit || !out) return AGP_ERR_INVALID_ARGUMENT;
  *out = fit->log_det;
  return AGP_OK;
}

int agp_fit_download_information(agp_context *ctx, const agp_fit *fit, double *information) {
  if (!ctx || !fit || !information || !fit->alpha) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  if (!fit->phantom.empty()) return fit_compact_vector(ctx, fit, fit->alpha, information, AGP_HOST);
  return copy_out(ctx, fit->alpha, fit->n, information, AGP_HOST);
}

int agp_fit_download_factor(agp_context *ctx, const agp_fit *fit, double *L, int64_t ld) {
  if (!ctx || !fit || !L || ld < fit_real_rows(fit)) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = fit->n;
  if (!fit->phantom.empty()) {
    // drop the phantom rows AND columns: they are decoupled (identity rows), what remains is the factor of the real matrix
    const long long nr = fit_real_rows(fit), ldt = round_up(nr, 2);
    int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * ((size_t)fit->lda * (size_t)n + (size_t)ldt * (size_t)n));
    if (st != AGP_OK) return st;
    double *full = ctx->ws_aux, *rows = full + (size_t)fit->lda * (size_t)n;
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(full, fit->A, sizeof(double) * (size_t)fit->lda * (size_t)n, hipMemcpyDeviceToDevice, ctx->stream));
    launch_zero_upper(ctx->stream, full, fit->lda, n);
    if ((st = fit_compact_matrix(ctx, fit, full, fit->lda, n, rows, ldt, AGP_DEVICE)) != AGP_OK) return st;  // rows: nr x n
    // columns: one strided copy per run of real columns
    long long pad = 0, real = 0;
    auto copy_cols = [&](long long p0, long long r0, long long len) -> int {
      return copy_out_2d(ctx, rows + p0 * ldt, ldt, nr, len, L + r0 * ld, ld, AGP_HOST);
    };
    for (const auto &ph : fit->phantom) {
      if (ph.first > pad) { if ((st = copy_cols(pad, real, ph.first - pad)) != AGP_OK) return st; real += ph.first - pad; }
      pad = ph.second;
    }
    if (n > pad && (st = copy_cols(pad, real, n - pad)) != AGP_OK) return st;
    return AGP_OK;
  }
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

// `bytes` of the context's pinned staging area (common.h: h_stage), grown on demand; nullptr if it cannot be had (the
// callers then stage through pageable memory and synchronise once).  The previous contents are dead: every user ends its
// call with a synchronisation of the stream that read them.
static void *host_stage(agp_context *ctx, size_t bytes) {
  if (ctx->h_stage_bytes < bytes) {
    if (ctx->h_stage) (void)hipHostFree(ctx->h_stage);
    ctx->h_stage = nullptr;
    ctx->h_stage_bytes = 0;
    const size_t want = (bytes + 65535) / 65536 * 65536;
    if (hipHostMalloc(&ctx->h_stage, want) != hipSuccess) { (void)hipGetLastError(); ctx->h_stage = nullptr; return nullptr; }
    ctx->h_stage_bytes = want;
  }
  return ctx->h_stage;
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
  // workspace: [A slabs | tile images | y slabs | yvar | logsum | quad | z slots of the fused panel launches | Gram table]
  const bool fused_panels = batched_fused_fits(ctx, n, count);
  const size_t table_elems = (gram_batch_table_bytes(count) + 7) / 8;
  const size_t elems = (size_t)count * ((size_t)stride_A + (size_t)stride_I + (size_t)np2) + (size_t)np2 +
                       2 * (size_t)round_up(count, 2) + (fused_panels ? (size_t)count * (size_t)np2 : 0) + table_elems;
  if ((st = ensure_ws(ctx, &ctx->ws_A, &ctx->ws_A_bytes, sizeof(double) * elems)) != AGP_OK) return st;
  double *A = ctx->ws_A, *invd = A + (size_t)count * (size_t)stride_A, *ys = invd + (size_t)count * (size_t)stride_I;
  double *yvar_d = ys + (size_t)count * (size_t)np2, *logsum = yvar_d + np2, *quad = logsum + round_up(count, 2);
  double *zpub = fused_panels ? quad + round_up(count, 2) : nullptr;
  void *table = quad + round_up(count, 2) + (fused_panels ? (size_t)count * (size_t)np2 : 0);
  const int loc = features[0]->location;
  const hipMemcpyKind kind = loc == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  for (int b = 0; b < count; ++b)
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(ys + (size_t)b * (size_t)np2, y + (size_t)b * (size_t)ldy, sizeof(double) * (size_t)n,
                                      kind, s));
  if (y_var) AGP_HIP_CHECK(ctx, hipMemcpyAsync(yvar_d, y_var, sizeof(double) * (size_t)n, kind, s));
  if (loc == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
  {  // one preparation launch: log sums and flags zeroed, the hand-over buffers of the fused panel launches sentinel-filled
    PrepArgs prep;
    prep.fill(logsum, 0ull, round_up(count, 2));
    prep.fill(ctx->d_flags, 0ull, 2);
    if (zpub) {
      prep.sentinel(invd, count * stride_I);
      prep.sentinel(zpub, count * np2);
    }
    launch_prep(s, prep);
  }
  std::vector<DeviceFeatures> dxs((size_t)count);
  const agp_features *last = nullptr;
  int last_b = -1;
  std::vector<FeatView> views((size_t)count);
  std::vector<const DevProgram *> hprogs((size_t)count);
  std::vector<double *> outs((size_t)count);
  for (int b = 0; b < count && st == AGP_OK; ++b) {
    // parameter vectors usually share one feature array: upload it once
    const bool same = last && features[b]->coords == last->coords && features[b]->scales == last->scales &&
                      features[b]->eq_id == last->eq_id;
    if (!same) {
      if ((st = to_device(ctx, features[b], false, &dxs[(size_t)b])) != AGP_OK) break;
      last = features[b];
      last_b = b;
    }
    views[(size_t)b] = dxs[(size_t)(same ? last_b : b)].v;
    views[(size_t)b].meas = 1;  // as_measurements(features), gp.hpp:288
    hprogs[(size_t)b] = &kernels[b]->prog;
    outs[(size_t)b] = A + (size_t)b * (size_t)stride_A;
  }
  bool gram_done = false;
  if (st == AGP_OK && count > 1) {  // all Gram matrices in ONE launch when the trees share a fast path (gram.hip)
    std::vector<const double *> diag((size_t)count, y_var ? yvar_d : nullptr);
    // (the descriptor table lives in the workspace and is uploaded from the context's pinned staging area: no allocation,
    // no synchronisation between the upload and the launch)
    gram_done = launch_gram_batch(s, count, hprogs.data(), views.data(), outs.data(), lda, y_var ? diag.data() : nullptr, nullptr, table,
                                  host_stage(ctx, gram_batch_table_bytes(count)));
  }
  for (int b = 0; b < count && st == AGP_OK && !gram_done; ++b) {
    const DevProgram *dprog = nullptr;
    if ((st = device_program(ctx, kernels[b], &dprog)) != AGP_OK) break;
    launch_gram(s, dprog, views[(size_t)b], views[(size_t)b], true, true, outs[(size_t)b], lda, y_var ? yvar_d : nullptr, nullptr,
                &kernels[b]->prog);
  }
  if (st == AGP_OK) {
    factor_lower_batched(s, A, stride_A, n, lda, invd, stride_I, ys, np2, count, ctx->d_flags, logsum, 0, zpub, np2);
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

// B independent fits of one shape in lock step: the Fit<GPFit> constructor (models/gp.hpp:61-69) for `count` datasets /
// parameter vectors at once.  Where the reference's users live - N of a few hundred to a few thousand
// (benchmarks/bench_predict.cc:20-40, the tuner loop tune/tune.hpp:276-290) - ONE fit is bound by the latency of its
// 128 serial pivots per panel (27-30 us per POTRF, config 2: 0.14 of the MFMA peak); a batch shares that latency and
// fills the chip with the trailing updates of all problems (factor_lower_batched).
int agp_fit_create_batch(agp_context *c, int count, const agp_kernel *const *kernels, const agp_features *const *features,
                         const double *y, int64_t ldy, const double *y_var, int64_t ldv, agp_fit **out, double *information,
                         int64_t ldi, double *log_det, int *status) {
  if (!c || count <= 0 || !kernels || !features || !y || !out || !status) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  for (int b = 0; b < count; ++b) { out[b] = nullptr; status[b] = AGP_ERR_INVALID_ARGUMENT; }
  const long long n = features[0] ? features[0]->n : 0;
  if (n <= 0 || (ldy != 0 && ldy < n) || (y_var && ldv != 0 && ldv < n) || (information && ldi < n)) return AGP_ERR_INVALID_ARGUMENT;
  int st = AGP_OK;
  for (int b = 0; b < count; ++b) {
    if (!kernels[b] || !features[b] || features[b]->n != n || features[b]->location != features[0]->location)
      return AGP_ERR_INVALID_ARGUMENT;
    if ((st = validate_features(features[b])) != AGP_OK) return st;
  }
  const long long lda = factor_ld(n), nblk = (n + NB - 1) / NB, np2 = round_up(n, 2), cp2 = round_up(count, 2);
  const long long stride_A = lda * n, stride_I = nblk * (36 * MB * MB);
  hipStream_t s = ctx->stream;
  // one allocation: [A slabs | tile images | information | z | logsum | flags (4 ints each) | y_var (scratch) | features]
  size_t feat_elems = 0;  // per problem: coordinates, equality ids, scale columns (8-byte units)
  for (int b = 0; b < count; ++b)
    feat_elems += (size_t)n * ((size_t)features[b]->dim + (features[b]->eq_id ? 1 : 0) + (size_t)features[b]->n_scale_columns);
  const size_t head_elems = (size_t)count * ((size_t)stride_A + (size_t)stride_I + 2 * (size_t)np2) + (size_t)cp2 + 2 * (size_t)cp2 +
                            (y_var ? (size_t)count * (size_t)np2 : 0);
  const size_t elems = head_elems + feat_elems;
  double *base = nullptr;
  if (ctx->pool_batch && ctx->pool_batch_bytes == sizeof(double) * elems) {
    base = ctx->pool_batch;
    ctx->pool_batch = nullptr;
    ctx->pool_batch_bytes = 0;
  } else {
    if (ctx->pool_batch) { (void)dev_release(ctx->pool_batch); ctx->pool_batch = nullptr; ctx->pool_batch_bytes = 0; }
    AGP_HIP_CHECK(ctx, hipMalloc(&base, sizeof(double) * elems));
  }
  double *A = base, *invd = A + (size_t)count * (size_t)stride_A, *alpha = invd + (size_t)count * (size_t)stride_I;
  double *z = alpha + (size_t)count * (size_t)np2, *logsum = z + (size_t)count * (size_t)np2;
  int *flags = reinterpret_cast<int *>(logsum + cp2);
  double *yvar_d = y_var ? logsum + 3 * cp2 : nullptr;
  std::vector<agp_fit *> fits((size_t)count, nullptr);
  auto fail = [&](int code) {
    (void)hipStreamSynchronize(s);
    for (auto *f : fits)
      if (f) { f->train.release(); delete f; }
    (void)hipFree(base);
    return code;
  };
#define BATCH_CHECK(expr)                                                     \
  do {                                                                        \
    hipError_t _e = (expr);                                                   \
    if (_e != hipSuccess) {                                                   \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);    \
      return fail(AGP_ERR_HIP);                                               \
    }                                                                         \
  } while (0)
  const int loc = features[0]->location;
  const hipMemcpyKind kind = loc == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  // targets (and their variances): constant strides on both sides - one pitched copy each
  BATCH_CHECK(hipMemcpy2DAsync(z, sizeof(double) * (size_t)np2, y, sizeof(double) * (size_t)(ldy ? ldy : n), sizeof(double) * (size_t)n,
                               (size_t)(ldy ? count : 1), kind, s));
  if (!ldy)
    for (int b = 1; b < count; ++b)  // (one target vector shared by all problems)
      BATCH_CHECK(hipMemcpyAsync(z + (size_t)b * (size_t)np2, y, sizeof(double) * (size_t)n, kind, s));
  if (y_var) {
    BATCH_CHECK(hipMemcpy2DAsync(yvar_d, sizeof(double) * (size_t)np2, y_var, sizeof(double) * (size_t)(ldv ? ldv : n), sizeof(double) * (size_t)n,
                                 (size_t)(ldv ? count : 1), kind, s));
    if (!ldv)
      for (int b = 1; b < count; ++b)
        BATCH_CHECK(hipMemcpyAsync(yvar_d + (size_t)b * (size_t)np2, y_var, sizeof(double) * (size_t)n, kind, s));
  }
  // train_features = features (gp.hpp:63): a copy per fit, inside the batch's allocation (no allocation per problem).
  // Device-resident inputs: ONE table-driven copy launch for all problems (pub.h: CopyItem) instead of a copy kernel per array
  std::vector<CopyItem> copies;
  long long copy_max = 0;
  auto stage = [&](double *dst, const void *src, long long words) -> hipError_t {
    if (loc == AGP_HOST) return hipMemcpyAsync(dst, src, sizeof(double) * (size_t)words, kind, s);
    copies.push_back(CopyItem{reinterpret_cast<unsigned long long *>(dst), static_cast<const unsigned long long *>(src), words});
    if (words > copy_max) copy_max = words;
    return hipSuccess;
  };
  double *fcur = base + head_elems;
  for (int b = 0; b < count; ++b) {
    agp_fit *fit = new (std::nothrow) agp_fit();
    if (!fit) return fail(AGP_ERR_INVALID_ARGUMENT);
    fits[(size_t)b] = fit;
    const agp_features *f = features[b];
    FeatView v;
    v.n = n; v.dim = f->dim; v.nsc = f->n_scale_columns; v.meas = 0;
    v.coords = fcur; v.ids = nullptr; v.scales = nullptr;
    BATCH_CHECK(stage(fcur, f->coords, n * (long long)f->dim));
    fcur += (size_t)n * (size_t)f->dim;
    if (f->eq_id) {
      v.ids = reinterpret_cast<const long long *>(fcur);
      BATCH_CHECK(stage(fcur, f->eq_id, n));
      fcur += n;
    }
    if (f->n_scale_columns > 0) {
      v.scales = fcur;
      BATCH_CHECK(stage(fcur, f->scales, n * (long long)f->n_scale_columns));
      fcur += (size_t)n * (size_t)f->n_scale_columns;
    }
    fit->train.v = v;  // (not owned: DeviceFeatures::release has nothing to free)
  }
  // device scratch behind the features: the copy table and the Gram table of the batched launches
  void *tables = nullptr;
  const size_t copy_bytes = (sizeof(CopyItem) * copies.size() + 15) / 16 * 16;
  const size_t gram_bytes = (gram_batch_table_bytes(count) + 15) / 16 * 16;
  // (fused panel launches for batches whose workgroups fit on the chip at once: they publish z through a slab of their own)
  const bool lookahead = (double)count * (double)n * (double)n >= 6e7 && n > 2 * NBO;
  const bool fused_panels = !lookahead && batched_fused_fits(ctx, n, count);
  const size_t zpub_bytes = fused_panels ? sizeof(double) * (size_t)count * (size_t)np2 : 0;
  const size_t table_bytes = copy_bytes + gram_bytes + zpub_bytes;
  if (dev_malloc(&tables, table_bytes) != hipSuccess) { (void)hipGetLastError(); tables = nullptr; }
  struct FreeTables { void *p; ~FreeTables() { if (p) (void)dev_free(p); } } free_tables{tables};
  // (both tables are built in the context's pinned staging area: no synchronisation between the uploads and the launches
  // that read them - the two mid-call synchronisations of round 5 were ~4 % of a batch of 256 fits of N = 512)
  char *pinned = tables ? static_cast<char *>(host_stage(ctx, copy_bytes + gram_bytes)) : nullptr;
  double *zpub = (tables && fused_panels) ? reinterpret_cast<double *>(static_cast<char *>(tables) + copy_bytes + gram_bytes) : nullptr;
  if (!copies.empty()) {
    if (tables) {
      const void *src = copies.data();
      if (pinned) { std::memcpy(pinned, copies.data(), sizeof(CopyItem) * copies.size()); src = pinned; }
      BATCH_CHECK(hipMemcpyAsync(tables, src, sizeof(CopyItem) * copies.size(), hipMemcpyHostToDevice, s));
      if (!pinned) BATCH_CHECK(hipStreamSynchronize(s));  // (pageable source)
      launch_copy_table(s, static_cast<const CopyItem *>(tables), (long long)copies.size(), copy_max);
    } else {
      for (const CopyItem &c : copies) BATCH_CHECK(hipMemcpyAsync(c.dst, c.src, sizeof(double) * (size_t)c.words, kind, s));
    }
  }
  if (loc == AGP_HOST) BATCH_CHECK(hipStreamSynchronize(s));
  // information = L^-T z of every problem in ONE launch (solve.hip: backsub_coop_kernel, blockIdx.y = problem) for sizes
  // of few 128-row blocks: its output vectors are the hand-over buffers and enter sentinel-filled
  const bool coop = ctx->tune.backsub_coop && n <= ctx->tune.backsub_coop_max && (n + NB - 1) / NB <= BACKSUB_DIRECT_BLOCKS;
  {
    PrepArgs prep;
    prep.fill(logsum, 0ull, 3 * cp2);  // log sums and flags
    if (coop) prep.sentinel(alpha, count * np2);
    if (zpub) {  // the hand-over buffers of the fused panel launches: every tile image and every z slot of the batch
      prep.sentinel(invd, count * stride_I);
      prep.sentinel(zpub, count * np2);
    }
    launch_prep(s, prep);
  }
  {
    std::vector<FeatView> views((size_t)count);
    std::vector<const DevProgram *> hprogs((size_t)count);
    std::vector<double *> outs((size_t)count);
    std::vector<const double *> diag((size_t)count, nullptr);
    std::vector<int *> nanf((size_t)count);
    for (int b = 0; b < count; ++b) {
      views[(size_t)b] = fits[(size_t)b]->train.v;
      views[(size_t)b].meas = 1;  // as_measurements(features), gp.hpp:288
      hprogs[(size_t)b] = &kernels[b]->prog;
      outs[(size_t)b] = A + (size_t)b * (size_t)stride_A;
      if (y_var) diag[(size_t)b] = yvar_d + (size_t)b * (size_t)np2;
      nanf[(size_t)b] = flags + 4 * b;
    }
    bool gram_done = false;
    if (tables && count > 1) {  // all Gram matrices in ONE launch when the trees share a fast path (gram.hip)
      gram_done = launch_gram_batch(s, count, hprogs.data(), views.data(), outs.data(), lda, y_var ? diag.data() : nullptr, nanf.data(),
                                    static_cast<char *>(tables) + copy_bytes, pinned ? pinned + copy_bytes : nullptr);
    }
    for (int b = 0; b < count && !gram_done; ++b) {
      const DevProgram *dprog = nullptr;
      if ((st = device_program(ctx, kernels[b], &dprog)) != AGP_OK) return fail(st);
      launch_gram(s, dprog, views[(size_t)b], views[(size_t)b], true, true, outs[(size_t)b], lda, diag[(size_t)b], nanf[(size_t)b],
                  &kernels[b]->prog);
    }
  }
  // (two streams once the trailing updates of the batch are long enough to hide the panel chain behind)
  if (lookahead)
    factor_lower_batched_lookahead(ctx, A, stride_A, n, lda, invd, stride_I, z, np2, count, flags, logsum, 4);
  else
    factor_lower_batched(s, A, stride_A, n, lda, invd, stride_I, z, np2, count, flags, logsum, 4, zpub, np2);
  // information = L^-T (L^-1 y), gp.hpp:68
  if (coop) {
    backward_solve_coop(s, A, n, lda, invd, z, alpha, flags, nullptr, count, stride_A, stride_I, np2, np2, 4);
  } else {
    BATCH_CHECK(hipMemcpyAsync(alpha, z, sizeof(double) * (size_t)count * (size_t)np2, hipMemcpyDeviceToDevice, s));
    backward_solve_vec_batched(s, A, stride_A, n, lda, invd, stride_I, alpha, np2, count);
  }
  std::vector<double> h_log((size_t)cp2);
  std::vector<int> h_flags(4 * (size_t)count);
  BATCH_CHECK(hipMemcpyAsync(h_log.data(), logsum, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, s));
  BATCH_CHECK(hipMemcpyAsync(h_flags.data(), flags, sizeof(int) * 4 * (size_t)count, hipMemcpyDeviceToHost, s));
  BATCH_CHECK(hipStreamSynchronize(s));
  BATCH_CHECK(hipGetLastError());
  for (int b = 0; b < count; ++b)
    if (h_flags[4 * (size_t)b + 2]) {  // a hand-over of the one-launch substitution timed out (its producer died)
      ctx->last_error = "batched back substitution: hand-over timed out";
      return fail(AGP_ERR_HIP);
    }
  // the information vectors of the good fits go to the caller BEFORE any handle is published: a failed copy must not leave
  // the caller with an error code AND live handles (a failed problem leaves its column untouched)
  if (information)
    for (int b = 0; b < count; ++b) {
      const int *fl = &h_flags[4 * (size_t)b];
      if (!fl[0] && !fl[1])
        BATCH_CHECK(hipMemcpy(information + (size_t)b * (size_t)ldi, alpha + (size_t)b * (size_t)np2, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
    }
  agp_fit_slab *slab = new (std::nothrow) agp_fit_slab();
  if (!slab) return fail(AGP_ERR_INVALID_ARGUMENT);
  slab->base = base;
  slab->bytes = sizeof(double) * elems;
  slab->refs = count;
  for (int b = 0; b < count; ++b) {
    agp_fit *fit = fits[(size_t)b];
    fit->ctx = ctx;
    fit->slab = slab;
    fit->device = ctx->device;
    fit->n = n;
    fit->lda = lda;
    fit->A_bytes = sizeof(double) * (size_t)stride_A;
    fit->A = A + (size_t)b * (size_t)stride_A;
    fit->invd = invd + (size_t)b * (size_t)stride_I;
    fit->alpha = alpha + (size_t)b * (size_t)np2;
    fit->z = z + (size_t)b * (size_t)np2;
    const int *fl = &h_flags[4 * (size_t)b];
    fit->failed_pivot = fl[1] ? (int64_t)fl[1] - 1 : -1;
    fit->log_det = 2. * h_log[(size_t)b];
    status[b] = fl[0] ? AGP_ERR_NAN_INPUT : (fl[1] ? AGP_ERR_NOT_POSITIVE_DEFINITE : AGP_OK);  // gp.hpp:66, then the factor
    if (log_det) log_det[b] = fit->log_det;
    out[b] = fit;
  }
#undef BATCH_CHECK
  return AGP_OK;
}

// ---- solve -----------------------------------------------------------------
int agp_solve(agp_context *ctx, const agp_fit *fit, const double *rhs, int64_t nrhs, double *out, int location) {
  if (!ctx || !fit || !rhs || !out || nrhs < 0) return AGP_ERR_INVALID_ARGUMENT;
  if (nrhs == 0) return AGP_OK;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const long long n = fit->n, ldb = round_up(n, 2);
  const hipMemcpyKind kind = location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  if (!fit->phantom.empty()) {
    // rhs / out hold the REAL rows: spread them over the padded rows (phantom rows zero), solve, gather
    const long long nr = fit_real_rows(fit);
    double *B = nullptr;
    AGP_HIP_CHECK(ctx, hipMalloc(&B, sizeof(double) * (size_t)ldb * (size_t)nrhs));
    int st = fit_expand_matrix(ctx, fit, rhs, nr, nrhs, B, ldb, location);
    if (st == AGP_OK) {
      forward_solve_mat_lookahead(ctx, fit->A, n, fit->lda, fit->invd, B, nrhs, ldb);
      backward_solve_mat(ctx->stream, fit->A, n, fit->lda, fit->invd, B, nrhs, ldb);
      st = fit_compact_matrix(ctx, fit, B, ldb, nrhs, out, nr, location);
    }
    (void)hipStreamSynchronize(ctx->stream);
    (void)hipFree(B);
    return st;
  }
  if (nrhs == 1 && n >= 1024) {
    // one right-hand side: the matrix kernels would run 2 N / 128 launches on a single column; the vector chains
    // (fused 128-row forward steps, blocked backward substitution) are 4x shorter (N = 16384: 11.3 -> 2.7 ms)
    const long long nblk = (n + NB - 1) / NB;
    const size_t wf = (size_t)nblk * NB * NB;
    int st1 = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * ((size_t)ldb + wf + backsolve_ws_elems(n)));
    if (st1 != AGP_OK) return st1;
    double *z = ctx->ws_aux, *Wfwd = z + ldb, *ws = Wfwd + wf;
    hipStream_t s = ctx->stream;
    AGP_HIP_CHECK(ctx, hipMemcpyAsync(z, rhs, sizeof(double) * (size_t)n, kind, s));
    if (location == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(s));
    if (const long long BW = backsolve_width(n)) {
      // both directions through 512-wide inverted diagonal blocks (ws holds the backward solve's staging + inverses;
      // the inverses are shared)
      double *xs = ws, *W = ws + round_up(n, 2);
      invert_wide_blocks(s, fit->A, n, fit->lda, fit->invd, BW, W);
      forward_solve_vec_wide(s, fit->A, n, fit->lda, W, BW, z, xs);
      backward_solve_vec_wide(s, fit->A, n, fit->lda, W, BW, z, xs);
      return copy_out(ctx, z, n, out, location);
    }
    invert_diag_blocks_forward(s, n, fit->invd, Wfwd);
    forward_solve_vec(s, fit->A, n, fit->lda, Wfwd, z, ws);  // ws[0 : n] as the staging vector
    backward_solve_vec_any(s, fit->A, n, fit->lda, fit->invd, z, ws);
    return copy_out(ctx, z, n, out, location);
  }
  int st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)ldb * (size_t)nrhs);
  if (st != AGP_OK) return st;
  AGP_HIP_CHECK(ctx, hipMemcpy2DAsync(ctx->ws_aux, sizeof(double) * (size_t)ldb, rhs, sizeof(double) * (size_t)n,
                                      sizeof(double) * (size_t)n, (size_t)nrhs, kind, ctx->stream));
  if (location == AGP_HOST) AGP_HIP_CHECK(ctx, hipStreamSynchronize(ctx->stream));
  forward_solve_mat_lookahead(ctx, fit->A, n, fit->lda, fit->invd, ctx->ws_aux, nrhs, ldb);
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
    AGP_HIP_CHECK(ctx, dev_malloc(&fit->A, fit->A_bytes));
  }
  AGP_HIP_CHECK(ctx, dev_malloc(&fit->invd, sizeof(double) * (size_t)nblk * (36 * MB * MB)));
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
  forward_solve_mat_lookahead(ctx, fit->A, n, fit->lda, fit->invd, R, n, ldr, /*rhs_lower=*/true);
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
  if (!fit->phantom.empty()) return fit_compact_vector(ctx, fit, diag, out, out_location);
  return copy_out(ctx, diag, fit->n, out, out_location);
}

int agp_loo_marginal(agp_context *ctx, const agp_fit *fit, const double *y, double *mean, double *variance,
                     int location) {
  if (!ctx || !fit || !y || !mean || !variance || !fit->A || !fit->alpha) return AGP_ERR_INVALID_ARGUMENT;
  if (!fit->phantom.empty()) return AGP_ERR_UNSUPPORTED;  // cross validation of a fit grown by agp_fit_update: refit instead
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
  // cross_cov = cov(train_features, features); mean = cross_cov^T information  (gp.hpp:361-363)
  if (out_location == AGP_DEVICE) {  // straight into the caller's buffer: no staging copy
    launch_predict_mean(ctx->stream, dprog, fit->train.v, dxs.v, fit->alpha, mean, &k->prog);
    if (hipStreamSynchronize(ctx->stream) != hipSuccess) { ctx->last_error = "agp_predict_mean: stream"; st = AGP_ERR_HIP; }
  } else {
    st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (size_t)m);
    if (st == AGP_OK) {
      launch_predict_mean(ctx->stream, dprog, fit->train.v, dxs.v, fit->alpha, ctx->ws_aux, &k->prog);
      st = copy_out(ctx, ctx->ws_aux, m, mean, out_location);
    }
  }
  dxs.release();
  return st;
}

// rows [o, o + cnt) of a device feature view (scale columns keep the stride of the whole vector)
static FeatView feature_range(const FeatView &v, long long o, long long cnt) {
  FeatView r = v;
  r.sstride = scale_stride(v);
  r.n = cnt;
  r.coords = v.coords + o * v.dim;
  r.ids = v.ids ? v.ids + o : nullptr;
  r.scales = v.scales ? v.scales + o : nullptr;
  return r;
}

// Test points per pass of a marginal prediction: the n x m block L^-1 K* is the only large buffer, and nothing couples
// the columns of a marginal prediction, so m is cut into passes that keep it at 2 GiB (AGP_PREDICT_CHUNK=<points>
// overrides; a joint prediction needs all columns at once).
static long long marginal_chunk(const agp_context *ctx, long long rows) {
  if (ctx->tune.predict_chunk > 0) return ctx->tune.predict_chunk;
  const long long c = (1LL << 28) / (rows > 0 ? rows : 1);
  return c < 1024 ? 1024 : (c > (1LL << 20) ? (1LL << 20) : c);  // (a million points: the Gram kernels' grids stay in range)
}

static int predict_common(agp_context *ctx, const agp_kernel *k, const agp_fit *fit, const agp_features *xs,
                          double *mean, double *var_or_cov, bool joint, int out_location) {
  if (!ctx || !k || !fit || !xs || !mean || !var_or_cov) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(xs);
  if (st != AGP_OK) return st;
  if (!fit->alpha || xs->dim != fit->train.v.dim) return AGP_ERR_INVALID_ARGUMENT;
  const long long m_all = xs->n, n = fit->n;
  if (m_all == 0) return AGP_OK;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dxs;
  if ((st = to_device(ctx, xs, false, &dxs)) != AGP_OK) return st;
  TraceRange tr_predict(joint ? "agp: predict joint (gp.hpp:103-113)" : "agp: predict marginal (gp.hpp:87-101)");
  const long long ldv = round_up(n, 2);
  const long long chunk = joint ? m_all : std::min(m_all, marginal_chunk(ctx, ldv));
  // workspace: V (n x chunk) | mean (chunk) | prior (chunk, or m x m for a joint prediction)
  const long long ldc = round_up(chunk, 2);
  const size_t v_elems = (size_t)ldv * (size_t)chunk;
  const size_t p_elems = joint ? (size_t)ldc * (size_t)chunk : (size_t)ldc;
  st = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * (v_elems + (size_t)ldc + p_elems));
  if (st != AGP_OK) { dxs.release(); return st; }
  double *V = ctx->ws_aux, *mean_d = V + v_elems, *prior = mean_d + ldc;
  hipStream_t s = ctx->stream;
  for (long long o = 0; o < m_all && st == AGP_OK; o += chunk) {
    const long long m = std::min(chunk, m_all - o);
    const FeatView xv = (o == 0 && m == m_all) ? dxs.v : feature_range(dxs.v, o, m);
    // mean (gp.hpp:82-85)
    launch_predict_mean(s, dprog, fit->train.v, xv, fit->alpha, mean_d, &k->prog);
    // cross_cov = cov(train_features, features)   (gp.hpp:316,337)
    launch_gram(s, dprog, fit->train.v, xv, false, false, V, ldv, nullptr, nullptr, &k->prog);
    fit_zero_phantom_rows(s, fit, V, ldv, m);
    // V = L^-1 K*  ;  explained = V^T V  (== K*^T K^-1 K*, gp.hpp:96,111)
    if (m == 1 && n >= 1024) {
      // a single test point: the vector chain (one fused launch per 128 rows) instead of the matrix kernels
      const long long nblk = (n + NB - 1) / NB;
      double *Wfwd = nullptr, *stage = nullptr;
      const long long BWp = backsolve_width(n);
      const size_t w_elems = BWp ? (size_t)(n / BWp) * (size_t)BWp * (size_t)BWp : (size_t)nblk * NB * NB;
      if (hipMalloc(&Wfwd, sizeof(double) * (w_elems + (size_t)round_up(n, 2))) != hipSuccess) {
        dxs.release();
        ctx->last_error = "hipMalloc (single-point prediction workspace)";
        return AGP_ERR_HIP;
      }
      stage = Wfwd + w_elems;
      if (BWp) {  // through 512-wide inverted diagonal blocks (see agp_solve)
        invert_wide_blocks(s, fit->A, n, fit->lda, fit->invd, BWp, Wfwd);
        forward_solve_vec_wide(s, fit->A, n, fit->lda, Wfwd, BWp, V, stage);
      } else {
        invert_diag_blocks_forward(s, n, fit->invd, Wfwd);
        forward_solve_vec(s, fit->A, n, fit->lda, Wfwd, V, stage);
      }
      (void)hipStreamSynchronize(s);
      (void)hipFree(Wfwd);
    } else {
      forward_solve_mat_lookahead(ctx, fit->A, n, fit->lda, fit->invd, V, m, ldv);
    }
    if (!joint) {
      launch_gram_diagonal(s, dprog, xv, prior);                      // gp.hpp:339-343
      launch_coldot(s, V, ldv, V, ldv, n, m, prior, 1.0, prior);      // gp.hpp:97-99
      st = copy_out(ctx, mean_d, m, mean + o, out_location);
      if (st == AGP_OK) st = copy_out(ctx, prior, m, var_or_cov + o, out_location);
    } else {
      launch_gram(s, dprog, xv, xv, true, false, prior, ldc, nullptr, nullptr, &k->prog);  // prior_cov, gp.hpp:317
      launch_gemm_nt_sub(s, prior, ldc, V, ldv, true, V, ldv, true, m, m, n, true);   // lower tiles
      launch_symmetrize(s, prior, ldc, m);
      st = copy_out(ctx, mean_d, m, mean, out_location);
      if (st == AGP_OK) st = copy_out_2d(ctx, prior, ldc, m, m, var_or_cov, m, out_location);
    }
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
