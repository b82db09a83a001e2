// shard_hip.hip — the GPU half of the sharded fit (shard.h): HipShardOps (the block arithmetic on HIP streams with
// the kernels of the single-GPU fit), the RCCL transport, and the C-ABI entry points agp_comm_* / agp_sharded_fit_*.
//
// Reference work replaced: GaussianProcessBase::_fit_impl (include/albatross/src/models/gp.hpp:281-294) + the
// Fit<GPFit> constructor (gp.hpp:61-69) for one dataset over the GPUs of a node.
//
// RCCL is bound at run time (dlopen of the ROCm installation's librccl.so.1, the one built against the HIP runtime
// this library links): processes that never create a communicator do not load it, and a torch wheel's private
// librccl elsewhere in the process is never picked up by accident.
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only; every function goes through the table below

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <thread>

#include "api_internal.h"
#include "shard_internal.h"
#include "trace.h"

namespace agp {
void panel_phase_public(agp_context *ctx, hipStream_t s, double *A, long long n, long long lda, double *img, double *y,
                        long long K0, long long kend);
void trsm_rows_wide(hipStream_t s, double *X, long long ld, long long nrows, long long w, const double *Lkk, long long ldl,
                    const double *img, const double *z, double *yrows);

// ---------------------------------------------------------------------------------------------------------------
// RCCL, bound at run time
// ---------------------------------------------------------------------------------------------------------------
struct RcclApi {
  void *handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommAbort)(ncclComm_t) = nullptr;
  ncclResult_t (*CommGetAsyncError)(ncclComm_t, ncclResult_t *) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*Broadcast)(const void *, void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  std::string error;
};

static RcclApi *rccl_api() {
  static RcclApi api;
  static bool tried = false;
  if (tried) return api.handle ? &api : nullptr;
  tried = true;
  const char *names[] = {getenv("AGP_RCCL_LIB"), "/opt/rocm/lib/librccl.so.1", "librccl.so.1", "librccl.so"};
  for (const char *nm : names) {
    if (!nm || !nm[0]) continue;
    api.handle = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
    if (api.handle) break;
    api.error = dlerror();
  }
  if (!api.handle) return nullptr;
#define AGP_BIND(field, sym)                                                  \
  api.field = reinterpret_cast<decltype(api.field)>(dlsym(api.handle, sym)); \
  if (!api.field) { api.error = std::string("missing symbol ") + sym; dlclose(api.handle); api.handle = nullptr; return nullptr; }
  AGP_BIND(GetUniqueId, "ncclGetUniqueId")
  AGP_BIND(CommInitRank, "ncclCommInitRank")
  AGP_BIND(CommDestroy, "ncclCommDestroy")
  AGP_BIND(CommAbort, "ncclCommAbort")
  AGP_BIND(CommGetAsyncError, "ncclCommGetAsyncError")
  AGP_BIND(GetErrorString, "ncclGetErrorString")
  AGP_BIND(Broadcast, "ncclBroadcast")
  AGP_BIND(AllGather, "ncclAllGather")
  AGP_BIND(AllReduce, "ncclAllReduce")
#undef AGP_BIND
  return &api;
}

double comm_timeout_seconds() {
  static double t = -1.;
  if (t < 0.) {
    const char *e = getenv("AGP_COMM_TIMEOUT_S");
    t = e ? atof(e) : 120.;
    if (!(t > 0.)) t = 120.;
  }
  return t;
}

// Wait for a stream with a deadline: a peer that died inside a collective must not hang this rank for ever.
static int wait_stream(agp_context *ctx, hipStream_t s, double timeout_s) {
  const auto t0 = std::chrono::steady_clock::now();
  int spins = 0;
  while (true) {
    const hipError_t e = hipStreamQuery(s);
    if (e == hipSuccess) return AGP_OK;
    if (e != hipErrorNotReady) {
      if (ctx) ctx->last_error = std::string("hipStreamQuery: ") + hipGetErrorString(e);
      return AGP_ERR_HIP;
    }
    if (++spins > 2000) std::this_thread::sleep_for(std::chrono::microseconds(50));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
      if (ctx) ctx->last_error = "timeout waiting for the device (a collective did not complete)";
      return AGP_ERR_COMM;
    }
  }
}

struct RcclComm : HostReducingComm {
  RcclApi *api = nullptr;
  ncclComm_t comm = nullptr;
  agp_context *ctx = nullptr;
  double *scratch = nullptr;  // device staging of the host-side all-reduce
  static constexpr long long SCRATCH = 4096;
  bool broken = false;

  ~RcclComm() override {
    if (ctx) (void)hipSetDevice(ctx->device);
    if (scratch) (void)hipFree(scratch);
    if (comm && api) {
      if (broken) (void)api->CommAbort(comm);
      else (void)api->CommDestroy(comm);
    }
  }
  void mark_broken() override { broken = true; }
  int check(ncclResult_t r, const char *what) {
    if (r == ncclSuccess) return AGP_OK;
    if (ctx) ctx->last_error = std::string(what) + ": " + api->GetErrorString(r);
    broken = true;
    return AGP_ERR_COMM;
  }
  int broadcast(ShardOps &ops, int q, double *buf, long long count, int root) override {
    return check(api->Broadcast(buf, buf, (size_t)count, ncclDouble, root, comm, (hipStream_t)ops.stream(q)), "ncclBroadcast");
  }
  int all_gather(ShardOps &ops, int q, const double *send, double *recv, long long count) override {
    return check(api->AllGather(send, recv, (size_t)count, ncclDouble, comm, (hipStream_t)ops.stream(q)), "ncclAllGather");
  }
  int all_reduce(ShardOps &ops, int q, double *buf, long long count, int op) override {
    return check(api->AllReduce(buf, buf, (size_t)count, ncclDouble, op == 1 ? ncclMax : ncclSum, comm,
                                (hipStream_t)ops.stream(q)), "ncclAllReduce");
  }
  int all_reduce_host(double *buf, long long count, int op) override {
    if (hipSetDevice(ctx->device) != hipSuccess) return AGP_ERR_HIP;
    hipStream_t s = ctx->stream3;
    for (long long off = 0; off < count; off += SCRATCH) {
      const long long c = count - off < SCRATCH ? count - off : SCRATCH;
      if (hipMemcpyAsync(scratch, buf + off, sizeof(double) * (size_t)c, hipMemcpyHostToDevice, s) != hipSuccess) return AGP_ERR_HIP;
      const int st = check(api->AllReduce(scratch, scratch, (size_t)c, ncclDouble, op == 1 ? ncclMax : ncclSum, comm, s), "ncclAllReduce");
      if (st != AGP_OK) return st;
      if (hipMemcpyAsync(buf + off, scratch, sizeof(double) * (size_t)c, hipMemcpyDeviceToHost, s) != hipSuccess) return AGP_ERR_HIP;
      const int sw = wait_stream(ctx, s, comm_timeout_seconds());
      if (sw != AGP_OK) { broken = true; return sw; }
    }
    return AGP_OK;
  }
};

// ---------------------------------------------------------------------------------------------------------------
// kernels of the HIP backend that the single-GPU path does not have
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void shard_copy2d_kernel(double *__restrict__ dst, long long ldd, const double *__restrict__ src,
                                                           long long lds, long long rows, long long cols) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r >= rows) return;
  for (long long c = blockIdx.y; c < cols; c += gridDim.y) dst[r + c * ldd] = src[r + c * lds];
}

// Pall[(i - k - 1) B + r][c] = recv[owner(i)][c][(li(i) - li0(owner, k)) B + r]: the all-gathered per-rank stacks back
// into global row order (snake ownership, see ShardPlan)
__global__ __launch_bounds__(256) void shard_gather_kernel(double *__restrict__ Pall, long long ldP, const double *__restrict__ recv,
                                                           long long cnt_rows, long long w, long long n, long long B, int world,
                                                           long long k) {
  const long long row = (long long)blockIdx.x * 256 + threadIdx.x;  // row of Pall
  const long long g = (k + 1) * B + row;                            // global row
  if (g >= n) return;
  const long long i = g / B, r = g - i * B;
  const long long rr = i % world, rnd = i / world;
  const int o = (int)((rnd & 1) ? world - 1 - rr : rr);
  long long li0 = (k + 1) / world;
  const long long gb = li0 * world + ((li0 & 1) ? world - 1 - o : o);
  if (gb <= k) ++li0;
  const double *src = recv + (long long)o * cnt_rows * w + (rnd - li0) * B + r;
  for (long long c = blockIdx.y; c < w; c += gridDim.y) Pall[row + c * ldP] = src[c * cnt_rows];
}

// the broadcast message of a diagonal block in ONE launch: L (w x w, ld = w) | tile images | z
__global__ __launch_bounds__(256) void shard_pack_msg_kernel(double *__restrict__ msg, long long B, const double *__restrict__ D,
                                                            long long ld, long long w, const double *__restrict__ img,
                                                            const double *__restrict__ z) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i < w * w) msg[i] = D[(i % w) + (i / w) * ld];
  if (i < 4 * SHARD_IMG) msg[B * B + i] = img[i];
  if (i < w) msg[B * B + 4 * SHARD_IMG + i] = z[i];
}

// ---- device-side pacing (shard.h: ShardOps::record / wait) ----
// record: ONE thread stores the record's sequence number behind everything enqueued on its stream so far (the kernel
// boundary before it has made the producers' writes visible device-wide; the store is a device-scope release).
__global__ void shard_signal_kernel(unsigned long long *flag, unsigned long long value) {
  __hip_atomic_store(flag, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// wait: one wave polls the flag until it has reached `need` (sequence numbers only grow) and ends; what follows on its
// stream starts behind it.  The spin is bounded (s_memrealtime ticks at 100 MHz): a producer that never comes - a dead
// peer inside a collective - raises *timeout_flag and lets the stream run on, so that the host's own deadline turns the
// fit into AGP_ERR_COMM instead of a hung GPU.
__global__ void shard_gate_kernel(const unsigned long long *flag, unsigned long long need, unsigned long long timeout_ticks,
                                  int *timeout_flag) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < need) {
    __builtin_amdgcn_s_sleep(8);
    if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
      if (timeout_flag) atomicExch(timeout_flag, 1);
      return;
    }
  }
}

// Consecutive records / waits on one stream travel as ONE launch: up to SHARD_MICRO_MAX operations executed in order by one
// thread (a dependent launch costs 2-5 us on its stream, more next to a bulk update that fills the chip).
constexpr int SHARD_MICRO_MAX = 6;
struct ShardMicroOps {
  unsigned long long *flag[SHARD_MICRO_MAX];
  unsigned long long value[SHARD_MICRO_MAX];
  int wait[SHARD_MICRO_MAX];  // 1: spin until *flag >= value, 0: store value
  int count;
};
__global__ void shard_micro_kernel(ShardMicroOps ops, unsigned long long timeout_ticks, int *timeout_flag) {
  if (threadIdx.x != 0) return;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < ops.count; ++i) {
    if (!ops.wait[i]) {
      __hip_atomic_store(ops.flag[i], ops.value[i], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      continue;
    }
    while (__hip_atomic_load(ops.flag[i], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < ops.value[i]) {
      __builtin_amdgcn_s_sleep(4);
      if (__builtin_amdgcn_s_memrealtime() - t0 > timeout_ticks) {
        if (timeout_flag) atomicExch(timeout_flag, 1);
        return;  // (the stores behind a wait that gave up are not made: their consumers run into their own deadline)
      }
    }
  }
}

__global__ __launch_bounds__(256) void shard_add_diag_kernel(double *A, long long ld, long long lrow0, long long gcol0, long long w,
                                                            const double *yvar) {
  const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
  if (r < w) A[(lrow0 + r) + (gcol0 + r) * ld] += yvar[gcol0 + r];
}

// full[g][c] (lower triangle, ldf) <- row-block stacks of all ranks (each nlb_max * B rows x n, ld_loc)
__global__ __launch_bounds__(256) void shard_unstack_kernel(double *__restrict__ full, long long ldf, const double *__restrict__ stacks,
                                                            long long ld_loc, long long per_rank, long long n, long long B, int world) {
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  const long long i = g / B, r = g - i * B;
  const long long rr = i % world, rnd = i / world;
  const int o = (int)((rnd & 1) ? world - 1 - rr : rr);
  const double *src = stacks + (long long)o * per_rank + rnd * B + r;
  for (long long c = blockIdx.y; c <= g; c += gridDim.y) full[g + c * ldf] = src[c * ld_loc];
}

// ---------------------------------------------------------------------------------------------------------------
// HipShardOps
// ---------------------------------------------------------------------------------------------------------------
struct HipShardOps : ShardOps {
  agp_context_impl *ctx;
  hipStream_t sq[3];
  hipEvent_t ev[EV_COUNT];
  bool ok = true;
  double timeout_s;
  // DEVICE pacing (default): one flag per event in device memory, sequence numbers that only grow (they live in the
  // context and continue from fit to fit: nothing to reset).  HOST pacing: HIP events (AGP_SHARD_HOST_PACING=1, or the
  // probe below found two of the queues sharing a hardware queue).
  bool device_pacing = true;
  unsigned long long *flags = nullptr;  // ctx->shard_flags: [EV_COUNT] events | [EV_COUNT ..] probe
  unsigned long long timeout_ticks = 0;
  // bulk-update timing (profiling only)
  std::vector<hipEvent_t> tev;
  std::vector<double> tflop;
  size_t tused = 0;

  explicit HipShardOps(agp_context_impl *c) : ctx(c), timeout_s(comm_timeout_seconds()) {
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // the collectives' queue: created with the first sharded call of the context and kept (creating and destroying a
    // stream costs ~0.4 ms per fit, sometimes tens of ms; a plain fit afterwards is as fast as before: profiles/r04)
    if (!ctx->stream_comm) ok = hipStreamCreateWithPriority(&ctx->stream_comm, hipStreamNonBlocking, hi) == hipSuccess;
    sq[QC] = ctx->stream_comm;
    sq[QP] = ctx->stream; sq[QB] = ctx->stream2;
    for (auto &e : ev) e = nullptr;
    timeout_ticks = (unsigned long long)(timeout_s * 1e8);
    if (ctx->shard_host_pacing < 0) ctx->shard_host_pacing = ctx->tune.shard_host_pacing ? 1 : 0;
    device_pacing = ok && ctx->shard_host_pacing == 0;
    if (device_pacing) {
      if (!ctx->shard_flags) {
        if (hipMalloc(&ctx->shard_flags, sizeof(unsigned long long) * 2 * EV_COUNT) != hipSuccess ||
            hipMemset(ctx->shard_flags, 0, sizeof(unsigned long long) * 2 * EV_COUNT) != hipSuccess) {
          (void)hipGetLastError();
          if (ctx->shard_flags) (void)hipFree(ctx->shard_flags);
          ctx->shard_flags = nullptr;
        }
      }
      flags = ctx->shard_flags;
      device_pacing = flags != nullptr;
    }
  }
  // first record / wait of this object: settle the pacing mode (entry points that only borrow the queues never pay for it)
  bool decided = false;
  void decide() {
    if (decided) return;
    decided = true;
    if ((device_pacing || allow_device) && !ctx->shard_probe_ok) {  // the queues are the context's own: the answer holds for its lifetime
      // (processes that share one GPU are time-sliced: a healthy gate can miss the short deadline while its producer's
      // process is not running - one more round with a long deadline before the queues are declared aliased; on a GPU of
      // its own the first round passes in ~40 us)
      const bool fine = probe_queues(25000000ull) || probe_queues(200000000ull);
      if (fine) ctx->shard_probe_ok = true;
      else { ctx->shard_host_pacing = 1; device_pacing = allow_device = false; }
    }
    if (!device_pacing)
      for (auto &e : ev)
        if (!e) ok = ok && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
  }
  ~HipShardOps() override {
    for (auto e : ev) if (e) (void)hipEventDestroy(e);
    for (auto e : tev) (void)hipEventDestroy(e);
  }
  // A gate kernel must never sit in FRONT of its producer in a hardware queue: the runtime maps HIP streams onto a few
  // hardware queues (GPU_MAX_HW_QUEUES) and serialises streams that share one.  One round of gates and signals between
  // every pair of queues the schedule pairs up, with a short deadline: if any gate runs into it, the queues alias and
  // this fit is paced by the host.  ~40 us when all is well.
  bool probe_queues(const unsigned long long short_ticks /* 100 MHz: 25000000 = 250 ms */) {
    unsigned long long *pf = flags + EV_COUNT;
    int *fail = ctx->d_flags + 3;
    unsigned long long &seq = ctx->shard_probe_seq;
    (void)hipMemsetAsync(fail, 0, sizeof(int), sq[QC]);
    if (hipStreamSynchronize(sq[QC]) != hipSuccess) return false;
    // every {consumer, producer} pair the schedule uses - the panel chain also gates on records of the bulk queue (the
    // previous step's U2) and, on one rank, the bulk queue on the chain - and the same pairs against the CU-MASKED stream,
    // which becomes the bulk queue in the chain-bound regime (to_masked_bulk): the answer is cached for the context's
    // lifetime, so it has to cover every stream a fit may put behind `QB`
    hipStream_t qs[4] = {sq[QC], sq[QP], sq[QB], (ctx->stream_masked && ctx->stream_masked != sq[QB]) ? ctx->stream_masked : nullptr};
    int slot = 0;
    for (int c = 0; c < 4; ++c)
      for (int pr = 0; pr < 4; ++pr) {
        if (c == pr || !qs[c] || !qs[pr]) continue;
        if (c >= 2 && pr >= 2) continue;  // (the two bulk streams never wait for each other inside a regime)
        ++seq;
        hipLaunchKernelGGL(shard_gate_kernel, dim3(1), dim3(64), 0, qs[c], pf + slot, seq, short_ticks, fail);
        hipLaunchKernelGGL(shard_signal_kernel, dim3(1), dim3(1), 0, qs[pr], pf + slot, seq);
        ++slot;
      }
    if (qs[3] && hipStreamSynchronize(qs[3]) != hipSuccess) { (void)hipGetLastError(); return false; }
    int h = 1;
    bool fine = true;
    for (int q = 0; q < 3; ++q) fine = fine && hipStreamSynchronize(sq[q]) == hipSuccess;
    fine = fine && hipMemcpy(&h, fail, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess && h == 0;
    if (!fine) {
      (void)hipGetLastError();
      (void)hipMemset(fail, 0, sizeof(int));
    }
    return fine;
  }
  void *stream(int q) override {
    flush(q); return sq[q]; }
  bool device_memory() const override { return true; }

  void factor_diag(int q, double *D, long long ld, long long w, long long pivot_base, double *img, double *zblk) override {
    flush(q);
    // the panel phase of the single-GPU factorisation on the w x w block alone; the pointers are shifted so that the
    // block sits at row / column `pivot_base` and a non-positive pivot is reported with its GLOBAL index
    panel_phase_public(ctx, sq[q], D - pivot_base * (ld + 1), pivot_base + w, ld, img - (pivot_base / NB) * (long long)SHARD_IMG,
                       zblk - pivot_base, pivot_base, pivot_base + w);
  }
  void trsm_rows(int q, double *X, long long ld, long long nrows, long long w, const double *Lkk, const double *img,
                 const double *z, double *yrows) override {
    flush(q);
    trsm_rows_wide(sq[q], X, ld, nrows, w, Lkk, w, img, z, yrows);
  }
  void gemm(int q, double *C, long long ldc, const double *P, long long ldp, const double *Q, long long ldq, long long M,
            long long N, long long K, bool tri, int bulk) override {
    flush(q);
    if (M <= 0 || N <= 0 || K <= 0) return;
    hipStream_t s = sq[q];
    const bool timed = bulk && ctx->profiling;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed) {
      while (tev.size() < tused + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        tev.push_back(e);
      }
      if (tev.size() >= tused + 2) { e0 = tev[tused]; e1 = tev[tused + 1]; }
    }
    if (bulk && tri && M == N && ldp == ldq) {
      // the single-GPU bulk update: full rounds of 128 x 128 workgroups + the 64 x 64 tail (gemm.hip)
      BulkTiming bt;
      bt.e0 = e0; bt.e1 = e1;
      launch_trailing_update(s, C, ldc, P, Q, ldp, M, K, e0 ? &bt : nullptr);
      if (e0 && bt.flops > 0.) {
        tflop.resize(tused / 2 + 1);
        tflop[tused / 2] = bt.flops;
        tused += 2;
      }
      return;
    }
    if (e0) (void)hipEventRecord(e0, s);
    launch_gemm_nt_sub(s, C, ldc, P, ldp, false, Q, ldq, false, M, N, K, tri);
    if (e0) {
      (void)hipEventRecord(e1, s);
      tflop.resize(tused / 2 + 1);
      tflop[tused / 2] = 2. * (double)K * (double)M * (double)N * (tri ? 0.5 : 1.);
      tused += 2;
    }
  }
  // Two regimes of a sharded fit, told apart by the flop of this rank's bulk update U2(k) of a step:
  //  * BULK-BOUND (U2 is longer than the step's panel chain): U2 on the ordinary bulk stream, all 256 CUs, and the schedule
  //    paced by the HOST.  A bulk update that fills every CU holds all VGPRs and LDS: every launch next to it - the
  //    one-thread record / wait kernels of the device pacing too - waits ~50 us for one of its workgroups to retire
  //    (profiles/r04/timeline_sharded_rccl1_dev.txt: each small kernel 54 us), and the chain is hidden behind U2 anyway.
  //  * CHAIN-BOUND (from step `switch_step` on): U2 on the context's CU-MASKED stream (224 of 256 CUs, api.hip), so that
  //    the panel chain's launches start at once on the 4 free CUs per XCD, and DEVICE pacing - no host in the loop.
  // The switch happens once per fit, at the top of a step, behind ONE drain of the three queues (so no wait ever
  // refers to a record of the other mechanism).  Threshold: 40 GFLOP of U2 per step (AGP_SHARD_MASK_GFLOP), from
  // scripts/sweep_shard_regime.sh on one rank's share (profiles/r04/sweep_shard_regime*.txt): N = 16384, 8 ranks: 9.5 ms
  // against 10.8 (host, unmasked throughout) and 9.6 (device, masked throughout); N = 65536: 258 against 260 and 279.
  // ONE rank running the multi-rank schedule (AGP_SHARD_FORCE_COMM, a test mode: every step an owner step next to a
  // full-size bulk update) stays with the host: 41.5 ms against 45-60.
  bool allow_device = false;   // device pacing is permitted (switch + probe) - used from switch_step on
  long long switch_step = 0;   // first block column of the chain-bound regime
  long long order_switch = 0;  // first block column from which the bulk update of EVERY rank is below the threshold (owner_first)
  bool bulk_masked = false;
  void begin(const ShardPlan &plan) override {
    allow_device = device_pacing;
    switch_step = 0;
    if (!ctx->stream_masked) {  // no masked stream: one regime (device pacing everywhere if permitted)
      switch_step = 0;
    } else {
      const double limit_gflop = plan.world > 1 ? ctx->tune.shard_mask_gflop : 0.;
      const long long B = plan.B, nlb = plan.n_local_blocks(plan.rank);
      for (long long k = 0; k + 2 < plan.nb; ++k) {
        double entries = 0.;
        for (long long li = plan.first_local_after(plan.rank, k + 1); li < nlb; ++li) {
          const long long i = plan.global_block(plan.rank, li);
          entries += (double)plan.width(i) * (double)((i + 1) * B - (k + 2) * B);
        }
        if (2. * (double)plan.width(k) * entries <= limit_gflop * 1e9) break;
        switch_step = k + 1;
      }
    }
    // ... and the step from which EVERY rank's bulk update is that small: the order of the collectives of a step
    // (owner_first) has to be the same on all ranks, so it cannot follow this rank's own regime switch
    order_switch = 0;
    if (ctx->stream_masked || plan.world > 1) {
      const double limit_gflop = plan.world > 1 ? ctx->tune.shard_mask_gflop : 0.;
      const long long Bq = plan.B;
      for (long long k = 0; k + 2 < plan.nb; ++k) {
        double worst = 0.;
        for (int r = 0; r < plan.world; ++r) {
          double entries = 0.;
          const long long nl = plan.n_local_blocks(r);
          for (long long li = plan.first_local_after(r, k + 1); li < nl; ++li) {
            const long long i = plan.global_block(r, li);
            entries += (double)plan.width(i) * (double)((i + 1) * Bq - (k + 2) * Bq);
          }
          if (entries > worst) worst = entries;
        }
        if (2. * (double)plan.width(k) * worst <= limit_gflop * 1e9) break;
        order_switch = k + 1;
      }
    }
    if (switch_step > 0) device_pacing = false;  // host pacing until the switch
    decided = false;
    decide();
    if (switch_step == 0) to_masked_bulk();
  }
  void to_masked_bulk() {
    if (bulk_masked || !ctx->stream_masked) return;
    sq[QB] = ctx->stream_masked;  // (nothing of this fit is on the unmasked stream, or the queues have just been drained)
    bulk_masked = true;
  }
  // chain-bound from switch_step on (begin): the same on every rank - the plan and the threshold are
  bool owner_first(long long k) override { return k >= order_switch; }
  int step_begin(long long k) override {
    if (k != switch_step || k == 0) return AGP_OK;
    const int st = sync_all();  // drain: every record so far has completed
    if (st != AGP_OK) return st;
    to_masked_bulk();
    if (allow_device && !device_pacing) {
      device_pacing = true;
      for (auto &r : recorded) r = false;  // (waits for records of the host-paced steps are satisfied by the drain)
    }
    return AGP_OK;
  }
  void update_staircase(int q, double *A, long long ld, const double *Q, long long ldq, const ShardPlan &plan, long long k) override {
    flush(q);
    // ONE launch over all own row blocks >= k + 2: the tiles right of a row block's own diagonal tile exit at once
    const long long B = plan.B, li2 = plan.first_local_after(plan.rank, k + 1);
    const long long M = plan.local_rows(plan.rank) - li2 * B, c0 = (k + 2) * B, N = plan.n - c0, K = plan.width(k);
    if (M <= 0 || N <= 0) return;
    hipStream_t s = sq[q];
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (ctx->profiling) {
      while (tev.size() < tused + 2) {
        hipEvent_t e;
        if (hipEventCreate(&e) != hipSuccess) break;
        tev.push_back(e);
      }
      if (tev.size() >= tused + 2) { e0 = tev[tused]; e1 = tev[tused + 1]; }
    }
    if (e0) (void)hipEventRecord(e0, s);
    launch_gemm_nt_sub_stair(s, A + li2 * B + c0 * ld, ld, A + li2 * B + k * B * ld, ld, Q, ldq, M, N, K, plan.world, plan.rank, li2, B, c0);
    if (e0) {
      (void)hipEventRecord(e1, s);
      double entries = 0.;  // algorithmic: the entries on / below the diagonal of the own row blocks' columns >= c0
      for (long long li = li2; li < plan.n_local_blocks(plan.rank); ++li) {
        const long long i = plan.global_block(plan.rank, li), wi = plan.width(i);
        entries += (double)wi * (double)(i * B - c0) + 0.5 * (double)wi * (double)(wi + 1);
      }
      tflop.resize(tused / 2 + 1);
      tflop[tused / 2] = 2. * (double)K * entries;
      tused += 2;
    }
  }
  void copy2d(int q, double *dst, long long ldd, const double *src, long long lds, long long rows, long long cols) override {
    flush(q);
    if (rows <= 0 || cols <= 0) return;
    const unsigned gy = (unsigned)(cols < 256 ? cols : 256);
    hipLaunchKernelGGL(shard_copy2d_kernel, dim3((unsigned)((rows + 255) / 256), gy), dim3(256), 0, sq[q], dst, ldd, src, lds, rows, cols);
  }
  void pack_msg(int q, double *msg, long long B, const double *D, long long ld, long long w, const double *img,
                const double *z) override {
    flush(q);
    const long long count = w * w > 4 * SHARD_IMG ? w * w : 4 * SHARD_IMG;
    hipLaunchKernelGGL(shard_pack_msg_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, sq[q], msg, B, D, ld, w, img, z);
  }
  void gather_panel(int q, double *Pall, long long ldP, const double *recv, long long cnt_rows, long long w, const ShardPlan &plan,
                    long long k) override {
    flush(q);
    const long long rows = plan.n - (k + 1) * plan.B;
    if (rows <= 0) return;
    const unsigned gy = (unsigned)(w < 128 ? w : 128);
    hipLaunchKernelGGL(shard_gather_kernel, dim3((unsigned)((rows + 255) / 256), gy), dim3(256), 0, sq[q], Pall, ldP, recv, cnt_rows, w,
                       plan.n, plan.B, plan.world, k);
  }
  void invert_diag(int q, const double *D, long long ld, long long w, const double *img, double *W) override {
    flush(q);
    launch_set_identity_batched(sq[q], W, w, w * w, w, 1);
    forward_solve_mat_batched(sq[q], D, 0, w, ld, img, 0, W, 0, w, w, /*rhs_lower=*/true, 1);
  }
  void invert_diag_batch(int q, const double *D, long long stride_D, long long ld, long long w, const double *img,
                         long long stride_img, double *W, long long stride_W, long long count) override {
    flush(q);
    launch_set_identity_batched(sq[q], W, w, stride_W, w, count);
    forward_solve_mat_batched(sq[q], D, stride_D, w, ld, img, stride_img, W, stride_W, w, w, /*rhs_lower=*/true, count);
  }
  void colvec_dot(int q, const double *W, long long ld, long long m, long long n, const double *v, double alpha, double beta,
                  const double *base, double *out) override {
    flush(q);
    launch_colvec_dot(sq[q], W, ld, m, n, v, alpha, beta, base, out);
  }
  void axpby(int q, long long n, double a, const double *x, double b, const double *y, double *out) override {
    flush(q);
    launch_axpby(sq[q], n, a, x, b, y, out);
  }
  void fill_zero(int q, double *p, long long count) override {
    flush(q);
    if (count > 0) (void)hipMemsetAsync(p, 0, sizeof(double) * (size_t)count, sq[q]);
  }
  int host_spin(int e) {
    const auto t0 = std::chrono::steady_clock::now();
    long long spins = 0;
    while (true) {
      const hipError_t r = hipEventQuery(ev[e]);
      if (r == hipSuccess) return AGP_OK;
      if (r != hipErrorNotReady) {
        ctx->last_error = std::string("hipEventQuery: ") + hipGetErrorString(r);
        return AGP_ERR_HIP;
      }
      if ((++spins & 0xfff) == 0 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > timeout_s) {
        ctx->last_error = "timeout waiting for an event of the sharded schedule (a peer or a stream stalled)";
        return AGP_ERR_COMM;
      }
    }
  }
  void record(int e, int q) override {
    decide();
    if (device_pacing) {
      push(q, flags + e, ++ctx->shard_seq[e], 0);
      recorded[e] = true;
      record_queue[e] = q;
    } else {
      (void)hipEventRecord(ev[e], sq[q]);
    }
  }
  int wait(int q, int e) override {
    decide();
    if (device_pacing) {
      if (recorded[e]) {  // (never recorded by this fit: nothing to wait for, like hipStreamWaitEvent)
        // the record must be ON its stream before anything can wait for it: a host-synchronous transport (callbacks)
        // drains the waiting queue before the host returns to flush the producer's
        if (record_queue[e] != q) flush(record_queue[e]);
        push(q, flags + e, ctx->shard_seq[e], 1);
      }
      return AGP_OK;
    }
    // host pacing: the panel chain may sit at a stream wait, the other queues are fed once their inputs are ready
    if (q == QP) { (void)hipStreamWaitEvent(sq[q], ev[e], 0); return AGP_OK; }
    return host_spin(e);
  }
  bool recorded[EV_COUNT] = {};
  int record_queue[EV_COUNT] = {};
  ShardMicroOps pending[3] = {};
  void push(int q, unsigned long long *flag, unsigned long long value, int is_wait) {
    ShardMicroOps &m = pending[q];
    if (m.count == SHARD_MICRO_MAX) flush(q);
    m.flag[m.count] = flag; m.value[m.count] = value; m.wait[m.count] = is_wait;
    ++m.count;
  }
  // launch the records / waits collected for queue q (before anything else is enqueued on it)
  void flush(int q) {
    ShardMicroOps &m = pending[q];
    if (m.count == 0) return;
    hipLaunchKernelGGL(shard_micro_kernel, dim3(1), dim3(64), 0, sq[q], m, timeout_ticks, ctx->d_flags + 3);
    m.count = 0;
  }
  int sync_all() override {
    for (int q = 0; q < 3; ++q) flush(q);
    for (int q = 0; q < 3; ++q) {
      const int st = wait_stream(ctx, sq[q], timeout_s);
      if (st != AGP_OK) return st;
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) { ctx->last_error = hipGetErrorString(e); return AGP_ERR_HIP; }
    return AGP_OK;
  }
  void status(double out[2]) override {
    int flags[4] = {0, 0, 0, 0};
    double scal[4] = {0., 0., 0., 0.};
    (void)hipMemcpy(flags, ctx->d_flags, sizeof(flags), hipMemcpyDeviceToHost);
    (void)hipMemcpy(scal, ctx->d_scalars, sizeof(scal), hipMemcpyDeviceToHost);
    out[0] = scal[0];
    out[1] = (double)flags[1];
    handover_timeout = flags[2] != 0;
    gate_timeout = flags[3] != 0;
  }
  bool gate_timeout = false;      // a wait of the device-paced schedule ran into the transport's deadline
  bool handover_timeout = false;  // a consumer of the fused panel kernel gave up waiting (chol.hip): the factor is garbage
  int to_host(int q, const double *dev, double *host, long long count) override {
    flush(q);
    if (hipMemcpyAsync(host, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, sq[q]) != hipSuccess) return AGP_ERR_HIP;
    return wait_stream(ctx, sq[q], timeout_s);
  }
  int from_host(int q, const double *host, double *dev, long long count) override {
    flush(q);
    if (hipMemcpyAsync(dev, host, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, sq[q]) != hipSuccess) return AGP_ERR_HIP;
    return wait_stream(ctx, sq[q], timeout_s);  // pageable source: must be consumed before the caller reuses it
  }
};

}  // namespace agp

using namespace agp;

// all-reduce of a device buffer that kernels on the context's MAIN stream produced / will consume: used by entry points
// outside the sharded dense fit (the sparse GP's group-sharded fit).  The minimal ShardOps a transport needs.
namespace agp {
struct MainStreamOps : ShardOps {
  agp_context *ctx;
  explicit MainStreamOps(agp_context *c) : ctx(c) {}
  void *stream(int) override { return ctx->stream; }
  bool device_memory() const override { return true; }
  int to_host(int, const double *dev, double *host, long long count) override {
    if (hipMemcpyAsync(host, dev, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess) return AGP_ERR_HIP;
    return wait_stream(ctx, ctx->stream, comm_timeout_seconds());
  }
  int from_host(int, const double *host, double *dev, long long count) override {
    if (hipMemcpyAsync(dev, host, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, ctx->stream) != hipSuccess) return AGP_ERR_HIP;
    return wait_stream(ctx, ctx->stream, comm_timeout_seconds());
  }
  void factor_diag(int, double *, long long, long long, long long, double *, double *) override {}
  void trsm_rows(int, double *, long long, long long, long long, const double *, const double *, const double *, double *) override {}
  void gemm(int, double *, long long, const double *, long long, const double *, long long, long long, long long, long long, bool, int) override {}
  void copy2d(int, double *, long long, const double *, long long, long long, long long) override {}
  void invert_diag(int, const double *, long long, long long, const double *, double *) override {}
  void colvec_dot(int, const double *, long long, long long, long long, const double *, double, double, const double *, double *) override {}
  void axpby(int, long long, double, const double *, double, const double *, double *) override {}
  void fill_zero(int, double *, long long) override {}
  void status(double out[2]) override { out[0] = out[1] = 0.; }
};

int comm_all_reduce_device(agp_context *ctx, agp_comm *comm, double *dev, long long count, int op) {
  if (!comm || !comm->impl || comm->impl->world == 1) return AGP_OK;
  MainStreamOps ops(ctx);
  const int st = comm->impl->all_reduce(ops, QP, dev, count, op);
  if (st != AGP_OK) return st;
  // the caller goes on with calls that synchronise the stream without a deadline (factorisations, stage timers): find a
  // collective that a dead peer never joins HERE, where it becomes AGP_ERR_COMM
  const int sw = wait_stream(ctx, ctx->stream, comm_timeout_seconds());
  return sw != AGP_OK ? sw : comm->impl->check_health();
}

int comm_wait_stream(agp_context *ctx, hipStream_t s) { return wait_stream(ctx, s, comm_timeout_seconds()); }
}  // namespace agp

struct agp_sharded_fit {
  agp_context_impl *ctx = nullptr;
  agp_comm *comm = nullptr;
  ShardPlan plan;
  double *A = nullptr;      // local stacked rows, column-major, ld
  long long ld = 0;
  size_t A_bytes = 0;
  double *work = nullptr;   // scratch of the schedule (ShardBuffers) followed by y
  size_t work_bytes = 0;
  ShardBuffers buf;
  double *y = nullptr;      // the local entries of z = L^-1 y (inside `work`)
  DeviceFeatures train;     // all training features (every rank holds the whole dataset)
  double log_det = 0.;
  int64_t failed_pivot = -1;
  double stage[8] = {0, 0, 0, 0, 0, 0, 0, 0};
};

extern "C" {

int agp_comm_unique_id(void *id) {
  if (!id) return AGP_ERR_INVALID_ARGUMENT;
  RcclApi *api = rccl_api();
  if (!api) return AGP_ERR_COMM;
  static_assert(sizeof(ncclUniqueId) <= AGP_COMM_ID_BYTES, "unique id size");
  ncclUniqueId u;
  if (api->GetUniqueId(&u) != ncclSuccess) return AGP_ERR_COMM;
  std::memset(id, 0, AGP_COMM_ID_BYTES);
  std::memcpy(id, &u, sizeof(u));
  return AGP_OK;
}

int agp_comm_create(agp_context *ctx, int nranks, int rank, const void *id, agp_comm **out) {
  if (!ctx || !id || !out || nranks < 1 || rank < 0 || rank >= nranks) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  RcclApi *api = rccl_api();
  if (!api) { ctx->last_error = "librccl could not be loaded"; return AGP_ERR_COMM; }
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  RcclComm *c = new (std::nothrow) RcclComm();
  if (!c) return AGP_ERR_INVALID_ARGUMENT;
  c->api = api;
  c->ctx = ctx;
  c->world = nranks;
  c->rank = rank;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  const ncclResult_t r = api->CommInitRank(&c->comm, nranks, u, rank);
  if (r != ncclSuccess) {
    ctx->last_error = std::string("ncclCommInitRank: ") + api->GetErrorString(r);
    c->comm = nullptr;
    delete c;
    return AGP_ERR_COMM;
  }
  if (hipMalloc(&c->scratch, sizeof(double) * RcclComm::SCRATCH) != hipSuccess) { delete c; return AGP_ERR_HIP; }
  agp_comm *h = new (std::nothrow) agp_comm();
  if (!h) { delete c; return AGP_ERR_INVALID_ARGUMENT; }
  h->impl = c;
  *out = h;
  return AGP_OK;
}

void agp_sharded_fit_destroy(agp_sharded_fit *f) {
  if (!f) return;
  if (f->ctx) (void)hipSetDevice(f->ctx->device);
  // park the two large buffers in the context for the next fit of the same size (kernels that used them were enqueued
  // on the context's streams, and so is whatever uses them next)
  if (f->A) {
    if (f->ctx && !f->ctx->pool_A) { f->ctx->pool_A = f->A; f->ctx->pool_A_bytes = f->A_bytes; }
    else (void)dev_release(f->A);  // (may have come from a dense factor's dev_malloc through pool_A)
  }
  if (f->work) {
    if (f->ctx && !f->ctx->pool_shard) { f->ctx->pool_shard = f->work; f->ctx->pool_shard_bytes = f->work_bytes; }
    else (void)dev_release(f->work);
  }
  f->train.release();
  delete f;
}

int64_t agp_sharded_fit_failed_pivot(const agp_sharded_fit *f) { return f ? f->failed_pivot : -1; }

int agp_sharded_fit_stage(const agp_sharded_fit *f, int stage, double *value) {
  if (!f || !value || stage < 0 || stage > 7) return AGP_ERR_INVALID_ARGUMENT;
  *value = f->stage[stage];
  return AGP_OK;
}

int agp_sharded_fit_create(agp_context *c, agp_comm *comm, const agp_kernel *k, const agp_features *x, const double *y,
                           const double *y_var, agp_sharded_fit **out, double *information, double *log_det) {
  if (!c || !k || !x || !y || !out) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(x);
  if (st != AGP_OK) return st;
  const long long n = x->n;
  if (n <= 0) return AGP_ERR_INVALID_ARGUMENT;
  HostReducingComm *tr = comm ? comm->impl : nullptr;
  const int world = tr ? tr->world : 1, rank = tr ? tr->rank : 0;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;

  agp_sharded_fit *f = new (std::nothrow) agp_sharded_fit();
  if (!f) return AGP_ERR_INVALID_ARGUMENT;
  f->ctx = ctx;
  f->comm = comm;
  long long block = NBO;
  {  // AGP_SHARD_BLOCK (tests: many row blocks at small n)
    const long long b = ctx->tune.shard_block;
    if (b == 128 || b == 256 || b == 512) block = b;
  }
  f->plan = ShardPlan(n, block, world, rank);
  f->plan.force_comm = tr && ctx->tune.shard_force_comm;
  const ShardPlan &plan = f->plan;
  const long long nlb = plan.n_local_blocks(rank), B = plan.B;
  f->ld = factor_ld(plan.max_local_blocks() * B);  // the same on every rank (agp_sharded_fit_replicate gathers the stacks)
  hipStream_t s = ctx->stream;
  double *yvar_d = nullptr;
#define SFIT_CHECK(expr)                                                     \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);   \
      if (yvar_d) (void)hipFree(yvar_d);                                     \
      agp_sharded_fit_destroy(f);                                            \
      return AGP_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)
  if ((st = to_device(ctx, x, true, &f->train)) != AGP_OK) { agp_sharded_fit_destroy(f); return st; }
  f->train.v.meas = 0;
  f->A_bytes = sizeof(double) * (size_t)f->ld * (size_t)n;
  if (ctx->pool_A && ctx->pool_A_bytes == f->A_bytes) {
    f->A = ctx->pool_A;
    ctx->pool_A = nullptr;
    ctx->pool_A_bytes = 0;
  } else {
    SFIT_CHECK(hipMalloc(&f->A, f->A_bytes));
  }
  const long long work_doubles = shard_work_doubles(plan);
  f->work_bytes = sizeof(double) * (size_t)(work_doubles + plan.max_local_blocks() * B + 8);
  if (ctx->pool_shard && ctx->pool_shard_bytes == f->work_bytes) {
    f->work = ctx->pool_shard;
    ctx->pool_shard = nullptr;
    ctx->pool_shard_bytes = 0;
  } else {
    SFIT_CHECK(hipMalloc(&f->work, f->work_bytes));
  }
  f->y = f->work + work_doubles;
  shard_carve(plan, f->work, &f->buf);
  const hipMemcpyKind kind = x->location == AGP_HOST ? hipMemcpyHostToDevice : hipMemcpyDeviceToDevice;
  // the local targets: block by block in local order
  for (long long li = 0; li < nlb; ++li) {
    const long long i = plan.global_block(rank, li);
    SFIT_CHECK(hipMemcpyAsync(f->y + li * B, y + i * B, sizeof(double) * (size_t)plan.width(i), kind, s));
  }
  if (y_var) {
    SFIT_CHECK(hipMalloc(&yvar_d, sizeof(double) * (size_t)n));
    SFIT_CHECK(hipMemcpyAsync(yvar_d, y_var, sizeof(double) * (size_t)n, kind, s));
  }
  SFIT_CHECK(hipMemsetAsync(ctx->d_flags, 0, 4 * sizeof(int), s));
  SFIT_CHECK(hipMemsetAsync(ctx->d_scalars, 0, 4 * sizeof(double), s));
  if (x->location == AGP_HOST) SFIT_CHECK(hipStreamSynchronize(s));

  // ---- Gram of the own row blocks: no communication (as_measurements(features), gp.hpp:288-290) ----
  const auto t0 = std::chrono::steady_clock::now();
  FeatView all = f->train.v;
  all.meas = 1;
  all.sstride = all.n;
  if (world == 1 && !plan.force_comm) {
    launch_gram(s, dprog, all, all, /*symmetric=*/true, /*lower_only=*/true, f->A, f->ld, yvar_d, ctx->d_flags, &k->prog);
  } else {
    for (long long li = 0; li < nlb; ++li) {
      const long long i = plan.global_block(rank, li), w = plan.width(i);
      FeatView rows = all, cols = all;
      rows.coords = all.coords + i * B * all.dim;
      rows.ids = all.ids ? all.ids + i * B : nullptr;
      rows.scales = all.scales ? all.scales + i * B : nullptr;
      rows.n = w;
      cols.n = i * B + w;
      launch_gram(s, dprog, rows, cols, /*symmetric=*/false, /*lower_only=*/false, f->A + li * B, f->ld, nullptr, ctx->d_flags,
                  &k->prog);
      if (yvar_d)
        hipLaunchKernelGGL(shard_add_diag_kernel, dim3((unsigned)((w + 255) / 256)), dim3(256), 0, s, f->A, f->ld, li * B, i * B, w,
                           yvar_d);
    }
  }
  int nan_flag = 0;
  SFIT_CHECK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
  SFIT_CHECK(hipStreamSynchronize(s));
  nan_flag = ctx->h_flags[0];
  const auto t1 = std::chrono::steady_clock::now();
  if (yvar_d) { (void)hipFree(yvar_d); yvar_d = nullptr; }
  if (tr) {  // ALBATROSS_ASSERT(!cov.hasNaN()), gp.hpp:66 - every rank must take the same exit
    double v = (double)nan_flag;
    if ((st = tr->all_reduce_host(&v, 1, 1)) != AGP_OK) { agp_sharded_fit_destroy(f); return st; }
    nan_flag = v > 0.;
  }
  if (nan_flag) { agp_sharded_fit_destroy(f); return AGP_ERR_NAN_INPUT; }

  // ---- factorisation + both substitutions ----
  TraceRange tr_factor("agp: sharded factor + substitutions (gp.hpp:61-69 over the ranks)");
  ShardResult res;
  if (!plan.multi()) {
    // ONE rank: the local matrix is the whole matrix - the single-GPU factorisation itself (chol.hip: factor_lower with
    // its two-stream look-ahead, api.hip: blocked backward substitution), exactly what agp_fit_create runs
    FactorTimers timers;
    if (ctx->profiling) {
      const size_t want = (size_t)(2 * (2 * ((n + NB - 1) / NB) + 4));
      while (ctx->gemm_events.size() < want) {
        hipEvent_t e;
        SFIT_CHECK(hipEventCreate(&e));
        ctx->gemm_events.push_back(e);
      }
      ctx->gemm_flops.assign(want / 2, 0.);
      timers.ev = ctx->gemm_events.data();
      timers.flops = ctx->gemm_flops.data();
      timers.n_ev = (int)want;
    }
    factor_lower(ctx, f->A, n, f->ld, f->buf.img_local, f->y, ctx->profiling ? &timers : nullptr);
    SFIT_CHECK(hipMemcpyAsync(ctx->h_flags, ctx->d_flags, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
    SFIT_CHECK(hipMemcpyAsync(ctx->h_scalars, ctx->d_scalars, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
    SFIT_CHECK(hipStreamSynchronize(s));
    res.log_det = 2. * ctx->h_scalars[0];
    res.bad_pivot = ctx->h_flags[1] ? (long long)ctx->h_flags[1] - 1 : -1;
    st = res.bad_pivot >= 0 ? AGP_ERR_NOT_POSITIVE_DEFINITE : AGP_OK;
    if (ctx->h_flags[2]) { agp_sharded_fit_destroy(f); return status_from_flags(ctx); }
    if (st == AGP_OK) {
      SFIT_CHECK(hipMemcpyAsync(f->buf.xfull, f->y, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s));
      const int st2 = ensure_ws(ctx, &ctx->ws_aux, &ctx->ws_aux_bytes, sizeof(double) * backsolve_ws_elems(n));
      if (st2 != AGP_OK) { agp_sharded_fit_destroy(f); return st2; }
      backward_solve_vec_any(s, f->A, n, f->ld, f->buf.img_local, f->buf.xfull, ctx->ws_aux);
      SFIT_CHECK(hipStreamSynchronize(s));
      SFIT_CHECK(hipGetLastError());
    }
    if (ctx->profiling) {
      double ms_sum = 0., flop = 0.;
      for (int i = 0; i + 1 < timers.used; i += 2) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, timers.ev[i], timers.ev[i + 1]);
        ms_sum += ms;
        flop += timers.flops[i / 2];
      }
      f->stage[3] = ms_sum;
      f->stage[4] = timers.used / 2;
      f->stage[5] = flop;
    }
  } else {
    HipShardOps ops(ctx);
    if (!ops.ok) { agp_sharded_fit_destroy(f); ctx->last_error = "stream / event creation failed"; return AGP_ERR_HIP; }
    st = shard_factor_solve(ops, tr, plan, f->A, f->ld, f->y, f->buf, &res);
    if (ops.handover_timeout && (st == AGP_OK || st == AGP_ERR_NOT_POSITIVE_DEFINITE)) {
      ctx->last_error = "panel kernel: hand-over of a diagonal block timed out";
      st = AGP_ERR_HIP;
    }
    if (ops.gate_timeout && (st == AGP_OK || st == AGP_ERR_NOT_POSITIVE_DEFINITE)) {
      ctx->last_error = "sharded schedule: a queue waited for another (or for a collective) past the transport's deadline";
      st = AGP_ERR_COMM;
    }
    if (ops.gate_timeout) {  // whatever the reason: the next fit of this context is paced by the host, which waits for nothing on the device
      ctx->shard_host_pacing = 1;
      ctx->shard_probe_ok = false;
    }
    f->stage[2] = ops.device_pacing ? 1. : 0.;  // the pacing the fit ENDED with
    f->stage[6] = res.enqueue_factor_ms + res.enqueue_solve_ms;
    f->stage[7] = res.total_ms;
    if (ctx->profiling) {
      double ms_sum = 0., flop = 0.;
      for (size_t i = 0; i + 1 < ops.tused; i += 2) {
        float ms = 0.f;
        (void)hipEventElapsedTime(&ms, ops.tev[i], ops.tev[i + 1]);
        ms_sum += ms;
        flop += ops.tflop[i / 2];
      }
      f->stage[3] = ms_sum;
      f->stage[4] = (double)(ops.tused / 2);
      f->stage[5] = flop;
    }
  }
  const auto t2 = std::chrono::steady_clock::now();
  f->stage[0] = std::chrono::duration<double, std::milli>(t1 - t0).count();
  f->stage[1] = std::chrono::duration<double, std::milli>(t2 - t1).count();
  f->log_det = res.log_det;
  f->failed_pivot = res.bad_pivot;
  if (st == AGP_ERR_NOT_POSITIVE_DEFINITE) { *out = f; return st; }  // the handle reports the pivot
  if (st != AGP_OK) { agp_sharded_fit_destroy(f); return st; }
  if (information) SFIT_CHECK(hipMemcpy(information, f->buf.xfull, sizeof(double) * (size_t)n, hipMemcpyDeviceToHost));
  if (log_det) *log_det = f->log_det;
  *out = f;
#undef SFIT_CHECK
  return AGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Marginal predictions straight from the SHARDED factor - no replication (SURVEY.md section 8e "Solve: distributed
// forward substitution"; gp.hpp:87-101).  For factors too large to replicate on every GPU; agp_sharded_fit_replicate +
// agp_predict_* with the test points split over the ranks remains the faster way while the factor fits (it needs no
// exchange at all).  Every rank passes the same test points and receives all means and variances.
//   mean      = K*^T information                      every rank has the whole information vector: no exchange
//   variance  = k** - colsum(V o V),  V = L^-1 K*     V by block rows: the owner of block row i solves
//               V_i = L_ii^-1 (K*_i - sum_{j<i} L_ij V_j) on its rows and broadcasts V_i (w x M), every rank subtracts
//               L_li V_i from its own later rows; the column sums of squares are all-reduced once at the end.
// Exchange per call: N x M doubles in nb broadcasts + one all-reduce of M doubles.
// ---------------------------------------------------------------------------------------------------------------
static int sharded_predict(agp_context *c, const agp_kernel *k, agp_sharded_fit *f, const agp_features *xs, double *mean,
                           double *variance, bool joint, int out_location) {
  if (!c || !k || !f || !xs || !mean || !variance || f->failed_pivot >= 0 || !f->A) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  if (ctx != f->ctx) return AGP_ERR_INVALID_ARGUMENT;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  int st = validate_features(xs);
  if (st != AGP_OK) return st;
  if (xs->dim != f->train.v.dim) return AGP_ERR_INVALID_ARGUMENT;
  const long long m_all = xs->n;
  if (m_all == 0) return AGP_OK;
  const ShardPlan &plan = f->plan;
  const long long n = plan.n, B = plan.B, nb = plan.nb;
  const int me = plan.rank;
  HostReducingComm *tr = f->comm ? f->comm->impl : nullptr;
  if (plan.multi() && !tr) return AGP_ERR_INVALID_ARGUMENT;
  const DevProgram *dprog = nullptr;
  if ((st = device_program(ctx, k, &dprog)) != AGP_OK) return st;
  DeviceFeatures dxs;
  if ((st = to_device(ctx, xs, false, &dxs)) != AGP_OK) return st;
  const long long n_loc = plan.multi() ? plan.local_rows(me) : n, nlb = plan.n_local_blocks(me);
  const long long ldk = round_up(std::max<long long>(n_loc, 2), 2);
  // (a joint prediction needs all test points at once: its m x m covariance couples them)
  const long long chunk = joint ? m_all : std::min<long long>(m_all, 4096), ldc = round_up(chunk, 2);
  // workspace: K*_loc (n_loc x chunk) | V_i (B x chunk, ld = B) | mean | prior | acc  (joint: prior and acc are m x m)
  double *ws = nullptr;
  const size_t sq = joint ? (size_t)ldc * (size_t)chunk : (size_t)ldc;
  const size_t ws_elems = (size_t)ldk * (size_t)chunk + (size_t)B * (size_t)chunk + (size_t)ldc + 2 * sq;
  if (hipMalloc(&ws, sizeof(double) * ws_elems) != hipSuccess) { dxs.release(); ctx->last_error = "hipMalloc (sharded prediction workspace)"; return AGP_ERR_HIP; }
  double *Kloc = ws, *Vi = Kloc + (size_t)ldk * (size_t)chunk, *mean_d = Vi + (size_t)B * (size_t)chunk, *prior = mean_d + ldc,
         *acc = prior + sq;
  HipShardOps ops(ctx);
  hipStream_t s = ctx->stream;
  FeatView train = f->train.v;
  for (long long o = 0; o < m_all && st == AGP_OK; o += chunk) {
    const long long m = std::min(chunk, m_all - o);
    FeatView xv = dxs.v;
    xv.sstride = scale_stride(dxs.v);
    xv.n = m;
    xv.coords = dxs.v.coords + o * dxs.v.dim;
    xv.ids = dxs.v.ids ? dxs.v.ids + o : nullptr;
    xv.scales = dxs.v.scales ? dxs.v.scales + o : nullptr;
    launch_predict_mean(s, dprog, train, xv, f->buf.xfull, mean_d, &k->prog);  // gp.hpp:82-85
    if (joint) launch_gram(s, dprog, xv, xv, true, false, prior, ldc, nullptr, nullptr, &k->prog);  // prior_cov, gp.hpp:317
    else launch_gram_diagonal(s, dprog, xv, prior);                            // gp.hpp:339-343
    if (!plan.multi()) {
      // one rank: the local matrix is the whole factor (agp_sharded_fit_create's single-GPU path)
      FeatView all = train;
      launch_gram(s, dprog, all, xv, false, false, Kloc, ldk, nullptr, nullptr, &k->prog);
      forward_solve_mat_lookahead(ctx, f->A, n, f->ld, f->buf.img_local, Kloc, m, ldk);
      if (joint) {  // gp.hpp:111: K** - V^T V (lower tiles, then mirrored)
        launch_gemm_nt_sub(s, prior, ldc, Kloc, ldk, true, Kloc, ldk, true, m, m, n, true);
        launch_symmetrize(s, prior, ldc, m);
      } else {
        launch_coldot(s, Kloc, ldk, Kloc, ldk, n, m, prior, 1.0, prior);  // gp.hpp:97-99
      }
    } else {
      // cross covariance of the own row blocks (cov(train_features, features), gp.hpp:337): no exchange
      for (long long li = 0; li < nlb; ++li) {
        const long long i = plan.global_block(me, li);
        FeatView rows = train;
        rows.sstride = scale_stride(train);
        rows.coords = train.coords + i * B * train.dim;
        rows.ids = train.ids ? train.ids + i * B : nullptr;
        rows.scales = train.scales ? train.scales + i * B : nullptr;
        rows.n = plan.width(i);
        launch_gram(s, dprog, rows, xv, false, false, Kloc + li * B, ldk, nullptr, nullptr, &k->prog);
      }
      (void)hipMemsetAsync(acc, 0, sizeof(double) * (joint ? (size_t)ldc * (size_t)m : (size_t)m), s);
      for (long long i = 0; i < nb && st == AGP_OK; ++i) {
        const int own = plan.owner(i);
        const long long w = plan.width(i);
        if (own == me) {
          const long long li = plan.local_index(i);
          double *Vrows = Kloc + li * B;  // w x m, ld = ldk
          forward_solve_mat(s, f->A + li * B + i * B * f->ld, w, f->ld, f->buf.img_local + li * 4 * SHARD_IMG, Vrows, m, ldk);
          if (!joint) launch_coldot(s, Vrows, ldk, Vrows, ldk, w, m, acc, 1.0, acc);  // acc -= colsum(V_i o V_i)
          ops.copy2d(QP, Vi, B, Vrows, ldk, w, m);
        }
        if (i == nb - 1) break;  // nobody has rows below the last block
        if ((st = tr->broadcast(ops, QP, Vi, B * m, own)) != AGP_OK) break;
        const long long li2 = plan.first_local_after(me, i), rows2 = n_loc - li2 * B;
        if (rows2 > 0)  // K*[own rows of blocks > i] -= L[those rows, block column i] V_i
          launch_gemm_nt_sub(s, Kloc + li2 * B, ldk, f->A + li2 * B + i * B * f->ld, f->ld, false, Vi, B, true, rows2, m, w, false);
      }
      if (joint) {
        // every rank's rows of V = L^-1 K* now stand in K*_loc: acc = - V_own^T V_own (lower tiles, one MFMA product over
        // the stacked own rows), summed over the ranks by ONE all-reduce of m x m doubles
        if (st == AGP_OK && n_loc > 0) launch_gemm_nt_sub(s, acc, ldc, Kloc, ldk, true, Kloc, ldk, true, m, m, n_loc, true);
        if (st == AGP_OK) st = tr->all_reduce(ops, QP, acc, ldc * m, 0);
        if (st == AGP_OK) {
          launch_axpby(s, ldc * m, 1.0, prior, 1.0, acc, prior);  // K** - sum over all ranks (gp.hpp:111)
          launch_symmetrize(s, prior, ldc, m);
        }
      } else {
        if (st == AGP_OK) st = tr->all_reduce(ops, QP, acc, m, 0);
        if (st == AGP_OK) launch_axpby(s, m, 1.0, prior, 1.0, acc, prior);  // k** - sum over all ranks
      }
    }
    if (st == AGP_OK) st = wait_stream(ctx, s, comm_timeout_seconds());
    if (st == AGP_OK && tr) st = tr->check_health();
    if (st == AGP_OK) st = copy_out(ctx, mean_d, m, mean + o, out_location);
    if (st == AGP_OK) st = joint ? copy_out_2d(ctx, prior, ldc, m, m, variance, m, out_location) : copy_out(ctx, prior, m, variance + o, out_location);
  }
  (void)hipStreamSynchronize(s);
  (void)hipFree(ws);
  dxs.release();
  return st;
}

int agp_sharded_predict_marginal(agp_context *c, const agp_kernel *k, agp_sharded_fit *f, const agp_features *xs, double *mean,
                                 double *variance, int out_location) {
  return sharded_predict(c, k, f, xs, mean, variance, false, out_location);
}

// gp_joint_prediction (gp.hpp:103-113) from the SHARDED factor: the distributed forward substitution of the marginal
// prediction, then K** - V^T V with every rank contributing the product of its own rows of V and ONE all-reduce of the
// m x m result.  For factors too large to replicate; all m test points at once (the rank-local block of V is n_loc x m).
int agp_sharded_predict_joint(agp_context *c, const agp_kernel *k, agp_sharded_fit *f, const agp_features *xs, double *mean,
                              double *covariance, int out_location) {
  return sharded_predict(c, k, f, xs, mean, covariance, true, out_location);
}

int agp_sharded_fit_replicate(agp_context *c, agp_sharded_fit *f, agp_fit **out) {
  if (!c || !f || !out || f->failed_pivot >= 0 || !f->A) return AGP_ERR_INVALID_ARGUMENT;
  agp_context_impl *ctx = static_cast<agp_context_impl *>(c);
  if (ctx != f->ctx) return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  const ShardPlan &plan = f->plan;
  const long long n = plan.n, B = plan.B, nlbm = plan.max_local_blocks();
  const int world = plan.world;
  HostReducingComm *tr = f->comm ? f->comm->impl : nullptr;
  if (plan.multi() && !tr) return AGP_ERR_INVALID_ARGUMENT;
  agp_fit *fit = new (std::nothrow) agp_fit();
  if (!fit) return AGP_ERR_INVALID_ARGUMENT;
  fit->ctx = ctx;
  fit->device = ctx->device;
  fit->n = n;
  fit->lda = factor_ld(n);
  fit->A_bytes = sizeof(double) * (size_t)fit->lda * (size_t)n;
  fit->log_det = f->log_det;
  const long long nblk = (n + NB - 1) / NB;
  double *stacks = nullptr, *imgs = nullptr;
  const long long per_rank = f->ld * n, img_per_rank = nlbm * 4 * SHARD_IMG;
#define REP_CHECK(expr)                                                      \
  do {                                                                       \
    hipError_t _e = (expr);                                                  \
    if (_e != hipSuccess) {                                                  \
      ctx->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);   \
      if (stacks && stacks != f->A) (void)hipFree(stacks);                   \
      if (imgs && imgs != f->buf.img_local) (void)hipFree(imgs);             \
      agp_fit_destroy(fit);                                                  \
      return AGP_ERR_HIP;                                                    \
    }                                                                        \
  } while (0)
  REP_CHECK(hipMalloc(&fit->A, fit->A_bytes));
  REP_CHECK(hipMalloc(&fit->invd, sizeof(double) * (size_t)nblk * SHARD_IMG));
  REP_CHECK(hipMalloc(&fit->alpha, sizeof(double) * (size_t)n));
  int st = AGP_OK;
  {
    HipShardOps ops(ctx);
    hipStream_t s = (hipStream_t)ops.stream(QC);
    if (plan.multi()) {
      REP_CHECK(hipMalloc(&stacks, sizeof(double) * (size_t)per_rank * (size_t)world));
      REP_CHECK(hipMalloc(&imgs, sizeof(double) * (size_t)img_per_rank * (size_t)world));
      st = tr->all_gather(ops, QC, f->A, stacks, per_rank);
      if (st == AGP_OK) st = tr->all_gather(ops, QC, f->buf.img_local, imgs, img_per_rank);
    } else {
      stacks = f->A;
      imgs = f->buf.img_local;
    }
    if (st == AGP_OK) {
      hipLaunchKernelGGL(shard_unstack_kernel, dim3((unsigned)((n + 255) / 256), 64), dim3(256), 0, s, fit->A, (long long)fit->lda, stacks,
                         f->ld, per_rank, n, B, world);
      if (!plan.multi()) {
        // one rank, single-GPU factorisation (factor_lower): one image per 128-block, contiguous whatever the row-block size
        (void)hipMemcpyAsync(fit->invd, imgs, sizeof(double) * (size_t)(nblk * SHARD_IMG), hipMemcpyDeviceToDevice, s);
      }
      for (long long i = 0; plan.multi() && i < plan.nb; ++i) {
        const long long sub = (plan.width(i) + NB - 1) / NB;  // 128-blocks of this row block
        (void)hipMemcpyAsync(fit->invd + i * (B / NB) * SHARD_IMG,
                             imgs + (long long)plan.owner(i) * img_per_rank + plan.local_index(i) * 4 * SHARD_IMG,
                             sizeof(double) * (size_t)(sub * SHARD_IMG), hipMemcpyDeviceToDevice, s);
      }
      (void)hipMemcpyAsync(fit->alpha, f->buf.xfull, sizeof(double) * (size_t)n, hipMemcpyDeviceToDevice, s);
      st = ops.sync_all();
      if (st == AGP_OK && tr) st = tr->check_health();
    }
  }
  if (stacks && stacks != f->A) (void)hipFree(stacks);
  if (imgs && imgs != f->buf.img_local) (void)hipFree(imgs);
  stacks = imgs = nullptr;
  if (st != AGP_OK) { agp_fit_destroy(fit); return st; }
  // train_features = features (gp.hpp:63): an owned copy, like agp_fit_create
  {
    agp_features view;
    view.n = f->train.v.n; view.dim = f->train.v.dim; view.n_scale_columns = f->train.v.nsc;
    view.coords = f->train.v.coords;
    view.eq_id = reinterpret_cast<const int64_t *>(f->train.v.ids);
    view.scales = f->train.v.scales;
    view.is_measurement = 0;
    view.location = AGP_DEVICE;
    if ((st = to_device(ctx, &view, true, &fit->train)) != AGP_OK) { agp_fit_destroy(fit); return st; }
    REP_CHECK(hipStreamSynchronize(ctx->stream));
  }
#undef REP_CHECK
  *out = fit;
  return AGP_OK;
}

}  // extern "C"
