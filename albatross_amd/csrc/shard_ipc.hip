// shard_ipc.hip — a DEVICE-ASYNCHRONOUS transport for the sharded entry points between processes that share ONE GPU
// (agp_comm_create_ipc): every collective is a handful of kernels on the caller's stream - peer stores into mailboxes
// opened with hipIpcOpenMemHandle, stream-ordered flags, bounded spins - and the host returns at once, exactly like
// RCCL.  What it is for: the pool's boxes have one GPU, where RCCL refuses a second rank per device and the callback
// transport (agp_comm_create_callbacks) is host-synchronous - it cannot show an ordering bug of ASYNCHRONOUS
// collectives (a buffer reused before a peer has read it, a queue that overtakes another).  With this transport the
// 2 ... 8-rank schedule runs with the same asynchrony it has over RCCL before it ever meets an 8-GPU node.  RCCL
// stays the transport of real multi-GPU runs (shard_hip.hip).
//
// Protocol.  Rank r owns a region [flags (W) | acks (W) | error word | two mailbox slots of `cap` doubles], exported
// with hipIpcGetMemHandle; the handles travel over the caller's host collectives (`bootstrap`, which also serve the
// control plane: agp_comm_all_reduce_host / agp_comm_barrier).  Collective number s (the same on every rank: all ranks
// issue the same collectives in the same order), slot s & 1:
//   push    every workgroup waits until every peer has acknowledged collective s - 2 (the slot is free), then the grid
//           copies this rank's contribution into the slot of every peer (broadcast: the root only)
//   signal  flags[me] of every peer <- s (release; stream order has completed the push), then wait for flags[p] >= s
//           of every peer p (acquire)
//   pull    mailbox -> destination (all-gather: a copy; all-reduce: the sum / max over ranks IN RANK ORDER, so every
//           rank computes bit-identical results; broadcast: a copy on the non-roots)
//   ack     acks[me] of every peer <- s
// Every spin is bounded (s_memrealtime, the transport's deadline): a dead peer raises the error word - the schedule's
// status check and check_health() turn it into AGP_ERR_COMM - and the stream runs on.
// Messages larger than a slot are sent in pieces.  Reference work replaced: none (the reference is single-process); this
// is test infrastructure of the multi-GPU path of models/gp.hpp:61-69.
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "api_internal.h"
#include "shard_internal.h"

namespace agp {
int comm_wait_stream(agp_context *ctx, hipStream_t s);  // shard_hip.hip
double comm_timeout_seconds();                           // shard_hip.hip

namespace {

constexpr int IPC_MAX_WORLD = 16;

struct IpcPeers {
  unsigned long long *flags[IPC_MAX_WORLD];  // flags[p]: rank p's flag array (W entries), written at index `me`
  unsigned long long *acks[IPC_MAX_WORLD];
  double *mbox[IPC_MAX_WORLD];               // rank p's two slots
  int world, me;
};

__device__ __forceinline__ bool ipc_spin(const unsigned long long *f, unsigned long long need, unsigned long long ticks, int *err) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < need) {
    __builtin_amdgcn_s_sleep(8);
    if (__builtin_amdgcn_s_memrealtime() - t0 > ticks) {
      atomicExch(err, 1);
      return false;
    }
  }
  return true;
}

// push: src (count doubles) -> slot of every peer at offset `off` (root >= 0: only the root pushes)
__global__ __launch_bounds__(256) void ipc_push_kernel(IpcPeers P, const double *__restrict__ src, long long count, long long off,
                                                       long long slot_off, unsigned long long seq, int root, unsigned long long ticks,
                                                       int *err) {
  if (root >= 0 && root != P.me) return;
  // the slot of every peer must be free: they acknowledged the collective that used it last (seq - 2)
  __shared__ int ok;
  if (threadIdx.x == 0) ok = 1;
  __syncthreads();
  if (seq > 2 && (int)threadIdx.x < P.world && (int)threadIdx.x != P.me)
    if (!ipc_spin(P.acks[P.me] + threadIdx.x, seq - 2, ticks, err)) ok = 0;
  __syncthreads();
  if (!ok) {
    // a peer never acknowledged: this rank pushes nothing - and says so in EVERY peer's error word (it sits behind the flag
    // and acknowledgement arrays of a region), so that the peers, which will pull stale mailbox contents, report
    // AGP_ERR_COMM too instead of returning wrong numbers with AGP_OK
    if (blockIdx.x == 0 && (int)threadIdx.x < P.world)
      (void)__hip_atomic_exchange(reinterpret_cast<int *>(P.flags[threadIdx.x] + 2 * IPC_MAX_WORLD), 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    return;
  }
  const long long stride = (long long)gridDim.x * 256;
  for (int p = 0; p < P.world; ++p) {
    if (root >= 0 && p == P.me) continue;  // the root keeps its own copy
    double *dst = P.mbox[p] + slot_off + off;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) dst[i] = src[i];
  }
}

__global__ void ipc_signal_kernel(IpcPeers P, unsigned long long seq, unsigned long long ticks, int *err) {
  const int p = (int)threadIdx.x;
  if (p >= P.world) return;
  // (a maximum, not a store: sequence numbers only grow, whatever the order in which two streams' collectives get here)
  (void)__hip_atomic_fetch_max(P.flags[p] + P.me, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
  (void)ipc_spin(P.flags[P.me] + p, seq, ticks, err);
}

// pull: kind 0 copy `count` doubles from offset `off` of the own slot; kind 1 / 2: dst[i] = sum / max over ranks (rank
// order) of slot[r * count + i]
__global__ __launch_bounds__(256) void ipc_pull_kernel(IpcPeers P, double *__restrict__ dst, long long count, long long off,
                                                       long long slot_off, int kind, int skip_rank) {
  if (skip_rank == P.me) return;
  const double *box = P.mbox[P.me] + slot_off;
  const long long stride = (long long)gridDim.x * 256;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < count; i += stride) {
    if (kind == 0) {
      dst[i] = box[off + i];
    } else {
      double v = box[i];
      for (int r = 1; r < P.world; ++r) {
        const double x = box[(long long)r * count + i];
        v = kind == 1 ? v + x : (x > v ? x : v);
      }
      dst[i] = v;
    }
  }
}

__global__ void ipc_ack_kernel(IpcPeers P, unsigned long long seq) {
  const int p = (int)threadIdx.x;
  if (p < P.world) (void)__hip_atomic_fetch_max(P.acks[p] + P.me, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

struct IpcComm : HostReducingComm {
  agp_context *ctx = nullptr;
  agp_comm_callbacks boot{};
  IpcPeers peers{};
  void *region = nullptr;                 // own allocation
  void *opened[IPC_MAX_WORLD] = {};       // peers' regions as mapped here
  long long cap = 0;                      // doubles per mailbox slot
  unsigned long long seq = 0;
  int *err = nullptr;                     // error word inside the own region
  unsigned long long ticks = 0;
  bool broken = false;

  static size_t header_bytes() { return 4096; }  // flags | acks | error word, page aligned
  ~IpcComm() override {
    if (ctx) (void)hipSetDevice(ctx->device);
    // nobody may still push into (or read flags of) a region that is about to go: host barrier first - unless a peer is
    // known to be gone
    (void)hipDeviceSynchronize();  // this rank's own queued pushes / acknowledgements first ...
    if (!broken && world > 1 && boot.all_reduce) {  // ... then everybody's: after the barrier no kernel of any rank touches a region
      double token = 0.;
      (void)boot.all_reduce(boot.user, &token, 1, 0);
    }
    for (int p = 0; p < world; ++p)
      if (opened[p]) (void)hipIpcCloseMemHandle(opened[p]);
    if (region) (void)hipFree(region);
  }
  void mark_broken() override { broken = true; }

  unsigned grid_for(long long count) const {
    const long long g = (count + 255) / 256;
    return (unsigned)(g < 1 ? 1 : (g > 512 ? 512 : g));
  }
  // one collective of at most `cap` doubles in the slot
  int piece(hipStream_t s, int kind, double *buf, const double *send, double *recv, long long count, int arg) {
    ++seq;
    const long long slot_off = (long long)(seq & 1) * cap;
    if (kind == 0) {  // broadcast, root = arg
      hipLaunchKernelGGL(ipc_push_kernel, dim3(grid_for(count)), dim3(256), 0, s, peers, buf, count, 0LL, slot_off, seq, arg, ticks, err);
      hipLaunchKernelGGL(ipc_signal_kernel, dim3(1), dim3(64), 0, s, peers, seq, ticks, err);
      hipLaunchKernelGGL(ipc_pull_kernel, dim3(grid_for(count)), dim3(256), 0, s, peers, buf, count, 0LL, slot_off, 0, arg);
    } else if (kind == 1) {  // all-gather
      hipLaunchKernelGGL(ipc_push_kernel, dim3(grid_for(count)), dim3(256), 0, s, peers, send, count, (long long)rank * count, slot_off,
                         seq, -1, ticks, err);
      hipLaunchKernelGGL(ipc_signal_kernel, dim3(1), dim3(64), 0, s, peers, seq, ticks, err);
      hipLaunchKernelGGL(ipc_pull_kernel, dim3(grid_for(count * world)), dim3(256), 0, s, peers, recv, count * world, 0LL, slot_off, 0, -1);
    } else {  // all-reduce, op = arg
      hipLaunchKernelGGL(ipc_push_kernel, dim3(grid_for(count)), dim3(256), 0, s, peers, buf, count, (long long)rank * count, slot_off, seq,
                         -1, ticks, err);
      hipLaunchKernelGGL(ipc_signal_kernel, dim3(1), dim3(64), 0, s, peers, seq, ticks, err);
      hipLaunchKernelGGL(ipc_pull_kernel, dim3(grid_for(count)), dim3(256), 0, s, peers, buf, count, 0LL, slot_off, arg == 1 ? 2 : 1, -1);
    }
    hipLaunchKernelGGL(ipc_ack_kernel, dim3(1), dim3(64), 0, s, peers, seq);
    if (hipGetLastError() != hipSuccess) { broken = true; return AGP_ERR_HIP; }
    return AGP_OK;
  }
  int broadcast(ShardOps &ops, int q, double *buf, long long count, int root) override {
    hipStream_t s = (hipStream_t)ops.stream(q);
    for (long long o = 0; o < count; o += cap) {
      const int st = piece(s, 0, buf + o, nullptr, nullptr, count - o < cap ? count - o : cap, root);
      if (st != AGP_OK) return st;
    }
    return AGP_OK;
  }
  int all_gather(ShardOps &ops, int q, const double *send, double *recv, long long count) override {
    hipStream_t s = (hipStream_t)ops.stream(q);
    const long long per = cap / world;  // doubles per rank and piece
    if (count <= per) return piece(s, 1, nullptr, send, recv, count, 0);
    // pieces: rank r's part of piece j lands at recv[r * count + j * per ...] - gather each piece into the slot and pull
    // it rank by rank
    for (long long o = 0; o < count; o += per) {
      const long long c = count - o < per ? count - o : per;
      ++seq;
      const long long slot_off = (long long)(seq & 1) * cap;
      hipLaunchKernelGGL(ipc_push_kernel, dim3(grid_for(c)), dim3(256), 0, s, peers, send + o, c, (long long)rank * c, slot_off, seq, -1,
                         ticks, err);
      hipLaunchKernelGGL(ipc_signal_kernel, dim3(1), dim3(64), 0, s, peers, seq, ticks, err);
      for (int r = 0; r < world; ++r)
        hipLaunchKernelGGL(ipc_pull_kernel, dim3(grid_for(c)), dim3(256), 0, s, peers, recv + (long long)r * count + o, c, (long long)r * c,
                           slot_off, 0, -1);
      hipLaunchKernelGGL(ipc_ack_kernel, dim3(1), dim3(64), 0, s, peers, seq);
    }
    if (hipGetLastError() != hipSuccess) { broken = true; return AGP_ERR_HIP; }
    return AGP_OK;
  }
  int all_reduce(ShardOps &ops, int q, double *buf, long long count, int op) override {
    hipStream_t s = (hipStream_t)ops.stream(q);
    const long long per = cap / world;
    for (long long o = 0; o < count; o += per) {
      const int st = piece(s, 2, buf + o, nullptr, nullptr, count - o < per ? count - o : per, op);
      if (st != AGP_OK) return st;
    }
    return AGP_OK;
  }
  int all_reduce_host(double *buf, long long count, int op) override {
    return boot.all_reduce(boot.user, buf, count, op) ? AGP_ERR_COMM : AGP_OK;
  }
  int check_health() override {
    int h = 0;
    if (hipMemcpy(&h, err, sizeof(int), hipMemcpyDeviceToHost) != hipSuccess) return AGP_ERR_HIP;
    if (h) {
      broken = true;
      if (ctx) ctx->last_error = "ipc transport: a peer did not arrive within the deadline";
      return AGP_ERR_COMM;
    }
    return AGP_OK;
  }
};

}  // namespace
}  // namespace agp

using namespace agp;

extern "C" int agp_comm_create_ipc(agp_context *ctx, int nranks, int rank, const agp_comm_callbacks *bootstrap, int64_t mailbox_doubles,
                                   agp_comm **out) {
  if (!ctx || !out || !bootstrap || !bootstrap->all_gather || !bootstrap->all_reduce || nranks < 1 || nranks > IPC_MAX_WORLD || rank < 0 ||
      rank >= nranks)
    return AGP_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  AGP_HIP_CHECK(ctx, hipSetDevice(ctx->device));
  IpcComm *c = new (std::nothrow) IpcComm();
  if (!c) return AGP_ERR_INVALID_ARGUMENT;
  c->ctx = ctx;
  c->boot = *bootstrap;
  c->world = nranks;
  c->rank = rank;
  c->cap = mailbox_doubles > 0 ? mailbox_doubles : (8LL << 20);  // 64 MiB per slot
  c->cap = (c->cap + nranks - 1) / nranks * nranks;
  c->ticks = (unsigned long long)(comm_timeout_seconds() * 1e8);
  const size_t bytes = IpcComm::header_bytes() + sizeof(double) * 2 * (size_t)c->cap;
  int st = AGP_OK;
  auto fail = [&](int code, const char *what) {
    ctx->last_error = what;
    c->broken = true;  // (no barrier in the destructor: the peers may have failed elsewhere)
    delete c;
    return code;
  };
  if (hipMalloc(&c->region, bytes) != hipSuccess) return fail(AGP_ERR_HIP, "hipMalloc (ipc region)");
  if (hipMemset(c->region, 0, IpcComm::header_bytes()) != hipSuccess) return fail(AGP_ERR_HIP, "hipMemset (ipc region)");
  // exchange the handles: 64 bytes = 8 doubles per rank over the bootstrap all-gather (bytes travel untouched)
  static_assert(sizeof(hipIpcMemHandle_t) % sizeof(double) == 0, "handle size");
  constexpr long long HD = sizeof(hipIpcMemHandle_t) / sizeof(double);
  hipIpcMemHandle_t mine;
  std::vector<double> send((size_t)HD + 1), all((size_t)(HD + 1) * (size_t)nranks);
  const bool got = hipIpcGetMemHandle(&mine, c->region) == hipSuccess;
  if (got) std::memcpy(send.data(), &mine, sizeof(mine));
  send[(size_t)HD] = got ? 1. : 0.;
  if (bootstrap->all_gather(bootstrap->user, send.data(), all.data(), HD + 1)) return fail(AGP_ERR_COMM, "ipc bootstrap all-gather failed");
  for (int p = 0; p < nranks; ++p)
    if (all[(size_t)p * (HD + 1) + HD] != 1.) st = AGP_ERR_COMM;
  if (st != AGP_OK) return fail(st, "hipIpcGetMemHandle failed on a rank (HSA_ENABLE_IPC_MODE_LEGACY=0 must be set)");
  c->peers.world = nranks;
  c->peers.me = rank;
  double ok = 1.;
  for (int p = 0; p < nranks; ++p) {
    void *base = c->region;
    if (p != rank) {
      hipIpcMemHandle_t h;
      std::memcpy(&h, &all[(size_t)p * (HD + 1)], sizeof(h));
      if (hipIpcOpenMemHandle(&c->opened[p], h, hipIpcMemLazyEnablePeerAccess) != hipSuccess) {
        (void)hipGetLastError();
        c->opened[p] = nullptr;
        ok = 0.;
        base = c->region;  // (placeholder; the communicator is not handed out)
      } else {
        base = c->opened[p];
      }
    }
    unsigned long long *w = static_cast<unsigned long long *>(base);
    c->peers.flags[p] = w;
    c->peers.acks[p] = w + IPC_MAX_WORLD;
    c->peers.mbox[p] = reinterpret_cast<double *>(static_cast<char *>(base) + IpcComm::header_bytes());
  }
  c->err = reinterpret_cast<int *>(static_cast<unsigned long long *>(c->region) + 2 * IPC_MAX_WORLD);
  // every rank must have every region mapped before anyone pushes (min over ranks)
  double neg = -ok;
  if (bootstrap->all_reduce(bootstrap->user, &neg, 1, 1)) return fail(AGP_ERR_COMM, "ipc bootstrap all-reduce failed");
  if (neg != -1.) return fail(AGP_ERR_COMM, "hipIpcOpenMemHandle failed on a rank");
  agp_comm *h = new (std::nothrow) agp_comm();
  if (!h) return fail(AGP_ERR_INVALID_ARGUMENT, "allocation");
  h->impl = c;
  *out = h;
  return AGP_OK;
}
