// shard.h — one Fit<GPFit> (include/albatross/src/models/gp.hpp:61-69) sharded ROW-block-cyclically over the
// GPUs of a node: the layout arithmetic, the backend-agnostic schedule (factorisation with the fused forward
// substitution, backward substitution) and the two interfaces it is written against:
//
//   ShardOps   block arithmetic on the rank-local stacked matrix.  HipShardOps (shard_hip.hip) = the kernels of
//              the single-GPU fit on HIP streams; CallbackShardOps = C function pointers (tests: numpy).
//   ShardComm  broadcast / all-gather / all-reduce of doubles.  RcclComm (RCCL on a HIP stream) or CallbackComm
//              (collectives supplied by the caller, on host memory).
//
// The schedule never touches matrix memory itself, so the same code runs on device memory with RCCL and on host
// memory with gloo (tests/test_distributed_cpu.py, world sizes 2-4).
#pragma once
#include <cstdint>

#include "../../include/albatross_amd.h"

namespace agp {

constexpr long long SHARD_IMG = 36 * 16 * 16;  // tile image of one 128 x 128 diagonal block (chol.hip)
// Back substitution: SHARD_SUPER consecutive row blocks form a SUPER-BLOCK (2048 rows at 512-row blocks).  Every rank keeps
// the diagonal super-blocks of L (the broadcast diagonal blocks + the first rows of the gathered panels), solves a
// super-block redundantly and exchanges ONE all-reduce per super-block instead of one per row block.
constexpr long long SHARD_SUPER = 4;

struct ShardPlan {
  long long n = 0, B = 512, nb = 0;
  int world = 1, rank = 0;
  // run the multi-rank schedule (pack, all-gather, re-ordering, broadcasts) even with ONE rank: lets a one-GPU box
  // drive every collective of the transport (AGP_SHARD_FORCE_COMM=1; tests)
  bool force_comm = false;
  bool multi() const { return world > 1 || force_comm; }

  ShardPlan() = default;
  ShardPlan(long long n_, long long B_, int world_, int rank_) : n(n_), B(B_), nb((n_ + B_ - 1) / B_), world(world_), rank(rank_) {}

  // boustrophedon ("snake") deal 0..G-1, G-1..0, ...: row block b carries ~b^2 of update work, the snake gives every
  // rank the same sum per pair of rounds (plain b mod G: rank G-1 gets up to 2x rank 0 at 4 blocks per rank)
  static int owner_of(long long b, int world) {
    const long long r = b % world, rnd = b / world;
    return (int)((rnd & 1) ? world - 1 - r : r);
  }
  int owner(long long b) const { return owner_of(b, world); }
  long long width(long long b) const { return (n - b * B < B) ? n - b * B : B; }
  // every rank owns exactly one block per round of G blocks: local index of global block b is b / G
  long long local_index(long long b) const { return b / world; }
  long long global_block(int r, long long li) const { return li * world + ((li & 1) ? world - 1 - r : r); }
  long long n_local_blocks(int r) const {
    long long c = nb / world;
    if (global_block(r, c) < nb) ++c;
    return c;
  }
  long long max_local_blocks() const { return (nb + world - 1) / world; }
  // only the globally last block can be narrower than B, and it is the last local block of its owner
  long long local_rows(int r) const {
    const long long c = n_local_blocks(r);
    if (c == 0) return 0;
    return (c - 1) * B + width(global_block(r, c - 1));
  }
  // smallest local index of rank r whose global block is > k
  long long first_local_after(int r, long long k) const {
    long long li = (k + 1) / world;
    if (global_block(r, li) <= k) ++li;
    return li;
  }
  // blocks > k owned by r
  long long blocks_after(int r, long long k) const {
    const long long c = n_local_blocks(r) - first_local_after(r, k);
    return c > 0 ? c : 0;
  }
  long long max_blocks_after(long long k) const {
    long long m = 0;
    for (int r = 0; r < world; ++r) {
      const long long c = blocks_after(r, k);
      if (c > m) m = c;
    }
    return m;
  }
  // super-blocks of the back substitution
  long long n_super() const { return (nb + SHARD_SUPER - 1) / SHARD_SUPER; }
  long long super_end_block(long long sb) const { return (sb + 1) * SHARD_SUPER < nb ? (sb + 1) * SHARD_SUPER : nb; }  // one past its last block
  long long super_end_row(long long sb) const { return super_end_block(sb) * B < n ? super_end_block(sb) * B : n; }
};

enum ShardQueue { QP = 0, QB = 1, QC = 2 };  // panel chain (high priority) | bulk updates | collectives
enum ShardEvent {
  EV_MSG = 0, EV_BCAST, EV_PACK, EV_GATHER, EV_TRSM, EV_U2_A, EV_U2_B, EV_DONE_B, EV_DONE_P, EV_COUNT
};

struct ShardOps {
  virtual ~ShardOps() {}
  // LL^T of the w x w diagonal block D (ld) in place + z <- L^-1 z on zblk; img: 4 tile images; pivot_base: global
  // index of the block's first row (for the reported pivot)
  virtual void factor_diag(int q, double *D, long long ld, long long w, long long pivot_base, double *img, double *zblk) = 0;
  virtual void trsm_rows(int q, double *X, long long ld, long long nrows, long long w, const double *Lkk, const double *img,
                         const double *z, double *yrows) = 0;
  // bulk != 0: one of the large trailing updates (timed when profiling is on)
  virtual void gemm(int q, double *C, long long ldc, const double *P, long long ldp, const double *Q, long long ldq,
                    long long M, long long N, long long K, bool tri, int bulk) = 0;
  // U2 of step k on the stacked local matrix A (ld): for every own row block i >= k + 2 (local rows li * B ..), its
  // columns (k + 2) B .. end of its own diagonal block -= X_i Q^T with X_i = A[rows of i, block column k] and
  // Q = the panel rows of the global blocks k + 2 .. (ldq).  Default: one gemm per row block.
  virtual void update_staircase(int q, double *A, long long ld, const double *Q, long long ldq, const ShardPlan &plan, long long k);
  virtual void copy2d(int q, double *dst, long long ldd, const double *src, long long lds, long long rows, long long cols) = 0;
  // Pall (rows of the global blocks k+1.., ldP) <- the all-gathered send buffers (default: one copy2d per block)
  virtual void gather_panel(int q, double *Pall, long long ldP, const double *recv, long long cnt_rows, long long w,
                            const ShardPlan &plan, long long k);
  // msg = [L (w x w, leading dimension w) at 0 | the 4 tile images at B * B | z (w) at B * B + 4 SHARD_IMG]
  // (default: three copy2d)
  virtual void pack_msg(int q, double *msg, long long B, const double *D, long long ld, long long w, const double *img,
                        const double *z);
  virtual void invert_diag(int q, const double *D, long long ld, long long w, const double *img, double *W) = 0;
  // `count` diagonal blocks of the same width at constant strides (default: one invert_diag each)
  virtual void invert_diag_batch(int q, const double *D, long long stride_D, long long ld, long long w, const double *img,
                                 long long stride_img, double *W, long long stride_W, long long count);
  virtual void colvec_dot(int q, const double *W, long long ld, long long m, long long n, const double *v, double alpha,
                          double beta, const double *base, double *out) = 0;
  virtual void axpby(int q, long long n, double a, const double *x, double b, const double *y, double *out) = 0;
  virtual void fill_zero(int q, double *p, long long count) = 0;
  // record(ev, q): `ev` completes when everything enqueued on q so far has.  wait(q, ev): whatever is enqueued on q
  // afterwards starts only once the LAST record of `ev` has completed (a no-op if it was never recorded).  How is the
  // backend's business.  HipShardOps: DEVICE pacing - a one-thread kernel behind the producer stores the record's
  // sequence number to a flag in device memory, a one-wave kernel in front of the consumer polls it with a bounded
  // spin: the host enqueues the whole factorisation ahead, no stream ever sits at a hipStreamWaitEvent (on this GPU a
  // stream blocked there slows the dependent launches of every OTHER stream: 121 instead of 44 us per 128 columns of
  // the panel chain, scripts/probe_chain.py) and no host thread spins on hipEventQuery in the data path.  Fallback
  // (HOST pacing, when the queues turn out to share a hardware queue): the round-3 scheme - the panel chain waits with
  // hipStreamWaitEvent, the collectives and bulk queues are fed by the host once their inputs are ready.
  // Returns AGP_OK, AGP_ERR_COMM after the transport's timeout (host pacing), AGP_ERR_HIP on a device error.
  virtual void record(int ev, int q) { (void)ev; (void)q; }
  virtual int wait(int q, int ev) { (void)q; (void)ev; return AGP_OK; }
  // hooks of the schedule: before the first step (the plan is known) and at the top of every block column k, before
  // anything of that step is enqueued.  A backend may change how it paces or where it runs its bulk updates there.
  virtual void begin(const ShardPlan &plan) { (void)plan; }
  virtual int step_begin(long long k) { (void)k; return AGP_OK; }
  // Is step k CHAIN-bound - the owner chain, not this rank's bulk update, decides when the step ends?  Then the owner of
  // block k + 1 solves its own block row first and its broadcast goes ahead of the step's all-gather (shard_sched.hip).
  // The answer must be the same on every rank (it orders the collectives).  Default: alternate, so that a backend
  // without a notion of regimes (the CPU test backend) exercises both orders.
  virtual bool owner_first(long long k) { return (k & 1) != 0; }
  // drain every queue; AGP_OK or an error status
  virtual int sync_all() { return AGP_OK; }
  // {sum of log L_ii over this rank's diagonal blocks, 1 + global index of its first non-positive pivot or 0}
  virtual void status(double out[2]) = 0;
  virtual bool device_memory() const { return false; }
  virtual int to_host(int q, const double *dev, double *host, long long count) { (void)q; (void)dev; (void)host; (void)count; return AGP_ERR_UNSUPPORTED; }
  virtual int from_host(int q, const double *host, double *dev, long long count) { (void)q; (void)dev; (void)host; (void)count; return AGP_ERR_UNSUPPORTED; }
  // a HIP stream for queue q (device backends; nullptr otherwise)
  virtual void *stream(int q) { (void)q; return nullptr; }
};

struct ShardComm {
  int world = 1, rank = 0;
  virtual ~ShardComm() {}
  // all three are enqueued on queue q of `ops` (device transports) or executed at once (host transports)
  virtual int broadcast(ShardOps &ops, int q, double *buf, long long count, int root) = 0;
  virtual int all_gather(ShardOps &ops, int q, const double *send, double *recv, long long count) = 0;
  virtual int all_reduce(ShardOps &ops, int q, double *buf, long long count, int op) = 0;  // op 0 sum, 1 max
  // a wait of the schedule timed out: the communicator may hold a collective no peer will join - abort, not destroy
  virtual void mark_broken() {}
  // after the queues have drained: did a device-side wait of the transport give up (AGP_ERR_COMM)?  RCCL and the host
  // transports report through their return codes and the stream deadlines: AGP_OK.
  virtual int check_health() { return AGP_OK; }
};

// scratch of the schedule, carved out of one allocation of shard_work_doubles(plan) doubles
struct ShardBuffers {
  // One message slot per block column: [L_kk (w x w, ld = w) | 4 tile images | z_k (B)].  Several ranks: EVERY slot is
  // kept (77 MB at N = 16384) - after the factorisation each rank holds all diagonal blocks, their images and all of z,
  // which is what the replicated back substitution needs.  One rank: two slots, used alternately.
  double *msgs = nullptr;
  long long msg_stride = 0, msg_slots = 0;
  double *msg(long long k) const { return msgs + (k % msg_slots) * msg_stride; }
  double *img_local = nullptr;           // tile images of the own diagonal blocks
  double *W = nullptr;                   // inverses of the diagonal blocks (B x B each): the own ones (one rank) / all nb
  // several ranks: per block column k the rows of L below its diagonal block INSIDE its super-block,
  // ((SHARD_SUPER - 1) B) x B at leading dimension ld_lcol, copied from the gathered panel
  double *lcol = nullptr;
  long long ld_lcol = 0;
  double *send = nullptr, *recv = nullptr;
  double *pall[2] = {nullptr, nullptr};
  long long ldp = 0;
  double *t = nullptr, *xfull = nullptr, *tmp = nullptr, *stat = nullptr;  // stat: 2 + 2 * world doubles
};
long long shard_msg_doubles(const ShardPlan &plan);
long long shard_work_doubles(const ShardPlan &plan);
void shard_carve(const ShardPlan &plan, double *work, ShardBuffers *out);

struct ShardResult {
  double log_det = 0.;
  long long bad_pivot = -1;  // global index of the first non-positive pivot, -1 if none
  // host time spent ENQUEUEING the factorisation / the back substitution (ms), and total until the device drained:
  // enqueue ~ total means the host, not the GPU, is the bottleneck
  double enqueue_factor_ms = 0., enqueue_solve_ms = 0., total_ms = 0.;
};

// LL^T of the staircase held in A (local stacked rows, ld) + z = L^-1 y + information = L^-T z.
// On return A holds the rank's rows of L, y its entries of z, buf.xfull the full information vector (replicated).
// Returns AGP_OK / AGP_ERR_NOT_POSITIVE_DEFINITE (the same on every rank) / a transport or HIP error.
int shard_factor_solve(ShardOps &ops, ShardComm *comm, const ShardPlan &plan, double *A, long long ld, double *y,
                       ShardBuffers &buf, ShardResult *result);

}  // namespace agp
