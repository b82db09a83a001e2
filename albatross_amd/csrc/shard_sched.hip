// shard_sched.hip — the backend-agnostic half of the sharded fit (shard.h): buffer carving, the schedule of
// the row-block-cyclic LL^T with one block column of look-ahead, the two substitutions, and the callback backends
// (CallbackShardOps / CallbackComm) behind agp_shard_factor_custom / agp_comm_create_callbacks.
//
// Reference work replaced: Fit<GPFit>::Fit (include/albatross/src/models/gp.hpp:61-69) - SerializableLDLT(cov)
// (eigen/serializable_ldlt.hpp:27) and information = ldlt.solve(y) - for ONE dataset over several GPUs.
//
// No HIP call is made in this file: with callback backends it runs on machines without a GPU.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "shard_internal.h"

namespace agp {

static long long round_even(long long x) { return (x + 1) / 2 * 2; }

long long shard_msg_doubles(const ShardPlan &p) { return p.B * p.B + 4 * SHARD_IMG + round_even(p.B); }

static long long pall_ld(const ShardPlan &p) {
  long long rows = p.n - p.B;
  if (rows < 2) rows = 2;
  long long ld = (rows + 7) / 8 * 8;
  if (ld % 256 == 0) ld += 8;  // consecutive columns on different HBM channels (api.hip: factor_ld)
  return ld;
}

static long long lcol_ld(const ShardPlan &p) { return (SHARD_SUPER - 1) * p.B; }

long long shard_work_doubles(const ShardPlan &p) {
  const long long nlb = p.max_local_blocks();
  const bool multi = p.multi();
  long long total = (multi ? p.nb : 2) * shard_msg_doubles(p) + nlb * 4 * SHARD_IMG + (multi ? p.nb : nlb) * p.B * p.B +
                    2 * round_even(p.n) + round_even(p.B) + 8 + 2 * round_even(p.world);
  if (multi) total += p.nb * lcol_ld(p) * p.B + nlb * p.B * p.B * (1 + p.world) + 2 * pall_ld(p) * p.B;
  return total;
}

void shard_carve(const ShardPlan &p, double *w, ShardBuffers *b) {
  const long long nlb = p.max_local_blocks(), msg = shard_msg_doubles(p);
  const bool multi = p.multi();
  b->msgs = w;
  b->msg_stride = msg;
  b->msg_slots = multi ? p.nb : 2;
  w += b->msg_slots * msg;
  b->img_local = w; w += nlb * 4 * SHARD_IMG;
  b->W = w; w += (multi ? p.nb : nlb) * p.B * p.B;
  b->t = w; w += round_even(p.n);
  b->xfull = w; w += round_even(p.n);
  b->tmp = w; w += round_even(p.B);
  b->stat = w; w += 8 + 2 * round_even(p.world);
  b->ldp = pall_ld(p);
  b->ld_lcol = lcol_ld(p);
  if (multi) {
    b->lcol = w; w += p.nb * b->ld_lcol * p.B;
    b->send = w; w += nlb * p.B * p.B;
    b->recv = w; w += nlb * p.B * p.B * p.world;
    b->pall[0] = w; w += b->ldp * p.B;
    b->pall[1] = w; w += b->ldp * p.B;
  } else {
    b->lcol = b->send = b->recv = b->pall[0] = b->pall[1] = nullptr;
  }
}

void ShardOps::invert_diag_batch(int q, const double *D, long long stride_D, long long ld, long long w, const double *img,
                                 long long stride_img, double *W, long long stride_W, long long count) {
  for (long long i = 0; i < count; ++i) invert_diag(q, D + i * stride_D, ld, w, img + i * stride_img, W + i * stride_W);
}

void ShardOps::pack_msg(int q, double *msg, long long B, const double *D, long long ld, long long w, const double *img,
                        const double *z) {
  copy2d(q, msg, w, D, ld, w, w);
  copy2d(q, msg + B * B, 4 * SHARD_IMG, img, 4 * SHARD_IMG, 4 * SHARD_IMG, 1);
  copy2d(q, msg + B * B + 4 * SHARD_IMG, B, z, B, w, 1);
}

void ShardOps::update_staircase(int q, double *A, long long ld, const double *Q, long long ldq, const ShardPlan &plan, long long k) {
  const long long B = plan.B, nlb = plan.n_local_blocks(plan.rank), w = plan.width(k);
  for (long long li = plan.first_local_after(plan.rank, k + 1); li < nlb; ++li) {
    const long long i = plan.global_block(plan.rank, li), wi = plan.width(i);
    const long long ncols = (i * B + wi) - (k + 2) * B;
    if (ncols > 0) gemm(q, A + li * B + (k + 2) * B * ld, ld, A + li * B + k * B * ld, ld, Q, ldq, wi, ncols, w, false, 1);
  }
}

void ShardOps::gather_panel(int q, double *Pall, long long ldP, const double *recv, long long cnt_rows, long long w,
                            const ShardPlan &plan, long long k) {
  for (long long i = k + 1; i < plan.nb; ++i) {
    const int o = plan.owner(i);
    const long long li = plan.local_index(i) - plan.first_local_after(o, k);
    copy2d(q, Pall + (i - k - 1) * plan.B, ldP, recv + (long long)o * cnt_rows * w + li * plan.B, cnt_rows, plan.width(i), w);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// The schedule.  Queues: QP = panel chain, QB = bulk updates, QC = collectives (see shard.h).  Step k:
//   [owner(k)]  D_k factored (done in the look-ahead part of step k - 1), message packed into its slot
//   QC          broadcast of the message
//   QP          X = A[own rows of blocks > k, block column k] L_kk^-T, y -= X z_k;  pack of X for the all-gather
//   [owner(k+1)] QP: D_{k+1} -= X_{k+1} X_{k+1}^T, factor D_{k+1}, pack its message   (look-ahead: runs while ...)
//   QC          ... all-gather of the packed panel rows, re-ordering into global row order (Pall); the rows of the panel
//               inside block k's super-block are kept (lcol) for the back substitution
//   QP          U1: block column k + 1 of the own row blocks >= k + 2
//   QB          U2: the columns right of block column k + 1 of the own row blocks >= k + 2
// Every dependency between queues is a record / wait pair of the backend (device-side flags in HipShardOps): the host
// enqueues all steps ahead and never blocks between them.
// With one rank the panel is used in place (no pack / gather), U1 covers D_{k+1} and U2 is ONE launch over the whole
// trailing triangle: the launch sequence of the single-GPU factorisation (chol.hip: factor_lower).
// ---------------------------------------------------------------------------------------------------------------
int shard_factor_solve(ShardOps &ops, ShardComm *comm, const ShardPlan &plan, double *A, long long ld, double *y,
                       ShardBuffers &buf, ShardResult *result) {
  const int me = plan.rank;
  const long long B = plan.B, nb = plan.nb, n = plan.n;
  const long long n_loc = plan.local_rows(me);
  const bool multi = plan.multi();
  if (multi && !comm) return AGP_ERR_INVALID_ARGUMENT;
  int st = AGP_OK;
  auto Aat = [&](long long lrow, long long gcol) { return A + lrow + gcol * ld; };
  auto msg_L = [&](long long k) { return buf.msg(k); };
  auto msg_img = [&](long long k) { return buf.msg(k) + B * B; };
  auto msg_z = [&](long long k) { return buf.msg(k) + B * B + 4 * SHARD_IMG; };
  const long long msg_count = shard_msg_doubles(plan);

  // diagonal block of global block b (owned by this rank): factor in place, keep its tile images, pack the message
  auto factor_and_pack = [&](long long b) {
    const long long li = plan.local_index(b), w = plan.width(b);
    double *D = Aat(li * B, b * B);
    double *img = buf.img_local + li * 4 * SHARD_IMG;
    ops.factor_diag(QP, D, ld, w, b * B, img, y + li * B);
    ops.pack_msg(QP, buf.msg(b), B, D, ld, w, img, y + li * B);
    ops.record(EV_MSG, QP);
  };

  const auto t_begin = std::chrono::steady_clock::now();
  auto ms_since = [&](std::chrono::steady_clock::time_point t) {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t).count();
  };
  ops.begin(plan);
  if (plan.owner(0) == me) factor_and_pack(0);
  // the message of block column b from its owner to everybody, on the collectives queue (the owner's pack is awaited there)
  auto broadcast_msg = [&](long long b) -> int {
    if (plan.owner(b) == me) {
      const int stw = ops.wait(QC, EV_MSG);
      if (stw != AGP_OK) return stw;
    }
    const int stb = comm->broadcast(ops, QC, buf.msg(b), msg_count, plan.owner(b));
    if (stb != AGP_OK) return stb;
    ops.record(EV_BCAST, QC);
    return AGP_OK;
  };
  if (multi && nb > 0 && (st = broadcast_msg(0)) != AGP_OK) { comm->mark_broken(); (void)ops.sync_all(); return st; }
  for (long long k = 0; k < nb && st == AGP_OK; ++k) {
    if ((st = ops.step_begin(k)) != AGP_OK) break;
    const int slot = (int)(k & 1);
    const long long w = plan.width(k);
    // (the broadcast of message k was enqueued in step k - 1, AHEAD of that step's all-gather - see below)
    if (multi && (st = ops.wait(QP, EV_BCAST)) != AGP_OK) break;
    if (k == nb - 1) break;
    const long long li0 = plan.first_local_after(me, k);
    const long long nrows = n_loc - li0 * B > 0 ? n_loc - li0 * B : 0;
    const long long w1 = plan.width(k + 1);
    const int o1 = plan.owner(k + 1);
    // Round 5: the OWNER of block k + 1 solves only ITS BLOCK ROW k + 1 first (its first row block behind k), brings
    // D_{k+1} up to date, factors it and packs the message - and the broadcast of that message goes onto the collectives
    // queue AHEAD of this step's all-gather.  The chain that a different rank carries every step is then
    //   broadcast(k) -> 512 rows of panel solve -> D_{k+1} update -> 4 POTRF steps -> pack -> broadcast(k + 1)
    // instead of  broadcast(k) -> ALL the owner's rows -> ... -> pack -> all-gather(k) -> broadcast(k + 1): the owner's
    // other rows, the all-gather and U1 / U2 of the step run beside it.
    const bool chain_step = multi && ops.owner_first(k);
    const bool own_next = chain_step && o1 == me && nrows > 0;
    const long long head = own_next ? (w1 < nrows ? w1 : nrows) : 0;  // rows solved ahead of the look-ahead (block row k + 1)
    if (nrows > 0 && !own_next)
      ops.trsm_rows(QP, Aat(li0 * B, k * B), ld, nrows, w, msg_L(k), msg_img(k), msg_z(k), y + li0 * B);
    if (head > 0)
      ops.trsm_rows(QP, Aat(li0 * B, k * B), ld, head, w, msg_L(k), msg_img(k), msg_z(k), y + li0 * B);
    const double *Q;
    long long ldq;
    const int ev_u2_prev = ((k - 1) & 1) ? EV_U2_B : EV_U2_A, ev_u2_cur = (k & 1) ? EV_U2_B : EV_U2_A;
    if (multi) {
      const long long cnt_rows = plan.max_blocks_after(k) * B;
      if (!chain_step) {  // bulk-bound step: this rank's panel rows go to the all-gather BEFORE the look-ahead, as in rounds 3-4
        if (nrows > 0) ops.copy2d(QP, buf.send, cnt_rows, Aat(li0 * B, k * B), ld, nrows, w);
        ops.record(EV_PACK, QP);
      }
      // look-ahead: the owner of block k + 1 has everything D_{k+1} still needs in its own panel rows
      // (U2(k - 1) wrote D_{k+1} and the columns U1(k) is about to update)
      if (k >= 1 && (st = ops.wait(QP, ev_u2_prev)) != AGP_OK) break;
      if (o1 == me) {
        const long long li1 = plan.local_index(k + 1);
        const double *X1 = Aat(li1 * B, k * B);
        ops.gemm(QP, Aat(li1 * B, (k + 1) * B), ld, X1, ld, X1, ld, w1, w1, w, true, 0);
        factor_and_pack(k + 1);
      }
      if (chain_step && (st = broadcast_msg(k + 1)) != AGP_OK) break;
      if (own_next && nrows > head)  // the owner's remaining rows, off its chain
        ops.trsm_rows(QP, Aat(li0 * B + head, k * B), ld, nrows - head, w, msg_L(k), msg_img(k), msg_z(k), y + li0 * B + head);
      if (chain_step) {
        if (nrows > 0) ops.copy2d(QP, buf.send, cnt_rows, Aat(li0 * B, k * B), ld, nrows, w);
        ops.record(EV_PACK, QP);
      }
      if ((st = ops.wait(QC, EV_PACK)) != AGP_OK) break;
      if ((st = ops.wait(QC, ev_u2_cur)) != AGP_OK) break;  // U2(k - 2) read pall[slot]
      st = comm->all_gather(ops, QC, buf.send, buf.recv, cnt_rows * w);
      if (st != AGP_OK) break;
      ops.gather_panel(QC, buf.pall[slot], buf.ldp, buf.recv, cnt_rows, w, plan, k);
      ops.record(EV_GATHER, QC);
      if (!chain_step && (st = broadcast_msg(k + 1)) != AGP_OK) break;  // (bulk-bound step: behind the all-gather, as before)
      if ((st = ops.wait(QP, EV_GATHER)) != AGP_OK) break;
      Q = buf.pall[slot];
      ldq = buf.ldp;
      // U1: block column k + 1 of ALL own row blocks >= k + 2 (they are contiguous local rows): one rectangle
      const long long li2 = plan.first_local_after(me, k + 1);
      const long long rows2 = n_loc - li2 * B;
      if (rows2 > 0) {
        ops.gemm(QP, Aat(li2 * B, (k + 1) * B), ld, Aat(li2 * B, k * B), ld, Q, ldq, rows2, w1, w, false, 0);
        // U2: the columns from (k + 2) B to the end of every row block's own diagonal block (a staircase)
        if ((st = ops.wait(QB, EV_GATHER)) != AGP_OK) break;
        ops.update_staircase(QB, A, ld, Q + B, ldq, plan, k);
      }
      // the panel rows inside block k's super-block, kept for the back substitution - on the bulk queue, behind U2 (off
      // the chain; pall[slot] is rewritten by the all-gather of step k + 2, which waits for the record below)
      {
        const long long rows_in = plan.super_end_row(k / SHARD_SUPER) - (k + 1) * B;
        if (rows_in > 0) {
          if (rows2 <= 0 && (st = ops.wait(QB, EV_GATHER)) != AGP_OK) break;
          ops.copy2d(QB, buf.lcol + k * buf.ld_lcol * B, buf.ld_lcol, buf.pall[slot], buf.ldp, rows_in, w);
        }
      }
      ops.record(ev_u2_cur, QB);
    } else {
      // one rank: the panel stays where it is
      ops.record(EV_TRSM, QP);
      Q = Aat((k + 1) * B, k * B);
      ldq = ld;
      if (k >= 1 && (st = ops.wait(QP, ev_u2_prev)) != AGP_OK) break;
      const long long below1 = n - (k + 1) * B;
      ops.gemm(QP, Aat((k + 1) * B, (k + 1) * B), ld, Q, ldq, Q, ldq, below1, w1, w, true, 0);  // U1 incl. D_{k+1}
      const long long below2 = n - (k + 2) * B;
      if (below2 > 0) {
        if ((st = ops.wait(QB, EV_TRSM)) != AGP_OK) break;
        const double *P2 = Aat((k + 2) * B, k * B);
        ops.gemm(QB, Aat((k + 2) * B, (k + 2) * B), ld, P2, ld, P2, ld, below2, below2, w, true, 1);
        ops.record(ev_u2_cur, QB);
      }
      factor_and_pack(k + 1);
    }
  }
  if (st != AGP_OK) {
    if (comm && st == AGP_ERR_COMM) comm->mark_broken();
    (void)ops.sync_all();
    return st;
  }

  // ---- information = L^-T z from the bottom (gp.hpp:68) ----------------------------------------------------------
  if (result) result->enqueue_factor_ms = ms_since(t_begin);
  const auto t_solve = std::chrono::steady_clock::now();
  // Everything of this phase - arithmetic and collectives - is enqueued on ONE queue (QC): the chain is serial
  // anyway, and all collectives of the communicator stay on one stream in one order.
  ops.record(EV_DONE_B, QB);
  st = ops.wait(QC, EV_DONE_B);
  ops.record(EV_DONE_P, QP);
  if (st == AGP_OK) st = ops.wait(QC, EV_DONE_P);
  const int QS = QC;
  const long long nlb = plan.n_local_blocks(me);
  ops.fill_zero(QS, buf.t, n);
  if (!multi && st == AGP_OK) {
    // one rank: row block by row block; t[c] collects sum_r L[r][c] x[r] over the rows solved so far
    for (long long li = 0; li < nlb; ++li) {
      const long long i = plan.global_block(me, li);
      ops.invert_diag(QS, Aat(li * B, i * B), ld, plan.width(i), buf.img_local + li * 4 * SHARD_IMG, buf.W + li * B * B);
    }
    ops.fill_zero(QS, buf.xfull, n);
    for (long long i = nb - 1; i >= 0; --i) {
      const long long w = plan.width(i), li = plan.local_index(i);
      ops.axpby(QS, w, 1., y + li * B, -1., buf.t + i * B, buf.tmp);
      ops.colvec_dot(QS, buf.W + li * B * B, w, w, w, buf.tmp, 1., 0., nullptr, buf.xfull + i * B);  // x_i = inv(L_ii)^T (z_i - S_i)
      if (i > 0) ops.colvec_dot(QS, Aat(li * B, 0), ld, w, i * B, buf.xfull + i * B, 1., 1., buf.t, buf.t);
    }
  } else if (st == AGP_OK) {
    // Several ranks: super-block by super-block (SHARD_SUPER row blocks).  Every rank holds every diagonal block and
    // z (the message slots) and the sub-diagonal blocks inside the super-blocks (lcol), so a super-block is solved
    // REDUNDANTLY by everybody - identical arithmetic on identical data: bit-identical x on every rank - and only the
    // contributions of a rank's own rows to the EARLIER super-blocks travel: one all-reduce of <= SHARD_SUPER * B
    // doubles per super-block (7 at N = 16384 instead of 32 + one of N doubles).
    const long long full = (n % B == 0) ? nb : nb - 1;  // diagonal blocks of full width (only the last can be narrower)
    if (full > 0) ops.invert_diag_batch(QS, msg_L(0), buf.msg_stride, B, B, msg_img(0), buf.msg_stride, buf.W, B * B, full);
    if (full < nb) ops.invert_diag(QS, msg_L(nb - 1), plan.width(nb - 1), plan.width(nb - 1), msg_img(nb - 1), buf.W + (nb - 1) * B * B);
    const long long nsb = plan.n_super();
    for (long long sb = nsb - 1; sb >= 0 && st == AGP_OK; --sb) {
      const long long b0 = sb * SHARD_SUPER, b1 = plan.super_end_block(sb), r0 = b0 * B, r1 = plan.super_end_row(sb);
      // t[r0, r1): the sum over ranks of the contributions of all later super-blocks (nothing yet for the last one)
      if (sb < nsb - 1) st = comm->all_reduce(ops, QS, buf.t + r0, r1 - r0, 0);
      if (st != AGP_OK) break;
      for (long long c = b1 - 1; c >= b0; --c) {  // left-looking inside the super-block
        const long long w = plan.width(c), below = r1 - (c + 1) * B;
        if (below > 0)  // t_c += L[rows of the super-block below c, c]^T x[those rows]
          ops.colvec_dot(QS, buf.lcol + c * buf.ld_lcol * B, buf.ld_lcol, below, w, buf.xfull + (c + 1) * B, 1., 1., buf.t + c * B,
                         buf.t + c * B);
        ops.axpby(QS, w, 1., msg_z(c), -1., buf.t + c * B, buf.tmp);
        ops.colvec_dot(QS, buf.W + c * B * B, w, w, w, buf.tmp, 1., 0., nullptr, buf.xfull + c * B);  // x_c = inv(L_cc)^T (z_c - t_c)
      }
      if (r0 > 0)  // the own rows of this super-block -> the columns of the earlier super-blocks
        for (long long c = b0; c < b1; ++c)
          if (plan.owner(c) == me) {
            const long long li = plan.local_index(c);
            ops.colvec_dot(QS, Aat(li * B, 0), ld, plan.width(c), r0, buf.xfull + c * B, 1., 1., buf.t, buf.t);
          }
    }
  }
  if (result) result->enqueue_solve_ms = ms_since(t_solve);
  const int st_sync = ops.sync_all();
  if (result) result->total_ms = ms_since(t_begin);
  if (st == AGP_OK) st = st_sync;
  if (st == AGP_OK && comm) st = comm->check_health();
  if (st != AGP_OK) {
    if (comm && st == AGP_ERR_COMM) comm->mark_broken();
    return st;
  }

  // ---- status: every rank learns the log-determinant and the first bad pivot (ONE all-gather of two doubles) ----
  double s[2];
  ops.status(s);
  double code = s[1] > 0. ? (double)(n + 1) - s[1] : 0.;  // larger = earlier pivot; max over ranks = the first one
  if (multi) {
    const int world = comm->world;
    std::vector<double> host((size_t)(2 + 2 * world));
    host[0] = s[0];
    host[1] = code;
    double *dev = buf.stat;
    if (ops.device_memory()) {
      if ((st = ops.from_host(QC, host.data(), dev, 2)) != AGP_OK) return st;
    } else {
      dev[0] = host[0]; dev[1] = host[1];
    }
    if ((st = comm->all_gather(ops, QC, dev, dev + 2, 2)) != AGP_OK) return st;
    if (ops.device_memory()) {
      if ((st = ops.to_host(QC, dev + 2, host.data() + 2, 2 * world)) != AGP_OK) return st;
    } else {
      if ((st = ops.sync_all()) != AGP_OK) return st;
      for (int i = 0; i < 2 * world; ++i) host[(size_t)(2 + i)] = dev[2 + i];
    }
    s[0] = 0.;
    code = 0.;
    for (int r = 0; r < world; ++r) {  // rank order: the same sum on every rank
      s[0] += host[(size_t)(2 + 2 * r)];
      if (host[(size_t)(3 + 2 * r)] > code) code = host[(size_t)(3 + 2 * r)];
    }
  }
  if (result) {
    result->log_det = 2. * s[0];
    result->bad_pivot = code > 0. ? (long long)((double)(n + 1) - code) - 1 : -1;
  }
  return code > 0. ? AGP_ERR_NOT_POSITIVE_DEFINITE : AGP_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// callback backends
// ---------------------------------------------------------------------------------------------------------------
int CallbackComm::staged(ShardOps &ops, int q, double *buf, long long count, int kind, int arg, const double *send,
                         long long send_count) {
  // host transport: device data is staged through host memory around the caller's collective
  if (!ops.device_memory()) {
    if (kind == 0) return cb.broadcast(cb.user, buf, count, arg) ? AGP_ERR_COMM : AGP_OK;
    if (kind == 1) return cb.all_gather(cb.user, send, buf, send_count) ? AGP_ERR_COMM : AGP_OK;
    return cb.all_reduce(cb.user, buf, count, arg) ? AGP_ERR_COMM : AGP_OK;
  }
  int st;
  if ((long long)stage.size() < count + send_count) stage.resize((size_t)(count + send_count));
  double *h = stage.data(), *hs = h + count;
  if (kind == 1) {
    if ((st = ops.to_host(q, send, hs, send_count)) != AGP_OK) return st;
    if (cb.all_gather(cb.user, hs, h, send_count)) return AGP_ERR_COMM;
  } else {
    if ((st = ops.to_host(q, buf, h, count)) != AGP_OK) return st;
    const int rc = kind == 0 ? cb.broadcast(cb.user, h, count, arg) : cb.all_reduce(cb.user, h, count, arg);
    if (rc) return AGP_ERR_COMM;
  }
  return ops.from_host(q, h, buf, count);
}

int CallbackComm::broadcast(ShardOps &ops, int q, double *buf, long long count, int root) {
  return staged(ops, q, buf, count, 0, root, nullptr, 0);
}
int CallbackComm::all_gather(ShardOps &ops, int q, const double *send, double *recv, long long count) {
  return staged(ops, q, recv, count * world, 1, 0, send, count);
}
int CallbackComm::all_reduce(ShardOps &ops, int q, double *buf, long long count, int op) {
  return staged(ops, q, buf, count, 2, op, nullptr, 0);
}
int CallbackComm::all_reduce_host(double *buf, long long count, int op) {
  return cb.all_reduce(cb.user, buf, count, op) ? AGP_ERR_COMM : AGP_OK;
}

}  // namespace agp

using namespace agp;

extern "C" {

int agp_comm_create_callbacks(int nranks, int rank, const agp_comm_callbacks *cb, agp_comm **out) {
  if (!out || !cb || nranks < 1 || rank < 0 || rank >= nranks || !cb->broadcast || !cb->all_gather || !cb->all_reduce)
    return AGP_ERR_INVALID_ARGUMENT;
  CallbackComm *c = new (std::nothrow) CallbackComm(*cb);
  if (!c) return AGP_ERR_INVALID_ARGUMENT;
  c->world = nranks;
  c->rank = rank;
  agp_comm *h = new (std::nothrow) agp_comm();
  if (!h) { delete c; return AGP_ERR_INVALID_ARGUMENT; }
  h->impl = c;
  *out = h;
  return AGP_OK;
}

void agp_comm_destroy(agp_comm *comm) {
  if (!comm) return;
  delete comm->impl;  // virtual: the RCCL transport destroys its communicator
  delete comm;
}

int agp_comm_size(const agp_comm *comm) { return comm && comm->impl ? comm->impl->world : 1; }
int agp_comm_rank(const agp_comm *comm) { return comm && comm->impl ? comm->impl->rank : 0; }

int agp_comm_all_reduce_host(agp_comm *comm, double *buf, int64_t count, int op) {
  if (!comm || !comm->impl || !buf || count < 0 || (op != 0 && op != 1)) return AGP_ERR_INVALID_ARGUMENT;
  if (count == 0) return AGP_OK;
  return comm->impl->all_reduce_host(buf, count, op);
}

int agp_comm_barrier(agp_comm *comm) {
  double token = 0.;
  return agp_comm_all_reduce_host(comm, &token, 1, 0);
}

int64_t agp_shard_local_rows(int64_t n, int64_t block, int nranks, int rank) {
  if (n <= 0 || block <= 0 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  return ShardPlan(n, block, nranks, rank).local_rows(rank);
}

int64_t agp_shard_global_row(int64_t n, int64_t block, int nranks, int rank, int64_t l) {
  if (n <= 0 || block <= 0 || nranks < 1 || rank < 0 || rank >= nranks || l < 0) return -1;
  const ShardPlan p(n, block, nranks, rank);
  if (l >= p.local_rows(rank)) return -1;
  return p.global_block(rank, l / block) * block + l % block;
}

int agp_shard_owner(int64_t block_index, int nranks) {
  if (block_index < 0 || nranks < 1) return -1;
  return ShardPlan::owner_of(block_index, nranks);
}

}  // extern "C"
