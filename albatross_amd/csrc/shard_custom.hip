// shard_custom.hip — TEST-ONLY entry points over the backend-agnostic sharded-fit schedule (shard_sched.hip) with the block
// arithmetic supplied by the caller: agp_debug_shard_factor_custom / agp_debug_shard_work_doubles.  Plain C++ (no HIP
// call): linked into libalbatross_amd_debug.so only - never into the product library - and compiled into
// examples/host_sanitize_check under AddressSanitizer / UBSan on machines without a GPU.
#include <cstdlib>
#include <cstring>

#include "shard_internal.h"

#define AGP_DEBUG_API_ __attribute__((visibility("default")))

using namespace agp;

static bool debug_force_comm() {
  const char *e = getenv("AGP_SHARD_FORCE_COMM");
  return e && e[0] == '1';
}

extern "C" {

// The schedule of shard_sched.hip (factorisation + both substitutions) on a rank-local matrix the CALLER built, with the
// block arithmetic supplied through callbacks instead of the HIP kernels: tests/ drives the library's C++ schedule
// with numpy block operations and gloo collectives on CPU-only machines, world size > 1.
//   A      local stacked rows (agp_shard_local_rows x n, leading dimension ld), lower staircase filled
//   y      the targets of the local rows, overwritten;  work: agp_debug_shard_work_doubles doubles of scratch
AGP_DEBUG_API_ int64_t agp_debug_shard_work_doubles(int64_t n, int64_t block, int nranks, int rank) {
  if (n <= 0 || block <= 0 || nranks < 1 || rank < 0 || rank >= nranks) return -1;
  ShardPlan p(n, block, nranks, rank);
  p.force_comm = debug_force_comm();
  return shard_work_doubles(p);
}

AGP_DEBUG_API_ int agp_debug_shard_factor_custom(const agp_shard_ops_callbacks *ops, agp_comm *comm, int64_t n, int64_t block, double *A,
                            int64_t ld, double *y, double *work, double *information, double *log_det,
                            int64_t *bad_pivot) {
  if (!ops || !A || !y || !work || n <= 0 || block <= 0 || block % 128 != 0) return AGP_ERR_INVALID_ARGUMENT;
  if (!ops->factor_diag || !ops->trsm_rows || !ops->gemm || !ops->copy2d || !ops->invert_diag || !ops->colvec_dot ||
      !ops->axpby || !ops->fill_zero)
    return AGP_ERR_INVALID_ARGUMENT;
  const int world = comm && comm->impl ? comm->impl->world : 1, rank = comm && comm->impl ? comm->impl->rank : 0;
  ShardPlan plan(n, block, world, rank);
  plan.force_comm = comm && comm->impl && debug_force_comm();
  if (ld < plan.local_rows(rank)) return AGP_ERR_INVALID_ARGUMENT;
  CallbackShardOps cops(*ops);
  ShardBuffers buf;
  shard_carve(plan, work, &buf);
  ShardResult res;
  const int st = shard_factor_solve(cops, comm ? comm->impl : nullptr, plan, A, ld, y, buf, &res);
  if (log_det) *log_det = res.log_det;
  if (bad_pivot) *bad_pivot = res.bad_pivot;
  if (st == AGP_OK && information) std::memcpy(information, buf.xfull, sizeof(double) * (size_t)n);
  return st;
}



}  // extern "C"
