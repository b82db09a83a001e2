// shard_internal.h — the handles behind agp_comm (shared by shard_sched.hip and shard_hip.hip)
#pragma once
#include <vector>

#include "shard.h"

namespace agp {

// a transport that can also serve the host-side control plane (agp_comm_all_reduce_host / agp_comm_barrier)
struct HostReducingComm : ShardComm {
  virtual int all_reduce_host(double *buf, long long count, int op) = 0;
};

// collectives supplied by the caller, on host memory (agp_comm_create_callbacks)
struct CallbackComm : HostReducingComm {
  agp_comm_callbacks cb;
  std::vector<double> stage;
  explicit CallbackComm(const agp_comm_callbacks &c) : cb(c) {}
  int broadcast(ShardOps &ops, int q, double *buf, long long count, int root) override;
  int all_gather(ShardOps &ops, int q, const double *send, double *recv, long long count) override;
  int all_reduce(ShardOps &ops, int q, double *buf, long long count, int op) override;
  int all_reduce_host(double *buf, long long count, int op) override;

 private:
  int staged(ShardOps &ops, int q, double *buf, long long count, int kind, int arg, const double *send, long long send_count);
};

}  // namespace agp

struct agp_comm {
  agp::HostReducingComm *impl = nullptr;
};
