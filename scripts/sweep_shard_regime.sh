#!/bin/bash
# regime threshold of the sharded schedule (shard_hip.hip: HipShardOps::begin; AGP_SHARD_MASK_GFLOP) on one
# rank's share of a G-rank fit and on the multi-rank schedule over an RCCL group of one
mkdir -p gpurun_out/r04
N=${1:-16384}
W=${2:-"8,0;8,7;4,0;2,0"}
for g in ${GF:-0 4 8 12 20 40 1000000}; do
  echo "== AGP_SHARD_MASK_GFLOP=$g"
  AGP_SHARD_MASK_GFLOP=$g WORLDS="$W" python scripts/time_sharded_rank.py $N 2>&1 | grep "N="
  if [ "$N" = "16384" ]; then AGP_SHARD_MASK_GFLOP=$g python scripts/time_sharded_rccl1.py 16384 2>&1 | grep "forced"; fi
done
