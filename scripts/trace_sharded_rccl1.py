"""Three N = 16384 fits through the multi-rank schedule with an RCCL communicator of size one (AGP_SHARD_FORCE_COMM=1) for
rocprofv3 --kernel-trace; scripts/trace_timeline.py analyses the last one."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd.distributed import Communicator, ShardedGaussianProcessFit
from bench import make_dataset

os.environ["AGP_SHARD_FORCE_COMM"] = "1"
ctx = ab.Context(0)
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
x, y = make_dataset(16384, 44)
comm = Communicator.rccl(ctx, 1, 0, Communicator.unique_id())
s = ShardedGaussianProcessFit(ctx, cov, comm)
for _ in range(3):
    s.fit(x, y)
print("done")
