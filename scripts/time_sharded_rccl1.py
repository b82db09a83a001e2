"""The multi-rank schedule with RCCL in the loop on ONE GPU: communicator of size one, AGP_SHARD_FORCE_COMM=1, so every
broadcast / all-gather / all-reduce of the schedule is a real RCCL call on the library's streams (moving data to itself).
Against the plain one-rank fit this shows what the collectives' launch latencies cost per step."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import albatross_amd as ab
from albatross_amd.distributed import Communicator, ShardedGaussianProcessFit
from bench import make_dataset

cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
for n in [int(a) for a in (sys.argv[1:] or ["16384"])]:
    x, y = make_dataset(n, 44)
    for forced in (False, True):
        if forced:
            os.environ["AGP_SHARD_FORCE_COMM"] = "1"
        else:
            os.environ.pop("AGP_SHARD_FORCE_COMM", None)
        ctx = ab.Context(0)  # (the switches are read when the context is created)
        comm = Communicator.rccl(ctx, 1, 0, Communicator.unique_id()) if forced else None
        s = ShardedGaussianProcessFit(ctx, cov, comm)
        ts = []
        for _ in range(4):
            t0 = time.perf_counter()
            res = s.fit(x, y)
            ts.append(time.perf_counter() - t0)
        print(f"N={n} {'RCCL group of one, multi-rank schedule forced' if forced else 'one rank, no transport':48s}: "
              f"{1e3 * min(ts[1:]):7.1f} ms per fit (host enqueue {s.stage(6):.1f} of {s.stage(7):.1f} ms)", flush=True)
        if comm is not None:
            comm.close()
        del s
        ctx.close()
