"""Three sharded fits of ONE rank's share (AGP_SHARD_FAKE_WORLD from the environment, default "8,0") for rocprofv3
--kernel-trace; scripts/trace_timeline.py analyses the last one."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from bench import make_dataset

os.environ.setdefault("AGP_SHARD_FAKE_WORLD", "8,0")
n = int(os.environ.get("TRACE_N", "16384"))
ctx = ab.Context(0)
lib = ctx._lib
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
x, y = make_dataset(n, 44)
s = cov.features(x).as_struct()
for _ in range(3):
    h = C.c_void_p()
    lib.agp_sharded_fit_create(ctx._h, None, ctx.kernel(cov), C.byref(s), C.c_void_p(y.ctypes.data), None, C.byref(h), None, None)
    lib.agp_sharded_fit_destroy(h)
print("done")
