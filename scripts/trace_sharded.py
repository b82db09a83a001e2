"""Three sharded fits of ONE rank's share (TRACE_WORLD="G,r" from the environment, default "8,0"; the transport that moves
nothing, agp_debug_comm_create_null) for rocprofv3
--kernel-trace; scripts/trace_timeline.py analyses the last one."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from bench import make_dataset

from albatross_amd import _capi as capi
world, rank = (int(v) for v in os.environ.get("TRACE_WORLD", "8,0").split(","))
n = int(os.environ.get("TRACE_N", "16384"))
ctx = ab.Context(0)
lib = ctx._lib
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
x, y = make_dataset(n, 44)
s = cov.features(x).as_struct()
comm = C.c_void_p()
assert capi.load_debug().agp_debug_comm_create_null(world, rank, C.byref(comm)) == 0
for _ in range(3):
    h = C.c_void_p()
    lib.agp_sharded_fit_create(ctx._h, comm, ctx.kernel(cov), C.byref(s), C.c_void_p(y.ctypes.data), None, C.byref(h), None, None)
    lib.agp_sharded_fit_destroy(h)
print("done")
