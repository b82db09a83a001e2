"""N = 4096 fits with device-resident inputs through the C-ABI (for rocprofv3 --hip-trace --kernel-trace --stats: which host
calls of a 2 ms fit cost what)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset, _device_features

n = int(os.environ.get("TRACE_N", "4096"))
ctx = ab.Context(0)
cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
kh = ctx.kernel(cov)
x, y = make_dataset(n, 42)
x_d, y_d = ctx.to_device(x), ctx.to_device(y)
feats = _device_features(capi, x_d, n)
ctx.synchronize()
ts = []
for _ in range(int(os.environ.get("TRACE_REPS", "30"))):
    h = C.c_void_p()
    t0 = time.perf_counter()
    st = ctx._lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None)
    ts.append(time.perf_counter() - t0)
    assert st == 0
    ctx._lib.agp_fit_destroy(h)
print(f"N={n}: best {1e3 * min(ts):.3f} ms, median {1e3 * sorted(ts)[len(ts) // 2]:.3f} ms")
