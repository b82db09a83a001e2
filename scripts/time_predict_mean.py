"""agp_predict_mean with resident inputs and outputs (what bench.py's `predict` block times): best of 20 per size."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset, _device_features

ctx = ab.Context(0)
lib = ctx._lib
n = int(os.environ.get("PM_N", "16384"))
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
kh = ctx.kernel(cov)
x, y = make_dataset(n, 44)
x_d, y_d = ctx.to_device(x), ctx.to_device(y)
feats = _device_features(capi, x_d, n)
h = C.c_void_p()
assert lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None) == 0
for m in [int(a) for a in sys.argv[1:]] or [4096, 65536]:
    xs, _ = make_dataset(m, 43)
    xs[: min(m, 8)] = x[: min(m, 8)]  # a few test points that ARE training points (the noise term's path)
    xs_d = ctx.to_device(xs)
    fx = _device_features(capi, xs_d, m)
    out = ctx.device_empty(m)
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        assert lib.agp_predict_mean(ctx._h, kh, h, C.byref(fx), C.c_void_p(out.ptr), capi.DEVICE) == 0
        ts.append(time.perf_counter() - t0)
    mu = out.numpy()
    print(f"N={n} M={m}: {1e3 * min(ts):.4f} ms = {m / min(ts) / 1e6:.1f} M pts/s; checksum {float(np.sum(mu)):.15e} first {mu[0]:.15e}", flush=True)
