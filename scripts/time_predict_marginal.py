"""agp_predict_marginal against one resident fit of bench.py's workload, test points and outputs resident in HBM (as the bench
line's `predict` block): one line per (N, M).  For scripts/ab.sh and the AGP_SOLVE_NBO switch."""
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
kh = ctx.kernel(ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1))
for n in [int(a) for a in sys.argv[1:]] or [4096, 16384]:
    x, y = make_dataset(n, 44)
    x_d, y_d = ctx.to_device(x), ctx.to_device(y)
    feats = _device_features(capi, x_d, n)
    h = C.c_void_p()
    assert lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None) == 0
    for m in (1024, 4096):
        xs, _ = make_dataset(m, 43)
        xs_d = ctx.to_device(xs)
        out_d = ctx.device_empty(2 * m)
        fx = _device_features(capi, xs_d, m)
        mean_p, var_p = C.c_void_p(out_d.ptr), C.c_void_p(out_d.ptr + 8 * m)
        call = lambda: lib.agp_predict_marginal(ctx._h, kh, h, C.byref(fx), mean_p, var_p, capi.DEVICE)  # noqa: E731
        assert call() == 0
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        v = out_d.numpy()[m:]
        print(f"N={n} M={m}: marginal best {1e3 * min(ts):.3f} ms, median {1e3 * sorted(ts)[2]:.3f} ms = {n * n * m / min(ts) / 1e12:.1f} TFLOP/s; "
              f"sum(var) {v.sum():.12e}", flush=True)
    lib.agp_fit_destroy(h)
