"""Wall time and GPU stage times of agp_fit_create on bench.py's workload (3-D SE + noise, inputs resident in HBM) at the
sizes given: one line per size.  No torch, no second HIP runtime; for scripts/ab.sh."""
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
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
kh = ctx.kernel(cov)
for n in [int(a) for a in sys.argv[1:]] or [16384]:
    x, y = make_dataset(n, 44)
    x_d, y_d = ctx.to_device(x), ctx.to_device(y)
    feats = _device_features(capi, x_d, n)
    ctx.synchronize()
    reps = int(os.environ.get("FIT_REPS", "0")) or max(8, min(200, int(0.5 / (3e-2 * (n / 16384.) ** 3 + 2e-4))))

    def fit():
        h = C.c_void_p()
        st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None)
        assert st == 0, st
        lib.agp_fit_destroy(h)
    for _ in range(3):
        fit()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fit()
        ts.append(time.perf_counter() - t0)
    ctx.set_profiling(True)
    fit()
    fit()
    st = [ctx.stage_ms(i) for i in range(6)]
    ctx.set_profiling(False)
    info = np.empty(n)
    hh = C.c_void_p()
    assert lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(hh), C.c_void_p(info.ctypes.data), None) == 0
    lib.agp_fit_destroy(hh)
    ts.sort()
    print(f"N={n}: best {1e3 * ts[0]:.3f} ms, median {1e3 * ts[len(ts) // 2]:.3f} ms ({reps} fits); stages gram {st[0]:.3f} factor {st[1]:.3f} "
          f"backsub {st[2]:.3f}; sum(information) {float(info.sum()):.15e}; bulk launches {st[4]:.0f} x {st[3] / max(st[4], 1):.4f} ms = {st[5] / max(st[3], 1e-9) / 1e9:.2f} TFLOP/s", flush=True)
    x_d.free()
    y_d.free()
