"""Stress of the polling kernels of small fits (one-launch back substitution, step launches): many fits at sizes around the
switch points, single and batched; every fit is checked against the first one of its size (bit-identical information
vectors) and no call may report a timed-out hand-over."""
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
cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
kh = ctx.kernel(cov)
reps = int(os.environ.get("STRESS_REPS", "3000"))
for n in (129, 512, 1000, 1280, 1920, 2047, 2048, 4096):
    x, y = make_dataset(n, 7)
    x_d, y_d = ctx.to_device(x), ctx.to_device(y)
    feats = _device_features(capi, x_d, n)
    ctx.synchronize()
    ref = None
    info = np.empty(n)
    t0 = time.perf_counter()
    r = max(50, reps * 512 // max(n, 512))
    for i in range(r):
        h = C.c_void_p()
        st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None)
        assert st == 0, (n, i, st, ctx.last_error() if hasattr(ctx, "last_error") else "")
        if i % 50 == 0:
            assert lib.agp_fit_download_information(ctx._h, h, info.ctypes.data_as(C.c_void_p)) == 0
            if ref is None:
                ref = info.copy()
            assert np.array_equal(ref, info), (n, i, np.abs(ref - info).max())
        lib.agp_fit_destroy(h)
    print(f"N={n}: {r} fits ok, {1e3 * (time.perf_counter() - t0) / r:.3f} ms each (incl. checks)", flush=True)
# batched
for n, B in ((512, 256), (1024, 64), (520, 32)):
    xs_d, feats = [], []
    ys = np.empty((n, B), order="F")
    for b in range(B):
        x, y = make_dataset(n, 100 + b)
        xs_d.append(ctx.to_device(x))
        ys[:, b] = y
        feats.append(_device_features(capi, xs_d[-1], n))
    y_d = ctx.to_device(ys.T.copy())
    kernels = (C.c_void_p * B)(*([kh] * B))
    fptrs = (C.c_void_p * B)(*[C.addressof(f) for f in feats])
    out = (C.c_void_p * B)()
    status = (C.c_int * B)()
    info = np.empty((n, B), order="F")
    ref = None
    for i in range(max(20, reps // 30)):
        st = lib.agp_fit_create_batch(ctx._h, B, kernels, fptrs, C.c_void_p(y_d.ptr), n, None, 0, out, info.ctypes.data_as(C.c_void_p), n, None, status)
        assert st == 0 and all(s == 0 for s in status), (n, B, i, st)
        if ref is None:
            ref = info.copy()
        assert np.array_equal(ref, info), (n, B, i)
        for b in range(B):
            lib.agp_fit_destroy(C.c_void_p(out[b]))
    print(f"N={n} B={B}: batches ok", flush=True)
