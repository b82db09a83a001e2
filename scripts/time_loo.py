import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
ctx = ab.Context(0)
n = 16384
rng = np.random.default_rng(44)
x = rng.uniform(0., 10., (n, 3)); y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
fit = fm.get_fit()
fit.leave_one_out(y)
t = time.perf_counter(); loo = fit.leave_one_out(y); dt = time.perf_counter() - t
print(f"LOO marginals of all {n} points: {dt*1e3:.1f} ms ({n**3/3/dt/1e12:.1f} TFLOP/s on N^3/3)")
perm = rng.permutation(n)
for gs in (512, 64):
    groups = [list(map(int, perm[g * gs:(g + 1) * gs])) for g in range(n // gs)]
    fit.held_out_predictions(y, groups[:2])
    t = time.perf_counter(); preds = fit.held_out_predictions(y, groups); dt = time.perf_counter() - t
    t = time.perf_counter(); pj = fit.held_out_predictions(y, groups, joint=True); dtj = time.perf_counter() - t
    print(f"leave-one-group-out, {len(groups)} groups of {gs}: marginals {dt*1e3:.1f} ms, joints {dtj*1e3:.1f} ms "
          f"(N refits of the other groups would be {len(groups)} fits)")
sizes = rng.integers(300, 700, size=40); bounds = np.cumsum(sizes); bounds = bounds[bounds < n]
ragged = [list(map(int, g)) for g in np.split(perm, bounds) if len(g)]
fit.held_out_predictions(y, ragged[:2])
t = time.perf_counter(); preds = fit.held_out_predictions(y, ragged); dt = time.perf_counter() - t
print(f"leave-one-group-out, {len(ragged)} ragged groups of {min(map(len, ragged))}..{max(map(len, ragged))}: marginals {dt*1e3:.1f} ms")
K = rng.standard_normal((4096, 4100)); K = K @ K.T / 4096 + np.eye(4096)
ab.DenseFactor(K, ctx)
t = time.perf_counter(); f = ab.DenseFactor(K, ctx); dt = time.perf_counter() - t
print(f"DenseFactor n=4096 (host matrix, incl. 134 MB upload): {dt*1e3:.1f} ms")
import ctypes as C
from albatross_amd import _capi as capi
Kf = np.asfortranarray(K)
h = C.c_void_p()
t = time.perf_counter()
st = ctx._lib.agp_factor_create(ctx._h, C.c_void_p(Kf.ctypes.data), 4096, 4096, 0, capi.HOST, C.byref(h)); dt = time.perf_counter() - t
print(f"agp_factor_create n=4096 from a host matrix: {dt*1e3:.1f} ms (status {st})")
ctx._lib.agp_fit_destroy(h)
t = time.perf_counter()
st = ctx._lib.agp_factor_create(ctx._h, C.c_void_p(Kf.ctypes.data), 4096, 4096, 0, capi.HOST, C.byref(h)); dt = time.perf_counter() - t
print(f"  second call: {dt*1e3:.1f} ms")
