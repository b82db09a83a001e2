"""What the library's profiling (HIP events around the bulk launches, stage timers) costs a N = 16384 fit: ms per fit with and without."""
import time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, torch, ctypes as C
torch.cuda.init()
import albatross_amd as ab
from albatross_amd import _capi as capi
import bench
n = 16384
ctx = ab.Context(0); lib = ctx._lib
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
kh = ctx.kernel(cov)
x, y = bench.make_dataset(n, 44)
xd = torch.from_numpy(x).cuda(); yd = torch.from_numpy(y).cuda()
f = capi.Features(); f.n, f.dim, f.n_scale_columns = n, 3, 0; f.coords = xd.data_ptr(); f.eq_id = None; f.scales = None; f.is_measurement = 0; f.location = capi.DEVICE
def fit():
    h = C.c_void_p()
    assert lib.agp_fit_create(ctx._h, kh, C.byref(f), C.c_void_p(yd.data_ptr()), None, C.byref(h), None, None) == 0
    lib.agp_fit_destroy(h)
for prof in (False, True, False, True):
    ctx.set_profiling(prof)
    for _ in range(3): fit()
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(15): fit()
    torch.cuda.synchronize()
    print("profiling", prof, round((time.perf_counter() - t) / 15 * 1e3, 3), "ms per fit")
