cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import ctypes as C, time, numpy as np, torch
torch.cuda.init()
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset, _device_features
n=16384
ctx=ab.Context(0); cov=ab.SquaredExponential(1.0,1.0)+ab.IndependentNoise(0.1); kh=ctx.kernel(cov)
x,y=make_dataset(n,44); x_d,y_d=torch.from_numpy(x).cuda(),torch.from_numpy(y).cuda(); f=_device_features(torch,capi,x_d,n); torch.cuda.synchronize()
def fit():
    h=C.c_void_p(); ctx._lib.agp_fit_create(ctx._h,kh,C.byref(f),C.c_void_p(y_d.data_ptr()),None,C.byref(h),None,None); ctx._lib.agp_fit_destroy(h)
fit(); fit()
for trial in range(3):
    t0=time.perf_counter(); torch.cuda.synchronize(); t1=time.perf_counter()
    for _ in range(5): fit()
    t2=time.perf_counter(); torch.cuda.synchronize(); t3=time.perf_counter(); torch.cuda.synchronize(); t4=time.perf_counter()
    print(f"sync before {1e3*(t1-t0):.3f} ms, 5 fits {1e3*(t2-t1):.2f} ms, sync after {1e3*(t3-t2):.3f} ms, sync again {1e3*(t4-t3):.3f} ms", flush=True)
ctx.set_profiling(True)
for trial in range(2):
    torch.cuda.synchronize(); t1=time.perf_counter()
    for _ in range(5):
        fit(); s=[ctx.stage_ms(k) for k in range(6)]
    t2=time.perf_counter(); torch.cuda.synchronize(); t3=time.perf_counter()
    print(f"profiling on: 5 fits + stage_ms {1e3*(t2-t1):.2f} ms, sync after {1e3*(t3-t2):.3f} ms", flush=True)
PY
