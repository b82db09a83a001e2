cd $GRAFT_REPO_ROOT
python3 - <<'PY'
import ctypes as C, time, os, numpy as np, torch
torch.cuda.init()
import torch.distributed as dist
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset, _device_features
n=16384
ctx=ab.Context(0); cov=ab.SquaredExponential(1.0,1.0)+ab.IndependentNoise(0.1); kh=ctx.kernel(cov)
x,y=make_dataset(n,44); x_d,y_d=torch.from_numpy(x).cuda(),torch.from_numpy(y).cuda(); f=_device_features(torch,capi,x_d,n); torch.cuda.synchronize()
def fit():
    h=C.c_void_p(); ctx._lib.agp_fit_create(ctx._h,kh,C.byref(f),C.c_void_p(y_d.data_ptr()),None,C.byref(h),None,None); ctx._lib.agp_fit_destroy(h)
ctx.set_profiling(True)
fit(); fit()
tick = torch.zeros(1, device="cuda")
for mode in ("plain", "tick", "plain", "tick", "ctxsync"):
    torch.cuda.synchronize(); t1=time.perf_counter()
    for _ in range(10):
        fit()
        if mode == "tick": tick.add_(1.0)
    t2=time.perf_counter()
    if mode == "ctxsync": ctx.synchronize()
    t2b=time.perf_counter(); torch.cuda.synchronize(); t3=time.perf_counter(); torch.cuda.synchronize(); t4=time.perf_counter()
    print(f"{mode}: 10 fits {1e3*(t2-t1):.2f} ms, closing torch sync {1e3*(t3-t2b):.3f} ms, again {1e3*(t4-t3):.3f} ms", flush=True)
PY
BENCH_DEBUG_STEPS=1 python3 bench.py --no-cpu-baseline --no-configs --no-predict 2>&1 | grep "per-step"
