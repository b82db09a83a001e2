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
for prof in (0,1,0,1):
    ctx.set_profiling(bool(prof))
    tc=[];td=[]
    for _ in range(6):
        h=C.c_void_p(); t0=time.perf_counter(); st=ctx._lib.agp_fit_create(ctx._h,kh,C.byref(f),C.c_void_p(y_d.data_ptr()),None,C.byref(h),None,None); t1=time.perf_counter(); ctx._lib.agp_fit_destroy(h); t2=time.perf_counter(); tc.append(t1-t0); td.append(t2-t1)
    print("profiling", prof, "create ms", [round(1e3*v,2) for v in tc], "destroy", [round(1e3*v,3) for v in td], flush=True)
PY
python3 bench.py --no-cpu-baseline --no-configs --no-predict 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
python3 bench.py --no-cpu-baseline --no-configs --no-predict --steps 20 --warmup 3 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'])"
