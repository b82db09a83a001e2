cd $GRAFT_REPO_ROOT
for i in 1 2; do python3 bench.py --no-cpu-baseline --no-configs --no-predict 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], d['ms_per_step'], d['stages_ms_per_fit'])"; done
TRACE_N=16384 python3 scripts/trace_config2_api.py 2>&1 | grep -v amdgpu
python3 - <<'PY'
import ctypes as C, time, numpy as np, torch
torch.cuda.init()
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset, _device_features
n=16384
ctx=ab.Context(0); cov=ab.SquaredExponential(1.0,1.0)+ab.IndependentNoise(0.1); kh=ctx.kernel(cov)
x,y=make_dataset(n,44); x_d,y_d=torch.from_numpy(x).cuda(),torch.from_numpy(y).cuda(); f=_device_features(torch,capi,x_d,n); torch.cuda.synchronize()
tc=[];td=[]
for _ in range(8):
    h=C.c_void_p(); t0=time.perf_counter(); st=ctx._lib.agp_fit_create(ctx._h,kh,C.byref(f),C.c_void_p(y_d.data_ptr()),None,C.byref(h),None,None); t1=time.perf_counter(); ctx._lib.agp_fit_destroy(h); t2=time.perf_counter(); tc.append(t1-t0); td.append(t2-t1)
print("create ms", [round(1e3*v,2) for v in tc]); print("destroy ms", [round(1e3*v,3) for v in td])
PY
