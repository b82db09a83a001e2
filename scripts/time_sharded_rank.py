"""Time ONE rank's share of a G-rank sharded fit on a one-GPU box (agp_debug_comm_create_null, libalbatross_amd_debug.so:
a transport that moves nothing, so the result is meaningless - the kernels, their shapes and the launch chain are those
of rank r of G; the fit itself runs in the product library)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset

ctx = ab.Context(0)
lib = ctx._lib
dbg = capi.load_debug()
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
for n in [int(a) for a in (sys.argv[1:] or ["16384"])]:
    x, y = make_dataset(n, 44)
    fs = cov.features(x)
    s = fs.as_struct()
    cases = [tuple(int(v) for v in c.split(",")) for c in os.environ["WORLDS"].split(";")] if os.environ.get("WORLDS") else \
        ((1, 0), (2, 0), (4, 0), (8, 0), (8, 7))
    for world, rank in cases:
        comm = C.c_void_p()
        if world > 1:
            assert dbg.agp_debug_comm_create_null(world, rank, C.byref(comm)) == 0
        times = []
        for it in range(4):
            h = C.c_void_p()
            t0 = time.perf_counter()
            st = lib.agp_sharded_fit_create(ctx._h, comm if world > 1 else None, ctx.kernel(cov), C.byref(s), C.c_void_p(y.ctypes.data), None, C.byref(h), None, None)
            times.append(time.perf_counter() - t0)
            assert st in (capi.AGP_OK, capi.AGP_ERR_NOT_POSITIVE_DEFINITE), st  # (garbage in the peers' buffers: any pivot may fail)
            stage = [C.c_double() for _ in range(8)]
            for i in (0, 1, 2, 6, 7):
                lib.agp_sharded_fit_stage(h, i, C.byref(stage[i]))
            lib.agp_sharded_fit_destroy(h)
        if world > 1:
            lib.agp_comm_destroy(comm)
        print(f"N={n} world={world} rank={rank}: {1e3*min(times[1:]):.1f} ms per call (gram {stage[0].value:.2f} ms, factor+solve {stage[1].value:.1f} ms; host enqueue {stage[6].value:.1f} of {stage[7].value:.1f} ms, {'device' if stage[2].value else 'host'} pacing)", flush=True)

# a plain single-GPU fit AFTER the sharded calls (the collectives' queue of the context now exists: does it disturb the
# stream-to-hardware-queue mapping of the plain fit?  DESIGN.md section 8, "a fourth stream")
model = ab.gp_from_covariance(cov, context=ctx)
x, y = make_dataset(16384, 44)
ds = ab.RegressionDataset(x, y)
ts = []
for _ in range(4):
    t0 = time.perf_counter()
    fm = model.fit(ds)
    ts.append(time.perf_counter() - t0)
    del fm
print(f"plain agp_fit_create N=16384 afterwards: {1e3 * min(ts[1:]):.1f} ms", flush=True)
