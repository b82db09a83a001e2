"""Shader clock held by the bulk trailing update during real fits (library built with -DAGP_CLOCK_PROBE by
scripts/clock_probe.sh): sum of s_memtime cycles / sum of s_memrealtime ticks over all workgroups of
agp::trailing_update_kernel.  NOTE: the probe library in this process is libalbatross_amd_debug.so - the fits run through
it, not through the product library."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from albatross_amd import _capi as capi
# the probe counters live in the library whose kernels run: make the debug library THE library of this process
capi.LIB_NAME = "libalbatross_amd_debug.so"
import albatross_amd as ab
from bench import make_dataset

lib = capi.load()
lib.agp_debug_mfma_kernel_clock.restype = C.c_int
lib.agp_debug_mfma_kernel_clock.argtypes = [C.POINTER(C.c_ulonglong), C.c_int]
ctx = ab.Context(0)
x, y = make_dataset(16384, 44)
model = ab.gp_from_covariance(ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1), context=ctx)
ds = ab.RegressionDataset(x, y)
for _ in range(3):
    model.fit(ds)
out = (C.c_ulonglong * 4)()
lib.agp_debug_mfma_kernel_clock(out, 1)
t0 = time.perf_counter()
reps = 10
for _ in range(reps):
    model.fit(ds)
dt = (time.perf_counter() - t0) / reps
lib.agp_debug_mfma_kernel_clock(out, 1)
cyc, ticks, wgs = out[0], out[1], out[2]
print(f"AGP_XCD_REMAP={os.environ.get('AGP_XCD_REMAP', '0')}: {1e3 * dt:.2f} ms per fit (host inputs); trailing_update_kernel: "
      f"{wgs // reps} workgroups per fit, {cyc / max(wgs, 1):.0f} cycles = {10. * ticks / max(wgs, 1):.0f} ns per workgroup, "
      f"SCLK held {cyc / max(ticks, 1) / 10.:.3f} GHz")
