"""The clock the chip holds under the fp64 bulk update (a -DAGP_BULK_STAMPS build of the two libraries:
scripts/build_variant.sh bulk_stamps -DAGP_BULK_STAMPS, copied over albatross_amd/*.so on the GPU box): back-to-back launches
of trailing_update_kernel at the sizes given, then cycles / ticks of one tile of the last launch.  `fit <N>`: 12 fits of
bench.py's workload back to back instead - the last bulk launch of a fit (the smallest trailing matrix that still takes the
bulk kernel) leaves the stamps."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
# (the stamps live in a device global of the module that ran the kernel: fits go through the debug library, which carries every
# product entry point as well)
capi.LIB_NAME = "libalbatross_amd_debug.so"
ctx = ab.Context(0)
dbg = capi.load_debug()
dbg.agp_debug_time_trailing_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
dbg.agp_debug_bulk_probe.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
if len(sys.argv) > 2 and sys.argv[1] == "fit":
    from bench import make_dataset
    n = int(sys.argv[2])
    x, y = make_dataset(n, 44)
    model = ab.gp_from_covariance(ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1), context=ctx)
    ds = ab.RegressionDataset(x, y)
    for _ in range(12):
        fm = model.fit(ds)
        del fm
    out = (C.c_ulonglong * 4)()
    dbg.agp_debug_bulk_probe(ctx._h, out)
    print(f"N={n}, 12 fits back to back: one tile of the last bulk launch: {out[0]} cycles in {out[1]} ticks of 10 ns = "
          f"{100. * out[0] / max(1, out[1]):.0f} MHz")
    sys.exit(0)
for M in [int(a) for a in sys.argv[1:]] or [15872, 30720]:
    ms = C.c_double()
    reps = max(5, int(400. / (2.3 * (M / 15872.) ** 2)))  # ~0.4 s of back-to-back launches
    st = dbg.agp_debug_time_trailing_update(ctx._h, M, 512, 0, reps, C.byref(ms))
    out = (C.c_ulonglong * 4)()
    dbg.agp_debug_bulk_probe(ctx._h, out)
    flop = M * (M + 1.) * 512
    mhz = 100. * out[0] / max(1, out[1])
    print(f"M={M}: {ms.value:.3f} ms = {flop / ms.value / 1e9:.1f} TFLOP/s over {reps} launches (status {st}); one tile: {out[0]} cycles in "
          f"{out[1]} ticks of 10 ns = {mhz:.0f} MHz -> the fp64 matrix peak at that clock: {78.6 * mhz / 2400.:.1f} TFLOP/s")
