"""Phase cycle sums of one workgroup of the bf16 x 3 bulk kernel (a -DAGP_BF16_STAMPS build of the two libraries:
scripts/build_variant.sh bf16_stamps -DAGP_BF16_STAMPS, copied over albatross_amd/*.so on the GPU box), and the clock the
chip held during the launch (s_memtime cycles / s_memrealtime 100 MHz ticks)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
dbg = capi.load_debug()
dbg.agp_debug_time_trailing_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
dbg.agp_debug_bf16_probe.argtypes = [C.c_void_p, C.POINTER(C.c_ulonglong)]
NAMES = ("barrier 'stage free'", "wait loads + 12 ds_write", "barrier 'stage full'", "24 ds_read + 96 MFMA")
for M in [int(a) for a in sys.argv[1:]] or [15872, 30720]:
    ms = C.c_double()
    st = dbg.agp_debug_time_trailing_update(ctx._h, M, 512, 5, 5, C.byref(ms))
    out = (C.c_ulonglong * 8)()
    dbg.agp_debug_bf16_probe(ctx._h, out)
    ph, cyc, ticks, nk = list(out[:4]), out[4], out[5], max(1, out[6])
    flop = M * (M + 1.) * 512
    print(f"M={M}: {ms.value:.3f} ms = {flop / ms.value / 1e9:.1f} TFLOP/s (status {st}); one workgroup's loop: {cyc} cycles in "
          f"{ticks} ticks of 10 ns = {100. * cyc / max(1, ticks):.0f} MHz; per chunk of 32 (of {nk}): {cyc / nk:.0f} cycles")
    print(f"    epilogue (C -= acc): {out[7]} cycles = {100. * out[7] / max(1, cyc + out[7]):.0f} % of loop + epilogue")
    for name, v in zip(NAMES, ph):
        print(f"    {name:28s} {v / nk:8.0f} cycles per chunk ({100. * v / max(1, cyc):.0f} %)")
