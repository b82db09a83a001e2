"""Bulk update C -= P P^T: fp64 MFMA (variant 0), fp32-product MFMA on an fp32 panel copy (4), bf16 x 3 on the BF16 pipe (5),
fp16 x 2 of power-of-two-scaled rows (6)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
dbg = capi.load_debug()
dbg.agp_debug_time_trailing_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
for M in [int(a) for a in sys.argv[1:]] or [15872, 30720]:
    for K in (512,):
        for variant, name in ((0, "fp64 MFMA"), (4, "fp32 MFMA (fp32 panel copy)"), (5, "bf16 x 3"), (6, "fp16 x 2 (scaled rows)")):
            ms = C.c_double()
            st = dbg.agp_debug_time_trailing_update(ctx._h, M, K, variant, 5, C.byref(ms))
            flop = M * (M + 1.) * K
            print(f"M={M} K={K} {name:30s} {ms.value:8.3f} ms  {flop / ms.value / 1e9:7.1f} TFLOP/s (status {st})", flush=True)
