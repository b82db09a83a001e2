"""Bulk trailing update against the depth K of one launch: the per-tile fixed cost (first loads, C read-modify-write, dispatch)
against the K-proportional MFMA loop.  `time_update_k.py [M]`"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_time_trailing_update.restype = C.c_int
lib.agp_debug_time_trailing_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
VARIANT = int(os.environ.get("VARIANT", "0"))  # 0: fp64 MFMA, 3: fp32-product kernel of the mixed-precision fit
Ms = [int(a) for a in sys.argv[1:]] or [15872, 8192]
for M in Ms:
    for K in (128, 256, 512, 768, 1024, 1536, 2048):
        ms = C.c_double()
        st = lib.agp_debug_time_trailing_update(ctx._h, M, K, VARIANT, 5, C.byref(ms))
        tiles = (M // 128) * (M // 128 + 1) // 2
        flop = 2.0 * K * 128 * 128 * tiles
        print(f"M={M:6d} K={K:5d}: {ms.value:8.3f} ms  {flop / ms.value / 1e9:6.1f} TF   {ms.value * 1e3 / (tiles / 512.0):7.1f} us per round of 512 tiles", flush=True)
