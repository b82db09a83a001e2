"""Isolated timing of the bulk trailing update: MFMA kernel (v0) vs DPP-broadcast VALU kernel (v2)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_time_trailing_update.restype = C.c_int
lib.agp_debug_time_trailing_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
for M, K in ((15872, 512), (12288, 512), (8192, 512), (8192, 256), (4096, 256), (4096, 128), (2048, 128)):
    line = f"M={M:6d} K={K:4d}:"
    for variant in (0, 2):
        ms = C.c_double()
        st = lib.agp_debug_time_trailing_update(ctx._h, M, K, variant, 5, C.byref(ms))
        flop = K * (M * M + M * 128.0)  # 2 K per entry of the lower 128-tiles ~ M^2/2 + diagonal tiles
        line += f"  v{variant}: {ms.value:8.3f} ms {flop / ms.value / 1e9:6.1f} TF" if st == 0 else f"  v{variant}: status {st}"
    print(line)
