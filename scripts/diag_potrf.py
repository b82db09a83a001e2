"""Cycle breakdown of one potrf_diag_kernel launch (library built with -DAGP_POTRF_TIMING)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_factor.restype = C.c_int
lib.agp_debug_factor.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
rng = np.random.default_rng(0)
n = 128
B = rng.standard_normal((n, n)); A = np.asfortranarray(B @ B.T + n * np.eye(n)); y = rng.standard_normal(n)
for rep in range(3):
    Ad = A.copy(order="F"); yd = y.copy(); ld = C.c_double(); bad = C.c_int64()
    assert lib.agp_debug_factor(ctx._h, Ad.ctypes.data, n, n, yd.ctypes.data, C.byref(ld), C.byref(bad)) == 0
t = (C.c_ulonglong * 64)()
lib.agp_debug_potrf_timing(t)
t = np.array(list(t), dtype=np.int64)
t0 = t[0]
print("load+sync", t[1] - t0, "first potrf16", t[2] - t[1], "sync", t[3] - t[2])
prev = t[3]
for jb in range(8):
    a = t[4 + 4 * jb]
    line = f"jb={jb}: stageA+sync {a - prev}"
    if jb < 7:
        line += f"  w0 syrk {t[5+4*jb]-a}  w0 potrf16+inv {t[6+4*jb]-t[5+4*jb]}  sync(wait others) {t[7+4*jb]-t[6+4*jb]}"
        prev = t[7 + 4 * jb]
    print(line)
print("loop total", t[40] - t[3], "epilogue", t[41] - t[40], "kernel total", t[41] - t0, "cycles (100 MHz ticks x ... s_memtime)")
