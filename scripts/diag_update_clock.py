import ctypes as C, sys
sys.path.insert(0,".")
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load()
lib.agp_debug_time_trailing_update.restype = C.c_int
lib.agp_debug_time_trailing_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
for M,K in ((15872,512),(31744,512),(8192,512)):
    ms = C.c_double()
    lib.agp_debug_time_trailing_update(ctx._h, M, K, 6, 5, C.byref(ms))
    fl = 2.*K*(M*(M+1)/2)
    print(M,K,ms.value,"ms",fl/ms.value/1e9,"TFLOP/s", flush=True)
