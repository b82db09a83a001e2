import ctypes as C, sys
sys.path.insert(0,".")
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_time_trailing_update.restype = C.c_int
lib.agp_debug_time_trailing_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
tot = {0:0.,4:0.,5:0.}
for j in range(1,31):
    M = 16384 - 512*(j+1)
    row = []
    for v in (0,4,5):
        ms = C.c_double()
        lib.agp_debug_time_trailing_update(ctx._h, M, 512, v, 5, C.byref(ms))
        tot[v] += ms.value
        row.append(ms.value)
    m = M//128; T = m*(m+1)//2
    print(M, T, round(T/512,2), " ".join(f"{r:.3f}" for r in row))
print("sum", tot)
