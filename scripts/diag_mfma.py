import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_mfma_clock.restype = C.c_int
lib.agp_debug_mfma_clock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.POINTER(C.c_double)]
out = (C.c_double * 3)()
for blocks, wps, nacc in [(256, 1, 8), (256, 2, 8), (256, 3, 8), (256, 4, 8), (256, 4, 4), (256, 4, 1), (256, 6, 4), (256, 8, 4), (256, 8, 1), (256, 2, 1), (256, 3, 4)]:
    st = lib.agp_debug_mfma_clock(ctx._h, blocks, wps, nacc, 20000, 1.1, 0.9, out)
    print(f"blocks={blocks:4d} waves/simd={wps} nacc={nacc}: cycles/mfma/wave={out[0]:.1f} -> per SIMD {out[0]/wps:.1f} clock={out[1]:.3f} GHz chip={out[2]:.1f} TF  st={st}")
