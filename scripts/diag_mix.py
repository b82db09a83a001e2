import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_mix_clock.restype = C.c_int
lib.agp_debug_mix_clock.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 5)()
for wps in (1, 2, 4):
    for variant in range(9):
        st = lib.agp_debug_mix_clock(ctx._h, wps, variant, 5000, out)
        print(f"waves/simd={wps} mfma={int(out[3])} vfma={int(out[4])}: cycles/iter/wave={out[0]:.0f} clock={out[1]:.3f} chip={out[2]:.1f} TF st={st}")
