"""Issue rate of the two fp64 MFMA shapes of gfx950 (bare loops): 16x16x4 vs 4x4x4 (4 blocks)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_mfma_shape.restype = C.c_int
lib.agp_debug_mfma_shape.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_double)]
out = (C.c_double * 2)()
for mode, nacc in ((0, 8), (1, 16), (1, 64), (2, 32), (2, 64)):
    for wps in (1, 2, 4):
        iters = 4000 if mode == 0 else 16000
        st = lib.agp_debug_mfma_shape(ctx._h, mode, nacc, wps, iters, out)
        print(("16x16x4" if mode == 0 else ("4x4x4x4b" if mode == 1 else "4x4x4x4b, 16 A x 4 B operand registers")), "nacc", nacc, "waves/simd", wps, "status", st, f"{out[0]:.1f} TFLOP/s  {out[1]:.3f} ms")

print("random operand mantissas (data-dependent power), 4x4x4x4b, 16 A x 4 B registers, nacc 64:")
for wps in (1, 2, 4):
    for iters in (16000, 64000):
        lib.agp_debug_mfma_shape(ctx._h, 2, 64, wps, -iters, out)
        print("  waves/simd", wps, "iters", iters, f"{out[0]:.1f} TFLOP/s  {out[1]:.3f} ms")
