"""The panel phase alone (agp_debug_panel_chain: POTRF + TRSM + inner updates of one 512-column outer block, nothing
else on the GPU): us per 128 columns.  Run with AGP_PANEL_FUSED=0 / 1 to compare the two-launch path with the fused
panel kernel."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab

ctx = ab.Context(0)
lib = ab._capi.load_debug()
lib.agp_debug_panel_chain.restype = C.c_int
lib.agp_debug_panel_chain.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]
print(f"AGP_PANEL_FUSED={os.environ.get('AGP_PANEL_FUSED', '(default: 1)')}")
for n, width in ((512, 512), (2048, 512), (4096, 512), (6656, 512), (16384, 512), (4096, 128), (16384, 128)):
    out = C.c_double()
    assert lib.agp_debug_panel_chain(ctx._h, n, width, 50, 0, C.byref(out)) == 0
    print(f"panel phase n={n:6d} width={width}: {out.value:8.1f} us per phase = {out.value / (width // 128):6.1f} us per 128 columns")
