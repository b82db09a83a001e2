"""v_fmac_f64 issue rate on gfx950 with VGPR / SGPR / DPP row_newbcast src0, and the DPP semantics."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_fmac_rate.restype = C.c_int
lib.agp_debug_fmac_rate.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for mode, name in ((0, "vgpr"), (1, "sgpr"), (2, "dpp row_newbcast")):
    for w in (1, 2):
        out = np.zeros(8); probe = np.zeros(64)
        st = lib.agp_debug_fmac_rate(ctx._h, w, mode, 2000, out.ctypes.data, probe.ctypes.data)
        print(f"{name:16s} waves/simd={w}: status {st} cycles/64fma={out[0]:.1f} clock={out[1]:.3f} GHz chip={out[2]:.1f} TF")
    if mode == 2:
        print("probe (b = lane, row_newbcast:5):", probe.astype(int).tolist())
