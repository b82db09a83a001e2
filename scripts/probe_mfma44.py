"""Lane maps of v_mfma_f64_4x4x4_4b_f64 on the device (one-hot operands), with and without A-block broadcast."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import albatross_amd as ab
from albatross_amd import _capi as capi
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_mfma44_probe.restype = C.c_int
lib.agp_debug_mfma44_probe.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
for mode in range(5):
    out = np.zeros(4096, dtype=np.uint64)
    assert lib.agp_debug_mfma44_probe(ctx._h, mode, C.c_void_p(out.ctypes.data)) == 0
    m = out.reshape(64, 64)
    print("mode", mode, "(cbsz=0)" if mode == 0 else f"(cbsz=2, abid={mode-1})")
    for la in range(64):
        hits = [(lb, [l for l in range(64) if (int(m[la, lb]) >> l) & 1]) for lb in range(64) if m[la, lb]]
        if la < 20 or la % 16 == 0:
            print("  A lane", la, "->", " ".join(f"B{lb}:D{d}" for lb, d in hits))
