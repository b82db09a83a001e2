"""Bulk update on a CU-masked stream + the panel chain on the main stream at the same time: does leaving a few CUs to
the chain pay?  (agp_debug_time_masked_update)"""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab

ctx = ab.Context(0)
lib = ab._capi.load_debug()
lib.agp_debug_time_masked_update.restype = C.c_int
lib.agp_debug_time_masked_update.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int,
                                             C.POINTER(C.c_double), C.POINTER(C.c_double)]


def run(M, mask_bits, chain_reps, reps=6, K=512):
    ms, us = C.c_double(), C.c_double()
    if mask_bits is None:
        st = lib.agp_debug_time_masked_update(ctx._h, M, K, 0, reps, None, 0, chain_reps, C.byref(ms), C.byref(us))
    else:
        words = np.zeros(8, dtype=np.uint32)
        for i in mask_bits:
            words[i // 32] |= np.uint32(1 << (i % 32))
        st = lib.agp_debug_time_masked_update(ctx._h, M, K, 0, reps, C.c_void_p(words.ctypes.data), 8, chain_reps, C.byref(ms), C.byref(us))
    assert st == 0, st
    return ms.value, us.value


masks = {
    "no mask": None,
    "all 256 set": list(range(256)),
    "first 224": list(range(224)),
    "drop i%8==7 (224)": [i for i in range(256) if i % 8 != 7],
    "drop i%32>=28 (224)": [i for i in range(256) if i % 32 < 28],
    "drop i%32>=30 (240)": [i for i in range(256) if i % 32 < 30],
    "first 240": list(range(240)),
}
for M in (7680, 4096):
    flop = M * M / 2 * 512 * 2
    for name, bits in masks.items():
        ms0, _ = run(M, bits, 0)
        # chain concurrently: enough panel phases to cover the bulk launches
        ms1, us1 = run(M, bits, 12)
        print(f"M={M} {name:22s}: bulk alone {ms0:6.3f} ms ({flop / ms0 / 1e9:5.1f} TF/s); with the chain alongside: bulk {ms1:6.3f} ms, "
              f"chain {us1:7.1f} us per 512-wide panel phase", flush=True)
