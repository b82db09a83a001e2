"""What does a dependent launch cost?  Sequences of tiny kernels on one stream (agp_debug_chain_probe) and the real
panel chain (agp_debug_panel_chain)."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab

ctx = ab.Context(0)
lib = ab._capi.load_debug()
lib.agp_debug_chain_probe.restype = C.c_int
lib.agp_debug_chain_probe.argtypes = [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int,
                                      C.c_int, C.POINTER(C.c_double)]
lib.agp_debug_panel_chain.restype = C.c_int
lib.agp_debug_panel_chain.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int, C.POINTER(C.c_double)]


def probe(kinds, wgs, spin=2000, touch=1, prio=1, reps=200):
    k = (C.c_int * len(kinds))(*kinds)
    w = (C.c_int * len(wgs))(*wgs)
    out = C.c_double()
    assert lib.agp_debug_chain_probe(ctx._h, k, w, len(kinds), reps, spin, touch, prio, C.byref(out)) == 0
    return out.value


print("spin 2000 clocks (~0.9 us), touch 2 KB per WG")
for name, kinds, wgs in [("same kernel, no LDS, 1 WG", [0], [1]), ("same, 73 KB LDS, 1 WG", [2], [1]),
                         ("same, 73 KB LDS, 24 WGs", [2], [24]), ("same, no LDS, 512 WGs", [0], [512]),
                         ("alternate 73 KB / 40 KB LDS, 24 WGs", [2, 1], [24, 24]),
                         ("alternate 77 KB(1 WG) / 73 KB(24) / 40 KB(96)", [3, 2, 1], [1, 24, 96]),
                         ("alternate no LDS(6) / 73 KB(24)", [0, 2], [6, 24]),
                         ("alternate 1 WG / 512 WGs no LDS", [0, 0], [1, 512]),
                         ("alternate 1 WG / 2048 WGs no LDS", [0, 0], [1, 2048])]:
    for prio in (1, 0):
        print(f"  {name:55s} stream prio {'high' if prio else 'low '}: {probe(kinds, wgs, prio=prio):7.2f} us per launch")
print("longer kernels: spin 24000 clocks (~10 us)")
for name, kinds, wgs in [("same, 73 KB LDS, 24 WGs", [2], [24]), ("alternate 77(1) / 73(24) / 40(96)", [3, 2, 1], [1, 24, 96])]:
    print(f"  {name:55s}: {probe(kinds, wgs, spin=24000):7.2f} us per launch")
print("large touch (each WG rewrites 256 KB):")
for name, kinds, wgs in [("same, no LDS, 64 WGs", [0], [64]), ("alternate 64 / 512 WGs", [0, 0], [64, 512])]:
    print(f"  {name:55s}: {probe(kinds, wgs, touch=128):7.2f} us per launch")
for blocked in (0, -1, 1, 2):
    for n, width in ((512, 512), (2560, 512), (16384, 512), (16384, 128)):
        out = C.c_double()
        assert lib.agp_debug_panel_chain(ctx._h, n, width, 50, blocked, C.byref(out)) == 0
        steps = width // 128
        what = {0: "alone", -1: "a 1-workgroup spinner kernel on another stream", -2: "another stream blocked in hipStreamWaitValue32 (no kernel)",
                1: "spinner + 1 stream waiting on its event", 2: "spinner + 2 streams waiting on its event"}[blocked]
        print(f"real panel phase n={n} width={width}, {what}: {out.value:8.1f} us per phase = "
              f"{out.value / steps:6.1f} us per 128 columns")
