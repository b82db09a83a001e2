"""Cycle stamps of the factoring workgroup of the last panel launch (a -DAGP_POTRF_TIMING build of the two libraries, see
scripts/build_variant.sh probe -DAGP_POTRF_TIMING): per micro step of the 128 x 128 POTRF, when each wave finished its own work and when the
barrier released them.  Wave 0: SYRK of the next diagonal tile + POTRF16 + INV16, waves 1-3: SYRK + y."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset

# (the stamps live in a device global of the module that ran the kernel: the fits must go through the SAME shared object
# that reads them back - the debug library, which carries every product entry point as well)
capi.LIB_NAME = "libalbatross_amd_debug.so"
ctx = ab.Context(0)
dbg = capi.load_debug()
dbg.agp_debug_potrf_probe.argtypes = [C.c_void_p, C.c_void_p]
cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
for n in [int(a) for a in sys.argv[1:]] or [512]:
    x, y = make_dataset(n, 42)
    model = ab.gp_from_covariance(cov, context=ctx)
    for _ in range(20):
        fm = model.fit(ab.RegressionDataset(x, y))
    out = (C.c_ulonglong * 128)()
    assert dbg.agp_debug_potrf_probe(ctx._h, out) == 0
    t = np.array(out[:], dtype=np.int64).reshape(4, 32)
    t0 = t[0, 0]
    print(f"N={n}: entry -> block loaded + first micro tile factored {t[0, 1] - t0} cycles; total to the last barrier {t[0, 15] - t0}")
    for jb in range(7):
        own = [int(t[w, 2 + 2 * jb] - t0) for w in range(4)]
        bar = int(t[0, 3 + 2 * jb] - t0)
        prev = int(t[0, 1 + 2 * jb] - t0)
        print(f"  micro step {jb}: waves done at {own}, barrier at {bar}  (stage A + B took {bar - prev})")
