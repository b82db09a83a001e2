"""Where one step launch of the chain-bound tail spends its time (library built with -DAGP_POTRF_TIMING): per launch,
by rows below the panel, microseconds from the start of workgroup 0."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from albatross_amd import _capi as capi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4608
ctx = ab.Context(0)
lib = capi.load_debug()
lib.agp_debug_factor.restype = C.c_int
lib.agp_debug_factor.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
lib.agp_debug_step_timing.argtypes = [C.c_void_p, C.c_int]
rng = np.random.default_rng(0)
B = rng.standard_normal((n, n)); A = np.asfortranarray(B @ B.T + n * np.eye(n)); y = rng.standard_normal(n)
target = int(os.environ.get("ROW_DETAIL", "0") or 0)
if target > 1:
    lib.agp_debug_row_target.argtypes = [C.c_longlong]
    lib.agp_debug_row_target(target)
for rep in range(3):
    lib.agp_debug_step_timing(None, 1)
    Ad = A.copy(order="F"); yd = y.copy(); ld = C.c_double(); bad = C.c_int64()
    assert lib.agp_debug_factor(ctx._h, Ad.ctypes.data, n, n, yd.ctypes.data, C.byref(ld), C.byref(bad)) == 0
t = (C.c_ulonglong * 1024)()
lib.agp_debug_step_timing(t, 0)
t = np.array(list(t), dtype=np.uint64).reshape(64, 16).astype(np.int64)
print("rows_below  wg0: prologue_done  end | row wgs: first_start last_end | trailing: first_start last_end   (us after wg0's start)  next launch's wg0 start")
prev = None
for slot in range(63, -1, -1):
    r = t[slot]
    if r[0] == 0:
        continue
    f = lambda v: f"{(v - r[0]) / 100.:7.1f}" if 0 < v < (1 << 62) else "      -"
    nxt = ""
    if slot > 0 and t[slot - 1][0] > 0:
        nxt = f"{(t[slot - 1][0] - r[0]) / 100.:7.1f}"
    print(f"{slot * 128:6d}      {f(r[1])} {f(r[2])} |  {f(r[6])} {f(r[3])} |  {f(r[4])} {f(r[5])}   {nxt}   | wg0 last diag tile out {f(r[8])} z out {f(r[7])} | rows: trsm done {f(r[9])} z seen {f(r[10])} last: wg {(int(r[11]) >> 4) & 0xffff} wave {int(r[11]) & 15}")

if os.environ.get("ROW_DETAIL"):
    rt = (C.c_ulonglong * 1024)()
    lib.agp_debug_row_timing.argtypes = [C.c_void_p]
    lib.agp_debug_row_timing(rt)
    rt = np.array(list(rt), dtype=np.uint64).reshape(128, 8).astype(np.int64)
    tr = target if target > 1 else 1408
    base = t[tr // 128][0]  # workgroup 0's start of that launch
    print(f"launch with {tr} rows below: workgroup, CU, start, pre-update done, TRSM done / (trailing) end  [us after wg0's start]")
    for w in range(128):
        r = rt[w]
        if r[0] == 0:
            continue
        g = lambda v: f"{(v - base) / 100.:7.1f}" if v > 0 else "      -"
        print(f"wg {w:3d} cu {int(r[4]):#06x}  {g(r[0])} {g(r[1])} {g(r[2])} {g(r[3])}   pre-update: staged0 {g(r[5])} mfma0 {g(r[6])} staged1 {g(r[7])}")
