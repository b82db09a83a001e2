"""Throughput of INDEPENDENT fits of bench.py's workload when T host threads, each with a context of its own, issue them
concurrently on one GPU (inputs resident in HBM): the chain-bound tail and the substitution of one fit run beside the bulk
phase of another.  Usage: time_fits_in_flight.py [N] [threads ...]"""
import ctypes as C
import os
import sys
import threading
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import albatross_amd as ab
from albatross_amd import _capi as capi
from bench import make_dataset, _device_features

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
counts = [int(a) for a in sys.argv[2:]] or [1, 2, 3]
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
x, y = make_dataset(n, 44)
FITS = 12


def worker(ctx, kh, feats, y_d, barrier, out, idx):
    lib = ctx._lib

    def fit():
        h = C.c_void_p()
        st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None)
        assert st == 0, st
        lib.agp_fit_destroy(h)
    for _ in range(2):
        fit()
    t0 = time.perf_counter()
    fit()
    single = time.perf_counter() - t0
    barrier.wait()
    # (started together the fits run in lock step - bulk phase beside bulk phase, tail beside tail - and gain nothing: thread i
    # starts i / T of a fit later)
    t_begin = time.perf_counter()
    time.sleep(idx * single / barrier.parties)
    for _ in range(FITS):
        fit()
    out[idx] = time.perf_counter() - t_begin


idle = [ab.Context(0) for _ in range(int(os.environ.get("IDLE_CONTEXTS", "0")))]  # (contexts that only exist: their streams count)
if idle and os.environ.get("IDLE_WARM"):  # ... and have worked before (pools, a mixed fit's buffers), like bench.py's main context
    nw = int(os.environ["IDLE_WARM"])
    xw, yw = make_dataset(nw, 44)
    mw = ab.gp_from_covariance(cov, context=idle[0])
    mw.precision = os.environ.get("IDLE_PRECISION", "fp64")
    fw = mw.fit(ab.RegressionDataset(xw, yw))
    del fw
for t in counts:
    ctxs = [ab.Context(0) for _ in range(t)]
    state = []
    for ctx in ctxs:
        x_d, y_d = ctx.to_device(x), ctx.to_device(y)
        state.append((ctx, ctx.kernel(cov), _device_features(capi, x_d, n), y_d, x_d))
    barrier = threading.Barrier(t)
    out = [0.] * t
    threads = [threading.Thread(target=worker, args=(s[0], s[1], s[2], s[3], barrier, out, i)) for i, s in enumerate(state)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    wall = max(out)
    print(f"N={n}: {t} thread(s) x {FITS} fits: {t * FITS / wall:.2f} fits/s in all ({1e3 * wall / FITS:.2f} ms per fit and thread)", flush=True)
    del state
    for ctx in ctxs:
        ctx.close()
