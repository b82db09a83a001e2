"""GPU tests of the device-resident boundary (agp_features.location = AGP_DEVICE,
out_location = AGP_DEVICE): exactly the path bench.py times.  Results must be
bit-identical to the host-pointer path (same kernels, same data)."""
import ctypes as C
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import albatross_amd as ab
from albatross_amd import _capi as capi
from conftest import synthetic_3d
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def dev_features(t, n, dim, meas=0):
    f = capi.Features()
    f.n, f.dim, f.n_scale_columns = n, dim, 0
    f.coords = t.data_ptr()
    f.eq_id = None
    f.scales = None
    f.is_measurement = meas
    f.location = capi.DEVICE
    return f


def test_device_resident_fit_and_predict_match_host_path(ctx):
    n, m = 1500, 200
    x, y = synthetic_3d(n, 7)
    xs, _ = synthetic_3d(m, 8)
    yvar = np.full(n, 0.02)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    lib, kh = ctx._lib, ctx.kernel(cov)
    xd, yd, vd = (torch.from_numpy(a).cuda() for a in (x, y, yvar))
    xsd = torch.from_numpy(xs).cuda()
    fx, fxs = dev_features(xd, n, 3), dev_features(xsd, m, 3)
    h = C.c_void_p()
    info = np.empty(n)
    logdet = C.c_double()
    st = lib.agp_fit_create(ctx._h, kh, C.byref(fx), C.c_void_p(yd.data_ptr()), C.c_void_p(vd.data_ptr()), C.byref(h),
                            C.c_void_p(info.ctypes.data), C.byref(logdet))
    assert st == capi.AGP_OK
    # host-pointer path through the Python mirror
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    assert np.array_equal(info, fm.get_fit().information)
    assert logdet.value == fm.get_fit().log_determinant
    ofit = orc.OracleFit(cov, x, y, yvar)
    assert np.abs(info - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
    # device outputs
    out = torch.empty(2 * m, dtype=torch.float64, device="cuda")
    st = lib.agp_predict_marginal(ctx._h, kh, h, C.byref(fxs), C.c_void_p(out.data_ptr()),
                                  C.c_void_p(out.data_ptr() + 8 * m), capi.DEVICE)
    assert st == capi.AGP_OK
    marg = fm.predict(xs).marginal()
    got = out.cpu().numpy()
    assert np.array_equal(got[:m], marg.mean) and np.array_equal(got[m:], marg.covariance)
    mean_d = torch.empty(m, dtype=torch.float64, device="cuda")
    assert lib.agp_predict_mean(ctx._h, kh, h, C.byref(fxs), C.c_void_p(mean_d.data_ptr()), capi.DEVICE) == capi.AGP_OK
    assert np.array_equal(mean_d.cpu().numpy(), fm.predict(xs).mean())
    # NLL with device inputs
    nll = C.c_double()
    assert lib.agp_nll(ctx._h, kh, C.byref(fx), C.c_void_p(yd.data_ptr()), C.c_void_p(vd.data_ptr()), C.byref(nll)) == 0
    assert abs(nll.value - orc.nll_with_variance(cov, x, y, yvar)) <= 1e-6 * n
    # Gram into a device buffer with a padded leading dimension
    ld = n + 6
    Kd = torch.full((ld * m,), float("nan"), dtype=torch.float64, device="cuda")
    assert lib.agp_gram(ctx._h, kh, C.byref(fx), C.byref(fxs), C.c_void_p(Kd.data_ptr()), ld, capi.DEVICE) == 0
    K = Kd.cpu().numpy().reshape(m, ld).T
    assert np.array_equal(K[:n], ctx.gram(cov, x, xs)) and np.all(np.isnan(K[n:]))
    lib.agp_fit_destroy(h)


def test_device_buffers_of_the_c_abi(ctx):
    """agp_device_malloc / agp_memcpy / agp_device_free (include/albatross_amd.h, "device memory"): the buffers bench.py's
    N = 1 path hands to the fit - no torch, no second HIP runtime.  Round trip is exact; a fit on them is bit-identical to
    the host-pointer path; invalid arguments are refused."""
    n = 700
    x, y = synthetic_3d(n, 21)
    xd, yd = ctx.to_device(x), ctx.to_device(y)
    assert np.array_equal(xd.numpy(), x) and np.array_equal(yd.numpy(), y)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    lib, kh = ctx._lib, ctx.kernel(cov)
    f = capi.Features()
    f.n, f.dim, f.n_scale_columns, f.coords, f.eq_id, f.scales, f.is_measurement, f.location = n, 3, 0, xd.ptr, None, None, 0, capi.DEVICE
    h = C.c_void_p()
    info = np.empty(n)
    assert lib.agp_fit_create(ctx._h, kh, C.byref(f), C.c_void_p(yd.ptr), None, C.byref(h), C.c_void_p(info.ctypes.data), None) == 0
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    assert np.array_equal(info, fm.get_fit().information)
    out = ctx.device_empty(n)
    assert lib.agp_predict_mean(ctx._h, kh, h, C.byref(f), C.c_void_p(out.ptr), capi.DEVICE) == 0
    assert np.array_equal(out.numpy(), fm.predict(x).mean())   # (agp_memcpy to the host waits for the context's streams)
    lib.agp_fit_destroy(h)
    p = C.c_void_p()
    assert lib.agp_device_malloc(ctx._h, 0, C.byref(p)) == capi.AGP_ERR_INVALID_ARGUMENT
    assert lib.agp_device_malloc(None, 8, C.byref(p)) == capi.AGP_ERR_INVALID_ARGUMENT
    assert lib.agp_memcpy(ctx._h, None, C.c_void_p(xd.ptr), 8, capi.HOST) == capi.AGP_ERR_INVALID_ARGUMENT
    assert lib.agp_memcpy(ctx._h, C.c_void_p(xd.ptr), C.c_void_p(x.ctypes.data), 8, 7) == capi.AGP_ERR_INVALID_ARGUMENT
    assert lib.agp_device_free(ctx._h, None) == capi.AGP_OK
    for d in (xd, yd, out):
        d.free()
    assert ctx.synchronize() is None


def test_two_contexts_are_independent(ctx):
    """Calls on distinct contexts are concurrent-safe (one context per host thread)."""
    import threading
    x, y = synthetic_3d(900, 3)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    want = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y)).get_fit().information
    results = {}

    def work(tag):
        c = ab.Context(0)
        for _ in range(5):
            results[tag] = ab.gp_from_covariance(cov, context=c).fit(ab.RegressionDataset(x, y)).get_fit().information
        c.close()

    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    for i in range(3):
        assert np.array_equal(results[i], want), (i, [float(np.abs(results[j] - want).max()) for j in range(3)],
                                                  [int((results[j] != want).sum()) for j in range(3)])


def test_bench_json_contract():
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1",
                                   "--n", "4096", "--no-cpu-baseline"], text=True, cwd=ROOT)
    line = [l for l in out.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline"):
        assert key in j, key
    assert j["dtype"] == "f64" and j["n_gpus"] == 1 and j["steps"] == 1 and j["higher_is_better"] is True
    assert j["vs_baseline"] is None and "workload" in j["config"] and "model" not in j["config"]
    r = j["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12
    assert j["value"] > 0 and abs(j["value"] - 1e3 / j["ms_per_step"]) < 1e-6 * j["value"]
    # the timing block is the LAST thing on the line (the driver's record keeps the final 2 kB verbatim) and this process
    # never loaded torch's HIP runtime
    assert list(j)[-1] == "timing" and list(j)[-2] == "stages_ms_per_fit" and line.rstrip().endswith("}}")
    t = j["timing"]
    assert t["hip_runtimes_in_process"] == 1 and t["ms_per_step"] == j["ms_per_step"] and t["max_step_index"] == 0
    assert len(json.dumps({"stages_ms_per_fit": j["stages_ms_per_fit"], "timing": t})) < 1500
