#!/usr/bin/env python3
"""bench.py — GP fits/sec (Gram + LL^T + solve) at N = 16384 fp64 on MI355X.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

A "step" is one `agp_fit_create` on one synthetic 3-D dataset whose inputs are
already resident in HBM: Gram of the measurement-wrapped features + target
variance on the diagonal, in-place LL^T, information vector K^-1 y and log|K|
(include/albatross/src/models/gp.hpp:281-294,61-69 in the reference).
Workload = BASELINE.json config 3's problem (3-D SquaredExponential(1,1) +
IndependentNoise(0.1), N = 16384, fp64), the size the metric is quoted on.

N > 1: `value` is ONE fit of the same problem row-block-sharded over the N GPUs (RCCL inside the library; strong
scaling); N independent fits ("replicas", weak scaling) are reported in an auxiliary block.

Rank 0 prints ONE JSON line (contract in the task statement) including
  roofline      for the dominant kernel (fp64 MFMA trailing update), from HIP
                events recorded on the library's stream around every launch
  cpu_baseline  the oracle (albatross-faithful port: serial Gram + unblocked
                pivoted LDL^T, 1 thread) timed on a bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_TRAIN = 16384
DIM = 3
MFMA_F64_PEAK_TFLOPS = 78.6  # MI355X datasheet FP64 matrix; cross-checked by agp_mfma_f64_peak


def mt19937_uniform(seed, count, lo=0., hi=10.):
    """`std::mt19937 gen(seed); std::uniform_real_distribution<double> dist(lo, hi)` of libstdc++, as the reference's
    benchmarks draw their features (benchmarks/bench_utils.h:25-34): generate_canonical<double, 53> takes two 32-bit
    draws, low word first.  Bit-identical to the compiled generator (tests/test_oracle_golden.py checks it on
    tests/golden/bench512.json)."""
    rs = np.random.RandomState(seed)  # init_genrand(seed) == std::mt19937(seed)
    raw = np.frombuffer(rs.bytes(8 * count), dtype="<u4").astype(np.float64)
    u = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0
    return lo + (hi - lo) * np.minimum(u, np.nextafter(1.0, 0.0))


def make_dataset(n, seed):
    """SURVEY.md section 8d config 3 generator: X ~ U[0,10]^3 from mt19937(seed), row-major;
    y = sum_k sin x_k + 0.1 cos(10 x_0)."""
    x = mt19937_uniform(seed, n * DIM).reshape(n, DIM)
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    return x, y


def _cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _oracle_cflags():
    try:
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "Makefile")) as f:
            for line in f:
                if line.startswith("CFLAGS"):
                    return line.split("=", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _hip_runtimes_loaded():
    """distinct libamdhip64 shared objects mapped into this process (1 = the library's own; torch.cuda would add its wheel's)"""
    try:
        with open("/proc/self/maps") as f:
            return len({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return -1


def _cpu_quota_cores():
    """cores this process may use: the smaller of its affinity mask and its cgroup CPU quota (the boxes of the pool show
    256 hardware threads and a quota of 16: `cpu.max` = "1600000 100000")"""
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    quota = None
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, period = f.read().split()[:2]
        if q != "max":
            quota = float(q) / float(period)
    except (OSError, ValueError):
        pass
    return usable, quota


def cpu_baseline(seconds_budget=60.0):
    """Oracle ("port") timed on host cores: albatross-faithful default = serial Gram + single-threaded unblocked pivoted
    LDL^T (core/model.hpp:20; Eigen 3.3's LDLT has no blocked or parallel path).  BASELINE.md section 2 / SURVEY 8d: the
    fit is timed at N in {1024, 2048, 4096, 6144, 8192} (about 45 s of CPU work), t(N) = a N^3 + b N^2 is fitted by least
    squares and evaluated at N = 16384; the record carries the samples, the coefficients and the largest relative
    residual of the fit, the host's CPU model, its core count and the oracle's compiler flags.  `value` comes from the
    three out-of-cache samples (N >= 4096: two parameters, three points - over-determined since round 6); the cubic
    through ALL samples is reported beside it and the two are printed as a range (`range_fits_per_sec`).  (Both are
    conservative for the CPU: at 16384 the unblocked factor works on a 2 GiB matrix, out of every cache.)"""
    import albatross_amd as ab
    from oracle import oracle_py as orc
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    sizes, samples, spent = (1024, 2048, 4096, 6144, 8192), [], 0.0
    for n in sizes:
        if samples:  # the next size costs (n / n_prev)^3 of the previous one: stop before the budget is blown
            n_prev, t_prev = samples[-1][0], samples[-1][2]
            if spent + t_prev * (n / n_prev) ** 3 * 1.3 > seconds_budget:
                break
        x, y = make_dataset(n, 44)
        t0 = time.perf_counter()
        _ = orc.gram(cov, x, x_meas=True, y_meas=True)
        t_gram = time.perf_counter() - t0
        del _
        t0 = time.perf_counter()
        fit = orc.OracleFit(cov, x, y)
        _ = fit.information
        dt = time.perf_counter() - t0
        del fit
        samples.append((n, t_gram, dt))
        spent += dt + t_gram
    ns = np.array([s[0] for s in samples], dtype=np.float64)
    ts = np.array([s[2] for s in samples])

    def rel_fit(nn, tt):
        """t(N) = a N^3 + b N^2 by relative least squares (every sample counts alike); b < 0 -> the pure cubic law"""
        if len(nn) >= 2:
            M = np.stack([nn ** 3, nn ** 2], axis=1)
            coef, *_ = np.linalg.lstsq(M / tt[:, None], np.ones_like(tt), rcond=None)
            if coef[0] <= 0. or coef[1] < 0.:
                coef = np.array([float(np.sum(nn ** 3 / tt) / np.sum((nn ** 3 / tt) ** 2)), 0.])
            return coef, float(np.abs((M @ coef) / tt - 1.).max())
        return np.array([tt[-1] / nn[-1] ** 3, 0.]), 0.

    # The prescribed fit over ALL samples, and the one `value` uses: the samples whose matrix no longer fits the host's
    # caches (N >= 4096: 128 MiB) - the unblocked factor is memory-bound there (its time per N^3 doubles between N = 1024
    # and 6144 on an EPYC 9575F), which is the regime of N = 16384 (2 GiB); the all-sample fit is reported beside it with
    # its residual, which says how badly one cubic describes both regimes.
    coef_all, resid_all = rel_fit(ns, ts)
    big = ns >= 4096
    coef, resid = rel_fit(ns[big], ts[big]) if big.sum() >= 1 else (coef_all, resid_all)
    scaled = float(coef[0] * N_TRAIN ** 3 + coef[1] * N_TRAIN ** 2)
    scaled_all = float(coef_all[0] * N_TRAIN ** 3 + coef_all[1] * N_TRAIN ** 2)
    _usable, _quota = _cpu_quota_cores()
    cores_host = int(max(1, min(_usable, _quota if _quota else _usable)))  # (threads beyond the cgroup quota only add contention)
    out = {"value": 1.0 / scaled, "unit": "fits/sec", "cores": 1, "kind": "port",
           "sample": "oracle fits (serial Gram + unblocked pivoted LDLT, 1 thread) at N = "
                     + ", ".join(f"{n}: {dt:.2f} s" for n, _, dt in samples)
                     + f"; t(N) = {coef[0]:.3e} N^3 + {coef[1]:.3e} N^2 through the out-of-cache samples (N >= 4096; relative least squares, "
                     f"max residual {100 * resid:.1f} %) evaluated at N = {N_TRAIN}: {scaled:.0f} s; one cubic through ALL samples: "
                     f"{coef_all[0]:.3e} N^3 + {coef_all[1]:.3e} N^2, max residual {100 * resid_all:.1f} %, {scaled_all:.0f} s",
           "samples": [{"n": int(n), "gram_s": tg, "fit_s": dt} for n, tg, dt in samples],
           "cubic_fit": {"a_n3": float(coef[0]), "b_n2": float(coef[1]), "max_rel_residual": resid, "samples": "N >= 4096"},
           "cubic_fit_all_samples": {"a_n3": float(coef_all[0]), "b_n2": float(coef_all[1]), "max_rel_residual": resid_all,
                                     "fits_per_sec_at_16384": 1.0 / scaled_all},
           "range_fits_per_sec": sorted([1.0 / scaled, 1.0 / scaled_all]),
           "cpu_model": _cpu_model(), "nproc": os.cpu_count() or 1, "usable_cores": cores_host, "compiler_flags": "gcc " + _oracle_cflags()}
    # BASELINE.md section 2, B2 "albatross-faithful, pooled": the Gram over all host cores (callers.hpp:134-166), the
    # factor unchanged (Eigen's LDLT has no parallel path): only the Gram's share of the fit changes
    try:
        n, t_gram, dt = samples[-1]
        xg, _y = make_dataset(n, 44)
        t0 = time.perf_counter()
        _ = orc.gram(cov, xg, x_meas=True, y_meas=True, threads=cores_host)
        t_pool = time.perf_counter() - t0
        del _
        # Gram ~ N^2, factor ~ N^3: scale the two shares separately
        gram_16k = t_gram * (N_TRAIN / n) ** 2
        pooled_16k = scaled - gram_16k + t_pool * (N_TRAIN / n) ** 2
        out["pooled_gram"] = {"value": 1.0 / pooled_16k, "unit": "fits/sec", "cores": cores_host,
                              "sample": f"Gram at N={n}: serial {t_gram:.2f} s, pooled over {cores_host} threads {t_pool:.2f} s; "
                                        f"the fit's Gram share (x (16384/{n})^2) exchanged, the single-threaded factor unchanged"}
    except Exception as exc:  # noqa: BLE001 - context only
        out["pooled_gram"] = {"error": f"{type(exc).__name__}: {exc}"}
    # For context ("strong CPU"): the same fit by a competent multi-core CPU implementation - oracle/strong_llt.c
    # (own code: a threaded SE Gram of the lower triangle, a blocked pthread-parallel LL^T with an AVX2 / AVX-512
    # micro-kernel) + the oracle's substitutions.  NOT the reference's algorithm (albatross factors
    # with Eigen's unblocked single-threaded LDL^T), so it is reported beside `value`, with the GFLOP/s it achieved and the
    # cores it was allowed (rounds 1-5 used scipy's LAPACK here and measured 49 GFLOP/s: its BLAS pool does not scale
    # inside the box's container, whose cgroup grants 16 of the 256 visible hardware threads).
    try:
        usable, quota = _cpu_quota_cores()
        threads = int(max(1, min(usable, quota if quota else usable)))
        L = orc.lib()
        L.orc_llt_blocked_isa.restype = C.c_int
        L.orc_strong_gram_se.restype = None
        L.orc_strong_gram_se.argtypes = [C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_int64, C.c_int]

        def strong_fit(m):
            xs, ys = make_dataset(m, 44)
            t0 = time.perf_counter()
            K = np.empty((m, m), order="F")  # (lower triangle only: all the factorisation and the substitutions read)
            L.orc_strong_gram_se(C.c_void_p(xs.ctypes.data), m, DIM, C.c_double(1.0), C.c_double(1.0), C.c_double(0.1),
                                 C.c_void_p(K.ctypes.data), m, threads)
            t_gram = time.perf_counter() - t0
            t0 = time.perf_counter()
            info = L.orc_llt_blocked(C.c_void_p(K.ctypes.data), m, m, threads)
            t_chol = time.perf_counter() - t0
            assert info == 0, info
            t0 = time.perf_counter()
            a = orc.llt_solve(K, ys)
            t_solve = time.perf_counter() - t0
            r = sampled_residual(xs, ys, a)
            return t_gram, t_chol, t_solve, r

        strong_fit(2048)  # (page in the code, start the clocks)
        tg, tc, ts_, r8 = strong_fit(8192)
        gflops8 = 8192 ** 3 / 3. / tc / 1e9
        if (tg + tc + ts_) * 6.5 < 25.0:  # N = 16384 itself when it fits the budget (~8x the factor, 4x Gram and solve)
            n_s = N_TRAIN
            tg, tc, ts_, r = strong_fit(N_TRAIN)
            total = tg + tc + ts_
            how = f"at N={N_TRAIN} itself"
        else:
            n_s, r = 8192, r8
            total = tg * 4. + tc * 8. + ts_ * 4.
            how = "at N=8192, scaled x4 / x8 / x4"
        gflops = n_s ** 3 / 3. / tc / 1e9
        out["strong_cpu"] = {"value": 1.0 / total, "unit": "fits/sec", "cores": threads,
                             "factor_gflops": gflops, "factor_gflops_n8192": gflops8, "vector_bits": int(L.orc_llt_blocked_isa()),
                             "usable_hardware_threads": usable, "cgroup_cpu_quota": quota, "self_check_residual": r,
                             "sample": f"threaded SE Gram (lower triangle, oracle/strong_llt.c) {tg:.2f} s + blocked pthread LL^T (oracle/strong_llt.c) {tc:.2f} s = "
                                       f"{gflops:.0f} GFLOP/s + substitutions {ts_:.2f} s {how}; {threads} threads "
                                       f"({usable} hardware threads visible, cgroup CPU quota {quota})",
                             # a competent dpotrf on >= 32 unrestricted cores does >= 0.5 TFLOP/s: below that this row does not
                             # say what a strong CPU would do, only what this container's share of one does
                             "is_strong": bool(gflops >= 500. or threads < 32)}
    except Exception as exc:  # noqa: BLE001 - context only
        out["strong_cpu"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


def synthetic_stations(n, seed):
    """stations lat ~ U[25, 50] deg, lon ~ U[-125, -65] deg, h ~ U[0, 3000] m -> ECEF
    (WGS-84, km); temperature = 60 - 0.0065 h 1.8 + a smooth field + N(0, 1.75).  Returns (ecef km, elevation m,
    temperature); the real file of the example has only 2133 rows (examples/temperature_example/gsod.csv)."""
    rng = np.random.default_rng(seed)
    lat = np.deg2rad(rng.uniform(25., 50., n))
    lon = np.deg2rad(rng.uniform(-125., -65., n))
    h = rng.uniform(0., 3000., n)
    a, f = 6378137.0, 1. / 298.257223563
    e2 = f * (2. - f)
    nu = a / np.sqrt(1. - e2 * np.sin(lat) ** 2)
    ecef = np.stack([(nu + h) * np.cos(lat) * np.cos(lon), (nu + h) * np.cos(lat) * np.sin(lon),
                     (nu * (1. - e2) + h) * np.sin(lat)], axis=1) / 1000.
    smooth = 8. * np.sin(3. * lat) * np.cos(2. * lon) + 5. * np.cos(5. * lon)
    temp = 60. - 0.0065 * h * 1.8 + smooth + rng.normal(0., 1.75, n)
    return ecef, h, temp


def temperature_covariance(ab):
    """The tuned covariance of examples/temperature_example/temperature_example.cc:34-85; the elevation scaling
    1 + factor * max(0, center - h) (temperature_example_utils.h:78-84) is supplied as an explicit scale column."""
    class Elevation(ab.ScalingFunction):
        def _call_impl(self, c):
            raise AssertionError("scale columns are supplied explicitly")

    cov = ab.ScalingTerm(Elevation()) * ab.Constant(5.07288) + ab.IndependentNoise(1.75027) \
        + ab.Exponential(1.10298, 1.0, ab.AngularDistance()) * ab.SquaredExponential(5835.56, 13.913, ab.RadialDistance())
    scale = lambda h: 1. + 0.000153439 * np.maximum(0., 4446.5 - h)  # noqa: E731
    return cov, scale


def sampled_residual(x, y, information, ell=1.0, sigma=1.0, noise=0.1, rows=64):
    """Self-check carried by every bench line: max_r |(K a)_r - y_r| / max|y| over `rows` sampled rows of
    K = SE(ell, sigma)(x, x) + noise^2 I, evaluated with numpy from the features (independent of the library and
    of the oracle).  cond(K) ~ 1.6e6 for the bench problem, so a correct fp64 fit lands around 1e-11."""
    n = x.shape[0]
    idx = np.unique(np.linspace(0, n - 1, rows).astype(np.int64))
    d2 = ((x[idx, None, :] - x[None, :, :]) ** 2).sum(axis=2)
    k = sigma * sigma * np.exp(-d2 / (ell * ell))
    r = k @ information + noise * noise * information[idx] - y[idx]
    return float(np.abs(r).max() / np.abs(y).max())


def _device_features(capi, x_d, n):
    """agp_features over coordinates resident in HBM (x_d: albatross_amd.DeviceArray - agp_device_malloc + agp_memcpy of the
    C-ABI; this process holds ONE HIP runtime, the library's)"""
    f = capi.Features()
    f.n, f.dim, f.n_scale_columns = n, DIM, 0
    f.coords = x_d.ptr
    f.eq_id = None
    f.scales = None
    f.is_measurement = 0
    f.location = capi.DEVICE
    return f


def fit_batch_rates(ab, ctx, sizes=None, batches=(1, 8, 32, 256)):
    """Small / medium N, where the reference's own workloads live (benchmarks/bench_predict.cc:20-40: N = 512; the tuner loop):
    fits per second of agp_fit_create_batch - B independent fits of one shape in lock step, inputs resident in HBM - with the
    aggregate fraction of the fp64 MFMA peak.  B = 1 is agp_fit_create.  Config 2's covariance (Matern-5/2 + noise)."""
    from albatross_amd import _capi as capi
    lib = ctx._lib
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    kh = ctx.kernel(cov)
    rows = []
    for n in (sizes or (512, 1024, 2048, 4096)):
        for B in batches:
            if B > 1 and B * n * n * 8 > 6e9:  # (the slab of a batch: a few GB are plenty to fill the chip)
                continue
            xs_d, feats = [], []
            ys = np.empty((n, B), order="F")
            for b in range(B):
                x, y = make_dataset(n, 1000 + b)
                xs_d.append(ctx.to_device(x))
                ys[:, b] = y
                feats.append(_device_features(capi, xs_d[-1], n))
            y_d = ctx.to_device(ys.T.copy())  # (B, n) C-order = n x B column-major
            ctx.synchronize()
            kernels = (C.c_void_p * B)(*([kh] * B))
            fptrs = (C.c_void_p * B)(*[C.addressof(f) for f in feats])
            out = (C.c_void_p * B)()
            status = (C.c_int * B)()

            # (the ctypes arguments are built ONCE: at N = 512 a fit is 0.17 ms and building them per call - byref, c_void_p,
            # tensor.data_ptr() - was 10-15 us of the harness, not of the library)
            h = C.c_void_p()
            h_ref, f0_ref, y_ptr, ctx_h = C.byref(h), C.byref(feats[0]), C.c_void_p(y_d.ptr), ctx._h
            fit_create, fit_create_batch, fit_destroy = lib.agp_fit_create, lib.agp_fit_create_batch, lib.agp_fit_destroy

            def step():
                if B == 1:
                    st = fit_create(ctx_h, kh, f0_ref, y_ptr, None, h_ref, None, None)
                    assert st == capi.AGP_OK, st
                    fit_destroy(h)
                else:
                    st = fit_create_batch(ctx_h, B, kernels, fptrs, y_ptr, n, None, 0, out, None, 0, None, status)
                    assert st == capi.AGP_OK and all(s == capi.AGP_OK for s in status), (st, list(status))
                    for b in range(B):
                        fit_destroy(C.c_void_p(out[b]))
            # warm the clock: short kernels after an idle gap run at whatever the GPU had dropped to - at least 50 ms of the
            # same work back to back before anything is timed (two driver runs of this table used to differ by 2x at N = 512)
            t_w = time.perf_counter()
            while time.perf_counter() - t_w < 0.05:
                step()
            reps = max(5, min(60, int(0.25 / (2e-4 * B * (n / 512.) ** 2))))
            ts = []
            for _ in range(reps):
                t0 = time.perf_counter()
                step()
                ts.append(time.perf_counter() - t0)
            t = min(ts)
            flop = B * n ** 3 / 3.
            row = {"n": n, "batch": B, "ms_per_batch": 1e3 * t, "ms_median": 1e3 * sorted(ts)[len(ts) // 2], "reps": reps,
                   "fits_per_sec": B / t, "frac_of_mfma_peak": flop / t / 1e12 / MFMA_F64_PEAK_TFLOPS, "tflops": flop / t / 1e12}
            if B == 1:
                # the GPU's own span of one fit (HIP events of the library's stage timers: Gram + factor + back substitution),
                # next to the wall time of the call
                lib.agp_set_profiling(ctx._h, 1)
                step()
                step()
                span = 0.
                for stage in (0, 1, 2):
                    ms = C.c_double()
                    lib.agp_last_stage_ms(ctx._h, stage, C.byref(ms))
                    span += ms.value
                lib.agp_set_profiling(ctx._h, 0)
                row["gpu_span_ms"] = span
            rows.append(row)
            del xs_d, y_d
    return rows


def other_configs(ab, ctx):
    """BASELINE.json configs 2, 4 and 5 on this GPU, each with its algorithmic work and fraction of peak - measured after
    the headline, outside `value`.  Host-resident inputs through the Python mirror (the uploads are << the fits)."""
    out = {}

    def best(f, reps):
        f()
        t = 1e9
        for _ in range(reps):
            t0 = time.perf_counter()
            f()
            t = min(t, time.perf_counter() - t0)
        return t

    # ---- config 2: 3-D Matern-5/2 + noise, N = 4096 fp64 dense fit + predict (M = 4096) ----
    try:
        n = m = 4096
        x, y = make_dataset(n, 42)
        xs, _ = make_dataset(m, 43)
        model = ab.gp_from_covariance(ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1), context=ctx)
        ds = ab.RegressionDataset(x, y)
        t_fit = best(lambda: model.fit(ds), 40)  # (2 ms fits: the clock needs a few of them back to back; best of 40)
        # the same fit as the headline times it: features and targets resident in HBM, straight through the C-ABI
        from albatross_amd import _capi as capi
        x_d, y_d = ctx.to_device(x), ctx.to_device(y)
        feats = _device_features(capi, x_d, n)
        kh = ctx.kernel(model.covariance_function_)
        ctx.synchronize()

        def fit_resident():
            h = C.c_void_p()
            st = ctx._lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None)
            assert st == capi.AGP_OK, st
            ctx._lib.agp_fit_destroy(h)
        t_fit_dev = best(fit_resident, 150)  # (0.3 s back to back: the clock ramps over bursts of 2 ms kernels)
        fm = model.fit(ds)
        p = fm.predict(xs)
        # predictions as the headline's `predict` block times them: test features and outputs resident in HBM, C-ABI
        hfit = C.c_void_p()
        st = ctx._lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(hfit), None, None)
        assert st == capi.AGP_OK, st
        xs_d = ctx.to_device(xs)
        fxs = _device_features(capi, xs_d, m)
        mean_d, var_d, cov_d = ctx.device_empty(m), ctx.device_empty(m), ctx.device_empty((m, m))
        ctx.synchronize()
        lib = ctx._lib
        t_mean = best(lambda: lib.agp_predict_mean(ctx._h, kh, hfit, C.byref(fxs), C.c_void_p(mean_d.ptr), capi.DEVICE), 20)
        t_marg = best(lambda: lib.agp_predict_marginal(ctx._h, kh, hfit, C.byref(fxs), C.c_void_p(mean_d.ptr),
                                                       C.c_void_p(var_d.ptr), capi.DEVICE), 10)
        t_joint = best(lambda: lib.agp_predict_joint(ctx._h, kh, hfit, C.byref(fxs), C.c_void_p(mean_d.ptr),
                                                     C.c_void_p(cov_d.ptr), capi.DEVICE), 5)
        # (and the device results are the Python mirror's, which the parity tests hold against the oracle)
        pj = p.joint()
        assert np.abs(cov_d.numpy() - pj.covariance).max() <= 1e-12 * np.abs(pj.covariance).max()
        assert np.abs(mean_d.numpy() - pj.mean).max() <= 1e-12 * max(1., np.abs(pj.mean).max())
        t_joint_host = best(p.joint, 2)
        lib.agp_fit_destroy(hfit)
        cov_d.free()
        del cov_d, pj
        fit_flop, marg_flop, joint_flop = n ** 3 / 3., float(n) * n * m, float(n) * n * m + float(n) * m * m
        out["config2"] = {
            "workload": "3-D Matern-5/2(2,1)+IndependentNoise(0.1), N=4096 fp64, features mt19937(42), predict at M=4096 mt19937(43)",
            "fit_ms": 1e3 * t_fit_dev, "fit_flop": fit_flop, "fit_frac_of_mfma_peak": fit_flop / t_fit_dev / 1e12 / MFMA_F64_PEAK_TFLOPS,
            "fit_ms_host_inputs_python": 1e3 * t_fit,
            "predict_mean_ms": 1e3 * t_mean, "predict_mean_pts_per_sec": m / t_mean,
            "predict_marginal_ms": 1e3 * t_marg, "predict_marginal_flop": marg_flop,
            "predict_marginal_frac_of_mfma_peak": marg_flop / t_marg / 1e12 / MFMA_F64_PEAK_TFLOPS,
            "predict_joint_ms": 1e3 * t_joint, "predict_joint_flop": joint_flop,
            "predict_joint_frac_of_mfma_peak": joint_flop / t_joint / 1e12 / MFMA_F64_PEAK_TFLOPS,
            "predict_joint_ms_with_download": 1e3 * t_joint_host,
            "note": "fit and predictions through the C-ABI with features, targets and outputs resident in HBM (as the headline); "
                    "predict_joint_ms_with_download: the Python mirror, incl. the 134 MB download of the M x M covariance"}
        del fm, p
    except Exception as exc:  # noqa: BLE001
        out["config2"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- small / medium N, batched: B independent fits of one shape in lock step (agp_fit_create_batch) ----
    try:
        out["small_n_batched"] = {"workload": "B independent fits (3-D Matern-5/2 + noise, inputs in HBM) per call; B = 1: agp_fit_create",
                                  "rows": fit_batch_rates(ab, ctx)}
    except Exception as exc:  # noqa: BLE001
        out["small_n_batched"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- the reference's OWN benchmark (benchmarks/bench_predict.cc:20-85, bench_utils.h:25-85): N = 512 1-D training
    #      points x ~ U[0, 10] from mt19937(seed), y = sin x + 0.1 cos 10 x, bench_covariance = SE(1, 1) + IndependentNoise(0.1);
    #      BM_gp_fit (seed 31), BM_gp_predict_joint (32 / 33), _marginal (34 / 35), _mean (36 / 37) at 512 test points.
    #      Through the drop-in surface with HOST inputs, as the reference's benchmark calls it (model.fit(dataset),
    #      fit_model.predict(features).joint()); B = 32: thirty-two such fits per call (ab.fit_batch).  `cpu_port_ms`: the
    #      oracle (the reference's algorithm restated, one thread) on the same inputs, best of three.
    try:
        from oracle import oracle_py as orc
        nt = 512
        covb = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)

        def bench_ds(seed):
            xf = mt19937_uniform(seed, nt)
            return xf, np.sin(xf) + 0.1 * np.cos(10. * xf)
        model = ab.gp_from_covariance(covb, context=ctx)
        x31, y31 = bench_ds(31)
        ds31 = ab.RegressionDataset(x31, y31)
        t_fit = best(lambda: model.fit(ds31), 300)
        dsets = [ab.RegressionDataset(*bench_ds(1000 + b)) for b in range(32)]
        t_fit32 = best(lambda: ab.fit_batch([model] * 32, dsets), 30)
        rb = {"workload": "benchmarks/bench_predict.cc: N = 512 1-D, bench_covariance SE(1,1)+IndependentNoise(0.1), 512 test points; "
                          "host inputs through the Python mirror of the reference's call surface",
              "fit_ms": 1e3 * t_fit, "fit_batch32_ms": 1e3 * t_fit32, "fits_per_sec_single": 1. / t_fit, "fits_per_sec_batch32": 32. / t_fit32}
        tcpu = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            orc.OracleFit(covb, x31, y31).information
            tcpu = min(tcpu, time.perf_counter() - t0)
        rb["fit_cpu_port_ms"] = 1e3 * tcpu
        for name, s_tr, s_te in (("joint", 32, 33), ("marginal", 34, 35), ("mean", 36, 37)):
            xtr, ytr = bench_ds(s_tr)
            xte = mt19937_uniform(s_te, nt)
            fm = model.fit(ab.RegressionDataset(xtr, ytr))
            pred = {"joint": lambda: fm.predict(xte).joint(), "marginal": lambda: fm.predict(xte).marginal(),
                    "mean": lambda: fm.predict(xte).mean()}[name]
            rb[f"predict_{name}_ms"] = 1e3 * best(pred, 100)
            ofit = orc.OracleFit(covb, xtr, ytr)
            ocall = {"joint": lambda: ofit.predict_joint(xte), "marginal": lambda: ofit.predict_marginal(xte),
                     "mean": lambda: ofit.predict_mean(xte)}[name]
            tcpu = 1e9
            for _ in range(3):
                t0 = time.perf_counter()
                ocall()
                tcpu = min(tcpu, time.perf_counter() - t0)
            rb[f"predict_{name}_cpu_port_ms"] = 1e3 * tcpu
            del fm
        out["reference_bench"] = rb
    except Exception as exc:  # noqa: BLE001
        out["reference_bench"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- config 4: temperature-example kernel, N = 32768, fp64 vs mixed precision ----
    try:
        n = 32768
        ecef, h, temp = synthetic_stations(n, 11)
        cov, scale = temperature_covariance(ab)
        ds = ab.RegressionDataset(ab.FeatureSet(ecef, [scale(h)]), temp - temp.mean())
        res = {}
        for prec in ("fp64", "mixed"):
            model = ab.gp_from_covariance(cov, context=ctx)
            model.precision = prec
            t = 1e9
            for _ in range(3):  # (the first fit of a size pays for two 8.6 GB allocations)
                t0 = time.perf_counter()
                fm = model.fit(ds)
                t = min(t, time.perf_counter() - t0)
                fit = fm.get_fit()
                fit.accept_mixed_log_determinant = True  # (reported as log_det_rel_err_vs_fp64)
                info, ld = fit.information.copy(), fit.log_determinant
                del fm, fit
            refinement = model.refinement_
            ctx.set_profiling(True)  # (one more fit for the stage split: the event records are kept out of the timed fits)
            try:
                fm = model.fit(ds)
                stages_ms = [ctx.stage_ms(i) for i in range(3)]
                del fm
            finally:
                ctx.set_profiling(False)
            res[prec] = (t, info, ld, refinement, stages_ms)
        flop = n ** 3 / 3.
        t64, a64, ld64, _, _ = res["fp64"]
        tmx, amx, ldm, (its, rel_res), stages = res["mixed"]
        out["config4"] = {
            "workload": "temperature-example covariance (ScalingTerm*Constant + IndependentNoise + Exponential<Angular>*SE<Radial>, "
                        "tuned values) on N=32768 synthetic stations",
            "fit_flop": flop, "fp64_fit_ms": 1e3 * t64, "fp64_frac_of_mfma_f64_peak": flop / t64 / 1e12 / MFMA_F64_PEAK_TFLOPS,
            "mixed_fit_ms": 1e3 * tmx, "mixed_speedup": t64 / tmx, "mixed_fp64_equivalent_tflops": flop / tmx / 1e12,
            "mixed_stages_ms": {"gram": stages[0], "factor": stages[1], "refinement_and_solve": stages[2]},
            "cg_steps": int(its), "cg_relative_residual": float(rel_res),
            "information_rel_err_vs_fp64": float(np.abs(amx - a64).max() / np.abs(a64).max()),
            "log_det_rel_err_vs_fp64": float(abs(ldm - ld64) / abs(ld64)),
            "log_det_abs_err_over_n": float(abs(ldm - ld64) / n),
            "mixed_products": ("fp32 MFMA (AGP_MIXED_BF16=0)" if os.environ.get("AGP_MIXED_BF16", "1") == "0" else
                               "bf16 x 3, six products per block (AGP_MIXED_F16=0)" if os.environ.get("AGP_MIXED_F16", "1") == "0" else
                               "fp16 x 2 planes of power-of-two-scaled rows, four exact products per block, fp32 accumulation "
                               "inside a launch (csrc/gemm_f16x2.hip)")}
    except Exception as exc:  # noqa: BLE001
        out["config4"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- config 5: sparse GP (PITC), N = 262144, 2048 inducing points, independent groups of 512 ----
    try:
        n, m, gs = 262144, 2048, 512
        rng = np.random.default_rng(n)
        x = np.sort(rng.uniform(0., n / 16., n))
        y = np.sin(x) + 0.1 * np.cos(10. * x) + 0.1 * rng.standard_normal(n)
        cov = ab.SquaredExponential(1.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))  # benchmarks/bench_utils.h:61-65
        u = np.linspace(x.min(), x.max(), m)

        def grouper(f):
            return np.searchsorted(x, np.asarray(f, dtype=np.float64).reshape(-1)) // gs
        grouper.vectorized = True
        model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "pitc", context=ctx)
        model.set_param("inducing_nugget", 1e-6)
        ds = ab.RegressionDataset(x, y)
        fm = model.fit(ds)
        t_fit = best(lambda: model.fit(ds), 2)
        xs = np.linspace(x.min(), x.max(), 4096)
        t_pred = best(lambda: fm.predict(xs).marginal(), 2)
        flop = 3. * m * m * n + float(n) * gs * m + n * float(gs) ** 2 / 3.
        out["config5"] = {
            "workload": "sparse GP (PITC), N=262144 1-D, 2048 inducing points, groups of 512, bench covariance",
            "fit_ms": 1e3 * t_fit, "fit_flop": flop, "fit_frac_of_mfma_peak": flop / t_fit / 1e12 / MFMA_F64_PEAK_TFLOPS,
            "predict_marginal_m4096_ms": 1e3 * t_pred, "nll": float(fm.get_fit().nll),
            "flop_formula": "3 m^2 n (P, W W^T, Q1) + n s m (A, W) + n s^2 / 3 (block LL^T)"}
        del fm
    except Exception as exc:  # noqa: BLE001
        out["config5"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- config 3's problem at larger N (one GPU, features and targets resident in HBM): the sizes at which the sharded fit
    #      of `--gpus N` has work to share (its N = 65536 block), and how far the factorisation gets from the tile's fixed costs ----
    try:
        from albatross_amd import _capi as capi
        rows = []
        cov3 = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
        kh3 = ctx.kernel(cov3)
        for n in (32768, 65536):
            x, y = make_dataset(n, 44)
            x_d, y_d = ctx.to_device(x), ctx.to_device(y)
            feats = _device_features(capi, x_d, n)
            ctx.synchronize()

            def fit_large():
                h = C.c_void_p()
                st = ctx._lib.agp_fit_create(ctx._h, kh3, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None)
                assert st == capi.AGP_OK, st
                ctx._lib.agp_fit_destroy(h)
            t = best(fit_large, 2)  # (one untimed fit pays for the allocation, best of two timed ones)
            flop = n ** 3 / 3.
            rows.append({"n": n, "fit_ms": 1e3 * t, "fits_per_sec": 1. / t, "tflops_overall": flop / t / 1e12,
                         "frac_of_mfma_peak": flop / t / 1e12 / MFMA_F64_PEAK_TFLOPS})
            del x_d, y_d
        out["config3_large_n"] = {"workload": "config 3's problem (3-D SE(1,1) + noise(0.1), mt19937(44)) at N = 32768 / 65536, fp64 dense fit, "
                                              "inputs resident in HBM, one GPU; n^3/3 flop over the whole fit", "rows": rows}
    except Exception as exc:  # noqa: BLE001
        out["config3_large_n"] = {"error": f"{type(exc).__name__}: {exc}"}

    # ---- independent fits in flight: T host threads, each with a context of its own, issue fits of the headline problem
    # (and of config 2's size) concurrently on this GPU - the chain-bound tail and the substitution of one fit run beside the
    # bulk phase of another.  The headline `value` stays ONE fit at a time (a fit's latency is its reciprocal); this is the
    # throughput a caller with several independent datasets or parameter vectors gets (the tuner's P + 1 objective evaluations,
    # tune/finite_difference.hpp:20-94, at sizes beyond agp_nll_batch's lock-step launches).  Thread i starts i / T of a fit
    # late (started together the fits run in lock step and gain nothing).  Which hardware queue the runtime gives a context's
    # streams differs from context to context and the result with it (34.5 or 37.5 fits/s at T = 2, profiles/r06/
    # time_fits_in_flight.txt): three attempts with fresh contexts, all reported, the best one quoted.
    try:
        import threading
        cov3 = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
        rows = []

        def attempt(n, x, y, threads, fits):
            ctxs = [ab.Context(ctx.device_id) for _ in range(threads)]
            try:
                state = []
                for c in ctxs:
                    x_d, y_d = c.to_device(x), c.to_device(y)
                    state.append((c, c.kernel(cov3), _device_features(capi, x_d, n), y_d, x_d))
                gate = threading.Barrier(threads)
                took, failed = [0.] * threads, []

                def run(i):
                    c, kh_i, feats_i, y_i, _ = state[i]

                    def one():
                        h = C.c_void_p()
                        st = c._lib.agp_fit_create(c._h, kh_i, C.byref(feats_i), C.c_void_p(y_i.ptr), None, C.byref(h), None, None)
                        if st != capi.AGP_OK:
                            failed.append(st)
                        c._lib.agp_fit_destroy(h)
                    for _ in range(2):
                        one()
                    t0 = time.perf_counter()
                    one()
                    single = time.perf_counter() - t0
                    gate.wait()
                    t0 = time.perf_counter()
                    time.sleep(i * single / threads)
                    for _ in range(fits):
                        one()
                    took[i] = time.perf_counter() - t0
                th = [threading.Thread(target=run, args=(i,)) for i in range(threads)]
                for t_ in th:
                    t_.start()
                for t_ in th:
                    t_.join()
                if failed:
                    raise RuntimeError(f"agp_fit_create returned {failed[0]}")
                del state
                return threads * fits / max(took)
            finally:
                for c in ctxs:
                    c.close()

        for n, fits in ((N_TRAIN, 10), (4096, 40)):
            x, y = make_dataset(n, 44)
            for threads in (1, 2):
                rates = [attempt(n, x, y, threads, fits) for _ in range(1 if threads == 1 else 3)]
                rows.append({"n": n, "threads": threads, "fits_per_sec": max(rates), "attempts_fits_per_sec": rates})
        out["fits_in_flight"] = {"workload": "independent fp64 fits of config 3's problem (and of its N = 4096 sub-problem), inputs "
                                             "resident in HBM, one context and one host thread per fit in flight, one GPU; "
                                             "fits_per_sec = the best attempt", "rows": rows}
    except Exception as exc:  # noqa: BLE001
        out["fits_in_flight"] = {"error": f"{type(exc).__name__}: {exc}"}
    return out


FALLBACK_EXIT = 5  # a replicas line was printed in place of the sharded measurement that was asked for


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n", "--train-points", dest="n", type=int, default=N_TRAIN,
                    help="training points (default 16384 = BASELINE config 3; 32768 / 65536: sizes where sharding one fit pays)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-predict", action="store_true")
    ap.add_argument("--no-configs", action="store_true", help="skip the `configs` block (BASELINE configs 2, 4, 5)")
    ap.add_argument("--force-sharded", action="store_true", help="N = 1: time the sharded entry point (one rank, no transport)")
    ap.add_argument("--multi-gpu", choices=["sharded", "replicas"], default="sharded",
                    help="N > 1: 'sharded' (default) = `value` is ONE fit row-block-sharded over all ranks (RCCL broadcast + "
                         "all-gather per block column, strong scaling); 'replicas' = one independent fit per rank, no "
                         "data-path collective (weak scaling).  The other mode is measured too and reported in an auxiliary block.")
    ap.add_argument("--no-fallback", action="store_true",
                    help="N > 1: fail instead of falling back to replicas when the sharded fit cannot run or fails its self-check")
    ap.add_argument("--allow-fallback", action="store_true",
                    help="N > 1: exit 0 when the replicas fallback was measured instead of the sharded fit (default: the labelled "
                         f"line is still printed, but the exit code is {FALLBACK_EXIT} - a dead RCCL path is not a green run)")
    ap.add_argument("--fallback-note", default="", help=argparse.SUPPRESS)
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` without a launcher: start the N ranks as CHILD processes (torch.distributed.run) before
    anything here has touched the GPU, relay rank 0's JSON line and the children's exit code.  No re-exec."""
    import signal
    import socket
    import subprocess
    import torch  # device_count() does not initialise the GPU
    visible = torch.cuda.device_count()
    if visible < args.gpus and os.environ.get("BENCH_SINGLE_DEVICE") != "1":
        sys.stderr.write(f"bench.py: --gpus {args.gpus} but {visible} device(s) visible (BENCH_SINGLE_DEVICE=1 runs every rank on "
                         "GPU 0 with gloo-staged collectives: a test mode, not a measurement)\n")
        return 2
    limit = float(os.environ.get("BENCH_LAUNCH_TIMEOUT_S", "1500"))

    def attempt(extra):
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv) + extra
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.setdefault("OMP_NUM_THREADS", "1")
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True)
        try:
            stdout, _ = p.communicate(timeout=limit)
        except subprocess.TimeoutExpired:
            try:
                os.killpg(p.pid, signal.SIGKILL)  # the process group this call started, nothing else
            except ProcessLookupError:
                pass
            stdout, _ = p.communicate()
            sys.stderr.write(f"bench.py: the {args.gpus}-rank run did not finish within {limit:.0f} s; killed\n")
            return 124, None
        line = None
        for ln in (stdout or "").splitlines():
            if ln.startswith('{"metric"'):
                line = ln
        return p.returncode, line

    # the children's arguments are rebuilt from the parsed values: torch.distributed.run's own parser would take a bare
    # `--n` for an abbreviation of `--nnodes`
    argv = ["--gpus", str(args.gpus), "--steps", str(args.steps), "--warmup", str(args.warmup), "--train-points", str(args.n),
            "--multi-gpu", args.multi_gpu]
    for flag, on in (("--no-cpu-baseline", args.no_cpu_baseline), ("--no-predict", args.no_predict), ("--no-configs", args.no_configs),
                     ("--force-sharded", args.force_sharded), ("--no-fallback", args.no_fallback),
                     ("--allow-fallback", args.allow_fallback)):
        if on:
            argv.append(flag)
    rc, line = attempt([])
    if rc == 0 and line:
        print(line, flush=True)
        return 0
    if rc != 0 and line and '"sharded_fallback"' in line:
        # the ranks agreed on the fallback themselves and left with FALLBACK_EXIT (torch.distributed.run turns that into 1):
        # the labelled line, a non-zero code
        print(line, flush=True)
        return FALLBACK_EXIT
    if args.multi_gpu == "sharded" and not args.no_fallback:
        note = f"the sharded {args.gpus}-rank run exited with code {rc}" + ("" if line else " and printed no result line")
        sys.stderr.write(f"bench.py: {note}; measuring {args.gpus} independent fits (replicas) instead\n")
        argv[argv.index("--multi-gpu") + 1] = "replicas"
        rc2, line2 = attempt(["--fallback-note", note])
        if line2 and (rc2 == 0 or '"sharded_fallback"' in line2):
            print(line2, flush=True)
            return 0 if args.allow_fallback else FALLBACK_EXIT
        return rc2 or 1
    if line:
        print(line, flush=True)
    return rc or 1


def run_rank(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    os.environ.setdefault("AGP_COMM_TIMEOUT_S", "60")  # a dead peer / deadlocked collective becomes an error within a minute

    import datetime

    import albatross_amd as ab
    from albatross_amd import _capi as capi

    # ONE HIP runtime in this process: the library's.  Device buffers come from the C-ABI (agp_device_malloc / agp_memcpy),
    # "torch.cuda.synchronize()" of the contract is agp_context_synchronize (hipDeviceSynchronize on the rank's device).
    # Rounds 1-5 let torch allocate the inputs: two runtimes (torch's ROCm 7.0 + the library's 7.2) in one process cost a
    # one-off 35-60 ms stall in some later synchronisation, which landed inside the driver's timed region in round 5.
    if capi.load().agp_device_count() <= 0:
        raise SystemExit("bench.py needs a GPU (no CPU fallback)")
    # BENCH_SINGLE_DEVICE=1 (testing the N > 1 code path on a one-GPU box): every rank uses GPU 0 and the collectives of
    # the sharded fit go over gloo through the library's callback transport - RCCL refuses two ranks per device.
    single_device = os.environ.get("BENCH_SINGLE_DEVICE") == "1"
    if single_device:
        local_rank = 0
    torch = dist = None
    if world > 1:
        # torch.distributed is the CONTROL plane only (rendezvous, exchange of the 128-byte RCCL id, agreeing on a
        # fallback): gloo on CPU tensors, finite timeout - torch never touches the GPU here.  The data path - and the
        # barrier / max-over-ranks of the timing - run on the library's own RCCL communicator.
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="gloo", timeout=datetime.timedelta(seconds=300))
    from albatross_amd.distributed import Communicator, ShardedGaussianProcessFit

    def all_agree(ok):
        """True iff `ok` on every rank (gloo)"""
        if world == 1:
            return bool(ok)
        t = torch.tensor([1.0 if ok else 0.0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        return t.item() > 0

    n = args.n
    ctx = ab.Context(local_rank)
    lib = ctx._lib
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    kh = ctx.kernel(cov)
    comm = None
    transport = "none"
    fallback_note = args.fallback_note
    want_sharded = (world > 1 and args.multi_gpu == "sharded") or (world == 1 and args.force_sharded)
    if world > 1 and want_sharded:
        if single_device:
            transport = "callbacks (BENCH_SINGLE_DEVICE test mode: collectives staged through host memory over gloo - not RCCL)"
            comm = Communicator.from_torch(ctx, transport="callbacks")
        else:
            err = ""
            try:
                comm = Communicator.from_torch(ctx, transport="rccl")
                transport = "rccl"
            except Exception as exc:  # noqa: BLE001 - every rank must take the same exit
                err = f"{type(exc).__name__}: {exc}"
            if not all_agree(comm is not None):
                if comm is not None:
                    comm._h = None  # peers may never have joined: do not run the communicator's destructor
                comm = None
                fallback_note = f"no RCCL communicator ({err or 'creation failed on another rank'})"
        if comm is not None:
            assert comm.world == world and comm.rank == rank

    def gloo_barrier():
        if world > 1:
            dist.barrier()

    def make_feats(ptr, count):
        f = capi.Features()
        f.n, f.dim, f.n_scale_columns = count, DIM, 0
        f.coords = ptr
        f.eq_id = None
        f.scales = None
        f.is_measurement = 0
        f.location = capi.DEVICE
        return f

    # inputs resident in HBM before the timed region.  Sharded: every rank holds the SAME dataset (one fit over all
    # ranks); replicas: one dataset per rank.
    def load(seed):
        xh, yh = make_dataset(n, seed)
        return xh, yh, ctx.to_device(xh), ctx.to_device(yh)

    sharded = want_sharded and (world == 1 or comm is not None)
    x_h, y_h, x_d, y_d = load(44 if sharded or world == 1 else 44 + rank)
    ctx.synchronize()
    feats = make_feats(x_d.ptr, n)
    sfit = ShardedGaussianProcessFit(ctx, cov, comm) if sharded else None

    def replica_step(want_information=False):
        h = C.c_void_p()
        info = np.empty(n) if want_information else None
        st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h),
                                None if info is None else C.c_void_p(info.ctypes.data), None)
        if st != capi.AGP_OK:
            raise RuntimeError(f"agp_fit_create failed: {lib.agp_status_string(st).decode()} "
                               f"{lib.agp_last_error(ctx._h).decode()}")
        lib.agp_fit_destroy(h)
        return info

    def sharded_step(want_information=False):
        res = sfit.fit(None, None, features_struct=feats, device_targets=y_d.ptr)
        return res.information if want_information else None

    # ---- self-check (and, for N > 1, the decision whether the sharded path is usable at all) ----
    self_check = None
    if sharded:
        ok, why = True, ""
        try:
            resid = sampled_residual(x_h, y_h, sharded_step(True))
            if os.environ.get("BENCH_TEST_SHARDED_FAILURE") == str(rank):  # test hook: this rank's self-check "fails"
                resid = 1.0
            self_check = {"rows": 64, "max_rel_residual": resid, "ok": bool(resid < 1e-8),
                          "what": "max_r |(K a)_r - y_r| / max|y| of a sharded fit before the timed region, numpy from the features"}
            if not self_check["ok"]:
                ok, why = False, f"sharded fit failed its self-check (residual {resid:.2e})"
        except Exception as exc:  # noqa: BLE001
            ok, why = False, f"{type(exc).__name__}: {exc}"
        if world > 1 and not all_agree(ok):
            sys.stderr.write(f"bench.py rank {rank}: sharded fit unusable ({why or 'failed on another rank'})\n")
            if args.no_fallback:
                sys.stderr.flush()
                os._exit(3)
            fallback_note = f"the sharded fit over {transport} failed on at least one rank" + (f" (rank {rank}: {why})" if why else "")
            comm._h = None  # broken or possibly mid-collective on a peer: never destroyed, the process ends with os._exit
            comm, sfit, sharded, transport = None, None, False, "none"
            x_h, y_h, x_d, y_d = load(44 + rank)
            feats = make_feats(x_d.ptr, n)
            ctx.synchronize()
        elif world == 1 and not ok:
            raise SystemExit(f"bench.py: {why}")
    fell_back = world > 1 and want_sharded and not sharded

    step = sharded_step if sharded else replica_step

    def barrier():
        tb0 = time.perf_counter()
        ctx.synchronize()
        tb1 = time.perf_counter()
        if comm is not None:
            comm.barrier()
        else:
            gloo_barrier()
        tb2 = time.perf_counter()
        ctx.synchronize()
        if os.environ.get("BENCH_DEBUG_STEPS"):
            sys.stderr.write(f"bench.py rank {rank}: barrier: sync {1e3 * (tb1 - tb0):.3f} ms, ranks {1e3 * (tb2 - tb1):.3f} ms, "
                             f"sync {1e3 * (time.perf_counter() - tb2):.3f} ms\n")

    def max_over_ranks(v):
        if comm is not None:
            return float(comm.all_reduce([v], "max")[0])
        if world > 1:
            t = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            return float(t.item())
        return v

    try:
        ctx.set_profiling(True)
        for _ in range(args.warmup):
            step()
        gemm_ms = gemm_flop = gemm_launches = 0.0
        gram_ms = factor_ms = solve_ms = 0.0
        barrier()
        t0 = time.perf_counter()
        step_walls = []
        for _ in range(args.steps):
            ts0 = time.perf_counter()
            step()  # returns after its streams have drained
            step_walls.append(time.perf_counter() - ts0)
            if sharded:
                gram_ms += sfit.stage(0)
                factor_ms += sfit.stage(1)
                gemm_ms += sfit.stage(3)
                gemm_launches += sfit.stage(4)
                gemm_flop += sfit.stage(5)
            else:
                gram_ms += ctx.stage_ms(0)
                factor_ms += ctx.stage_ms(1)
                solve_ms += ctx.stage_ms(2)
                gemm_ms += ctx.stage_ms(3)
                gemm_launches += ctx.stage_ms(4)
                gemm_flop += ctx.stage_ms(5)
        t_loop = time.perf_counter() - t0
        barrier()
        elapsed = max_over_ranks(time.perf_counter() - t0)
        # (the stage events are for the headline's live roofline only: left on, every later measurement of this process - the
        # first row of the small-N table - paid five event records per fit, 20 us at N = 512)
        ctx.set_profiling(False)
        if os.environ.get("BENCH_DEBUG_STEPS"):
            sys.stderr.write(f"bench.py rank {rank}: per-step wall ms {[round(1e3 * v, 2) for v in step_walls]}, loop {1e3 * t_loop:.2f} ms, "
                             f"with closing barrier {1e3 * elapsed:.2f} ms\n")

        if not sharded:  # self-check of the replica path: one more fit of the timed problem, outside the timed region
            resid = sampled_residual(x_h, y_h, replica_step(True))
            self_check = {"rows": 64, "max_rel_residual": resid, "ok": bool(resid < 1e-8),
                          "what": "max_r |(K a)_r - y_r| / max|y| of one more fit of the timed problem, numpy from the features"}

        # ---- auxiliary (N > 1, sharded): N independent fits, one per GPU ("replicas") ----
        aux = None
        if world > 1 and sharded:
            replica_step()
            barrier()
            tr = time.perf_counter()
            for _ in range(3):
                replica_step()
            barrier()
            tr = max_over_ranks(time.perf_counter() - tr)
            aux = {"replicas": {"fits_per_sec": 3 * world / tr, "scaling": "weak",
                                "note": "one independent fit per GPU (every rank a copy of the problem), no collective"}}
    except Exception as exc:  # noqa: BLE001
        # a failed or timed-out collective leaves the other ranks inside theirs: report, and leave with a non-zero code at
        # once (no destructors, no re-exec: the launcher starts fresh children)
        sys.stderr.write(f"bench.py rank {rank}: {type(exc).__name__}: {exc}\n")
        sys.stderr.flush()
        os._exit(3)

    # ---- secondary: predict points/sec at M = 4096 (per GPU) against one resident fit ----
    # N = 1: the fit of the timed problem.  N > 1: the test points are the partitioned unit - every rank holds a resident fit
    # (its own) and predicts its own 4096 points, no collective; `*_pts_per_sec` is the whole-job rate (N x 4096 points /
    # max-over-ranks time).  Sharded runs add the distributed marginal prediction straight from the sharded factor.
    predict = None
    if not args.no_predict and (world > 1 or not sharded):
        try:
            m = 4096
            xs_h, _ = make_dataset(m, 43 + 1000 * rank)
            xs_d = ctx.to_device(xs_h)
            out_d = ctx.device_empty(2 * m)
            fx = make_feats(xs_d.ptr, m)
            h = C.c_void_p()
            st = lib.agp_fit_create(ctx._h, kh, C.byref(feats), C.c_void_p(y_d.ptr), None, C.byref(h), None, None)
            if st != capi.AGP_OK:
                raise RuntimeError(f"agp_fit_create failed: {lib.agp_status_string(st).decode()}")
            mean_p, var_p = C.c_void_p(out_d.ptr), C.c_void_p(out_d.ptr + 8 * m)

            def timed(fn, reps):
                fn()
                barrier()
                t = time.perf_counter()
                for _ in range(reps):
                    fn()
                barrier()
                return max_over_ranks(time.perf_counter() - t) / reps

            t_mean = timed(lambda: lib.agp_predict_mean(ctx._h, kh, h, C.byref(fx), mean_p, capi.DEVICE), 5)
            t_marg = timed(lambda: lib.agp_predict_marginal(ctx._h, kh, h, C.byref(fx), mean_p, var_p, capi.DEVICE), 3)
            lib.agp_fit_destroy(h)
            predict = {"m_per_gpu": m, "m": m * world, "mean_pts_per_sec": world * m / t_mean,
                       "marginal_pts_per_sec": world * m / t_marg, "mean_ms": 1e3 * t_mean, "marginal_ms": 1e3 * t_marg}
            if world > 1:
                predict["scaling"] = "weak"
                predict["note"] = "test points partitioned over the ranks, each against a resident fit on its GPU; no collective"
            if world > 1 and sharded:
                xq, _ = make_dataset(m, 43)  # collective call: the same test points on every rank
                t_sh = timed(lambda: sfit.predict_marginal(xq), 2)
                predict["sharded_factor"] = {"m": m, "marginal_pts_per_sec": m / t_sh, "marginal_ms": 1e3 * t_sh,
                                             "note": "agp_sharded_predict_marginal: the distributed forward substitution "
                                                     "against the sharded factor (not replicated), host features in and out"}
        except Exception as exc:  # noqa: BLE001 - see above: peers may be inside a collective
            sys.stderr.write(f"bench.py rank {rank}: {type(exc).__name__}: {exc}\n")
            sys.stderr.flush()
            os._exit(3)

    # ---- auxiliary (N > 1, sharded), LAST thing that touches the communicator: the same sharded fit at N = 65536, where the
    # 8-rank schedule is COMPUTE-bound (one rank's share 227-250 ms against 1.67 s on one GPU; at N = 16384 it is bound by
    # the owner chain) - the first real scaling curve should show both regimes.  Outside `value`.  A failure here must not
    # cost the headline line: it is recorded, the line goes out, and the ranks leave without running destructors.
    aux_big, aux_broken = None, False
    if world > 1 and sharded and not single_device and not args.no_configs and n == N_TRAIN:
        try:
            nb = 65536
            xb, yb = make_dataset(nb, 45)
            xb_d, yb_d = ctx.to_device(xb), ctx.to_device(yb)
            fb = make_feats(xb_d.ptr, nb)
            ctx.synchronize()
            big = ShardedGaussianProcessFit(ctx, cov, comm)
            res = big.fit(None, None, features_struct=fb, device_targets=yb_d.ptr)  # warm-up + self-check
            resid_b = sampled_residual(xb, yb, res.information)
            barrier()
            tb = time.perf_counter()
            big.fit(None, None, features_struct=fb, device_targets=yb_d.ptr)
            barrier()
            tb = max_over_ranks(time.perf_counter() - tb)
            aux_big = {"n": nb, "ms_per_fit": 1e3 * tb, "fits_per_sec": 1.0 / tb, "scaling": "strong",
                       "tflops": nb ** 3 / 3. / tb / 1e12, "max_rel_residual": resid_b,
                       "one_gpu_reference_ms": 1670.,
                       "one_gpu_reference_source": "constant from profiles/r04/fit_vs_n.txt, not measured in this run",
                       "note": "ONE fit of N = 65536 row-block-sharded over all ranks (compute-bound regime)"}
            del big, xb_d, yb_d
        except Exception as exc:  # noqa: BLE001
            aux_big = {"error": f"{type(exc).__name__}: {exc}"}
            aux_broken = True

    # HBM traffic of the dominant kernel: PMC counters cannot be read from inside this process, so the per-launch figure
    # comes from the committed rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE in separate runs, gfx950 x2 read
    # correction applied) of the single-GPU run; null if absent or not applicable.
    traffic = traffic_src = None
    if world == 1 and not sharded and n == N_TRAIN:
        for rnd in ("r06", "r05", "r04", "r03", "r02", "r01"):
            try:
                with open(os.path.join(ROOT, "profiles", rnd, "pmc_traffic.json")) as fh:
                    traffic = json.load(fh)["traffic_bytes_per_launch"]
                traffic_src = f"profiles/{rnd}/pmc_traffic.json"
                break
            except (OSError, KeyError, ValueError):
                continue

    configs = None
    if rank == 0 and world == 1 and not sharded and not args.no_configs:
        x_d.free()
        y_d.free()
        configs = other_configs(ab, ctx)

    if rank == 0:
        # sharded: one fit per step over all ranks; replicas: every rank fits its own dataset
        fits = args.steps if (sharded or world == 1) else args.steps * world
        achieved = (gemm_flop / 1e12) / (gemm_ms * 1e-3) if gemm_ms > 0 else 0.0  # 0: no launch of that kernel at this size
        if sharded and world > 1:
            parallelism = (f"ONE fit row-block-sharded (512-row blocks, snake-cyclic) over {world} GPUs: per block column one RCCL "
                           "broadcast of the factored diagonal block + one RCCL all-gather of the panel rows, one block column of "
                           "look-ahead; back substitution: one all-reduce per 2048-row super-block")
            kernel_name = "agp::gemm_nt_sub_kernel (fp64 MFMA updates of rank 0's own row blocks, K=512)"
        else:
            parallelism = "1 GPU" if world == 1 else f"{world} independent fits, one per GPU, no data-path collective"
            if fell_back or fallback_note:
                parallelism = "FALLBACK - " + parallelism + f" (the sharded single-fit path was not used: {fallback_note})"
            kernel_name = "agp::trailing_update_kernel (fp64 MFMA bulk trailing update C -= P P^T, K=512)"
        stages = {"gram": gram_ms / args.steps, "factor": factor_ms / args.steps, "backward_solve": solve_ms / args.steps,
                  "trailing_update_kernels": gemm_ms / args.steps}
        if sharded:
            stages["note"] = "sharded entry point: `factor` is host wall time of factorisation + both substitutions"
        out = {
            "metric": f"GP fits/sec (Gram+Chol+solve) at N={n} fp64",
            "value": fits / elapsed,
            "unit": "fits/sec",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True,
            "scaling": "strong" if (sharded and world > 1) else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"dense GP fit, N={n}, 3-D SquaredExponential(1,1)+IndependentNoise(0.1), features from "
                                   "mt19937(44), inputs resident in HBM (BASELINE config 3 problem)",
                       "parallelism": parallelism,
                       "transport": transport, "n_ranks": (comm.world if comm is not None else (world if not sharded else 1))},
            "self_check": self_check,
            "roofline": {
                "bound": "mfma", "kernel": kernel_name,
                "achieved": achieved, "peak": MFMA_F64_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved / MFMA_F64_PEAK_TFLOPS,
                # NOT measured by this run (counters cannot be read from inside the benchmark process): the constant of the
                # committed rocprofv3 --pmc passes of the same command, named in traffic_source; null when there is none
                "traffic": traffic,
                "traffic_unit": "bytes per launch" if traffic_src else None,
                "traffic_source": f"constant from {traffic_src}, not measured in this run" if traffic_src else None,
                "launches_per_fit": gemm_launches / args.steps,
                "avg_launch_ms": gemm_ms / max(gemm_launches, 1.0),
                "flop_per_fit": gemm_flop / args.steps,
                # the whole fit against the same peak (n^3/3 flop over ms_per_step), and where the wall time of a step goes:
                # GPU stage events of the library (Gram / factor / back substitution) against the wall clock of the loop
                "whole_fit_frac": (n ** 3 / 3.) / (elapsed / args.steps) / 1e12 / MFMA_F64_PEAK_TFLOPS,
                "fit_stages_ms": stages,
                "wall_minus_stages_ms": 1e3 * elapsed / args.steps - (stages["gram"] + stages["factor"] + stages["backward_solve"]),
            },
        }
        if fell_back or fallback_note:
            out["sharded_fallback"] = fallback_note
        if predict is not None:
            out["predict"] = predict
        if aux is not None:
            out.update(aux)
        if aux_big is not None:
            out["n65536_sharded"] = aux_big
        if configs is not None:
            out["configs"] = configs
        if not args.no_cpu_baseline and world == 1:  # rank 0 at N = 1 only
            out["cpu_baseline"] = cpu_baseline()
        # LAST on the line (the driver's record keeps the final 2 kB verbatim): where a step's wall time goes.
        # ms_per_step_loop_only = the same K steps without the closing barrier (every step returns after its streams have
        # drained); closing_barrier_ms = what that barrier cost; max_step_ms / max_step_index = the slowest step.
        out["stages_ms_per_fit"] = stages
        out["timing"] = {"value": fits / elapsed, "ms_per_step": 1e3 * elapsed / args.steps,
                         "ms_per_step_loop_only": 1e3 * sum(step_walls) / args.steps,
                         "closing_barrier_ms": 1e3 * (elapsed - t_loop),
                         "min_step_ms": 1e3 * min(step_walls), "max_step_ms": 1e3 * max(step_walls),
                         "max_step_index": int(np.argmax(step_walls)),
                         "wall_minus_stages_ms": out["roofline"]["wall_minus_stages_ms"],
                         "roofline_frac": out["roofline"]["frac"], "avg_launch_ms": out["roofline"]["avg_launch_ms"],
                         "hip_runtimes_in_process": _hip_runtimes_loaded()}
        cb = out.get("cpu_baseline") or {}
        if cb.get("value"):
            lo, hi = cb.get("range_fits_per_sec", [cb["value"], cb["value"]])
            out["timing"]["gpu_over_faithful_cpu"] = [round(fits / elapsed / hi), round(fits / elapsed / lo)]  # (range: both cubic fits)
            sc = cb.get("strong_cpu") or {}
            if sc.get("value"):
                out["timing"]["gpu_over_strong_cpu"] = round(fits / elapsed / sc["value"], 1)
                out["timing"]["strong_cpu_gflops"] = round(sc["factor_gflops"])
        print(json.dumps(out), flush=True)
    bad_check = self_check is not None and not self_check["ok"]
    if aux_broken:  # the communicator may be mid-collective on a peer: no destructors
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(4 if bad_check else 0)
    if fell_back:
        # the abandoned communicator must not run its destructors (peers may sit in a collective of it)
        gloo_barrier()
        sys.stdout.flush()
        sys.stderr.flush()
        os._exit(4 if bad_check else (0 if args.allow_fallback else FALLBACK_EXIT))
    if comm is not None:
        comm.barrier()
        comm.close()
    if world > 1:
        dist.destroy_process_group()
    ctx.close()
    if bad_check:
        raise SystemExit("bench.py: self-check failed (see self_check in the JSON line)")


def main():
    args = parse_args()
    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if args.gpus > 1 and not under_launcher:
        sys.exit(launch_ranks(args, sys.argv[1:]))
    run_rank(args)


if __name__ == "__main__":
    main()
