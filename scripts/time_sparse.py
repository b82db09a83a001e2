"""BASELINE config 5: SparseGP PITC, N = 262144, 2048 inducing points, independent groups of 512."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab

ctx = ab.Context(0)
CASES = ((32768, 1024, 512), (262144, 2048, 512), (262144, 2048, -512))
if len(sys.argv) > 1:  # e.g. `time_sparse.py 1`: config 5 only (for a kernel trace)
    CASES = tuple(CASES[int(a)] for a in sys.argv[1:])
for n, m, gs in CASES:
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., n / 16., n))              # 1-D, ~16 points per unit length
    y = np.sin(x) + 0.1 * np.cos(10. * x) + 0.1 * rng.standard_normal(n)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))  # bench covariance
    u = np.linspace(x.min(), x.max(), m)
    rank = np.argsort(np.argsort(x))
    if gs > 0:
        group_of = {float(xi): int(r // gs) for xi, r in zip(x, rank)}
    else:  # ragged: group sizes uniform in [0.5, 1.3] * |gs|
        sizes = rng.integers(int(0.5 * -gs), int(1.3 * -gs), size=2 * n // -gs)
        bounds = np.cumsum(sizes)
        gid = np.searchsorted(bounds, np.arange(n), side="right")
        group_of = {float(xi): int(gid[r]) for xi, r in zip(x, rank)}
    sorted_x = np.sort(x)
    if gs > 0:
        def grouper(f):  # group = rank of the feature // group size; accepts one feature or the whole array
            return np.searchsorted(sorted_x, np.asarray(f, dtype=np.float64).reshape(-1)) // gs if np.ndim(f) else group_of[float(f)]
        grouper.vectorized = True
    else:
        grouper = lambda f: group_of[float(f)]
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "pitc", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    ds = ab.RegressionDataset(x, y)
    t = time.perf_counter(); fm = model.fit(ds); t1 = time.perf_counter() - t
    t = time.perf_counter(); fm = model.fit(ds); t2 = time.perf_counter() - t
    t = time.perf_counter(); model._components(ds); th = time.perf_counter() - t
    ctx.set_profiling(True) if hasattr(ctx, "set_profiling") else None
    xs = np.linspace(x.min(), x.max(), 4096)
    fm.predict(xs).marginal()
    t = time.perf_counter(); p = fm.predict(xs).marginal(); tp = time.perf_counter() - t
    resid = np.sqrt(np.mean((fm.predict(x[::64]).mean() - np.sin(x[::64]) - 0.1 * np.cos(10. * x[::64])) ** 2))
    print(f"N={n} m={m} groups of {gs}: fit {t1*1e3:.0f} ms (second {t2*1e3:.0f} ms), nll {fm.get_fit().nll:.3f}, "
          f"host grouping/reordering {th*1e3:.0f} ms, predict marginal M=4096 {tp*1e3:.1f} ms, rms error vs truth {resid:.3f}")
