"""Marginal prediction at M >> the bench's 4096: the test points pass in slices that keep the N x M workspace at 2 GiB."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab

ctx = ab.Context(0)
n = 16384
rng = np.random.default_rng(0)
x = rng.uniform(0., 10., (n, 3))
y = np.sin(x).sum(axis=1)
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
for M in (4096, 65536, 524288):
    xs = rng.uniform(0., 10., (M, 3))
    fm.predict(xs[:4096]).marginal()
    t = time.perf_counter(); p = fm.predict(xs).marginal(); dt = time.perf_counter() - t
    t = time.perf_counter(); mu = fm.predict(xs).mean(); dm = time.perf_counter() - t
    assert np.all(np.isfinite(p.covariance)) and np.abs(p.mean - mu).max() < 1e-9
    print(f"N={n} M={M}: marginal {dt*1e3:.1f} ms ({M/dt/1e3:.0f} k pts/s), mean only {dm*1e3:.2f} ms ({M/dm/1e6:.1f} M pts/s)")
