"""FITC (every observation its own group: the reference's LeaveOneOutGrouper) with more groups than the y extent of a
grid, against the oracle."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from oracle import oracle_py as orc

ctx = ab.Context(0)
n, m = 70000, 64
rng = np.random.default_rng(0)
x = np.sort(rng.uniform(0., 100., n))
y = np.sin(x) + 0.1 * rng.standard_normal(n)
cov = ab.SquaredExponential(2.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))
sorted_x = x.copy()
grouper = lambda f: np.searchsorted(sorted_x, np.asarray(f, dtype=np.float64).reshape(-1)) if np.ndim(f) else int(np.searchsorted(sorted_x, float(f)))
grouper.vectorized = True
u = np.linspace(0., 100., m)
model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "fitc", context=ctx)
model.set_param("inducing_nugget", 1e-6)
t = time.perf_counter()
fm = model.fit(ab.RegressionDataset(x, y))
print(f"FITC n={n} ({n} groups of one), m={m}: fit {time.perf_counter() - t:.3f} s, nll {fm.get_fit().nll:.6f}")
t = time.perf_counter()
ofit = orc.OracleSparseFit(cov, x, np.arange(n), y, None, u, 1e-8, 1e-6)
print(f"oracle {time.perf_counter() - t:.1f} s, nll {ofit.nll:.6f}")
v = ofit.information
print("information rel err", np.abs(fm.get_fit().information - v).max() / np.abs(v).max(), "nll diff", abs(fm.get_fit().nll - ofit.nll))
assert np.abs(fm.get_fit().information - v).max() <= 1e-6 * np.abs(v).max() and abs(fm.get_fit().nll - ofit.nll) <= 1e-8 * n
print("ok")
