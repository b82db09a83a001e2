import sys, time
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import albatross_amd as ab
from conftest import synthetic_3d
ctx = ab.Context(0)
for n in (4096, 16384):
    x, y = synthetic_3d(n, 1)
    fm = ab.gp_from_covariance(ab.SquaredExponential(1., 1.) + ab.IndependentNoise(0.1), context=ctx).fit(ab.RegressionDataset(x, y))
    fit = fm.get_fit()
    for m in (1, 8, 64):
        B = np.asfortranarray(np.random.default_rng(0).standard_normal((n, m)))
        fit.solve(B if m > 1 else B[:, 0])
        t0 = time.perf_counter()
        for _ in range(3):
            out = fit.solve(B if m > 1 else B[:, 0])
        print(n, "nrhs", m, f"{(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
