"""Predict marginal / joint at few test points (M = 1 .. 512) against one N = 16384 fit."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import numpy as np
import albatross_amd as ab
from conftest import synthetic_3d
ctx = ab.Context(0)
for n in (4096, 16384):
    x, y = synthetic_3d(n, 1)
    fm = ab.gp_from_covariance(ab.SquaredExponential(1., 1.) + ab.IndependentNoise(0.1), context=ctx).fit(ab.RegressionDataset(x, y))
    for m in (1, 8, 64, 512, 1024, 2048, 4096):
        xs, _ = synthetic_3d(m, 2)
        fm.predict(xs).marginal()
        t0 = time.perf_counter()
        for _ in range(3):
            fm.predict(xs).marginal()
        tm = (time.perf_counter() - t0) / 3
        t0 = time.perf_counter()
        for _ in range(3):
            fm.predict(xs).joint()
        print(f"N={n} M={m}: marginal {tm*1e3:.2f} ms, joint {(time.perf_counter() - t0) / 3 * 1e3:.2f} ms")
