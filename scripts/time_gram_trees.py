"""Gram (lower triangle, fit path) and fused predict-mean time for the covariance trees of the BASELINE configs:
fast path (radial<Euclidean> + noise), sum-of-products evaluator, postfix interpreter (AGP_GRAM_SOP=0)."""
import ctypes as C
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
import albatross_amd as ab
from conftest import synthetic_stations, temperature_covariance, synthetic_3d

ctx = ab.Context(0)
n = 16384
x, y = synthetic_3d(n, 44)
ecef, h, temp = synthetic_stations(n, 11)
tcov, scale = temperature_covariance(ab)
cases = [("config 3: SE + noise (fast path)", ab.SquaredExponential(1., 1.) + ab.IndependentNoise(0.1), ab.FeatureSet(x), y),
         ("config 2: Matern52 + noise (fast path)", ab.Matern52(2., 1.) + ab.IndependentNoise(0.1), ab.FeatureSet(x), y),
         ("config 4: scaling*const + noise + exp<angular>*se<radial>", tcov, ab.FeatureSet(ecef, [scale(h)]), temp - temp.mean()),
         ("se*matern52 + meas(noise) + const", ab.SquaredExponential(3., 1.) * ab.Matern52(2., 1.) + ab.measurement_only(ab.IndependentNoise(0.1)) + ab.Constant(0.5), ab.FeatureSet(x), y)]
for name, cov, feats, yy in cases:
    model = ab.gp_from_covariance(cov, context=ctx)
    ctx.set_profiling(True)
    ds = ab.RegressionDataset(feats, yy)
    fm = model.fit(ds)
    fm = model.fit(ds)
    gram_ms = ctx.stage_ms(0)
    xs = feats if not isinstance(feats, ab.FeatureSet) else ab.FeatureSet(feats.coords[:4096], None if feats.scales is None else list(feats.scales[:4096].T))
    fm.predict(xs).mean()
    t0 = time.perf_counter()
    for _ in range(5):
        fm.predict(xs).mean()
    pm = (time.perf_counter() - t0) / 5
    print(f"{name:62s}: gram {gram_ms:6.3f} ms, predict mean (M=4096, incl. transfers) {1e3 * pm:6.3f} ms", flush=True)
