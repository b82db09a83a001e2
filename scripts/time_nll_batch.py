"""Tuner objective batching: P + 1 log-likelihood evaluations of a finite-difference gradient,
one agp_nll each vs one agp_nll_batch."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab

ctx = ab.Context(0)
for n in (256, 1024, 4096):
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10., (n, 3))
    y = np.sin(x).sum(axis=1) + 0.1 * rng.standard_normal(n)
    cov = ab.Constant(0.5) + ab.Matern52(2.0, 1.0) + ab.SquaredExponential(5.0, 0.5) + ab.IndependentNoise(0.1)
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(x, y)
    base = model.get_params()
    sets = [{}] + [{k: v + 1e-6} for k, v in base.items()]
    model.log_likelihoods(ds, sets)
    reps = 5
    t = time.perf_counter()
    for _ in range(reps):
        batch = model.log_likelihoods(ds, sets)
    tb = (time.perf_counter() - t) / reps
    t = time.perf_counter()
    for _ in range(reps):
        single = []
        for s in sets:
            for k, v in s.items():
                model.set_param(k, v)
            single.append(model.log_likelihood(ds))
            for k in s:
                model.set_param(k, base[k])
    ts = (time.perf_counter() - t) / reps
    print(f"N={n}: {len(sets)} evaluations: one by one {ts*1e3:.2f} ms, batched {tb*1e3:.2f} ms ({ts/tb:.1f}x), "
          f"max |diff| {np.abs(np.array(single) - batch).max():.2e}")
