"""Sanity + timing beyond the bench size: N = 20000 (not a multiple of 128) and N = 32768."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab

ctx = ab.Context(0)
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
model = ab.gp_from_covariance(cov, context=ctx)
for n in (20000, 32768):
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10. * (n / 16384) ** (1 / 3), (n, 3))  # same point density as the bench
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    ds = ab.RegressionDataset(x, y)
    t = time.perf_counter(); fm = model.fit(ds); t_fit = time.perf_counter() - t
    t = time.perf_counter(); fm2 = model.fit(ds); t_fit2 = time.perf_counter() - t
    alpha = fm2.get_fit().information
    # residual of K alpha = y on sampled rows, Gram computed row-block-wise on the device
    rows = np.sort(rng.choice(n, 512, replace=False))
    Kr = ctx.gram(cov, ab.Measurement(x[rows]), ab.Measurement(x))
    resid = np.abs(Kr @ alpha - y[rows]).max()
    t = time.perf_counter(); ll = model.log_likelihood(ds); t_nll = time.perf_counter() - t
    expect = -0.5 * (fm2.get_fit().log_determinant + y @ alpha + n * np.log(2 * np.pi))
    print(f"N={n}: fit {t_fit*1e3:.1f} ms (second {t_fit2*1e3:.1f} ms, {n**3/3/t_fit2/1e12:.1f} TFLOP/s), nll {t_nll*1e3:.1f} ms, "
          f"max |K alpha - y| on 512 rows = {resid:.2e}, loglik {ll:.6f} vs from fit {expect:.6f}")
    assert resid < 1e-8 and abs(ll - expect) < 1e-6 * n
    del fm, fm2
print("ok")
