"""BASELINE config 2: 3-D Matern-5/2 + noise, N = 4096 fp64 dense fit + predict (M = 4096) on one MI355X."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import albatross_amd as ab

def data(n, seed):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0., 10., size=(n, 3))
    return x, np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])

ctx = ab.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [4096]:
    x, y = data(n, 42)
    xs, _ = data(4096, 43)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(x, y)
    def best(f, reps=7):
        f(); t = 1e9
        for _ in range(reps):
            t0 = time.perf_counter(); f(); t = min(t, time.perf_counter() - t0)
        return t
    t_fit = best(lambda: model.fit(ds), 40)  # (short fits: the clock needs a few of them back to back)
    fm = model.fit(ds)
    t_ll = best(lambda: model.log_likelihood(ds))
    p = fm.predict(xs)
    t_mean = best(lambda: p.mean())
    t_marg = best(lambda: p.marginal())
    t_joint = best(lambda: p.joint(), 3)
    print(f"N={n}: fit {1e3*t_fit:.2f} ms ({1/t_fit:.0f} fits/s, host inputs), log_likelihood {1e3*t_ll:.2f} ms, "
          f"predict M=4096: mean {1e3*t_mean:.2f} ms, marginal {1e3*t_marg:.2f} ms, joint {1e3*t_joint:.2f} ms (incl. 134 MB download)")
