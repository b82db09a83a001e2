"""BASELINE config 5 (N = 262144, m = 2048, groups of 512) on 1 GPU, and ONE rank's share of it on 8 GPUs (1/8 of the
groups, the same m: what agp_sparse_fit_create_sharded does per rank apart from two 32 MiB all-reduces and four
m-vectors)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab

ctx = ab.Context(0)
n_full, m, gs = 262144, 2048, 512
rng = np.random.default_rng(n_full)
x = np.sort(rng.uniform(0., n_full / 16., n_full))
y = np.sin(x) + 0.1 * np.cos(10. * x) + 0.1 * rng.standard_normal(n_full)
cov = ab.SquaredExponential(1.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))
u = np.linspace(x.min(), x.max(), m)
sorted_x = np.sort(x)


def grouper(f):
    r = np.searchsorted(sorted_x, np.asarray(f, dtype=np.float64).reshape(-1)) // gs
    return r if np.ndim(f) else int(r[0])


grouper.vectorized = True
model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "pitc", context=ctx)
model.set_param("inducing_nugget", 1e-6)
groups = np.arange(n_full) // gs
for world in (1, 2, 4, 8):
    mine = (groups % world) == 0
    ds = ab.RegressionDataset(x[mine], y[mine])
    model.fit(ds)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        model.fit(ds)
        ts.append(time.perf_counter() - t0)
    print(f"config 5, rank 0 of {world}: {int(mine.sum())} observations, m = {m}: {1e3 * min(ts):7.1f} ms per fit", flush=True)
