"""Run three N=16384 fits (for rocprofv3 --kernel-trace); scripts/trace_timeline.py analyses the last one."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab

n = int(os.environ.get("TRACE_N", "16384"))
ctx = ab.Context(0)
cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
model = ab.gp_from_covariance(cov, context=ctx)
rng = np.random.default_rng(0)
x = rng.uniform(0., 10., (n, 3))
y = np.sin(x).sum(axis=1)
ds = ab.RegressionDataset(x, y)
for _ in range(3):
    fm = model.fit(ds)
    del fm
print("done")
