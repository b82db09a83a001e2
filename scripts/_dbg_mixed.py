import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
import bench
ctx = ab.Context(0)
n = 32768
ecef, h, temp = bench.synthetic_stations(n, 11)
cov, scale = bench.temperature_covariance(ab)
ds = ab.RegressionDataset(ab.FeatureSet(ecef, [scale(h)]), temp - temp.mean())
for prof in (False, True):
    ctx.set_profiling(prof)
    for prec in ("fp64", "mixed"):
        model = ab.gp_from_covariance(cov, context=ctx)
        model.precision = prec
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); fm = model.fit(ds); ts.append(time.perf_counter() - t0); del fm
        print(f"profiling={prof} {prec}: " + " ".join(f"{1e3*t:.1f}" for t in ts), flush=True)
