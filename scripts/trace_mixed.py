"""Three mixed-precision fits of BASELINE config 4 (N = 32768 by default) for rocprofv3 --kernel-trace --stats: which kernels
the panel stream of the mixed factorisation spends its time in."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import albatross_amd as ab
from conftest import synthetic_stations, temperature_covariance

n = int(os.environ.get("TRACE_N", "32768"))
ctx = ab.Context(0)
ecef, h, temp = synthetic_stations(n, 11)
cov, scale = temperature_covariance(ab)
ds = ab.RegressionDataset(ab.FeatureSet(ecef, [scale(h)]), temp - temp.mean())
model = ab.gp_from_covariance(cov, context=ctx)
model.precision = "mixed"
for rep in range(3):
    t0 = time.perf_counter()
    fm = model.fit(ds)
    print(f"mixed fit N={n}: {1e3 * (time.perf_counter() - t0):.1f} ms", flush=True)
    del fm
