"""log|K| of the mixed-precision factor against the fp64 factor, per product path (default fp16 x 2 with four products,
AGP_F16X2_TERMS=3, AGP_MIXED_F16=0 = bf16 x 3, AGP_MIXED_BF16=0 = fp32 MFMA) and covariance function: the figures
include/albatross_amd.h states for agp_fit_create_mixed.  One process per path (the switches are read at context creation).
Usage: python scripts/mixed_logdet_errors.py            (spawns the four paths)"""
import os, subprocess, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))

PATHS = (("fp16 x 2, 4 products (default)", {}), ("fp16 x 2, 3 products", {"AGP_F16X2_TERMS": "3"}),
         ("bf16 x 3", {"AGP_MIXED_F16": "0"}), ("fp32 MFMA", {"AGP_MIXED_BF16": "0"}))


def child():
    import albatross_amd as ab
    from conftest import synthetic_stations, temperature_covariance, synthetic_3d
    ctx = ab.Context(0)
    cases = []
    x, y = synthetic_3d(5300, 5 + 5300)
    cases.append(("Matern-5/2(2,1)+noise(0.1) N=5300", ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1), ab.FeatureSet(x), y))
    for n in (8192, 32768):
        x, y = synthetic_3d(n, 44)
        cases.append((f"SE(1,1)+noise(0.1) N={n}", ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1), ab.FeatureSet(x), y))
    ecef, h, temp = synthetic_stations(32768, 11)
    cov, scale = temperature_covariance(ab)
    cases.append(("temperature example N=32768", cov, ab.FeatureSet(ecef, [scale(h)]), temp - temp.mean()))
    for name, cov, train, y in cases:
        n = len(y)
        ds = ab.RegressionDataset(train, y)
        f64 = ab.gp_from_covariance(cov, context=ctx).fit(ds).get_fit()
        ld64 = f64.log_determinant
        del f64
        mm = ab.gp_from_covariance(cov, context=ctx)
        mm.precision = "mixed"
        fit = mm.fit(ds).get_fit()
        fit.accept_mixed_log_determinant = True
        err = fit.log_determinant - ld64
        print(f"    {name:36s} log det error {err:+.4f} = {abs(err) / n:.2e} N = {abs(err) / abs(ld64):.1e} relative; CG steps {mm.refinement_[0]}",
              flush=True)
        del fit


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for name, env in PATHS:
            print(name, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env={**os.environ, **env}, check=False)
