"""fp64 fit vs mixed-precision fit (agp_fit_create_mixed) on BASELINE config 4's workload: the temperature-example
covariance on N synthetic stations.  Usage: python scripts/time_mixed.py [N ...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import albatross_amd as ab
from conftest import synthetic_stations, temperature_covariance, synthetic_3d

ctx = ab.Context(0)
for n in [int(a) for a in sys.argv[1:]] or [16384, 32768]:
    for name in ("config4", "config3"):
        if name == "config4":
            ecef, h, temp = synthetic_stations(n, 11)
            cov, scale = temperature_covariance(ab)
            train, y = ab.FeatureSet(ecef, [scale(h)]), temp - temp.mean()
        else:
            x, y = synthetic_3d(n, 44)
            cov, train = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1), ab.FeatureSet(x)
        ds = ab.RegressionDataset(train, y)
        out = {}
        for prec in ("fp64", "mixed", "mixed0"):
            model = ab.gp_from_covariance(cov, context=ctx)
            model.precision = prec[:5]
            if prec == "mixed0":
                model.max_refinements = 0  # factor + first solve only
            best = 1e9
            for rep in range(3):
                t0 = time.perf_counter()
                fm = model.fit(ds)
                dt = time.perf_counter() - t0
                best = min(best, dt)
                fit = fm.get_fit()
                fit.accept_mixed_log_determinant = True  # (reported as log_det_rel_err_vs_fp64)
                info, ld = fit.information.copy(), fit.log_determinant
                del fm, fit
            out[prec] = (best, info, ld, model.refinement_)
        t64, a64, ld64, _ = out["fp64"]
        tmx, amx, ldm, (its, res) = out["mixed"]
        print(f"{name} N={n}: fp64 fit {1e3*t64:.1f} ms, mixed fit {1e3*tmx:.1f} ms ({t64/tmx:.2f}x; without refinement "
              f"{1e3*out['mixed0'][0]:.1f} ms, first residual {out['mixed0'][3][1]:.1e}), CG steps {its}, "
              f"relative residual {res:.1e}, |a_mixed - a_fp64|/|a_fp64| {np.abs(amx-a64).max()/np.abs(a64).max():.1e}, "
              f"log det {ld64:.6f} vs {ldm:.6f} (rel {abs(ldm-ld64)/abs(ld64):.1e})", flush=True)
