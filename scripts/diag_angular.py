import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import albatross_amd as ab
from oracle import oracle_py as orc
ctx = ab.Context(0)
rng = np.random.default_rng(127 * 31 + 2)
x = rng.uniform(0.5, 10., (127, 2))
x[3] = x[1]
for nm, cov in [("exp_ang", ab.Exponential(1.1, 1.0, ab.AngularDistance())),
                ("se_rad", ab.SquaredExponential(6.0, 3.7, ab.RadialDistance()))]:
    got = ctx.gram(cov, x); want = orc.gram(cov, x)
    d = np.abs(got - want)
    i, j = np.unravel_index(np.argmax(d / np.abs(want)), d.shape)
    print(nm, "max abs", d.max(), "max rel", (d / np.abs(want)).max(), "at", i, j, got[i, j], want[i, j])
    theta_g = -1.1 * np.log(got[i, j]); theta_w = -1.1 * np.log(want[i, j])
    print("   theta", theta_g, theta_w, "x_i", x[i], "x_j", x[j])
    bad = np.argwhere(d > 4e-16 + 2e-14 * np.abs(want))
    print("   n bad", len(bad), bad[:10].tolist())
cov = ab.Exponential(1.1, 1.0, ab.AngularDistance()) * ab.SquaredExponential(6.0, 3.7, ab.RadialDistance()) + ab.measurement_only(ab.IndependentNoise(1.75))
for meas in (False, True):
    got = ctx.gram(cov, ab.Measurement(x) if meas else x); want = orc.gram(cov, x, x_meas=meas)
    d = np.abs(got - want)
    bad = np.argwhere(d > 4e-16 * np.abs(want).max() + 2e-14 * np.abs(want))
    print("composite meas", meas, "max abs", d.max(), "n bad", len(bad), bad[:6].tolist())
    for i, j in bad[:4]:
        print("    ", i, j, got[i, j], want[i, j])
