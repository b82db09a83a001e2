"""Repeated use of every entry point; device memory in use must return to its starting level."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
torch.cuda.init()
import albatross_amd as ab

def used():
    free, total = torch.cuda.mem_get_info()
    return (total - free) / 2**20

ctx = ab.Context(0)
rng = np.random.default_rng(0)
n = 1500
x = rng.uniform(0., 10., (n, 2)); y = np.sin(x).sum(axis=1) + 0.1 * rng.standard_normal(n)
cov = ab.Matern52(2.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.2))
model = ab.gp_from_covariance(cov, context=ctx)
ds = ab.RegressionDataset(x, y)
xs = rng.uniform(0., 10., (200, 2))
groups = [list(range(g * 100, (g + 1) * 100)) for g in range(15)]
ragged = [list(range(0, 130)), list(range(130, 400)), list(range(400, 700)), list(range(700, 1100)), list(range(1100, 1500))]
u = rng.uniform(0., 10., (64, 2))
sparse = ab.sparse_gp_from_covariance(cov, lambda f: int(f[0] // 2.5), ab.FixedInducingPoints(u), "s", context=ctx)
sparse.set_param("inducing_nugget", 1e-6)

def cycle():
    fm = model.fit(ds)
    fm.predict(xs).joint(); fm.predict_with_measurement_noise(xs).marginal(); fm.predict(xs).mean()
    model.log_likelihood(ds)
    model.log_likelihoods(ds, [{}, {"sigma_matern_52": 1.1}])
    fit = fm.get_fit()
    fit.solve(np.ones((n, 3))); fit.inverse_diagonal(); fit.leave_one_out(y)
    fit.held_out_predictions(y, groups, joint=True); fit.held_out_predictions(y, ragged); fit.inverse_blocks(groups[:3])
    fm.update(ab.RegressionDataset(xs, np.zeros(200))).predict(xs[:10]).joint()
    sf = sparse.fit(ds)
    sf.predict(xs).joint(); sf.update(ab.RegressionDataset(xs, np.zeros(200))).predict(xs).marginal()
    sparse.log_likelihood(ds)
    rb = ab.rebase_inducing_points(sf, rng.uniform(0., 10., (40, 2)))   # pivoted L D L^T + pivoted QR
    rb.predict(xs).marginal(); rb.update(ab.RegressionDataset(xs, np.zeros(200))).predict(xs).joint()
    ab.DenseFactor(np.eye(300) * 2.0, ctx).solve(np.ones(300))
    ctx.gram(cov, x[:100], xs)

cycle(); cycle()
ctx.synchronize(); torch.cuda.synchronize()
base = used()
for i in range(30):
    cycle()
ctx.synchronize(); torch.cuda.synchronize()
import gc; gc.collect()
after = used()
print(f"device memory in use: {base:.1f} MiB after warm-up, {after:.1f} MiB after 30 more cycles (delta {after - base:+.1f} MiB)")
assert after - base < 64.0, "device memory grows with use"
print("ok")
