"""GPU parity of the full path: gp.fit(dataset) / fit.predict(x) /
model.log_likelihood(dataset) through the C-ABI vs the oracle (pivoted LDL^T,
the reference's algorithm) and vs the golden fixtures.

Stated fp64 tolerances: information / predictive mean /
variance relative error <= 1e-8 (ill-conditioned toy case: its fixture's own
1e-7), log-likelihood absolute error <= 1e-6 * N."""
import numpy as np
import pytest

import albatross_amd as ab
from conftest import golden, synthetic_3d
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def simple_cov(c):
    return ab.SquaredExponential(c["squared_exponential_length_scale"], c["sigma_squared_exponential"])


def test_toy_linear_golden(ctx):
    g = golden("toy_linear.json")
    cov = simple_cov(g["cov"]) + ab.measurement_only(ab.IndependentNoise(g["cov"]["sigma_independent_noise"]))
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(np.array(g["x"]), np.array(g["y"]))
    fm = model.fit(ds)
    tol = g["tolerance_rel"]
    assert rel(fm.get_fit().information, g["information"]) <= tol
    assert abs(fm.get_fit().log_determinant - g["log_det"]) <= 1e-8 * abs(g["log_det"])
    assert abs(-model.log_likelihood(ds) - g["nll"]) <= 1e-7 * abs(g["nll"])
    for p in g["predictions"]:
        pred = fm.predict(np.array(p["xs"]))
        joint = pred.joint()
        assert rel(joint.mean, p["mean"]) <= tol
        assert np.abs(joint.covariance - np.array(p["cov"])).max() <= 1e-5
        assert rel(pred.mean(), p["mean"]) <= tol
        marg = pred.marginal()
        assert np.abs(marg.covariance - np.diag(np.array(p["cov"]))).max() <= 1e-5
    # expect_predict_variants_consistent (test_models.h:324-432): 1e-8 between variants
    pred = fm.predict(np.array([0.1, 1.1, 2.2]))
    assert np.abs(pred.mean() - pred.joint().mean).max() < 1e-8
    assert np.abs(pred.marginal().covariance - np.diag(pred.joint().covariance)).max() < 1e-8 * 1e4


def test_bench512_golden(ctx):
    g = golden("bench512.json")
    cov = simple_cov(g["cov"]) + ab.IndependentNoise(g["cov"]["sigma_independent_noise"])
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(np.array(g["x"]), np.array(g["y"]))
    fm = model.fit(ds)
    tol = 1e-8
    assert rel(fm.get_fit().information, g["information"]) <= tol
    assert abs(fm.get_fit().log_determinant - g["log_det"]) <= 1e-9 * abs(g["log_det"])
    assert abs(-model.log_likelihood(ds) - g["nll"]) <= 1e-6 * 512
    marg = fm.predict(np.array(g["xs"])).marginal()
    assert np.abs(marg.mean - np.array(g["mean"])).max() <= tol
    assert np.abs(marg.covariance - np.array(g["variance"])).max() <= tol


CASES = [
    ("matern52+noise 3-D", lambda: ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1), 3),
    ("se+noise 3-D", lambda: ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1), 3),
    ("exp*se + meas-noise 2-D", lambda: ab.Exponential(4.0, 1.3) * ab.SquaredExponential(6.0, 1.1)
     + ab.measurement_only(ab.IndependentNoise(0.2)), 2),
]


@pytest.mark.parametrize("label,make,dim", CASES)
@pytest.mark.parametrize("n,m", [(1, 3), (17, 5), (128, 64), (129, 65), (640, 200), (1500, 333)])
def test_fit_predict_matches_oracle(ctx, label, make, dim, n, m):
    rng = np.random.default_rng(n + 13 * m)
    x = rng.uniform(0., 10., (n, dim))
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    yvar = rng.uniform(0.0, 0.05, n)
    xs = rng.uniform(0., 10., (m, dim))
    cov = make()
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar))
    fm = model.fit(ds)
    ofit = orc.OracleFit(cov, x, y, yvar)  # pivoted LDL^T, as the reference
    assert rel(fm.get_fit().information, ofit.information) <= 1e-8
    assert abs(fm.get_fit().log_determinant - ofit.log_determinant) <= 1e-6 * n
    assert abs(-model.log_likelihood(ds) - orc.nll(cov, x, y)) <= 1e-6 * n  # no target variance: gp.hpp:442-451
    pred = fm.predict(xs)
    om, ov = ofit.predict_marginal(xs)
    marg = pred.marginal()
    assert rel(marg.mean, om) <= 1e-8 and rel(pred.mean(), om) <= 1e-8
    assert np.abs(marg.covariance - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-9
    jm, jc = ofit.predict_joint(xs)
    joint = pred.joint()
    assert rel(joint.mean, jm) <= 1e-8
    assert np.abs(joint.covariance - jc).max() <= 1e-8 * np.abs(jc).max() + 1e-9
    assert np.array_equal(joint.covariance, joint.covariance.T)
    # predict_with_measurement_noise wraps the test features (fit_model.hpp:54-62)
    om2, ov2 = ofit.predict_marginal(xs, xs_meas=True)
    marg2 = fm.predict_with_measurement_noise(xs).marginal()
    assert np.abs(marg2.covariance - ov2).max() <= 1e-8 * np.abs(ov2).max() + 1e-9
    # CovarianceRepresentation::solve
    B = rng.standard_normal((n, 4))
    assert rel(fm.get_fit().solve(B), ofit.solve(B)) <= 1e-8


def test_mean_function_is_removed_and_added(ctx):
    # tests/test_gp.cc:344-371,464-506
    rng = np.random.default_rng(2)
    x = np.linspace(0., 10., 60)
    y = 3. * x + 1. + 0.01 * rng.standard_normal(60)
    cov = ab.SquaredExponential(2., 1.) + ab.measurement_only(ab.IndependentNoise(0.1))
    with_mean = ab.gp_from_covariance_and_mean(cov, ab.LinearMean(3., 1.), context=ctx).fit(ab.RegressionDataset(x, y))
    without = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    far = np.array([-20., 40.])
    assert np.abs(with_mean.predict(far).mean() - (3. * far + 1.)).max() < 1e-3
    assert np.linalg.norm(with_mean.predict(far).mean() - without.predict(far).mean()) > 1.


def test_linear_mean_gp_matches_golden_and_oracle(ctx):
    """MeanFunction::remove_from / add_to (mean_function.hpp:86-107), LinearMean (polynomials.hpp:92-106) through
    fit / predict / log_likelihood: against tests/golden/toy_linear_mean.json (MakeGaussianProcessWithMean,
    test_models.h:75-96, and the model of tests/test_gp.cc:344-371; expected values from scipy) and against the oracle."""
    g = golden("toy_linear_mean.json")
    x, y = np.array(g["x"]), np.array(g["y"])
    mean = ab.LinearMean(g["mean"]["slope"], g["mean"]["offset"])
    for mdl in g["models"]:
        c = mdl["cov"]
        cov = ab.SquaredExponential(c["squared_exponential_length_scale"], c["sigma_squared_exponential"]) \
            + ab.measurement_only(ab.IndependentNoise(c["sigma_independent_noise"]))
        tol = mdl["tolerance_rel"]
        model = ab.gp_from_covariance_and_mean(cov, mean, context=ctx)
        ds = ab.RegressionDataset(x, y)
        fm = model.fit(ds)
        ofit = orc.OracleFit(cov, x, y, mean=mean)
        assert rel(fm.get_fit().information, np.array(mdl["information"])) <= tol
        assert rel(fm.get_fit().information, ofit.information) <= tol
        for p in mdl["predictions"]:
            xs, want = np.array(p["xs"]), np.array(p["mean"])
            pred = fm.predict(xs)
            for got in (pred.mean(), pred.marginal().mean, pred.joint().mean):
                assert rel(got, want) <= tol and rel(got, ofit.predict_mean(xs)) <= tol
            assert np.abs(pred.joint().covariance - np.array(p["cov"])).max() <= 1e-5 * c["sigma_squared_exponential"] ** 2
        assert abs(-model.log_likelihood(ds) - mdl["nll"]) <= 1e-7 * abs(mdl["nll"])
        assert abs(-model.log_likelihood(ds) - orc.nll(cov, x, y, mean=mean)) <= 1e-7 * abs(mdl["nll"])
    # composed mean functions, 3-D features (LinearMean reads the first coordinate), target variance
    rng = np.random.default_rng(11)
    x3 = rng.uniform(0., 10., (200, 3))
    y3 = np.sin(x3).sum(axis=1) + 0.7 * x3[:, 0] - 2.
    yv = rng.uniform(0.01, 0.05, 200)
    xs3 = rng.uniform(0., 10., (30, 3))
    comp = ab.LinearMean(0.7, -2.) + ab.LinearMean(0.1, 0.) * ab.LinearMean(0., 0.5)
    cov3 = ab.Matern52(2., 1.) + ab.IndependentNoise(0.1)
    model = ab.gp_from_covariance_and_mean(cov3, comp, context=ctx)
    ds3 = ab.RegressionDataset(x3, ab.MarginalDistribution(y3, yv))
    fm = model.fit(ds3)
    ofit = orc.OracleFit(cov3, x3, y3, yv, mean=comp)
    assert rel(fm.get_fit().information, ofit.information) <= 1e-8
    om, ov = ofit.predict_marginal(xs3)
    marg = fm.predict(xs3).marginal()
    assert rel(marg.mean, om) <= 1e-8 and np.abs(marg.covariance - ov).max() <= 1e-8
    assert abs(-model.log_likelihood(ds3) - orc.nll(cov3, x3, y3, mean=comp)) <= 1e-6 * 200


def test_fit_from_prediction_with_mean(ctx):
    """tests/test_gp.cc:344-371 (test_model_from_prediction_with_mean): fit_from_prediction must remove the mean
    function from the prediction before it builds the new fit (gp.hpp:236-245)."""
    g = golden("toy_linear_mean.json")
    x, y = np.array(g["x"]), np.array(g["y"])
    cov = ab.SquaredExponential(2., 1.) + ab.measurement_only(ab.IndependentNoise(0.1))
    model = ab.gp_from_covariance_and_mean(cov, ab.LinearMean(g["mean"]["slope"], g["mean"]["offset"]), context=ctx)
    fm = model.fit(ab.RegressionDataset(x, y))
    feats = np.array([1.3, 4.2, 7.1])
    pred = fm.predict(feats).joint()
    again = model.fit_from_prediction(feats, pred).predict(feats).joint()
    assert np.linalg.norm(again.mean - pred.mean) <= 1e-6                 # the reference's own bars
    assert np.linalg.norm(again.covariance - pred.covariance) <= 1e-6
    want = np.array(g["models"][1]["predictions"][0]["mean"])
    assert rel(again.mean, want) <= 1e-8


def test_not_positive_definite_and_nan_are_reported(ctx):
    x = np.array([0., 0., 1.])  # duplicate point, no noise: singular Gram
    model = ab.gp_from_covariance(ab.SquaredExponential(1., 1.), context=ctx)
    model.pivoted_fallback = False  # the un-pivoted device factor alone: reported with the failing pivot
    with pytest.raises(ab.NotPositiveDefiniteError, match="pivot 1"):
        model.fit(ab.RegressionDataset(x, np.zeros(3)))
    with pytest.raises(ab.NanInputError):
        model.fit(ab.RegressionDataset(np.array([0., np.nan, 2.]), np.zeros(3)))
    with pytest.raises(ab.NanInputError):
        model.log_likelihood(ab.RegressionDataset(np.array([0., np.nan, 2.]), np.zeros(3)))


def test_config2_n4096_matern(ctx):
    """BASELINE config 2: 3-D Matern-5/2 + noise, N = 4096, dense fit + predict.
    Checked against the oracle's un-pivoted LL^T (the pivoted LDL^T takes
    minutes at this size) plus residual properties."""
    x, y = synthetic_3d(4096, 42)
    xs, _ = synthetic_3d(512, 43)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    model = ab.gp_from_covariance(cov, context=ctx)
    fm = model.fit(ab.RegressionDataset(x, y))
    ofit = orc.OracleFit(cov, x, y, threads=8, use_llt=True)
    assert rel(fm.get_fit().information, ofit.information) <= 1e-8
    assert abs(fm.get_fit().log_determinant - ofit.log_determinant) <= 1e-6 * 4096
    om, ov = ofit.predict_marginal(xs)
    marg = fm.predict(xs).marginal()
    assert rel(marg.mean, om) <= 1e-8
    assert np.abs(marg.covariance - ov).max() <= 1e-8


def test_config3_n16384_properties(ctx):
    """BASELINE config 3 (N = 16384, SE + noise) through size-independent
    properties: K alpha = y residual, L L^T = K on sampled rows, solve round
    trip, NLL consistent with the fit's log-det and information."""
    n = 16384
    x, y = synthetic_3d(n, 44)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(x, y)
    fm = model.fit(ds)
    alpha = fm.get_fit().information
    K = ctx.gram(cov, ab.Measurement(x))
    resid = np.abs(K @ alpha - y).max()
    assert resid <= 1e-9 * (np.abs(K).sum(axis=1).max() * np.abs(alpha).max())
    nll = -model.log_likelihood(ds)
    expect = 0.5 * (fm.get_fit().log_determinant + y @ alpha + n * np.log(2 * np.pi))
    assert abs(nll - expect) <= 1e-6 * n
    rows = np.random.default_rng(1).choice(n, 64, replace=False)
    L = fm.get_fit().factor()
    assert np.abs(L[rows] @ L.T - K[rows]).max() <= 1e-11
    B = np.random.default_rng(2).standard_normal((n, 2))
    X = fm.get_fit().solve(B)
    assert np.abs(K @ X - B).max() <= 1e-8 * np.abs(B).max() * 10
    # The leading 2048 x 2048 block of L is the LL^T factor of the leading principal sub-matrix - the first four outer blocks
    # of the N = 16384 schedule (two-stream look-ahead, bulk launches that carry the next block column, left-looking inner
    # panels), pinned to the ORACLE on the sub-problem of the first 2048 points: log-determinant and solves through the
    # oracle's own un-pivoted LL^T (threads = 8, as config 2), and - an LL^T factor with a positive diagonal is unique -
    # element by element against LAPACK's factor of the oracle's Gram matrix.
    import scipy.linalg as sla
    m = 2048
    L11 = np.ascontiguousarray(L[:m, :m])
    assert np.all(np.triu(L11, 1) == 0.) and np.all(np.diag(L11) > 0.)
    ofit = orc.OracleFit(cov, x[:m], y[:m], threads=8, use_llt=True)
    assert abs(2. * np.log(np.diag(L11)).sum() - ofit.log_determinant) <= 1e-6 * m
    Bs = np.random.default_rng(3).standard_normal((m, 2))
    assert rel(sla.cho_solve((L11, True), Bs), ofit.solve(Bs)) <= 1e-8
    assert rel(sla.cho_solve((L11, True), y[:m]), ofit.information) <= 1e-8
    Ko = orc.gram(cov, x[:m], x_meas=True)
    assert np.abs(L11 - np.linalg.cholesky(Ko)).max() <= 1e-9


@pytest.mark.parametrize("n", [1, 50, 128, 300, 1000, 1700])
def test_leave_one_out_fast_path(ctx, n):
    """tests/test_cross_validation.cc:419-446 / test_serializable_ldlt.cc:52-66 restated:
    the LOO fast path equals the oracle's (which equals brute-force refits, see
    tests/test_oracle_golden.py) at 1e-8."""
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10., (n, 2))
    y = np.sin(x).sum(axis=1) + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.05, n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    ofit = orc.OracleFit(cov, x, y, yvar)
    d = fm.get_fit().inverse_diagonal()
    od = ofit.inverse_diagonal()
    assert np.abs(d - od).max() <= 1e-8 * od.max()
    loo = fm.get_fit().leave_one_out(y)
    om, ov = ofit.loo_marginal(y)
    assert np.abs(loo.mean - om).max() <= 1e-8 * max(np.abs(om).max(), 1.)
    assert np.abs(loo.covariance - ov).max() <= 1e-8 * ov.max()


def test_leave_one_out_n16384_property(ctx):
    """At the bench size: (K^-1)_ii from the fast path vs a direct solve of K x = e_i on sampled i."""
    n = 16384
    x, y = synthetic_3d(n, 44)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    d = fm.get_fit().inverse_diagonal()
    idx = np.random.default_rng(0).choice(n, 6, replace=False)
    E = np.zeros((n, 6))
    E[idx, np.arange(6)] = 1.
    X = fm.get_fit().solve(E)
    assert np.abs(X[idx, np.arange(6)] - d[idx]).max() <= 1e-9 * d.max()
    loo = fm.get_fit().leave_one_out(y)
    assert np.all(loo.covariance > 0) and np.abs(loo.mean - y).max() < 1.0


def _elevation_model(ctx):
    class Elevation(ab.ScalingFunction):
        _params = {"elevation_scaling_center": 4.0, "elevation_scaling_factor": 0.3}

        def get_name(self):
            return "elevation_scaling"

        def _call_impl(self, c):
            p = self.get_params()
            return 1. + p["elevation_scaling_factor"] * np.maximum(p["elevation_scaling_center"] - np.asarray(c)[:, 2], 0.)

    cov = ab.ScalingTerm(Elevation()) * ab.Constant(0.5) + ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    return cov, ab.gp_from_covariance_and_mean(cov, ab.LinearMean(), context=ctx)


def _with_overrides(ctx, cov, mean_function, overrides):
    import copy
    m = ab.gp_from_covariance_and_mean(copy.deepcopy(cov), copy.deepcopy(mean_function), context=ctx)
    m.set_param_values(overrides)
    return m


@pytest.mark.parametrize("n", [60, 256, 700])
def test_batched_log_likelihoods_match_oracle(ctx, n):
    """agp_nll_batch against the ORACLE's log_likelihood (gp.hpp:442-451) per parameter vector
    - the evaluations of the tuner's finite-difference gradient (tune/finite_difference.hpp:20-94) - including a
    ScalingTerm parameter, a mean-function parameter, and a parameter vector that is not positive definite; and
    against agp_nll one by one."""
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10., (n, 3))
    y = np.sin(x).sum(axis=1) + 0.5 * x[:, 0] + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.03, n)  # must be ignored by log_likelihood
    cov, model = _elevation_model(ctx)
    ds = ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar))
    base = model.get_params()
    sets = [{}]
    for name in base:  # forward differences, epsilon as in compute_gradient
        sets.append({name: base[name] + 1e-6 * max(1., abs(base[name]))})
    sets.append({"matern_52_length_scale": 0.5, "sigma_matern_52": 3.0})
    sets.append({"slope": 0.4, "offset": -1.0})
    got = model.log_likelihoods(ds, sets)
    for overrides, g in zip(sets, got):
        m = _with_overrides(ctx, cov, model.mean_function_, overrides)
        want = -orc.nll(m.covariance_function_, x, y, mean=m.mean_function_)
        assert abs(g - want) <= 1e-8 * n, overrides
        assert abs(g - m.log_likelihood(ds)) <= 1e-9 * max(1., abs(want)), overrides
    # not positive definite: duplicate points without noise
    xd = np.concatenate([x[:10], x[:10]])
    bad = ab.gp_from_covariance(ab.SquaredExponential(1.0, 1.0), context=ctx)
    out = bad.log_likelihoods(ab.RegressionDataset(xd, np.zeros(20)), [{}, {"sigma_squared_exponential": 2.0}])
    assert np.all(np.isnan(out))


def test_batched_log_likelihoods_many_parameter_sets(ctx):
    """More parameter vectors in one batch than the context's kernel cache holds (64): every slot must still be
    evaluated with ITS parameters (the batch owns its kernel handles; the cache evicts one entry at a time)."""
    n, count = 96, 75
    rng = np.random.default_rng(77)
    x = rng.uniform(0., 10., (n, 3))
    y = np.sin(x).sum(axis=1)
    cov = ab.SquaredExponential(1.5, 1.0) + ab.IndependentNoise(0.2)
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(x, y)
    sets = [{"squared_exponential_length_scale": 0.8 + 0.03 * i, "sigma_independent_noise": 0.1 + 0.01 * (i % 7)}
            for i in range(count)]
    got = model.log_likelihoods(ds, sets)
    for overrides, g in zip(sets, got):
        m = _with_overrides(ctx, cov, model.mean_function_, overrides)
        assert abs(g + orc.nll(m.covariance_function_, x, y)) <= 1e-8 * n, overrides
    # the cache itself: handles handed out stay valid while more than 64 distinct kernels pass through it
    handles = {}
    for i in range(count):
        m = _with_overrides(ctx, cov, model.mean_function_, sets[i])
        handles[i] = ctx.kernel(m.covariance_function_).value
        assert len(ctx._kernels) <= ctx.KERNEL_CACHE
    m_last = _with_overrides(ctx, cov, model.mean_function_, sets[count - 1])
    assert ctx.kernel(m_last.covariance_function_).value == handles[count - 1]  # most recent entry still cached
    assert abs(m_last.log_likelihood(ds) + orc.nll(m_last.covariance_function_, x, y)) <= 1e-8 * n


@pytest.mark.parametrize("seed", range(12))
def test_random_models_fit_predict_match_oracle(ctx, seed):
    """End-to-end parity sweep: a random composed covariance function (+ noise so that it is positive
    definite), random target variances, fit / log-likelihood / all three prediction types vs the oracle's
    pivoted LDL^T, plain and with measurement noise."""
    from test_gram_gpu import _random_tree
    rng = np.random.default_rng(5000 + seed)
    dim = int(rng.integers(1, 4))
    cov = _random_tree(rng, 2, dim) + ab.measurement_only(ab.IndependentNoise(0.3)) + ab.SquaredExponential(1.5, 1.0)
    n, m = int(rng.integers(150, 400)), 45
    x = rng.uniform(0.5, 5.0, (n, dim)) if dim > 1 else rng.uniform(0.5, 5.0, n)
    xs = rng.uniform(0.5, 5.0, (m, dim)) if dim > 1 else rng.uniform(0.5, 5.0, m)
    y = rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.1, n) if seed % 2 else None
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar))
    fm = model.fit(ds)
    ofit = orc.OracleFit(cov, x, y, yvar)
    info = ofit.information
    assert np.abs(fm.get_fit().information - info).max() <= 1e-8 * np.abs(info).max(), cov.get_name()
    assert abs(model.log_likelihood(ds) + orc.nll(cov, x, y)) <= 1e-8 * n
    for meas in (False, True):
        om, ov = ofit.predict_marginal(xs, xs_meas=meas)
        _, oj = ofit.predict_joint(xs, xs_meas=meas)
        pred = fm.predict_with_measurement_noise(xs) if meas else fm.predict(xs)
        scale = max(1., np.abs(om).max())
        assert np.abs(pred.mean() - om).max() <= 1e-8 * scale
        mg, jt = pred.marginal(), pred.joint()
        vs = max(np.abs(ov).max(), 1e-3)
        assert np.abs(mg.covariance - ov).max() <= 1e-8 * vs and np.abs(jt.covariance - oj).max() <= 1e-8 * np.abs(oj).max()


def test_semi_definite_model_falls_back_to_the_pivoted_factor(ctx):
    """An "unobservable" model (tests/test_gp.cc:20-33 style): duplicated observations without noise make the
    Gram matrix singular.  The reference's pivoted LDL^T goes through; so does the mirror (device Gram +
    agp_ldlt_*), with the oracle's predictions."""
    rng = np.random.default_rng(8)
    x = rng.uniform(0., 10., 40)
    x = np.concatenate([x, x[:7]])  # exact duplicates
    f = np.sin(x)
    cov = ab.SquaredExponential(2.0, 1.0) + ab.Constant(0.5)
    model = ab.gp_from_covariance(cov, context=ctx)
    fm = model.fit(ab.RegressionDataset(x, f))
    assert isinstance(fm.get_fit().train_covariance, ab.PivotedLDLT)
    ofit = orc.OracleFit(cov, x, f)
    # the information vector of a singular system is rounding noise amplified by 1e6 (the oracle's Gram differs
    # from the device's by 1 ulp): only what the data determine is comparable - the predictions
    assert np.all(np.isfinite(fm.get_fit().information))
    xs = np.linspace(0.5, 9.5, 21)
    om, ov = ofit.predict_marginal(xs)
    pm = fm.predict(xs).marginal()
    assert np.abs(pm.mean - om).max() <= 1e-4 * max(1., np.abs(om).max())
    assert np.abs(pm.covariance - ov).max() <= 1e-6
    # and on the SAME Gram matrix the device factor is the oracle's, bit for bit
    K = ctx.gram(cov, ab.Measurement(x))
    packed, tr, ok = orc.ldlt(K)
    ldlt = fm.get_fit().train_covariance
    assert np.array_equal(ldlt.transpositions(), tr) and np.array_equal(np.tril(ldlt.matrix_ldlt()), np.tril(packed))


@pytest.mark.parametrize("n", [2048, 2560])
def test_ill_conditioned_fit_wide_backward_path(ctx, n):
    """N a multiple of 512 takes the 512-row backward substitution (explicit inverses of the 512 x 512 diagonal
    blocks): an ill-conditioned covariance (cond ~ 1e8) must still solve K a = y to fp64 working accuracy."""
    x, y = synthetic_3d(n, 91)
    cov = ab.SquaredExponential(2.0, 1.0) + ab.IndependentNoise(3e-3)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    a = fm.get_fit().information
    K = ctx.gram(cov, ab.Measurement(x))
    assert np.abs(K @ a - y).max() <= 1e-9 * np.abs(K).sum(axis=1).max() * np.abs(a).max()
    ofit = orc.OracleFit(cov, ab.FeatureSet(x), y, use_llt=True)
    assert rel(a, ofit.information) <= 1e-6   # cond(K) * eps ~ 1e-8 for either path


def test_single_point_prediction_uses_the_vector_chain(ctx):
    """M = 1 and N >= 1024: the forward substitution runs as one fused launch per 128 rows (agp_predict_*)"""
    n = 1500
    x, y = synthetic_3d(n, 17)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    ofit = orc.OracleFit(cov, ab.FeatureSet(x), y)
    xs = np.array([[3.3, 4.4, 5.5]])
    om, ov = ofit.predict_marginal(ab.FeatureSet(xs))
    pm = fm.predict(xs).marginal()
    pj = fm.predict(xs).joint()
    assert abs(pm.mean[0] - om[0]) <= 1e-8 * abs(om[0]) and abs(pm.covariance[0] - ov[0]) <= 1e-8 * abs(ov[0])
    assert abs(pj.covariance[0, 0] - ov[0]) <= 1e-8 * abs(ov[0])


def test_marginal_prediction_in_slices_matches_one_pass(make_ctx, monkeypatch):
    """Marginal predictions pass over the test points in slices that bound the n x M workspace (2 GiB by default;
    AGP_PREDICT_CHUNK forces the slice): same numbers as one pass, for the dense and the sparse model, with a scaling
    term in the covariance (the sub-views keep the stride of the scale columns)."""
    rng = np.random.default_rng(21)
    n, M = 700, 2500
    x = rng.uniform(0.5, 5., (n, 2))
    y = np.sin(x).sum(axis=1) + 0.1 * rng.standard_normal(n)
    xs = rng.uniform(0.5, 5., (M, 2))
    class Ramp(ab.ScalingFunction):
        _params = {}

        def get_name(self):
            return "ramp"

        def _call_impl(self, c):
            return 1. + 0.1 * np.asarray(c)[:, 0]

    cov = ab.ScalingTerm(Ramp()) * ab.Constant(0.7) + ab.Matern52(1.5, 1.2) + ab.measurement_only(ab.IndependentNoise(0.2))
    results = []
    for chunk in ("0", "1000"):  # one pass; three slices, the last one partial
        monkeypatch.setenv("AGP_PREDICT_CHUNK", chunk)
        ctx = make_ctx()  # (the switch is read when the context is created)
        model = ab.gp_from_covariance(cov, context=ctx)
        fm = model.fit(ab.RegressionDataset(x, y))
        sparse = ab.sparse_gp_from_covariance(cov, lambda f: int(f[0] // 1.0), ab.FixedInducingPoints(x[:60]), "s", context=ctx)
        sparse.set_param("inducing_nugget", 1e-6)
        sfm = sparse.fit(ab.RegressionDataset(x, y))
        dense = fm.predict_with_measurement_noise(xs).marginal()
        sm = sfm.predict_with_measurement_noise(xs).marginal()
        results.append((dense.mean, dense.covariance, sm.mean, sm.covariance))
    for a, b in zip(results[0], results[1]):
        assert np.abs(a - b).max() <= 1e-12 * max(1., np.abs(a).max())
    om, ov = orc.OracleFit(cov, x, y).predict_marginal(xs, xs_meas=True)
    assert np.abs(results[1][0] - om).max() <= 1e-9 * np.abs(om).max()
    assert np.abs(results[1][1] - ov).max() <= 1e-9 * np.abs(ov).max()
