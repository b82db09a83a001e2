"""GPU tests of the rows next to the fit path: dense-matrix factor
(SerializableLDLT(MatrixXd)), dense negative_log_likelihood, BlockSymmetric and
FitModel::update — restating tests/test_evaluate.cc:20-44,
tests/test_serializable_ldlt.cc:34-85, tests/test_block_utils.cc:125-147 and
tests/test_gp.cc:182-219 through the C-ABI."""
import numpy as np
import pytest

import albatross_amd as ab
from conftest import golden
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def spd(n, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n + 3))
    return G @ G.T / n + np.eye(n)


def test_mvn_nll_golden_on_device(ctx):
    g = golden("mvn_nll.json")  # scipy known answer 6.0946974293510134 (tests/test_evaluate.cc:20-44)
    nll = ab.negative_log_likelihood(np.array(g["x"]), np.array(g["cov"]), context=ctx)
    assert abs(nll - g["nll"]) < g["tolerance_build"]
    # univariate shortcut (likelihood.hpp:57-60)
    assert abs(ab.negative_log_likelihood(np.array([0.3]), np.array([[2.0]]), context=ctx)
               - 0.5 * (np.log(2 * np.pi * 2.0) + 0.09 / 2.0)) < 1e-15


@pytest.mark.parametrize("n", [2, 17, 128, 300, 1000])
def test_dense_factor_matches_oracle(ctx, n):
    A = spd(n, n)
    B = np.random.default_rng(n + 1).standard_normal((n, 3))
    f = ab.DenseFactor(A, ctx)
    packed, tr, ok = orc.ldlt(A)
    X = orc.ldlt_solve(packed, tr, B)
    assert np.abs(f.solve(B) - X).max() <= 1e-10 * np.abs(X).max()          # solve equality
    assert abs(f.log_determinant - orc.ldlt_logdet(packed)) <= 1e-8 * n     # logdet, 1e-8
    assert np.abs(f.inverse_diagonal() - np.diag(np.linalg.inv(A))).max() <= 1e-8  # inverse diagonal, 1e-8
    L = f.factor()
    assert np.abs(L @ L.T - A).max() <= 1e-12 * np.abs(A).max() * n
    dev = B[:, 0]
    assert abs(ab.negative_log_likelihood(dev, A, context=ctx) - orc.nll_dense(dev, A)) <= 1e-9 * n


def test_dense_factor_reads_only_the_lower_triangle(ctx):
    A = spd(200, 9)
    want = np.linalg.solve(A, np.ones(200))
    for order in ("F", "C"):  # column-major: uplo = 0; row-major: handed over untransposed, uplo = 1
        M = np.array(np.tril(A) + np.triu(np.full_like(A, 1e9), 1), order=order)  # garbage above the diagonal
        assert np.abs(ab.DenseFactor(M, ctx).solve(np.ones(200)) - want).max() <= 1e-10 * np.abs(want).max()


def test_dense_factor_error_paths(ctx):
    A = spd(50, 1)
    A[30, 30] = -1.
    with pytest.raises(ab.NotPositiveDefiniteError, match="pivot 30"):
        ab.DenseFactor(A, ctx)
    A = spd(50, 2)
    A[40, 3] = np.nan
    with pytest.raises(ab.NanInputError):
        ab.DenseFactor(A, ctx)


def test_block_symmetric_matches_dense(ctx):
    # tests/test_block_utils.cc:125-147
    n, m = 300, 40
    M = spd(n + m, 5)
    A, Bm, Cm = M[:n, :n], M[:n, n:], M[n:, n:]
    fa = ab.DenseFactor(A, ctx)
    S = Cm - Bm.T @ np.linalg.solve(A, Bm)
    bs = ab.BlockSymmetric(fa, Bm, ab.DenseFactor(S, ctx))
    rhs = np.random.default_rng(0).standard_normal((n + m, 4))
    want = np.linalg.solve(M, rhs)
    assert bs.rows() == n + m
    assert np.abs(bs.solve(rhs) - want).max() <= 1e-10 * np.abs(want).max()


def test_device_compositions_nest_and_predict(ctx):
    """agp_solver_* (include/albatross_amd.h): BlockSymmetric over a BlockSymmetric over a pivoted L D L^T, ExplainedCovariance
    over a pivoted factor - solves against numpy - and the generic _predict_impl (agp_solver_predict, gp.hpp:305-366) of a fit
    that was updated twice on a pivoted factor against the oracle's full fit."""
    n, m1, m2 = 120, 30, 17
    M = spd(n + m1 + m2, 9)
    A, B1, C1 = M[:n, :n], M[:n, n:n + m1], M[n:n + m1, n:n + m1]
    fa = ab.PivotedLDLT(A, ctx)
    S1 = C1 - B1.T @ np.linalg.solve(A, B1)
    bs1 = ab.BlockSymmetric(fa, B1, ab.DenseFactor(S1, ctx))
    M1 = M[:n + m1, :n + m1]
    B2, C2 = M[:n + m1, n + m1:], M[n + m1:, n + m1:]
    S2 = C2 - B2.T @ np.linalg.solve(M1, B2)
    bs2 = ab.BlockSymmetric(bs1, B2, ab.PivotedLDLT(S2, ctx))
    rhs = np.random.default_rng(1).standard_normal((n + m1 + m2, 5))
    want = np.linalg.solve(M, rhs)
    assert bs2.rows() == n + m1 + m2
    assert np.abs(bs2.solve(rhs) - want).max() <= 1e-9 * np.abs(want).max()
    assert np.abs(bs2.solve(rhs[:, 0]) - want[:, 0]).max() <= 1e-9 * np.abs(want).max()
    inner = spd(n, 4) - np.eye(n)
    ec = ab.ExplainedCovariance(fa, inner, ctx)
    want = np.linalg.solve(A, inner @ np.linalg.solve(A, rhs[:n]))
    assert np.abs(ec.solve(rhs[:n]) - want).max() <= 1e-9 * np.abs(want).max()
    # a model fitted through the pivoted factor, updated twice (BlockSymmetric on the device), predicted generically
    rng = np.random.default_rng(5)
    x = rng.uniform(0., 10., (260, 2))
    y = np.sin(x).sum(axis=1) + 0.05 * rng.standard_normal(260)
    var = np.full(260, 0.1)
    xs = rng.uniform(0., 10., (40, 2))
    cov = ab.SquaredExponential(1.5, 1.0) + ab.Constant(2.0)
    model = ab.gp_from_covariance(cov, context=ctx)
    first, second, third = slice(0, 150), slice(150, 210), slice(210, 260)
    fm = model._fit_pivoted(ab.RegressionDataset(x[first], ab.MarginalDistribution(y[first], var[first])),
                            cov.features(x[first]), y[first].copy(), var[first])
    fm = fm.update(ab.RegressionDataset(x[second], ab.MarginalDistribution(y[second], var[second])))
    fm = fm.update(ab.RegressionDataset(x[third], ab.MarginalDistribution(y[third], var[third])))
    assert isinstance(fm.get_fit().train_covariance, ab.BlockSymmetric)
    ofit = orc.OracleFit(cov, x, y, var)
    om, ov = ofit.predict_marginal(xs)
    ojm, ojc = ofit.predict_joint(xs)
    assert np.abs(fm.predict(xs).mean() - om).max() <= 1e-8 * np.abs(om).max()
    marg = fm.predict(xs).marginal()
    assert np.abs(marg.covariance - ov).max() <= 1e-7 * np.abs(ov).max() + 1e-9
    joint = fm.predict(xs).joint()
    assert np.abs(joint.covariance - ojc).max() <= 1e-7 * np.abs(ojc).max() + 1e-9
    assert np.abs(fm.get_fit().information - ofit.information).max() <= 1e-7 * np.abs(ofit.information).max()


def test_update_equals_full_fit(ctx):
    # tests/test_gp.cc:182-219: a partial fit followed by update == a full fit
    rng = np.random.default_rng(3)
    n = 400
    x = rng.uniform(0., 10., (n, 2))
    y = np.sin(x).sum(axis=1) + 0.05 * rng.standard_normal(n)
    var = np.full(n, 0.1)
    xs = rng.uniform(0., 10., (25, 2))
    cov = ab.SquaredExponential(1.5, 1.0) + ab.Constant(2.0)   # noise only through the target variance
    model = ab.gp_from_covariance(cov, context=ctx)
    first, second, third = slice(0, 250), slice(250, 330), slice(330, n)
    full = model.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, var)))
    full_pred = full.predict(xs).joint()
    split = model.fit(ab.RegressionDataset(x[first], ab.MarginalDistribution(y[first], var[first])))
    first_pred = split.predict(xs).joint()
    split = split.update(ab.RegressionDataset(x[second], ab.MarginalDistribution(y[second], var[second])))
    split = split.update(ab.RegressionDataset(x[third], ab.MarginalDistribution(y[third], var[third])))  # nested update
    split_pred = split.predict(xs).joint()
    assert np.allclose(split_pred.mean, full_pred.mean, rtol=1e-9, atol=1e-10)
    assert np.linalg.norm(split_pred.covariance - full_pred.covariance) <= 1e-6
    assert np.linalg.norm(split_pred.mean - first_pred.mean) > 1e-3  # and it is not the partial fit
    marg = split.predict(xs).marginal()
    assert np.abs(marg.covariance - np.diag(full_pred.covariance)).max() <= 1e-8
    assert np.abs(split.predict(xs).mean() - full_pred.mean).max() <= 1e-9
    # the updated information vector is the full fit's
    assert np.abs(split.get_fit().information - full.get_fit().information).max() \
        <= 1e-8 * np.abs(full.get_fit().information).max()


def test_fit_from_prediction_round_trip(ctx):
    """tests/test_gp.cc:343-371 (test_model_from_prediction_with_mean): the fit built from a joint prediction
    reproduces it (1e-6), mean function included exactly once; and it keeps predicting like the original
    model elsewhere when the prediction points carry the information (tests/test_gp.cc:308-341 in spirit)."""
    g = golden("toy_linear.json")
    x, y = np.array(g["x"]), np.array(g["y"])
    cov = ab.SquaredExponential(2.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))
    model = ab.gp_from_covariance_and_mean(cov, ab.LinearMean(slope=1.0, offset=5.0), context=ctx)
    fit_model = model.fit(ab.RegressionDataset(x, y))
    features = np.array([1.3, 4.2, 7.1])
    pred = fit_model.predict(features).joint()
    from_pred = model.fit_from_prediction(features, pred)
    # a fit whose solver is an ExplainedCovariance lives on the device too (agp_solver_explained / agp_solver_predict): no
    # host arithmetic between the solves
    again = from_pred.predict(features).joint()
    assert np.linalg.norm(again.mean - pred.mean) <= 1e-6
    assert np.linalg.norm(again.covariance - pred.covariance) <= 1e-6
    # ExplainedCovariance::solve = A^-1 B A^-1 (representations.hpp:80-82) against numpy
    A, B = spd(40, 3), spd(40, 4) - np.eye(40)
    rhs = np.random.default_rng(0).standard_normal((40, 3))
    ec = ab.ExplainedCovariance(A, B, ctx)
    want = np.linalg.solve(A, B @ np.linalg.solve(A, rhs))
    assert np.abs(ec.solve(rhs) - want).max() <= 1e-10 * np.abs(want).max()
    # dense inducing set: the rebuilt model agrees with the original at new points
    dense_pts = np.linspace(0., 9., 19)
    rebuilt = model.fit_from_prediction(dense_pts, fit_model.predict(dense_pts).joint())
    xs = np.array([0.7, 3.3, 8.4])
    a, b = fit_model.predict(xs).joint(), rebuilt.predict(xs).joint()
    assert np.abs(a.mean - b.mean).max() <= 1e-5 and np.abs(a.covariance - b.covariance).max() <= 1e-5


@pytest.mark.parametrize("n0,m1,m2", [(250, 80, 70), (256, 128, 40), (1000, 300, 1), (130, 1, 129)])
def test_device_update_matches_oracle_update(ctx, n0, m1, m2):
    """agp_fit_update (the resident factor grows by a block row) against the oracle's literal _update_impl +
    BlockSymmetric restatement (gp.hpp:384-414): information, all three predictions, solve, log-determinant; nested;
    sizes that are / are not multiples of 128 (phantom rows); a covariance with measurement-only noise AND
    IndependentNoise, where update differs from a full fit, so that the reference's exact semantics are pinned."""
    rng = np.random.default_rng(n0 + m1)
    n = n0 + m1 + m2
    x = rng.uniform(0., 10., (n, 3))
    x[n0 + 1] = x[3]  # a new observation AT an old point: IndependentNoise correlates them (plain features, by value)
    y = np.sin(x).sum(axis=1) + 0.05 * rng.standard_normal(n)
    var = rng.uniform(0.05, 0.1, n)
    xs = rng.uniform(0., 10., (33, 3))
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.2) + ab.measurement_only(ab.IndependentNoise(0.3))
    model = ab.gp_from_covariance_and_mean(cov, ab.LinearMean(0.2, -0.5), context=ctx)
    a, b, c = slice(0, n0), slice(n0, n0 + m1), slice(n0 + m1, n)
    fm = model.fit(ab.RegressionDataset(x[a], ab.MarginalDistribution(y[a], var[a])))
    of = orc.OracleFit(cov, x[a], y[a], var[a], mean=model.mean_function_)
    for sl in (b, c):
        fm = fm.update(ab.RegressionDataset(x[sl], ab.MarginalDistribution(y[sl], var[sl])))
        of = of.update(x[sl], y[sl], var[sl])
        assert isinstance(fm.get_fit(), ab.GPFit)  # still a device factor, not a host composition
        info = of.information
        assert fm.get_fit().rows() == of.n
        assert np.abs(fm.get_fit().information - info).max() <= 1e-8 * np.abs(info).max()
        om, ov = of.predict_marginal(xs)
        _, oj = of.predict_joint(xs)
        pred = fm.predict(xs)
        assert np.abs(pred.mean() - om).max() <= 1e-8 * np.abs(om).max()
        marg, joint = pred.marginal(), pred.joint()
        assert np.abs(marg.covariance - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-10
        assert np.abs(joint.covariance - oj).max() <= 1e-8 * np.abs(oj).max() + 1e-10
        rhs = rng.standard_normal((of.n, 3))
        assert np.abs(fm.get_fit().solve(rhs) - of.solve(rhs)).max() <= 1e-8 * np.abs(of.solve(rhs)).max()
        assert np.abs(fm.get_fit().solve(rhs[:, 0]) - of.solve(rhs[:, 0])).max() <= 1e-8 * np.abs(of.solve(rhs[:, 0])).max()
    # the grown factor really is the LL^T of the block matrix the reference's BlockSymmetric inverts
    L = fm.get_fit().factor()
    K = orc.gram(cov, x[a], x_meas=True) + np.diag(var[a])
    M = np.block([[K, orc.gram(cov, x[a], x[n0:])], [orc.gram(cov, x[n0:], x[a]), orc.gram(cov, x[n0:]) + np.diag(var[n0:])]])
    assert np.abs(L @ L.T - M).max() <= 1e-10 * np.abs(M).max()
    assert abs(fm.get_fit().log_determinant - np.linalg.slogdet(M)[1]) <= 1e-8 * n
    if n0 % 128 or (n0 + m1) % 128:  # the grown factor carries phantom rows: its cross-validation entry points decline
        with pytest.raises(ab.AlbatrossAmdError):
            fm.get_fit().leave_one_out(y)


def test_device_update_reports_not_positive_definite(ctx):
    rng = np.random.default_rng(1)
    x = rng.uniform(0., 10., (200, 2))
    y = np.sin(x).sum(axis=1)
    model = ab.gp_from_covariance(ab.SquaredExponential(1.5, 1.0) + ab.IndependentNoise(0.1), context=ctx)
    fm = model.fit(ab.RegressionDataset(x, y))
    xn = rng.uniform(0., 10., (20, 2))
    xn[7] = xn[2]  # a duplicate among the NEW points whose (nonsensical) negative variance makes the Schur complement
    bad_var = np.zeros(20)  # indefinite at exactly that pivot: reported with its index in the grown fit
    bad_var[7] = -1e-3
    cov_nn = ab.SquaredExponential(1.5, 1.0)
    fm2 = ab.gp_from_covariance(cov_nn, context=ctx)
    fit0 = fm2.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, np.full(200, 0.1))))
    with pytest.raises(ab.NotPositiveDefiniteError, match="pivot 207"):
        fit0.update(ab.RegressionDataset(xn, ab.MarginalDistribution(np.zeros(20), bad_var)))
    assert fm.update(ab.RegressionDataset(xn[:5], np.zeros(5))).get_fit().rows() == 205
