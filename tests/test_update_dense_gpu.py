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
    again = model.fit_from_prediction(features, pred).predict(features).joint()
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
