"""agp_fit_create_batch (include/albatross_amd.h): B independent fits of one shape in lock step - the Fit<GPFit> constructor
(models/gp.hpp:61-69) for several datasets / parameter vectors at once - against the oracle and against the one-by-one
fits, incl. the blocked batched factorisation at sizes with partial last panels and several outer blocks."""
import numpy as np
import pytest

import albatross_amd as ab
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def _problems(n, count, seed, with_variance):
    rng = np.random.default_rng(seed)
    models_cov, datasets = [], []
    for b in range(count):
        x = rng.uniform(0., 10., (n, 3))
        y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0]) + 0.05 * b
        yvar = rng.uniform(0.01, 0.05, n) if with_variance and b % 2 == 0 else None
        if yvar is not None:
            x[5] = x[2]  # duplicate point: IndependentNoise fires off the diagonal too (the target variance keeps K definite)
        # different parameter vectors AND different trees across the batch
        cov = (ab.Matern52(1.5 + 0.25 * b, 1.0) if b % 3 else ab.SquaredExponential(1.0 + 0.1 * b, 1.2)) + ab.IndependentNoise(0.1 + 0.01 * b)
        models_cov.append(cov)
        datasets.append((x, y, yvar))
    return models_cov, datasets


@pytest.mark.parametrize("n,count,with_variance", [(100, 3, False), (512, 8, True), (700, 5, True), (1300, 4, False), (2048, 3, True)])
def test_fit_batch_matches_oracle_and_single_fits(ctx, n, count, with_variance):
    covs, data = _problems(n, count, n + count, with_variance)
    models = [ab.gp_from_covariance(c, context=ctx) for c in covs]
    datasets = [ab.RegressionDataset(x, y if v is None else ab.MarginalDistribution(y, v)) for x, y, v in data]
    fms = ab.fit_batch(models, datasets)
    assert len(fms) == count
    xs = np.random.default_rng(3).uniform(0., 10., (50, 3))
    for b, (fm, cov, (x, y, v)) in enumerate(zip(fms, covs, data)):
        ofit = orc.OracleFit(cov, x, y, v)
        info = fm.get_fit().information
        assert np.abs(info - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max(), b
        assert abs(fm.get_fit().log_determinant - ofit.log_determinant) <= 1e-6 * n
        single = models[b].fit(datasets[b]).get_fit()
        assert np.abs(info - single.information).max() <= 1e-9 * np.abs(info).max()
        # the handles are ordinary fits: predictions, solve
        om, ov = ofit.predict_marginal(xs)
        marg = fm.predict(xs).marginal()
        assert np.abs(marg.mean - om).max() <= 1e-8 * np.abs(om).max()
        assert np.abs(marg.covariance - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-9
        rhs = np.random.default_rng(b).standard_normal(n)
        assert np.abs(fm.get_fit().solve(rhs) - ofit.solve(rhs)).max() <= 1e-8 * np.abs(ofit.solve(rhs)).max()
    # destroying the fits in any order releases the shared allocation with the last one
    del fms[1]
    del fms


def test_fit_batch_reports_the_failing_problem(ctx):
    n = 300
    covs, data = _problems(n, 3, 11, False)
    covs[1] = ab.SquaredExponential(1., 1.)  # no noise + a duplicated point: singular at pivot 5
    data[1][0][5] = data[1][0][2]
    models = [ab.gp_from_covariance(c, context=ctx) for c in covs]
    datasets = [ab.RegressionDataset(x, y) for x, y, _ in data]
    with pytest.raises(ab.NotPositiveDefiniteError, match="problem 1 .pivot 5."):
        ab.fit_batch(models, datasets)
    xn = data[2][0].copy()
    xn[7, 1] = np.nan
    datasets[1] = ab.RegressionDataset(data[0][0] + 0.5, data[1][1])  # (no duplicate point any more)
    models[1] = ab.gp_from_covariance(ab.Matern52(2., 1.) + ab.IndependentNoise(0.1), context=ctx)
    datasets[2] = ab.RegressionDataset(xn, data[2][1])
    with pytest.raises(ab.NanInputError, match="problem 2"):
        ab.fit_batch(models, datasets)


@pytest.mark.parametrize("n,count,same_tree,with_variance", [(256, 40, True, False), (520, 24, False, True), (1100, 50, True, True)])
def test_fit_batch_large_batches_match_oracle(ctx, n, count, same_tree, with_variance):
    """Round 5: large batches of Fit<GPFit> constructions (gp.hpp:61-69) - uniform trees go through ONE Gram launch for the
    whole batch (csrc/gram.hip: gram_fast_batch_kernel), mixed trees through a launch each, the training features of
    device-resident problems through one table-driven copy launch; sizes with a partial last panel."""
    rng = np.random.default_rng(n + count)
    covs, data = [], []
    for b in range(count):
        x = rng.uniform(0., 10., (n, 3))
        y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0]) + 0.05 * b
        yvar = rng.uniform(0.01, 0.05, n) if with_variance and b % 2 == 0 else None
        if same_tree or b % 3:
            cov = ab.Matern52(1.5 + 0.02 * b, 1.0) + ab.IndependentNoise(0.1 + 0.002 * b)
        else:
            cov = ab.SquaredExponential(1.0 + 0.01 * b, 1.2) + ab.IndependentNoise(0.15)
        covs.append(cov)
        data.append((x, y, yvar))
    models = [ab.gp_from_covariance(c, context=ctx) for c in covs]
    datasets = [ab.RegressionDataset(x, y if v is None else ab.MarginalDistribution(y, v)) for x, y, v in data]
    fms = ab.fit_batch(models, datasets)
    xs = np.random.default_rng(3).uniform(0., 10., (20, 3))
    check = range(count) if n <= 600 else range(0, count, 7)  # (the oracle's unblocked LDL^T is the slow side)
    for b in check:
        fm, cov, (x, y, v) = fms[b], covs[b], data[b]
        ofit = orc.OracleFit(cov, x, y, v)
        info = fm.get_fit().information
        assert np.abs(info - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max(), b
        assert abs(fm.get_fit().log_determinant - ofit.log_determinant) <= 1e-6 * n, b
        om, ov = ofit.predict_marginal(xs)
        marg = fm.predict(xs).marginal()
        assert np.abs(marg.mean - om).max() <= 1e-8 * np.abs(om).max()
        assert np.abs(marg.covariance - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-9
    # every problem against its own one-at-a-time fit (the step-launch path): same factor up to rounding
    for b in range(count):
        single = models[b].fit(datasets[b]).get_fit()
        info = fms[b].get_fit().information
        assert np.abs(info - single.information).max() <= 1e-9 * np.abs(info).max(), b
        assert abs(fms[b].get_fit().log_determinant - single.log_determinant) <= 1e-9 * n


def test_fit_batch_large_batch_reports_failures(ctx):
    n, count = 200, 12
    rng = np.random.default_rng(5)
    covs = [ab.SquaredExponential(1.2, 1.0) + ab.IndependentNoise(0.1) for _ in range(count)]
    data = [(rng.uniform(0., 10., (n, 3)), rng.standard_normal(n)) for _ in range(count)]
    covs[7] = ab.SquaredExponential(1., 1.)  # no noise + a duplicated point: singular
    data[7][0][5] = data[7][0][2]
    models = [ab.gp_from_covariance(c, context=ctx) for c in covs]
    with pytest.raises(ab.NotPositiveDefiniteError, match="problem 7 .pivot 5."):
        ab.fit_batch(models, [ab.RegressionDataset(x, y) for x, y in data])
