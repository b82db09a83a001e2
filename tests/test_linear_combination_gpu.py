"""LinearCombination<X> features (core/linear_combination.hpp, LinearCombinationCaller callers.hpp:321-396;
SURVEY.md section 8a row a4): Gram of the expanded points on the device, contraction with the coefficients on the
host, factorisation and solves on the device.  Parity against the oracle's Gram + the defining double sum, and
the reference's own property tests (tests/test_gp.cc:395-462: a sum / difference constraint observed as a
LinearCombination feature is honoured by the posterior)."""
import numpy as np
import pytest

import albatross_amd as ab
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def _cov():
    return ab.SquaredExponential(1.5, 2.0) + ab.Constant(0.7) + ab.measurement_only(ab.IndependentNoise(0.2))


def _oracle_lc_gram(cov, xs, ys=None, x_meas=False, y_meas=False):
    """the defining sums of LinearCombinationCaller over the oracle's pairwise covariance"""
    def expand(fs):
        pts, owner, coef = [], [], []
        for j, f in enumerate(fs):
            if isinstance(f, ab.LinearCombination):
                for v, a in zip(f.values, f.coefficients):
                    pts.append(v); owner.append(j); coef.append(a)
            else:
                pts.append(np.atleast_1d(np.float64(f))); owner.append(j); coef.append(1.)
        return np.stack(pts), np.array(owner), np.array(coef)
    px, ox, cx = expand(xs)
    py, oy, cy = (px, ox, cx) if ys is None else expand(ys)
    G = orc.gram(cov, px, None if ys is None else py, x_meas=x_meas, y_meas=y_meas if ys is not None else x_meas)
    out = np.zeros((len(xs), len(xs) if ys is None else len(ys)))
    for a in range(G.shape[0]):
        for b in range(G.shape[1]):
            out[ox[a], oy[b]] += cx[a] * cy[b] * G[a, b]
    return out


def test_gram_of_linear_combinations_matches_the_defining_sum(ctx):
    rng = np.random.default_rng(5)
    feats = [0.3, ab.LinearCombination([0.1, 0.9, 2.5], [1., -2., 0.5]), 1.7, ab.LinearCombination([1.7, 0.3]), 4.0]
    other = [ab.LinearCombination([0.2, 0.3], [0.5, 0.5]), 0.9, 3.3]
    cov = _cov()
    for meas in (False, True):
        fx = ab.Measurement(feats) if meas else feats
        K = ctx.gram(cov, fx)
        want = _oracle_lc_gram(cov, feats, x_meas=meas)
        assert np.abs(K - want).max() <= 1e-13 * np.abs(want).max()
        Kc = ctx.gram(cov, fx, other)
        wantc = _oracle_lc_gram(cov, feats, other, x_meas=meas, y_meas=False)
        assert np.abs(Kc - wantc).max() <= 1e-13 * np.abs(wantc).max()
    # an IndependentNoise term counts equal constituents: feature 3 = x(1.7) + x(0.3) shares its noise with
    # features 0 (0.3) and 2 (1.7) when both sides are measurements (noise.hpp:37-43 inside the double sum)
    Km = ctx.gram(cov, ab.Measurement(feats))
    Kp = ctx.gram(cov, feats)
    assert abs((Km - Kp)[3, 0] - 0.2 ** 2) < 1e-13 and abs((Km - Kp)[3, 2] - 0.2 ** 2) < 1e-13
    assert abs((Km - Kp)[3, 3] - 2 * 0.2 ** 2) < 1e-13


def test_fit_and_predict_with_linear_combination_observations(ctx):
    """dense formulas on the oracle's covariance: information, mean, joint covariance, log-likelihood"""
    rng = np.random.default_rng(2)
    x = np.sort(rng.uniform(0., 6., 40))
    feats = list(x[:30]) + [ab.LinearCombination(x[30:35], rng.standard_normal(5)),
                            ab.LinearCombination(x[35:40]), ab.LinearCombination([x[0], x[1]], [1., -1.])]
    y = rng.standard_normal(len(feats))
    yv = np.full(len(feats), 0.01)
    cov = _cov()
    model = ab.gp_from_covariance_and_mean(cov, ab.LinearMean(0.3, -1.0), context=ctx)
    ds = ab.RegressionDataset(feats, ab.MarginalDistribution(y, yv))
    fm = model.fit(ds)
    K = _oracle_lc_gram(cov, feats, x_meas=True) + np.diag(yv)
    mean_at = lambda fs: np.array([(f.coefficients @ (0.3 * np.concatenate(f.values) - 1.0)) if isinstance(f, ab.LinearCombination)
                                   else 0.3 * f - 1.0 for f in fs])
    dev = y - mean_at(feats)
    info = np.linalg.solve(K, dev)
    assert np.abs(fm.get_fit().information - info).max() <= 1e-9 * np.abs(info).max()
    tests = [0.5, 2.2, ab.LinearCombination([1.0, 5.0], [0.5, 0.5]), 5.9]
    cross = _oracle_lc_gram(cov, feats, tests, x_meas=True, y_meas=False)
    prior = _oracle_lc_gram(cov, tests)
    want_mean = cross.T @ info + mean_at(tests)
    want_cov = prior - cross.T @ np.linalg.solve(K, cross)
    joint = fm.predict(tests).joint()
    assert np.abs(joint.mean - want_mean).max() <= 1e-9 * np.abs(want_mean).max()
    assert np.abs(joint.covariance - want_cov).max() <= 1e-9 * np.abs(prior).max()
    marg = fm.predict(tests).marginal()
    assert np.abs(marg.covariance - np.diag(want_cov)).max() <= 1e-9 * np.abs(prior).max()
    assert np.abs(fm.predict(tests).mean() - want_mean).max() <= 1e-9 * np.abs(want_mean).max()
    # log_likelihood (gp.hpp:442-451) uses covariance_function_(measurement_features) alone - no target variance - so it
    # is checked on the observations whose covariance is non-singular by itself (the last feature is an exact
    # combination of two directly observed points and shares their measurement noise)
    K0 = _oracle_lc_gram(cov, feats[:-1], x_meas=True)
    dev0 = dev[:-1]
    sign, logdet = np.linalg.slogdet(K0)
    want_ll = -0.5 * (logdet + dev0 @ np.linalg.solve(K0, dev0) + len(dev0) * np.log(2 * np.pi))
    ds0 = ab.RegressionDataset(feats[:-1], ab.MarginalDistribution(y[:-1], yv[:-1]))
    assert abs(model.log_likelihood(ds0) - want_ll) <= 1e-9 * abs(want_ll)


@pytest.mark.parametrize("coefs", [None, [1., -1.]])
def test_constraint_observed_as_linear_combination_is_honoured(ctx, coefs):
    """tests/test_gp.cc:395-462: observe `sum_i c_i f(x_i) = 0` with variance 1e-5 next to the data; the
    posterior at those points then satisfies the constraint."""
    rng = np.random.default_rng(9)
    x = np.linspace(0., 5., 25)
    y = np.sin(x) + 1.5 + 0.05 * rng.standard_normal(25)
    pts = [0.7, 2.9] if coefs is not None else [0.4, 1.9, 3.3, 4.6]
    constraint = ab.LinearCombination(pts, coefs)
    feats = list(x) + [constraint]
    targets = ab.MarginalDistribution(np.concatenate([y, [0.]]), np.concatenate([np.full(25, 0.05 ** 2), [1e-5]]))
    model = ab.gp_from_covariance(ab.SquaredExponential(1.2, 2.0) + ab.Constant(3.0), context=ctx)
    fm = model.fit(ab.RegressionDataset(feats, targets))
    pred = fm.predict(pts).joint()
    c = np.ones(len(pts)) if coefs is None else np.array(coefs)
    # (the reference checks 1e-6 on a model whose data cannot contradict the constraint; here 25 noisy
    # observations pull against it, so the residual is of the order of sqrt(1e-5) * a few)
    assert abs(c @ pred.mean) <= 0.05
    assert abs(c @ pred.covariance @ c) <= 2e-5  # ~ the observation variance of the constraint
    free = model.fit(ab.RegressionDataset(list(x), ab.MarginalDistribution(y, np.full(25, 0.05 ** 2))))
    assert abs(c @ free.predict(pts).joint().mean) > 0.1  # ... which the unconstrained fit does not satisfy
