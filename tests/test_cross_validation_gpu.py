"""GPU tests of leave-one-GROUP-out cross validation: SerializableLDLT::inverse_blocks,
details::held_out_predictions and model.cross_validate(), restating
tests/test_serializable_ldlt.cc:40-66, tests/test_cross_validation.cc:56-72,156-321 through the C-ABI."""
import numpy as np
import pytest

import albatross_amd as ab
from conftest import synthetic_3d
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def make_case(n, seed, with_variance=True):
    rng = np.random.default_rng(seed)
    x = rng.uniform(0., 10., (n, 2))
    y = np.sin(x).sum(axis=1) + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.05, n) if with_variance else None
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    return x, y, yvar, cov


def random_groups(n, k, seed):
    """k ragged groups covering a random subset of the indices, in shuffled order."""
    rng = np.random.default_rng(seed)
    perm = rng.permutation(n)[: max(k, int(0.8 * n))]
    cuts = np.sort(rng.choice(np.arange(1, len(perm)), size=k - 1, replace=False)) if k > 1 else []
    return [list(map(int, g)) for g in np.split(perm, cuts)]


@pytest.mark.parametrize("n,k", [(40, 5), (300, 7), (700, 3)])
def test_inverse_blocks_match_oracle(ctx, n, k):
    x, y, yvar, cov = make_case(n, n)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    ofit = orc.OracleFit(cov, x, y, yvar)
    groups = random_groups(n, k, n + 1) + [[0], [n - 1, 0]]
    got = fm.get_fit().inverse_blocks(groups)
    want = ofit.inverse_blocks(groups)
    scale = max(np.abs(w).max() for w in want)
    for a, b in zip(got, want):
        assert a.shape == b.shape and np.abs(a - b).max() <= 1e-9 * scale
    # single-index blocks == inverse_diagonal (serializable_ldlt.hpp:181-199)
    d = fm.get_fit().inverse_diagonal()
    singles = fm.get_fit().inverse_blocks([[i] for i in (0, 5, n - 1)])
    assert np.allclose([s[0, 0] for s in singles], d[[0, 5, n - 1]], rtol=1e-10)


@pytest.mark.parametrize("n,k", [(40, 5), (300, 7), (700, 3)])
def test_held_out_predictions_match_oracle(ctx, n, k):
    x, y, yvar, cov = make_case(n, 3 * n)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    ofit = orc.OracleFit(cov, x, y, yvar)
    groups = random_groups(n, k, n + 2)
    want = ofit.held_out(y, groups, joint=True)
    marg = fm.get_fit().held_out_predictions(y, groups)
    joint = fm.get_fit().held_out_predictions(y, groups, joint=True)
    for (wm, wv, wj), m, j in zip(want, marg, joint):
        assert np.abs(m.mean - wm).max() <= 1e-8 * max(1., np.abs(wm).max())
        assert np.abs(m.covariance - wv).max() <= 1e-8 * wv.max()
        assert np.abs(j.mean - wm).max() <= 1e-8 * max(1., np.abs(wm).max())
        assert np.abs(j.covariance - wj).max() <= 1e-8 * np.abs(wj).max()
        assert np.abs(np.diag(j.covariance) - m.covariance).max() <= 1e-10 * wv.max()


def test_cross_validate_equals_brute_force_refits(ctx):
    """tests/test_cross_validation.cc:202-260 (test_leave_one_out_equivalences): the fast path equals
    fitting on the other groups and predicting the held-out one (predictions of the measurements:
    the held-out features are wrapped like the training ones)."""
    n = 120
    x, y, yvar, cov = make_case(n, 11, with_variance=False)
    x1 = x[:, 0].copy()
    cov1 = ab.SquaredExponential(3.0, 2.0) + ab.measurement_only(ab.IndependentNoise(0.2))
    model = ab.gp_from_covariance(cov1, context=ctx)
    ds = ab.RegressionDataset(x1, y)
    grouper = lambda f: int(f // 2.5)  # group_by_interval
    cv = model.cross_validate().predict(ds, grouper)
    fast_m, fast_j = cv.marginals(), cv.joints()
    assert list(fast_m.keys()) == sorted(fast_m.keys()) and len(fast_m) == 4
    for key, idx in cv.indexer_.items():
        idx = np.asarray(idx)
        rest = np.setdiff1d(np.arange(n), idx)
        fm = model.fit(ab.RegressionDataset(x1[rest], y[rest]))
        brute = fm.predict_with_measurement_noise(x1[idx]).joint()
        assert np.abs(fast_m[key].mean - brute.mean).max() <= 1e-7
        assert np.abs(fast_j[key].covariance - brute.covariance).max() <= 1e-7
        assert np.abs(fast_m[key].covariance - np.diag(brute.covariance)).max() <= 1e-7
    # concatenated forms are in dataset order (concatenate_mean_predictions)
    mean = cv.mean()
    for key, idx in cv.indexer_.items():
        assert np.array_equal(mean[np.asarray(idx)], fast_m[key].mean)
    assert cv.marginal().covariance.shape == (n,)


def test_leave_one_out_grouper_equals_loo_fast_path(ctx):
    """tests/test_cross_validation.cc:156-199: LeaveOneOutGrouper through cross_validate() ==
    leave_one_out_conditional."""
    n = 200
    x, y, yvar, cov = make_case(n, 5)
    model = ab.gp_from_covariance(cov, context=ctx)
    ds = ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar))
    m = model.cross_validate().predict(ds, ab.LeaveOneOutGrouper()).marginal()
    loo = model.fit(ds).get_fit().leave_one_out(y)
    assert np.abs(m.mean - loo.mean).max() <= 1e-8
    assert np.abs(m.covariance - loo.covariance).max() <= 1e-8 * loo.covariance.max()
    scores = model.cross_validate().scores(ab.root_mean_square_error, ds, ab.LeaveOneOutGrouper())
    assert scores.shape == (n,) and np.allclose(scores, np.abs(loo.mean - y), atol=1e-8)


def test_group_edge_cases(ctx):
    n = 50
    x, y, yvar, cov = make_case(n, 9)
    fit = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y)).get_fit()
    assert fit.inverse_blocks([]) == []
    assert fit.held_out_predictions(y, []) == []
    out = fit.held_out_predictions(y, [[], [3, 4]])
    assert out[0].mean.shape == (0,) and out[1].mean.shape == (2,)
    with pytest.raises(IndexError):
        fit.inverse_blocks([[0, n]])
    # one group holding everything: B = K^-1, prediction = y - K alpha = prior mean 0 ... = y - y
    everything = fit.held_out_predictions(y, [list(range(n))], joint=True)[0]
    assert np.abs(everything.mean).max() <= 1e-8


def test_leave_one_group_out_config3_property(ctx):
    """N = 16384 (BASELINE config 3), 32 groups of 512: blocks of K^-1 against direct solves of
    K X = E on sampled columns; held-out means stay close to the truth."""
    n = 16384
    x, y = synthetic_3d(n, 44)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    fit = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y)).get_fit()
    perm = np.random.default_rng(1).permutation(n)
    groups = [list(map(int, perm[g * 512:(g + 1) * 512])) for g in range(32)]
    blocks = fit.inverse_blocks(groups[:2])
    g0 = np.asarray(groups[0])
    cols = g0[:4]
    E = np.zeros((n, 4))
    E[cols, np.arange(4)] = 1.
    X = fit.solve(E)
    assert np.abs(blocks[0][:, :4] - X[g0]).max() <= 1e-9 * np.abs(blocks[0]).max()
    preds = fit.held_out_predictions(y, groups)
    err = np.concatenate([p.mean - y[np.asarray(g)] for p, g in zip(preds, groups)])
    assert np.all(np.concatenate([p.covariance for p in preds]) > 0)
    assert np.sqrt(np.mean(err ** 2)) < 0.5


@pytest.mark.parametrize("n,m", [(240, 40), (600, 150), (1024, 128), (900, 300)])
def test_equal_size_groups_take_the_batched_path(ctx, n, m):
    """Groups of one size are processed in lock step (batched launches); same answers as the oracle."""
    x, y, yvar, cov = make_case(n, 7 * n)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    ofit = orc.OracleFit(cov, x, y, yvar)
    perm = np.random.default_rng(n).permutation(n)
    groups = [list(map(int, perm[g * m:(g + 1) * m])) for g in range(n // m)]
    want = ofit.held_out(y, groups, joint=True)
    blocks = fm.get_fit().inverse_blocks(groups)
    wb = ofit.inverse_blocks(groups)
    scale = max(np.abs(b).max() for b in wb)
    for a, b in zip(blocks, wb):
        assert np.abs(a - b).max() <= 1e-9 * scale
    marg = fm.get_fit().held_out_predictions(y, groups)
    joint = fm.get_fit().held_out_predictions(y, groups, joint=True)
    for (wm, wv, wj), mg, jt in zip(want, marg, joint):
        assert np.abs(mg.mean - wm).max() <= 1e-8 * max(1., np.abs(wm).max())
        assert np.abs(mg.covariance - wv).max() <= 1e-8 * wv.max()
        assert np.abs(jt.covariance - wj).max() <= 1e-8 * np.abs(wj).max()
        assert np.abs(jt.mean - wm).max() <= 1e-8 * max(1., np.abs(wm).max())
