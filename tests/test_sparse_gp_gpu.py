"""GPU tests of the sparse Gaussian process: SparseGaussianProcessRegression
fit / predict / log_likelihood (include/albatross/src/models/sparse_gp.hpp) through the C-ABI,
against the CPU oracle's restatement (pivoted Householder QR, block LDLT) and restating
tests/test_sparse_gp.cc:48-133 (test_sanity) and :172-218 (test_likelihood)."""
import numpy as np
import pytest

import albatross_amd as ab
from conftest import golden
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def interval_grouper(width):
    return lambda f: int(np.floor(np.atleast_1d(f)[0] / width))


def toy_linear():
    g = golden("toy_linear.json")  # make_toy_linear_data(), tests/lib/albatross/test/test_utils.h:42-60
    return np.array(g["x"]), np.array(g["y"])


def simple_cov(length_scale):
    # make_simple_covariance_function(): SE(100, 100) + measurement_only(IndependentNoise(0.1)), test_models.h:26-30
    return ab.SquaredExponential(length_scale, 100.0) + ab.measurement_only(ab.IndependentNoise(0.1))


@pytest.mark.parametrize("n,m,width,dim", [(60, 7, 5.0, 1), (400, 40, 2.5, 1), (900, 130, 1.7, 3)])
def test_sparse_fit_and_predict_match_oracle(ctx, n, m, width, dim):
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 20., (n, dim)) if dim > 1 else rng.uniform(0., 20., n)
    col = x[:, 0] if dim > 1 else x
    y = np.sin(col) + 0.3 * col + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.04, n)
    cov = ab.Matern52(4.0, 2.0) + ab.measurement_only(ab.IndependentNoise(0.2))
    u = rng.uniform(0., 20., (m, dim)) if dim > 1 else np.linspace(0., 20., m)
    grouper = interval_grouper(width)
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    model.set_param("measurement_nugget", 1e-10)
    ds = ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar))
    fm = model.fit(ds)
    keys = np.array([grouper(f) for f in x])
    ofit = orc.OracleSparseFit(cov, x, keys, y, yvar, u, 1e-10, 1e-6)
    v = ofit.information
    assert np.abs(fm.get_fit().information - v).max() <= 1e-7 * np.abs(v).max()
    assert abs(fm.get_fit().nll - ofit.nll) <= 1e-8 * n
    assert abs(model.log_likelihood(ds) + ofit.nll) <= 1e-8 * n
    xs = rng.uniform(0., 20., (37, dim)) if dim > 1 else np.linspace(0.01, 19.9, 37)
    om, ov, oj = ofit.predict(xs, xs_meas=True, joint=True)
    scale = max(1., np.abs(om).max())
    pred = fm.predict_with_measurement_noise(xs)
    assert np.abs(pred.mean() - om).max() <= 1e-8 * scale
    marg, joint = pred.marginal(), pred.joint()
    assert np.abs(marg.mean - om).max() <= 1e-8 * scale and np.abs(marg.covariance - ov).max() <= 1e-8 * ov.max()
    assert np.abs(joint.covariance - oj).max() <= 1e-8 * np.abs(oj).max()
    assert np.abs(joint.covariance - joint.covariance.T).max() == 0.
    # latent prediction (no measurement wrapper): the measurement-only noise drops out of K_**
    lm, lv = ofit.predict(xs, xs_meas=False)
    lat = fm.predict(xs).marginal()
    assert np.abs(lat.covariance - lv).max() <= 1e-8 * ov.max() and np.abs(lat.mean - lm).max() <= 1e-8 * scale


@pytest.mark.parametrize("n,gs,m", [(512, 128, 30), (900, 300, 64), (1200, 150, 100)])
def test_uniform_groups_batched_path_matches_oracle(ctx, n, gs, m):
    """Equal group sizes take the batched (lock-step) block path; ragged ones the per-block path above."""
    rng = np.random.default_rng(n + gs)
    x = np.sort(rng.uniform(0., 30., n))
    y = np.sin(x) + 0.2 * x + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.04, n)
    cov = ab.SquaredExponential(2.5, 1.5) + ab.measurement_only(ab.IndependentNoise(0.2))
    u = np.linspace(0., 30., m)
    rank = {float(v): i for i, v in enumerate(x)}
    grouper = lambda f: rank[float(f)] // gs
    perm = rng.permutation(n)  # the caller's order is arbitrary: the mirror regroups
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    fm = model.fit(ab.RegressionDataset(x[perm], ab.MarginalDistribution(y[perm], yvar[perm])))
    keys = np.array([grouper(f) for f in x])
    ofit = orc.OracleSparseFit(cov, x, keys, y, yvar, u, 1e-8, 1e-6)
    v = ofit.information
    assert np.abs(fm.get_fit().information - v).max() <= 1e-7 * np.abs(v).max()
    assert abs(fm.get_fit().nll - ofit.nll) <= 1e-8 * n
    xs = np.linspace(0.5, 29.5, 40)
    om, ov, oj = ofit.predict(xs, xs_meas=True, joint=True)
    j = fm.predict_with_measurement_noise(xs).joint()
    assert np.abs(j.mean - om).max() <= 1e-8 * max(1., np.abs(om).max())
    assert np.abs(j.covariance - oj).max() <= 1e-8 * np.abs(oj).max()


@pytest.mark.parametrize("dim,cov_name", [(2, "matern52"), (3, "se"), (3, "matern32")])
def test_uniform_groups_gram_blocks_in_one_launch(ctx, dim, cov_name):
    """Equal groups + a radial fast-path kernel: the K_gg blocks of all groups come from ONE launch (launch_gram_blocks,
    csrc/gram.hip, blockIdx.z = group) - 2-D / 3-D features, target variances, measurement-only noise."""
    n, gs, m = 768, 128, 48
    rng = np.random.default_rng(dim * 10 + len(cov_name))
    x = rng.uniform(0., 12., (n, dim))
    x = x[np.argsort(x[:, 0])]
    y = np.sin(x[:, 0]) + 0.1 * x.sum(axis=1) + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.04, n)
    radial = {"matern52": ab.Matern52(3.0, 1.5), "se": ab.SquaredExponential(2.5, 1.5), "matern32": ab.Matern32(3.0, 1.2)}[cov_name]
    cov = radial + ab.measurement_only(ab.IndependentNoise(0.2))
    u = rng.uniform(0., 12., (m, dim))
    index = {tuple(v): i for i, v in enumerate(x)}
    grouper = lambda f: index[tuple(np.atleast_1d(f))] // gs
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    fm = model.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    ofit = orc.OracleSparseFit(cov, x, np.arange(n) // gs, y, yvar, u, 1e-8, 1e-6)
    v = ofit.information
    assert np.abs(fm.get_fit().information - v).max() <= 1e-7 * np.abs(v).max()
    assert abs(fm.get_fit().nll - ofit.nll) <= 1e-8 * n
    xs = rng.uniform(0., 12., (30, dim))
    om, ov, oj = ofit.predict(xs, xs_meas=True, joint=True)
    j = fm.predict_with_measurement_noise(xs).joint()
    assert np.abs(j.mean - om).max() <= 1e-8 * max(1., np.abs(om).max())
    assert np.abs(j.covariance - oj).max() <= 1e-8 * np.abs(oj).max()


def test_wide_substitution_path_matches_oracle(ctx):
    """Many more observations than inducing points (n >= 8 m, m a multiple of 512): P = L_u^-1 K_uf and Q1 = L1^-1 W go
    through the out-of-place substitution on explicitly inverted 512 x 512 diagonal blocks (forward_solve_wide,
    csrc/solve.hip) instead of the in-place 128-row chain."""
    n, m, gs = 8192, 1024, 256
    rng = np.random.default_rng(1)
    x = np.sort(rng.uniform(0., n / 16., n))
    y = np.sin(x) + 0.1 * rng.standard_normal(n)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))
    u = np.linspace(x.min(), x.max(), m)
    rank = {float(v): i for i, v in enumerate(x)}
    grouper = lambda f: rank[float(f)] // gs
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    fm = model.fit(ab.RegressionDataset(x, y))
    ofit = orc.OracleSparseFit(cov, x, np.arange(n) // gs, y, None, u, 1e-8, 1e-6)
    v = ofit.information
    assert np.abs(fm.get_fit().information - v).max() <= 1e-7 * np.abs(v).max()
    assert abs(fm.get_fit().nll - ofit.nll) <= 1e-8 * n
    xs = np.linspace(x.min() + 0.3, x.max() - 0.3, 50)
    om, ov, oj = ofit.predict(xs, xs_meas=True, joint=True)
    j = fm.predict_with_measurement_noise(xs).joint()
    assert np.abs(j.mean - om).max() <= 1e-8 * max(1., np.abs(om).max())
    assert np.abs(j.covariance - oj).max() <= 1e-8 * np.abs(oj).max()


@pytest.mark.parametrize("length_scale,sparse_thr,really_sparse_thr", [(1000., 1e-2, 0.5), (100., 1e-2, 0.5),
                                                                        (10., 5e-2, 100.)])
def test_sanity_against_direct_gp(ctx, length_scale, sparse_thr, really_sparse_thr):
    """tests/test_sparse_gp.cc:48-133: 8 uniformly spaced inducing points track the direct GP, 3 do worse."""
    x, y = toy_linear()
    cov = simple_cov(length_scale)
    ds = ab.RegressionDataset(x, y)
    grouper = interval_grouper(5.0)  # LeaveOneIntervalOut / get_group, :22-29
    direct = ab.gp_from_covariance(cov, context=ctx).fit(ds)
    xs = np.linspace(0.01, 9.9, 11)
    dp = direct.predict_with_measurement_noise(xs).joint()
    errs = []
    for num in (8, 3):
        sp = ab.sparse_gp_from_covariance(cov, grouper, ab.UniformlySpacedInducingPoints(num), "sparse", context=ctx)
        sp.set_param_value("inducing_nugget", 1e-3)
        sp.set_param_value("measurement_nugget", 1e-12)
        p = sp.fit(ds).predict_with_measurement_noise(xs).joint()
        errs.append((np.linalg.norm(p.mean - dp.mean), np.linalg.norm(p.covariance - dp.covariance)))
    (sparse_err, sparse_cov), (really_err, really_cov) = errs
    assert sparse_err < sparse_thr and really_err < really_sparse_thr
    assert really_err > sparse_err - 1e-4
    assert sparse_cov < sparse_thr and really_cov < really_sparse_thr and really_cov > sparse_cov


def test_likelihood_equals_dense_equivalent(ctx):
    """tests/test_sparse_gp.cc:172-218: the sparse log likelihood is the dense one of K = Q_ff with the
    group blocks replaced by K_ff (+ noise + nuggets); reference tolerance 1e-2 absolute, here 1e-9
    relative (the value is ~ -6.8e3 and the 2 x 2 K_uu has a condition number of 1e12)."""
    n = 12
    x = np.arange(n, dtype=float)
    rng = np.random.default_rng(3)
    y = 5. * np.sin(x * 10.) + 0.1 * rng.standard_normal(n)  # make_toy_sine_data(5, 10, 0.1, 12) shape
    cov = simple_cov(100.)
    grouper = interval_grouper(5.0)
    strategy = ab.UniformlySpacedInducingPoints(2)
    sparse = ab.sparse_gp_from_covariance(cov, grouper, strategy, "sparse", context=ctx)
    u = strategy(cov, x)
    K_uu = ctx.gram(cov, u) + sparse.get_params()["inducing_nugget"] * np.eye(2)
    K_fu = ctx.gram(cov, ab.Measurement(x), u)
    K = K_fu @ np.linalg.solve(K_uu, K_fu.T)
    K_ff = ctx.gram(cov, ab.Measurement(x))
    keys = np.array([grouper(f) for f in x])
    for key in np.unique(keys):
        idx = np.nonzero(keys == key)[0]
        K[np.ix_(idx, idx)] = K_ff[np.ix_(idx, idx)]
    K[np.diag_indices(n)] += sparse.get_params()["measurement_nugget"]
    expected = -ab.negative_log_likelihood(y, K, context=ctx)
    assert abs(sparse.log_likelihood(ab.RegressionDataset(x, y)) - expected) <= 1e-9 * abs(expected)


def test_inducing_points_on_the_data_recover_the_dense_gp(ctx):
    """Property at a size the oracle does not reach: with u = the training points the FITC/PITC
    posterior is the exact GP posterior (Q_ff = K_ff up to the nuggets)."""
    n = 3000
    rng = np.random.default_rng(0)
    x = rng.uniform(0., 10., (n, 3))
    y = np.sin(x).sum(axis=1) + 0.1 * rng.standard_normal(n)
    cov = ab.SquaredExponential(2.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.3))
    ds = ab.RegressionDataset(x, y)
    dense = ab.gp_from_covariance(cov, context=ctx).fit(ds)
    sparse = ab.sparse_gp_from_covariance(cov, interval_grouper(1.0), ab.FixedInducingPoints(x), "sparse", context=ctx)
    sparse.set_param("inducing_nugget", 1e-9)
    sparse.set_param("measurement_nugget", 1e-9)
    fm = sparse.fit(ds)
    xs = rng.uniform(0., 10., (200, 3))
    d, s = dense.predict(xs).marginal(), fm.predict(xs).marginal()
    assert np.abs(d.mean - s.mean).max() <= 1e-5 and np.abs(d.covariance - s.covariance).max() <= 1e-5
    assert abs(sparse.log_likelihood(ds) - ab.gp_from_covariance(cov, context=ctx).log_likelihood(ds)) <= 1e-4 * n


def test_sparse_error_paths(ctx):
    x, y = toy_linear()
    cov = simple_cov(100.)
    with pytest.raises(ValueError):
        ab.sparse_gp_from_covariance(cov, None, None)
    dup = ab.sparse_gp_from_covariance(cov, interval_grouper(5.0), ab.FixedInducingPoints(np.array([1.0, 1.0, 2.0])),
                                       "dup", context=ctx)
    dup.set_param("inducing_nugget", 0.0)
    # an exactly singular K_uu (a repeated inducing point, no nugget): LL^T rejects it and the reference's pivoted
    # L D L^T / QR take over - the fit the reference itself would produce, rank deficient, with finite predictions
    fm = dup.fit(ab.RegressionDataset(x, y))
    keys = np.floor(x / 5.).astype(np.int64)
    ofit = orc.OracleSparseFit(cov, x, keys, y, None, np.array([1.0, 1.0, 2.0]), 1e-8, 0.0)
    assert fm.get_fit().numerical_rank == ofit.numerical_rank == 2
    xs = np.linspace(0.1, 9.9, 5)
    om, ov = ofit.predict(xs)
    p = fm.predict(xs).marginal()
    assert np.all(np.isfinite(p.mean)) and np.abs(p.mean - om).max() <= 1e-6 * np.abs(om).max()
    # the pivot that is zero in exact arithmetic comes out as +-1e-12 from the last bits of K_uu (the device's exp differs
    # from libm's by an ulp), and D^-1/2 either zeroes or amplifies that direction: the variance is only defined to the
    # size of the term that direction carries
    assert np.all(np.isfinite(p.covariance)) and np.abs(p.covariance - ov).max() <= 0.05 * ov.max()
    bad = ab.sparse_gp_from_covariance(cov, interval_grouper(5.0), ab.UniformlySpacedInducingPoints(4), context=ctx)
    yn = y.copy()
    with pytest.raises(KeyError):
        bad.set_param("no_such_parameter", 1.0)
    assert np.isfinite(bad.log_likelihood(ab.RegressionDataset(x, yn)))


def test_update_equals_full_fit(ctx):
    """tests/test_sparse_gp.cc:293-371 (test_update): fit without the first group, update with it ==
    fit on everything (1e-6 on mean / covariance), while the partial fit is clearly different."""
    x, y = toy_linear()
    cov = simple_cov(100.)
    grouper = interval_grouper(5.0)
    u = np.linspace(x.min(), x.max(), 8)  # FixedInducingPoints(min, max, 8)
    sparse = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    sparse.set_param_value("inducing_nugget", 1e-3)
    sparse.set_param_value("measurement_nugget", 1e-12)
    keys = np.array([grouper(f) for f in x])
    held = keys == keys.min()
    full = sparse.fit(ab.RegressionDataset(x, y))
    partial = sparse.fit(ab.RegressionDataset(x[~held], y[~held]))
    updated = partial.update(ab.RegressionDataset(x[held], y[held]))
    xs = np.linspace(0.01, 9.9, 11)
    fp, pp, up = (f.predict_with_measurement_noise(xs).joint() for f in (full, partial, updated))
    assert np.linalg.norm(pp.mean - fp.mean) > 1e-2 and np.linalg.norm(pp.covariance - fp.covariance) > 1e-1
    assert np.linalg.norm(up.mean - fp.mean) < 1e-6 and np.linalg.norm(up.covariance - fp.covariance) < 1e-6
    assert np.abs(updated.get_fit().information - full.get_fit().information).max() < 1e-6
    # the old handle is untouched and still predicts as before
    assert np.array_equal(partial.predict_with_measurement_noise(xs).joint().mean, pp.mean)


@pytest.mark.parametrize("n,m,gs", [(600, 40, 100), (1024, 96, 128)])
def test_update_matches_oracle_and_chains(ctx, n, m, gs):
    rng = np.random.default_rng(n)
    x = np.sort(rng.uniform(0., 30., n))
    y = np.sin(x) + 0.2 * x + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.04, n)
    cov = ab.Matern52(3.0, 1.5) + ab.measurement_only(ab.IndependentNoise(0.2))
    u = np.linspace(0., 30., m)
    rank = {float(v): i for i, v in enumerate(x)}
    grouper = lambda f: rank[float(f)] // gs
    keys = np.array([grouper(f) for f in x])
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    first, second, third = keys < 2, (keys >= 2) & (keys < 4), keys >= 4
    ds = lambda sel: ab.RegressionDataset(x[sel], ab.MarginalDistribution(y[sel], yvar[sel]))
    chained = model.fit(ds(first)).update(ds(second)).update(ds(third))
    o = orc.OracleSparseFit(cov, x[first], keys[first], y[first], yvar[first], u, 1e-8, 1e-6)
    o = o.update(x[second], keys[second], y[second], yvar[second], 1e-8, 1e-6)
    o = o.update(x[third], keys[third], y[third], yvar[third], 1e-8, 1e-6)
    v = o.information
    assert np.abs(chained.get_fit().information - v).max() <= 1e-6 * np.abs(v).max()
    xs = np.linspace(0.5, 29.5, 33)
    om, ov, oj = o.predict(xs, xs_meas=True, joint=True)
    j = chained.predict_with_measurement_noise(xs).joint()
    assert np.abs(j.mean - om).max() <= 1e-7 * max(1., np.abs(om).max())
    assert np.abs(j.covariance - oj).max() <= 1e-7 * np.abs(oj).max()
    full = model.fit(ds(np.ones(n, dtype=bool))).predict_with_measurement_noise(xs).joint()
    assert np.abs(j.mean - full.mean).max() <= 1e-6 and np.abs(j.covariance - full.covariance).max() <= 1e-6


def test_vectorized_grouper_gives_the_same_fit(ctx):
    """a grouper marked `vectorized` is called once on the whole feature array; same groups, same fit"""
    rng = np.random.default_rng(4)
    x = np.sort(rng.uniform(0., 40., 600))
    y = np.sin(x) + 0.1 * rng.standard_normal(600)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))
    u = np.linspace(0., 40., 40)
    scalar = lambda f: int(float(f) // 5.)
    def vec(f):
        return (np.asarray(f, dtype=np.float64).reshape(-1) // 5.).astype(np.int64)
    vec.vectorized = True
    fits = []
    for g in (scalar, vec):
        m = ab.sparse_gp_from_covariance(cov, g, ab.FixedInducingPoints(u), "pitc", context=ctx)
        fits.append(m.fit(ab.RegressionDataset(x, y)).get_fit())
    assert np.array_equal(fits[0].information, fits[1].information) and fits[0].nll == fits[1].nll


def _toy_model(ctx, strategy, inducing_nugget=None, measurement_nugget=None):
    model = ab.sparse_gp_from_covariance(simple_cov(100.0), interval_grouper(5.0), strategy, "sparse", context=ctx)
    if inducing_nugget is not None:
        model.set_param("inducing_nugget", inducing_nugget)
    if measurement_nugget is not None:
        model.set_param("measurement_nugget", measurement_nugget)
    return model


def test_rebase_inducing_points(ctx):
    """tests/test_sparse_gp.cc:374-416 (test_rebase_inducing_points) on the device, with the reference's thresholds, and
    every rebased fit against the oracle's restatement of fit_from_prediction (sparse_gp.hpp:406-461).  K_zz of 51
    points under a length scale of 100 has numerical rank 5: the path runs the pivoted L D L^T and the pivoted QR."""
    x, y = toy_linear()
    u = np.linspace(x.min(), x.max(), 8)
    model = _toy_model(ctx, ab.FixedInducingPoints(u), 1e-3, 1e-12)
    full = model.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y)))
    xs = np.linspace(0.01, 9.9, 11)
    full_mean = full.predict_with_measurement_noise(xs).joint().mean
    keys = np.floor(x / 5.).astype(np.int64)
    ofull = orc.OracleSparseFit(simple_cov(100.0), x, keys, y, None, u, 1e-12, 1e-3)

    low = ab.rebase_inducing_points(full, np.array([5.]))
    low_mean = low.predict_with_measurement_noise(xs).joint().mean
    assert np.linalg.norm(low_mean - full_mean) > 10.
    olow = ofull.rebase(np.array([5.]))
    assert np.abs(low.get_fit().information - olow.information).max() <= 1e-9 * np.abs(olow.information).max()
    om, ov = olow.predict(xs, xs_meas=True)
    lp = low.predict_with_measurement_noise(xs).marginal()
    assert np.abs(lp.mean - om).max() <= 1e-8 * np.abs(om).max() and np.abs(lp.covariance - ov).max() <= 1e-8 * ov.max()

    z = np.linspace(0.01, 9.9, 51)
    high = ab.rebase_inducing_points(full, z)
    hp = high.predict_with_measurement_noise(xs).joint()
    assert np.linalg.norm(hp.mean - full_mean) < 1e-6
    ohigh = ofull.rebase(z)
    om, ov, oj = ohigh.predict(xs, xs_meas=True, joint=True)
    # the rebased fit is defined through a numerically singular K_zz: parity is on what it predicts.  The predictive
    # covariance K_** - Q_** + S_** cancels terms of the size of the prior variance (1e4) down to 1e-2: its bar is
    # relative to the prior (a different pivot at a rounding-level tie changes it by ~1e-7 of that).
    prior = 100.0 ** 2
    assert np.abs(hp.mean - om).max() <= 1e-6 and np.abs(hp.covariance - oj).max() <= 5e-7 * prior
    assert high.get_fit().numerical_rank < 51 and ohigh.numerical_rank < 51

    low_high = ab.rebase_inducing_points(low, z)
    assert np.linalg.norm(low_high.predict_with_measurement_noise(xs).joint().mean - full_mean) > 10.


def test_rebase_and_update(ctx):
    """tests/test_sparse_gp.cc:418-456 (test_rebase_and_update): fit the first group, rebase to the inducing points of
    the whole data set, update with the remaining groups == direct fit (4e-3 on the mean, 8e-3 on the covariance), and
    the device chain against the oracle's."""
    x, y = toy_linear()
    keys = np.floor(x / 5.).astype(np.int64)
    model = _toy_model(ctx, ab.UniformlySpacedInducingPoints(10))
    u_all = np.linspace(x.min(), x.max(), 10)
    first = keys == keys.min()
    fit = model.fit(ab.RegressionDataset(x[first], ab.MarginalDistribution(y[first])))
    fit = ab.rebase_inducing_points(fit, u_all)
    ofit = orc.OracleSparseFit(simple_cov(100.0), x[first], keys[first], y[first], None,
                               np.linspace(x[first].min(), x[first].max(), 10)).rebase(u_all)
    for k in np.unique(keys[~first]):
        sel = keys == k
        fit = fit.update(ab.RegressionDataset(x[sel], ab.MarginalDistribution(y[sel])))
        ofit = ofit.update(x[sel], keys[sel], y[sel], None)
    direct = model.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y)))
    xs = np.linspace(0.1, 9.9, 5)
    ip, dp = fit.predict(xs).joint(), direct.predict(xs).joint()
    assert np.linalg.norm(ip.mean - dp.mean) < 4e-3
    assert np.linalg.norm(ip.covariance - dp.covariance) < 8e-3
    om, _, oc = ofit.predict(xs, joint=True)
    assert np.linalg.norm(ip.mean - om) < 4e-3 and np.linalg.norm(ip.covariance - oc) < 8e-3


def test_rebase_well_conditioned_matches_oracle(ctx):
    """fit_from_prediction where nothing is singular (short length scale, few inducing points): information, rank and
    predictions agree with the oracle to rounding; then an update of the rebased fit (the pivoted-QR update)."""
    rng = np.random.default_rng(5)
    n = 300
    x = np.sort(rng.uniform(0., 30., n))
    y = np.sin(x) + 0.2 * x + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.04, n)
    cov = ab.Matern52(3.0, 1.5) + ab.measurement_only(ab.IndependentNoise(0.2))
    grouper = interval_grouper(3.0)
    keys = np.array([grouper(f) for f in x])
    u = np.linspace(0., 30., 12)
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    old = x < 20.
    fm = model.fit(ab.RegressionDataset(x[old], ab.MarginalDistribution(y[old], yvar[old])))
    ofit = orc.OracleSparseFit(cov, x[old], keys[old], y[old], yvar[old], u, 1e-8, 1e-6)
    z = np.linspace(0.5, 29.5, 16)
    rb, orb = ab.rebase_inducing_points(fm, z), ofit.rebase(z)
    assert rb.get_fit().numerical_rank == orb.numerical_rank == 16
    assert np.abs(rb.get_fit().information - orb.information).max() <= 1e-7 * np.abs(orb.information).max()
    xs = np.linspace(0.2, 29.8, 41)
    om, ov, oj = orb.predict(xs, xs_meas=True, joint=True)
    p = rb.predict_with_measurement_noise(xs).joint()
    assert np.abs(p.mean - om).max() <= 1e-7 * np.abs(om).max()
    assert np.abs(p.covariance - oj).max() <= 1e-7 * np.abs(oj).max()
    new = ~old
    up = rb.update(ab.RegressionDataset(x[new], ab.MarginalDistribution(y[new], yvar[new])))
    oup = orb.update(x[new], keys[new], y[new], yvar[new], 1e-8, 1e-6)
    assert np.abs(up.get_fit().information - oup.information).max() <= 1e-7 * np.abs(oup.information).max()
    om, ov, oj = oup.predict(xs, xs_meas=True, joint=True)
    p = up.predict_with_measurement_noise(xs).joint()
    assert np.abs(p.mean - om).max() <= 1e-7 * np.abs(om).max()
    assert np.abs(p.covariance - oj).max() <= 1e-7 * np.abs(oj).max()


@pytest.mark.parametrize("n,m,width", [(60, 7, 5.0), (400, 40, 2.5)])
def test_pivoted_fit_matches_oracle(make_ctx, monkeypatch, n, m, width):
    """The reference's own algorithm on the device (pivoted L D L^T of K_uu, column-pivoted QR of B: the path that takes
    over where LL^T / CholeskyQR2 reject the matrices), forced on a well-conditioned problem: fit, likelihood, rank,
    predictions and an update against the oracle."""
    monkeypatch.setenv("AGP_SPARSE_PIVOTED", "1")
    ctx = make_ctx()  # (the switch is read when the context is created)
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 20., n)
    y = np.sin(x) + 0.3 * x + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.04, n)
    cov = ab.Matern52(4.0, 2.0) + ab.measurement_only(ab.IndependentNoise(0.2))
    u = np.linspace(0., 20., m)
    grouper = interval_grouper(width)
    keys = np.array([grouper(f) for f in x])
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    model.set_param("measurement_nugget", 1e-10)
    old = keys != keys.max()
    fm = model.fit(ab.RegressionDataset(x[old], ab.MarginalDistribution(y[old], yvar[old])))
    ofit = orc.OracleSparseFit(cov, x[old], keys[old], y[old], yvar[old], u, 1e-10, 1e-6)
    v = ofit.information
    assert fm.get_fit().numerical_rank == ofit.numerical_rank == m
    assert np.abs(fm.get_fit().information - v).max() <= 1e-8 * np.abs(v).max()
    assert abs(fm.get_fit().nll - ofit.nll) <= 1e-9 * n
    xs = np.linspace(0.01, 19.9, 37)
    om, ov, oj = ofit.predict(xs, xs_meas=True, joint=True)
    p = fm.predict_with_measurement_noise(xs).joint()
    assert np.abs(p.mean - om).max() <= 1e-9 * np.abs(om).max()
    assert np.abs(p.covariance - oj).max() <= 1e-9 * np.abs(oj).max()
    new = ~old
    up = fm.update(ab.RegressionDataset(x[new], ab.MarginalDistribution(y[new], yvar[new])))
    oup = ofit.update(x[new], keys[new], y[new], yvar[new], 1e-10, 1e-6)
    assert np.abs(up.get_fit().information - oup.information).max() <= 1e-8 * np.abs(oup.information).max()
    om, ov, oj = oup.predict(xs, xs_meas=True, joint=True)
    p = up.predict_with_measurement_noise(xs).joint()
    assert np.abs(p.mean - om).max() <= 1e-9 * np.abs(om).max()
    assert np.abs(p.covariance - oj).max() <= 1e-9 * np.abs(oj).max()


def test_singular_inducing_covariance_falls_back_to_the_pivoted_path(ctx):
    """make_simple_covariance_function() with the DEFAULT nuggets (1e-8) on 10 inducing points under a length scale of
    100 (the model of tests/test_sparse_gp.cc:418-456): B^T B is singular to working precision, LL^T rejects it, the
    pivoted path produces the reference's fit - against the oracle, which runs the same algorithm on the CPU."""
    x, y = toy_linear()
    keys = np.floor(x / 5.).astype(np.int64)
    model = _toy_model(ctx, ab.UniformlySpacedInducingPoints(10))
    fm = model.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y)))
    u = np.linspace(x.min(), x.max(), 10)
    ofit = orc.OracleSparseFit(simple_cov(100.0), x, keys, y, None, u)
    assert fm.get_fit().numerical_rank <= 10
    xs = np.linspace(0.1, 9.9, 5)
    om, _, oc = ofit.predict(xs, joint=True)
    p = fm.predict(xs).joint()
    assert np.linalg.norm(p.mean - om) < 1e-6 * np.linalg.norm(om)
    assert np.abs(p.covariance - oc).max() <= 5e-7 * 100.0 ** 2
    assert abs(fm.get_fit().nll - ofit.nll) <= 1e-6 * max(1., abs(ofit.nll))


def test_rebase_many_points_2d_matches_oracle(ctx):
    """fit_from_prediction with 150 new inducing points in two dimensions: the blocked paths of the pivoted
    factorisations (64-wide diagonal blocks of the L D L^T and R^T substitutions, the blocked L D L^T itself)."""
    rng = np.random.default_rng(11)
    n = 500
    x = rng.uniform(0., 10., (n, 2))
    y = np.sin(x[:, 0]) * np.cos(x[:, 1]) + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.04, n)
    cov = ab.Matern32(1.5, 1.2) + ab.measurement_only(ab.IndependentNoise(0.2))
    grouper = lambda f: int(f[0] // 2.0) * 8 + int(f[1] // 2.0)
    keys = np.array([grouper(f) for f in x])
    u = rng.uniform(0., 10., (90, 2))
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "sparse", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    fm = model.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    ofit = orc.OracleSparseFit(cov, x, keys, y, yvar, u, 1e-8, 1e-6)
    z = rng.uniform(0., 10., (150, 2))
    rb, orb = ab.rebase_inducing_points(fm, z), ofit.rebase(z)
    assert rb.get_fit().numerical_rank == orb.numerical_rank
    xs = rng.uniform(0., 10., (60, 2))
    om, ov, oj = orb.predict(xs, xs_meas=True, joint=True)
    p = rb.predict_with_measurement_noise(xs).joint()
    assert np.abs(p.mean - om).max() <= 1e-6 * np.abs(om).max()
    assert np.abs(p.covariance - oj).max() <= 1e-6 * np.abs(oj).max()


def test_fit_from_prediction_device_resident_inputs(ctx):
    """agp_sparse_fit_from_prediction with the prediction already in HBM (location = AGP_DEVICE, a leading dimension
    larger than m) gives the fit it gives for host inputs."""
    import ctypes as C
    import torch
    from albatross_amd import _capi as capi
    rng = np.random.default_rng(3)
    z = np.linspace(0.5, 9.5, 20)
    cov = ab.Matern52(2.0, 1.5) + ab.measurement_only(ab.IndependentNoise(0.2))
    K = orc.gram(cov, z)
    mean = rng.standard_normal(20)
    Cm = 0.3 * K + 0.05 * np.eye(20)
    fz = cov.features(z)
    sz = fz.as_struct()
    lib = ctx._lib
    infos, ranks = [], []
    ld = 24
    pad = np.zeros((ld, 20), order="F")
    pad[:20] = Cm
    dev_mean = torch.from_numpy(mean).cuda()
    dev_cov = torch.from_numpy(np.ascontiguousarray(pad.T)).cuda()  # memory of the column-major ld x 20 array
    host_cov = np.asfortranarray(Cm)  # kept alive: .ctypes.data of a temporary would dangle
    for args in ((mean.ctypes.data, host_cov.ctypes.data, 20, capi.HOST),
                 (dev_mean.data_ptr(), dev_cov.data_ptr(), ld, capi.DEVICE)):
        h = C.c_void_p()
        info = np.zeros(20)
        rank = C.c_int64()
        st = lib.agp_sparse_fit_from_prediction(ctx._h, ctx.kernel(cov), C.byref(sz), C.c_void_p(args[0]), C.c_void_p(args[1]),
                                                args[2], args[3], 1e-8, C.byref(h), info.ctypes.data_as(C.c_void_p), C.byref(rank))
        assert st == capi.AGP_OK
        ranks.append(rank.value)
        xs = np.linspace(0., 10., 7)
        fs = cov.features(xs).as_struct()
        m, v = np.zeros(7), np.zeros(7)
        assert lib.agp_sparse_predict_marginal(ctx._h, ctx.kernel(cov), h, C.byref(fs), m.ctypes.data_as(C.c_void_p),
                                               v.ctypes.data_as(C.c_void_p), capi.HOST) == capi.AGP_OK
        lib.agp_sparse_fit_destroy(h)
        infos.append((info, m, v))
    for a, b in zip(infos[0], infos[1]):
        assert np.array_equal(a, b)
    ofit = orc.OracleSparseFit.from_prediction(cov, z, mean, Cm)
    assert ranks[0] == ranks[1] == ofit.numerical_rank
    assert np.abs(infos[0][0] - ofit.information).max() <= 1e-6 * np.abs(ofit.information).max()


def test_rebased_fit_predicts_many_points(ctx):
    """A fit in pivoted form predicts 200 000 points in one call (more right-hand sides than the y extent of a grid:
    the permutation kernels stride over them) and agrees with the oracle on a sample."""
    x, y = toy_linear()
    u = np.linspace(x.min(), x.max(), 8)
    model = _toy_model(ctx, ab.FixedInducingPoints(u), 1e-3, 1e-12)
    full = model.fit(ab.RegressionDataset(x, ab.MarginalDistribution(y)))
    rb = ab.rebase_inducing_points(full, np.linspace(0.5, 9.5, 6))
    xs = np.linspace(0., 10., 200000)
    p = rb.predict(xs).marginal()
    assert np.all(np.isfinite(p.mean)) and np.all(np.isfinite(p.covariance))
    keys = np.floor(x / 5.).astype(np.int64)
    orb = orc.OracleSparseFit(simple_cov(100.0), x, keys, y, None, u, 1e-12, 1e-3).rebase(np.linspace(0.5, 9.5, 6))
    idx = np.arange(0, 200000, 9973)
    om, ov = orb.predict(xs[idx])
    assert np.abs(p.mean[idx] - om).max() <= 1e-7 * np.abs(om).max()
    assert np.abs(p.covariance[idx] - ov).max() <= 5e-7 * 100.0 ** 2
