"""CPU tests (-m "not gpu"): pin the oracle against the reference's own golden
vectors (tests/golden/*.json, sources cited there) and check its internal
consistency.  The oracle is the checker of the GPU parity tests."""
import numpy as np
import pytest

import albatross_amd as ab
from conftest import golden
from oracle import oracle_py as orc


@pytest.mark.parametrize("name,cls", [("matern52.json", ab.Matern52), ("matern32.json", ab.Matern32)])
def test_matern_gpytorch_oracle(name, cls):
    g = golden(name)  # tests/test_radial.cc:212-353,355-489, tolerance 1e-15
    K = orc.gram(cls(g["length_scale"], g["sigma"]), np.array(g["x"]))
    assert np.abs(K - np.array(g["K"])).max() < g["tolerance_abs"]


def test_mvn_negative_log_likelihood():
    g = golden("mvn_nll.json")  # tests/test_evaluate.cc:20-44
    nll = orc.nll_dense(np.array(g["x"]), np.array(g["cov"]))
    assert abs(nll - g["nll"]) < g["tolerance_build"]


@pytest.mark.parametrize("cls", [ab.Exponential, ab.SquaredExponential, ab.Matern32, ab.Matern52])
def test_radial_edge_cases(cls):
    g = golden("radial_edges.json")  # tests/test_radial.cc:52-66
    cov = cls(g["length_scale"], g["sigma"])
    s2 = g["sigma"] ** 2
    assert orc.eval_pair(cov, [np.pi], 0, [np.pi], 0) == s2
    assert abs(orc.eval_pair(cov, [np.pi], 0, [np.pi + 1e-16], 0) - s2) < 1e-8
    assert orc.eval_pair(cov, [0.], 0, [1e32], 0) == 0.


def test_distance_metrics():
    g = golden("distances.json")  # tests/test_distance_metrics.cc:20-75 (EXPECT_DOUBLE_EQ)
    for key, metric in (("euclidean", ab.EuclideanDistance), ("radial", ab.RadialDistance),
                        ("angular", ab.AngularDistance)):
        # Exponential(l=1, sigma=1) = exp(-d): recover d = -log k
        cov = ab.Exponential(1., 1., metric())
        for x, y, d in g[key]:
            k = orc.eval_pair(cov, np.array([x], dtype=float), 0, np.array([y], dtype=float), 0)
            assert abs(-np.log(k) - d) <= 4 * np.finfo(float).eps * max(1., d)


def test_measurement_noise_algebra():
    # tests/test_covariance_functions.cc:33-93
    radial = ab.SquaredExponential()
    noise = ab.IndependentNoise()
    meas_noise = ab.measurement_only(noise)
    total = radial + meas_noise
    prod = meas_noise * radial
    prod_of_sum = noise * total
    x = np.array([0., 1., 2.])

    def call(cov, a_meas, b_meas, i=0, j=0):
        return orc.eval_pair(cov, x, i, x, j, a_meas, b_meas)

    assert call(meas_noise, False, False) == 0.
    assert call(meas_noise, False, True) == 0.
    assert call(meas_noise, True, False) == 0.
    assert call(meas_noise, True, True) > 0.
    assert call(radial, False, False) > 0.
    for am, bm in ((True, True), (True, False), (False, True)):
        assert call(radial, am, bm) == call(radial, False, False)
    assert call(total, True, True) > call(total, False, False) > 0.
    for am, bm in ((True, True), (True, False), (False, True)):
        assert call(total, am, bm) == call(radial, am, bm) + call(meas_noise, am, bm)
    assert call(prod, False, False) == 0.
    assert call(prod, True, True) == call(radial, True, True) * call(meas_noise, True, True) > 0.
    assert call(prod, True, False) == 0. and call(prod, False, True) == 0.
    assert call(prod_of_sum, False, False) == call(noise, False, False) * call(total, False, False) > 0.
    assert call(prod_of_sum, True, True) == call(noise, True, True) * call(total, True, True)
    assert call(prod_of_sum, True, False) == call(prod_of_sum, False, False)
    assert call(prod_of_sum, False, True) == call(prod_of_sum, True, False)


def test_product_short_circuit():
    # covariance_function.hpp:362-366: rhs is skipped when lhs == 0, so 0 * inf stays 0
    x = np.array([0., 1.])
    huge = ab.Constant(1e200) * ab.Constant(1e200)  # overflows to inf
    cov = ab.IndependentNoise(1.) * huge
    assert orc.eval_pair(cov, x, 0, x, 1) == 0.
    assert np.isinf(orc.eval_pair(cov, x, 0, x, 0))


def test_toy_linear_gp():
    g = golden("toy_linear.json")
    c = g["cov"]
    cov = ab.SquaredExponential(c["squared_exponential_length_scale"], c["sigma_squared_exponential"]) \
        + ab.measurement_only(ab.IndependentNoise(c["sigma_independent_noise"]))
    x, y = np.array(g["x"]), np.array(g["y"])
    K = orc.gram(cov, x, x_meas=True)
    assert np.abs(K - np.array(g["K_train"])).max() <= 1e-12 * np.abs(K).max()
    tol = g["tolerance_rel"]
    for use_llt in (False, True):
        fit = orc.OracleFit(cov, x, y, use_llt=use_llt)
        info = np.array(g["information"])
        assert np.abs(fit.information - info).max() <= tol * np.abs(info).max()
        assert abs(fit.log_determinant - g["log_det"]) <= 1e-8 * abs(g["log_det"])
        for p in g["predictions"]:
            mean, covm = fit.predict_joint(np.array(p["xs"]))
            assert np.abs(mean - np.array(p["mean"])).max() <= tol * np.abs(np.array(p["mean"])).max()
            assert np.abs(covm - np.array(p["cov"])).max() <= 1e-5  # kappa ~ 1e8 on sigma^2 = 1e4
            m2, var = fit.predict_marginal(np.array(p["xs"]))
            assert np.allclose(m2, mean, rtol=0, atol=1e-9 * np.abs(mean).max())
            assert np.abs(var - np.diag(covm)).max() <= 1e-8 * 1e4
    assert abs(orc.nll(cov, x, y) - g["nll"]) <= 1e-7 * abs(g["nll"])


def test_bench512_gp():
    g = golden("bench512.json")
    c = g["cov"]
    cov = ab.SquaredExponential(c["squared_exponential_length_scale"], c["sigma_squared_exponential"]) \
        + ab.IndependentNoise(c["sigma_independent_noise"])
    x, y, xs = np.array(g["x"]), np.array(g["y"]), np.array(g["xs"])
    tol = g["tolerance_rel"]
    for use_llt in (False, True):
        fit = orc.OracleFit(cov, x, y, use_llt=use_llt)
        info = np.array(g["information"])
        assert np.abs(fit.information - info).max() <= tol * np.abs(info).max()
        assert abs(fit.log_determinant - g["log_det"]) <= tol * abs(g["log_det"])
        mean, var = fit.predict_marginal(xs)
        assert np.abs(mean - np.array(g["mean"])).max() <= tol
        assert np.abs(var - np.array(g["variance"])).max() <= tol
    assert abs(orc.nll(cov, x, y) - g["nll"]) <= tol * abs(g["nll"])


def test_serial_equals_pooled_gram():
    # tests/test_callers.cc:225-266: bitwise equality for pool sizes 1..32
    rng = np.random.default_rng(5)
    x = rng.uniform(0., 10., (600, 3))
    cov = ab.Matern52(2., 1.) + ab.IndependentNoise(0.1)
    serial = orc.gram(cov, x)
    cross = orc.gram(cov, x, x[:77])
    for threads in (2, 3, 8, 32):
        assert np.array_equal(serial, orc.gram(cov, x, threads=threads))
        assert np.array_equal(cross, orc.gram(cov, x, x[:77], threads=threads))


def test_ldlt_matches_llt_and_numpy():
    # tests/test_serializable_ldlt.cc:34-85 property checks, restated
    rng = np.random.default_rng(7)
    for n in (1, 2, 17, 130):
        A = rng.standard_normal((n, n))
        A = A @ A.T + n * np.eye(n)
        B = rng.standard_normal((n, 3))
        packed, tr, ok = orc.ldlt(A)
        assert ok
        X = orc.ldlt_solve(packed, tr, B)
        assert np.abs(A @ X - B).max() < 1e-10
        L, info = orc.llt(A)
        assert info == 0
        assert np.abs(orc.llt_solve(L, B) - X).max() < 1e-11
        sign, logdet = np.linalg.slogdet(A)
        assert abs(orc.ldlt_logdet(packed) - logdet) < 1e-8 and abs(orc.llt_logdet(L) - logdet) < 1e-8


def test_ldlt_pivoting_handles_semidefinite():
    # the reference's pivoted LDLT tolerates rank deficiency (tests/test_gp.cc:20-33)
    v = np.array([[1.], [2.], [3.]])
    A = v @ v.T
    packed, tr, ok = orc.ldlt(A)
    x = orc.ldlt_solve(packed, tr, A @ np.array([1., 1., 1.]))
    assert np.all(np.isfinite(x))
    L, info = orc.llt(A)
    assert info != 0  # un-pivoted LL^T reports the failed pivot instead


def test_ldlt_pivot_order_is_a_function_of_the_diagonal():
    """Eigen 3.3's LDLT (the reference's BlockDiagonalLDLT of the sparse GP's A blocks, linalg/block_diagonal.hpp:24-313,
    models/sparse_gp.hpp:688-697) searches its pivot on the trailing diagonal BEFORE that diagonal has seen the updates of
    the columns already eliminated (the factorisation is left-looking: only A_kk is updated at step k).  The transposition
    sequence is therefore known from diag(A) alone, and for a positive definite block L D^1/2 is the LL^T factor of the
    pre-permuted block - which is why the device factors the A blocks in lock step by LL^T (csrc/sparse_api.hip): the
    products W W^T, W y_w, y_w^T y_w it feeds are invariant under a permutation inside a group, and where a pivot is not
    positive sqrt_solve (serializable_ldlt.hpp:99-109) has no finite value in the reference either."""
    rng = np.random.default_rng(21)
    for n in (5, 64, 200):
        G = rng.standard_normal((n, n))
        A = G @ G.T + np.diag(rng.uniform(0.1, 50., n))
        packed, tr, ok = orc.ldlt(A)
        assert ok
        d = np.abs(np.diag(A)).copy()
        perm = np.arange(n)
        for k in range(n):  # the search of the factorisation, on the diagonal alone
            big = k + int(np.argmax(d[k:]))
            assert tr[k] == big
            d[[k, big]] = d[[big, k]]
            perm[[k, big]] = perm[[big, k]]
        Lunit = np.tril(packed, -1) + np.eye(n)
        D = np.diag(packed).copy()
        assert np.all(D > 0.)
        Lc, info = orc.llt(A[np.ix_(perm, perm)])
        assert info == 0
        assert np.abs(Lunit * np.sqrt(D)[None, :] - np.tril(Lc)).max() <= 1e-12 * np.abs(Lc).max()
        B = rng.standard_normal((n, 4))
        want = orc.ldlt_sqrt_solve(packed, tr, B)  # D^-1/2 L^-1 P B
        got = np.linalg.solve(np.tril(Lc), B[perm])
        assert np.abs(got - want).max() <= 1e-10 * np.abs(want).max()


def test_oracle_nan_input_is_reported():
    cov = ab.SquaredExponential(1., 1.)
    x = np.array([0., np.nan, 2.])
    with pytest.raises(FloatingPointError):
        orc.OracleFit(cov, x, np.zeros(3))


def test_loo_fast_path_equals_brute_force():
    # tests/test_cross_validation.cc:419-446: fast path == brute-force refits (1e-8)
    rng = np.random.default_rng(1)
    n = 60
    x = rng.uniform(0, 10, (n, 2))
    y = np.sin(x).sum(1) + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.05, n)
    cov = ab.Matern52(2., 1.) + ab.IndependentNoise(0.1)
    K = orc.gram(cov, x, x_meas=True) + np.diag(yvar)
    for use_llt in (False, True):
        f = orc.OracleFit(cov, x, y, yvar, use_llt=use_llt)
        assert np.abs(f.inverse_diagonal() - np.diag(np.linalg.inv(K))).max() < 1e-8  # test_serializable_ldlt.cc:52-59
        m, v = f.loo_marginal(y)
        for i in range(n):
            idx = np.r_[0:i, i + 1:n]
            sol = np.linalg.solve(K[np.ix_(idx, idx)], K[idx, i])
            assert abs(m[i] - sol @ y[idx]) < 1e-8 and abs(v[i] - (K[i, i] - K[idx, i] @ sol)) < 1e-8


def test_group_held_out_equals_brute_force():
    # tests/test_serializable_ldlt.cc:40-66 (inverse blocks == blocks of cov.inverse()) and
    # tests/test_cross_validation.cc:202-321 (leave-one-group-out == refit without the group), 1e-8
    rng = np.random.default_rng(2)
    n = 70
    x = rng.uniform(0, 10, (n, 2))
    y = np.sin(x).sum(1) + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.05, n)
    cov = ab.Matern52(2., 1.) + ab.IndependentNoise(0.1)
    K = orc.gram(cov, x, x_meas=True) + np.diag(yvar)
    Kinv = np.linalg.inv(K)
    perm = rng.permutation(n)
    groups = [list(map(int, perm[:9])), list(map(int, perm[9:40])), [int(perm[40])], list(map(int, perm[41:]))]
    for use_llt in (False, True):
        f = orc.OracleFit(cov, x, y, yvar, use_llt=use_llt)
        for blk, g in zip(f.inverse_blocks(groups), groups):
            assert np.abs(blk - Kinv[np.ix_(g, g)]).max() < 1e-8
        for (m, v, J), g in zip(f.held_out(y, groups, joint=True), groups):
            rest = np.setdiff1d(np.arange(n), g)
            sol = np.linalg.solve(K[np.ix_(rest, rest)], K[np.ix_(rest, g)])
            assert np.abs(m - sol.T @ y[rest]).max() < 1e-8
            C = K[np.ix_(g, g)] - K[np.ix_(rest, g)].T @ sol
            assert np.abs(J - C).max() < 1e-8 and np.abs(v - np.diag(C)).max() < 1e-8


def test_sparse_gp_oracle_equals_dense_formulas():
    # models/sparse_gp.hpp:129-243 (the documented identities) and tests/test_sparse_gp.cc:172-218:
    # the QR-based restatement equals the textbook FITC/PITC expressions evaluated densely
    rng = np.random.default_rng(0)
    n, m = 90, 12
    x = np.sort(rng.uniform(0, 20, n))
    y = np.sin(x) + 0.1 * rng.standard_normal(n) + 0.3 * x
    yvar = rng.uniform(0.01, 0.03, n)
    cov = ab.SquaredExponential(3., 2.) + ab.measurement_only(ab.IndependentNoise(0.2))
    u = np.linspace(x.min(), x.max(), m)
    keys = np.floor(x / 5.).astype(np.int64)
    mn, inn = 1e-12, 1e-3
    f = orc.OracleSparseFit(cov, x, keys, y, yvar, u, mn, inn)
    Kuu = orc.gram(cov, u) + inn * np.eye(m)
    Kfu = orc.gram(cov, x, u, x_meas=True)
    Kff = orc.gram(cov, x, x_meas=True)
    Q = Kfu @ np.linalg.solve(Kuu, Kfu.T)
    K = Q.copy()
    for k in np.unique(keys):
        idx = np.nonzero(keys == k)[0]
        K[np.ix_(idx, idx)] = Kff[np.ix_(idx, idx)]
    K += np.diag(yvar) + mn * np.eye(n)
    A = K - Q
    nll = 0.5 * (np.linalg.slogdet(K)[1] + y @ np.linalg.solve(K, y) + n * np.log(2 * np.pi))
    assert abs(f.nll - nll) < 1e-9 * n and f.numerical_rank == m
    Sigma = np.linalg.inv(Kuu + Kfu.T @ np.linalg.solve(A, Kfu))
    v = Sigma @ Kfu.T @ np.linalg.solve(A, y)
    assert np.abs(f.information - v).max() < 1e-9 * np.abs(v).max()
    xs = np.linspace(0.01, 19.9, 11)
    Ksu = orc.gram(cov, xs, u, x_meas=True)
    C = orc.gram(cov, xs, x_meas=True) - Ksu @ np.linalg.solve(Kuu, Ksu.T) + Ksu @ Sigma @ Ksu.T
    mean, var, J = f.predict(xs, xs_meas=True, joint=True)
    assert np.abs(mean - Ksu @ v).max() < 1e-9 and np.abs(J - C).max() < 1e-10 and np.abs(var - np.diag(C)).max() < 1e-10
    # shuffled input order: the grouping / reordering is internal (reordered_inds, :645-662)
    perm = rng.permutation(n)
    g = orc.OracleSparseFit(cov, x[perm], keys[perm], y[perm], yvar[perm], u, mn, inn)
    assert np.abs(g.information - f.information).max() < 1e-9 * np.abs(v).max()


def test_sparse_gp_oracle_update_equals_full_fit():
    # tests/test_sparse_gp.cc:293-371: partial fit + update with the held-out group == full fit (1e-6)
    rng = np.random.default_rng(0)
    n = 90
    x = np.sort(rng.uniform(0, 20, n))
    y = np.sin(x) + 0.1 * rng.standard_normal(n) + 0.3 * x
    yvar = rng.uniform(0.01, 0.03, n)
    cov = ab.SquaredExponential(3., 2.) + ab.measurement_only(ab.IndependentNoise(0.2))
    u = np.linspace(x.min(), x.max(), 12)
    keys = np.floor(x / 5.).astype(np.int64)
    full = orc.OracleSparseFit(cov, x, keys, y, yvar, u, 1e-12, 1e-3)
    held = keys == keys.min()
    part = orc.OracleSparseFit(cov, x[~held], keys[~held], y[~held], yvar[~held], u, 1e-12, 1e-3)
    upd = part.update(x[held], keys[held], y[held], yvar[held], 1e-12, 1e-3)
    xs = np.linspace(0.01, 19.9, 11)
    fm, fv, fj = full.predict(xs, xs_meas=True, joint=True)
    pm, pv, pj = part.predict(xs, xs_meas=True, joint=True)
    um, uv, uj = upd.predict(xs, xs_meas=True, joint=True)
    assert np.linalg.norm(pm - fm) > 1e-2 and np.linalg.norm(pj - fj) > 1e-1
    assert np.linalg.norm(um - fm) < 1e-9 and np.linalg.norm(uj - fj) < 1e-9


def _toy_sparse_problem():
    """make_toy_linear_data + make_simple_covariance_function + LeaveOneIntervalOut (tests/test_sparse_gp.cc:22-29,
    tests/lib/albatross/test/test_models.h:26-30)."""
    g = golden("toy_linear.json")
    x, y = np.array(g["x"]), np.array(g["y"])
    cov = ab.SquaredExponential(100., 100.) + ab.measurement_only(ab.IndependentNoise(0.1))
    keys = np.floor(x / 5.).astype(np.int64)
    return cov, x, y, keys


def test_sparse_gp_oracle_rebase_inducing_points():
    # tests/test_sparse_gp.cc:374-416 (test_rebase_inducing_points), same thresholds
    cov, x, y, keys = _toy_sparse_problem()
    u = np.linspace(x.min(), x.max(), 8)
    full = orc.OracleSparseFit(cov, x, keys, y, None, u, 1e-12, 1e-3)
    xs = np.linspace(0.01, 9.9, 11)
    full_mean = full.predict(xs, xs_meas=True)[0]
    low = full.rebase(np.array([5.]))
    assert np.linalg.norm(low.predict(xs, xs_meas=True)[0] - full_mean) > 10.  # a single point loses information
    high = full.rebase(np.linspace(0.01, 9.9, 51))
    assert np.linalg.norm(high.predict(xs, xs_meas=True)[0] - full_mean) < 1e-6  # more points: nothing changes
    low_high = low.rebase(np.linspace(0.01, 9.9, 51))
    assert np.linalg.norm(low_high.predict(xs, xs_meas=True)[0] - full_mean) > 10.  # what was lost stays lost


def test_sparse_gp_oracle_rebase_and_update():
    # tests/test_sparse_gp.cc:418-456 (test_rebase_and_update): fit the first group, rebase to the inducing points of
    # the whole data set, update with the other groups == direct fit (4e-3 on the mean, 8e-3 on the covariance)
    cov, x, y, keys = _toy_sparse_problem()
    u_all = np.linspace(x.min(), x.max(), 10)
    first = keys == keys.min()
    u_first = np.linspace(x[first].min(), x[first].max(), 10)
    fit = orc.OracleSparseFit(cov, x[first], keys[first], y[first], None, u_first)
    fit = fit.rebase(u_all)
    for k in np.unique(keys[~first]):
        sel = keys == k
        fit = fit.update(x[sel], keys[sel], y[sel], None)
    direct = orc.OracleSparseFit(cov, x, keys, y, None, u_all)
    xs = np.linspace(0.1, 9.9, 5)
    im, _, ic = fit.predict(xs, joint=True)
    dm, _, dc = direct.predict(xs, joint=True)
    assert np.linalg.norm(im - dm) < 4e-3
    assert np.linalg.norm(ic - dc) < 8e-3


def _has_multiple():
    """tests/lib/albatross/test/test_covariance_utils.h:42-62 (HasMultiple): one _call_impl overload per pair of
    alternatives X=0, Y=1, W=2, V=3: (X,X)=1, (X,Y)=3, (Y,Y)=5, (W,W)=7, (V,V)=11."""
    c = lambda v: ab.Constant(np.sqrt(v))
    return (ab.only_for_alternatives(c(1.), 0) + ab.only_for_alternatives(c(3.), 0, 1) + ab.only_for_alternatives(c(5.), 1)
            + ab.only_for_alternatives(c(7.), 2) + ab.only_for_alternatives(c(11.), 3))


def test_variant_dispatch_values_of_the_reference():
    """tests/test_covariance_function.cc:57-170: cov(variant holding a, variant holding b) = the overload for (a, b) in
    either order, 0 where the covariance function has none (X-W, Y-W, ...)."""
    cov = _has_multiple()
    X, Y, W, V = 0, 1, 2, 3
    feats = ab.VariantFeatures([X, Y, W, V], [0., 0., 0., 0.])
    K = orc.gram(cov, feats)
    want = np.array([[1., 3., 0., 0.],
                     [3., 5., 0., 0.],
                     [0., 0., 7., 0.],
                     [0., 0., 0., 11.]])
    assert np.abs(K - want).max() <= 4 * np.finfo(float).eps * 11.   # Constant(sigma) returns sigma * sigma
    assert np.array_equal(K == 0., want == 0.)                          # the undefined pairs are exactly 0


def test_sum_and_product_ignore_a_side_without_a_caller():
    """covariance_function.hpp:266-294, 357-389: a sum / product whose one side has no caller for the pair of
    types keeps the OTHER side alone (it does not become 0); with no caller on either side the pair contributes 0."""
    only_xx = ab.only_for_alternatives(ab.Constant(np.sqrt(2.)), 0)        # defined for (X, X) only
    anywhere = ab.Constant(np.sqrt(3.))
    feats = ab.VariantFeatures([0, 1], [0., 0.])
    Kp = orc.gram(only_xx * anywhere, feats)
    assert abs(Kp[0, 0] - 6.) < 1e-14 and abs(Kp[1, 1] - 3.) < 1e-14 and abs(Kp[0, 1] - 3.) < 1e-14   # (Y,Y), (X,Y): rhs alone
    Ks = orc.gram(only_xx + anywhere, feats)
    assert abs(Ks[0, 0] - 5.) < 1e-14 and abs(Ks[1, 1] - 3.) < 1e-14
    only_yy = ab.only_for_alternatives(ab.Constant(np.sqrt(5.)), 1)
    Kn = orc.gram(only_xx * only_yy, feats)                                 # no pair has both; each diagonal pair has one
    assert abs(Kn[0, 0] - 2.) < 1e-14 and abs(Kn[1, 1] - 5.) < 1e-14 and Kn[0, 1] == 0. and Kn[1, 0] == 0.
    # a defined side that EVALUATES to 0 still annihilates a product (the lhs == 0 short circuit, :361-365)
    meas = ab.measurement_only(ab.IndependentNoise(0.5))
    assert orc.gram(meas * anywhere, feats)[0, 0] == 0.


def test_toy_linear_gp_with_linear_mean():
    """GPs with a mean function (MakeGaussianProcessWithMean, test_models.h:75-96; tests/test_gp.cc:344-371):
    remove_from before the fit, add_to after the prediction (mean_function.hpp:86-107, polynomials.hpp:92-106)."""
    g = golden("toy_linear_mean.json")
    x, y = np.array(g["x"]), np.array(g["y"])
    mean = ab.LinearMean(g["mean"]["slope"], g["mean"]["offset"])
    assert np.array_equal(orc.mean_vector(mean, ab.SquaredExponential(), x), g["mean"]["slope"] * x + g["mean"]["offset"])
    for mdl in g["models"]:
        c = mdl["cov"]
        cov = ab.SquaredExponential(c["squared_exponential_length_scale"], c["sigma_squared_exponential"]) \
            + ab.measurement_only(ab.IndependentNoise(c["sigma_independent_noise"]))
        tol = mdl["tolerance_rel"]
        fit = orc.OracleFit(cov, x, y, mean=mean)
        info = np.array(mdl["information"])
        assert np.abs(fit.information - info).max() <= tol * np.abs(info).max()
        for p in mdl["predictions"]:
            xs, want = np.array(p["xs"]), np.array(p["mean"])
            for got in (fit.predict_mean(xs), fit.predict_marginal(xs)[0], fit.predict_joint(xs)[0]):
                assert np.abs(got - want).max() <= tol * np.abs(want).max()
            assert np.abs(fit.predict_joint(xs)[1] - np.array(p["cov"])).max() <= 1e-5 * c["sigma_squared_exponential"] ** 2
        assert abs(orc.nll(cov, x, y, mean=mean) - mdl["nll"]) <= 1e-7 * abs(mdl["nll"])


def test_mean_function_composition():
    # SumOfMeanFunctions / ProductOfMeanFunctions incl. the `output != 0` short circuit (mean_function.hpp:150-155,221-227)
    x = np.array([-1., 0., 2., 5.])
    cov = ab.SquaredExponential()
    lin, lin2, zero = ab.LinearMean(2., 1.), ab.LinearMean(0.5, 0.), ab.ZeroMean()
    assert np.array_equal(orc.mean_vector(lin + lin2, cov, x), (2. * x + 1.) + 0.5 * x)
    assert np.array_equal(orc.mean_vector(lin * lin2, cov, x), (2. * x + 1.) * (0.5 * x))
    assert np.array_equal(orc.mean_vector(zero * lin, cov, x), np.zeros(4))
    # 0 * inf stays 0: the right-hand side is not evaluated where the left one is zero
    inf = [("linear", 0.5, 0.), ("constant", np.inf), ("product",)]
    assert np.array_equal(orc.mean_vector(inf, cov, x), np.array([-np.inf, 0., np.inf, np.inf]))
    for m in (lin + lin2, lin * lin2, zero * lin, (lin + zero) * lin2):
        assert np.array_equal(orc.mean_vector(m, cov, x), m(x))  # the host mirror evaluates the same
    # ZeroMean leaves the target untouched (mean_function.hpp:90-92,101-103), also NaN entries
    t = np.array([1., np.nan, 3., 4.])
    assert np.array_equal(orc.remove_mean(zero, cov, x, t), t, equal_nan=True)
    assert np.array_equal(orc.add_mean(lin, cov, x, orc.remove_mean(lin, cov, x, np.ones(4))), np.ones(4))


def test_log_likelihood_ignores_target_variance():
    """GaussianProcessBase::log_likelihood (gp.hpp:442-451) evaluates covariance_function_(measurement_features)
    alone; negative_log_likelihood on K + diag(var) is a different number."""
    rng = np.random.default_rng(4)
    x = rng.uniform(0., 5., 40)
    y = np.sin(x)
    cov = ab.SquaredExponential(1., 1.) + ab.measurement_only(ab.IndependentNoise(0.2))
    K = orc.gram(cov, x, x_meas=True)
    assert abs(orc.nll(cov, x, y) - orc.nll_dense(y, K)) < 1e-10
    var = rng.uniform(0.1, 0.2, 40)
    assert abs(orc.nll_with_variance(cov, x, y, var) - orc.nll_dense(y, K + np.diag(var))) < 1e-10
    assert abs(orc.nll_with_variance(cov, x, y, var) - orc.nll(cov, x, y)) > 1e-3


def test_bench_generator_is_libstdcxx_mt19937():
    """bench.py's dataset generator against the compiled one: tests/golden/bench512.json holds
    `std::mt19937 gen(4); std::uniform_real_distribution<double>(0, 10)` (benchmarks/bench_utils.h:25-34) drawn by a C++
    program (tests/golden/make_golden.py)."""
    import bench
    g = golden("bench512.json")
    assert np.array_equal(bench.mt19937_uniform(4, 512), np.array(g["x"]))
    x, y = bench.make_dataset(100, 44)
    assert x.shape == (100, 3) and x.min() >= 0. and x.max() < 10.
    assert np.array_equal(x.reshape(-1), bench.mt19937_uniform(44, 300))  # row-major fill


def test_oracle_update_equals_full_fit():
    """_update_impl / BlockSymmetric restated in the oracle (gp.hpp:384-414, block_symmetric.hpp:46-98) on the
    reference's own property: a partial fit followed by update (also nested) == a full fit (tests/test_gp.cc:182-219,
    tests/test_block_utils.cc:125-147) - when the noise enters through the target variance only."""
    rng = np.random.default_rng(0)
    n = 90
    x = rng.uniform(0., 10., (n, 2))
    y = np.sin(x).sum(axis=1)
    var = rng.uniform(0.05, 0.15, n)
    cov = ab.SquaredExponential(1.5, 1.0) + ab.Constant(2.0)
    full = orc.OracleFit(cov, x, y, var)
    upd = orc.OracleFit(cov, x[:50], y[:50], var[:50]).update(x[50:70], y[50:70], var[50:70]).update(x[70:], y[70:], var[70:])
    assert np.abs(upd.information - full.information).max() <= 1e-10 * np.abs(full.information).max()
    xs = rng.uniform(0., 10., (7, 2))
    for a, b in zip(upd.predict_joint(xs), full.predict_joint(xs)):
        assert np.abs(a - b).max() <= 1e-10
    rhs = rng.standard_normal((n, 3))
    assert np.abs(upd.solve(rhs) - full.solve(rhs)).max() <= 1e-9 * np.abs(full.solve(rhs)).max()
    # with MEASUREMENT-ONLY noise in the covariance function the update is NOT the full fit: the new block's prior is
    # evaluated on plain features (gp.hpp:388-396), the measurement noise of the new observations is left out
    covm = ab.SquaredExponential(1.5, 1.0) + ab.measurement_only(ab.IndependentNoise(0.3))
    fullm = orc.OracleFit(covm, x, y, var)
    updm = orc.OracleFit(covm, x[:50], y[:50], var[:50]).update(x[50:], y[50:], var[50:])
    assert np.abs(updm.information - fullm.information).max() > 1e-3 * np.abs(fullm.information).max()


@pytest.mark.parametrize("n,threads", [(1, 1), (37, 2), (300, 3), (777, 8)])
def test_strong_cpu_llt_matches_the_oracle_llt(n, threads):
    """oracle/strong_llt.c (bench.py's `strong_cpu` context row: blocked, pthread-parallel LL^T with an AVX2 / AVX-512
    micro-kernel, own code) against the oracle's plain left-looking LL^T and numpy's: same factor to rounding, a
    non-positive pivot reported at the same index."""
    rng = np.random.default_rng(n)
    X = rng.standard_normal((n, n + 3))
    A = X @ X.T / n + 2. * np.eye(n)
    Lb = np.tril(orc.llt_blocked(A, threads))
    Lo, info = orc.llt(A)
    assert info == 0
    assert np.abs(Lb - np.tril(Lo)).max() <= 1e-13 * np.abs(Lo).max()
    assert np.abs(Lb - np.linalg.cholesky(A)).max() <= 1e-13 * np.abs(Lo).max()
    if n >= 37:
        A[20, 20] = -1.
        with pytest.raises(FloatingPointError, match="pivot 20 "):
            orc.llt_blocked(A, threads)
        assert orc.llt(A)[1] == 21
