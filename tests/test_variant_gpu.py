"""variant<> feature vectors (VariantForwarder, covariance_functions/callers.hpp:419-544):
every point carries the index of the alternative it holds; a covariance term declared with
`only_for_alternatives(cov, a, b)` is what a `_call_impl(const A &, const B &)` overload is in the reference, and
pairs of alternatives without a term contribute exactly 0 (AGP_OP_TYPE_PAIR in the device program)."""
import numpy as np
import pytest

import albatross_amd as ab
from oracle import oracle_py as orc
from test_oracle_golden import _has_multiple

pytestmark = pytest.mark.gpu


def test_reference_dispatch_table_on_the_device(ctx):
    cov = _has_multiple()
    feats = ab.VariantFeatures([0, 1, 2, 3, 1, 0], [0., 0., 0., 0., 0., 0.])
    K = ctx.gram(cov, feats)
    assert np.array_equal(K, orc.gram(cov, feats))
    assert K[0, 1] == K[1, 0] and abs(K[0, 1] - 3.) < 1e-14 and K[0, 2] == 0. and K[1, 2] == 0. and abs(K[2, 2] - 7.) < 1e-14
    Kc = ctx.gram(cov, feats, ab.VariantFeatures([2, 0], [0., 0.]))
    assert np.array_equal(Kc, orc.gram(cov, feats, ab.VariantFeatures([2, 0], [0., 0.])))


def test_gp_on_two_observation_types(ctx):
    """Two alternatives living in different spaces: A = a 1-D position t, B = a 3-D point.  f_A and f_B are independent
    processes plus a shared constant offset seen by both: cov = SE_t on (A, A) + Matern on (B, B) + Constant on every pair.
    Device Gram, fit and predictions against the oracle."""
    rng = np.random.default_rng(8)
    nA, nB = 180, 220
    tA = np.sort(rng.uniform(0., 10., nA))
    pB = rng.uniform(0., 4., (nB, 3))
    alt = np.array([0] * nA + [1] * nB)
    order = rng.permutation(nA + nB)                      # interleave the two kinds
    values = [tA[i] if i < nA else pB[i - nA] for i in order]
    feats = ab.VariantFeatures(alt[order], values)
    cov = (ab.only_for_alternatives(ab.SquaredExponential(1.5, 1.0), 0) + ab.only_for_alternatives(ab.Matern52(2.0, 0.8), 1)
           + ab.Constant(0.5) + ab.IndependentNoise(0.1))
    K = ctx.gram(cov, ab.Measurement(feats))
    Ko = orc.gram(cov, feats, x_meas=True)
    assert np.abs(K - Ko).max() <= 1e-14 * np.abs(Ko).max()
    a_idx, b_idx = np.where(alt[order] == 0)[0], np.where(alt[order] == 1)[0]
    assert np.allclose(K[np.ix_(a_idx, b_idx)], 0.25, atol=1e-15)   # across the kinds only the shared constant
    y = rng.standard_normal(nA + nB)
    model = ab.gp_from_covariance(cov, context=ctx)
    fm = model.fit(ab.RegressionDataset(feats, y))
    ofit = orc.OracleFit(cov, feats, y)
    assert np.abs(fm.get_fit().information - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
    tests = ab.VariantFeatures([0, 1, 0, 1], [2.5, [1., 2., 3.], 7.7, [0.3, 0.2, 3.9]])
    om, ov = ofit.predict_marginal(tests)
    pred = fm.predict(tests).marginal()
    assert np.abs(pred.mean - om).max() <= 1e-8 * np.abs(om).max()
    assert np.abs(pred.covariance - ov).max() <= 1e-8 * np.abs(ov).max()
    assert abs(-model.log_likelihood(ab.RegressionDataset(feats, y)) - orc.nll(cov, feats, y)) <= 1e-6 * (nA + nB)


def test_products_and_sums_with_undefined_sides_on_the_device(ctx):
    """device == oracle for the `ignore the side without a caller` rules (covariance_function.hpp:266-294, 357-389)"""
    rng = np.random.default_rng(3)
    alt = rng.integers(0, 3, 60)
    feats = ab.VariantFeatures(alt, [rng.uniform(0., 5.) if a < 2 else rng.uniform(0., 5., 2) for a in alt])
    se0 = ab.only_for_alternatives(ab.SquaredExponential(1.3, 1.1), 0)
    m1 = ab.only_for_alternatives(ab.Matern32(2.0, 0.7), 1)
    x01 = ab.only_for_alternatives(ab.Constant(0.4), 0, 1)
    for cov in (se0 * ab.Constant(1.5) + m1, (se0 + m1) * (x01 + ab.Constant(0.9)), se0 * m1 + x01,
                ab.measurement_only(ab.IndependentNoise(0.3)) * se0 + m1 * ab.Constant(2.0)):
        for meas in (False, True):
            f = ab.Measurement(feats) if meas else feats
            K = ctx.gram(cov, f)
            Ko = orc.gram(cov, feats, x_meas=meas)
            assert np.abs(K - Ko).max() <= 1e-14 * max(np.abs(Ko).max(), 1.), cov.get_name()
