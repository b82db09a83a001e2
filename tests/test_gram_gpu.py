"""GPU parity: agp_gram (HIP) vs the oracle's compute_covariance_matrix
restatement and vs the reference's golden vectors.

Tolerance for a kernel entry: |gpu - oracle| <= 4e-16 * scale + 2e-14 * |value|
with scale = max |K| (device exp() and glibc exp() are each < 1 ulp; a 1-ulp
difference in the exponent argument t costs t * eps relative).  Index /
equality work (noise, nugget, measurement flags) is exact."""
import numpy as np
import pytest

import albatross_amd as ab
from conftest import golden, synthetic_3d
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def close(got, want):
    scale = np.abs(want).max() if want.size else 1.
    return np.all(np.abs(got - want) <= 4e-16 * scale + 2e-14 * np.abs(want))


@pytest.mark.parametrize("name,cls", [("matern52.json", ab.Matern52), ("matern32.json", ab.Matern32)])
def test_matern_golden(ctx, name, cls):
    g = golden(name)
    K = ctx.gram(cls(g["length_scale"], g["sigma"]), np.array(g["x"]))
    # the reference's own bar (tests/test_radial.cc:350,486) is 1e-15 absolute
    assert np.abs(K - np.array(g["K"])).max() < 1e-15


@pytest.mark.parametrize("cls", [ab.Exponential, ab.SquaredExponential, ab.Matern32, ab.Matern52])
def test_radial_edge_cases(ctx, cls):
    g = golden("radial_edges.json")
    cov = cls(g["length_scale"], g["sigma"])
    s2 = g["sigma"] ** 2
    assert ctx.gram(cov, [np.pi], [np.pi])[0, 0] == s2
    assert abs(ctx.gram(cov, [np.pi], [np.pi + 1e-16])[0, 0] - s2) < 1e-8
    assert ctx.gram(cov, [0.], [1e32])[0, 0] == 0.


def test_distance_metrics(ctx):
    g = golden("distances.json")
    for key, metric in (("euclidean", ab.EuclideanDistance), ("radial", ab.RadialDistance),
                        ("angular", ab.AngularDistance)):
        cov = ab.Exponential(1., 1., metric())
        for x, y, d in g[key]:
            k = ctx.gram(cov, np.array([x], dtype=float), np.array([y], dtype=float))[0, 0]
            assert abs(-np.log(k) - d) <= 8 * np.finfo(float).eps * max(1., d)


COVS = {
    "se+noise": lambda: ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1),
    "matern52+noise": lambda: ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1),
    "matern32*exp+nugget": lambda: ab.Matern32(3.0, 2.0) * ab.Exponential(5.0, 1.5) + ab.Nugget(1e-3),
    "const+se_radial": lambda: ab.Constant(0.7) + ab.SquaredExponential(4.0, 1.2, ab.RadialDistance()),
    "exp_angular*se_radial": lambda: ab.Exponential(1.1, 1.0, ab.AngularDistance())
    * ab.SquaredExponential(6.0, 3.7, ab.RadialDistance()) + ab.measurement_only(ab.IndependentNoise(1.75)),
}


@pytest.mark.parametrize("which", sorted(COVS))
@pytest.mark.parametrize("n,dim", [(1, 3), (5, 1), (127, 2), (128, 3), (129, 3), (777, 3), (300, 5), (64, 8)])
def test_symmetric_gram_matches_oracle(ctx, which, n, dim):
    rng = np.random.default_rng(n * 31 + dim)
    x = rng.uniform(0.5, 10., (n, dim))
    if n > 4:
        x[3] = x[1]  # duplicate feature: noise must appear OFF the diagonal too (noise.hpp:37-43)
    cov = COVS[which]()
    for meas in (False, True):
        xa = ab.Measurement(x) if meas else x
        got = ctx.gram(cov, xa)
        want = orc.gram(cov, x, x_meas=meas)
        assert close(got, want)
        assert np.array_equal(got, got.T)
        if n > 4 and "noise" in which and (meas or "measurement" not in cov.get_name()):
            assert got[3, 1] == got[1, 1]


@pytest.mark.parametrize("which", sorted(COVS))
def test_cross_gram_matches_oracle(ctx, which):
    rng = np.random.default_rng(11)
    x = rng.uniform(0.5, 10., (333, 3))
    y = rng.uniform(0.5, 10., (95, 3))
    y[7] = x[20]
    cov = COVS[which]()
    assert close(ctx.gram(cov, x, y), orc.gram(cov, x, y))
    assert close(ctx.gram(cov, ab.Measurement(x), ab.Measurement(y)), orc.gram(cov, x, y, True, True))
    assert close(ctx.gram(cov, x, ab.Measurement(y)), orc.gram(cov, x, y, False, True))


def test_polynomial_and_scaling_terms(ctx):
    class Elevation(ab.ScalingFunction):
        _params = {"center": 4.0, "factor": 0.3}

        def _call_impl(self, c):
            return 1. + self._params["factor"] * np.maximum(self._params["center"] - np.asarray(c)[:, 0], 0.)

    x = np.linspace(-3., 7., 150)
    cov = ab.Polynomial(1, 100.) + ab.SquaredExponential(3.5, 5.7) + ab.measurement_only(ab.IndependentNoise(1.0))
    assert close(ctx.gram(cov, ab.Measurement(x)), orc.gram(cov, x, x_meas=True))
    cov2 = ab.ScalingTerm(Elevation()) * ab.Constant(5.07) + ab.Polynomial(2, 0.5)
    assert close(ctx.gram(cov2, x, x[:40]), orc.gram(cov2, x, x[:40]))
    assert close(ctx.gram(cov2, x), orc.gram(cov2, x))


def test_measurement_noise_algebra_exact(ctx):
    # tests/test_covariance_functions.cc:33-93 through the device path
    radial = ab.SquaredExponential()
    noise = ab.IndependentNoise()
    meas_noise = ab.measurement_only(noise)
    total = radial + meas_noise
    prod = meas_noise * radial
    f = np.array([0.])
    m = ab.Measurement(f)

    def call(cov, a, b):
        return ctx.gram(cov, a, b)[0, 0]

    assert call(meas_noise, f, f) == 0. and call(meas_noise, f, m) == 0. and call(meas_noise, m, f) == 0.
    assert call(meas_noise, m, m) == 0.1 * 0.1
    assert call(radial, m, m) == call(radial, f, f) == call(radial, m, f) > 0.
    assert call(total, m, m) == call(radial, m, m) + call(meas_noise, m, m)
    assert call(total, m, f) == call(radial, m, f)
    assert call(prod, f, f) == 0. and call(prod, m, f) == 0.
    assert call(prod, m, m) == call(radial, m, m) * call(meas_noise, m, m)


def test_product_short_circuit_on_device(ctx):
    x = np.array([0., 1.])
    cov = ab.IndependentNoise(1.) * (ab.Constant(1e200) * ab.Constant(1e200))
    K = ctx.gram(cov, x)
    assert K[1, 0] == 0. and K[0, 1] == 0. and np.isinf(K[0, 0])


def test_gram_config3_shape_properties(ctx):
    """N = 16384 (BASELINE config 3) through size-independent properties:
    symmetry, unit-plus-noise diagonal, and a sampled block against the oracle."""
    x, _ = synthetic_3d(16384, 44)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    K = ctx.gram(cov, x)
    assert np.array_equal(K, K.T)
    assert np.all(np.diag(K) == 1.0 + 0.1 * 0.1)
    idx = np.random.default_rng(0).choice(16384, 300, replace=False)
    assert close(K[np.ix_(idx, idx)], orc.gram(cov, x[idx]))


def _random_tree(rng, depth, dim):
    """A random composed covariance function over every leaf / operator the descriptor can express."""
    metrics = [ab.EuclideanDistance, ab.RadialDistance] + ([ab.AngularDistance] if dim > 1 else [])
    if depth == 0 or rng.random() < 0.3:
        kind = rng.integers(0, 9)
        if kind < 4:
            cls = [ab.SquaredExponential, ab.Exponential, ab.Matern32, ab.Matern52][kind]
            metric = metrics[rng.integers(len(metrics))]
            if cls is ab.SquaredExponential and metric is ab.AngularDistance:
                metric = ab.RadialDistance  # static_assert in radial.hpp:138-141: SE over angles is not PSD
            return cls(float(rng.uniform(0.5, 6.0)), float(rng.uniform(0.3, 2.0)), metric())
        if kind == 4:
            return ab.Constant(float(rng.uniform(0.2, 2.0)))
        if kind == 5:
            return ab.IndependentNoise(float(rng.uniform(0.05, 0.5)))
        if kind == 6:
            return ab.Nugget(float(rng.uniform(0.01, 0.1)))
        if kind == 7:
            return ab.Polynomial(int(rng.integers(0, 3)), float(rng.uniform(0.1, 0.5)))
        return ab.measurement_only(ab.IndependentNoise(float(rng.uniform(0.05, 0.5))))
    op = rng.integers(0, 5)
    lhs, rhs = _random_tree(rng, depth - 1, dim), _random_tree(rng, depth - 1, dim)
    if op < 2:
        return lhs + rhs
    if op < 4:
        return lhs * rhs
    return ab.measurement_only(lhs) + rhs


@pytest.mark.parametrize("evaluator", ["sop", "interpreter"])
@pytest.mark.parametrize("seed", range(24))
def test_random_covariance_trees_match_oracle(make_ctx, seed, evaluator, monkeypatch):
    """Parity sweep of both generic evaluators (sum-of-products where the tree expands to few products, the
    postfix interpreter otherwise or when AGP_GRAM_SOP=0): random sums / products / measurement-only wrappers of
    every leaf, symmetric and cross Gram, plain and Measurement<> features, with repeated points (equality terms)."""
    monkeypatch.setenv("AGP_GRAM_SOP", "1" if evaluator == "sop" else "0")
    ctx = make_ctx()  # (the switch is read when the context is created)
    rng = np.random.default_rng(1000 + seed)
    dim = int(rng.integers(1, 4))
    cov = _random_tree(rng, 3, dim)
    n, m = 137, 61
    x = rng.uniform(0.5, 5.0, (n, dim)) if dim > 1 else rng.uniform(0.5, 5.0, n)
    x[5] = x[17]  # equal features: IndependentNoise / Nugget fire off the diagonal too
    xs = rng.uniform(0.5, 5.0, (m, dim)) if dim > 1 else rng.uniform(0.5, 5.0, m)
    xs[3] = x[9]
    for x_meas in (False, True):
        got = ctx.gram(cov, ab.Measurement(x) if x_meas else x)
        want = orc.gram(cov, x, x_meas=x_meas)
        assert close(got, want), cov.get_name()
        assert np.array_equal(got, got.T)
        for y_meas in (False, True):
            gc = ctx.gram(cov, ab.Measurement(x) if x_meas else x, ab.Measurement(xs) if y_meas else xs)
            wc = orc.gram(cov, x, xs, x_meas=x_meas, y_meas=y_meas)
            assert close(gc, wc), cov.get_name()


def test_exp_neg_accuracy(ctx):
    """The library's own exp(-t) (csrc/cov_eval.h: Cody-Waite reduction + degree-13 polynomial, 22 instructions)
    against the correctly rounded value (mpmath, 40 digits): <= 1.5 ulp over the whole range, exact at the edge cases."""
    import ctypes as C
    import mpmath as mp
    mp.mp.dps = 40
    lib = ab._capi.load_debug()
    lib.agp_debug_exp_neg.restype = C.c_int
    lib.agp_debug_exp_neg.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(0)
    t = np.concatenate([rng.uniform(0., 40., 4000), rng.uniform(0., 1., 2000), rng.uniform(40., 700., 1000),
                        np.array([0., 1e-300, 0.34657359027997264, 0.6931471805599453, 36., 708., 745., 1e6, np.inf, np.nan])])
    out = np.empty_like(t)
    assert lib.agp_debug_exp_neg(ctx._h, C.c_void_p(t.ctypes.data), t.size, C.c_void_p(out.ctypes.data)) == 0
    assert out[-5] < 1e-300 and out[-4] < 1e-300 and out[-3] == 0. and out[-2] == 0. and np.isnan(out[-1])  # 708, 745: denormal range
    assert out[-10] == 1.
    worst = 0.
    for ti, vi in zip(t[:-5], out[:-5]):
        exact = mp.exp(-mp.mpf(float(ti)))
        worst = max(worst, float(abs(mp.mpf(float(vi)) - exact) / mp.mpf(float(np.spacing(float(exact))))))
    assert worst <= 1.5, worst


def test_acos_fast_accuracy(ctx):
    """The library's own acos for the angular metric (csrc/cov_eval.h: reduction to [0, 1/4], degree-12 polynomial,
    correctly rounded sqrt; ~45 instructions against 93) against the correctly rounded value (mpmath, 40 digits)."""
    import ctypes as C
    import mpmath as mp
    mp.mp.dps = 40
    lib = ab._capi.load_debug()
    lib.agp_debug_acos_fast.restype = C.c_int
    lib.agp_debug_acos_fast.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.uniform(-1., 1., 3000), 1. - 10. ** rng.uniform(-16., -0.3, 3000),
                        -1. + 10. ** rng.uniform(-16., -0.3, 1000), rng.uniform(-0.5, 0.5, 1000),
                        np.array([0., 0.5, -0.5, 1., -1., np.nextafter(0.5, 0.), np.nextafter(1., 0.), 1.5, -2., np.nan])])
    out = np.empty_like(t)
    assert lib.agp_debug_acos_fast(ctx._h, C.c_void_p(t.ctypes.data), t.size, C.c_void_p(out.ctypes.data)) == 0
    assert np.isnan(out[-3:]).all()  # outside [-1, 1] and NaN: NaN, like acos
    assert out[-7] == 0. and out[-6] == np.pi and out[-10] == np.pi / 2
    worst = 0.
    for ti, vi in zip(t[:-3], out[:-3]):
        exact = mp.acos(mp.mpf(float(ti)))
        if exact == 0:
            assert vi == 0.
            continue
        worst = max(worst, float(abs(mp.mpf(float(vi)) - exact) / mp.mpf(float(np.spacing(float(exact))))))
    assert worst <= 1.5, worst
