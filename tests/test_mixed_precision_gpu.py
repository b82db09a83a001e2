"""BASELINE.json configs[3] / SURVEY.md section 8d config 4: the mixed-precision fit (agp_fit_create_mixed: the products of
the bulk trailing updates from split 16-bit planes with fp32 accumulation inside a launch - or, AGP_MIXED_BF16=0, on the fp32
MFMA -, fp64 panel chain and accumulation between launches, fp64 conjugate-gradient refinement of the information vector)
against the oracle and against the all-fp64 fit.

Stated tolerances: information vector and predicted means 1e-8 relative (the same as the fp64 path: the refinement
runs to a 1e-12 relative residual); predictive variances 1e-4 relative; log-determinant on the default path (fp16 x 2
planes of scaled rows, four products: csrc/gemm_f16x2.hip): 4e-6 N absolute here (measured, profiles/r06/
mixed_log_determinant_by_path.txt: 1.9e-6 N on Matern-5/2 + noise at N = 5300, 1.4e-6 N on config 3's SE(1,1) + noise(0.1) at
N = 8192 - inside the 2e-6 N log-likelihood bar of the fp64 path, but close to it; BASELINE config 4's covariance at its own
size: 0.5e-6 N, and tests/test_full_size_configs_gpu.py holds it and the other paths: include/albatross_amd.h, agp_fit_create_mixed)."""
import ctypes as C

import numpy as np
import pytest

import albatross_amd as ab
from albatross_amd import _capi as capi
from conftest import synthetic_3d, synthetic_stations, temperature_covariance
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _p(a):
    return C.c_void_p(a.ctypes.data)


@pytest.mark.parametrize("M,K", [(256, 16), (640, 128), (1000, 256), (1418, 512), (130, 32)])
def test_fp32_product_update_kernel(ctx, M, K):
    """trailing_update_f32_kernel: C(fp64) -= fl32(P) fl32(P)^T with fp32 accumulation inside the launch."""
    lib = capi.load_debug()
    lib.agp_debug_trailing_update.restype = C.c_int
    lib.agp_debug_trailing_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                              C.c_int64, C.c_int]
    rng = np.random.default_rng(M + K)
    ldc, ldp = M + 8 - (M % 2), M + 10 - (M % 2)
    Cm = np.asfortranarray(rng.standard_normal((ldc, M)))
    P = np.asfortranarray(rng.standard_normal((ldp, K)))
    P32 = P[:M].astype(np.float32).astype(np.float64)
    want = Cm[:M] - P32 @ P32.T
    got = Cm.copy(order="F")
    assert lib.agp_debug_trailing_update(ctx._h, _p(got), ldc, _p(P), ldp, M, K, 3) == 0
    low = np.tril_indices(M)
    err = np.abs(got[:M][low] - want[low]).max()
    bound = np.abs(P[:M]).sum(axis=1).max() ** 2
    assert err <= 2. ** -22 * bound                      # fp32 accumulation of K products
    exact = Cm[:M] - P[:M] @ P[:M].T
    assert np.abs(got[:M][low] - exact[low]).max() > 1e-12  # ... and it really is the fp32 path
    assert np.array_equal(got[M:], Cm[M:])               # padding rows untouched


@pytest.mark.parametrize("n", [300, 1100, 2300, 5300])
def test_mixed_fit_matches_oracle(ctx, n):
    x, y = synthetic_3d(n, 5 + n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    model = ab.gp_from_covariance(cov, context=ctx)
    model.precision = "mixed"
    fm = model.fit(ab.RegressionDataset(x, y))
    its, res = model.refinement_
    ofit = orc.OracleFit(cov, ab.FeatureSet(x), y)
    # (up to 4608 rows the factorisation has no bulk update at all - chol.hip: step_below, one fp64 launch per panel - so
    # nothing is rounded to fp32 and the first solve already meets the tolerance)
    assert res <= 1e-12 and (its >= 1 or n <= 4608), (its, res)
    assert rel(fm.get_fit().information, ofit.information) <= 1e-8
    xs, _ = synthetic_3d(200, 77)
    om, ov = ofit.predict_marginal(ab.FeatureSet(xs))
    pred = fm.predict(xs).marginal()
    assert rel(pred.mean, om) <= 1e-8
    assert np.abs(pred.covariance - ov).max() <= 1e-4 * np.abs(ov).max()
    with pytest.raises(ab.AlbatrossAmdError, match="mixed-precision factor"):  # opt-in: the fp32 rounding is in it
        fm.get_fit().log_determinant
    fm.get_fit().accept_mixed_log_determinant = True
    assert abs(fm.get_fit().log_determinant - ofit.log_determinant) <= 4e-6 * n  # (measured 1.9e-6 n at n = 5300 on the Matern kernel: at the 2e-6 n bar of the fp64 path)


def test_mixed_fit_config4_kernel(ctx):
    """The temperature-example covariance on synthetic stations: mixed vs all-fp64 vs oracle (small N)."""
    n = 1500
    ecef, h, temp = synthetic_stations(n, 11)
    cov, scale = temperature_covariance(ab)
    train = ab.FeatureSet(ecef, [scale(h)])
    y = temp - temp.mean()
    m64 = ab.gp_from_covariance(cov, context=ctx)
    f64 = m64.fit(ab.RegressionDataset(train, y))
    mm = ab.gp_from_covariance(cov, context=ctx)
    mm.precision = "mixed"
    fmx = mm.fit(ab.RegressionDataset(train, y))
    its, res = mm.refinement_
    assert res <= 1e-12, (its, res)
    ofit = orc.OracleFit(cov, train, y)
    assert rel(f64.get_fit().information, ofit.information) <= 1e-8
    assert rel(fmx.get_fit().information, ofit.information) <= 1e-8
    fmx.get_fit().accept_mixed_log_determinant = True
    assert abs(fmx.get_fit().log_determinant - ofit.log_determinant) <= 4e-6 * n


def test_mixed_fit_large_property(ctx):
    """Size-independent property at N = 8192: the returned information vector solves K a = y (checked against an
    independently built Gram matrix), and agrees with the all-fp64 fit."""
    n = 8192
    x, y = synthetic_3d(n, 44)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    mm = ab.gp_from_covariance(cov, context=ctx)
    mm.precision = "mixed"
    fmx = mm.fit(ab.RegressionDataset(x, y))
    its, res = mm.refinement_
    a = fmx.get_fit().information
    K = ctx.gram(cov, ab.Measurement(x))
    assert np.linalg.norm(K @ a - y) <= 1e-11 * np.linalg.norm(y)
    assert res <= 1e-12 and 1 <= its <= 30, (its, res)
    f64 = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    assert rel(a, f64.get_fit().information) <= 1e-8
    fmx.get_fit().accept_mixed_log_determinant = True
    assert abs(fmx.get_fit().log_determinant - f64.get_fit().log_determinant) <= 4e-6 * n  # (config 3's kernel: measured 1.4e-6 n)


@pytest.mark.parametrize("n", [3072, 4608])
def test_mixed_fit_wide_sweeps_property(ctx, n):
    """Multiples of 512 below 8192: the preconditioner sweeps run through 512-wide inverted blocks and their transposed
    copies, out of place (api.hip: refine_information); the factor's working matrix is a lower-triangle COPY of the kept
    covariance.  Property: K a = y against an independently built Gram matrix, and agreement with the all-fp64 fit."""
    x, y = synthetic_3d(n, 7 + n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.05)
    mm = ab.gp_from_covariance(cov, context=ctx)
    mm.precision = "mixed"
    fmx = mm.fit(ab.RegressionDataset(x, y))
    its, res = mm.refinement_
    a = fmx.get_fit().information
    K = ctx.gram(cov, ab.Measurement(x))
    assert np.linalg.norm(K @ a - y) <= 1e-11 * np.linalg.norm(y)
    assert res <= 1e-12 and its <= 30, (its, res)
    f64 = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    assert rel(a, f64.get_fit().information) <= 1e-8


def test_mixed_fit_badly_scaled_diagonal(ctx):
    """K = D (SE + 0.01 I) D with D spanning 1e-4 ... 1e+4 (a ScalingTerm factor, scaling_function.hpp:58-112, and the noise as
    per-target variances (0.1 d)^2, gp.hpp:64): the diagonal of the covariance spans 16 orders of magnitude.  The fp16 x 2
    products of the mixed factorisation (csrc/gemm_f16x2.hip) scale every row by a power of two taken from the diagonal
    first - without that the large rows overflow fp16 and the small ones vanish, and a scale taken from the wrong row
    would wreck the factor; with it the factor is as good a preconditioner as on the unscaled problem: same 1e-8
    agreement of the information vector with the all-fp64 fit (entry by entry in the unscaled problem's units), a handful
    of CG steps, log-determinant within 4e-6 N."""
    n = 6144  # (> 4608: the factorisation has bulk updates)
    x, y0 = synthetic_3d(n, 91)
    rng = np.random.default_rng(5)
    d = 10. ** rng.uniform(-4., 4., size=n)

    class Given(ab.ScalingFunction):
        def _call_impl(self, c):
            raise AssertionError("scale columns are supplied explicitly")

    cov = ab.ScalingTerm(Given()) * ab.SquaredExponential(1.0, 1.0)
    ds = ab.RegressionDataset(ab.FeatureSet(x, [d]), ab.MarginalDistribution(d * y0, (0.1 * d) ** 2))
    f64 = ab.gp_from_covariance(cov, context=ctx)
    f64.pivoted_fallback = False
    f64 = f64.fit(ds)
    mm = ab.gp_from_covariance(cov, context=ctx)
    mm.precision = "mixed"
    mm.pivoted_fallback = False  # (a failed mixed factorisation must fail the test, not fall back to the pivoted LDL^T)
    fmx = mm.fit(ds)
    its, res = mm.refinement_
    assert 1 <= its <= 10, (its, res)
    a64, amx = np.array(f64.get_fit().information), np.array(fmx.get_fit().information)
    # a = D^-1 (SE + 0.01 I)^-1 y0: compare D a, the solution of the unscaled problem
    assert np.abs(d * amx - d * a64).max() <= 1e-8 * np.abs(d * a64).max()
    fmx.get_fit().accept_mixed_log_determinant = True
    assert abs(fmx.get_fit().log_determinant - f64.get_fit().log_determinant) <= 4e-6 * n


def test_mixed_fit_reports_nan_input(ctx):
    """gp.hpp:66 (ALBATROSS_ASSERT(!cov.hasNaN())) on the mixed path: the NaN is seen by the lower-triangle copy that
    feeds the factorisation (reduce.hip: copy_lower_kernel), not by a second evaluation of the covariance."""
    x, y = synthetic_3d(2048, 9)
    x[100, 1] = np.nan
    model = ab.gp_from_covariance(ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1), context=ctx)
    model.precision = "mixed"
    with pytest.raises(ab.NanInputError):
        model.fit(ab.RegressionDataset(x, y))


def test_mixed_fit_reports_failures(ctx):
    x, y = synthetic_3d(300, 3)
    x[17] = x[3]  # duplicate point and no noise: singular
    model = ab.gp_from_covariance(ab.SquaredExponential(1.0, 1.0), context=ctx)
    model.precision = "mixed"
    model.pivoted_fallback = False
    with pytest.raises(ab.NotPositiveDefiniteError):
        model.fit(ab.RegressionDataset(x, y))
    model.precision = "fp16"
    with pytest.raises(ValueError):
        model.fit(ab.RegressionDataset(x, y))
