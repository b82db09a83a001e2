"""BASELINE.json configs[3] and configs[4] at their STATED sizes (VERDICT r01 "configs_untested"):

  config 4  temperature-example spatial kernel (examples/temperature_example/temperature_example.cc:34-85) on
            N = 32768 synthetic stations, mixed-precision fit against the all-fp64 fit
  config 5  sparse GP (PITC), N = 262144, m = 2048 inducing points, independent groups of 512
            (models/sparse_gp.hpp:129-243, 354-381, 631-706)

The CPU oracle cannot run these sizes, so parity is shown the way the task prescribes for full sizes: the oracle on
a sub-problem / a sampled block of the same data, plus size-independent properties of the full-size result
(residual of the normal equations, update == full fit, linearity in the targets, the PITC defining identity
evaluated group by group with the separately tested dense primitives)."""
import numpy as np
import pytest

import albatross_amd as ab
from conftest import synthetic_3d, synthetic_stations, temperature_covariance
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def rel(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def _subset(fs, idx, is_measurement):
    return ab.FeatureSet(fs.coords[idx], None if fs.scales is None else list(fs.scales[idx].T),
                         None if fs.eq_id is None else fs.eq_id[idx], is_measurement)


def test_config4_n32768_mixed(ctx):
    n = 32768
    ecef, h, temp = synthetic_stations(n, 4)
    cov, scale = temperature_covariance(ab)
    train = ab.FeatureSet(ecef, [scale(h)])
    y = temp - temp.mean()

    # (1) the Gram matrix of THIS data against the oracle on a sampled 300 x 300 block (entries depend on the two
    # points only; rows == cols so that the IndependentNoise diagonal is in the block)
    rng = np.random.default_rng(0)
    idx = np.sort(rng.choice(n, 300, replace=False))
    Kb = ctx.gram(cov, _subset(train, idx, True))
    Ko = orc.gram(cov, _subset(train, idx, False), x_meas=True)
    assert np.all(np.abs(Kb - Ko) <= 4e-16 * np.abs(Ko).max() + 2e-14 * np.abs(Ko))
    # ... and the oracle fit of that sub-problem against the device's (same kernel, same data, small n)
    sub = _subset(train, idx, False)
    ofit = orc.OracleFit(cov, sub, y[idx])
    fsub = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(sub, y[idx]))
    assert rel(fsub.get_fit().information, ofit.information) <= 1e-8

    # (2) full size: all-fp64 fit and mixed-precision fit
    m64 = ab.gp_from_covariance(cov, context=ctx)
    f64 = m64.fit(ab.RegressionDataset(train, y))
    a64 = np.array(f64.get_fit().information)
    ld64 = f64.get_fit().log_determinant
    mm = ab.gp_from_covariance(cov, context=ctx)
    mm.precision = "mixed"
    fmx = mm.fit(ab.RegressionDataset(train, y))
    its, res = mm.refinement_
    amx = np.array(fmx.get_fit().information)
    # the refinement stops on the recurrence residual (1e-12) and reports the TRUE one, which sits at the fp64 floor of
    # this system (cond ~ 2e6, n = 32768): a few 1e-13 above
    assert res <= 2e-12 and 1 <= its <= 50, (its, res)
    assert rel(amx, a64) <= 1e-8                                   # the stated bar for the information vector
    # log|K| of the mixed factor: the bulk products come from two fp16 planes of power-of-two-scaled rows (csrc/gemm_f16x2.hip:
    # four exact partial products, fp32 accumulation inside a launch, fp64 between launches) - MEASURED 0.017 absolute = 3.7e-7
    # relative on this kernel at this size (profiles/r06/time_mixed.txt; round 5's bf16 x 3 products: 0.027), inside the
    # log-likelihood bar of BASELINE.md (|nll error| <= 1e-6 N, i.e. |log_det error| <= 2e-6 N = 0.066).  The bf16 x 3 path
    # (AGP_MIXED_F16=0) and the fp32-MFMA fallback (AGP_MIXED_BF16=0) are held to their own measured bounds by the next test.
    fmx.get_fit().accept_mixed_log_determinant = True  # (opt-in: the bound is per covariance function, include/albatross_amd.h)
    assert abs(fmx.get_fit().log_determinant - ld64) <= 2e-6 * n, abs(fmx.get_fit().log_determinant - ld64)

    # (3) size-independent property: both information vectors solve K a = y, with K rebuilt independently of the
    # fit in row blocks (measurement-wrapped features, as_measurements, gp.hpp:288-290)
    r64 = np.empty(n)
    rmx = np.empty(n)
    allm = _subset(train, np.arange(n), True)
    for lo in range(0, n, 2048):
        rows = np.arange(lo, lo + 2048)
        Krows = ctx.gram(cov, _subset(train, rows, True), allm)
        r64[rows] = Krows @ a64 - y[rows]
        rmx[rows] = Krows @ amx - y[rows]
    ynorm = np.linalg.norm(y)
    assert np.linalg.norm(r64) <= 1e-10 * ynorm, np.linalg.norm(r64) / ynorm
    assert np.linalg.norm(rmx) <= 1e-10 * ynorm, np.linalg.norm(rmx) / ynorm

    # (4) predictions of the two fits agree (means to 1e-8; variances keep the fp32 rounding of the factor)
    es, hs, _ = synthetic_stations(512, 5)
    xs = ab.FeatureSet(es, [scale(hs)])
    p64, pmx = f64.predict(xs).marginal(), fmx.predict(xs).marginal()
    assert rel(pmx.mean, p64.mean) <= 1e-8
    assert np.abs(pmx.covariance - p64.covariance).max() <= 1e-4 * np.abs(p64.covariance).max()
    assert np.all(p64.covariance > 0.)


def test_config4_n32768_mixed_log_determinant_bounds(make_ctx, monkeypatch):
    """What the mixed factor's log|K| is good for, pinned per path and per covariance function at N = 32768.  MEASURED
    (profiles/r06/time_mixed.txt): default path (fp16 x 2, four products) on config 3's kernel (SE(1,1) + noise(0.1)): 0.062
    absolute = 1.9e-6 N - AT the 2e-6 N bar, asserted at 4e-6 N (round 5's bf16 x 3: 0.14); the bf16 x 3 path (AGP_MIXED_F16=0)
    on config 4's covariance: 0.027, asserted at the bar; the fp32-MFMA fallback (AGP_MIXED_BF16=0) on config 4's
    covariance: 1.3e-5 relative, asserted at 5e-5 and asserted OUTSIDE the bar.  The information vector meets 1e-8 on all
    of them (the refinement is fp64)."""
    n = 32768
    # --- config 3's kernel, default (fp16 x 2) path
    ctx = make_ctx()
    x, y = synthetic_3d(n, 44)
    cov3 = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    ds = ab.RegressionDataset(x, y)
    f64 = ab.gp_from_covariance(cov3, context=ctx).fit(ds)
    ld64, a64 = f64.get_fit().log_determinant, np.array(f64.get_fit().information)
    del f64
    mm = ab.gp_from_covariance(cov3, context=ctx)
    mm.precision = "mixed"
    fit = mm.fit(ds).get_fit()
    fit.accept_mixed_log_determinant = True
    assert abs(fit.log_determinant - ld64) <= 4e-6 * n, abs(fit.log_determinant - ld64)
    assert rel(fit.information, a64) <= 1e-8
    del fit
    ctx.close()
    # --- config 4's covariance on the bf16 x 3 path and on the fp32-MFMA fallback path
    ecef, h, temp = synthetic_stations(n, 11)
    cov, scale = temperature_covariance(ab)
    ds = ab.RegressionDataset(ab.FeatureSet(ecef, [scale(h)]), temp - temp.mean())
    ld64 = a64 = None
    for switch, inside in (("AGP_MIXED_F16", True), ("AGP_MIXED_BF16", False)):
        monkeypatch.setenv(switch, "0")
        ctx = make_ctx()
        if ld64 is None:
            f64 = ab.gp_from_covariance(cov, context=ctx).fit(ds)
            ld64, a64 = f64.get_fit().log_determinant, np.array(f64.get_fit().information)
            del f64
        mm = ab.gp_from_covariance(cov, context=ctx)
        mm.precision = "mixed"
        fit = mm.fit(ds).get_fit()
        fit.accept_mixed_log_determinant = True
        err = abs(fit.log_determinant - ld64)
        if inside:
            assert err <= 2e-6 * n, err
        else:
            assert err <= 5e-5 * abs(ld64), err / abs(ld64)
            assert err > 2e-6 * n, "the fp32 fallback is expected OUTSIDE the bar: if it is inside now, tighten include/albatross_amd.h"
        assert rel(fit.information, a64) <= 1e-8
        del fit
        ctx.close()


def _pitc_problem(n, m, seed):
    """bench covariance (benchmarks/bench_utils.h:61-65) on a sorted 1-D line, ~16 points per unit length, so that
    every group of 512 consecutive points spans ~32 length scales"""
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0., n / 16., n))
    y = np.sin(x) + 0.1 * np.cos(10. * x) + 0.1 * rng.standard_normal(n)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))
    u = np.linspace(x.min(), x.max(), m)
    return x, y, cov, u


def _rank_grouper(x, gs):
    sorted_x = np.sort(x)

    def grouper(f):
        r = np.searchsorted(sorted_x, np.asarray(f, dtype=np.float64).reshape(-1)) // gs
        return r if np.ndim(f) else int(r[0])
    grouper.vectorized = True
    return grouper


def test_config5_subproblem_matches_oracle(ctx):
    """The config-5 problem family at a size the oracle's literal QR restatement finishes in seconds: same kernel,
    point density, groups of 512; n = 8192, m = 128."""
    n, m, gs = 8192, 128, 512
    x, y, cov, u = _pitc_problem(n, m, 3)
    model = ab.sparse_gp_from_covariance(cov, _rank_grouper(x, gs), ab.FixedInducingPoints(u), "pitc", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    ds = ab.RegressionDataset(x, y)
    fm = model.fit(ds)
    keys = np.arange(n) // gs
    ofit = orc.OracleSparseFit(cov, x, keys, y, None, u, model.get_params()["measurement_nugget"], 1e-6)
    v = ofit.information
    assert np.abs(fm.get_fit().information - v).max() <= 1e-7 * np.abs(v).max()
    assert abs(fm.get_fit().nll - ofit.nll) <= 1e-8 * n
    xs = np.linspace(x.min(), x.max(), 64)
    om, ov = ofit.predict(xs)
    marg = fm.predict(xs).marginal()
    assert np.abs(marg.mean - om).max() <= 1e-8 * max(1., np.abs(om).max())
    assert np.abs(marg.covariance - ov).max() <= 1e-8 * ov.max()


def test_config5_n262144_pitc(ctx):
    n, m, gs = 262144, 2048, 512
    x, y, cov, u = _pitc_problem(n, m, n)
    grouper = _rank_grouper(x, gs)
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "pitc", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    mnug = model.get_params()["measurement_nugget"]
    fm = model.fit(ab.RegressionDataset(x, y))
    v = np.array(fm.get_fit().information)
    assert v.shape == (m,) and np.all(np.isfinite(v)) and np.isfinite(fm.get_fit().nll)

    # (1) the defining identity of the information vector (sparse_gp.hpp:206-210):
    #       (K_uu + K_uf A^-1 K_fu) v = K_uf A^-1 y,   A = blockdiag_g(K_gg - K_gu K_uu^-1 K_ug) (+ nuggets)
    # accumulated group by group from pieces that are tested on their own against the oracle: the device Gram
    # (agp_gram) and the dense factor / solve (agp_factor_create / agp_solve); the sums run in numpy.
    # A random quarter of the groups would not do (the identity is global), so ALL 512 groups are visited.
    Kuu = ctx.gram(cov, u)
    Kuu[np.diag_indices(m)] += 1e-6                                       # inducing_nugget, sparse_gp.hpp:676-677
    Kuu_f = ab.DenseFactor(Kuu, ctx)
    lhs = Kuu @ v
    rhs = np.zeros(m)
    for g in range(n // gs):
        sl = slice(g * gs, (g + 1) * gs)
        Kug = ctx.gram(cov, u, ab.Measurement(x[sl]))                      # m x gs
        Kgg = ctx.gram(cov, ab.Measurement(x[sl]))                         # as_measurements, :649-650
        A = Kgg - Kug.T @ Kuu_f.solve(Kug)                                 # K_ff - Q_ff, :688-690
        A[np.diag_indices(gs)] += mnug                                     # measurement_nugget, :692-696
        t = np.linalg.solve(A, np.column_stack([Kug.T @ v, y[sl]]))
        lhs += Kug @ t[:, 0]
        rhs += Kug @ t[:, 1]
    assert np.linalg.norm(lhs - rhs) <= 1e-7 * np.linalg.norm(rhs), np.linalg.norm(lhs - rhs) / np.linalg.norm(rhs)

    # (2) update == full fit (tests/test_sparse_gp.cc:293-371, the reference's bar is 1e-6): fit on the first half of
    # the groups, fold the second half in through _update_impl (sparse_gp.hpp:322-371)
    half = n // 2
    fm_half = model.fit(ab.RegressionDataset(x[:half], y[:half]))
    fm_upd = fm_half.update(ab.RegressionDataset(x[half:], y[half:]))
    assert rel(fm_upd.get_fit().information, v) <= 1e-6

    # (3) linearity in the targets: v(y1 + 2 y2) = v(y1) + 2 v(y2)
    rng = np.random.default_rng(1)
    y2 = np.cos(0.3 * x) + 0.1 * rng.standard_normal(n)
    v2 = np.array(model.fit(ab.RegressionDataset(x, y2)).get_fit().information)
    v12 = np.array(model.fit(ab.RegressionDataset(x, y + 2. * y2)).get_fit().information)
    assert rel(v12, v + 2. * v2) <= 1e-7

    # (4) predictions (mean = K_*u v, sparse_gp.hpp:447-458): finite, positive variances, and the mean is the cross
    # Gram applied to the information vector
    xs = np.linspace(x.min() + 1., x.max() - 1., 4096)
    marg = fm.predict(xs).marginal()
    assert np.all(marg.covariance > 0.) and np.all(np.isfinite(marg.mean))
    assert rel(marg.mean, ctx.gram(cov, xs, u) @ v) <= 1e-10
