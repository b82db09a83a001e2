"""GPU tests of the C++ drop-in surface (include/albatross_amd/albatross.hpp):
build the examples with g++ against the C-ABI library, run them, and check
their output against the oracle on the same data."""
import os
import subprocess

import numpy as np
import pytest

import albatross_amd as ab
from oracle import oracle_py as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EX = os.path.join(ROOT, "examples")


def golden_toy_y():
    import json
    with open(os.path.join(os.path.dirname(__file__), "golden", "toy_linear.json")) as fh:
        return json.load(fh)["y"]


def run(binary, *args):
    subprocess.check_call(["make", "-s", "-C", EX])
    out = subprocess.check_output([os.path.join(EX, binary), *args], text=True)
    rows = {}
    for line in out.strip().splitlines():
        key, *vals = line.split(",")
        rows.setdefault(key, []).append(vals)
    return rows


def test_headers_compile_without_gpu():
    """not a gpu test: the C++ surface builds with plain g++ and links the C-ABI"""
    subprocess.check_call(["make", "-s", "-C", EX, "clean"])
    subprocess.check_call(["make", "-s", "-C", EX])
    assert os.path.exists(os.path.join(EX, "sinc_example"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["radial", "radial_only"])
def test_sinc_example_config1(mode):
    """BASELINE config 1: sinc_example, 1-D SquaredExponential GP, N = 256."""
    rows = run("sinc_example", mode, "256")
    train = np.array(rows["train"], dtype=float)
    pred = np.array(rows["pred"], dtype=float)
    assert train.shape == (256, 2) and pred.shape == (161, 4)
    x, y = train[:, 0], train[:, 1]
    if mode == "radial":
        cov = ab.Polynomial(1, 100.) + ab.SquaredExponential(3.5, 5.7) + ab.measurement_only(ab.IndependentNoise(1.0))
    else:
        cov = ab.SquaredExponential(3.5, 100.) + ab.measurement_only(ab.IndependentNoise(1.0))
    params = {k: float(v) for k, v in rows["params"]}
    assert params == cov.get_params()
    ofit = orc.OracleFit(cov, x, y)
    om, ov = ofit.predict_marginal(pred[:, 0], xs_meas=True)  # predict_with_measurement_noise
    assert np.abs(pred[:, 1] - om).max() <= 1e-7 * np.abs(om).max()
    assert np.abs(pred[:, 2] - ov).max() <= 1e-7 * np.abs(ov).max()
    assert abs(float(rows["loglik"][0][0]) + orc.nll(cov, x, y)) <= 1e-6 * 256
    # the GP should recover the truth inside the data range (the example's purpose)
    inside = (pred[:, 0] > -8) & (pred[:, 0] < 21)
    assert np.abs(pred[inside, 1] - pred[inside, 3]).max() < 3.0


@pytest.mark.gpu
def test_cpp_api_matches_oracle():
    rows = run("cpp_api_check")
    one = {k: v[0][0] for k, v in rows.items() if len(v) == 1 and len(v[0]) == 1}
    # measurement / noise algebra is exact (tests/test_covariance_functions.cc:33-93)
    assert float(one["algebra_meas_ff"]) == 0. and float(one["algebra_meas_mf"]) == 0.
    assert float(one["algebra_meas_mm"]) == 0.1 * 0.1
    assert float(one["algebra_sum_mm_minus_parts"]) == 0. and float(one["algebra_prod_mm_minus_parts"]) == 0.
    assert float(one["algebra_prod_ff"]) == 0.
    assert one["name"] == "(((elevation_scaling*constant)+matern_52[euclidean_distance])+independent_noise)"
    assert float(one["param_sigma_constant"]) == 0.7 and float(one["param_matern_52_length_scale"]) == 2.0

    class Elevation(ab.ScalingFunction):
        _params = {"elevation_scaling_center": 4.0, "elevation_scaling_factor": 0.3}

        def get_name(self):
            return "elevation_scaling"

        def _call_impl(self, c):
            return 1. + 0.3 * np.maximum(4.0 - np.asarray(c)[:, 2], 0.)

    cov = ab.ScalingTerm(Elevation()) * ab.Constant(0.7) + ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    xr = np.array(rows["x"], dtype=float)
    x, y = xr[:, 1:4], xr[:, 4]
    xs = np.array(rows["xs"], dtype=float)[:, 1:4]
    ofit = orc.OracleFit(cov, x, y)
    info = np.array(rows["info"], dtype=float)[:, 1]
    assert np.abs(info - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
    assert abs(float(one["logdet"]) - ofit.log_determinant) <= 1e-6 * 400
    assert abs(float(one["loglik"]) + orc.nll(cov, x, y)) <= 1e-6 * 400
    pred = np.array(rows["pred"], dtype=float)
    om, ov = ofit.predict_marginal(xs)
    assert np.abs(pred[:, 1] - om).max() <= 1e-8 * np.abs(om).max()
    assert np.abs(pred[:, 2] - om).max() <= 1e-8 * np.abs(om).max()
    assert np.abs(pred[:, 3] - ov).max() <= 1e-8 and np.abs(pred[:, 4] - ov).max() <= 1e-8
    loo = np.array(rows["loo"], dtype=float)
    lm, lv = ofit.loo_marginal(y)
    assert np.abs(loo[:, 1] - lm).max() <= 1e-8 * np.abs(lm).max() and np.abs(loo[:, 2] - lv).max() <= 1e-8 * lv.max()
    assert np.abs(loo[:, 3] - ofit.inverse_diagonal()).max() <= 1e-8 * ofit.inverse_diagonal().max()
    # LinearCombination features: the C++ surface against the Python mirror (itself checked against the defining
    # double sum over the oracle's covariance in tests/test_linear_combination_gpu.py)
    lc_feats = [ab.LinearCombination([0.4 * i]) for i in range(12)] + [
        ab.LinearCombination([0.7, 2.9], [1., -1.]), ab.LinearCombination([0.4, 1.9, 3.3, 4.6], [0.25] * 4)]
    lc_y = np.array([np.sin(0.4 * i) + 1.5 for i in range(12)] + [0., 1.])
    lc_v = np.array([0.05 ** 2] * 12 + [1e-5, 1e-5])
    lc_cov = ab.SquaredExponential(1.2, 2.0) + ab.Constant(3.0) + ab.measurement_only(ab.IndependentNoise(0.2))
    lc_model = ab.gp_from_covariance_and_mean(lc_cov, ab.LinearMean(0.3, -1.0))
    Kpy = lc_model._ctx().gram(lc_cov, ab.Measurement(lc_feats))
    grow = np.array(rows["lc_gram_row"], dtype=float)
    assert np.abs(grow[:, 1] - Kpy[12]).max() <= 1e-13 * np.abs(Kpy).max()
    assert np.abs(grow[:, 2] - Kpy[13]).max() <= 1e-13 * np.abs(Kpy).max()
    assert np.abs(grow[:, 3] - np.diag(Kpy)).max() <= 1e-13 * np.abs(Kpy).max()
    lc_fm = lc_model.fit(ab.RegressionDataset(lc_feats, ab.MarginalDistribution(lc_y, lc_v)))
    lc_info = np.array(rows["lc_info"], dtype=float)[:, 1]
    assert np.abs(lc_info - lc_fm.get_fit().information).max() <= 1e-9 * np.abs(lc_info).max()
    lcp = np.array(rows["lc_pred"], dtype=float)
    want = lc_fm.predict([0.7, 2.9, 0.4, 1.9, 3.3, 4.6]).joint()
    assert np.abs(lcp[:, 1] - want.mean).max() <= 1e-9 * np.abs(want.mean).max()
    assert np.abs(lcp[:, 2] - np.diag(want.covariance)).max() <= 1e-9
    cm0, cm1, cv0, cv1 = (float(v) for v in rows["lc_constraints"][0])
    wq = lc_fm.predict([lc_feats[12], lc_feats[13]]).joint()   # predictions AT the two combined features
    assert abs(cm0 - wq.mean[0]) < 1e-9 and abs(cm1 - wq.mean[1]) < 1e-9
    assert abs(cv0 - wq.covariance[0, 0]) < 1e-9 and abs(cv1 - wq.covariance[1, 1]) < 1e-9
    assert abs((lcp[0, 1] - lcp[1, 1]) - cm0) < 1e-9             # = the same combination of the point predictions
    # variant features: the reference's dispatch table and a two-kinds-of-observations GP against the oracle
    vg = np.array(rows["variant_gram"], dtype=float)[:, 1:]
    want = np.array([[1., 3., 0., 0.], [3., 5., 0., 0.], [0., 0., 7., 0.], [0., 0., 0., 11.]])
    assert np.abs(vg - want).max() <= 1e-14 and np.array_equal(vg == 0., want == 0.)
    alt = [i % 2 for i in range(30)]
    vals = [0.3 * i if i % 2 == 0 else [0.2 * i, 1. + 0.1 * i] for i in range(30)]
    vfeats = ab.VariantFeatures(alt, vals)
    vy = np.array([np.sin(0.4 * i) for i in range(30)])
    vcov = (ab.only_for_alternatives(ab.SquaredExponential(1.5, 1.0), 0) + ab.only_for_alternatives(ab.Matern52(2.0, 0.8), 1)
            + ab.Constant(0.5) + ab.IndependentNoise(0.1))
    vfit = orc.OracleFit(vcov, vfeats, vy)
    vinfo = np.array(rows["variant_info"], dtype=float)[:, 1]
    assert np.abs(vinfo - vfit.information).max() <= 1e-8 * np.abs(vfit.information).max()
    vm, vv = vfit.predict_marginal(ab.VariantFeatures([0, 1], [2.5, [1., 2.]]))
    vp = [float(v) for v in rows["variant_pred"][0]]
    assert abs(vp[0] - vm[0]) <= 1e-8 and abs(vp[1] - vm[1]) <= 1e-8 and abs(vp[2] - vv[0]) <= 1e-8 and abs(vp[3] - vv[1]) <= 1e-8
    vs = [float(v) for v in rows["variant_scaling"][0]]   # f(2.0) = 1.2, f(5.0) = 1.5, the 2-D alternative scales by 1
    assert np.allclose(vs, [1.2 * 1.2 * 0.49, 1.2 * 0.49, 0.49, 1.2 * 1.5 * 0.49], rtol=1e-14)
    # leave-one-group-out: fast path == refit per fold (the brute-force predict(test) is the latent
    # prediction: the held-out noise term only appears in the fast path's covariance diagonal)
    assert int(one["cv_groups"]) == 4 and float(one["cv_mean_diff"]) < 1e-7 and float(one["cv_cov_diff"]) < 1e-7
    cvr = np.array(rows["cv"], dtype=float)
    groups = [list(map(int, cvr[cvr[:, 1] == k, 0])) for k in sorted(set(cvr[:, 1]))]
    want = ofit.held_out(y, groups)
    for (wm, wv), g in zip(want, groups):
        sel = np.array([np.nonzero(cvr[:, 0] == i)[0][0] for i in g])
        assert np.abs(cvr[sel, 2] - wm).max() <= 1e-8 * np.abs(wm).max() and np.abs(cvr[sel, 3] - wv).max() <= 1e-8
    assert float(one["cv_loo_diff"]) < 1e-9
    assert float(one["from_prediction_mean_diff"]) < 1e-6 and float(one["from_prediction_cov_diff"]) < 1e-6
    # GaussianProcessRegression::fit(dataset, Communicator): the sharded entry point + replicated factor (one rank here)
    assert int(one["sharded_ranks"]) == 1
    assert float(one["sharded_information_diff"]) < 1e-10 and float(one["sharded_prediction_diff"]) < 1e-10
    assert float(one["sharded_logdet_diff"]) < 1e-9
    # GaussianProcessRegression::fit_batch: three datasets in lock step (agp_fit_create_batch) == one fit at a time
    assert int(one["batch_count"]) == 3 and float(one["batch_information_diff"]) < 1e-9
    assert float(one["batch_prediction_diff"]) < 1e-10 and float(one["batch_logdet_diff"]) < 1e-9
    assert int(one["nll_batch_count"]) == 7 and float(one["nll_batch_diff"]) < 1e-8  # agp_nll_batch == agp_nll
    # sparse GP through the C++ surface: close to the direct GP (test_sparse_gp.cc:115-133 thresholds) and
    # equal to the oracle's QR-based restatement
    assert float(one["sparse_mean_err"]) < 1e-2 and float(one["sparse_cov_err"]) < 1e-2
    assert float(one["sparse_update_mean_diff"]) < 1e-6 and float(one["sparse_update_cov_diff"]) < 1e-6
    # rebase_inducing_points through the C++ surface, thresholds of tests/test_sparse_gp.cc:374-416
    assert float(one["sparse_rebase_low_diff"]) > 10. and float(one["sparse_rebase_high_diff"]) < 1e-6
    assert int(one["sparse_rebase_high_rank"]) < 51
    assert float(one["sparse_sharded_information_diff"]) < 1e-9  # SparseGaussianProcessRegression::fit(dataset, comm)
    tx = np.arange(10.)
    ty = np.array(golden_toy_y())
    scov = ab.SquaredExponential(100., 100.) + ab.measurement_only(ab.IndependentNoise(0.1))
    u8 = np.linspace(0., 9., 8)
    so = orc.OracleSparseFit(scov, tx, np.floor(tx / 5.).astype(np.int64), ty, None, u8, 1e-12, 1e-3)
    sinfo = np.array(rows["sparse_info"], dtype=float)[:, 1]
    assert np.abs(sinfo - so.information).max() <= 1e-6 * np.abs(so.information).max()
    assert abs(float(one["sparse_loglik"]) + so.nll) <= 1e-7
    sp = np.array(rows["sparse_pred"], dtype=float)
    txs = 0.01 + (9.9 - 0.01) * np.arange(11) / 10.
    smean, svar = so.predict(txs)
    assert np.abs(sp[:, 1] - smean).max() <= 1e-7 and np.abs(sp[:, 2] - smean).max() <= 1e-7
    assert np.abs(sp[:, 3] - svar).max() <= 1e-7
    # semi-definite model through the pivoted factor: interpolates the (duplicated) observations
    pp = [float(v) for v in rows["pivoted_pred"][0]]
    assert abs(pp[0] - np.sin(0.5)) < 1e-6 and abs(pp[2] - np.sin(3.5)) < 1e-6 and abs(float(one["pivoted_var0"])) < 1e-6
    assert abs(float(one["mvn_nll"]) - 6.0946974293510134) < 1e-12  # tests/test_evaluate.cc:26,41
    assert abs(float(one["mvn_logdet"]) - np.linalg.slogdet(np.array([[1, .9, .8], [.9, 1, .9], [.8, .9, 1.]]))[1]) < 1e-13
    assert float(one["update_mean_diff"]) < 1e-8 and float(one["update_cov_diff"]) < 1e-6  # tests/test_gp.cc:213
    assert float(one["joint_asymmetry"]) == 0.
    assert float(one["solve_residual"]) < 1e-10
    assert "not positive definite" in one["singular"] and "pivot 1" in one["singular"]


@pytest.mark.gpu
def test_temperature_example_config4_kernel():
    """BASELINE config 4's spatial kernel (examples/temperature_example) in fp64 on synthetic
    stations: a user feature type (Station: ECEF coords, equality by ECEF, elevation-dependent
    ScalingTerm) through the C++ surface vs the oracle."""
    rows = run("temperature_example", "1500", "100", "mixed")
    its, res, dmean, dvar = (float(v) for v in rows["mixed"][0])
    assert res <= 1e-12 and dmean <= 1e-8 and dvar <= 1e-4, rows["mixed"]  # mixed-precision fit of the same model
    st = np.array(rows["station"], dtype=float)
    pr = np.array(rows["pred"], dtype=float)
    assert rows["name"][0][0] == ("(((elevation_scaled*constant)+independent_noise)+"
                                  "(exponential[angular_distance]*squared_exponential[radial_distance]))")

    class Elev(ab.ScalingFunction):
        def _call_impl(self, c):
            raise AssertionError("scale columns are supplied explicitly")

    cov = ab.ScalingTerm(Elev()) * ab.Constant(5.07288) + ab.IndependentNoise(1.75027) \
        + ab.Exponential(1.10298, 1.0, ab.AngularDistance()) * ab.SquaredExponential(5835.56, 13.913, ab.RadialDistance())
    scale = lambda h: 1. + 0.000153439 * np.maximum(0., 4446.5 - h)
    train = ab.FeatureSet(st[:, :3], [scale(st[:, 3])])
    test = ab.FeatureSet(pr[:, :3], [scale(pr[:, 3])])
    ofit = orc.OracleFit(cov, train, st[:, 4])
    om, ov = ofit.predict_marginal(test)
    assert np.abs(pr[:, 4] - om).max() <= 1e-8 * np.abs(om).max()
    assert np.abs(pr[:, 5] - ov).max() <= 1e-8 * np.abs(ov).max()
    assert abs(float(rows["loglik"][0][0]) + orc.nll(cov, train, st[:, 4])) <= 1e-6 * 1500
    # the prediction placed ON a station: IndependentNoise<Station> is not measurement-only, so the
    # equal feature (x == y by ECEF) shares its noise and the observed value is reproduced exactly
    assert abs(pr[0, 5]) < 1e-8 and abs(pr[0, 4] - st[3, 4]) < 1e-8


@pytest.mark.gpu
def test_cpp_bench_fit_runs():
    """examples/bench_fit.cpp (BASELINE config 3 through the C++ surface only) at a small size: exit code 0 means
    the fit reproduces its targets; the log-likelihood is checked against the oracle."""
    rows = run("bench_fit", "600", "2")
    assert float(rows["fit"][0][0]) > 0.
    # the data come from std::mt19937(44): same engine in numpy (init_genrand seeding), and libstdc++'s
    bits = np.random.MT19937()
    bits._legacy_seeding(44)
    # std::uniform_real_distribution<double> on mt19937 consumes two 32-bit words per draw (generate_canonical)
    raw = bits.random_raw(600 * 3 * 2).astype(np.float64)
    u = (raw[0::2] + raw[1::2] * 4294967296.0) / 18446744073709551616.0 * 10.
    x = u.reshape(600, 3)
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    assert abs(float(rows["loglik"][0][0]) + orc.nll(cov, x, y)) <= 1e-6 * 600
