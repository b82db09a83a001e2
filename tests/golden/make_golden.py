#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ (run HERE, in the build
container; /root/reference is not available on the GPU box).

Fixtures are DATA taken from the reference's own tests, or produced from the
reference's data generators restated in a few lines of C++ (libstdc++
mt19937 + normal/uniform distributions, exactly what the reference's test
helpers call) with expected outputs from numpy / scipy / mpmath — an
implementation independent of both the oracle and the HIP library:

  matern52.json, matern32.json  15x15 gpytorch oracle matrices
        /root/reference/tests/test_radial.cc:205-353,355-489   (tol 1e-15)
  mvn_nll.json                  scipy multivariate-normal known answer
        /root/reference/tests/test_evaluate.cc:20-44           (6.0946974293510134)
  radial_edges.json             k(pi,pi), k(0,1e32) edge cases
        /root/reference/tests/test_radial.cc:52-66
  distances.json                distance-metric known values
        /root/reference/tests/test_distance_metrics.cc:20-75
  toy_linear.json               make_toy_linear_data() + make_simple_covariance_function()
  toy_linear_mean.json          the same data under GPs with a LinearMean (test_models.h:75-96, test_gp.cc:344-371)
        tests/lib/albatross/test/test_utils.h:42-60, test_models.h:26-30,
        test_models.cc:134-148, tests/test_gp.cc:464-490
  bench512.json                 benchmarks/bench_utils.h:25-85 (N=512 1-D)
  measurement_algebra.json      tests/test_covariance_functions.cc:33-93
"""
import json
import os
import re
import subprocess
import tempfile

import numpy as np
import scipy.linalg
import scipy.stats

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


def dump(name, obj):
    with open(os.path.join(OUT, name), "w") as f:
        json.dump(obj, f, indent=1)
    print("wrote", name)


def parse_array(text, name):
    """Pull the numeric initialiser of `name{{...}};` out of a C++ test file."""
    start = text.index(name)
    end = text.index(";", start)
    return [float(t) for t in re.findall(r"-?\d+\.\d+e[+-]\d+", text[start:end])]


def matern_fixtures():
    text = open(f"{REF}/tests/test_radial.cc").read()
    x = [-100, -10, -5, -2, -1, -0.01, -1e-05, 0, 1e-05, 0.01, 1, 2, 5, 10, 100]
    for nm, var in (("matern52", "kOracleMatern52Y"), ("matern32", "kOracleMatern32Y")):
        vals = parse_array(text, var + "{{")
        assert len(vals) == 225, len(vals)
        K = np.array(vals).reshape(15, 15)
        assert np.abs(K - K.T).max() < 1e-15  # gpytorch output is symmetric to 1 ulp only
        dump(nm + ".json", {
            "source": f"tests/test_radial.cc {var} (gpytorch)", "x": x,
            "length_scale": 22.2, "sigma": 1.0, "K": K.tolist(), "tolerance_abs": 1e-15})


def mvn_fixture():
    x = np.array([-1., 0., 1.])
    cov = np.array([[1., .9, .8], [.9, 1., .9], [.8, .9, 1.]])
    ref_value = 6.0946974293510134  # tests/test_evaluate.cc:26,41
    sp = -scipy.stats.multivariate_normal.logpdf(x, np.zeros(3), cov)
    assert abs(sp - ref_value) < 1e-12
    dump("mvn_nll.json", {"source": "tests/test_evaluate.cc:20-44", "x": x.tolist(),
                          "cov": cov.tolist(), "nll": ref_value, "tolerance_reference": 1e-6,
                          "tolerance_build": 1e-12})


def edges_fixture():
    # tests/test_radial.cc:52-66 with each kernel's default sigma (radial.hpp:17)
    dump("radial_edges.json", {
        "source": "tests/test_radial.cc:52-66",
        "kernels": ["Exponential", "SquaredExponential", "Matern32", "Matern52"],
        "sigma": 10.0, "length_scale": 100000.0,
        "cases": [
            {"x": np.pi, "y": np.pi, "expect": "sigma^2", "exact": True},
            {"x": np.pi, "y": np.pi + 1e-16, "expect": "sigma^2", "tolerance_abs": 1e-8},
            {"x": 0.0, "y": 1e32, "expect": "0", "exact": True}]})
    dump("distances.json", {
        "source": "tests/test_distance_metrics.cc:20-75 (EXPECT_DOUBLE_EQ = 4 ulp)",
        "euclidean": [[[1, 1, 1], [1, 1, 2], 1.0], [[1, 1, 1], [2, 2, 2], 3 ** 0.5], [[2, 2, 2], [2, 2, 2], 0.0]],
        "radial": [[[0, 0, 1], [0, 0, 1], 0.0], [[0, 0, 1], [0, 1, 0], 0.0], [[0, 1, 1], [1, 0, 0], 2 ** 0.5 - 1]],
        "angular": [[[0, 0, 1], [0, 0, 1], 0.0], [[0, 0, 1], [0, 0, -1], np.pi], [[0, 0, 1], [0, 1, 0], np.pi / 2]]})


CXX_GEN = r"""
#include <cstdio>
#include <cmath>
#include <random>
#include <cstdint>
// make_toy_linear_data: tests/lib/albatross/test/test_utils.h:42-60
// random_features / random_dataset: benchmarks/bench_utils.h:25-34,74-85
int main() {
  {
    std::mt19937 gen; gen.seed(3);
    std::normal_distribution<> d{0., 0.1};
    std::printf("toy");
    for (int i = 0; i < 10; ++i) std::printf(" %.17g", 5. + double(i) * 1. + d(gen));
    std::printf("\n");
  }
  {
    std::mt19937 gen(4u);
    std::uniform_real_distribution<double> dist(0., 10.);
    std::printf("bench");
    for (int i = 0; i < 512; ++i) std::printf(" %.17g", dist(gen));
    std::printf("\n");
  }
  return 0;
}
"""


def run_cxx():
    with tempfile.TemporaryDirectory() as td:
        src = os.path.join(td, "gen.cc")
        open(src, "w").write(CXX_GEN)
        exe = os.path.join(td, "gen")
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, src])
        out = subprocess.check_output([exe], text=True)
    rows = {}
    for line in out.strip().splitlines():
        k, *v = line.split()
        rows[k] = np.array([float(t) for t in v])
    return rows


def se(xa, xb, ell, sigma):
    d = np.abs(np.asarray(xa)[:, None] - np.asarray(xb)[None, :])
    return sigma * sigma * np.exp(-(d / ell) ** 2)


def gp_expected(K, Kx, Kxx, y):
    """Independent numpy/scipy evaluation of gp.hpp:61-113."""
    c = scipy.linalg.cho_factor(K, lower=True)
    alpha = scipy.linalg.cho_solve(c, y)
    logdet = 2 * np.sum(np.log(np.diag(c[0])))
    mean = Kx.T @ alpha
    E = scipy.linalg.cho_solve(c, Kx)
    cov = Kxx - Kx.T @ E
    nll = 0.5 * (logdet + y @ alpha + len(y) * np.log(2 * np.pi))
    return alpha, logdet, mean, cov, nll


def toy_fixture(rows):
    y = rows["toy"]
    assert abs(y[0] - 5.0184128196853504) < 1e-15 and abs(y[9] - 13.881048261401078) < 1e-14
    x = np.arange(10, dtype=float)
    # make_simple_covariance_function(): SE(100,100) + measurement_only(IndependentNoise(0.1))
    K = se(x, x, 100., 100.) + 0.01 * np.eye(10)
    out = {"source": "test_utils.h:42-60 make_toy_linear_data(); test_models.h:26-30",
           "x": x.tolist(), "y": y.tolist(),
           "cov": {"squared_exponential_length_scale": 100., "sigma_squared_exponential": 100.,
                   "sigma_independent_noise": 0.1},
           "K_train": K.tolist(), "predictions": []}
    for xs in ([0.1, 1.1, 2.2], [-20., 0.01], x.tolist()):
        xs = np.array(xs)
        alpha, logdet, mean, cov, nll = gp_expected(K, se(x, xs, 100., 100.), se(xs, xs, 100., 100.), y)
        out["predictions"].append({"xs": xs.tolist(), "mean": mean.tolist(), "cov": cov.tolist()})
    out["information"] = alpha.tolist()
    out["log_det"] = float(logdet)
    out["nll"] = float(nll)
    out["tolerance_rel"] = 1e-7  # kappa(K) ~ 1e8: agreement ~ kappa * eps
    dump("toy_linear.json", out)


def toy_mean_fixture(rows):
    """The same toy data under a GP WITH a mean function: MakeGaussianProcessWithMean
    (tests/lib/albatross/test/test_models.h:75-96: simple covariance + LinearMean{offset 5, slope 1}) and the model of
    tests/test_gp.cc:344-371 (SE(2, 1) + measurement_only(IndependentNoise(0.1)) + the same LinearMean, predicted at
    {1.3, 4.2, 7.1}).  Expected values: numpy/scipy on y - m(x) (remove_from, gp.hpp:291-292) with m(x*) added to the
    predicted mean (add_to, gp.hpp:322,346,364); log-likelihood of the mean-removed targets (gp.hpp:442-451)."""
    y = rows["toy"]
    x = np.arange(10, dtype=float)
    a, b = 5., 1.  # offset, slope
    models = []
    for (ell, sigma, xs_list) in ((100., 100., ([0.1, 1.1, 2.2], [-20., 0.01])), (2., 1., ([1.3, 4.2, 7.1], [-3., 12.5]))):
        K = se(x, x, ell, sigma) + 0.01 * np.eye(10)
        z = y - (b * x + a)
        preds = []
        for xs in xs_list:
            xs = np.array(xs)
            alpha, logdet, mean, cov, nll = gp_expected(K, se(x, xs, ell, sigma), se(xs, xs, ell, sigma), z)
            preds.append({"xs": xs.tolist(), "mean": (mean + b * xs + a).tolist(), "cov": cov.tolist()})
        models.append({"cov": {"squared_exponential_length_scale": ell, "sigma_squared_exponential": sigma,
                               "sigma_independent_noise": 0.1},
                       "information": alpha.tolist(), "log_det": float(logdet), "nll": float(nll), "predictions": preds,
                       "tolerance_rel": 1e-7 if ell == 100. else 1e-10})
    dump("toy_linear_mean.json", {
        "source": "test_utils.h:42-60 make_toy_linear_data(5, 1); test_models.h:75-96; tests/test_gp.cc:344-371",
        "x": x.tolist(), "y": y.tolist(), "mean": {"offset": a, "slope": b}, "models": models})


def bench_fixture(rows):
    x = rows["bench"]
    y = np.sin(x) + 0.1 * np.cos(10. * x)
    K = se(x, x, 1., 1.) + 0.01 * (x[:, None] == x[None, :])
    xs = np.linspace(0., 10., 64)
    alpha, logdet, mean, cov, nll = gp_expected(K, se(x, xs, 1., 1.), se(xs, xs, 1., 1.) + 0.01 * np.eye(64), y)
    dump("bench512.json", {
        "source": "benchmarks/bench_utils.h:25-85 random_dataset(512, seed 4), bench_covariance()",
        "x": x.tolist(), "y": y.tolist(), "xs": xs.tolist(),
        "cov": {"squared_exponential_length_scale": 1., "sigma_squared_exponential": 1.,
                "sigma_independent_noise": 0.1},
        "information": alpha.tolist(), "log_det": float(logdet), "nll": float(nll),
        "mean": mean.tolist(), "variance": np.diag(cov).tolist(), "tolerance_rel": 1e-9})


def algebra_fixture():
    # tests/test_covariance_functions.cc:33-93: meas_noise = measurement_only(IndependentNoise(sigma)),
    # cov = SE + meas_noise evaluated on plain / Measurement<> arguments.
    dump("measurement_algebra.json", {
        "source": "tests/test_covariance_functions.cc:33-93",
        "sigma_noise": 0.1, "length_scale": 100.0, "sigma": 100.0, "x": 1.0, "y": 2.0,
        "rules": [
            "meas_noise(x, x) == 0", "meas_noise(Meas(x), x) == 0", "meas_noise(x, Meas(x)) == 0",
            "meas_noise(Meas(x), Meas(x)) == sigma_noise^2", "meas_noise(Meas(x), Meas(y)) == 0",
            "(se + meas_noise)(x, y) == se(x, y)",
            "(se + meas_noise)(Meas(x), Meas(x)) == se(x, x) + sigma_noise^2",
            "(se + meas_noise)(Meas(x), Meas(y)) == se(x, y)"]})


if __name__ == "__main__":
    matern_fixtures()
    mvn_fixture()
    edges_fixture()
    rows = run_cxx()
    toy_fixture(rows)
    toy_mean_fixture(rows)
    bench_fixture(rows)
    algebra_fixture()
