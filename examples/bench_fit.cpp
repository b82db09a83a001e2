// bench_fit.cpp — BASELINE config 3 through the C++ surface only (no Python in the loop):
// 3-D SquaredExponential(1, 1) + IndependentNoise(0.1), N training points (default 16384) with
// x ~ U[0, 10]^3 (std::mt19937(44)), y = sum_k sin x_k + 0.1 cos(10 x_0) (value
// distributions after benchmarks/bench_utils.h:25-85), host-resident inputs:
//   gp.fit(dataset) per step, then predict(M = 4096).mean() / .marginal().
// Usage: bench_fit [n] [steps]           prints CSV rows: fit,<ms>,<fits/s> / predict_mean,<ms> / predict_marginal,<ms>
#include <array>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <random>
#include <string>

#include <albatross_amd/albatross.hpp>

using namespace albatross;
using P3 = std::array<double, 3>;

static RegressionDataset<P3> make(int n, unsigned seed) {
  std::mt19937 gen(seed);
  std::uniform_real_distribution<double> u(0., 10.);
  std::vector<P3> x(static_cast<std::size_t>(n));
  Vector y(static_cast<std::size_t>(n));
  for (int i = 0; i < n; ++i) {
    for (int d = 0; d < 3; ++d) x[i][d] = u(gen);
    y[i] = std::sin(x[i][0]) + std::sin(x[i][1]) + std::sin(x[i][2]) + 0.1 * std::cos(10. * x[i][0]);
  }
  return RegressionDataset<P3>(x, y);
}

int main(int argc, char *argv[]) {
  const int n = argc > 1 ? std::stoi(argv[1]) : 16384, steps = argc > 2 ? std::stoi(argv[2]) : 10;
  const auto data = make(n, 44);
  const auto test = make(4096, 43);
  auto model = gp_from_covariance(SquaredExponential<EuclideanDistance>(1.0, 1.0) + IndependentNoise<P3>(0.1));
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto ms = [](auto a, auto b) { return std::chrono::duration<double, std::milli>(b - a).count(); };
  for (int w = 0; w < 2; ++w) (void)model.fit(data);
  const auto t0 = now();
  for (int s = 0; s < steps; ++s) (void)model.fit(data);
  const double fit_ms = ms(t0, now()) / steps;
  std::printf("fit,%.3f,%.3f\n", fit_ms, 1e3 / fit_ms);
  const auto fm = model.fit(data);
  (void)fm.predict(test.features).mean();
  auto t1 = now();
  const auto mean = fm.predict(test.features).mean();
  std::printf("predict_mean,%.3f\n", ms(t1, now()));
  (void)fm.predict(test.features).marginal();
  t1 = now();
  const auto marg = fm.predict(test.features).marginal();
  std::printf("predict_marginal,%.3f\n", ms(t1, now()));
  // sanity: the posterior mean at the training points reproduces the targets to within the noise level
  const auto back = fm.predict(std::vector<P3>(data.features.begin(), data.features.begin() + 64)).mean();
  double worst = 0.;
  for (int i = 0; i < 64; ++i) worst = std::max(worst, std::fabs(back[i] - data.targets.mean[i]));
  std::printf("train_residual_max,%.3g\n", worst);
  std::printf("loglik,%.10g\n", model.log_likelihood(data));
  return worst < 0.5 && mean.size() == 4096 && marg.covariance.size() == 4096 ? 0 : 1;
}
