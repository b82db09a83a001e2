// sinc_example.cpp — the reference's examples/sinc_example.cc (BASELINE config 1:
// 1-D SquaredExponential GP, N = 256) written against the drop-in C++ surface.
// Same data generator (examples/example_utils.h:41-96: std::default_random_engine,
// x ~ U[-10, 23], y = x sqrt(2) + 3.14159 + 10 sinc(x - 3) + N(0, 1)), same models
// (sinc_example.cc:71-88), same output (predict_with_measurement_noise on 161
// grid points over [-20, 33], example_utils.h:146-170) — as CSV on stdout:
//   train,<x>,<y>            one line per training point
//   params,<name>,<value>
//   loglik,<value>
//   pred,<x>,<mean>,<variance>,<truth>
// Usage: sinc_example [radial|radial_only] [n]
#include <cmath>
#include <cstdio>
#include <random>
#include <string>

#include <albatross_amd/albatross.hpp>

static double sinc(double x) { return x == 0 ? 1. : std::sin(x) / x; }
static double truth(double x) { return x * std::sqrt(2.) + 3.14159 + 10. * sinc(x - 3.); }

static std::vector<double> random_points_on_line(int n, double low, double high) {
  std::default_random_engine generator;
  std::uniform_real_distribution<double> distribution(low, high);
  std::vector<double> xs;
  for (int i = 0; i < n; i++) xs.push_back(distribution(generator));
  return xs;
}

static std::vector<double> uniform_points_on_line(std::size_t n, double low, double high) {
  std::vector<double> xs;
  for (std::size_t i = 0; i < n; i++) xs.push_back(low + (double)i / (double)(n - 1) * (high - low));
  return xs;
}

static albatross::RegressionDataset<double> create_train_data(int n, double low, double high, double noise_sd) {
  auto xs = random_points_on_line(n, low, high);
  std::default_random_engine generator;
  std::normal_distribution<double> noise_distribution(0., noise_sd);
  albatross::Vector ys(xs.size());
  for (std::size_t i = 0; i < xs.size(); i++) ys[i] = truth(xs[i]) + noise_distribution(generator);
  return albatross::RegressionDataset<double>(xs, ys);
}

template <typename ModelType>
static void run_model(ModelType &model, albatross::RegressionDataset<double> &data, double low, double high) {
  for (const auto &kv : model.get_params()) std::printf("params,%s,%.17g\n", kv.first.c_str(), kv.second);
  const auto fit_model = model.fit(data);
  std::printf("loglik,%.17g\n", model.log_likelihood(data));
  const auto grid_xs = uniform_points_on_line(161, low - 10., high + 10.);
  const auto prediction = fit_model.predict_with_measurement_noise(grid_xs).marginal();
  for (std::size_t i = 0; i < grid_xs.size(); ++i)
    std::printf("pred,%.17g,%.17g,%.17g,%.17g\n", grid_xs[i], prediction.mean[i], prediction.covariance[i], truth(grid_xs[i]));
}

int main(int argc, char *argv[]) {
  const std::string mode = argc > 1 ? argv[1] : "radial";
  const int n = argc > 2 ? std::stoi(argv[2]) : 256;
  const double low = -10., high = 23., meas_noise_sd = 1.;
  using namespace albatross;
  RegressionDataset<double> data = create_train_data(n, low, high, meas_noise_sd);
  for (std::size_t i = 0; i < data.features.size(); ++i) std::printf("train,%.17g,%.17g\n", data.features[i], data.targets.mean[i]);

  IndependentNoise<double> indep_noise(meas_noise_sd);
  if (mode == "radial_only") {
    const SquaredExponential<EuclideanDistance> squared_exponential(3.5, 100.);
    auto cov = squared_exponential + measurement_only(indep_noise);
    auto model = gp_from_covariance(cov);
    run_model(model, data, low, high);
  } else {
    const Polynomial<1> linear(100.);
    const SquaredExponential<EuclideanDistance> squared_exponential(3.5, 5.7);
    auto cov = linear + squared_exponential + measurement_only(indep_noise);
    auto model = gp_from_covariance(cov);
    run_model(model, data, low, high);
  }
  return 0;
}
