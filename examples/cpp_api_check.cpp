// cpp_api_check.cpp — exercises the C++ drop-in surface the way the reference's
// tests do (tests/test_covariance_functions.cc:33-93, tests/test_gp.cc,
// tests/lib/albatross/test/test_models.h): 3-D features, scaling terms,
// parameter get/set, fit / predict variants / solve.  Prints "key,value" lines
// that tests/test_cpp_host_gpu.py checks against the oracle.
#include <array>
#include <cmath>
#include <cstdio>
#include <variant>
#include <random>

#include <albatross_amd/albatross.hpp>

using namespace albatross;
using P3 = std::array<double, 3>;

struct Elevation {  // a ScalingFunction, cf. examples/temperature_example/temperature_example_utils.h:60-90
  double center = 4.0, factor = 0.3;
  std::string get_name() const { return "elevation_scaling"; }
  ParameterStore get_params() const { return {{"elevation_scaling_center", center}, {"elevation_scaling_factor", factor}}; }
  void set_param(const std::string &n, double v) { (n == "elevation_scaling_center" ? center : factor) = v; }
  double _call_impl(const P3 &x) const { return 1. + factor * std::fmax(center - x[2], 0.); }
};

int main() {
  // --- measurement / noise algebra (exact) ---
  SquaredExponential<EuclideanDistance> radial;
  IndependentNoise<double> noise;
  auto meas_noise = measurement_only(noise);
  auto sum = radial + meas_noise;
  auto prod = meas_noise * radial;
  const double f = 0.;
  const Measurement<double> m(f);
  std::printf("algebra_meas_ff,%.17g\n", meas_noise.call(f, f));
  std::printf("algebra_meas_mm,%.17g\n", meas_noise.call(m, m));
  std::printf("algebra_meas_mf,%.17g\n", meas_noise.call(m, f));
  std::printf("algebra_sum_mm_minus_parts,%.17g\n", sum.call(m, m) - (radial.call(m, m) + meas_noise.call(m, m)));
  std::printf("algebra_prod_ff,%.17g\n", prod.call(f, f));
  std::printf("algebra_prod_mm_minus_parts,%.17g\n", prod.call(m, m) - radial.call(m, m) * meas_noise.call(m, m));

  // --- 3-D Matern + scaling term model, seeded data ---
  std::mt19937 gen(42);
  std::uniform_real_distribution<double> u(0., 10.);
  const int n = 400, ms = 50;
  std::vector<P3> x(n), xs(ms);
  Vector y(n);
  for (int i = 0; i < n; ++i) {
    x[i] = {u(gen), u(gen), u(gen)};
    y[i] = std::sin(x[i][0]) + std::sin(x[i][1]) + std::sin(x[i][2]) + 0.1 * std::cos(10. * x[i][0]);
  }
  for (int i = 0; i < ms; ++i) xs[i] = {u(gen), u(gen), u(gen)};
  auto cov = ScalingTerm<Elevation>() * Constant(0.5) + Matern52<EuclideanDistance>(2.0, 1.0) + IndependentNoise<P3>(0.1);
  auto model = gp_from_covariance(cov, "cpp_check");
  std::printf("name,%s\n", cov.get_name().c_str());
  model.set_param("sigma_constant", 0.7);
  for (const auto &kv : model.get_params()) std::printf("param_%s,%.17g\n", kv.first.c_str(), kv.second);
  RegressionDataset<P3> data(x, y);
  const auto fm = model.fit(data);
  std::printf("loglik,%.17g\n", model.log_likelihood(data));
  std::printf("logdet,%.17g\n", fm.get_fit().log_determinant);
  for (int i = 0; i < n; ++i) std::printf("x,%d,%.17g,%.17g,%.17g,%.17g\n", i, x[i][0], x[i][1], x[i][2], y[i]);
  for (int i = 0; i < ms; ++i) std::printf("xs,%d,%.17g,%.17g,%.17g\n", i, xs[i][0], xs[i][1], xs[i][2]);
  for (int i = 0; i < n; ++i) std::printf("info,%d,%.17g\n", i, fm.get_fit().information[i]);
  const auto pred = fm.predict(xs);
  const auto mean = pred.mean();
  const auto marg = pred.marginal();
  const auto joint = pred.joint();
  for (int i = 0; i < ms; ++i)
    std::printf("pred,%d,%.17g,%.17g,%.17g,%.17g\n", i, mean[i], marg.mean[i], marg.covariance[i], joint.covariance(i, i));
  double asym = 0.;
  for (int i = 0; i < ms; ++i)
    for (int j = 0; j < ms; ++j) asym = std::fmax(asym, std::fabs(joint.covariance(i, j) - joint.covariance(j, i)));
  std::printf("joint_asymmetry,%.17g\n", asym);
  // the same fit through a communicator (here of ONE rank: the RCCL bootstrap, the sharded entry point and the
  // replication of the factor run; with N processes every rank would pass the same dataset)
  {
    const Communicator comm(1, 0, Communicator::unique_id());
    const auto sfm = model.fit(data, comm);
    double di = 0., dp = 0.;
    for (int i = 0; i < n; ++i) di = std::fmax(di, std::fabs(sfm.get_fit().information[i] - fm.get_fit().information[i]));
    const auto smarg = sfm.predict(xs).marginal();
    for (int i = 0; i < ms; ++i)
      dp = std::fmax(dp, std::fmax(std::fabs(smarg.mean[i] - marg.mean[i]), std::fabs(smarg.covariance[i] - marg.covariance[i])));
    std::printf("sharded_ranks,%d\n", comm.size());
    std::printf("sharded_information_diff,%.17g\n", di);
    std::printf("sharded_prediction_diff,%.17g\n", dp);
    std::printf("sharded_logdet_diff,%.17g\n", std::fabs(sfm.get_fit().log_determinant - fm.get_fit().log_determinant));
  }
  // several datasets of one size in lock step (agp_fit_create_batch): the same numbers as one fit at a time
  {
    std::vector<RegressionDataset<P3>> batch;
    for (int b = 0; b < 3; ++b) {
      Vector yb(n);
      for (int i = 0; i < n; ++i) yb[i] = y[i] * (1. + 0.5 * b);
      batch.emplace_back(x, yb);
    }
    const auto fms = model.fit_batch(batch);
    double di = 0., dp = 0., dl = 0.;
    for (int b = 0; b < 3; ++b) {
      const auto one = model.fit(batch[b]);
      for (int i = 0; i < n; ++i) di = std::fmax(di, std::fabs(fms[b].get_fit().information[i] - one.get_fit().information[i]));
      const auto pb = fms[b].predict(xs).marginal(), po = one.predict(xs).marginal();
      for (int i = 0; i < ms; ++i) dp = std::fmax(dp, std::fmax(std::fabs(pb.mean[i] - po.mean[i]), std::fabs(pb.covariance[i] - po.covariance[i])));
      dl = std::fmax(dl, std::fabs(fms[b].get_fit().log_determinant - one.get_fit().log_determinant));
    }
    std::printf("batch_count,%d\n", (int)fms.size());
    std::printf("batch_information_diff,%.17g\n", di);
    std::printf("batch_prediction_diff,%.17g\n", dp);
    std::printf("batch_logdet_diff,%.17g\n", dl);
  }
  // CovarianceRepresentation::solve round trip: K (K^-1 e_0) = e_0
  Matrix rhs(n, 1);
  rhs(0, 0) = 1.;
  const Matrix sol = fm.get_fit().solve(rhs);
  const Matrix K = model.get_covariance()(as_measurements(x));  // the model holds its own copy of cov
  double resid = 0.;
  for (int i = 0; i < n; ++i) {
    double s = 0.;
    for (int j = 0; j < n; ++j) s += K(i, j) * sol(j, 0);
    resid = std::fmax(resid, std::fabs(s - (i == 0 ? 1. : 0.)));
  }
  std::printf("solve_residual,%.17g\n", resid);
  // leave-one-out fast path
  const auto loo = fm.get_fit().leave_one_out(y);
  const auto kinv = fm.get_fit().inverse_diagonal();
  for (int i = 0; i < n; ++i) std::printf("loo,%d,%.17g,%.17g,%.17g\n", i, loo.mean[i], loo.covariance[i], kinv[i]);
  // leave-one-GROUP-out through model.cross_validate() (tests/test_cross_validation.cc:56-72,202-321):
  // groups by the integer part of x[0] / 2.5; fast path vs refit-per-fold
  {
    const auto grouper = [](const P3 &p) { return static_cast<int>(p[0] / 2.5); };
    const auto cv = model.cross_validate().predict(data, grouper);
    const auto fast = cv.joints();
    const auto brute = cv.predictions();
    double dmean = 0., dcov = 0.;
    for (const auto &kv : fast) {
      const auto &b = brute.at(kv.first);
      for (std::size_t a = 0; a < kv.second.size(); ++a) {
        dmean = std::fmax(dmean, std::fabs(kv.second.mean[a] - b.mean[a]));
        for (std::size_t c = 0; c < kv.second.size(); ++c)
          dcov = std::fmax(dcov, std::fabs(kv.second.covariance(static_cast<std::int64_t>(a), static_cast<std::int64_t>(c)) -
                                           b.covariance(static_cast<std::int64_t>(a), static_cast<std::int64_t>(c))));
      }
    }
    std::printf("cv_groups,%zu\n", fast.size());
    std::printf("cv_mean_diff,%.17g\n", dmean);
    std::printf("cv_cov_diff,%.17g\n", dcov);
    const auto cvm = cv.marginal();
    for (const auto &kv : cv.indexer())
      for (std::size_t a = 0; a < kv.second.size(); ++a)
        std::printf("cv,%zu,%d,%.17g,%.17g\n", kv.second[a], kv.first, cvm.mean[kv.second[a]], cvm.covariance[kv.second[a]]);
    // LeaveOneOutGrouper == the LOO fast path
    const auto loo_cv = model.cross_validate().predict(data, LeaveOneOutGrouper()).marginal();
    double dl = 0.;
    for (int i = 0; i < n; ++i) dl = std::fmax(dl, std::fabs(loo_cv.mean[i] - loo.mean[i]) + std::fabs(loo_cv.covariance[i] - loo.covariance[i]));
    std::printf("cv_loo_diff,%.17g\n", dl);
  }
  // dense-matrix factor, dense NLL (tests/test_evaluate.cc:20-44) and update == full fit (tests/test_gp.cc:182-219)
  {
    Matrix c3(3, 3);
    const double vals[9] = {1., .9, .8, .9, 1., .9, .8, .9, 1.};
    for (int i = 0; i < 9; ++i) c3.data[i] = vals[i];
    std::printf("mvn_nll,%.17g\n", negative_log_likelihood(Vector{-1., 0., 1.}, c3));
    std::printf("mvn_logdet,%.17g\n", SerializableLDLT(c3).log_determinant());
    auto ucov = SquaredExponential<EuclideanDistance>(1.5, 1.0) + Constant(2.0);
    auto umodel = gp_from_covariance(ucov);
    std::vector<P3> xa(x.begin(), x.begin() + 250), xb(x.begin() + 250, x.begin() + 330), xc(x.begin() + 330, x.end());
    Vector ya(y.begin(), y.begin() + 250), yb(y.begin() + 250, y.begin() + 330), yc(y.begin() + 330, y.end());
    const auto full = umodel.fit(RegressionDataset<P3>(x, MarginalDistribution(y, Vector(n, 0.1))));
    const auto part = umodel.fit(RegressionDataset<P3>(xa, MarginalDistribution(ya, Vector(250, 0.1))));
    const auto up1 = update(part, RegressionDataset<P3>(xb, MarginalDistribution(yb, Vector(80, 0.1))));
    const auto up2 = up1.update(RegressionDataset<P3>(xc, MarginalDistribution(yc, Vector(yc.size(), 0.1))));
    const auto fj = full.predict(xs).joint();
    const auto uj = up2.predict(xs).joint();
    double dm = 0., dc = 0.;
    for (int i = 0; i < ms; ++i) {
      dm = std::fmax(dm, std::fabs(fj.mean[i] - uj.mean[i]));
      for (int j = 0; j < ms; ++j) dc = std::fmax(dc, std::fabs(fj.covariance(i, j) - uj.covariance(i, j)));
    }
    std::printf("update_mean_diff,%.17g\n", dm);
    std::printf("update_cov_diff,%.17g\n", dc);
  }
  // fit_from_prediction round trip (tests/test_gp.cc:343-371)
  {
    const std::vector<P3> pts(xs.begin(), xs.begin() + 5);
    const auto jp = fm.predict(pts).joint();
    const auto again = model.fit_from_prediction(pts, jp).predict_joint(pts);
    double dm = 0., dc = 0.;
    for (int i = 0; i < 5; ++i) {
      dm = std::fmax(dm, std::fabs(again.mean[i] - jp.mean[i]));
      for (int j = 0; j < 5; ++j) dc = std::fmax(dc, std::fabs(again.covariance(i, j) - jp.covariance(i, j)));
    }
    std::printf("from_prediction_mean_diff,%.17g\n", dm);
    std::printf("from_prediction_cov_diff,%.17g\n", dc);
  }
  // batched log likelihoods == single calls (the tuner's finite-difference gradient, tune/finite_difference.hpp:20-94)
  {
    std::vector<ParameterStore> sets(1);
    for (const auto &kv : model.get_params()) sets.push_back({{kv.first, kv.second + 1e-6}});
    const Vector batch = model.log_likelihoods(data, sets);
    double worst = 0.;
    for (std::size_t b = 0; b < sets.size(); ++b) {
      auto m2 = model;
      m2.set_param_values(sets[b]);
      worst = std::fmax(worst, std::fabs(batch[b] - m2.log_likelihood(data)));
    }
    std::printf("nll_batch_count,%zu\n", sets.size());
    std::printf("nll_batch_diff,%.17g\n", worst);
  }
  // sparse GP (tests/test_sparse_gp.cc:48-133 test_sanity shape): toy linear data, LeaveOneIntervalOut groups
  {
    std::vector<double> tx(10);
    Vector ty(10);
    const double toy_y[10] = {5.01841281968535, 5.899404483909093, 6.9658019644108045, 7.995527586269562,
                              9.027844091455382, 9.941910600141895, 10.984848510737773, 11.885256581824562,
                              12.938899996351795, 13.881048261401078};  // make_toy_linear_data(): tests/golden/toy_linear.json
    for (int i = 0; i < 10; ++i) { tx[i] = i; ty[i] = toy_y[i]; }
    auto scov = SquaredExponential<EuclideanDistance>(100., 100.) + measurement_only(IndependentNoise<double>(0.1));
    const auto grouper = [](const double &f) { return static_cast<long>(std::floor(f / 5.)); };
    auto sparse = sparse_gp_from_covariance(scov, grouper, UniformlySpacedInducingPoints(8), "sparse");
    sparse.set_param_value(details::inducing_nugget_name(), 1e-3);
    sparse.set_param_value(details::measurement_nugget_name(), 1e-12);
    RegressionDataset<double> tds(tx, ty);
    const auto sfit = sparse.fit(tds);
    const auto dfit = gp_from_covariance(scov, "direct").fit(tds);
    std::vector<double> txs(11);
    for (int i = 0; i < 11; ++i) txs[i] = 0.01 + (9.9 - 0.01) * i / 10.;
    const auto sp = sfit.predict_with_measurement_noise(txs).joint();
    const auto dp = dfit.predict_with_measurement_noise(txs).joint();
    double em = 0., ec = 0.;
    for (int i = 0; i < 11; ++i) {
      em += (sp.mean[i] - dp.mean[i]) * (sp.mean[i] - dp.mean[i]);
      for (int j = 0; j < 11; ++j) ec += (sp.covariance(i, j) - dp.covariance(i, j)) * (sp.covariance(i, j) - dp.covariance(i, j));
    }
    std::printf("sparse_mean_err,%.17g\n", std::sqrt(em));
    std::printf("sparse_cov_err,%.17g\n", std::sqrt(ec));
    std::printf("sparse_loglik,%.17g\n", sparse.log_likelihood(tds));
    for (std::size_t i = 0; i < sfit.get_fit().information.size(); ++i) std::printf("sparse_info,%zu,%.17g\n", i, sfit.get_fit().information[i]);
    // update == full fit (tests/test_sparse_gp.cc:293-371): hold out the first interval, then fold it back in
    {
      std::vector<double> xa, xb;
      Vector ya, yb;
      for (int i = 0; i < 10; ++i) (tx[i] < 5. ? xa : xb).push_back(tx[i]), (tx[i] < 5. ? ya : yb).push_back(ty[i]);
      struct FixedInducingPoints {  // tests/test_sparse_gp.cc:219-235: the same inducing points for every dataset
        std::vector<double> operator()(const decltype(scov) &, const std::vector<double> &) const {
          return UniformlySpacedInducingPoints(8)(0, std::vector<double>{0., 9.});
        }
      };
      auto fixed = sparse_gp_from_covariance(scov, grouper, FixedInducingPoints(), "sparse_fixed");
      fixed.set_param_value(details::inducing_nugget_name(), 1e-3);
      fixed.set_param_value(details::measurement_nugget_name(), 1e-12);
      const auto upd = fixed.fit(RegressionDataset<double>(xb, yb)).update(RegressionDataset<double>(xa, ya));
      const auto up = upd.predict_with_measurement_noise(txs).joint();
      const auto fp = fixed.fit(tds).predict_with_measurement_noise(txs).joint();
      double um = 0., uc = 0.;
      for (int i = 0; i < 11; ++i) {
        um = std::fmax(um, std::fabs(up.mean[i] - fp.mean[i]));
        for (int j = 0; j < 11; ++j) uc = std::fmax(uc, std::fabs(up.covariance(i, j) - fp.covariance(i, j)));
      }
      std::printf("sparse_update_mean_diff,%.17g\n", um);
      std::printf("sparse_update_cov_diff,%.17g\n", uc);
    }
    // rebase_inducing_points (tests/test_sparse_gp.cc:374-416): one point loses information, 51 points change nothing
    {
      double low = 0., high = 0.;
      const auto lp = rebase_inducing_points(sfit, std::vector<double>{5.}).predict_with_measurement_noise(txs).joint();
      std::vector<double> z(51);
      for (int i = 0; i < 51; ++i) z[i] = 0.01 + (9.9 - 0.01) * i / 50.;
      const auto hfit = rebase_inducing_points(sfit, z);
      const auto hp = hfit.predict_with_measurement_noise(txs).joint();
      for (int i = 0; i < 11; ++i) {
        low += (lp.mean[i] - sp.mean[i]) * (lp.mean[i] - sp.mean[i]);
        high += (hp.mean[i] - sp.mean[i]) * (hp.mean[i] - sp.mean[i]);
      }
      std::printf("sparse_rebase_low_diff,%.17g\n", std::sqrt(low));
      std::printf("sparse_rebase_high_diff,%.17g\n", std::sqrt(high));
      std::printf("sparse_rebase_high_rank,%lld\n", static_cast<long long>(hfit.numerical_rank()));
    }
    {  // the sparse fit through a communicator of one rank (agp_sparse_fit_create_sharded)
      const Communicator comm(1, 0, Communicator::unique_id());
      const auto cfit = sparse.fit(tds, comm);
      double d = 0.;
      for (std::size_t i = 0; i < sfit.get_fit().information.size(); ++i)
        d = std::fmax(d, std::fabs(cfit.get_fit().information[i] - sfit.get_fit().information[i]));
      std::printf("sparse_sharded_information_diff,%.17g\n", d);
    }
    const auto sm = sfit.predict(txs).marginal();
    for (int i = 0; i < 11; ++i) std::printf("sparse_pred,%d,%.17g,%.17g,%.17g\n", i, sfit.predict(txs).mean()[i], sm.mean[i], sm.covariance[i]);
  }
  // pivoted LDL^T fallback for a semi-definite covariance (duplicated observations, no noise)
  {
    std::vector<double> dx = {0.5, 1.5, 2.5, 3.5, 0.5, 2.5};
    Vector dy(dx.size());
    for (std::size_t i = 0; i < dx.size(); ++i) dy[i] = std::sin(dx[i]);
    auto pm = gp_from_covariance(SquaredExponential<EuclideanDistance>(1.5, 1.0) + Constant(0.3));
    const auto pf = fit_pivoted(pm, RegressionDataset<double>(dx, dy));
    const auto pj = pf.predict_joint(std::vector<double>{0.5, 2.0, 3.5});
    std::printf("pivoted_pred,%.17g,%.17g,%.17g\n", pj.mean[0], pj.mean[1], pj.mean[2]);
    std::printf("pivoted_var0,%.17g\n", pj.covariance(0, 0));
  }
  // LinearCombination features (core/linear_combination.hpp; tests/test_gp.cc:395-462): 12 plain observations
  // (single-term combinations) + "f(0.7) - f(2.9) = 0" and "mean of f at four points = 1" observed with variance 1e-5
  {
    using LC = LinearCombination<double>;
    std::vector<LC> feats;
    Vector ty, tv;
    for (int i = 0; i < 12; ++i) {
      const double xi = 0.4 * i;
      feats.push_back(LC({xi}));
      ty.push_back(std::sin(xi) + 1.5);
      tv.push_back(0.05 * 0.05);
    }
    feats.push_back(LC({0.7, 2.9}, Vector{1., -1.}));
    ty.push_back(0.); tv.push_back(1e-5);
    feats.push_back(LC({0.4, 1.9, 3.3, 4.6}, Vector{0.25, 0.25, 0.25, 0.25}));
    ty.push_back(1.); tv.push_back(1e-5);
    auto lm = gp_from_covariance_and_mean(SquaredExponential<EuclideanDistance>(1.2, 2.0) + Constant(3.0) +
                                              measurement_only(IndependentNoise<double>(0.2)),
                                          LinearMean{0.3, -1.0});
    const Matrix K = lm.get_covariance()(as_measurements(feats));
    for (std::size_t i = 0; i < feats.size(); ++i) std::printf("lc_gram_row,%zu,%.17g,%.17g,%.17g\n", i, K(12, (long)i), K(13, (long)i), K((long)i, (long)i));
    const auto lf = lm.fit(RegressionDataset<LC>(feats, MarginalDistribution(ty, tv)));
    for (std::size_t i = 0; i < lf.information.size(); ++i) std::printf("lc_info,%zu,%.17g\n", i, lf.information[i]);
    const std::vector<double> pts = {0.7, 2.9, 0.4, 1.9, 3.3, 4.6};
    const auto lj = lf.predict_joint(pts);
    for (std::size_t i = 0; i < pts.size(); ++i) std::printf("lc_pred,%zu,%.17g,%.17g\n", i, lj.mean[i], lj.covariance((long)i, (long)i));
    const auto lq = lf.predict_joint(std::vector<LC>{feats[12], feats[13]});
    std::printf("lc_constraints,%.17g,%.17g,%.17g,%.17g\n", lq.mean[0], lq.mean[1], lq.covariance(0, 0), lq.covariance(1, 1));
  }
  // variant features (VariantForwarder, callers.hpp:419-544) with the reference's dispatch table
  // (tests/lib/albatross/test/test_covariance_utils.h:42-62: (X,X)=1, (X,Y)=3, (Y,Y)=5, (W,W)=7, (V,V)=11, else 0)
  {
    using A2 = std::array<double, 2>;
    using A3 = std::array<double, 3>;
    using A4 = std::array<double, 4>;
    using F = std::variant<double, A2, A3, A4>;  // alternatives X, Y, W, V
    auto c = [](double v) { return Constant(std::sqrt(v)); };
    auto has_multiple = only_for_alternatives<0>(c(1.)) + only_for_alternatives<0, 1>(c(3.)) + only_for_alternatives<1>(c(5.)) +
                        only_for_alternatives<2>(c(7.)) + only_for_alternatives<3>(c(11.));
    const std::vector<F> fs = {F(0.), F(A2{0., 0.}), F(A3{0., 0., 0.}), F(A4{0., 0., 0., 0.})};
    const Matrix K = has_multiple(fs);
    for (int i = 0; i < 4; ++i) std::printf("variant_gram,%d,%.17g,%.17g,%.17g,%.17g\n", i, K(i, 0), K(i, 1), K(i, 2), K(i, 3));
    // a GP over two kinds of observations: 1-D positions (alternative 0) and 2-D points (alternative 1)
    using G = std::variant<double, A2>;
    std::vector<G> gx;
    Vector gy;
    for (int i = 0; i < 30; ++i) {
      if (i % 2 == 0) gx.push_back(G(0.3 * i)); else gx.push_back(G(A2{0.2 * i, 1. + 0.1 * i}));
      gy.push_back(std::sin(0.4 * i));
    }
    auto gcov = only_for_alternatives<0>(SquaredExponential<EuclideanDistance>(1.5, 1.0)) +
                only_for_alternatives<1>(Matern52<EuclideanDistance>(2.0, 0.8)) + Constant(0.5) + IndependentNoise<G>(0.1);
    auto gm = gp_from_covariance(gcov);
    const auto gf = gm.fit(RegressionDataset<G>(gx, gy));
    for (std::size_t i = 0; i < gx.size(); ++i) std::printf("variant_info,%zu,%.17g\n", i, gf.get_fit().information[i]);
    // a ScalingTerm whose function only knows the first alternative scales the other by 1 (scaling_function.hpp:92-112)
    struct TimeScale {
      std::string get_name() const { return "time_scale"; }
      ParameterStore get_params() const { return {}; }
      void set_param(const std::string &, double) {}
      double _call_impl(const double &t) const { return 1. + 0.1 * t; }
    };
    const auto sc = ScalingTerm<TimeScale>(TimeScale()) * Constant(0.7);
    const Matrix Ks = sc(std::vector<G>{G(2.0), G(A2{1., 2.}), G(5.0)});
    std::printf("variant_scaling,%.17g,%.17g,%.17g,%.17g\n", Ks(0, 0), Ks(0, 1), Ks(1, 1), Ks(0, 2));
    const auto gp = gf.predict(std::vector<G>{G(2.5), G(A2{1., 2.})}).marginal();
    std::printf("variant_pred,%.17g,%.17g,%.17g,%.17g\n", gp.mean[0], gp.mean[1], gp.covariance[0], gp.covariance[1]);
  }
  // a singular covariance is reported, not silently factored
  try {
    std::vector<double> dup = {0., 0., 1.};
    gp_from_covariance(SquaredExponential<EuclideanDistance>(1., 1.)).fit(RegressionDataset<double>(dup, Vector{0., 0., 0.}));
    std::printf("singular,not_reported\n");
  } catch (const std::runtime_error &e) {
    std::printf("singular,%s\n", e.what());
  }
  return 0;
}
