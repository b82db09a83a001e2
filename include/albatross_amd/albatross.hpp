// albatross.hpp — C++ host-side mirror of albatross's dense-GP call surface
// over the C-ABI of the MI355X engine (include/albatross_amd.h).
//
//   #include <albatross_amd/albatross.hpp>     // instead of <albatross/GP>
//   using namespace albatross;
//   auto cov = SquaredExponential<EuclideanDistance>(3.5, 5.7) +
//              measurement_only(IndependentNoise<double>(1.0));
//   auto model = gp_from_covariance(cov);
//   auto fit_model = model.fit(dataset);                 // RegressionDataset<double>
//   auto pred = fit_model.predict(xs).marginal();        // .mean() / .joint()
//   double ll = model.log_likelihood(dataset);
//
// Same names, argument meaning and composition rules as the reference
// (include/albatross/src/...): covariance_functions/*.hpp, models/gp.hpp:170-537,
// core/model.hpp, core/fit_model.hpp, core/prediction.hpp, core/dataset.hpp,
// core/distribution.hpp.  Every Gram / factor / solve / predict is executed by
// the HIP library; nothing here computes on the CPU.  The reference's Eigen
// containers are replaced by the minimal column-major `Vector` / `Matrix`
// below (Eigen is not a dependency of this engine).
#ifndef ALBATROSS_AMD_ALBATROSS_HPP
#define ALBATROSS_AMD_ALBATROSS_HPP

#include <algorithm>
#include <array>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <utility>
#include <variant>
#include <vector>

#include "../albatross_amd.h"

namespace albatross {

// ---------------------------------------------------------------------------
// containers
// ---------------------------------------------------------------------------
using Vector = std::vector<double>;

struct Matrix {  // column-major, like Eigen::MatrixXd
  std::int64_t rows_ = 0, cols_ = 0;
  std::vector<double> data;
  Matrix() = default;
  Matrix(std::int64_t r, std::int64_t c) : rows_(r), cols_(c), data(static_cast<std::size_t>(r * c), 0.) {}
  std::int64_t rows() const { return rows_; }
  std::int64_t cols() const { return cols_; }
  double &operator()(std::int64_t i, std::int64_t j) { return data[static_cast<std::size_t>(i + j * rows_)]; }
  double operator()(std::int64_t i, std::int64_t j) const { return data[static_cast<std::size_t>(i + j * rows_)]; }
  Vector diagonal() const {
    Vector d(static_cast<std::size_t>(rows_ < cols_ ? rows_ : cols_));
    for (std::size_t i = 0; i < d.size(); ++i) d[i] = (*this)(static_cast<std::int64_t>(i), static_cast<std::int64_t>(i));
    return d;
  }
};

// core/distribution.hpp
struct MarginalDistribution {
  Vector mean;
  Vector covariance;  // diagonal; empty = no target variance
  MarginalDistribution() = default;
  explicit MarginalDistribution(Vector mean_) : mean(std::move(mean_)) {}
  MarginalDistribution(Vector mean_, Vector diag) : mean(std::move(mean_)), covariance(std::move(diag)) {}
  std::size_t size() const { return mean.size(); }
};

struct JointDistribution {
  Vector mean;
  Matrix covariance;
  std::size_t size() const { return mean.size(); }
  MarginalDistribution marginal() const { return MarginalDistribution(mean, covariance.diagonal()); }
};

// core/dataset.hpp
template <typename FeatureType>
struct RegressionDataset {
  std::vector<FeatureType> features;
  MarginalDistribution targets;
  RegressionDataset() = default;
  RegressionDataset(std::vector<FeatureType> f, MarginalDistribution t) : features(std::move(f)), targets(std::move(t)) {}
  RegressionDataset(std::vector<FeatureType> f, Vector t) : features(std::move(f)), targets(std::move(t)) {}
  std::size_t size() const { return features.size(); }
};

// measurement.hpp:18-53
template <typename X>
struct Measurement {
  X value;
  Measurement() : value() {}
  Measurement(const X &x) : value(x) {}
};

template <typename X>
std::vector<Measurement<X>> as_measurements(const std::vector<X> &features) {
  return std::vector<Measurement<X>>(features.begin(), features.end());
}

// core/linear_combination.hpp:18-44: a feature that is sum_i coefficients[i] * values[i].  The covariance
// functions and models below accept std::vector<LinearCombination<X>> wherever they accept std::vector<X>:
// LinearCombinationCaller (covariance_functions/callers.hpp:321-396) applies the double sum at the top of the
// caller chain, so the Gram matrix of the EXPANDED points is built on the device and contracted with the
// coefficients there too (agp_gram_combined; predictions: agp_solver_predict_combined).  (The reference mixes plain and combined features through variant<X, LinearCombination<X>>;
// here a plain feature in such a vector is the combination of itself: LinearCombination<X>({x}).)
template <typename X>
struct LinearCombination {
  LinearCombination() = default;
  explicit LinearCombination(const std::vector<X> &values_) : values(values_), coefficients(values_.size(), 1.) {}
  LinearCombination(const std::vector<X> &values_, const Vector &coefficients_) : values(values_), coefficients(coefficients_) {
    if (values.size() != coefficients.size()) throw std::invalid_argument("values and coefficients differ in size");
  }
  bool operator==(const LinearCombination &other) const { return values == other.values && coefficients == other.coefficients; }
  std::vector<X> values;
  Vector coefficients;
};

// ---------------------------------------------------------------------------
// feature flattening: how a feature type becomes the POD record of the C-ABI.
// Specialise FeatureTraits<X> for user types (coords, optional equality id).
// ---------------------------------------------------------------------------
template <typename X, typename Enable = void>
struct FeatureTraits;

template <>
struct FeatureTraits<double> {
  static constexpr int dim = 1;
  static constexpr bool has_eq_id = false;
  static void coords(const double &x, double *out) { out[0] = x; }
  static std::int64_t eq_id(const double &) { return 0; }
};

template <std::size_t N>
struct FeatureTraits<std::array<double, N>> {
  static_assert(N >= 1 && N <= AGP_MAX_DIM, "feature dimension must be 1..AGP_MAX_DIM");
  static constexpr int dim = static_cast<int>(N);
  static constexpr bool has_eq_id = false;
  static void coords(const std::array<double, N> &x, double *out) {
    for (std::size_t d = 0; d < N; ++d) out[d] = x[d];
  }
  static std::int64_t eq_id(const std::array<double, N> &) { return 0; }
};

// std::vector<std::variant<T0, T1, ...>> (the reference uses mapbox::variant; VariantForwarder,
// covariance_functions/callers.hpp:419-544): the POD record is zero-padded to the widest alternative, equality is
// "same alternative and equal value" (a hash of both as equality id), and every point carries the index of the
// alternative it holds in the last scale column, where only_for_alternatives<A, B>(cov) terms read it.
template <typename... Ts>
struct FeatureTraits<std::variant<Ts...>> {
  static constexpr int dim = std::max({FeatureTraits<Ts>::dim...});
  static constexpr bool has_eq_id = true;
  static void coords(const std::variant<Ts...> &x, double *out) {
    for (int d = 0; d < dim; ++d) out[d] = 0.;
    std::visit([out](const auto &v) { FeatureTraits<std::decay_t<decltype(v)>>::coords(v, out); }, x);
  }
  static std::int64_t eq_id(const std::variant<Ts...> &x) {
    double c[dim];
    coords(x, c);
    std::uint64_t h = 1469598103934665603ull ^ static_cast<std::uint64_t>(x.index());  // FNV-1a over index + coordinate bits
    h *= 1099511628211ull;
    for (int d = 0; d < dim; ++d) {
      std::uint64_t bits;
      static_assert(sizeof(bits) == sizeof(double), "double is 64 bits");
      std::memcpy(&bits, &c[d], sizeof(bits));
      for (int b = 0; b < 8; ++b) {
        h ^= (bits >> (8 * b)) & 0xffu;
        h *= 1099511628211ull;
      }
    }
    return static_cast<std::int64_t>(h & 0x7fffffffffffffffull);
  }
};

namespace detail {

constexpr int kAlternativeColumn = AGP_MAX_SCALE_COLUMNS - 1;  // scale column that carries the alternative index
template <typename X>
double alternative_index(const X &) { return 0.; }  // plain feature types are alternative 0
template <typename... Ts>
double alternative_index(const std::variant<Ts...> &x) { return static_cast<double>(x.index()); }

// f(x) of a ScalingTerm; a feature type the scaling function has no _call_impl for is ignored, i.e. scales by 1
// (the one-sided overloads of scaling_function.hpp:92-112); a variant is visited
template <typename F, typename X, typename = void>
struct has_scaling_call : std::false_type {};
template <typename F, typename X>
struct has_scaling_call<F, X, std::void_t<decltype(std::declval<const F &>()._call_impl(std::declval<const X &>()))>>
    : std::true_type {};
template <typename F, typename X>
double scale_of(const F &f, const X &x) {
  if constexpr (has_scaling_call<F, X>::value) return f._call_impl(x);
  else return 1.;
}
template <typename F, typename... Ts>
double scale_of(const F &f, const std::variant<Ts...> &x) {
  return std::visit([&f](const auto &v) { return scale_of(f, v); }, x);
}

inline void check(int status, agp_context *ctx, const char *what) {
  if (status == AGP_OK) return;
  std::string msg = std::string("albatross_amd: ") + what + ": " + agp_status_string(status);
  if (status == AGP_ERR_HIP && ctx) msg += std::string(" (") + agp_last_error(ctx) + ")";
  throw std::runtime_error(msg);
}

struct ContextHolder {
  agp_context *ctx = nullptr;
  explicit ContextHolder(int device) { check(agp_context_create(device, &ctx), nullptr, "agp_context_create"); }
  ~ContextHolder() { agp_context_destroy(ctx); }
  ContextHolder(const ContextHolder &) = delete;
  ContextHolder &operator=(const ContextHolder &) = delete;
};

inline int &default_device() {
  static int device = 0;
  return device;
}

// One context per HOST THREAD: contexts are independent and thread-safe against each other (a context itself serves one
// thread at a time), so threads that fit independent datasets run concurrently on the GPU - one fit's chain-bound tail beside
// another's bulk phase (bench.py configs.fits_in_flight).  Objects keep the context they were made with alive.
inline std::shared_ptr<ContextHolder> default_context() {
  static thread_local std::shared_ptr<ContextHolder> c = std::make_shared<ContextHolder>(default_device());
  return c;
}

struct KernelHolder {
  agp_kernel *k = nullptr;
  explicit KernelHolder(const std::vector<agp_kernel_node> &nodes) {
    check(agp_kernel_create(nodes.data(), static_cast<int>(nodes.size()), &k), nullptr, "agp_kernel_create");
  }
  ~KernelHolder() { agp_kernel_destroy(k); }
  KernelHolder(const KernelHolder &) = delete;
  KernelHolder &operator=(const KernelHolder &) = delete;
};

template <typename X>
struct unwrap {
  using type = X;
  static constexpr bool is_measurement = false;
  static const X &get(const X &x) { return x; }
};
template <typename X>
struct unwrap<Measurement<X>> {
  using type = X;
  static constexpr bool is_measurement = true;
  static const X &get(const Measurement<X> &m) { return m.value; }
};

// LinearCombination features -> the points they are made of + (owner, coefficient) per point
template <typename F>
struct expansion {
  static constexpr bool expands = false;
};
template <typename X>
struct expansion<LinearCombination<X>> {
  static constexpr bool expands = true;
  using point = X;
  static const LinearCombination<X> &get(const LinearCombination<X> &f) { return f; }
  static point make(const X &x) { return x; }
};
template <typename X>
struct expansion<Measurement<LinearCombination<X>>> {  // MeasurementForwarder sits outside LinearCombinationCaller
  static constexpr bool expands = true;
  using point = Measurement<X>;
  static const LinearCombination<X> &get(const Measurement<LinearCombination<X>> &f) { return f.value; }
  static point make(const X &x) { return Measurement<X>(x); }
};

template <typename F, bool = expansion<F>::expands>
struct Expanded {  // plain features: nothing to do
  static constexpr bool expands = false;
};
template <typename F>
struct Expanded<F, true> {
  static constexpr bool expands = true;
  std::vector<typename expansion<F>::point> points;
  std::vector<std::size_t> owner;
  std::vector<double> coefficient;
  std::size_t n = 0;
  explicit Expanded(const std::vector<F> &features) : n(features.size()) {
    for (std::size_t j = 0; j < features.size(); ++j) {
      const auto &lc = expansion<F>::get(features[j]);
      for (std::size_t i = 0; i < lc.values.size(); ++i) {
        points.push_back(expansion<F>::make(lc.values[i]));
        owner.push_back(j);
        coefficient.push_back(lc.coefficients[i]);
      }
    }
  }
  // combination j = expanded points offsets()[j] .. offsets()[j + 1] (the members were appended feature by feature)
  std::vector<std::int64_t> offsets() const {
    std::vector<std::int64_t> off(n + 1, 0);
    for (std::size_t a = 0; a < owner.size(); ++a) ++off[owner[a] + 1];
    for (std::size_t j = 0; j < n; ++j) off[j + 1] += off[j];
    return off;
  }
};

// mean_function(feature): sum_i a_i m(x_i) for a LinearCombination (callers.hpp:386-396)
template <typename Mean, typename P>
double mean_at(const Mean &m, const P &x) { return m._call_impl(unwrap<P>::get(x)); }
template <typename Mean, typename X>
double mean_at(const Mean &m, const LinearCombination<X> &x) {
  double s = 0.;
  for (std::size_t i = 0; i < x.values.size(); ++i) s += x.coefficients[i] * m._call_impl(x.values[i]);
  return s;
}
template <typename Mean, typename X>
double mean_at(const Mean &m, const Measurement<LinearCombination<X>> &x) { return mean_at(m, x.value); }

// flattened feature vector + the agp_features view over it
struct Flat {
  std::vector<double> coords, scales;
  std::vector<std::int64_t> ids;
  agp_features view{};
};

template <typename Cov, typename F>
Flat flatten(const Cov &cov, const std::vector<F> &features, bool force_measurement = false) {
  using U = unwrap<F>;
  using X = typename U::type;
  using T = FeatureTraits<X>;
  Flat f;
  const std::size_t n = features.size();
  f.coords.resize(n * T::dim);
  const int ncol = cov.n_scale_columns();
  f.scales.resize(n * static_cast<std::size_t>(ncol));
  if (T::has_eq_id) f.ids.resize(n);
  std::vector<double> tmp(static_cast<std::size_t>(ncol));
  for (std::size_t i = 0; i < n; ++i) {
    const X &x = U::get(features[i]);
    T::coords(x, f.coords.data() + i * T::dim);
    if (T::has_eq_id) f.ids[i] = T::eq_id(x);
    if (ncol > 0) {
      int col = 0;
      cov.template fill_scales<X>(x, tmp.data(), col);
      for (int c = 0; c < ncol; ++c) f.scales[static_cast<std::size_t>(c) * n + i] = tmp[static_cast<std::size_t>(c)];
    }
  }
  f.view.n = static_cast<std::int64_t>(n);
  f.view.dim = T::dim;
  f.view.n_scale_columns = ncol;
  f.view.coords = f.coords.data();
  f.view.eq_id = T::has_eq_id ? f.ids.data() : nullptr;
  f.view.scales = ncol > 0 ? f.scales.data() : nullptr;
  f.view.is_measurement = (U::is_measurement || force_measurement) ? 1 : 0;
  f.view.location = AGP_HOST;
  return f;
}

inline agp_kernel_node node(int op, int metric = 0, int column = 0, int order = 0, double p0 = 0., double p1 = 0.,
                            double p2 = 0., double p3 = 0.) {
  agp_kernel_node nd{};
  nd.op = op; nd.metric = metric; nd.column = column; nd.order = order;
  nd.params[0] = p0; nd.params[1] = p1; nd.params[2] = p2; nd.params[3] = p3;
  return nd;
}

}  // namespace detail

using ParameterStore = std::map<std::string, double>;

// ---------------------------------------------------------------------------
// distance metrics (distance_metrics.hpp:30-90): tags; the math runs on device
// ---------------------------------------------------------------------------
struct EuclideanDistance {
  static constexpr int metric = AGP_METRIC_EUCLIDEAN;
  std::string get_name() const { return "euclidean_distance"; }
};
struct RadialDistance {
  static constexpr int metric = AGP_METRIC_RADIAL;
  std::string get_name() const { return "radial_distance"; }
};
struct AngularDistance {
  static constexpr int metric = AGP_METRIC_ANGULAR;
  std::string get_name() const { return "angular_distance"; }
};

template <class LHS, class RHS> class SumOfCovarianceFunctions;
template <class LHS, class RHS> class ProductOfCovarianceFunctions;

// ---------------------------------------------------------------------------
// CovarianceFunction CRTP base (covariance_function.hpp:63-217)
// ---------------------------------------------------------------------------
template <typename Derived>
class CovarianceFunction {
 public:
  const Derived &derived() const { return *static_cast<const Derived *>(this); }
  Derived &derived() { return *static_cast<Derived *>(this); }

  std::string get_name() const { return derived().name(); }

  std::vector<agp_kernel_node> program() const {
    std::vector<agp_kernel_node> nodes;
    int column = 0;
    derived().emit(nodes, column);
    return nodes;
  }

  int n_scale_columns() const {
    std::vector<agp_kernel_node> nodes;
    int column = 0;
    derived().emit(nodes, column);
    for (const auto &nd : nodes)
      if (nd.op == AGP_OP_TYPE_PAIR) {  // the alternative index lives in the last column
        if (column > detail::kAlternativeColumn) throw std::invalid_argument("too many ScalingTerms next to only_for_alternatives");
        return detail::kAlternativeColumn + 1;
      }
    return column;
  }

  void set_param_values(const ParameterStore &values) {
    for (const auto &kv : values) derived().set_param(kv.first, kv.second);
  }
  double get_param_value(const std::string &name) const { return derived().get_params().at(name); }

  // cov(xs): symmetric Gram, callers.hpp:107-166
  template <typename F>
  Matrix operator()(const std::vector<F> &xs) const {
    if constexpr (detail::expansion<F>::expands) {  // LinearCombinationCaller, callers.hpp:336-347: Gram of the expanded
      const detail::Expanded<F> ex(xs);              // points and its contraction on the device (agp_gram_combined)
      auto ctx = detail::default_context();
      detail::KernelHolder k(program());
      detail::Flat fx = detail::flatten(derived(), ex.points);
      const std::vector<std::int64_t> off = ex.offsets();
      Matrix out(static_cast<std::int64_t>(ex.n), static_cast<std::int64_t>(ex.n));
      if (ex.n > 0)
        detail::check(agp_gram_combined(ctx->ctx, k.k, &fx.view, static_cast<std::int64_t>(ex.n), off.data(), ex.coefficient.data(),
                                        nullptr, 0, nullptr, nullptr, out.data.data(), static_cast<std::int64_t>(ex.n), AGP_HOST),
                      ctx->ctx, "agp_gram_combined");
      return out;
    } else {
    auto ctx = detail::default_context();
    detail::KernelHolder k(program());
    detail::Flat fx = detail::flatten(derived(), xs);
    Matrix out(fx.view.n, fx.view.n);
    if (fx.view.n > 0)
      detail::check(agp_gram(ctx->ctx, k.k, &fx.view, nullptr, out.data.data(), fx.view.n, AGP_HOST), ctx->ctx, "agp_gram");
    return out;
    }
  }

  // cov(xs, ys): cross Gram, callers.hpp:38-102
  template <typename F, typename G>
  Matrix operator()(const std::vector<F> &xs, const std::vector<G> &ys) const {
    if constexpr (detail::expansion<F>::expands || detail::expansion<G>::expands) {  // callers.hpp:336-376, on the device
      auto ctx = detail::default_context();
      detail::KernelHolder k(program());
      std::vector<std::int64_t> xoff, yoff;
      std::vector<double> xc, yc;
      std::int64_t nx = static_cast<std::int64_t>(xs.size()), ny = static_cast<std::int64_t>(ys.size());
      detail::Flat fx, fy;
      if constexpr (detail::expansion<F>::expands) {
        const detail::Expanded<F> ex(xs);
        fx = detail::flatten(derived(), ex.points);
        xoff = ex.offsets();
        xc = ex.coefficient;
      } else {
        fx = detail::flatten(derived(), xs);
      }
      if constexpr (detail::expansion<G>::expands) {
        const detail::Expanded<G> ey(ys);
        fy = detail::flatten(derived(), ey.points);
        yoff = ey.offsets();
        yc = ey.coefficient;
      } else {
        fy = detail::flatten(derived(), ys);
      }
      Matrix out(nx, ny);
      if (nx > 0 && ny > 0)
        detail::check(agp_gram_combined(ctx->ctx, k.k, &fx.view, nx, xoff.empty() ? nullptr : xoff.data(), xoff.empty() ? nullptr : xc.data(),
                                        &fy.view, ny, yoff.empty() ? nullptr : yoff.data(), yoff.empty() ? nullptr : yc.data(),
                                        out.data.data(), nx, AGP_HOST),
                      ctx->ctx, "agp_gram_combined");
      return out;
    } else {
    auto ctx = detail::default_context();
    detail::KernelHolder k(program());
    detail::Flat fx = detail::flatten(derived(), xs), fy = detail::flatten(derived(), ys);
    Matrix out(fx.view.n, fy.view.n);
    if (fx.view.n > 0 && fy.view.n > 0)
      detail::check(agp_gram(ctx->ctx, k.k, &fx.view, &fy.view, out.data.data(), fx.view.n, AGP_HOST), ctx->ctx, "agp_gram");
    return out;
    }
  }

  // cov(x, y) for two single features (CovarianceFunction::call)
  template <typename F, typename G>
  double call(const F &x, const G &y) const {
    return (*this)(std::vector<F>{x}, std::vector<G>{y})(0, 0);
  }

  template <typename Other>
  SumOfCovarianceFunctions<Derived, Other> operator+(const CovarianceFunction<Other> &other) const {
    return SumOfCovarianceFunctions<Derived, Other>(derived(), other.derived());
  }
  template <typename Other>
  ProductOfCovarianceFunctions<Derived, Other> operator*(const CovarianceFunction<Other> &other) const {
    return ProductOfCovarianceFunctions<Derived, Other>(derived(), other.derived());
  }
};

constexpr double default_length_scale = 100000.;  // radial.hpp:16
constexpr double default_radial_sigma = 10.;      // radial.hpp:17

#define ALBATROSS_AMD_RADIAL(ClassName, OP, LS_NAME, SIGMA_NAME, PRETTY)                                   \
  template <class DistanceMetricType>                                                                      \
  class ClassName : public CovarianceFunction<ClassName<DistanceMetricType>> {                             \
   public:                                                                                                 \
    ClassName(double length_scale_ = default_length_scale, double sigma_ = default_radial_sigma)           \
        : length_scale(length_scale_), sigma(sigma_) {}                                                    \
    std::string name() const { return std::string(PRETTY "[") + distance_metric_.get_name() + "]"; }        \
    ParameterStore get_params() const { return {{LS_NAME, length_scale}, {SIGMA_NAME, sigma}}; }            \
    bool has_param(const std::string &n) const { return n == LS_NAME || n == SIGMA_NAME; }                  \
    void set_param(const std::string &n, double v) {                                                       \
      if (n == LS_NAME) length_scale = v;                                                                  \
      else if (n == SIGMA_NAME) sigma = v;                                                                 \
      else throw std::out_of_range("unknown parameter " + n);                                              \
    }                                                                                                      \
    void emit(std::vector<agp_kernel_node> &nodes, int &) const {                                          \
      nodes.push_back(detail::node(OP, DistanceMetricType::metric, 0, 0, length_scale, sigma));            \
    }                                                                                                      \
    template <typename X> void fill_scales(const X &, double *, int &) const {}                           \
    double length_scale, sigma;                                                                            \
    DistanceMetricType distance_metric_;                                                                   \
  };

// radial.hpp:131-189, 239-287, 421-459, 491-529
ALBATROSS_AMD_RADIAL(SquaredExponential, AGP_OP_SQUARED_EXPONENTIAL, "squared_exponential_length_scale",
                     "sigma_squared_exponential", "squared_exponential")
ALBATROSS_AMD_RADIAL(Exponential, AGP_OP_EXPONENTIAL, "exponential_length_scale", "sigma_exponential", "exponential")
ALBATROSS_AMD_RADIAL(Matern32, AGP_OP_MATERN32, "matern_32_length_scale", "sigma_matern_32", "matern_32")
ALBATROSS_AMD_RADIAL(Matern52, AGP_OP_MATERN52, "matern_52_length_scale", "sigma_matern_52", "matern_52")
#undef ALBATROSS_AMD_RADIAL

// The SquaredExponential is not PSD under a great-circle distance (radial.hpp:138-141)
template <>
class SquaredExponential<AngularDistance>;

#define ALBATROSS_AMD_ONE_PARAM(ClassDecl, ClassName, OP, PNAME, DEFAULT, PRETTY)                         \
  ClassDecl class ClassName : public CovarianceFunction<ClassName> {                                      \
   public:                                                                                                \
    explicit ClassName(double v = DEFAULT) : value(v) {}                                                  \
    std::string name() const { return PRETTY; }                                                           \
    ParameterStore get_params() const { return {{PNAME, value}}; }                                        \
    bool has_param(const std::string &n) const { return n == PNAME; }                                     \
    void set_param(const std::string &n, double v) {                                                      \
      if (n != PNAME) throw std::out_of_range("unknown parameter " + n);                                  \
      value = v;                                                                                          \
    }                                                                                                     \
    void emit(std::vector<agp_kernel_node> &nodes, int &) const { nodes.push_back(detail::node(OP, 0, 0, 0, value)); } \
    template <typename X> void fill_scales(const X &, double *, int &) const {}                          \
    double value;                                                                                         \
  };

// polynomials.hpp:31-61, nugget.hpp:32-49
ALBATROSS_AMD_ONE_PARAM(, Constant, AGP_OP_CONSTANT, "sigma_constant", 10., "constant")
ALBATROSS_AMD_ONE_PARAM(, Nugget, AGP_OP_NUGGET, "nugget_sigma", 1e-8, "nugget")
#undef ALBATROSS_AMD_ONE_PARAM

// noise.hpp:20-44
template <typename Observed>
class IndependentNoise : public CovarianceFunction<IndependentNoise<Observed>> {
 public:
  explicit IndependentNoise(double sigma_noise = 0.1) : sigma_independent_noise(sigma_noise) {}
  std::string name() const { return "independent_noise"; }
  ParameterStore get_params() const { return {{"sigma_independent_noise", sigma_independent_noise}}; }
  bool has_param(const std::string &n) const { return n == "sigma_independent_noise"; }
  void set_param(const std::string &n, double v) {
    if (!has_param(n)) throw std::out_of_range("unknown parameter " + n);
    sigma_independent_noise = v;
  }
  void emit(std::vector<agp_kernel_node> &nodes, int &) const {
    nodes.push_back(detail::node(AGP_OP_INDEPENDENT_NOISE, 0, 0, 0, sigma_independent_noise));
  }
  template <typename X> void fill_scales(const X &, double *, int &) const {}
  double sigma_independent_noise;
};

// polynomials.hpp:63-90 (1-D features)
template <int order>
class Polynomial : public CovarianceFunction<Polynomial<order>> {
  static_assert(order >= 0 && order <= 3, "device path supports Polynomial<order> for order <= 3");

 public:
  explicit Polynomial(double sigma = 10.) { sigmas.fill(sigma); }
  std::string name() const { return "polynomial_" + std::to_string(order); }
  ParameterStore get_params() const {
    ParameterStore p;
    for (int i = 0; i <= order; ++i) p["sigma_polynomial_" + std::to_string(i)] = sigmas[static_cast<std::size_t>(i)];
    return p;
  }
  bool has_param(const std::string &n) const { return get_params().count(n) > 0; }
  void set_param(const std::string &n, double v) {
    for (int i = 0; i <= order; ++i)
      if (n == "sigma_polynomial_" + std::to_string(i)) { sigmas[static_cast<std::size_t>(i)] = v; return; }
    throw std::out_of_range("unknown parameter " + n);
  }
  void emit(std::vector<agp_kernel_node> &nodes, int &) const {
    nodes.push_back(detail::node(AGP_OP_POLYNOMIAL, 0, 0, order, sigmas[0], sigmas[1], sigmas[2], sigmas[3]));
  }
  template <typename X> void fill_scales(const X &, double *, int &) const {}
  std::array<double, 4> sigmas{};
};

// scaling_function.hpp:58-112: cov(x, y) = f(x) f(y).  ScalingFunction needs
//   double _call_impl(const X &) const;  std::string get_name() const;
//   ParameterStore get_params() const;   void set_param(name, value);
template <typename ScalingFunction>
class ScalingTerm : public CovarianceFunction<ScalingTerm<ScalingFunction>> {
 public:
  ScalingTerm() = default;
  explicit ScalingTerm(const ScalingFunction &f) : scaling_function_(f) {}
  std::string name() const { return scaling_function_.get_name(); }
  ParameterStore get_params() const { return scaling_function_.get_params(); }
  bool has_param(const std::string &n) const { return get_params().count(n) > 0; }
  void set_param(const std::string &n, double v) { scaling_function_.set_param(n, v); }
  void emit(std::vector<agp_kernel_node> &nodes, int &column) const {
    nodes.push_back(detail::node(AGP_OP_SCALING, 0, column, 0));
    ++column;
  }
  template <typename X>
  void fill_scales(const X &x, double *out, int &column) const {
    out[column++] = detail::scale_of(scaling_function_, x);  // evaluated once per point, not per pair
  }

 private:
  ScalingFunction scaling_function_;
};

namespace detail {
// index groups -> offsets / indices of the C-ABI (agp_fit_inverse_blocks, agp_held_out_predictions)
inline void flatten_groups(const std::vector<std::vector<std::size_t>> &groups, std::vector<std::int64_t> *offsets,
                           std::vector<std::int64_t> *indices) {
  offsets->assign(1, 0);
  indices->clear();
  for (const auto &g : groups) {
    for (std::size_t i : g) indices->push_back(static_cast<std::int64_t>(i));
    offsets->push_back(static_cast<std::int64_t>(indices->size()));
  }
  if (indices->empty()) indices->push_back(0);  // keep data() non-null for empty inputs
}

template <class LHS, class RHS, int OP, char SYM, typename Self>
class Binary : public CovarianceFunction<Self> {
 public:
  Binary() = default;
  Binary(const LHS &l, const RHS &r) : lhs_(l), rhs_(r) {}
  std::string name() const { return "(" + lhs_.get_name() + std::string(1, SYM) + rhs_.get_name() + ")"; }
  ParameterStore get_params() const {  // map_join, covariance_function.hpp:235-237
    ParameterStore p = lhs_.get_params();
    for (const auto &kv : rhs_.get_params()) p[kv.first] = kv.second;
    return p;
  }
  bool has_param(const std::string &n) const { return lhs_.has_param(n) || rhs_.has_param(n); }
  void set_param(const std::string &n, double v) {  // set_param_if_exists_in_any, :239-242
    bool done = false;
    if (lhs_.has_param(n)) { lhs_.set_param(n, v); done = true; }
    if (rhs_.has_param(n)) { rhs_.set_param(n, v); done = true; }
    if (!done) throw std::out_of_range("unknown parameter " + n);
  }
  void emit(std::vector<agp_kernel_node> &nodes, int &column) const {
    lhs_.emit(nodes, column);
    rhs_.emit(nodes, column);
    nodes.push_back(detail::node(OP));
  }
  template <typename X>
  void fill_scales(const X &x, double *out, int &column) const {
    lhs_.template fill_scales<X>(x, out, column);
    rhs_.template fill_scales<X>(x, out, column);
  }

 protected:
  LHS lhs_;
  RHS rhs_;
};
}  // namespace detail

// covariance_function.hpp:222-325
template <class LHS, class RHS>
class SumOfCovarianceFunctions
    : public detail::Binary<LHS, RHS, AGP_OP_SUM, '+', SumOfCovarianceFunctions<LHS, RHS>> {
  using Base = detail::Binary<LHS, RHS, AGP_OP_SUM, '+', SumOfCovarianceFunctions<LHS, RHS>>;
 public:
  using Base::Base;
};

// covariance_function.hpp:330-420 (rhs skipped when lhs == 0, :362-366 — on device)
template <class LHS, class RHS>
class ProductOfCovarianceFunctions
    : public detail::Binary<LHS, RHS, AGP_OP_PRODUCT, '*', ProductOfCovarianceFunctions<LHS, RHS>> {
  using Base = detail::Binary<LHS, RHS, AGP_OP_PRODUCT, '*', ProductOfCovarianceFunctions<LHS, RHS>>;
 public:
  using Base::Base;
};

// measurement.hpp:70-106
template <typename SubCovariance>
class MeasurementOnly : public CovarianceFunction<MeasurementOnly<SubCovariance>> {
 public:
  MeasurementOnly() = default;
  explicit MeasurementOnly(const SubCovariance &sub) : sub_cov_(sub) {}
  std::string name() const { return "measurement[" + sub_cov_.get_name() + "]"; }
  ParameterStore get_params() const { return sub_cov_.get_params(); }
  bool has_param(const std::string &n) const { return sub_cov_.has_param(n); }
  void set_param(const std::string &n, double v) { sub_cov_.set_param(n, v); }
  void emit(std::vector<agp_kernel_node> &nodes, int &column) const {
    sub_cov_.emit(nodes, column);
    nodes.push_back(detail::node(AGP_OP_MEASUREMENT_ONLY));
  }
  template <typename X>
  void fill_scales(const X &x, double *out, int &column) const { sub_cov_.template fill_scales<X>(x, out, column); }

 private:
  SubCovariance sub_cov_;
};

template <typename SubCovariance>
MeasurementOnly<SubCovariance> measurement_only(const SubCovariance &cov) {
  return MeasurementOnly<SubCovariance>(cov);
}

// A covariance term that is defined for ONE pair (A, B) of alternatives of a variant feature type, in either order:
// what a `_call_impl(const TA &, const TB &)` overload is in the reference.  VariantForwarder
// (covariance_functions/callers.hpp:419-544) returns 0 for pairs of alternatives without an overload; a covariance
// function with several overloads (tests/lib/albatross/test/test_covariance_utils.h:42-62) is the sum of one such
// term per overload.  A, B: alternative indices (std::variant::index()).
template <typename SubCovariance>
class OnlyForAlternatives : public CovarianceFunction<OnlyForAlternatives<SubCovariance>> {
 public:
  OnlyForAlternatives() = default;
  OnlyForAlternatives(const SubCovariance &sub, int a, int b) : sub_cov_(sub), a_(a), b_(b) {}
  std::string name() const {
    return "alternatives[" + std::to_string(a_) + "," + std::to_string(b_) + "][" + sub_cov_.get_name() + "]";
  }
  ParameterStore get_params() const { return sub_cov_.get_params(); }
  bool has_param(const std::string &n) const { return sub_cov_.has_param(n); }
  void set_param(const std::string &n, double v) { sub_cov_.set_param(n, v); }
  void emit(std::vector<agp_kernel_node> &nodes, int &column) const {
    sub_cov_.emit(nodes, column);
    agp_kernel_node nd = detail::node(AGP_OP_TYPE_PAIR, 0, detail::kAlternativeColumn, 0);
    nd.params[0] = a_;
    nd.params[1] = b_;
    nodes.push_back(nd);
  }
  template <typename X>
  void fill_scales(const X &x, double *out, int &column) const {
    sub_cov_.template fill_scales<X>(x, out, column);
    out[detail::kAlternativeColumn] = detail::alternative_index(x);
  }

 private:
  SubCovariance sub_cov_;
  int a_ = 0, b_ = 0;
};

template <int A, int B = A, typename SubCovariance>
OnlyForAlternatives<SubCovariance> only_for_alternatives(const SubCovariance &cov) {
  return OnlyForAlternatives<SubCovariance>(cov, A, B);
}

// ---------------------------------------------------------------------------
// mean functions (mean_function.hpp:86-107,274-276; polynomials.hpp:92-106)
// ---------------------------------------------------------------------------
struct ZeroMean {
  std::string get_name() const { return "zero_mean"; }
  ParameterStore get_params() const { return {}; }
  bool has_param(const std::string &) const { return false; }
  void set_param(const std::string &n, double) { throw std::out_of_range("unknown parameter " + n); }
  template <typename X> double _call_impl(const X &) const { return 0.; }
};

struct LinearMean {
  double slope = 0., offset = 0.;
  std::string get_name() const { return "linear"; }
  ParameterStore get_params() const { return {{"slope", slope}, {"offset", offset}}; }
  bool has_param(const std::string &n) const { return n == "slope" || n == "offset"; }
  void set_param(const std::string &n, double v) {
    if (n == "slope") slope = v;
    else if (n == "offset") offset = v;
    else throw std::out_of_range("unknown parameter " + n);
  }
  double _call_impl(const double &x) const { return slope * x + offset; }
};

// ---------------------------------------------------------------------------
// Fit<GPFit<...>> (gp.hpp:43-77): the factor lives on the device
// ---------------------------------------------------------------------------
template <typename FeatureType>
struct GPFit {
  std::vector<FeatureType> train_features;
  Vector information;
  double log_determinant = 0.;
  std::shared_ptr<detail::ContextHolder> context;
  std::shared_ptr<agp_fit> handle;  // train_covariance (CovarianceRepresentation)

  std::int64_t rows() const { return static_cast<std::int64_t>(train_features.size()); }

  // SerializableLDLT::inverse_diagonal, eigen/serializable_ldlt.hpp:181-199
  Vector inverse_diagonal() const {
    Vector out(train_features.size());
    detail::check(agp_fit_inverse_diagonal(context->ctx, handle.get(), out.data(), AGP_HOST), context->ctx,
                  "agp_fit_inverse_diagonal");
    return out;
  }

  // leave-one-out predictive marginals of every training point
  // (held_out_predictions with singleton groups, evaluation/cross_validation_utils.hpp:165-232)
  MarginalDistribution leave_one_out(const Vector &target_mean) const {
    MarginalDistribution out(Vector(train_features.size()), Vector(train_features.size()));
    detail::check(agp_loo_marginal(context->ctx, handle.get(), target_mean.data(), out.mean.data(),
                                   out.covariance.data(), AGP_HOST),
                  context->ctx, "agp_loo_marginal");
    return out;
  }

  // SerializableLDLT::inverse_blocks, eigen/serializable_ldlt.hpp:137-179: (K^-1)[I_g, I_g] per index group
  std::vector<Matrix> inverse_blocks(const std::vector<std::vector<std::size_t>> &blocks) const {
    std::vector<std::int64_t> offsets, indices;
    detail::flatten_groups(blocks, &offsets, &indices);
    std::size_t elems = 0;
    for (const auto &b : blocks) elems += b.size() * b.size();
    std::vector<double> flat(elems ? elems : 1);
    detail::check(agp_fit_inverse_blocks(context->ctx, handle.get(), static_cast<std::int64_t>(blocks.size()),
                                         offsets.data(), indices.data(), flat.data(), AGP_HOST),
                  context->ctx, "agp_fit_inverse_blocks");
    std::vector<Matrix> out;
    std::size_t pos = 0;
    for (const auto &b : blocks) {
      Matrix m(static_cast<std::int64_t>(b.size()), static_cast<std::int64_t>(b.size()));
      std::copy(flat.begin() + static_cast<std::ptrdiff_t>(pos), flat.begin() + static_cast<std::ptrdiff_t>(pos + b.size() * b.size()),
                m.data.begin());
      pos += b.size() * b.size();
      out.push_back(std::move(m));
    }
    return out;
  }

  // details::held_out_predictions, evaluation/cross_validation_utils.hpp:165-232: per group the
  // joint prediction of its targets from all other groups (mean, full covariance), no refit
  std::vector<JointDistribution> held_out_predictions(const Vector &target_mean,
                                                      const std::vector<std::vector<std::size_t>> &groups) const {
    std::vector<std::int64_t> offsets, indices;
    detail::flatten_groups(groups, &offsets, &indices);
    std::size_t elems = 0;
    for (const auto &g : groups) elems += g.size() * g.size();
    Vector mean(indices.size() ? indices.size() : 1), var(indices.size() ? indices.size() : 1);
    std::vector<double> flat(elems ? elems : 1);
    detail::check(agp_held_out_predictions(context->ctx, handle.get(), target_mean.data(),
                                           static_cast<std::int64_t>(groups.size()), offsets.data(), indices.data(),
                                           mean.data(), var.data(), flat.data(), AGP_HOST),
                  context->ctx, "agp_held_out_predictions");
    std::vector<JointDistribution> out;
    std::size_t pos = 0;
    for (std::size_t g = 0; g < groups.size(); ++g) {
      const std::size_t m = groups[g].size(), o = static_cast<std::size_t>(offsets[g]);
      JointDistribution j;
      j.mean = Vector(mean.begin() + static_cast<std::ptrdiff_t>(o), mean.begin() + static_cast<std::ptrdiff_t>(o + m));
      j.covariance = Matrix(static_cast<std::int64_t>(m), static_cast<std::int64_t>(m));
      std::copy(flat.begin() + static_cast<std::ptrdiff_t>(pos), flat.begin() + static_cast<std::ptrdiff_t>(pos + m * m),
                j.covariance.data.begin());
      pos += m * m;
      out.push_back(std::move(j));
    }
    return out;
  }

  // train_covariance.solve(rhs), gp.hpp:42-45
  Matrix solve(const Matrix &rhs) const {
    Matrix out(rhs.rows(), rhs.cols());
    detail::check(agp_solve(context->ctx, handle.get(), rhs.data.data(), rhs.cols(), out.data.data(), AGP_HOST),
                  context->ctx, "agp_solve");
    return out;
  }
};

// ---------------------------------------------------------------------------
// Eigen::SerializableLDLT(const MatrixXd &) (eigen/serializable_ldlt.hpp:27):
// device LL^T of a dense symmetric positive-definite matrix (lower triangle read)
// ---------------------------------------------------------------------------
class SerializableLDLT {
 public:
  SerializableLDLT() = default;
  explicit SerializableLDLT(const Matrix &x) : context_(detail::default_context()), n_(x.rows()) {
    agp_fit *h = nullptr;
    const int st = agp_factor_create(context_->ctx, x.data.data(), x.rows(), x.rows(), /*uplo=*/0, AGP_HOST, &h);
    if (st != AGP_OK) {
      const long long pivot = h ? static_cast<long long>(agp_fit_failed_pivot(h)) : -1;
      agp_fit_destroy(h);
      const std::string what = "agp_factor_create (pivot " + std::to_string(pivot) + ")";
      detail::check(st, context_->ctx, what.c_str());
    }
    auto ctx = context_;
    handle_ = std::shared_ptr<agp_fit>(h, [ctx](agp_fit *p) { agp_fit_destroy(p); });
  }
  std::int64_t rows() const { return n_; }
  Matrix solve(const Matrix &rhs) const {
    Matrix out(rhs.rows(), rhs.cols());
    detail::check(agp_solve(context_->ctx, handle_.get(), rhs.data.data(), rhs.cols(), out.data.data(), AGP_HOST),
                  context_->ctx, "agp_solve");
    return out;
  }
  Vector solve(const Vector &rhs) const {
    Vector out(rhs.size());
    detail::check(agp_solve(context_->ctx, handle_.get(), rhs.data(), 1, out.data(), AGP_HOST), context_->ctx, "agp_solve");
    return out;
  }
  double log_determinant() const {  // serializable_ldlt.hpp:128-135
    double v = 0.;
    detail::check(agp_fit_log_determinant(handle_.get(), &v), context_->ctx, "agp_fit_log_determinant");
    return v;
  }
  Vector inverse_diagonal() const {  // serializable_ldlt.hpp:181-199
    Vector out(static_cast<std::size_t>(n_));
    detail::check(agp_fit_inverse_diagonal(context_->ctx, handle_.get(), out.data(), AGP_HOST), context_->ctx,
                  "agp_fit_inverse_diagonal");
    return out;
  }
  // this factor as a device-side CovarianceRepresentation (agp_solver): what the compositions below are built from
  std::shared_ptr<agp_solver> device_solver() const {
    agp_solver *sv = nullptr;
    detail::check(agp_solver_from_fit(context_->ctx, handle_.get(), &sv), context_->ctx, "agp_solver_from_fit");
    auto keep = handle_;
    return std::shared_ptr<agp_solver>(sv, [keep](agp_solver *p) { agp_solver_destroy(p); });
  }
  const std::shared_ptr<detail::ContextHolder> &context() const { return context_; }

 private:
  std::shared_ptr<detail::ContextHolder> context_;
  std::shared_ptr<agp_fit> handle_;
  std::int64_t n_ = 0;
};

// ---------------------------------------------------------------------------
// Eigen::LDLT<MatrixXd, Lower> as SerializableLDLT wraps it (eigen/serializable_ldlt.hpp:27): the diagonally
// pivoted P A P^T = L D L^T on the device, for symmetric matrices that are only SEMI-definite (the un-pivoted
// factor above rejects those).  Same operation order as the reference's unblocked algorithm.
// ---------------------------------------------------------------------------
class PivotedLDLT {
 public:
  PivotedLDLT() = default;
  explicit PivotedLDLT(const Matrix &x) : context_(detail::default_context()), n_(x.rows()) {
    agp_ldlt *h = nullptr;
    int ok = 1;
    detail::check(agp_ldlt_create(context_->ctx, x.data.data(), x.rows(), x.rows(), /*uplo=*/0, AGP_HOST, &h, &ok), context_->ctx,
                  "agp_ldlt_create");
    success_ = ok != 0;
    auto ctx = context_;
    handle_ = std::shared_ptr<agp_ldlt>(h, [ctx](agp_ldlt *p) { agp_ldlt_destroy(p); });
  }
  std::int64_t rows() const { return n_; }
  bool success() const { return success_; }  // info() == Eigen::Success
  Matrix solve(const Matrix &rhs) const {    // P^T L^-T D^+ L^-1 P rhs
    Matrix out(rhs.rows(), rhs.cols());
    detail::check(agp_ldlt_solve(context_->ctx, handle_.get(), rhs.data.data(), rhs.cols(), out.data.data(), AGP_HOST), context_->ctx,
                  "agp_ldlt_solve");
    return out;
  }
  Vector solve(const Vector &rhs) const {
    Vector out(rhs.size());
    detail::check(agp_ldlt_solve(context_->ctx, handle_.get(), rhs.data(), 1, out.data(), AGP_HOST), context_->ctx, "agp_ldlt_solve");
    return out;
  }
  Matrix sqrt_solve(const Matrix &rhs) const {  // D^-1/2 L^-1 P rhs (serializable_ldlt.hpp:99-109)
    Matrix out(rhs.rows(), rhs.cols());
    detail::check(agp_ldlt_sqrt_solve(context_->ctx, handle_.get(), rhs.data.data(), rhs.cols(), out.data.data(), AGP_HOST),
                  context_->ctx, "agp_ldlt_sqrt_solve");
    return out;
  }
  Vector vectorD() const {
    Vector d(static_cast<std::size_t>(n_));
    detail::check(agp_ldlt_vector_d(handle_.get(), d.data()), context_->ctx, "agp_ldlt_vector_d");
    return d;
  }
  std::vector<std::int64_t> transpositionsP() const {
    std::vector<std::int64_t> tr(static_cast<std::size_t>(n_));
    detail::check(agp_ldlt_transpositions(handle_.get(), tr.data()), context_->ctx, "agp_ldlt_transpositions");
    return tr;
  }
  double log_determinant() const {  // serializable_ldlt.hpp:128-135
    double s = 0.;
    for (double d : vectorD()) s += std::log(d);
    return s;
  }
  std::shared_ptr<agp_solver> device_solver() const {
    agp_solver *sv = nullptr;
    detail::check(agp_solver_from_ldlt(context_->ctx, handle_.get(), &sv), context_->ctx, "agp_solver_from_ldlt");
    auto keep = handle_;
    return std::shared_ptr<agp_solver>(sv, [keep](agp_solver *p) { agp_solver_destroy(p); });
  }
  const std::shared_ptr<detail::ContextHolder> &context() const { return context_; }

 private:
  std::shared_ptr<detail::ContextHolder> context_;
  std::shared_ptr<agp_ldlt> handle_;
  std::int64_t n_ = 0;
  bool success_ = true;
};

// negative_log_likelihood(deviation, covariance), evaluation/likelihood.hpp:53-66
inline double negative_log_likelihood(const Vector &deviation, const Matrix &covariance) {
  auto ctx = detail::default_context();
  double out = 0.;
  detail::check(agp_nll_dense(ctx->ctx, deviation.data(), covariance.data.data(), covariance.rows(), covariance.rows(),
                              /*uplo=*/0, AGP_HOST, &out),
                ctx->ctx, "agp_nll_dense");
  return out;
}

namespace detail {
// solve / predict through a device-side CovarianceRepresentation (agp_solver_*, include/albatross_amd.h)
inline Matrix solver_solve(const std::shared_ptr<ContextHolder> &ctx, const agp_solver *sv, const Matrix &rhs) {
  Matrix out(rhs.rows(), rhs.cols());
  if (rhs.cols() > 0 && rhs.rows() > 0)
    check(agp_solver_solve(ctx->ctx, sv, rhs.data.data(), rhs.cols(), out.data.data(), AGP_HOST), ctx->ctx, "agp_solver_solve");
  return out;
}
}  // namespace detail

// linalg/block_symmetric.hpp:46-115 - on the device (agp_solver_block_symmetric): Ai_B = A.solve(B) is computed and kept in
// HBM, a solve is device solves + MFMA products; nothing of the block algebra runs on the host.
template <typename Solver>
struct BlockSymmetric {
  BlockSymmetric() = default;
  BlockSymmetric(const Solver &A_, const Matrix &B_, const SerializableLDLT &S_) : A(A_), S(S_), context_(S_.context()) {
    sub_a_ = A.device_solver();
    sub_s_ = S.device_solver();
    agp_solver *sv = nullptr;
    detail::check(agp_solver_block_symmetric(context_->ctx, sub_a_.get(), B_.data.data(), B_.rows(), AGP_HOST, sub_s_.get(), &sv),
                  context_->ctx, "agp_solver_block_symmetric");
    auto ka = sub_a_, ks = sub_s_;
    handle_ = std::shared_ptr<agp_solver>(sv, [ka, ks](agp_solver *p) { agp_solver_destroy(p); });
  }
  std::int64_t rows() const { return A.rows() + S.rows(); }
  Matrix solve(const Matrix &rhs) const { return detail::solver_solve(context_, handle_.get(), rhs); }  // block_symmetric.hpp:75-98
  std::shared_ptr<agp_solver> device_solver() const { return handle_; }
  const std::shared_ptr<detail::ContextHolder> &context() const { return context_; }
  Solver A;
  SerializableLDLT S;

 private:
  std::shared_ptr<detail::ContextHolder> context_;
  std::shared_ptr<agp_solver> sub_a_, sub_s_, handle_;
};

template <typename ModelType, typename FeatureType> class FitModel;

// core/prediction.hpp:115-224 — lazy prediction
template <typename ModelType, typename FitFeature, typename PredictFeature>
class Prediction {
 public:
  Prediction(const FitModel<ModelType, FitFeature> *fm, std::vector<PredictFeature> features)
      : fm_(fm), features_(std::move(features)) {}
  Vector mean() const { return fm_->predict_mean_(features_); }
  MarginalDistribution marginal() const { return fm_->predict_marginal_(features_); }
  JointDistribution joint() const { return fm_->predict_joint_(features_); }

 private:
  const FitModel<ModelType, FitFeature> *fm_;
  std::vector<PredictFeature> features_;
};

// core/fit_model.hpp:18-114
template <typename ModelType, typename FeatureType>
class FitModel {
 public:
  FitModel(const ModelType &model, GPFit<FeatureType> fit) : model_(model), fit_(std::move(fit)) {}
  const GPFit<FeatureType> &get_fit() const { return fit_; }
  const ModelType &get_model() const { return model_; }

  template <typename P>
  Prediction<ModelType, FeatureType, P> predict(const std::vector<P> &features) const {
    return Prediction<ModelType, FeatureType, P>(this, features);
  }
  // fit_model.hpp:54-62
  template <typename P>
  Prediction<ModelType, FeatureType, Measurement<P>> predict_with_measurement_noise(const std::vector<P> &features) const {
    return Prediction<ModelType, FeatureType, Measurement<P>>(this, as_measurements(features));
  }

  // FitModel::update, core/fit_model.hpp:68-81 (defined after the free function update() below)
  FitModel update(const RegressionDataset<FeatureType> &dataset) const;

  // _predict_impl, gp.hpp:305-366
  template <typename P>
  Vector predict_mean_(const std::vector<P> &xs) const {
    detail::KernelHolder k(model_.get_covariance().program());
    detail::Flat f = detail::flatten(model_.get_covariance(), xs);
    Vector mean(xs.size());
    if (!xs.empty())
      detail::check(agp_predict_mean(fit_.context->ctx, k.k, fit_.handle.get(), &f.view, mean.data(), AGP_HOST),
                    fit_.context->ctx, "agp_predict_mean");
    model_.add_mean(xs, &mean);
    return mean;
  }
  template <typename P>
  MarginalDistribution predict_marginal_(const std::vector<P> &xs) const {
    detail::KernelHolder k(model_.get_covariance().program());
    detail::Flat f = detail::flatten(model_.get_covariance(), xs);
    MarginalDistribution out(Vector(xs.size()), Vector(xs.size()));
    if (!xs.empty())
      detail::check(agp_predict_marginal(fit_.context->ctx, k.k, fit_.handle.get(), &f.view, out.mean.data(),
                                         out.covariance.data(), AGP_HOST),
                    fit_.context->ctx, "agp_predict_marginal");
    model_.add_mean(xs, &out.mean);
    return out;
  }
  template <typename P>
  JointDistribution predict_joint_(const std::vector<P> &xs) const {
    detail::KernelHolder k(model_.get_covariance().program());
    detail::Flat f = detail::flatten(model_.get_covariance(), xs);
    JointDistribution out;
    out.mean.resize(xs.size());
    out.covariance = Matrix(static_cast<std::int64_t>(xs.size()), static_cast<std::int64_t>(xs.size()));
    if (!xs.empty())
      detail::check(agp_predict_joint(fit_.context->ctx, k.k, fit_.handle.get(), &f.view, out.mean.data(),
                                      out.covariance.data.data(), AGP_HOST),
                    fit_.context->ctx, "agp_predict_joint");
    model_.add_mean(xs, &out.mean);
    return out;
  }

 private:
  ModelType model_;
  GPFit<FeatureType> fit_;
};

// ---------------------------------------------------------------------------
// update(): Fit<GPFit<BlockSymmetric<Solver>, F>> (gp.hpp:384-414).  The updated
// fit predicts through the generic CovarianceRepresentation form of _predict_impl
// (gp.hpp:305-366): device Gram + solver.solve().
// ---------------------------------------------------------------------------
template <typename ModelType, typename FeatureType, typename Solver>
class UpdatedFitModel {
 public:
  UpdatedFitModel(const ModelType &model, std::vector<FeatureType> features, BlockSymmetric<Solver> cov, Vector info)
      : train_features(std::move(features)), train_covariance(std::move(cov)), information(std::move(info)), model_(model) {}

  std::int64_t rows() const { return train_covariance.rows(); }
  Matrix solve(const Matrix &rhs) const { return train_covariance.solve(rhs); }

  // _predict_impl over a generic CovarianceRepresentation (gp.hpp:305-366), in HBM: agp_solver_predict
  JointDistribution predict_joint(const std::vector<FeatureType> &xs) const {
    JointDistribution out;
    out.mean.assign(xs.size(), 0.);
    out.covariance = Matrix(static_cast<std::int64_t>(xs.size()), static_cast<std::int64_t>(xs.size()));
    device_predict(xs, &out.mean, out.covariance.data.data(), 2);
    return out;
  }
  Vector predict_mean(const std::vector<FeatureType> &xs) const {
    Vector mean(xs.size(), 0.);
    device_predict(xs, &mean, nullptr, 0);
    return mean;
  }

  // a further update nests the solvers, exactly like the reference's types do
  UpdatedFitModel<ModelType, FeatureType, BlockSymmetric<Solver>> update(const RegressionDataset<FeatureType> &d) const {
    return update_impl<ModelType, FeatureType, BlockSymmetric<Solver>>(model_, *this, train_covariance, train_features,
                                                                        information, d);
  }

  template <typename M2, typename F2, typename S2, typename Self>
  static UpdatedFitModel<M2, F2, S2> update_impl(const M2 &model, const Self &self, const S2 &solver,
                                                 const std::vector<F2> &old_features, const Vector &old_information,
                                                 const RegressionDataset<F2> &d) {
    JointDistribution pred = self.predict_joint(d.features);                       // gp.hpp:388-389
    const std::size_t m = d.features.size(), n = old_features.size();
    Vector delta(m);
    for (std::size_t i = 0; i < m; ++i) delta[i] = d.targets.mean[i] - pred.mean[i];
    if (!d.targets.covariance.empty())
      for (std::size_t i = 0; i < m; ++i) pred.covariance(static_cast<std::int64_t>(i), static_cast<std::int64_t>(i)) += d.targets.covariance[i];
    const SerializableLDLT S_ldlt(pred.covariance);                                 // gp.hpp:393
    const Matrix cross = model.get_covariance()(old_features, d.features);          // gp.hpp:395-396
    BlockSymmetric<S2> new_cov(solver, cross, S_ldlt);                              // gp.hpp:398-399
    const Vector Si_delta = S_ldlt.solve(delta);
    // [information - Ai_B Si_delta ; Si_delta] (gp.hpp:403-407) from the Ai_B the new solver holds in HBM: one mat-vec on the device
    Vector info(n + m);
    detail::check(agp_solver_update_information(new_cov.context()->ctx, new_cov.device_solver().get(), old_information.data(), Si_delta.data(),
                                                info.data(), AGP_HOST),
                  new_cov.context()->ctx, "agp_solver_update_information");
    std::vector<F2> feats = old_features;
    feats.insert(feats.end(), d.features.begin(), d.features.end());
    return UpdatedFitModel<M2, F2, S2>(model, std::move(feats), std::move(new_cov), std::move(info));
  }

  std::vector<FeatureType> train_features;
  BlockSymmetric<Solver> train_covariance;
  Vector information;

 private:
  template <typename P>
  void device_predict(const std::vector<P> &xs, Vector *mean, double *second, int mode) const {
    if (xs.empty()) return;
    const auto &ctx = train_covariance.context();
    detail::KernelHolder k(model_.get_covariance().program());
    detail::Flat ftr = detail::flatten(model_.get_covariance(), train_features);
    detail::Flat fxs = detail::flatten(model_.get_covariance(), xs);
    detail::check(agp_solver_predict(ctx->ctx, k.k, train_covariance.device_solver().get(), &ftr.view, information.data(), &fxs.view,
                                     mean->data(), second, mode, AGP_HOST),
                  ctx->ctx, "agp_solver_predict");
    model_.add_mean(xs, mean);
  }
  ModelType model_;
};

// covariance_functions/representations.hpp:64-96: S^-1 = A^-1 B A^-1, A through its factor, B kept as it is
struct ExplainedCovariance {
  ExplainedCovariance() = default;
  ExplainedCovariance(const SerializableLDLT &outer_ldlt_, const Matrix &inner_) : outer_ldlt(outer_ldlt_), inner(inner_) { make(); }
  ExplainedCovariance(const Matrix &outer, const Matrix &inner_) : outer_ldlt(outer), inner(inner_) { make(); }
  std::int64_t rows() const { return inner.rows(); }
  std::int64_t cols() const { return inner.cols(); }
  // :80-82 - outer^-1 (inner (outer^-1 rhs)), the product with the inner matrix on the MFMA between the two device solves
  Matrix solve(const Matrix &rhs) const { return detail::solver_solve(outer_ldlt.context(), handle_.get(), rhs); }
  std::shared_ptr<agp_solver> device_solver() const { return handle_; }
  const std::shared_ptr<detail::ContextHolder> &context() const { return outer_ldlt.context(); }
  SerializableLDLT outer_ldlt;
  Matrix inner;

 private:
  void make() {
    sub_ = outer_ldlt.device_solver();
    agp_solver *sv = nullptr;
    detail::check(agp_solver_explained(outer_ldlt.context()->ctx, sub_.get(), inner.data.data(), inner.rows(), AGP_HOST, &sv),
                  outer_ldlt.context()->ctx, "agp_solver_explained");
    auto keep = sub_;
    handle_ = std::shared_ptr<agp_solver>(sv, [keep](agp_solver *p) { agp_solver_destroy(p); });
  }
  std::shared_ptr<agp_solver> sub_, handle_;
};

// FitModel over a Fit<GPFit<Representation, F>> whose solver is any CovarianceRepresentation
// (gp.hpp:42-45): predictions through the generic _predict_impl (gp.hpp:305-366), device Gram + solve().
template <typename ModelType, typename FeatureType, typename Representation>
class RepresentationFitModel {
 public:
  RepresentationFitModel(const ModelType &model, std::vector<FeatureType> features, Representation cov, Vector info)
      : train_features(std::move(features)), train_covariance(std::move(cov)), information(std::move(info)), model_(model) {}

  // _predict_impl over a generic CovarianceRepresentation (gp.hpp:305-366) in HBM: agp_solver_predict, or - LinearCombination
  // features on either side (callers.hpp:321-396) - agp_solver_predict_combined
  template <typename P>
  Vector predict_mean(const std::vector<P> &xs) const {
    Vector mean(xs.size(), 0.);
    device_predict(xs, &mean, nullptr, 0);
    return mean;
  }
  template <typename P>
  JointDistribution predict_joint(const std::vector<P> &xs) const {
    JointDistribution out;
    out.mean.assign(xs.size(), 0.);
    out.covariance = Matrix(static_cast<std::int64_t>(xs.size()), static_cast<std::int64_t>(xs.size()));
    device_predict(xs, &out.mean, out.covariance.data.data(), 2);
    return out;
  }

  std::vector<FeatureType> train_features;
  Representation train_covariance;
  Vector information;

 private:
  // one side of a prediction: the (expanded) points, and - for LinearCombination features - offsets and coefficients
  template <typename F>
  struct Side {
    detail::Flat flat;
    std::vector<std::int64_t> offsets;
    std::vector<double> coefficients;
    std::int64_t count = 0;
    bool combined = false;
  };
  template <typename F>
  Side<F> side_of(const std::vector<F> &features) const {
    Side<F> sd;
    if constexpr (detail::expansion<F>::expands) {
      const detail::Expanded<F> ex(features);
      sd.flat = detail::flatten(model_.get_covariance(), ex.points);
      sd.offsets = ex.offsets();
      sd.coefficients = ex.coefficient;
      sd.count = static_cast<std::int64_t>(ex.n);
      sd.combined = true;
    } else {
      sd.flat = detail::flatten(model_.get_covariance(), features);
      sd.count = sd.flat.view.n;
    }
    return sd;
  }
  template <typename P>
  void device_predict(const std::vector<P> &xs, Vector *mean, double *second, int mode) const {
    if (xs.empty()) return;
    auto ctx = detail::default_context();
    detail::KernelHolder k(model_.get_covariance().program());
    const auto tr = side_of(train_features);
    const auto te = side_of(xs);
    auto solver = train_covariance.device_solver();
    if (tr.combined || te.combined) {
      detail::check(agp_solver_predict_combined(ctx->ctx, k.k, solver.get(), &tr.flat.view, tr.count, tr.combined ? tr.offsets.data() : nullptr,
                                                tr.combined ? tr.coefficients.data() : nullptr, information.data(), &te.flat.view, te.count,
                                                te.combined ? te.offsets.data() : nullptr, te.combined ? te.coefficients.data() : nullptr,
                                                mean->data(), second, mode, AGP_HOST),
                    ctx->ctx, "agp_solver_predict_combined");
    } else {
      detail::check(agp_solver_predict(ctx->ctx, k.k, solver.get(), &tr.flat.view, information.data(), &te.flat.view, mean->data(), second, mode,
                                       AGP_HOST),
                    ctx->ctx, "agp_solver_predict");
    }
    model_.add_mean(xs, mean);
  }
  ModelType model_;
};

// The reference's route for covariances that are only positive semi-definite ("unobservable" models,
// tests/test_gp.cc:20-33): Fit<GPFit<SerializableLDLT>> with the pivoted factor (gp.hpp:61-69).  model.fit()
// reports such inputs as "not positive definite"; this is the explicit fallback.
template <typename ModelType, typename FeatureType>
RepresentationFitModel<ModelType, FeatureType, PivotedLDLT> fit_pivoted(const ModelType &model,
                                                                        const RegressionDataset<FeatureType> &dataset) {
  Matrix K = model.get_covariance()(as_measurements(dataset.features));  // gp.hpp:288-290
  if (!dataset.targets.covariance.empty())
    for (std::size_t i = 0; i < dataset.features.size(); ++i)
      K(static_cast<std::int64_t>(i), static_cast<std::int64_t>(i)) += dataset.targets.covariance[i];  // gp.hpp:65
  Vector y = dataset.targets.mean;
  Vector zero(y.size(), 0.);
  model.add_mean(dataset.features, &zero);  // remove_from == subtract what add_to adds
  for (std::size_t i = 0; i < y.size(); ++i) y[i] -= zero[i];
  PivotedLDLT ldlt(K);                                                   // gp.hpp:67
  Vector info = ldlt.solve(y);                                           // gp.hpp:68
  return RepresentationFitModel<ModelType, FeatureType, PivotedLDLT>(model, dataset.features, std::move(ldlt), std::move(info));
}

// update(fit_model, dataset), core/fit_model.hpp:117-120 -> _update_impl, gp.hpp:384-414.  The reference returns a fit
// whose solver is BlockSymmetric<Solver>; here the RESIDENT factor grows by one block row on the device (agp_fit_update:
// triangular solve + SYRK on MFMA, LL^T of the Schur complement), so the result is an ordinary FitModel again: nested
// updates, predictions and solves all stay on the device factor.  (UpdatedFitModel above remains for solvers that are
// not a device factor: PivotedLDLT, ExplainedCovariance.)
template <typename ModelType, typename FeatureType>
FitModel<ModelType, FeatureType> update(const FitModel<ModelType, FeatureType> &fm, const RegressionDataset<FeatureType> &d) {
  if (d.features.size() != d.targets.size()) throw std::invalid_argument("features and targets differ in size");
  const ModelType &model = fm.get_model();
  const GPFit<FeatureType> &old = fm.get_fit();
  detail::KernelHolder k(model.get_covariance().program());
  detail::Flat f = detail::flatten(model.get_covariance(), d.features);
  Vector y = d.targets.mean;
  Vector zero(y.size(), 0.);
  model.add_mean(d.features, &zero);  // remove_from == subtract what add_to adds (ModelBase::update)
  for (std::size_t i = 0; i < y.size(); ++i) y[i] -= zero[i];
  GPFit<FeatureType> fit;
  fit.train_features = old.train_features;                                   // concatenate(...), gp.hpp:387
  fit.train_features.insert(fit.train_features.end(), d.features.begin(), d.features.end());
  fit.information.resize(fit.train_features.size());
  fit.context = old.context;
  agp_fit *h = nullptr;
  const double *yvar = d.targets.covariance.empty() ? nullptr : d.targets.covariance.data();
  const int st = agp_fit_update(old.context->ctx, k.k, old.handle.get(), &f.view, y.data(), yvar, &h, fit.information.data(),
                                &fit.log_determinant);
  if (st != AGP_OK) {
    const long long pivot = h ? static_cast<long long>(agp_fit_failed_pivot(h)) : -1;
    agp_fit_destroy(h);
    std::string what = "agp_fit_update";
    if (st == AGP_ERR_NOT_POSITIVE_DEFINITE) what += " (pivot " + std::to_string(pivot) + ")";
    detail::check(st, old.context->ctx, what.c_str());
  }
  auto ctx = old.context;
  fit.handle = std::shared_ptr<agp_fit>(h, [ctx](agp_fit *p) { agp_fit_destroy(p); });
  return FitModel<ModelType, FeatureType>(model, std::move(fit));
}

template <typename ModelType, typename FeatureType>
FitModel<ModelType, FeatureType> FitModel<ModelType, FeatureType>::update(const RegressionDataset<FeatureType> &dataset) const {
  return albatross::update(*this, dataset);
}

// ---------------------------------------------------------------------------
// Several GPUs: one process per GPU, one Communicator per process (agp_comm_*: RCCL over xGMI inside the library).
// Not in the reference (a single-process library); `model.fit(dataset, comm)` is the same fit with the Gram matrix
// and its factorisation sharded row-block-wise over the ranks (gp.hpp:61-69, 281-294).
// ---------------------------------------------------------------------------
// The GPU of this process (call before anything else touches the library; default 0).  With one process per GPU:
// set_device(local_rank).
inline void set_device(int device) { detail::default_device() = device; }

class Communicator {
 public:
  using UniqueId = std::array<unsigned char, AGP_COMM_ID_BYTES>;
  // ncclGetUniqueId: rank 0 calls it and hands the bytes to every rank by any means (MPI, a file, a TCP store)
  static UniqueId unique_id() {
    UniqueId id{};
    detail::check(agp_comm_unique_id(id.data()), nullptr, "agp_comm_unique_id");
    return id;
  }
  // collective over all ranks (ncclCommInitRank on this process's GPU)
  Communicator(int nranks, int rank, const UniqueId &id) : context_(detail::default_context()) {
    agp_comm *c = nullptr;
    detail::check(agp_comm_create(context_->ctx, nranks, rank, id.data(), &c), context_->ctx, "agp_comm_create");
    auto ctx = context_;
    handle_ = std::shared_ptr<agp_comm>(c, [ctx](agp_comm *p) { agp_comm_destroy(p); });
  }
  int size() const { return agp_comm_size(handle_.get()); }
  int rank() const { return agp_comm_rank(handle_.get()); }
  void barrier() const { detail::check(agp_comm_barrier(handle_.get()), context_->ctx, "agp_comm_barrier"); }
  // in place on host doubles; every rank receives the result
  void all_reduce_sum(Vector *v) const {
    detail::check(agp_comm_all_reduce_host(handle_.get(), v->data(), static_cast<std::int64_t>(v->size()), 0), context_->ctx,
                  "agp_comm_all_reduce_host");
  }
  agp_comm *handle() const { return handle_.get(); }

 private:
  std::shared_ptr<detail::ContextHolder> context_;
  std::shared_ptr<agp_comm> handle_;
};

// ---------------------------------------------------------------------------
// GaussianProcessRegression (gp.hpp:170-505)
// ---------------------------------------------------------------------------
template <typename CovFunc, typename MeanFunc = ZeroMean>
class GaussianProcessRegression {
 public:
  GaussianProcessRegression() = default;
  explicit GaussianProcessRegression(const CovFunc &cov, const std::string &name = "gaussian_process_regression")
      : covariance_function_(cov), model_name_(name) {}
  GaussianProcessRegression(const CovFunc &cov, const MeanFunc &mean, const std::string &name = "gaussian_process_regression")
      : covariance_function_(cov), mean_function_(mean), model_name_(name) {}

  std::string get_name() const { return model_name_; }
  const CovFunc &get_covariance() const { return covariance_function_; }
  const MeanFunc &get_mean() const { return mean_function_; }

  ParameterStore get_params() const {  // gp.hpp:255-258
    ParameterStore p = covariance_function_.get_params();
    for (const auto &kv : mean_function_.get_params()) p[kv.first] = kv.second;
    return p;
  }
  void set_param(const std::string &n, double v) {  // gp.hpp:260-268
    if (covariance_function_.has_param(n)) covariance_function_.set_param(n, v);
    else if (mean_function_.has_param(n)) mean_function_.set_param(n, v);
    else throw std::out_of_range("unknown parameter " + n);
  }
  void set_param_values(const ParameterStore &values) {
    for (const auto &kv : values) set_param(kv.first, kv.second);
  }

  template <typename P>
  void add_mean(const std::vector<P> &xs, Vector *mean) const {  // mean_function_.add_to, mean_function.hpp:86-95
    if (std::is_same<MeanFunc, ZeroMean>::value) return;
    for (std::size_t i = 0; i < xs.size(); ++i) (*mean)[i] += detail::mean_at(mean_function_, xs[i]);
  }

  // Datasets of LinearCombination features: Fit<GPFit<SerializableLDLT>> (gp.hpp:61-69) from the contracted
  // covariance matrix; Gram of the expanded points, factorisation and solves on the device.
  template <typename X>
  RepresentationFitModel<GaussianProcessRegression, LinearCombination<X>, SerializableLDLT> fit(
      const std::vector<LinearCombination<X>> &features, const MarginalDistribution &targets) const {
    if (features.size() != targets.size()) throw std::invalid_argument("features and targets differ in size");
    Matrix K = covariance_function_(as_measurements(features));  // gp.hpp:288-290
    if (!targets.covariance.empty())
      for (std::size_t i = 0; i < features.size(); ++i)
        K(static_cast<std::int64_t>(i), static_cast<std::int64_t>(i)) += targets.covariance[i];  // gp.hpp:65
    Vector y = targets.mean;
    Vector m(y.size(), 0.);
    add_mean(features, &m);  // mean_function_.remove_from, gp.hpp:291-292
    for (std::size_t i = 0; i < y.size(); ++i) y[i] -= m[i];
    SerializableLDLT ldlt(K);        // gp.hpp:67
    Vector info = ldlt.solve(y);     // gp.hpp:68
    return RepresentationFitModel<GaussianProcessRegression, LinearCombination<X>, SerializableLDLT>(*this, features, std::move(ldlt),
                                                                                                   std::move(info));
  }
  template <typename X>
  auto fit(const RegressionDataset<LinearCombination<X>> &dataset) const {
    return fit(dataset.features, dataset.targets);
  }

  // ModelBase::fit (core/model.hpp:137-152) -> _fit_impl (gp.hpp:281-294)
  template <typename FeatureType>
  FitModel<GaussianProcessRegression, FeatureType> fit(const RegressionDataset<FeatureType> &dataset) const {
    return fit(dataset.features, dataset.targets);
  }

  template <typename FeatureType>
  FitModel<GaussianProcessRegression, FeatureType> fit(const std::vector<FeatureType> &features,
                                                       const MarginalDistribution &targets) const {
    if (features.size() != targets.size()) throw std::invalid_argument("features and targets differ in size");
    if (!targets.covariance.empty() && targets.covariance.size() != targets.size())
      throw std::invalid_argument("target covariance must be diagonal (one variance per target)");
    auto ctx = detail::default_context();
    detail::KernelHolder k(covariance_function_.program());
    detail::Flat f = detail::flatten(covariance_function_, features);  // as_measurements happens inside agp_fit_create
    Vector y = targets.mean;                                            // mean_function_.remove_from, gp.hpp:291-292
    if (!std::is_same<MeanFunc, ZeroMean>::value)
      for (std::size_t i = 0; i < y.size(); ++i) y[i] -= mean_function_._call_impl(detail::unwrap<FeatureType>::get(features[i]));
    GPFit<FeatureType> fit;
    fit.train_features = features;
    fit.information.resize(features.size());
    fit.context = ctx;
    agp_fit *h = nullptr;
    const double *yvar = targets.covariance.empty() ? nullptr : targets.covariance.data();
    int st;
    if (mixed_precision.enabled) {
      // fp32 MFMA products in the bulk updates of the factorisation, information vector refined to fp64
      st = agp_fit_create_mixed(ctx->ctx, k.k, &f.view, y.data(), yvar, mixed_precision.max_iterations,
                                mixed_precision.tolerance, &h, fit.information.data(), &fit.log_determinant,
                                &mixed_precision.iterations, &mixed_precision.residual);
    } else {
      st = agp_fit_create(ctx->ctx, k.k, &f.view, y.data(), yvar, &h, fit.information.data(), &fit.log_determinant);
    }
    if (st != AGP_OK) {
      const long long pivot = h ? static_cast<long long>(agp_fit_failed_pivot(h)) : -1;
      agp_fit_destroy(h);
      std::string what = "agp_fit_create";
      if (st == AGP_ERR_NOT_POSITIVE_DEFINITE) what += " (pivot " + std::to_string(pivot) + ")";
      detail::check(st, ctx->ctx, what.c_str());
    }
    fit.handle = std::shared_ptr<agp_fit>(h, [ctx](agp_fit *p) { agp_fit_destroy(p); });
    return FitModel<GaussianProcessRegression, FeatureType>(*this, std::move(fit));
  }

  // `fit` for SEVERAL datasets of one size in lock step (agp_fit_create_batch): the regime of the reference's own workloads
  // (benchmarks/bench_predict.cc:20-40: N = 512; one fit per tuner step), where a single fit is bound by the latency of its
  // serial pivots - a batch shares it.  Every dataset is fitted with THIS model (its current parameters); throws like `fit`
  // for the first dataset whose covariance has NaN or is not positive definite.
  template <typename FeatureType>
  std::vector<FitModel<GaussianProcessRegression, FeatureType>> fit_batch(const std::vector<RegressionDataset<FeatureType>> &datasets) const {
    std::vector<FitModel<GaussianProcessRegression, FeatureType>> out;
    if (datasets.empty()) return out;
    if (mixed_precision.enabled) throw std::invalid_argument("fit_batch: fp64 models only");
    const std::size_t count = datasets.size(), n = datasets[0].features.size();
    auto ctx = detail::default_context();
    detail::KernelHolder k(covariance_function_.program());
    std::vector<detail::Flat> flats;
    flats.reserve(count);
    Vector y(n * count);
    bool have_var = false;
    for (const auto &d : datasets) have_var = have_var || !d.targets.covariance.empty();
    Vector yv(have_var ? n * count : 0, 0.);
    for (std::size_t b = 0; b < count; ++b) {
      const auto &d = datasets[b];
      if (d.features.size() != n || d.targets.size() != n) throw std::invalid_argument("fit_batch: every dataset must have the same number of points");
      if (!d.targets.covariance.empty() && d.targets.covariance.size() != n)
        throw std::invalid_argument("target covariance must be diagonal (one variance per target)");
      flats.push_back(detail::flatten(covariance_function_, d.features));
      for (std::size_t i = 0; i < n; ++i) {
        double v = d.targets.mean[i];  // mean_function_.remove_from, gp.hpp:291-292
        if (!std::is_same<MeanFunc, ZeroMean>::value) v -= mean_function_._call_impl(detail::unwrap<FeatureType>::get(d.features[i]));
        y[b * n + i] = v;
        if (have_var && !d.targets.covariance.empty()) yv[b * n + i] = d.targets.covariance[i];
      }
    }
    std::vector<const agp_kernel *> kernels(count, k.k);
    std::vector<const agp_features *> views(count);
    for (std::size_t b = 0; b < count; ++b) views[b] = &flats[b].view;
    std::vector<agp_fit *> handles(count, nullptr);
    std::vector<int> status(count, AGP_OK);
    std::vector<double> info(n * count), logdet(count);
    const int st = agp_fit_create_batch(ctx->ctx, (int)count, kernels.data(), views.data(), y.data(), (std::int64_t)n,
                                        have_var ? yv.data() : nullptr, (std::int64_t)n, handles.data(), info.data(), (std::int64_t)n,
                                        logdet.data(), status.data());
    detail::check(st, ctx->ctx, "agp_fit_create_batch");
    std::vector<std::shared_ptr<agp_fit>> owned;
    for (agp_fit *h : handles) owned.emplace_back(h, [ctx](agp_fit *p) { agp_fit_destroy(p); });
    for (std::size_t b = 0; b < count; ++b)
      if (status[b] != AGP_OK) {
        std::string what = "agp_fit_create_batch: problem " + std::to_string(b);
        if (status[b] == AGP_ERR_NOT_POSITIVE_DEFINITE) what += " (pivot " + std::to_string((long long)agp_fit_failed_pivot(handles[b])) + ")";
        detail::check(status[b], ctx->ctx, what.c_str());
      }
    for (std::size_t b = 0; b < count; ++b) {
      GPFit<FeatureType> fit;
      fit.train_features = datasets[b].features;
      fit.information.assign(info.begin() + (std::ptrdiff_t)(b * n), info.begin() + (std::ptrdiff_t)((b + 1) * n));
      fit.log_determinant = logdet[b];
      fit.context = ctx;
      fit.handle = owned[b];
      out.emplace_back(*this, std::move(fit));
    }
    return out;
  }

  // The same fit over the GPUs of a communicator: every rank passes the SAME dataset, the Gram matrix and its LL^T are
  // sharded row-block-wise (agp_sharded_fit_create), then the factor is replicated (agp_sharded_fit_replicate) so that
  // every rank holds an ordinary FitModel and predicts its own share of the test points - predictions are independent
  // per point (gp.hpp:82-113), nothing further is exchanged.  Collective.
  template <typename FeatureType>
  FitModel<GaussianProcessRegression, FeatureType> fit(const RegressionDataset<FeatureType> &dataset,
                                                       const Communicator &comm) const {
    const std::vector<FeatureType> &features = dataset.features;
    const MarginalDistribution &targets = dataset.targets;
    if (features.size() != targets.size()) throw std::invalid_argument("features and targets differ in size");
    auto ctx = detail::default_context();
    detail::KernelHolder k(covariance_function_.program());
    detail::Flat f = detail::flatten(covariance_function_, features);
    Vector y = targets.mean;  // mean_function_.remove_from, gp.hpp:291-292
    if (!std::is_same<MeanFunc, ZeroMean>::value)
      for (std::size_t i = 0; i < y.size(); ++i) y[i] -= mean_function_._call_impl(detail::unwrap<FeatureType>::get(features[i]));
    GPFit<FeatureType> fit;
    fit.train_features = features;
    fit.information.resize(features.size());
    fit.context = ctx;
    const double *yvar = targets.covariance.empty() ? nullptr : targets.covariance.data();
    agp_sharded_fit *sf = nullptr;
    int st = agp_sharded_fit_create(ctx->ctx, comm.handle(), k.k, &f.view, y.data(), yvar, &sf, fit.information.data(),
                                    &fit.log_determinant);
    if (st != AGP_OK) {
      const long long pivot = sf ? static_cast<long long>(agp_sharded_fit_failed_pivot(sf)) : -1;
      agp_sharded_fit_destroy(sf);
      std::string what = "agp_sharded_fit_create";
      if (st == AGP_ERR_NOT_POSITIVE_DEFINITE) what += " (pivot " + std::to_string(pivot) + ")";
      detail::check(st, ctx->ctx, what.c_str());
    }
    agp_fit *h = nullptr;
    st = agp_sharded_fit_replicate(ctx->ctx, sf, &h);
    agp_sharded_fit_destroy(sf);
    detail::check(st, ctx->ctx, "agp_sharded_fit_replicate");
    fit.handle = std::shared_ptr<agp_fit>(h, [ctx](agp_fit *p) { agp_fit_destroy(p); });
    return FitModel<GaussianProcessRegression, FeatureType>(*this, std::move(fit));
  }

  // core/model.hpp:154-156
  auto cross_validate() const;

  // Not in the reference: opt-in mixed-precision fit (agp_fit_create_mixed; BASELINE configs[3]).  `iterations` and
  // `residual` report the refinement of the last fit.
  struct MixedPrecision {
    bool enabled = false;
    int max_iterations = 50;
    double tolerance = 1e-12;
    mutable int iterations = 0;
    mutable double residual = 0.;
  };
  MixedPrecision mixed_precision;

  // fit_from_prediction (gp.hpp:236-245) -> gp_fit_from_prediction (gp.hpp:139-153): the model that reproduces a
  // joint prediction at `features`
  template <typename FeatureType>
  RepresentationFitModel<GaussianProcessRegression, FeatureType, ExplainedCovariance> fit_from_prediction(
      const std::vector<FeatureType> &features, const JointDistribution &prediction) const {
    Vector mean = prediction.mean;  // mean_function_.remove_from, :240
    if (!std::is_same<MeanFunc, ZeroMean>::value)
      for (std::size_t i = 0; i < mean.size(); ++i) mean[i] -= mean_function_._call_impl(detail::unwrap<FeatureType>::get(features[i]));
    const Matrix prior = covariance_function_(features);  // :243
    const SerializableLDLT prior_ldlt(prior);
    Matrix inner = prior;
    for (std::size_t e = 0; e < inner.data.size(); ++e) inner.data[e] -= prediction.covariance.data[e];
    return RepresentationFitModel<GaussianProcessRegression, FeatureType, ExplainedCovariance>(
        *this, features, ExplainedCovariance(prior_ldlt, inner), prior_ldlt.solve(mean));
  }

  // log_likelihood(dataset) for several parameter vectors in ONE batched device pass (agp_nll_batch): the
  // evaluations compute_gradient (tune/finite_difference.hpp:20-94) and ModelTuner (tune/tune.hpp:151-161)
  // make one after the other.  Every entry overrides some of the current parameters; a parameter vector whose
  // covariance is not positive definite yields NaN.
  template <typename FeatureType>
  Vector log_likelihoods(const RegressionDataset<FeatureType> &dataset, const std::vector<ParameterStore> &parameter_sets) const {
    const std::size_t count = parameter_sets.size(), n = dataset.features.size();
    Vector out(count);
    if (count == 0) return out;
    auto ctx = detail::default_context();
    std::vector<GaussianProcessRegression> models(count, *this);
    std::vector<std::unique_ptr<detail::KernelHolder>> kernels;
    std::vector<detail::Flat> flats(count);
    std::vector<const agp_kernel *> kptr(count);
    std::vector<const agp_features *> fptr(count);
    std::vector<double> Y(n * count);
    for (std::size_t b = 0; b < count; ++b) {
      models[b].set_param_values(parameter_sets[b]);
      kernels.emplace_back(new detail::KernelHolder(models[b].covariance_function_.program()));
      flats[b] = detail::flatten(models[b].covariance_function_, dataset.features);
      kptr[b] = kernels.back()->k;
      for (std::size_t i = 0; i < n; ++i) {
        double v = dataset.targets.mean[i];
        if (!std::is_same<MeanFunc, ZeroMean>::value)
          v -= models[b].mean_function_._call_impl(detail::unwrap<FeatureType>::get(dataset.features[i]));
        Y[b * n + i] = v;
      }
    }
    for (std::size_t b = 0; b < count; ++b) fptr[b] = &flats[b].view;  // after the vector stopped moving
    // like log_likelihood below: the target variance is NOT part of the covariance (gp.hpp:442-451)
    detail::check(agp_nll_batch(ctx->ctx, static_cast<int>(count), kptr.data(), fptr.data(), Y.data(), static_cast<std::int64_t>(n),
                                nullptr, out.data()),
                  ctx->ctx, "agp_nll_batch");
    for (double &v : out) v = -v;
    return out;
  }

  // gp.hpp:442-451 (prior_log_likelihood() is outside the hot path and not included).  As in the reference the
  // covariance is covariance_function_(measurement_features) alone: dataset.targets.covariance is NOT added.
  template <typename FeatureType>
  double log_likelihood(const RegressionDataset<FeatureType> &dataset) const {
    auto ctx = detail::default_context();
    detail::KernelHolder k(covariance_function_.program());
    detail::Flat f = detail::flatten(covariance_function_, dataset.features);
    Vector y = dataset.targets.mean;
    if (!std::is_same<MeanFunc, ZeroMean>::value)
      for (std::size_t i = 0; i < y.size(); ++i)
        y[i] -= mean_function_._call_impl(detail::unwrap<FeatureType>::get(dataset.features[i]));
    double nll = 0.;
    detail::check(agp_nll(ctx->ctx, k.k, &f.view, y.data(), nullptr, &nll), ctx->ctx, "agp_nll");
    return -nll;
  }

 private:
  CovFunc covariance_function_;
  MeanFunc mean_function_;
  std::string model_name_ = "gaussian_process_regression";
};

// ---------------------------------------------------------------------------
// Cross validation: indexing/group_by.hpp (LeaveOneOutGrouper, GroupIndexer),
// evaluation/cross_validation.hpp:28-330 (model.cross_validate().predict(dataset, grouper)),
// GP fast path gp_cross_validated_predictions, models/gp.hpp:465-482
// ---------------------------------------------------------------------------
struct LeaveOneOutGrouper {};

template <typename GroupKey>
using GroupIndexer = std::map<GroupKey, std::vector<std::size_t>>;

template <typename FeatureType, typename Grouper>
auto group_indexer(const std::vector<FeatureType> &features, const Grouper &grouper)
    -> GroupIndexer<decltype(grouper(features[0]))> {
  GroupIndexer<decltype(grouper(features[0]))> out;
  for (std::size_t i = 0; i < features.size(); ++i) out[grouper(features[i])].push_back(i);
  return out;
}

template <typename FeatureType>
GroupIndexer<std::size_t> group_indexer(const std::vector<FeatureType> &features, const LeaveOneOutGrouper &) {
  GroupIndexer<std::size_t> out;
  for (std::size_t i = 0; i < features.size(); ++i) out[i] = {i};
  return out;
}

template <typename ModelType, typename FeatureType, typename GroupKey>
class CrossValidationPrediction {
 public:
  CrossValidationPrediction(const ModelType &model, const RegressionDataset<FeatureType> &dataset,
                            const GroupIndexer<GroupKey> &indexer)
      : model_(model), dataset_(dataset), indexer_(indexer) {}

  const GroupIndexer<GroupKey> &indexer() const { return indexer_; }

  // ONE fit, then held_out_predictions (gp.hpp:465-482)
  std::map<GroupKey, JointDistribution> joints() const {
    std::vector<std::vector<std::size_t>> groups;
    for (const auto &kv : indexer_) groups.push_back(kv.second);
    const auto fit_model = model_.fit(dataset_);
    const auto preds = fit_model.get_fit().held_out_predictions(dataset_.targets.mean, groups);
    std::map<GroupKey, JointDistribution> out;
    std::size_t g = 0;
    for (const auto &kv : indexer_) out[kv.first] = preds[g++];
    return out;
  }
  std::map<GroupKey, MarginalDistribution> marginals() const {
    std::map<GroupKey, MarginalDistribution> out;
    for (const auto &kv : joints()) out[kv.first] = kv.second.marginal();
    return out;
  }
  std::map<GroupKey, Vector> means() const {
    std::map<GroupKey, Vector> out;
    for (const auto &kv : joints()) out[kv.first] = kv.second.mean;
    return out;
  }
  // concatenate_*_predictions: group results scattered back to dataset order
  MarginalDistribution marginal() const {
    MarginalDistribution out(Vector(dataset_.size()), Vector(dataset_.size()));
    const auto m = marginals();
    for (const auto &kv : indexer_) {
      const auto &p = m.at(kv.first);
      for (std::size_t a = 0; a < kv.second.size(); ++a) {
        out.mean[kv.second[a]] = p.mean[a];
        out.covariance[kv.second[a]] = p.covariance[a];
      }
    }
    return out;
  }
  Vector mean() const { return marginal().mean; }

  // generic path (cross_validation.hpp:20-43): refit on the other groups, predict the held-out one
  std::map<GroupKey, JointDistribution> predictions() const {
    std::map<GroupKey, JointDistribution> out;
    for (const auto &kv : indexer_) {
      std::vector<bool> held(dataset_.size(), false);
      for (std::size_t i : kv.second) held[i] = true;
      RegressionDataset<FeatureType> train;
      std::vector<FeatureType> test;
      const bool has_var = !dataset_.targets.covariance.empty();
      for (std::size_t i = 0; i < dataset_.size(); ++i) {
        if (held[i]) continue;
        train.features.push_back(dataset_.features[i]);
        train.targets.mean.push_back(dataset_.targets.mean[i]);
        if (has_var) train.targets.covariance.push_back(dataset_.targets.covariance[i]);
      }
      for (std::size_t i : kv.second) test.push_back(dataset_.features[i]);
      out[kv.first] = model_.fit(train).predict(test).joint();
    }
    return out;
  }

 private:
  ModelType model_;
  RegressionDataset<FeatureType> dataset_;
  GroupIndexer<GroupKey> indexer_;
};

template <typename ModelType>
class CrossValidation {
 public:
  explicit CrossValidation(const ModelType &model) : model_(model) {}
  template <typename FeatureType, typename Grouper>
  auto predict(const RegressionDataset<FeatureType> &dataset, const Grouper &grouper) const {
    auto indexer = group_indexer(dataset.features, grouper);
    using Key = typename decltype(indexer)::key_type;
    return CrossValidationPrediction<ModelType, FeatureType, Key>(model_, dataset, indexer);
  }

 private:
  ModelType model_;
};

template <typename ModelType>
CrossValidation<ModelType> cross_validate(const ModelType &model) { return CrossValidation<ModelType>(model); }

template <typename CovFunc, typename MeanFunc>
auto GaussianProcessRegression<CovFunc, MeanFunc>::cross_validate() const {
  return CrossValidation<GaussianProcessRegression<CovFunc, MeanFunc>>(*this);
}

// factories, gp.hpp:507-537
template <typename CovFunc>
GaussianProcessRegression<CovFunc, ZeroMean> gp_from_covariance(const CovFunc &cov,
                                                                const std::string &name = "gaussian_process_regression") {
  return GaussianProcessRegression<CovFunc, ZeroMean>(cov, name);
}

template <typename CovFunc, typename MeanFunc>
GaussianProcessRegression<CovFunc, MeanFunc> gp_from_covariance_and_mean(
    const CovFunc &cov, const MeanFunc &mean, const std::string &name = "gaussian_process_regression") {
  return GaussianProcessRegression<CovFunc, MeanFunc>(cov, mean, name);
}

// ---------------------------------------------------------------------------
// SparseGaussianProcessRegression (models/sparse_gp.hpp:245-797): FITC / PITC.
// Grouping, reordering and the inducing-point strategy are host bookkeeping exactly as in
// compute_internal_components (:631-706); K_uu, K_fu, the blocks of A, Sigma and every prediction
// run on the device (agp_sparse_*).
// ---------------------------------------------------------------------------
namespace details {
constexpr double DEFAULT_NUGGET = 1e-8;  // :22
inline std::string measurement_nugget_name() { return "measurement_nugget"; }
inline std::string inducing_nugget_name() { return "inducing_nugget"; }
}  // namespace details

struct UniformlySpacedInducingPoints {  // :36-49
  explicit UniformlySpacedInducingPoints(std::size_t num_points_ = 10) : num_points(num_points_) {}
  template <typename CovarianceFunction>
  std::vector<double> operator()(const CovarianceFunction &, const std::vector<double> &features) const {
    double lo = features.at(0), hi = features.at(0);
    for (double f : features) { lo = f < lo ? f : lo; hi = f > hi ? f : hi; }
    std::vector<double> out(num_points);
    for (std::size_t i = 0; i < num_points; ++i)  // linspace(min, max, num_points)
      out[i] = num_points > 1 ? lo + (hi - lo) * static_cast<double>(i) / static_cast<double>(num_points - 1) : lo;
    return out;
  }
  std::size_t num_points;
};

// Fit<SparseGPFit<InducingFeature>> (:92-124)
template <typename InducingFeature>
struct SparseGPFit {
  std::vector<InducingFeature> train_features;  // the inducing points
  Vector information;
  double negative_log_likelihood = 0.;
  std::shared_ptr<agp_sparse_fit> handle;
  std::shared_ptr<detail::ContextHolder> context;
};

template <typename ModelType, typename InducingFeature>
class SparseFitModel;

template <typename ModelType, typename InducingFeature, typename PredictFeature>
class SparsePrediction {
 public:
  SparsePrediction(const SparseFitModel<ModelType, InducingFeature> *fm, std::vector<PredictFeature> features)
      : fm_(fm), features_(std::move(features)) {}
  Vector mean() const { return fm_->predict_(features_, 0).mean; }
  MarginalDistribution marginal() const { return fm_->predict_(features_, 1).marginal(); }
  JointDistribution joint() const { return fm_->predict_(features_, 2); }

 private:
  const SparseFitModel<ModelType, InducingFeature> *fm_;
  std::vector<PredictFeature> features_;
};

template <typename ModelType, typename InducingFeature>
class SparseFitModel {
 public:
  SparseFitModel(const ModelType &model, SparseGPFit<InducingFeature> fit) : model_(model), fit_(std::move(fit)) {}
  const SparseGPFit<InducingFeature> &get_fit() const { return fit_; }
  const ModelType &get_model() const { return model_; }
  // Fit<SparseGPFit>::numerical_rank (:98): the rank of the pivoted QR where the fit is in that form, else the number of
  // inducing points
  std::int64_t numerical_rank() const { return agp_sparse_fit_numerical_rank(fit_.handle.get()); }

  template <typename P>
  SparsePrediction<ModelType, InducingFeature, P> predict(const std::vector<P> &features) const {
    return SparsePrediction<ModelType, InducingFeature, P>(this, features);
  }
  template <typename P>
  SparsePrediction<ModelType, InducingFeature, Measurement<P>> predict_with_measurement_noise(
      const std::vector<P> &features) const {
    return SparsePrediction<ModelType, InducingFeature, Measurement<P>>(this, as_measurements(features));
  }

  // FitModel::update -> _update_impl (:322-371): fold further observations into the fit (the inducing points stay)
  template <typename FeatureType>
  SparseFitModel update(const RegressionDataset<FeatureType> &dataset) const {
    const auto grouped = model_.group(dataset);
    detail::KernelHolder k(model_.get_covariance().program());
    detail::Flat fx = detail::flatten(model_.get_covariance(), grouped.features);
    SparseGPFit<InducingFeature> fit;
    fit.train_features = fit_.train_features;
    fit.context = fit_.context;
    fit.information.resize(fit_.information.size());
    fit.negative_log_likelihood = std::nan("");
    agp_context *c = fit.context->ctx;
    agp_sparse_fit *h = nullptr;
    detail::check(agp_sparse_fit_update(c, k.k, fit_.handle.get(), &fx.view, static_cast<std::int64_t>(grouped.offsets.size() - 1),
                                        grouped.offsets.data(), grouped.y.data(), grouped.yv.empty() ? nullptr : grouped.yv.data(),
                                        model_.measurement_nugget(), &h, fit.information.data()),
                  c, "agp_sparse_fit_update");
    auto ctx = fit.context;
    fit.handle = std::shared_ptr<agp_sparse_fit>(h, [ctx](agp_sparse_fit *p) { agp_sparse_fit_destroy(p); });
    return SparseFitModel(model_, std::move(fit));
  }

  // _predict_impl x 3 (:447-521); mode 0: mean, 1: marginal (diagonal filled), 2: joint
  template <typename P>
  JointDistribution predict_(const std::vector<P> &xs, int mode) const {
    detail::KernelHolder k(model_.get_covariance().program());
    detail::Flat f = detail::flatten(model_.get_covariance(), xs);
    JointDistribution out;
    out.mean.resize(xs.size());
    agp_context *c = fit_.context->ctx;
    if (xs.empty()) return out;
    if (mode == 0) {
      detail::check(agp_sparse_predict_mean(c, k.k, fit_.handle.get(), &f.view, out.mean.data(), AGP_HOST), c,
                    "agp_sparse_predict_mean");
    } else if (mode == 1) {
      Vector var(xs.size());
      detail::check(agp_sparse_predict_marginal(c, k.k, fit_.handle.get(), &f.view, out.mean.data(), var.data(), AGP_HOST), c,
                    "agp_sparse_predict_marginal");
      out.covariance = Matrix(static_cast<std::int64_t>(xs.size()), static_cast<std::int64_t>(xs.size()));
      for (std::size_t i = 0; i < xs.size(); ++i) out.covariance(static_cast<std::int64_t>(i), static_cast<std::int64_t>(i)) = var[i];
    } else {
      out.covariance = Matrix(static_cast<std::int64_t>(xs.size()), static_cast<std::int64_t>(xs.size()));
      detail::check(agp_sparse_predict_joint(c, k.k, fit_.handle.get(), &f.view, out.mean.data(), out.covariance.data.data(),
                                             AGP_HOST),
                    c, "agp_sparse_predict_joint");
    }
    model_.add_mean(xs, &out.mean);  // mean_function_.add_to (:457,473,516)
    return out;
  }

 private:
  ModelType model_;
  SparseGPFit<InducingFeature> fit_;
};

template <typename CovFunc, typename MeanFunc, typename GrouperFunction, typename InducingPointStrategy>
class SparseGaussianProcessRegression {
 public:
  SparseGaussianProcessRegression() = default;
  SparseGaussianProcessRegression(const CovFunc &cov, const MeanFunc &mean, const GrouperFunction &grouper,
                                  const InducingPointStrategy &strategy, const std::string &name)
      : covariance_function_(cov), mean_function_(mean), independent_group_function_(grouper),
        inducing_point_strategy_(strategy), model_name_(name) {}

  std::string get_name() const { return model_name_; }
  const CovFunc &get_covariance() const { return covariance_function_; }
  InducingPointStrategy get_inducing_point_strategy() const { return inducing_point_strategy_; }
  GrouperFunction get_grouper_function() const { return independent_group_function_; }

  ParameterStore get_params() const {  // :300-306
    ParameterStore p = mean_function_.get_params();
    for (const auto &kv : covariance_function_.get_params()) p[kv.first] = kv.second;
    p[details::measurement_nugget_name()] = measurement_nugget_;
    p[details::inducing_nugget_name()] = inducing_nugget_;
    return p;
  }
  void set_param(const std::string &n, double v) {  // :308-320
    if (n == details::measurement_nugget_name()) measurement_nugget_ = v;
    else if (n == details::inducing_nugget_name()) inducing_nugget_ = v;
    else if (covariance_function_.has_param(n)) covariance_function_.set_param(n, v);
    else if (mean_function_.has_param(n)) mean_function_.set_param(n, v);
    else throw std::out_of_range("unknown parameter " + n);
  }
  void set_param_value(const std::string &n, double v) { set_param(n, v); }

  template <typename P>
  void add_mean(const std::vector<P> &xs, Vector *mean) const {
    if (std::is_same<MeanFunc, ZeroMean>::value) return;
    for (std::size_t i = 0; i < xs.size(); ++i) (*mean)[i] += detail::mean_at(mean_function_, xs[i]);
  }

  // _fit_impl (:354-381)
  template <typename FeatureType>
  auto fit(const RegressionDataset<FeatureType> &dataset) const {
    using U = typename std::decay<decltype(inducing_point_strategy_(covariance_function_, dataset.features)[0])>::type;
    SparseGPFit<U> fit;
    run(dataset, &fit, true);
    return SparseFitModel<SparseGaussianProcessRegression, U>(*this, std::move(fit));
  }

  // The same fit with the observations split BY GROUP over the ranks of a communicator: `dataset` holds THIS rank's
  // groups (whole groups per rank; the inducing point strategy must return the same points on every rank - e.g. fixed
  // ones); every rank receives the same fit.  Collective.
  template <typename FeatureType>
  auto fit(const RegressionDataset<FeatureType> &dataset, const Communicator &comm) const {
    using U = typename std::decay<decltype(inducing_point_strategy_(covariance_function_, dataset.features)[0])>::type;
    SparseGPFit<U> fit;
    run(dataset, &fit, true, &comm);
    return SparseFitModel<SparseGaussianProcessRegression, U>(*this, std::move(fit));
  }

  // :524-596 (prior_log_likelihood() is outside the hot path and not included)
  template <typename FeatureType>
  double log_likelihood(const RegressionDataset<FeatureType> &dataset) const {
    using U = typename std::decay<decltype(inducing_point_strategy_(covariance_function_, dataset.features)[0])>::type;
    SparseGPFit<U> fit;
    run(dataset, &fit, false);
    return -fit.negative_log_likelihood;
  }

  // fit_from_prediction (:406-461): the fit on `new_inducing_points` that reproduces `prediction`, a joint distribution
  // made AT those points.  Like the reference, the mean is used as given (the mean function is not removed from it).
  template <typename FeatureType>
  auto fit_from_prediction(const std::vector<FeatureType> &new_inducing_points, const JointDistribution &prediction) const {
    const std::size_t m = new_inducing_points.size();
    if (m == 0 || prediction.mean.size() != m || prediction.covariance.rows() != static_cast<std::int64_t>(m) ||
        prediction.covariance.cols() != static_cast<std::int64_t>(m))
      throw std::invalid_argument("the prediction must be a joint distribution over the new inducing points");
    SparseGPFit<FeatureType> fit;
    fit.train_features = new_inducing_points;
    fit.context = detail::default_context();
    fit.information.resize(m);
    fit.negative_log_likelihood = std::nan("");
    agp_context *c = fit.context->ctx;
    detail::KernelHolder k(covariance_function_.program());
    detail::Flat fz = detail::flatten(covariance_function_, new_inducing_points);
    agp_sparse_fit *h = nullptr;
    detail::check(agp_sparse_fit_from_prediction(c, k.k, &fz.view, prediction.mean.data(), prediction.covariance.data.data(),
                                                 static_cast<std::int64_t>(m), AGP_HOST, inducing_nugget_, &h,
                                                 fit.information.data(), nullptr),
                  c, "agp_sparse_fit_from_prediction");
    auto ctx = fit.context;
    fit.handle = std::shared_ptr<agp_sparse_fit>(h, [ctx](agp_sparse_fit *p) { agp_sparse_fit_destroy(p); });
    return SparseFitModel<SparseGaussianProcessRegression, FeatureType>(*this, std::move(fit));
  }

  double measurement_nugget() const { return measurement_nugget_; }

  // the host half of compute_internal_components (:642-668): group_by(features, grouper).indexers() in key
  // order, reordered_inds, subsets of features / targets
  template <typename FeatureType>
  struct Grouped {
    std::vector<FeatureType> features;
    std::vector<std::int64_t> offsets;
    Vector y, yv;
  };
  template <typename FeatureType>
  Grouped<FeatureType> group(const RegressionDataset<FeatureType> &dataset) const {
    const std::size_t n = dataset.features.size();
    if (n != dataset.targets.size()) throw std::invalid_argument("features and targets differ in size");
    using Key = typename std::decay<decltype(independent_group_function_(dataset.features[0]))>::type;
    std::map<Key, std::vector<std::size_t>> indexer;
    for (std::size_t i = 0; i < n; ++i) indexer[independent_group_function_(dataset.features[i])].push_back(i);
    std::vector<std::size_t> reordered_inds;
    Grouped<FeatureType> g;
    g.offsets.assign(1, 0);
    for (const auto &kv : indexer) {
      reordered_inds.insert(reordered_inds.end(), kv.second.begin(), kv.second.end());
      g.offsets.push_back(static_cast<std::int64_t>(reordered_inds.size()));
    }
    g.features.resize(n);
    g.y.resize(n);
    const bool has_var = !dataset.targets.covariance.empty();
    if (has_var) g.yv.resize(n);
    for (std::size_t a = 0; a < n; ++a) {
      g.features[a] = dataset.features[reordered_inds[a]];
      g.y[a] = dataset.targets.mean[reordered_inds[a]];  // y is copied BEFORE the mean function is removed (:664-668)
      if (has_var) g.yv[a] = dataset.targets.covariance[reordered_inds[a]];
    }
    return g;
  }

 private:
  template <typename FeatureType, typename U>
  void run(const RegressionDataset<FeatureType> &dataset, SparseGPFit<U> *fit, bool keep, const Communicator *comm = nullptr) const {
    const Grouped<FeatureType> grouped = group(dataset);
    const std::vector<FeatureType> &features = grouped.features;
    const std::vector<std::int64_t> &offsets = grouped.offsets;
    const Vector &y = grouped.y, &yv = grouped.yv;
    const bool has_var = !yv.empty();
    fit->train_features = inducing_point_strategy_(covariance_function_, dataset.features);
    if (fit->train_features.empty()) throw std::invalid_argument("Empty inducing points!");  // :361
    fit->context = detail::default_context();
    agp_context *c = fit->context->ctx;
    detail::KernelHolder k(covariance_function_.program());
    detail::Flat fx = detail::flatten(covariance_function_, features);
    detail::Flat fu = detail::flatten(covariance_function_, fit->train_features);
    fit->information.resize(fit->train_features.size());
    agp_sparse_fit *h = nullptr;
    if (comm)  // this rank's groups of one fit spread over all ranks (agp_sparse_fit_create_sharded; collective)
      detail::check(agp_sparse_fit_create_sharded(c, comm->handle(), k.k, &fx.view, static_cast<std::int64_t>(offsets.size() - 1),
                                                  offsets.data(), y.data(), has_var ? yv.data() : nullptr, &fu.view,
                                                  measurement_nugget_, inducing_nugget_, keep ? &h : nullptr,
                                                  fit->information.data(), &fit->negative_log_likelihood),
                    c, "agp_sparse_fit_create_sharded");
    else
      detail::check(agp_sparse_fit_create(c, k.k, &fx.view, static_cast<std::int64_t>(offsets.size() - 1), offsets.data(), y.data(),
                                          has_var ? yv.data() : nullptr, &fu.view, measurement_nugget_, inducing_nugget_,
                                          keep ? &h : nullptr, fit->information.data(), &fit->negative_log_likelihood),
                    c, "agp_sparse_fit_create");
    auto ctx = fit->context;
    if (keep) fit->handle = std::shared_ptr<agp_sparse_fit>(h, [ctx](agp_sparse_fit *p) { agp_sparse_fit_destroy(p); });
  }

  CovFunc covariance_function_;
  MeanFunc mean_function_;
  GrouperFunction independent_group_function_;
  InducingPointStrategy inducing_point_strategy_;
  std::string model_name_ = "sparse_gaussian_process_regression";
  double measurement_nugget_ = details::DEFAULT_NUGGET;  // initialize_params, :292-298
  double inducing_nugget_ = details::DEFAULT_NUGGET;
};

// factories, :740-775
template <typename CovFunc, typename MeanFunc, typename GrouperFunction, typename InducingPointStrategy>
auto sparse_gp_from_covariance_and_mean(const CovFunc &cov, const MeanFunc &mean, const GrouperFunction &grouper,
                                        const InducingPointStrategy &strategy, const std::string &model_name) {
  return SparseGaussianProcessRegression<CovFunc, MeanFunc, GrouperFunction, InducingPointStrategy>(cov, mean, grouper, strategy,
                                                                                                   model_name);
}

template <typename CovFunc, typename GrouperFunction, typename InducingPointStrategy>
auto sparse_gp_from_covariance(const CovFunc &cov, const GrouperFunction &grouper, const InducingPointStrategy &strategy,
                               const std::string &model_name) {
  return sparse_gp_from_covariance_and_mean(cov, ZeroMean(), grouper, strategy, model_name);
}

// rebase_inducing_points (:714-725): a fit relative to new inducing points, from the old fit's joint prediction at them.
// NOT equivalent to fitting with the new inducing points: information may be lost.
template <typename ModelType, typename InducingFeature, typename NewFeatureType>
auto rebase_inducing_points(const SparseFitModel<ModelType, InducingFeature> &fit_model,
                            const std::vector<NewFeatureType> &new_inducing_points) {
  return fit_model.get_model().fit_from_prediction(new_inducing_points, fit_model.predict(new_inducing_points).joint());
}

// GaussianProcessNegativeLogLikelihood, gp.hpp:542-550: the tuner's objective
struct GaussianProcessNegativeLogLikelihood {
  template <typename FeatureType, typename CovFunc, typename MeanFunc>
  double operator()(const RegressionDataset<FeatureType> &dataset,
                    const GaussianProcessRegression<CovFunc, MeanFunc> &model) const {
    return -model.log_likelihood(dataset);
  }
};

}  // namespace albatross

#endif  // ALBATROSS_AMD_ALBATROSS_HPP
