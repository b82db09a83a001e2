"""Host-side mirror of albatross's composable covariance functions.

Each class keeps the reference's name, parameter names and call semantics
(include/albatross/src/covariance_functions/*.hpp) and flattens itself into the
postfix `agp_kernel_node` program the HIP library evaluates per pair.  No
arithmetic on the Gram happens here: `cov(xs)`, `cov(xs, ys)` go through the
C-ABI (`agp_gram`), which has no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from ._capi import KernelNode


# ---------------------------------------------------------------------------
# distance metrics (distance_metrics.hpp:30-90) — tags only; math is on device
# ---------------------------------------------------------------------------
class EuclideanDistance:
    metric = capi.METRIC_EUCLIDEAN

    def get_name(self):
        return "euclidean_distance"


class RadialDistance:
    metric = capi.METRIC_RADIAL

    def get_name(self):
        return "radial_distance"


class AngularDistance:
    metric = capi.METRIC_ANGULAR

    def get_name(self):
        return "angular_distance"


# ---------------------------------------------------------------------------
# flattened feature vector handed to the C-ABI
# ---------------------------------------------------------------------------
class FeatureSet:
    """POD view of `std::vector<Feature>`: row-major coords, optional equality
    ids and per-point scale columns (ScalingTerm values f(x_i))."""

    def __init__(self, coords, scales=None, eq_id=None, is_measurement=False):
        a = np.asarray(coords, dtype=np.float64)
        if a.ndim == 1:
            a = a.reshape(-1, 1)
        if a.ndim != 2 or not (1 <= a.shape[1] <= capi.MAX_DIM):
            raise ValueError(f"features must be n x dim with 1 <= dim <= {capi.MAX_DIM}")
        self.coords = np.ascontiguousarray(a)
        self.n, self.dim = self.coords.shape
        self.scales = None
        if scales is not None and len(scales) > 0:
            s = np.asfortranarray(np.column_stack([np.asarray(c, dtype=np.float64) for c in scales]))
            if s.shape != (self.n, len(scales)):
                raise ValueError("scale columns must have one value per feature")
            self.scales = s
        self.eq_id = None if eq_id is None else np.ascontiguousarray(eq_id, dtype=np.int64)
        self.is_measurement = bool(is_measurement)

    @property
    def n_scale_columns(self):
        return 0 if self.scales is None else self.scales.shape[1]

    def as_struct(self):
        f = capi.Features()
        f.n = self.n
        f.dim = self.dim
        f.n_scale_columns = self.n_scale_columns
        f.coords = self.coords.ctypes.data
        f.eq_id = None if self.eq_id is None else self.eq_id.ctypes.data
        f.scales = None if self.scales is None else self.scales.ctypes.data
        f.is_measurement = 1 if self.is_measurement else 0
        f.location = capi.HOST
        return f


class Measurement:
    """Measurement<X> tag for a whole feature vector (measurement.hpp:18-53)."""

    def __init__(self, values):
        self.values = values


class _AlternativeIndex:
    """Pseudo scaling function: the scale column that carries the alternative index of every point of a
    variant<> feature vector (read by AGP_OP_TYPE_PAIR).  Plain feature vectors are alternative 0."""

    def __call__(self, coords):
        return np.zeros(len(coords))


_ALTERNATIVE_INDEX = _AlternativeIndex()


class VariantFeatures:
    """std::vector<variant<T0, T1, ...>>: element i holds alternative `alternatives[i]` with value `values[i]`
    (a scalar or a coordinate vector; alternatives may have different dimensions, the POD record is zero-padded
    to the largest).  See only_for_alternatives()."""

    def __init__(self, alternatives, values):
        self.alternatives = np.asarray(alternatives, dtype=np.int64).reshape(-1)
        vals = [np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in values]
        if len(vals) != self.alternatives.shape[0]:
            raise ValueError("one alternative index per value")
        dim = max([v.shape[0] for v in vals] + [1])
        self.coords = np.zeros((len(vals), dim))
        for i, v in enumerate(vals):
            self.coords[i, :v.shape[0]] = v

    def __len__(self):
        return self.alternatives.shape[0]


class LinearCombination:
    """LinearCombination<X> (core/linear_combination.hpp:18-44): a feature that is sum_i coefficients[i] * values[i].
    cov(a, b) = sum_ij a_i b_j cov(x_i, y_j) and mean(a) = sum_i a_i mean(x_i), applied at the TOP of the caller chain
    (LinearCombinationCaller, covariance_functions/callers.hpp:321-396).  Here: the Gram matrix of the expanded
    points is built on the device and contracted with the coefficients on the host."""

    def __init__(self, values, coefficients=None):
        self.values = [np.atleast_1d(np.asarray(v, dtype=np.float64)) for v in values]
        self.coefficients = (np.ones(len(self.values)) if coefficients is None
                             else np.asarray(coefficients, dtype=np.float64).reshape(-1))
        if len(self.values) != self.coefficients.shape[0]:
            raise ValueError("values and coefficients differ in size")  # linear_combination.hpp:33

    def __eq__(self, other):  # linear_combination.hpp:36-38
        return (isinstance(other, LinearCombination) and len(self.values) == len(other.values)
                and all(np.array_equal(a, b) for a, b in zip(self.values, other.values))
                and np.array_equal(self.coefficients, other.coefficients))


def has_linear_combinations(features):
    values = features.values if isinstance(features, Measurement) else features
    return isinstance(values, (list, tuple)) and any(isinstance(f, LinearCombination) for f in values)


def expand_linear_combinations(features):
    """(expanded features, C): C is the (number of expanded points) x (number of features) coefficient matrix; a
    plain feature is the combination of itself with coefficient 1.  A Measurement<> wrapper stays on the expanded
    vector (MeasurementForwarder sits outside LinearCombinationCaller in the DefaultCaller chain, callers.hpp:546-553)."""
    meas = isinstance(features, Measurement)
    values = features.values if meas else features
    points, rows, cols, coefs = [], [], [], []
    for j, f in enumerate(values):
        if isinstance(f, LinearCombination):
            for v, a in zip(f.values, f.coefficients):
                rows.append(len(points)); cols.append(j); coefs.append(a); points.append(v)
        else:
            rows.append(len(points)); cols.append(j); coefs.append(1.); points.append(np.atleast_1d(np.asarray(f, dtype=np.float64)))
    C = np.zeros((len(points), len(values)))
    C[rows, cols] = coefs
    pts = np.stack(points) if points else np.zeros((0, 1))
    return (Measurement(pts) if meas else pts), C


def expand_with_offsets(features):
    """(expanded features, offsets, coefficients) for agp_gram_combined: combination j = expanded points
    offsets[j] .. offsets[j + 1] with the given coefficients (a plain feature is the combination of itself)."""
    ex, Cm = expand_linear_combinations(features)
    rows, cols = np.nonzero(Cm != 0.) if Cm.size else (np.zeros(0, int), np.zeros(0, int))
    n_exp, n = Cm.shape
    # the expansion appends the members of combination 0, then of 1, ...: owners are non-decreasing; zero coefficients
    # still occupy their slot, so offsets come from the construction order, not from the non-zeros
    owner = np.zeros(n_exp, dtype=np.int64)
    values = features.values if isinstance(features, Measurement) else features
    pos = 0
    for j, f in enumerate(values):
        cnt = len(f.values) if isinstance(f, LinearCombination) else 1
        owner[pos:pos + cnt] = j
        pos += cnt
    offsets = np.zeros(n + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(np.bincount(owner, minlength=n))
    coefficients = np.ascontiguousarray(Cm[np.arange(n_exp), owner], dtype=np.float64)
    return ex, offsets, coefficients


def as_measurements(features):
    return features if isinstance(features, Measurement) else Measurement(features)


# ---------------------------------------------------------------------------
# CovarianceFunction base (covariance_function.hpp:63-217)
# ---------------------------------------------------------------------------
class CovarianceFunction:
    def __add__(self, other):
        return SumOfCovarianceFunctions(self, other)

    def __mul__(self, other):
        return ProductOfCovarianceFunctions(self, other)

    # --- parameters (ParameterHandlingMixin subset) -------------------------
    def get_params(self):
        return dict(self._params)

    def get_param_value(self, name):
        return self.get_params()[name]

    def set_param(self, name, value):
        if name not in self._params:
            raise KeyError(name)
        self._params[name] = float(value)

    def set_param_values(self, values):
        for k, v in values.items():
            self.set_param(k, v)

    set_params = set_param_values

    def get_name(self):
        return self.name()

    # --- flattening -----------------------------------------------------------
    def _emit(self, nodes, scalers):
        raise NotImplementedError

    def program(self):
        nodes, scalers = [], []
        self._emit(nodes, scalers)
        if len(nodes) > capi.MAX_KERNEL_NODES:
            raise ValueError("covariance function has too many terms for the device program")
        if len(scalers) > capi.MAX_SCALE_COLUMNS:
            raise ValueError("too many ScalingTerms")
        depth = peak = 0
        for nd in nodes:
            if nd.op in (capi.OP_SUM, capi.OP_PRODUCT):
                depth -= 1
            elif nd.op != capi.OP_MEASUREMENT_ONLY:
                depth += 1
            peak = max(peak, depth)
        if peak > capi.MAX_STACK:
            raise ValueError("covariance function nests too deeply for the device program")
        return nodes, scalers

    def program_nodes(self):
        return self.program()[0]

    def features(self, x, is_measurement=False):
        """Flatten a feature vector for this covariance function: evaluates
        every ScalingTerm's f(x_i) once per point (scaling_function.hpp:79-83)."""
        if isinstance(x, Measurement):
            return self.features(x.values, is_measurement=True)
        if isinstance(x, FeatureSet):
            if x.is_measurement == bool(is_measurement):
                return x
            return FeatureSet(x.coords, None if x.scales is None else list(x.scales.T), x.eq_id, is_measurement)
        _, scalers = self.program()
        alternatives = None
        if isinstance(x, VariantFeatures):
            alternatives, x = x.alternatives.astype(np.float64), x.coords
        coords = np.asarray(x, dtype=np.float64)
        if coords.ndim == 1:
            coords = coords.reshape(-1, 1)  # scaling functions always see n x dim
        cols = [(alternatives if (s is _ALTERNATIVE_INDEX and alternatives is not None)
                 else np.asarray(s(coords), dtype=np.float64).reshape(-1)) for s in scalers]
        eq_id = None
        if alternatives is not None:
            # variant equality = same alternative AND equal value (the zero-padded coordinates alone would make
            # a 1-D alternative at t equal to a 3-D one at (t, 0, 0)): a 63-bit hash of both, identical for equal
            # features of any two feature vectors
            eq_id = np.array([hash((int(a), row.tobytes())) & 0x7fffffffffffffff
                              for a, row in zip(alternatives, np.ascontiguousarray(coords))], dtype=np.int64)
        return FeatureSet(coords, cols, eq_id, is_measurement)

    # --- calls ------------------------------------------------------------------
    def __call__(self, xs, ys=None):
        """cov(xs) / cov(xs, ys): compute_covariance_matrix (callers.hpp:38-166)."""
        from .gp import default_context
        return default_context().gram(self, xs, ys)

    def diagonal(self, xs):
        from .gp import default_context
        return default_context().gram_diagonal(self, xs)


def _node(op, metric=0, column=0, order=0, params=()):
    nd = KernelNode()
    nd.op = op
    nd.metric = metric
    nd.column = column
    nd.order = order
    for i, p in enumerate(params):
        nd.params[i] = float(p)
    return nd


default_length_scale = 100000.0  # radial.hpp:16
default_radial_sigma = 10.0      # radial.hpp:17


class _Radial(CovarianceFunction):
    _op = None
    _ls = _sg = _nm = None

    def __init__(self, length_scale=default_length_scale, sigma=default_radial_sigma,
                 distance_metric=None):
        self.distance_metric_ = distance_metric or EuclideanDistance()
        if isinstance(self.distance_metric_, type):
            self.distance_metric_ = self.distance_metric_()
        self._params = {self._ls: float(length_scale), self._sg: float(sigma)}

    def name(self):
        return f"{self._nm}[{self.distance_metric_.get_name()}]"

    def _emit(self, nodes, scalers):
        nodes.append(_node(self._op, metric=self.distance_metric_.metric,
                           params=(self._params[self._ls], self._params[self._sg])))


class SquaredExponential(_Radial):
    """sigma^2 exp(-(d/l)^2)  (radial.hpp:131-189)."""
    _op = capi.OP_SQUARED_EXPONENTIAL
    _ls, _sg, _nm = "squared_exponential_length_scale", "sigma_squared_exponential", "squared_exponential"

    def __init__(self, length_scale=default_length_scale, sigma=default_radial_sigma, distance_metric=None):
        super().__init__(length_scale, sigma, distance_metric)
        if isinstance(self.distance_metric_, AngularDistance):
            # static_assert in radial.hpp:138-141
            raise TypeError("SquaredExponential covariance with AngularDistance is not PSD.")


class Exponential(_Radial):
    """sigma^2 exp(-|d|/l)  (radial.hpp:239-287)."""
    _op = capi.OP_EXPONENTIAL
    _ls, _sg, _nm = "exponential_length_scale", "sigma_exponential", "exponential"


class Matern32(_Radial):
    """radial.hpp:421-459"""
    _op = capi.OP_MATERN32
    _ls, _sg, _nm = "matern_32_length_scale", "sigma_matern_32", "matern_32"


class Matern52(_Radial):
    """radial.hpp:491-529"""
    _op = capi.OP_MATERN52
    _ls, _sg, _nm = "matern_52_length_scale", "sigma_matern_52", "matern_52"


class Constant(CovarianceFunction):
    """polynomials.hpp:31-61"""

    def __init__(self, sigma_constant=10.0):
        self._params = {"sigma_constant": float(sigma_constant)}

    def name(self):
        return "constant"

    def _emit(self, nodes, scalers):
        nodes.append(_node(capi.OP_CONSTANT, params=(self._params["sigma_constant"],)))


class Polynomial(CovarianceFunction):
    """Polynomial<order> on 1-D features (polynomials.hpp:63-90)."""

    def __init__(self, order, sigma=10.0):
        if not 0 <= order <= 3:
            raise ValueError("device path supports Polynomial<order> for order <= 3")
        self.order = order
        self._params = {f"sigma_polynomial_{i}": float(sigma) for i in range(order + 1)}

    def name(self):
        return f"polynomial_{self.order}"

    def _emit(self, nodes, scalers):
        nodes.append(_node(capi.OP_POLYNOMIAL, order=self.order,
                           params=[self._params[f"sigma_polynomial_{i}"] for i in range(self.order + 1)]))


class IndependentNoise(CovarianceFunction):
    """sigma^2 iff x == y  (noise.hpp:20-44)."""

    def __init__(self, sigma_noise=0.1):
        self._params = {"sigma_independent_noise": float(sigma_noise)}

    def name(self):
        return "independent_noise"

    def _emit(self, nodes, scalers):
        nodes.append(_node(capi.OP_INDEPENDENT_NOISE, params=(self._params["sigma_independent_noise"],)))


class Nugget(CovarianceFunction):
    """nugget.hpp:32-49 (default_nugget_noise = 1e-8)"""

    def __init__(self, nugget_sigma=1e-8):
        self._params = {"nugget_sigma": float(nugget_sigma)}

    def name(self):
        return "nugget"

    def _emit(self, nodes, scalers):
        nodes.append(_node(capi.OP_NUGGET, params=(self._params["nugget_sigma"],)))


class ScalingFunction:
    """Base of a deterministic scaling f(x) (scaling_function.hpp:18-33).
    Subclasses define `_call_impl(coords) -> n values` (vectorised over the
    feature vector) and optional parameters in `self._params`."""
    _params = {}

    def get_name(self):
        return type(self).__name__

    def get_params(self):
        return dict(self._params)

    def set_param(self, name, value):
        if "_params" not in self.__dict__:  # class-level defaults: copy on first write
            self._params = dict(type(self)._params)
        self._params[name] = float(value)

    def __call__(self, coords):
        return self._call_impl(coords)


class ScalingTerm(CovarianceFunction):
    """cov(x, y) = f(x) f(y)  (scaling_function.hpp:58-112)."""

    def __init__(self, scaling_function):
        self.scaling_function_ = scaling_function

    @property
    def _params(self):
        return self.scaling_function_._params

    def get_params(self):
        return self.scaling_function_.get_params()

    def set_param(self, name, value):
        if name not in self.scaling_function_._params:
            raise KeyError(name)
        self.scaling_function_.set_param(name, value)

    def name(self):
        return self.scaling_function_.get_name()

    def _emit(self, nodes, scalers):
        nodes.append(_node(capi.OP_SCALING, column=len(scalers)))
        scalers.append(self.scaling_function_)


class _Binary(CovarianceFunction):
    _op = None
    _sym = "?"

    def __init__(self, lhs, rhs):
        self.lhs_, self.rhs_ = lhs, rhs

    def name(self):
        return f"({self.lhs_.get_name()}{self._sym}{self.rhs_.get_name()})"

    def get_params(self):  # map_join, covariance_function.hpp:235-237
        out = dict(self.lhs_.get_params())
        out.update(self.rhs_.get_params())
        return out

    def set_param(self, name, value):  # set_param_if_exists_in_any, :239-242
        done = False
        for side in (self.lhs_, self.rhs_):
            if name in side.get_params():
                side.set_param(name, value)
                done = True
        if not done:
            raise KeyError(name)

    def _emit(self, nodes, scalers):
        self.lhs_._emit(nodes, scalers)
        self.rhs_._emit(nodes, scalers)
        nodes.append(_node(self._op))


class SumOfCovarianceFunctions(_Binary):
    """covariance_function.hpp:222-325"""
    _op = capi.OP_SUM
    _sym = "+"


class ProductOfCovarianceFunctions(_Binary):
    """covariance_function.hpp:330-420 (rhs skipped when lhs == 0, :362-366)"""
    _op = capi.OP_PRODUCT
    _sym = "*"


class MeasurementOnly(CovarianceFunction):
    """measurement.hpp:70-106"""

    def __init__(self, sub_cov):
        self.sub_cov_ = sub_cov

    def name(self):
        return f"measurement[{self.sub_cov_.get_name()}]"

    def get_params(self):
        return self.sub_cov_.get_params()

    def set_param(self, name, value):
        self.sub_cov_.set_param(name, value)

    def _emit(self, nodes, scalers):
        self.sub_cov_._emit(nodes, scalers)
        nodes.append(_node(capi.OP_MEASUREMENT_ONLY))


def measurement_only(cov):
    return MeasurementOnly(cov)


class OnlyForAlternatives(CovarianceFunction):
    """A covariance function that is defined for ONE pair of alternatives (a, b) of a variant<> feature type (in
    either order): what a `_call_impl(const A &, const B &)` overload is in the reference.  VariantForwarder
    (covariance_functions/callers.hpp:419-544) returns 0 for every pair of alternatives without an overload; a
    covariance function with several overloads (tests/lib/albatross/test/test_covariance_utils.h:42-62) is the sum of
    one of these per overload."""

    def __init__(self, sub_cov, a, b=None):
        self.sub_cov_ = sub_cov
        self.a_, self.b_ = int(a), int(a if b is None else b)

    def name(self):
        return f"alternatives[{self.a_},{self.b_}][{self.sub_cov_.get_name()}]"

    def get_params(self):
        return self.sub_cov_.get_params()

    def set_param(self, name, value):
        self.sub_cov_.set_param(name, value)

    def _emit(self, nodes, scalers):
        self.sub_cov_._emit(nodes, scalers)
        if _ALTERNATIVE_INDEX not in scalers:
            scalers.append(_ALTERNATIVE_INDEX)  # one shared column for every gate of the program
        nodes.append(_node(capi.OP_TYPE_PAIR, column=scalers.index(_ALTERNATIVE_INDEX), params=(self.a_, self.b_)))


def only_for_alternatives(cov, a, b=None):
    return OnlyForAlternatives(cov, a, b)


def nodes_to_array(nodes):
    return (KernelNode * len(nodes))(*nodes)
