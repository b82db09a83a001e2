"""Host-side mirror of albatross's SparseGaussianProcessRegression
(include/albatross/src/models/sparse_gp.hpp) over the sparse entry points of the C-ABI.

    model = sparse_gp_from_covariance(cov, grouper, UniformlySpacedInducingPoints(8), "sparse")
    model.set_param("inducing_nugget", 1e-3)
    fit_model = model.fit(dataset)
    fit_model.predict_with_measurement_noise(xs).joint()

Grouping, reordering and the choice of inducing points are O(n) host bookkeeping exactly as in
compute_internal_components (:631-706); K_uu, K_fu, the blocks of A, Sigma and every prediction
are computed on the device.  There is no CPU fallback.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from .covariance import CovarianceFunction, Measurement
from .gp import (JointDistribution, MarginalDistribution, Prediction, RegressionDataset, ZeroMean, _ptr, _values_of,
                 default_context)

DEFAULT_NUGGET = 1e-8  # details::DEFAULT_NUGGET (:22)


class UniformlySpacedInducingPoints:
    """UniformlySpacedInducingPoints (:36-49): linspace over the range of 1-D features."""

    def __init__(self, num_points=10):
        self.num_points = num_points

    def __call__(self, cov, features):
        f = np.asarray(features, dtype=np.float64).reshape(-1)
        return np.linspace(f.min(), f.max(), self.num_points)


class FixedInducingPoints:
    """An InducingPointStrategy that returns given features (tests/test_sparse_gp.cc:219-235)."""

    def __init__(self, points):
        self.points = points

    def __call__(self, cov, features):
        return self.points


class SparseGPFit:
    """Fit<SparseGPFit<InducingFeature>> (:92-124): inducing features, K_uu factor, Sigma factor, information."""

    def __init__(self, ctx, handle, train_features, nll):
        self._ctx, self._h = ctx, handle
        self.train_features = train_features
        self.m = int(ctx._lib.agp_sparse_fit_size(handle))
        self.nll = nll

    @property
    def numerical_rank(self):
        return int(self._ctx._lib.agp_sparse_fit_numerical_rank(self._h))

    def __del__(self):
        if getattr(self, "_h", None) and self._ctx._h:
            self._ctx._lib.agp_sparse_fit_destroy(self._h)
            self._h = None

    @property
    def information(self):
        out = np.empty(self.m)
        self._ctx._check(self._ctx._lib.agp_sparse_fit_information(self._ctx._h, self._h, _ptr(out)),
                         "agp_sparse_fit_information")
        return out


class SparseFitModel:
    """FitModel<SparseGaussianProcessRegression, Fit<SparseGPFit>>."""

    def __init__(self, model, fit):
        self._model, self._fit = model, fit

    def get_fit(self):
        return self._fit

    def get_model(self):
        return self._model

    def predict(self, features):
        return Prediction(self, features)

    def predict_with_measurement_noise(self, features):
        return Prediction(self, features if isinstance(features, Measurement) else Measurement(features))

    def update(self, dataset, targets=None):
        """FitModel::update -> _update_impl (:322-371): fold further observations into the fit; the inducing
        points stay.  Returns a new SparseFitModel."""
        if targets is not None:
            dataset = RegressionDataset(dataset, targets)
        m, ctx = self._model, self._model._ctx()
        cov = m.covariance_function_
        reordered, offsets, y, yv = m._group(dataset)
        fx = cov.features(reordered)
        sx = fx.as_struct()
        h = C.c_void_p()
        ctx._check(ctx._lib.agp_sparse_fit_update(ctx._h, ctx.kernel(cov), self._fit._h, C.byref(sx), len(offsets) - 1,
                                                  _ptr(offsets), _ptr(y), _ptr(yv), m.measurement_nugget_, C.byref(h),
                                                  None), "agp_sparse_fit_update")
        return SparseFitModel(m, SparseGPFit(ctx, h, self._fit.train_features, float("nan")))

    def _call(self, fn, features, n_out):
        m, ctx = self._model, self._model._ctx()
        fs = m.covariance_function_.features(features)
        s = fs.as_struct()
        mean = np.empty(fs.n)
        if n_out == 0:
            args = (_ptr(mean),)
            extra = None
        elif n_out == 1:
            extra = np.empty(fs.n)
            args = (_ptr(mean), _ptr(extra))
        else:
            extra = np.empty((fs.n, fs.n), order="F")
            args = (_ptr(mean), _ptr(extra))
        ctx._check(getattr(ctx._lib, fn)(ctx._h, ctx.kernel(m.covariance_function_), self._fit._h, C.byref(s), *args,
                                         capi.HOST), fn)
        return mean + m.mean_function_(fs.coords), extra  # mean_function_.add_to (:457,473,516)

    def _predict_mean(self, features):
        return self._call("agp_sparse_predict_mean", features, 0)[0]

    def _predict_marginal(self, features):
        return MarginalDistribution(*self._call("agp_sparse_predict_marginal", features, 1))

    def _predict_joint(self, features):
        return JointDistribution(*self._call("agp_sparse_predict_joint", features, 2))


class SparseGaussianProcessRegression:
    """SparseGaussianProcessRegression<CovFunc, MeanFunc, GrouperFunction, InducingPointStrategy> (:245-712)."""

    def __init__(self, covariance_function, mean_function=None, grouper_function=None, inducing_point_strategy=None,
                 model_name="sparse_gaussian_process_regression", context=None):
        if not isinstance(covariance_function, CovarianceFunction):
            raise TypeError("covariance_function must be an albatross_amd CovarianceFunction")
        if grouper_function is None or inducing_point_strategy is None:
            raise ValueError("a grouper function and an inducing point strategy are required")
        self.covariance_function_ = covariance_function
        self.mean_function_ = mean_function or ZeroMean()
        self.independent_group_function_ = grouper_function
        self.inducing_point_strategy_ = inducing_point_strategy
        self.model_name_ = model_name
        self._context = context
        self.measurement_nugget_ = DEFAULT_NUGGET  # initialize_params (:292-298)
        self.inducing_nugget_ = DEFAULT_NUGGET

    def _ctx(self):
        return self._context or default_context()

    def get_name(self):
        return self.model_name_

    def get_covariance(self):
        return self.covariance_function_

    def get_params(self):  # :300-306
        out = dict(self.mean_function_.get_params())
        out.update(self.covariance_function_.get_params())
        out["measurement_nugget"] = self.measurement_nugget_
        out["inducing_nugget"] = self.inducing_nugget_
        return out

    def set_param(self, name, value):  # :308-320
        if name == "measurement_nugget":
            self.measurement_nugget_ = float(value)
        elif name == "inducing_nugget":
            self.inducing_nugget_ = float(value)
        elif name in self.covariance_function_.get_params():
            self.covariance_function_.set_param(name, value)
        elif name in self.mean_function_.get_params():
            self.mean_function_.set_param(name, value)
        else:
            raise KeyError(name)

    set_param_value = set_param

    def set_param_values(self, values):
        for k, v in values.items():
            self.set_param(k, v)

    def _components(self, dataset):
        """The host part of compute_internal_components (:642-668) plus the inducing point strategy (:358-360)."""
        reordered, offsets, y, yv = self._group(dataset)
        u = self.inducing_point_strategy_(self.covariance_function_, _values_of(dataset.features))
        if len(u) == 0:
            raise ValueError("Empty inducing points!")  # :361
        return reordered, offsets, y, yv, u

    def _group(self, dataset):
        """group_by(features, grouper).indexers() in key order, reordered_inds, subset of features / targets."""
        feats = _values_of(dataset.features)
        n = len(feats)
        grouper = self.independent_group_function_
        arr = np.asarray(feats, dtype=np.float64)
        keys_of = None
        if getattr(grouper, "vectorized", False):
            # a grouper that maps the whole feature array to an array of keys (one call instead of n Python calls:
            # 45 -> 8 ms of host time at n = 262144); mark it with `grouper.vectorized = True`
            keys_of = np.asarray(grouper(arr))
            if keys_of.shape != (n,):
                raise ValueError("a vectorized grouper must return one key per feature")
        if keys_of is not None:
            uniq, inverse = np.unique(keys_of, return_inverse=True)  # sorted keys, like group_by's std::map
            order = np.argsort(inverse, kind="stable").astype(np.int64)
            offsets = np.zeros(len(uniq) + 1, dtype=np.int64)
            offsets[1:] = np.cumsum(np.bincount(inverse, minlength=len(uniq)))
        else:
            groups = {}
            for i in range(n):
                groups.setdefault(grouper(feats[i]), []).append(i)
            keys = sorted(groups.keys())
            order = np.concatenate([np.asarray(groups[k], dtype=np.int64) for k in keys])
            offsets = np.zeros(len(keys) + 1, dtype=np.int64)
            offsets[1:] = np.cumsum([len(groups[k]) for k in keys])
        reordered = arr[order]
        y = np.ascontiguousarray(np.asarray(dataset.targets.mean, dtype=np.float64)[order])  # y BEFORE remove_from, :664-668
        yv = None
        if dataset.targets.covariance is not None:
            yv = np.ascontiguousarray(np.asarray(dataset.targets.covariance, dtype=np.float64)[order])
        return reordered, offsets, y, yv

    def _create(self, dataset, want_fit, comm=None):
        ctx = self._ctx()
        cov = self.covariance_function_
        reordered, offsets, y, yv, u = self._components(dataset)
        fx, fu = cov.features(reordered), cov.features(u)
        sx, su = fx.as_struct(), fu.as_struct()
        h = C.c_void_p()
        nll = C.c_double()
        if comm is None:
            st = ctx._lib.agp_sparse_fit_create(ctx._h, ctx.kernel(cov), C.byref(sx), len(offsets) - 1, _ptr(offsets),
                                                _ptr(y), _ptr(yv), C.byref(su), self.measurement_nugget_,
                                                self.inducing_nugget_, C.byref(h) if want_fit else None, None,
                                                C.byref(nll))
        else:  # this rank's groups only; the m x m sums over observations are all-reduced inside the library
            st = ctx._lib.agp_sparse_fit_create_sharded(ctx._h, comm._h, ctx.kernel(cov), C.byref(sx), len(offsets) - 1,
                                                        _ptr(offsets), _ptr(y), _ptr(yv), C.byref(su), self.measurement_nugget_,
                                                        self.inducing_nugget_, C.byref(h) if want_fit else None, None,
                                                        C.byref(nll))
        ctx._check(st, "agp_sparse_fit_create")
        return (SparseGPFit(ctx, h, u, nll.value) if want_fit else None), nll.value

    def fit(self, dataset, targets=None, comm=None):
        """_fit_impl (:354-381).  comm (albatross_amd.distributed.Communicator): `dataset` holds THIS rank's groups of
        one fit spread over all ranks (whole groups per rank; the inducing point strategy must return the same points
        on every rank); every rank receives the same fit."""
        if targets is not None:
            dataset = RegressionDataset(dataset, targets)
        fit, _ = self._create(dataset, True, comm)
        return SparseFitModel(self, fit)

    def log_likelihood(self, dataset, comm=None):
        """:524-596, without the parameter priors (out of scope)."""
        return -self._create(dataset, False, comm)[1]

    def fit_from_prediction(self, new_inducing_points, prediction):
        """fit_from_prediction (:406-461): the fit on `new_inducing_points` that reproduces `prediction`, a
        JointDistribution made AT those points.  Like the reference, the mean is used as given (the mean function is not
        removed from it)."""
        ctx, cov = self._ctx(), self.covariance_function_
        fz = cov.features(new_inducing_points)
        sz = fz.as_struct()
        mean = np.ascontiguousarray(prediction.mean, dtype=np.float64)
        covariance = np.asfortranarray(prediction.covariance, dtype=np.float64)
        if mean.shape != (fz.n,) or covariance.shape != (fz.n, fz.n):
            raise ValueError("the prediction must be a joint distribution over the new inducing points")
        h = C.c_void_p()
        ctx._check(ctx._lib.agp_sparse_fit_from_prediction(ctx._h, ctx.kernel(cov), C.byref(sz), _ptr(mean), _ptr(covariance),
                                                           fz.n, capi.HOST, self.inducing_nugget_, C.byref(h), None, None),
                   "agp_sparse_fit_from_prediction")
        return SparseFitModel(self, SparseGPFit(ctx, h, new_inducing_points, float("nan")))


def rebase_inducing_points(fit_model, new_inducing_points):
    """rebase_inducing_points (:714-725): a fit relative to new inducing points, from the old fit's joint prediction at
    them.  NOT equivalent to fitting with the new inducing points: information may be lost."""
    return fit_model.get_model().fit_from_prediction(new_inducing_points, fit_model.predict(new_inducing_points).joint())


def sparse_gp_from_covariance_and_mean(covariance_function, mean_function, grouper_function, strategy,
                                       model_name="sparse_gaussian_process_regression", context=None):
    """:740-757"""
    return SparseGaussianProcessRegression(covariance_function, mean_function, grouper_function, strategy, model_name,
                                           context)


def sparse_gp_from_covariance(covariance_function, grouper_function, strategy,
                              model_name="sparse_gaussian_process_regression", context=None):
    """:759-775"""
    return SparseGaussianProcessRegression(covariance_function, None, grouper_function, strategy, model_name, context)
