"""Host-side mirror of albatross's GaussianProcessRegression call surface.

Mirrors include/albatross/src/models/gp.hpp (GaussianProcessBase :170-463,
GaussianProcessRegression :487-505, factories :507-537), core/model.hpp,
core/fit_model.hpp and core/prediction.hpp: `gp_from_covariance(cov).fit(dataset)
.predict(features).mean()/.marginal()/.joint()`, `model.log_likelihood(dataset)`.
Every Gram / factor / solve / predict goes through the C-ABI in
include/albatross_amd.h (HIP kernels); nothing is computed on the CPU here.
"""
import ctypes as C

import numpy as np

from . import _capi as capi
from .covariance import (CovarianceFunction, FeatureSet, LinearCombination, Measurement, expand_linear_combinations,
                         expand_with_offsets, has_linear_combinations, nodes_to_array)


class AlbatrossAmdError(RuntimeError):
    def __init__(self, status, detail=""):
        self.status = status
        name = capi.load().agp_status_string(status).decode()
        super().__init__(f"albatross_amd: {name}" + (f" ({detail})" if detail else ""))


class NotPositiveDefiniteError(AlbatrossAmdError):
    pass


class NanInputError(AlbatrossAmdError):
    pass


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class Context:
    """agp_context: HIP streams + workspaces of one GPU."""

    def __init__(self, device_id=0):
        self._lib = capi.load()
        h = C.c_void_p()
        st = self._lib.agp_context_create(device_id, C.byref(h))
        if st != capi.AGP_OK:
            raise AlbatrossAmdError(st, "agp_context_create")
        self._h = h
        self.device_id = device_id
        self._kernels = {}

    def close(self):
        if getattr(self, "_h", None):
            for kh in self._kernels.values():
                self._lib.agp_kernel_destroy(kh)
            self._kernels = {}
            self._lib.agp_context_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # --- helpers ---------------------------------------------------------------
    def _check(self, st, what):
        if st == capi.AGP_OK:
            return
        detail = what
        if st == capi.AGP_ERR_HIP:
            detail += ": " + self._lib.agp_last_error(self._h).decode()
        if st == capi.AGP_ERR_NOT_POSITIVE_DEFINITE:
            raise NotPositiveDefiniteError(st, detail)
        if st == capi.AGP_ERR_NAN_INPUT:
            raise NanInputError(st, detail)
        raise AlbatrossAmdError(st, detail)

    def kernel(self, cov):
        """agp_kernel for the CURRENT parameter values of `cov` (cached by the
        flattened program bytes, so a tuner revisiting parameters re-uses it)."""
        nodes = cov.program_nodes()
        arr = nodes_to_array(nodes)
        key = bytes(arr)
        kh = self._kernels.pop(key, None)
        if kh is None:
            kh = C.c_void_p()
            self._check(self._lib.agp_kernel_create(arr, len(nodes), C.byref(kh)), "agp_kernel_create")
            # least-recently-used eviction, ONE entry at a time: a handle returned by this method stays valid for
            # the next KERNEL_CACHE - 1 calls at least (callers that collect more handles than that before using
            # them make private ones, see private_kernel)
            while len(self._kernels) >= self.KERNEL_CACHE:
                oldest = next(iter(self._kernels))
                self._lib.agp_kernel_destroy(self._kernels.pop(oldest))
        self._kernels[key] = kh  # (re-)inserted last = most recently used
        return kh

    KERNEL_CACHE = 64

    def private_kernel(self, cov):
        """an agp_kernel the CALLER owns (agp_kernel_destroy when done): outside the cache, never evicted"""
        nodes = cov.program_nodes()
        kh = C.c_void_p()
        self._check(self._lib.agp_kernel_create(nodes_to_array(nodes), len(nodes), C.byref(kh)), "agp_kernel_create")
        return kh

    def synchronize(self):
        """waits for everything queued on any of the context's streams (device-wide)"""
        self._check(self._lib.agp_context_synchronize(self._h), "synchronize")

    def to_device(self, array):
        """a DeviceArray holding a copy of `array` (fp64 / int64 numpy) in this context's HBM: what AGP_DEVICE arguments
        point at (agp_device_malloc + agp_memcpy; no second HIP runtime in the process)"""
        a = np.ascontiguousarray(array)
        d = DeviceArray(self, a.nbytes, a.dtype, a.shape)
        self._check(self._lib.agp_memcpy(self._h, C.c_void_p(d.ptr), C.c_void_p(a.ctypes.data), a.nbytes, capi.DEVICE), "agp_memcpy")
        return d

    def device_empty(self, shape, dtype=np.float64):
        shape = tuple(np.atleast_1d(shape).astype(np.int64).tolist())
        return DeviceArray(self, int(np.prod(shape)) * np.dtype(dtype).itemsize, np.dtype(dtype), shape)

    def set_profiling(self, enabled):
        self._check(self._lib.agp_set_profiling(self._h, 1 if enabled else 0), "set_profiling")

    def stage_ms(self, stage):
        v = C.c_double()
        self._check(self._lib.agp_last_stage_ms(self._h, stage, C.byref(v)), "last_stage_ms")
        return v.value

    def gram(self, cov, xs, ys=None):
        """compute_covariance_matrix (callers.hpp:38-166) on the device."""
        if has_linear_combinations(xs) or (ys is not None and has_linear_combinations(ys)):
            # LinearCombinationCaller (callers.hpp:321-396): Gram of the expanded points AND its contraction with the
            # coefficients on the device (agp_gram_combined)
            ex, xoff, xc = expand_with_offsets(xs)
            fx = cov.features(ex)
            sx = fx.as_struct()
            kh = self.kernel(cov)
            nx = len(xoff) - 1
            if ys is None:
                out = np.empty((nx, nx), order="F")
                st = self._lib.agp_gram_combined(self._h, kh, C.byref(sx), nx, _ptr(xoff), _ptr(xc), None, 0, None, None,
                                                 _ptr(out), max(nx, 1), capi.HOST)
            else:
                ey, yoff, yc = expand_with_offsets(ys)
                fy = cov.features(ey)
                sy = fy.as_struct()
                ny = len(yoff) - 1
                out = np.empty((nx, ny), order="F")
                st = self._lib.agp_gram_combined(self._h, kh, C.byref(sx), nx, _ptr(xoff), _ptr(xc), C.byref(sy), ny, _ptr(yoff),
                                                 _ptr(yc), _ptr(out), max(nx, 1), capi.HOST)
            self._check(st, "agp_gram_combined")
            return out
        fx = cov.features(xs)
        sx = fx.as_struct()
        kh = self.kernel(cov)
        if ys is None:
            out = np.empty((fx.n, fx.n), order="F")
            st = self._lib.agp_gram(self._h, kh, C.byref(sx), None, _ptr(out), max(fx.n, 1), capi.HOST)
        else:
            fy = cov.features(ys)
            sy = fy.as_struct()
            out = np.empty((fx.n, fy.n), order="F")
            st = self._lib.agp_gram(self._h, kh, C.byref(sx), C.byref(sy), _ptr(out), max(fx.n, 1), capi.HOST)
        self._check(st, "agp_gram")
        return out

    def gram_diagonal(self, cov, xs):
        if has_linear_combinations(xs):
            return np.diag(self.gram(cov, xs)).copy()
        f = cov.features(xs)
        return np.array([self.gram(cov, FeatureSet(f.coords[i:i + 1],
                                                   None if f.scales is None else list(f.scales[i:i + 1].T),
                                                   None if f.eq_id is None else f.eq_id[i:i + 1],
                                                   f.is_measurement))[0, 0] for i in range(f.n)])


_default_context = None


class DeviceArray:
    """`nbytes` of device memory owned by a Context (agp_device_malloc); `.ptr` is the integer device address that the
    AGP_DEVICE arguments of the C-ABI take, `.numpy()` downloads (after the context's streams have drained)."""

    def __init__(self, ctx, nbytes, dtype=np.float64, shape=None):
        self._ctx = ctx
        self.nbytes = int(nbytes)
        self.dtype = np.dtype(dtype)
        self.shape = tuple(shape) if shape is not None else (self.nbytes // self.dtype.itemsize,)
        p = C.c_void_p()
        ctx._check(ctx._lib.agp_device_malloc(ctx._h, max(self.nbytes, 8), C.byref(p)), "agp_device_malloc")
        self.ptr = p.value

    def numpy(self):
        out = np.empty(self.shape, dtype=self.dtype)
        self._ctx._check(self._ctx._lib.agp_memcpy(self._ctx._h, C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr), self.nbytes, capi.HOST),
                         "agp_memcpy")
        return out

    def free(self):
        if getattr(self, "ptr", None) and getattr(self._ctx, "_h", None):
            self._ctx._lib.agp_device_free(self._ctx._h, C.c_void_p(self.ptr))
        self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def default_context():
    global _default_context
    if _default_context is None:
        _default_context = Context(0)
    return _default_context


# ---------------------------------------------------------------------------
# core containers (core/dataset.hpp, core/distribution.hpp)
# ---------------------------------------------------------------------------
class MarginalDistribution:
    """mean + diagonal covariance (core/distribution.hpp)."""

    def __init__(self, mean, covariance=None):
        self.mean = np.asarray(mean, dtype=np.float64)
        self.covariance = None if covariance is None else np.asarray(covariance, dtype=np.float64)

    def size(self):
        return self.mean.shape[0]


class JointDistribution:
    def __init__(self, mean, covariance):
        self.mean = np.asarray(mean, dtype=np.float64)
        self.covariance = np.asarray(covariance, dtype=np.float64)

    def size(self):
        return self.mean.shape[0]

    def marginal(self):
        return MarginalDistribution(self.mean, np.diag(self.covariance).copy())


class RegressionDataset:
    """RegressionDataset<Feature>{features, targets} (core/dataset.hpp)."""

    def __init__(self, features, targets):
        self.features = features
        if not isinstance(targets, MarginalDistribution):
            targets = MarginalDistribution(targets)
        self.targets = targets

    def size(self):
        return self.targets.size()


class MeanFunction:
    """MeanFunction<Derived> (covariance_functions/mean_function.hpp:18-134): `m(coords)` is the mean vector
    (operator()(std::vector<X>), :73-84); `+` / `*` compose (:136-271).  `nodes()` is the function flattened to
    postfix tuples ("zero",) / ("linear", slope, offset) / ("sum",) / ("product",), like get_name() a description."""

    def get_params(self):
        return {}

    def set_param(self, name, value):
        raise KeyError(name)

    def __add__(self, other):
        return SumOfMeanFunctions(self, other)

    def __mul__(self, other):
        return ProductOfMeanFunctions(self, other)


class ZeroMean(MeanFunction):
    """mean_function.hpp:274-276"""

    def __call__(self, coords):
        return np.zeros(len(coords))

    def nodes(self):
        return [("zero",)]


class LinearMean(MeanFunction):
    """slope * x + offset on 1-D features (polynomials.hpp:92-106)."""

    def __init__(self, slope=0., offset=0.):
        self._params = {"slope": float(slope), "offset": float(offset)}

    def get_params(self):
        return dict(self._params)

    def set_param(self, name, value):
        if name not in self._params:
            raise KeyError(name)
        self._params[name] = float(value)

    def __call__(self, coords):
        x = np.asarray(coords, dtype=np.float64).reshape(len(coords), -1)[:, 0]
        return self._params["slope"] * x + self._params["offset"]

    def nodes(self):
        return [("linear", self._params["slope"], self._params["offset"])]


class _BinaryMean(MeanFunction):
    def __init__(self, lhs, rhs):
        self.lhs_, self.rhs_ = lhs, rhs

    def get_params(self):  # map_join(lhs_.get_params(), rhs_.get_params())
        out = dict(self.lhs_.get_params())
        out.update(self.rhs_.get_params())
        return out

    def set_param(self, name, value):  # set_param_if_exists_in_any
        done = False
        for side in (self.lhs_, self.rhs_):
            if name in side.get_params():
                side.set_param(name, value)
                done = True
        if not done:
            raise KeyError(name)


class SumOfMeanFunctions(_BinaryMean):
    """mean_function.hpp:136-190"""

    def __call__(self, coords):
        return self.lhs_(coords) + self.rhs_(coords)

    def nodes(self):
        return self.lhs_.nodes() + self.rhs_.nodes() + [("sum",)]


class ProductOfMeanFunctions(_BinaryMean):
    """mean_function.hpp:192-260: lhs * rhs with rhs skipped where lhs == 0 (:221-227)"""

    def __call__(self, coords):
        out = np.array(self.lhs_(coords), dtype=np.float64)
        nz = out != 0.
        if nz.any():
            out[nz] *= np.asarray(self.rhs_(coords), dtype=np.float64)[nz]
        return out

    def nodes(self):
        return self.lhs_.nodes() + self.rhs_.nodes() + [("product",)]


def _values_of(features):
    return features.values if isinstance(features, Measurement) else features


def _mean_at(mean_function, cov, features):
    """mean_function(features) including LinearCombination features: sum_i a_i m(x_i) (callers.hpp:386-396)"""
    if has_linear_combinations(features):
        ex, Cx = expand_linear_combinations(_values_of(features))
        return Cx.T @ mean_function(np.asarray(ex, dtype=np.float64))
    return mean_function(cov.features(features).coords)


# ---------------------------------------------------------------------------
# Fit<GPFit<...>> (gp.hpp:43-77): device factor + information vector
# ---------------------------------------------------------------------------
class _DeviceSolver:
    """A CovarianceRepresentation that lives on the device as an `agp_solver` (include/albatross_amd.h): the handle is made on
    first use and goes with the object.  Subclasses provide `_make_solver()`."""
    _sv = None

    def _solver(self):
        if self._sv is None:
            self._sv = self._make_solver()
        return self._sv

    def _drop_solver(self):
        try:
            if getattr(self, "_sv", None) and self._ctx._h:
                self._ctx._lib.agp_solver_destroy(self._sv)
        except Exception:
            pass
        self._sv = None

    def _solve_through_handle(self, rhs):
        rhs = np.asarray(rhs, dtype=np.float64)
        r2 = np.asfortranarray(rhs.reshape(rhs.shape[0], -1))
        out = np.empty_like(r2, order="F")
        self._ctx._check(self._ctx._lib.agp_solver_solve(self._ctx._h, self._solver(), _ptr(r2), r2.shape[1], _ptr(out), capi.HOST),
                         "agp_solver_solve")
        return out.reshape(rhs.shape, order="F")


class GPFit(_DeviceSolver):

    def _make_solver(self):
        h = C.c_void_p()
        self._ctx._check(self._ctx._lib.agp_solver_from_fit(self._ctx._h, self._h, C.byref(h)), "agp_solver_from_fit")
        return h

    def __init__(self, ctx, handle, n, train_features):
        self._ctx = ctx
        self._h = handle
        self.n = n
        self.train_features = train_features
        self._information = None

    def __del__(self):
        self._drop_solver()
        try:
            if getattr(self, "_h", None) and self._ctx._h:
                self._ctx._lib.agp_fit_destroy(self._h)
            self._h = None
        except Exception:
            pass

    @property
    def information(self):
        if self._information is None:
            out = np.empty(self.n)
            self._ctx._check(self._ctx._lib.agp_fit_download_information(self._ctx._h, self._h, _ptr(out)),
                             "download_information")
            self._information = out
        return self._information

    # a mixed-precision factor (agp_fit_create_mixed): its log-determinant carries the rounding of the bulk products.
    # MEASURED (include/albatross_amd.h; tests/test_full_size_configs_gpu.py holds each figure): default fp16 x 2 path 0.5e-6 N
    # on BASELINE config 4's covariance at N = 32768, 1.9e-6 N on config 3's kernel - inside the 2e-6 N log-likelihood bar,
    # the latter just; bf16 x 3 (AGP_MIXED_F16=0) 0.8e-6 N / 4.3e-6 N; fp32 fallback (AGP_MIXED_BF16=0) 1.3e-5 relative.
    # The bound depends on the covariance function and is not proven for an arbitrary one, so reading it stays an explicit opt-in.
    mixed_precision = False
    accept_mixed_log_determinant = False

    @property
    def log_determinant(self):
        if self.mixed_precision and not self.accept_mixed_log_determinant:
            raise AlbatrossAmdError(capi.AGP_ERR_UNSUPPORTED,
                                    "log_determinant of a mixed-precision factor carries the rounding of the bulk products "
                                    "(measured 0.5e-6 N ... 1.9e-6 N on the default fp16 x 2 path, up to 4.9e-6 N on bf16 x 3, 1.9e-5 N on the fp32 "
                                    "fallback; whether that meets the 2e-6 N bar depends on the covariance function): use model.log_likelihood "
                                    "(always fp64) or set fit.accept_mixed_log_determinant = True")
        v = C.c_double()
        self._ctx._check(self._ctx._lib.agp_fit_log_determinant(self._h, C.byref(v)), "log_determinant")
        return v.value

    def rows(self):
        return self.n

    def solve(self, rhs):
        """train_covariance.solve(rhs) (CovarianceRepresentation, gp.hpp:42-45)."""
        rhs = np.asarray(rhs, dtype=np.float64)
        b = np.asfortranarray(rhs.reshape(self.n, -1, order="F"))
        out = np.empty_like(b, order="F")
        self._ctx._check(self._ctx._lib.agp_solve(self._ctx._h, self._h, _ptr(b), b.shape[1], _ptr(out), capi.HOST),
                         "agp_solve")
        return out.reshape(rhs.shape, order="F")

    def inverse_diagonal(self):
        """SerializableLDLT::inverse_diagonal (serializable_ldlt.hpp:181-199)."""
        out = np.empty(self.n)
        self._ctx._check(self._ctx._lib.agp_fit_inverse_diagonal(self._ctx._h, self._h, _ptr(out), capi.HOST),
                         "agp_fit_inverse_diagonal")
        return out

    def leave_one_out(self, target_mean):
        """Leave-one-out predictive marginals of every training point
        (held_out_predictions with singleton groups, cross_validation_utils.hpp:165-232)."""
        y = np.ascontiguousarray(target_mean, dtype=np.float64)
        mean, var = np.empty(self.n), np.empty(self.n)
        self._ctx._check(self._ctx._lib.agp_loo_marginal(self._ctx._h, self._h, _ptr(y), _ptr(mean), _ptr(var),
                                                         capi.HOST), "agp_loo_marginal")
        return MarginalDistribution(mean, var)

    @staticmethod
    def _flatten_groups(groups, n):
        offsets = np.zeros(len(groups) + 1, dtype=np.int64)
        if len(groups):
            offsets[1:] = np.cumsum([len(g) for g in groups])
        parts = [np.asarray(g, dtype=np.int64).reshape(-1) for g in groups]
        indices = np.ascontiguousarray(np.concatenate(parts)) if parts else np.zeros(0, dtype=np.int64)
        if indices.size and (indices.min() < 0 or indices.max() >= n):
            raise IndexError("group index out of range")
        return offsets, indices

    def inverse_blocks(self, groups):
        """SerializableLDLT::inverse_blocks (serializable_ldlt.hpp:137-179): [(K^-1)[I_g, I_g] for g in groups]."""
        offsets, indices = self._flatten_groups(groups, self.n)
        out = np.empty(int(sum(len(g) ** 2 for g in groups)))
        self._ctx._check(self._ctx._lib.agp_fit_inverse_blocks(self._ctx._h, self._h, len(groups), _ptr(offsets),
                                                               _ptr(indices), _ptr(out), capi.HOST),
                         "agp_fit_inverse_blocks")
        blocks, pos = [], 0
        for g in groups:
            m = len(g)
            blocks.append(out[pos:pos + m * m].reshape(m, m, order="F").copy())
            pos += m * m
        return blocks

    def held_out_predictions(self, target_mean, groups, joint=False):
        """details::held_out_predictions (cross_validation_utils.hpp:165-232): for every index group the
        prediction of its targets from all other groups, without refitting.  Returns one
        MarginalDistribution (joint=False) or JointDistribution (joint=True) per group."""
        y = np.ascontiguousarray(target_mean, dtype=np.float64)
        if y.shape[0] != self.n:
            raise ValueError("target size")
        offsets, indices = self._flatten_groups(groups, self.n)
        total = int(offsets[-1])
        mean, var = np.empty(total), np.empty(total)
        jb = np.empty(int(sum(len(g) ** 2 for g in groups))) if joint else None
        self._ctx._check(self._ctx._lib.agp_held_out_predictions(
            self._ctx._h, self._h, _ptr(y), len(groups), _ptr(offsets), _ptr(indices), _ptr(mean), _ptr(var),
            _ptr(jb), capi.HOST), "agp_held_out_predictions")
        out, pos = [], 0
        for gi, g in enumerate(groups):
            m, o = len(g), int(offsets[gi])
            if joint:
                out.append(JointDistribution(mean[o:o + m].copy(), jb[pos:pos + m * m].reshape(m, m, order="F").copy()))
                pos += m * m
            else:
                out.append(MarginalDistribution(mean[o:o + m].copy(), var[o:o + m].copy()))
        return out

    def factor(self):
        L = np.empty((self.n, self.n), order="F")
        self._ctx._check(self._ctx._lib.agp_fit_download_factor(self._ctx._h, self._h, _ptr(L), self.n),
                         "download_factor")
        return L


class DenseFactor(_DeviceSolver):
    """`Eigen::SerializableLDLT(const MatrixXd &)` (eigen/serializable_ldlt.hpp:27): the
    device LL^T of a dense symmetric positive-definite matrix (lower triangle read)."""
    _make_solver = GPFit._make_solver

    def __init__(self, matrix, context=None):
        self._ctx = context or default_context()
        K = np.asarray(matrix, dtype=np.float64)
        if K.ndim != 2 or K.shape[0] != K.shape[1]:
            raise ValueError("square matrix expected")
        if not (K.flags.f_contiguous or K.flags.c_contiguous):
            K = np.asfortranarray(K)
        # the LOWER triangle of the array is what is read.  A C-ordered (row-major) array seen
        # column-major is its transpose, whose UPPER triangle that is: no host copy, uplo = 1.
        uplo = 0 if K.flags.f_contiguous else 1
        self.n = K.shape[0]
        h = C.c_void_p()
        st = self._ctx._lib.agp_factor_create(self._ctx._h, _ptr(K), self.n, self.n, uplo, capi.HOST, C.byref(h))
        if st != capi.AGP_OK:
            pivot = self._ctx._lib.agp_fit_failed_pivot(h) if h else -1
            if h:
                self._ctx._lib.agp_fit_destroy(h)
            self._ctx._check(st, f"agp_factor_create (pivot {pivot})")
        self._h = h

    def __del__(self):
        self._drop_solver()
        try:
            if getattr(self, "_h", None) and self._ctx._h:
                self._ctx._lib.agp_fit_destroy(self._h)
            self._h = None
        except Exception:
            pass

    def rows(self):
        return self.n

    solve = GPFit.solve
    inverse_diagonal = GPFit.inverse_diagonal
    factor = GPFit.factor
    mixed_precision = False  # (always an fp64 factor)
    log_determinant = GPFit.log_determinant


class PivotedLDLT(_DeviceSolver):
    """Eigen::LDLT<MatrixXd, Lower> as SerializableLDLT wraps it (eigen/serializable_ldlt.hpp:27): the
    diagonally pivoted P A P^T = L D L^T on the device, for symmetric matrices that are only semi-definite
    (the un-pivoted DenseFactor rejects those).  Same operation order as the reference's unblocked
    algorithm: matrix_ldlt(), vector_d() and transpositions() are bit-identical to the CPU restatement."""

    def __init__(self, matrix, context=None):
        self._ctx = context or default_context()
        K = np.asarray(matrix, dtype=np.float64)
        if K.ndim != 2 or K.shape[0] != K.shape[1]:
            raise ValueError("square matrix expected")
        if not (K.flags.f_contiguous or K.flags.c_contiguous):
            K = np.asfortranarray(K)
        uplo = 0 if K.flags.f_contiguous else 1  # lower triangle of the array, whatever its memory order
        self.n = K.shape[0]
        h, ok = C.c_void_p(), C.c_int(1)
        self._ctx._check(self._ctx._lib.agp_ldlt_create(self._ctx._h, _ptr(K), self.n, self.n, uplo, capi.HOST,
                                                        C.byref(h), C.byref(ok)), "agp_ldlt_create")
        self._h = h
        self.success = bool(ok.value)  # info() == Eigen::Success

    def _make_solver(self):
        h = C.c_void_p()
        self._ctx._check(self._ctx._lib.agp_solver_from_ldlt(self._ctx._h, self._h, C.byref(h)), "agp_solver_from_ldlt")
        return h

    def __del__(self):
        self._drop_solver()
        try:
            if getattr(self, "_h", None) and self._ctx._h:
                self._ctx._lib.agp_ldlt_destroy(self._h)
            self._h = None
        except Exception:
            pass

    def rows(self):
        return self.n

    def solve(self, rhs):
        """LDLT::solve: P^T L^-T D^+ L^-1 P rhs."""
        rhs = np.asarray(rhs, dtype=np.float64)
        r2 = np.asfortranarray(rhs.reshape(rhs.shape[0], -1))
        out = np.empty_like(r2, order="F")
        self._ctx._check(self._ctx._lib.agp_ldlt_solve(self._ctx._h, self._h, _ptr(r2), r2.shape[1], _ptr(out),
                                                       capi.HOST), "agp_ldlt_solve")
        return out.reshape(rhs.shape, order="F")

    def sqrt_solve(self, rhs):
        """SerializableLDLT::sqrt_solve (:99-109): D^-1/2 L^-1 P rhs."""
        rhs = np.asarray(rhs, dtype=np.float64)
        r2 = np.asfortranarray(rhs.reshape(rhs.shape[0], -1))
        out = np.empty_like(r2, order="F")
        self._ctx._check(self._ctx._lib.agp_ldlt_sqrt_solve(self._ctx._h, self._h, _ptr(r2), r2.shape[1], _ptr(out),
                                                            capi.HOST), "agp_ldlt_sqrt_solve")
        return out.reshape(rhs.shape, order="F")

    def vector_d(self):
        d = np.empty(self.n)
        self._ctx._check(self._ctx._lib.agp_ldlt_vector_d(self._h, _ptr(d)), "agp_ldlt_vector_d")
        return d

    def transpositions(self):
        tr = np.empty(self.n, dtype=np.int64)
        self._ctx._check(self._ctx._lib.agp_ldlt_transpositions(self._h, _ptr(tr)), "agp_ldlt_transpositions")
        return tr

    def matrix_ldlt(self):
        out = np.empty((self.n, self.n), order="F")
        self._ctx._check(self._ctx._lib.agp_ldlt_download(self._ctx._h, self._h, _ptr(out), self.n), "agp_ldlt_download")
        return out

    @property
    def log_determinant(self):
        """serializable_ldlt.hpp:128-135: sum(log(vectorD)) (NaN / -inf for non-positive pivots, like the reference)."""
        with np.errstate(divide="ignore", invalid="ignore"):
            return float(np.log(self.vector_d()).sum())


def negative_log_likelihood(deviation, covariance, context=None):
    """negative_log_likelihood(deviation, covariance) (evaluation/likelihood.hpp:53-66)."""
    ctx = context or default_context()
    d = np.ascontiguousarray(deviation, dtype=np.float64)
    K = np.asarray(covariance, dtype=np.float64)
    if not (K.flags.f_contiguous or K.flags.c_contiguous):
        K = np.asfortranarray(K)
    uplo = 0 if K.flags.f_contiguous else 1
    out = C.c_double()
    ctx._check(ctx._lib.agp_nll_dense(ctx._h, _ptr(d), _ptr(K), d.shape[0], K.shape[0], uplo, capi.HOST,
                                      C.byref(out)), "agp_nll_dense")
    return out.value


class BlockSymmetric(_DeviceSolver):
    """linalg/block_symmetric.hpp:46-115: solver of [[A, B], [B^T, C]] from a solver of A, Ai_B = A^-1 B and the factor of
    the Schur complement S = C - B^T A^-1 B - ON THE DEVICE (agp_solver_block_symmetric): Ai_B is computed and kept in HBM,
    a solve is device solves + MFMA products.  Only solvers that are not a plain device factor end up here (pivoted
    LDL^T fits, fit_from_prediction); fits on the device factor are updated by agp_fit_update."""

    def __init__(self, A, B, S):
        self._ctx = A._ctx
        self.A, self.S = A, S  # (kept alive: the device object borrows their solvers)
        self._B = np.asfortranarray(B, dtype=np.float64)
        if self._B.shape != (A.rows(), S.rows()):
            raise ValueError("BlockSymmetric: B must be rows(A) x rows(S)")

    def _make_solver(self):
        h = C.c_void_p()
        self._ctx._check(self._ctx._lib.agp_solver_block_symmetric(self._ctx._h, self.A._solver(), _ptr(self._B), self._B.shape[0], capi.HOST,
                                                                   self.S._solver(), C.byref(h)), "agp_solver_block_symmetric")
        return h

    def __del__(self):
        self._drop_solver()

    def rows(self):
        return self.A.rows() + self.S.rows()

    def solve(self, rhs):  # block_symmetric.hpp:75-98
        return self._solve_through_handle(rhs)

    def update_information(self, information, si_delta):
        """gp.hpp:403-407: [information - Ai_B Si_delta ; Si_delta] from the Ai_B this solver holds in HBM
        (agp_solver_update_information: one mat-vec on the device)."""
        info = np.ascontiguousarray(information, dtype=np.float64)
        sd = np.ascontiguousarray(si_delta, dtype=np.float64)
        out = np.empty(info.shape[0] + sd.shape[0])
        self._ctx._check(self._ctx._lib.agp_solver_update_information(self._ctx._h, self._solver(), _ptr(info), _ptr(sd), _ptr(out), capi.HOST),
                         "agp_solver_update_information")
        return out


class ExplainedCovariance(_DeviceSolver):
    """ExplainedCovariance (covariance_functions/representations.hpp:64-96): S^-1 = A^-1 B A^-1 with the outer matrix A
    held through its factor and the inner matrix B kept as it is, because B may be singular - on the device
    (agp_solver_explained): the product with B between the two solves is an MFMA product in HBM."""

    def __init__(self, outer, inner, context=None):
        self.outer_ldlt = outer if isinstance(outer, (DenseFactor, PivotedLDLT)) else DenseFactor(outer, context)
        self._ctx = self.outer_ldlt._ctx
        self.inner = np.asfortranarray(inner, dtype=np.float64)

    def _make_solver(self):
        h = C.c_void_p()
        self._ctx._check(self._ctx._lib.agp_solver_explained(self._ctx._h, self.outer_ldlt._solver(), _ptr(self.inner), self.inner.shape[0],
                                                             capi.HOST, C.byref(h)), "agp_solver_explained")
        return h

    def __del__(self):
        self._drop_solver()

    def rows(self):
        return self.inner.shape[0]

    def solve(self, rhs):  # representations.hpp:80-82
        return self._solve_through_handle(rhs)


class UpdatedGPFit:
    """Fit<GPFit<Representation, F>> whose solver is not the plain LL^T factor: a pivoted L D L^T (semi-definite covariances),
    BlockSymmetric<Solver> from update() (gp.hpp:384-414) or ExplainedCovariance from fit_from_prediction (gp.hpp:139-153).
    The solver is a device object (`agp_solver`) and predictions go through agp_solver_predict - the generic
    CovarianceRepresentation form of _predict_impl (gp.hpp:305-366) in HBM - agp_solver_predict_combined when either side holds
    LinearCombination features (their covariance matrices are built and contracted on the device too)."""

    def __init__(self, train_features, train_covariance, information):
        self.train_features = train_features
        self.train_covariance = train_covariance
        self.information = information
        self.n = information.shape[0]

    def rows(self):
        return self.n

    def solve(self, rhs):
        return self.train_covariance.solve(rhs)


class Prediction:
    """Lazy prediction object (core/prediction.hpp:115-224)."""

    def __init__(self, fit_model, features):
        self._fm = fit_model
        self._features = features

    def mean(self):
        return self._fm._predict_mean(self._features)

    def marginal(self):
        return self._fm._predict_marginal(self._features)

    def joint(self):
        return self._fm._predict_joint(self._features)


class FitModel:
    """FitModel<Model, Fit> (core/fit_model.hpp:18-114)."""

    def __init__(self, model, fit):
        self._model = model
        self._fit = fit

    def get_fit(self):
        return self._fit

    def get_model(self):
        return self._model

    def predict(self, features):
        return Prediction(self, features)

    def predict_with_measurement_noise(self, features):
        """fit_model.hpp:54-62: wraps the test features in Measurement<>."""
        return Prediction(self, features if isinstance(features, Measurement) else Measurement(features))

    def update(self, dataset, targets=None):
        """FitModel::update (core/fit_model.hpp:68-81) -> _update_impl (gp.hpp:384-414):
        condition the fit on further observations through the Schur complement instead of
        re-factoring; returns a new FitModel whose fit holds a BlockSymmetric solver."""
        if targets is not None:
            dataset = RegressionDataset(dataset, targets)
        m, ctx = self._model, self._model._ctx()
        feats = _values_of(dataset.features)
        if isinstance(self._fit, GPFit) and not has_linear_combinations(feats):
            return self._update_on_device(dataset, feats)
        pred = self.predict(feats).joint()                                   # gp.hpp:388-389
        delta = np.asarray(dataset.targets.mean, dtype=np.float64) - pred.mean
        S = np.array(pred.covariance)
        if dataset.targets.covariance is not None:
            S[np.diag_indices_from(S)] += dataset.targets.covariance         # pred.covariance += targets.covariance
        S_ldlt = DenseFactor(S, ctx)                                          # gp.hpp:393
        old_feats = _values_of(self._fit.train_features)
        cross = ctx.gram(m.covariance_function_, old_feats, feats)            # gp.hpp:395-396
        solver_a = self._fit.train_covariance if isinstance(self._fit, UpdatedGPFit) else self._fit
        new_cov = BlockSymmetric(solver_a, cross, S_ldlt)                     # gp.hpp:398-399
        Si_delta = S_ldlt.solve(delta)
        info = new_cov.update_information(self._fit.information, Si_delta)   # gp.hpp:403-407, on the device
        new_feats = np.concatenate([np.asarray(old_feats, dtype=np.float64).reshape(self._fit.rows(), -1),
                                    np.asarray(feats, dtype=np.float64).reshape(len(delta), -1)])
        return FitModel(m, UpdatedGPFit(new_feats, new_cov, info))

    def _update_on_device(self, dataset, feats):
        """agp_fit_update: the resident factor grows by one block row (V^T = (L^-1 B)^T, L_S from the Schur complement);
        triangular solve, SYRK, LL^T of the new block and the back substitution all run in the HIP library."""
        m, ctx = self._model, self._model._ctx()
        cov = m.covariance_function_
        fs = cov.features(feats)
        y, yv = m._targets(fs, dataset.targets)      # mean function removed (ModelBase::update -> remove_from)
        s = fs.as_struct()
        h = C.c_void_p()
        st = ctx._lib.agp_fit_update(ctx._h, ctx.kernel(cov), self._fit._h, C.byref(s), _ptr(y), _ptr(yv), C.byref(h), None, None)
        if st != capi.AGP_OK:
            pivot = ctx._lib.agp_fit_failed_pivot(h) if h else -1
            if h:
                ctx._lib.agp_fit_destroy(h)
            ctx._check(st, f"agp_fit_update (pivot {pivot})")
        old_feats = _values_of(self._fit.train_features)
        try:
            new_feats = np.concatenate([np.asarray(old_feats, dtype=np.float64).reshape(self._fit.rows(), -1),
                                        np.asarray(feats, dtype=np.float64).reshape(fs.n, -1)])
        except (TypeError, ValueError):  # feature containers numpy cannot stack: keep them side by side
            new_feats = (old_feats, feats)
        return FitModel(m, GPFit(ctx, h, self._fit.rows() + fs.n, new_feats))

    # --- _predict_impl (gp.hpp:305-366) ------------------------------------------
    def _host_predict(self, features, want):
        """_predict_impl written against a generic CovarianceRepresentation (gp.hpp:305-366), for fits whose solver is not the
        plain LL^T factor: agp_solver_predict - cross covariance, solve, explained covariance all in HBM."""
        m, ctx = self._model, self._model._ctx()
        cov = m.covariance_function_
        if not (has_linear_combinations(features) or has_linear_combinations(self._fit.train_features)):
            fs = cov.features(features)
            ftr = cov.features(self._fit.train_features)
            s_xs, s_tr = fs.as_struct(), ftr.as_struct()
            info = np.ascontiguousarray(self._fit.information, dtype=np.float64)
            mode = {"mean": 0, "marginal": 1, "joint": 2}[want]
            mean = np.empty(fs.n)
            second = None if mode == 0 else (np.empty(fs.n) if mode == 1 else np.empty((fs.n, fs.n), order="F"))
            ctx._check(ctx._lib.agp_solver_predict(ctx._h, ctx.kernel(cov), self._fit.train_covariance._solver(), C.byref(s_tr), _ptr(info),
                                                   C.byref(s_xs), _ptr(mean), None if second is None else _ptr(second), mode, capi.HOST),
                       "agp_solver_predict")
            mean = mean + m.mean_function_(fs.coords)  # mean_function_.add_to, gp.hpp:364
            if want == "mean":
                return mean
            return MarginalDistribution(mean, second) if want == "marginal" else JointDistribution(mean, second)
        # LinearCombination features on either side (callers.hpp:321-396): the same composition with the contracted covariance
        # matrices of agp_gram_combined, all of it in HBM (agp_solver_predict_combined)
        def side(f):
            if has_linear_combinations(f):
                ex, off, co = expand_with_offsets(f)
                return cov.features(ex), len(off) - 1, off, co
            fs_ = cov.features(f)
            return fs_, fs_.n, None, None
        ftr, ntr, troff, trco = side(self._fit.train_features)
        fxs, nxs, xoff, xco = side(features)
        s_tr, s_xs = ftr.as_struct(), fxs.as_struct()
        solver = self._fit.train_covariance._solver() if isinstance(self._fit, UpdatedGPFit) else self._fit._solver()
        info = np.ascontiguousarray(self._fit.information, dtype=np.float64)
        mode = {"mean": 0, "marginal": 1, "joint": 2}[want]
        mean = np.empty(nxs)
        second = None if mode == 0 else (np.empty(nxs) if mode == 1 else np.empty((nxs, nxs), order="F"))
        ctx._check(ctx._lib.agp_solver_predict_combined(
            ctx._h, ctx.kernel(cov), solver, C.byref(s_tr), ntr, None if troff is None else _ptr(troff), None if trco is None else _ptr(trco),
            _ptr(info), C.byref(s_xs), nxs, None if xoff is None else _ptr(xoff), None if xco is None else _ptr(xco), _ptr(mean),
            None if second is None else _ptr(second), mode, capi.HOST), "agp_solver_predict_combined")
        mean = mean + _mean_at(m.mean_function_, cov, features)  # mean_function_.add_to, gp.hpp:364
        if want == "mean":
            return mean
        return MarginalDistribution(mean, second) if want == "marginal" else JointDistribution(mean, second)

    def _xs(self, features):
        fs = self._model.covariance_function_.features(features)
        return fs, fs.as_struct()

    def _predict_mean(self, features):
        if isinstance(self._fit, UpdatedGPFit):
            return self._host_predict(features, "mean")
        m, ctx = self._model, self._model._ctx()
        fs, s = self._xs(features)
        mean = np.empty(fs.n)
        ctx._check(ctx._lib.agp_predict_mean(ctx._h, ctx.kernel(m.covariance_function_), self._fit._h, C.byref(s),
                                             _ptr(mean), capi.HOST), "agp_predict_mean")
        return mean + m.mean_function_(fs.coords)  # mean_function_.add_to, gp.hpp:364

    def _predict_marginal(self, features):
        if isinstance(self._fit, UpdatedGPFit):
            return self._host_predict(features, "marginal")
        m, ctx = self._model, self._model._ctx()
        fs, s = self._xs(features)
        mean, var = np.empty(fs.n), np.empty(fs.n)
        ctx._check(ctx._lib.agp_predict_marginal(ctx._h, ctx.kernel(m.covariance_function_), self._fit._h,
                                                 C.byref(s), _ptr(mean), _ptr(var), capi.HOST),
                   "agp_predict_marginal")
        return MarginalDistribution(mean + m.mean_function_(fs.coords), var)

    def _predict_joint(self, features):
        if isinstance(self._fit, UpdatedGPFit):
            return self._host_predict(features, "joint")
        m, ctx = self._model, self._model._ctx()
        fs, s = self._xs(features)
        mean, cov = np.empty(fs.n), np.empty((fs.n, fs.n), order="F")
        ctx._check(ctx._lib.agp_predict_joint(ctx._h, ctx.kernel(m.covariance_function_), self._fit._h, C.byref(s),
                                              _ptr(mean), _ptr(cov), capi.HOST), "agp_predict_joint")
        return JointDistribution(mean + m.mean_function_(fs.coords), cov)


class GaussianProcessRegression:
    """GaussianProcessRegression<CovFunc, MeanFunc> (gp.hpp:487-505)."""

    def __init__(self, covariance_function, mean_function=None, model_name="gaussian_process_regression",
                 context=None):
        if not isinstance(covariance_function, CovarianceFunction):
            raise TypeError("covariance_function must be an albatross_amd CovarianceFunction")
        self.covariance_function_ = covariance_function
        self.mean_function_ = mean_function or ZeroMean()
        self.model_name_ = model_name
        self._context = context

    def _ctx(self):
        return self._context or default_context()

    def get_name(self):
        return self.model_name_

    def get_covariance(self):
        return self.covariance_function_

    def get_mean(self):
        return self.mean_function_

    # ParameterHandlingMixin subset (gp.hpp:255-268)
    def get_params(self):
        out = dict(self.covariance_function_.get_params())
        out.update(self.mean_function_.get_params())
        return out

    def set_param(self, name, value):
        if name in self.covariance_function_.get_params():
            self.covariance_function_.set_param(name, value)
        elif name in self.mean_function_.get_params():
            self.mean_function_.set_param(name, value)
        else:
            raise KeyError(name)

    def set_param_values(self, values):
        for k, v in values.items():
            self.set_param(k, v)

    set_params = set_param_values

    def _targets(self, fs, targets):
        y = np.ascontiguousarray(targets.mean - self.mean_function_(fs.coords), dtype=np.float64)  # remove_from
        yv = None
        if targets.covariance is not None:
            yv = np.ascontiguousarray(targets.covariance, dtype=np.float64)
            if yv.ndim != 1 or yv.shape[0] != y.shape[0]:
                raise ValueError("target covariance must be the diagonal (one variance per target)")
        if y.shape[0] != fs.n:
            raise ValueError("features and targets differ in size")
        return y, yv

    def fit(self, dataset, targets=None):
        """ModelBase::fit (core/model.hpp:137-152) -> _fit_impl (gp.hpp:281-294)."""
        if targets is not None:
            dataset = RegressionDataset(dataset, targets)
        ctx = self._ctx()
        if has_linear_combinations(dataset.features):
            return self._fit_dense(dataset)
        fs = self.covariance_function_.features(_values_of(dataset.features))
        y, yv = self._targets(fs, dataset.targets)
        s = fs.as_struct()
        h = C.c_void_p()
        if self.precision == "mixed":
            # agp_fit_create_mixed: fp32-product bulk updates + fp64 conjugate-gradient refinement of the
            # information vector (BASELINE configs[3]); `refinement_` keeps (iterations, relative residual)
            it, res = C.c_int(0), C.c_double(0.)
            st = ctx._lib.agp_fit_create_mixed(ctx._h, ctx.kernel(self.covariance_function_), C.byref(s), _ptr(y),
                                               _ptr(yv), int(self.max_refinements), float(self.refinement_tolerance),
                                               C.byref(h), None, None, C.byref(it), C.byref(res))
            self.refinement_ = (it.value, res.value)
        elif self.precision == "fp64":
            st = ctx._lib.agp_fit_create(ctx._h, ctx.kernel(self.covariance_function_), C.byref(s), _ptr(y), _ptr(yv),
                                         C.byref(h), None, None)
        else:
            raise ValueError(f"precision must be 'fp64' or 'mixed', not {self.precision!r}")
        if st != capi.AGP_OK:
            pivot = ctx._lib.agp_fit_failed_pivot(h) if h else -1
            if h:
                ctx._lib.agp_fit_destroy(h)
            if st == capi.AGP_ERR_NOT_POSITIVE_DEFINITE and self.pivoted_fallback:
                return self._fit_pivoted(dataset, fs, y, yv)
            ctx._check(st, f"agp_fit_create (pivot {pivot})")
        fit = GPFit(ctx, h, fs.n, dataset.features)
        fit.mixed_precision = self.precision == "mixed"
        return FitModel(self, fit)

    pivoted_fallback = True
    # 'fp64' (the reference's arithmetic) or 'mixed' (fp32 MFMA products in the bulk updates of the factorisation,
    # information vector refined to fp64; see agp_fit_create_mixed in include/albatross_amd.h)
    precision = "fp64"
    max_refinements = 50
    refinement_tolerance = 1e-12
    refinement_ = None

    def _fit_pivoted(self, dataset, fs, y, yv):
        """The reference's own route for covariances that are only positive SEMI-definite ("unobservable" models,
        tests/test_gp.cc:20-33): Fit<GPFit<SerializableLDLT>> with the pivoted L D L^T (gp.hpp:61-69).  Gram and
        factor run on the device; predictions go through the generic CovarianceRepresentation form of
        _predict_impl.  Set `pivoted_fallback = False` to get NotPositiveDefiniteError instead."""
        ctx = self._ctx()
        feats = _values_of(dataset.features)
        K = ctx.gram(self.covariance_function_, Measurement(feats))  # as_measurements(features), gp.hpp:288-290
        if yv is not None:
            K[np.diag_indices_from(K)] += yv                          # gp.hpp:65
        if np.isnan(K).any():
            raise NanInputError(capi.AGP_ERR_NAN_INPUT, "covariance has NaN")  # gp.hpp:66
        ldlt = PivotedLDLT(K, ctx)                                    # gp.hpp:67
        return FitModel(self, UpdatedGPFit(feats, ldlt, ldlt.solve(y)))  # gp.hpp:68

    def _fit_dense(self, dataset):
        """Datasets with LinearCombination features: Fit<GPFit<SerializableLDLT>> (gp.hpp:61-69) from the contracted
        covariance matrix.  Gram of the expanded points, factorisation and solves run on the device; the
        contraction with the coefficients is host work."""
        ctx = self._ctx()
        feats = _values_of(dataset.features)
        y = np.ascontiguousarray(dataset.targets.mean - _mean_at(self.mean_function_, self.covariance_function_, feats))
        K = ctx.gram(self.covariance_function_, Measurement(feats))   # as_measurements(features), gp.hpp:288-290
        if dataset.targets.covariance is not None:
            K[np.diag_indices_from(K)] += np.asarray(dataset.targets.covariance, dtype=np.float64)  # gp.hpp:65
        if np.isnan(K).any():
            raise NanInputError(capi.AGP_ERR_NAN_INPUT, "covariance has NaN")                     # gp.hpp:66
        try:
            factor = DenseFactor(K, ctx)
        except NotPositiveDefiniteError:
            if not self.pivoted_fallback:
                raise
            factor = PivotedLDLT(K, ctx)
        return FitModel(self, UpdatedGPFit(feats, factor, factor.solve(y)))

    def cross_validate(self):
        return CrossValidation(self)

    def fit_from_prediction(self, features, prediction):
        """fit_from_prediction (gp.hpp:236-245) -> gp_fit_from_prediction (gp.hpp:139-153): the model that
        reproduces a joint prediction at `features`: information = prior^-1 (mean - mean_function),
        train_covariance = ExplainedCovariance(prior_ldlt, prior - prediction.covariance)."""
        ctx = self._ctx()
        feats = _values_of(features)
        fs = self.covariance_function_.features(feats)
        mean = np.asarray(prediction.mean, dtype=np.float64) - self.mean_function_(fs.coords)  # remove_from, :240
        prior = ctx.gram(self.covariance_function_, feats)                                      # :243
        try:
            prior_ldlt = DenseFactor(prior, ctx)
        except NotPositiveDefiniteError:  # low-rank priors (tests/test_gp.cc:308-341): the pivoted factor
            prior_ldlt = PivotedLDLT(prior, ctx)
        cov = ExplainedCovariance(prior_ldlt, prior - np.asarray(prediction.covariance, dtype=np.float64))
        info = prior_ldlt.solve(mean)
        return FitModel(self, UpdatedGPFit(feats, cov, info))

    def log_likelihoods(self, dataset, parameter_sets):
        """log_likelihood(dataset) for several parameter vectors at once (agp_nll_batch): the evaluations
        compute_gradient (tune/finite_difference.hpp:20-94) and ModelTuner (tune/tune.hpp:151-161) make one
        after the other.  parameter_sets: iterable of {name: value} overrides of the current parameters.
        A parameter vector whose covariance is not positive definite gives NaN."""
        import copy
        ctx = self._ctx()
        models = []
        for overrides in parameter_sets:
            m = copy.copy(self)
            m.covariance_function_ = copy.deepcopy(self.covariance_function_)
            m.mean_function_ = copy.deepcopy(self.mean_function_)
            m.set_param_values(overrides)
            models.append(m)
        count = len(models)
        if count == 0:
            return np.zeros(0)
        feats = _values_of(dataset.features)
        fsets, structs, ys = [], [], []
        yv = None
        for m in models:
            fs = m.covariance_function_.features(feats)
            y, yv = m._targets(fs, dataset.targets)
            fsets.append(fs)
            structs.append(fs.as_struct())
            ys.append(y)
        n = fsets[0].n
        Y = np.asfortranarray(np.stack(ys, axis=1))
        # private handles for the duration of the call: the context's cache may evict while the batch is assembled
        handles = []
        try:
            for m in models:
                handles.append(ctx.private_kernel(m.covariance_function_))
            kernels = (C.c_void_p * count)(*handles)
            fptrs = (C.c_void_p * count)(*[C.addressof(st) for st in structs])
            out = np.empty(count)
            # log_likelihood ignores the target variance (gp.hpp:442-451): y_var = NULL
            ctx._check(ctx._lib.agp_nll_batch(ctx._h, count, kernels, fptrs, _ptr(Y), n, None, _ptr(out)),
                       "agp_nll_batch")
        finally:
            for kh in handles:
                ctx._lib.agp_kernel_destroy(kh)
        return -out

    def log_likelihood(self, dataset):
        """gp.hpp:442-451 (without priors: the parameter-prior subsystem is out of scope).  Like the reference, the
        covariance is covariance_function_(measurement_features) ALONE: dataset.targets.covariance is not added."""
        ctx = self._ctx()
        if has_linear_combinations(dataset.features):
            feats = _values_of(dataset.features)
            K = ctx.gram(self.covariance_function_, Measurement(feats))
            dev = dataset.targets.mean - _mean_at(self.mean_function_, self.covariance_function_, feats)
            return -negative_log_likelihood(dev, K, ctx)
        fs = self.covariance_function_.features(_values_of(dataset.features))
        y, _ = self._targets(fs, dataset.targets)
        s = fs.as_struct()
        out = C.c_double()
        ctx._check(ctx._lib.agp_nll(ctx._h, ctx.kernel(self.covariance_function_), C.byref(s), _ptr(y), None,
                                    C.byref(out)), "agp_nll")
        return -out.value


class LeaveOneOutGrouper:
    """LeaveOneOutGrouper (indexing/group_by.hpp): every observation is its own group."""

    def __call__(self, feature, index=None):
        return index


def group_indexer(features, grouper):
    """dataset.group_by(grouper).indexers(): ordered {key: [indices]} (std::map order = sorted keys)."""
    feats = _values_of(features)
    groups = {}
    for i in range(len(feats)):
        key = i if isinstance(grouper, LeaveOneOutGrouper) else grouper(feats[i])
        groups.setdefault(key, []).append(i)
    return dict(sorted(groups.items(), key=lambda kv: kv[0]))


class CrossValidationPrediction:
    """Prediction<CrossValidation<Model>, Feature, GroupIndexer> (evaluation/cross_validation.hpp:28-260).

    means() / marginals() / joints() use the GP fast path (gp_cross_validated_predictions, gp.hpp:465-482:
    ONE fit, then held_out_predictions); predictions() is the generic refit-per-fold path."""

    def __init__(self, model, dataset, indexer):
        self.model_, self.dataset_, self.indexer_ = model, dataset, indexer
        self._fit_model = None

    def _fit(self):
        if self._fit_model is None:
            self._fit_model = self.model_.fit(self.dataset_)
        return self._fit_model.get_fit()

    def _held_out(self, joint):
        groups = list(self.indexer_.values())
        preds = self._fit().held_out_predictions(self.dataset_.targets.mean, groups, joint=joint)
        return dict(zip(self.indexer_.keys(), preds))

    def predictions(self):
        """predict_fold for every group: fit on the rest, predict the group (cross_validation.hpp:20-43)."""
        feats = _values_of(self.dataset_.features)
        y, yv = self.dataset_.targets.mean, self.dataset_.targets.covariance
        n = len(feats)
        out = {}
        for key, idx in self.indexer_.items():
            held = np.zeros(n, dtype=bool)
            held[np.asarray(idx)] = True
            train = np.nonzero(~held)[0]
            tr_feats = [feats[i] for i in train] if isinstance(feats, list) else feats[train]
            te_feats = [feats[i] for i in idx] if isinstance(feats, list) else feats[np.asarray(idx)]
            targets = MarginalDistribution(y[train], None if yv is None else yv[train])
            out[key] = self.model_.fit(RegressionDataset(tr_feats, targets)).predict(te_feats)
        return out

    def means(self):
        return {k: p.mean for k, p in self._held_out(False).items()}

    def marginals(self):
        return self._held_out(False)

    def joints(self):
        return self._held_out(True)

    def _concatenate(self, per_group):
        n = self.dataset_.size()
        out = np.empty(n)
        for key, idx in self.indexer_.items():
            out[np.asarray(idx)] = per_group[key]
        return out

    def mean(self):
        """concatenate_mean_predictions: group results scattered back to dataset order."""
        return self._concatenate(self.means())

    def marginal(self):
        m = self.marginals()
        return MarginalDistribution(self._concatenate({k: p.mean for k, p in m.items()}),
                                    self._concatenate({k: p.covariance for k, p in m.items()}))


class CrossValidation:
    """model.cross_validate() (core/model.hpp:154-156, evaluation/cross_validation.hpp:262-330)."""

    def __init__(self, model):
        self.model_ = model

    def predict(self, dataset, grouper):
        indexer = grouper if isinstance(grouper, dict) else group_indexer(dataset.features, grouper)
        return CrossValidationPrediction(self.model_, dataset, indexer)

    def predictions(self, dataset, grouper):
        return self.predict(dataset, grouper).predictions()

    def scores(self, metric, dataset, grouper):
        """cross_validated_scores: metric(prediction of the group, truth of the group) per group."""
        pred = self.predict(dataset, grouper)
        marg = pred.marginals()
        y = dataset.targets.mean
        return np.array([metric(marg[k], MarginalDistribution(y[np.asarray(idx)])) for k, idx in pred.indexer_.items()])


def root_mean_square_error(prediction, truth):
    """RootMeanSquareError (evaluation/prediction_metrics.hpp)."""
    d = prediction.mean - truth.mean
    return float(np.sqrt(np.mean(d * d)))


def fit_batch(models, datasets):
    """`models[b].fit(datasets[b])` for several problems of ONE size in lock step (agp_fit_create_batch): the regime of the
    reference's own workloads (benchmarks/bench_predict.cc: N = 512; one fit per tuner step), where a single fit is bound by the
    latency of its serial pivots - a batch shares it.  models: GaussianProcessRegression objects on one context (one model
    may appear several times); datasets: as many RegressionDatasets with the same number of points.  Returns the FitModels;
    raises like `fit` for the first problem whose covariance has NaN or is not positive definite."""
    if len(models) != len(datasets) or not models:
        raise ValueError("fit_batch: as many datasets as models, at least one")
    ctx = models[0]._ctx()
    count = len(models)
    fsets, structs, ys, yvs = [], [], [], []
    for m, ds in zip(models, datasets):
        if m.precision != "fp64" or has_linear_combinations(ds.features) or m._ctx() is not ctx:
            raise ValueError("fit_batch: fp64 models on one context, plain features")
        fs = m.covariance_function_.features(_values_of(ds.features))
        y, yv = m._targets(fs, ds.targets)
        fsets.append(fs)
        structs.append(fs.as_struct())
        ys.append(y)
        yvs.append(yv)
    n = fsets[0].n
    if any(fs.n != n for fs in fsets):
        raise ValueError("fit_batch: every dataset must have the same number of points")
    Y = np.asfortranarray(np.stack(ys, axis=1))
    have_var = any(v is not None for v in yvs)
    V = np.asfortranarray(np.stack([np.zeros(n) if v is None else v for v in yvs], axis=1)) if have_var else None
    handles = []
    out = (C.c_void_p * count)()
    status = (C.c_int * count)()
    try:
        for m in models:  # private handles: the context's small kernel cache may evict while the batch is assembled
            handles.append(ctx.private_kernel(m.covariance_function_))
        kernels = (C.c_void_p * count)(*handles)
        fptrs = (C.c_void_p * count)(*[C.addressof(st) for st in structs])
        ctx._check(ctx._lib.agp_fit_create_batch(ctx._h, count, kernels, fptrs, _ptr(Y), n, _ptr(V) if have_var else None, n, out, None,
                                                 0, None, status), "agp_fit_create_batch")
    finally:
        for kh in handles:
            ctx._lib.agp_kernel_destroy(kh)
    fits = [GPFit(ctx, C.c_void_p(out[b]), n, datasets[b].features) for b in range(count)]
    for b in range(count):
        if status[b] != capi.AGP_OK:
            pivot = ctx._lib.agp_fit_failed_pivot(fits[b]._h)
            ctx._check(status[b], f"agp_fit_create_batch: problem {b} (pivot {pivot})")
    return [FitModel(m, f) for m, f in zip(models, fits)]


def gp_from_covariance(covariance_function, model_name="gaussian_process_regression", context=None):
    """gp.hpp:507-521"""
    return GaussianProcessRegression(covariance_function, None, model_name, context)


def gp_from_covariance_and_mean(covariance_function, mean_function, model_name="gaussian_process_regression",
                                context=None):
    """gp.hpp:523-537"""
    return GaussianProcessRegression(covariance_function, mean_function, model_name, context)
