"""ctypes wrapper of oracle/liboracle.so — TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg;
never by anything under albatross_amd/.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from albatross_amd._capi import Features, KernelNode

_HERE = os.path.dirname(os.path.abspath(__file__))
_lib = None


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build()
        L = C.CDLL(path)
        PF = C.POINTER(Features)
        PN = C.POINTER(KernelNode)
        V = C.c_void_p
        I64 = C.c_int64
        L.orc_eval.restype = C.c_double
        L.orc_eval.argtypes = [PN, C.c_int, PF, I64, PF, I64]
        L.orc_gram_cross.argtypes = [PN, C.c_int, PF, PF, V, I64]
        L.orc_gram_sym.argtypes = [PN, C.c_int, PF, V, I64]
        L.orc_gram_sym_pooled.argtypes = [PN, C.c_int, PF, V, I64, C.c_int]
        L.orc_gram_cross_pooled.argtypes = [PN, C.c_int, PF, PF, V, I64, C.c_int]
        L.orc_ldlt.restype = C.c_int
        L.orc_ldlt.argtypes = [V, I64, I64, V]
        L.orc_ldlt_solve.argtypes = [V, I64, I64, V, V, I64, I64]
        L.orc_ldlt_sqrt_solve.argtypes = [V, I64, I64, V, V, I64, I64]
        L.orc_ldlt_logdet.restype = C.c_double
        L.orc_ldlt_logdet.argtypes = [V, I64, I64]
        L.orc_llt.restype = I64
        L.orc_llt.argtypes = [V, I64, I64]
        L.orc_llt_blocked.restype = I64
        L.orc_llt_blocked.argtypes = [V, I64, I64, C.c_int]
        L.orc_llt_solve.argtypes = [V, I64, I64, V, I64, I64]
        L.orc_llt_logdet.restype = C.c_double
        L.orc_llt_logdet.argtypes = [V, I64, I64]
        L.orc_fit_create.restype = V
        L.orc_fit_create.argtypes = [PN, C.c_int, PF, V, V, C.c_int, C.c_int, C.POINTER(C.c_int)]
        L.orc_fit_destroy.argtypes = [V]
        L.orc_fit_information.argtypes = [V, V]
        L.orc_fit_update.restype = C.c_void_p
        L.orc_fit_update.argtypes = [V, PN, C.c_int, PF, V, V]
        L.orc_fit_logdet.restype = C.c_double
        L.orc_fit_logdet.argtypes = [V]
        L.orc_fit_solve.argtypes = [V, V, I64]
        L.orc_fit_inverse_diagonal.argtypes = [V, V]
        L.orc_fit_loo_marginal.argtypes = [V, V, V, V]
        L.orc_fit_inverse_blocks.argtypes = [V, I64, V, V, V]
        L.orc_fit_held_out.restype = C.c_int
        L.orc_fit_held_out.argtypes = [V, V, I64, V, V, V, V, V]
        L.orc_sparse_fit_create.restype = V
        L.orc_sparse_fit_create.argtypes = [PN, C.c_int, PF, V, V, V, PF, C.c_double, C.c_double]
        L.orc_sparse_fit_from_prediction.restype = V
        L.orc_sparse_fit_from_prediction.argtypes = [PN, C.c_int, PF, V, V]
        L.orc_sparse_fit_update.restype = V
        L.orc_sparse_fit_update.argtypes = [V, PN, C.c_int, PF, V, V, V, C.c_double, C.c_double]
        L.orc_sparse_fit_destroy.argtypes = [V]
        L.orc_sparse_fit_information.argtypes = [V, V]
        L.orc_sparse_fit_rank.restype = I64
        L.orc_sparse_fit_rank.argtypes = [V]
        L.orc_sparse_fit_nll.restype = C.c_double
        L.orc_sparse_fit_nll.argtypes = [V]
        L.orc_sparse_predict.argtypes = [V, PN, C.c_int, PF, V, V, V]
        L.orc_nll_dense.restype = C.c_double
        L.orc_nll_dense.argtypes = [V, V, I64, I64]
        L.orc_nll.restype = C.c_double
        L.orc_nll.argtypes = [PN, C.c_int, PF, V]
        L.orc_nll_with_variance.restype = C.c_double
        L.orc_nll_with_variance.argtypes = [PN, C.c_int, PF, V, V]
        L.orc_mean_vector.restype = None
        L.orc_mean_vector.argtypes = [V, C.c_int, PF, V]
        L.orc_mean_apply.restype = None
        L.orc_mean_apply.argtypes = [V, C.c_int, PF, C.c_double, V]
        L.orc_predict_mean.argtypes = [V, PN, C.c_int, PF, V]
        L.orc_predict_marginal.argtypes = [V, PN, C.c_int, PF, V, V]
        L.orc_predict_joint.argtypes = [V, PN, C.c_int, PF, V, V]
        _lib = L
    return _lib


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _prog(cov):
    nodes = cov.program_nodes()
    arr = (KernelNode * len(nodes))(*nodes)
    return arr, len(nodes)


def _feat(cov, x, is_measurement=False):
    """host FeatureSet -> (Features struct, keepalive)"""
    fs = cov.features(x, is_measurement=is_measurement)
    return fs.as_struct(), fs


def eval_pair(cov, x, i, y, j, x_meas=False, y_meas=False):
    p, n = _prog(cov)
    fx, kx = _feat(cov, x, x_meas)
    fy, ky = _feat(cov, y, y_meas)
    return lib().orc_eval(p, n, C.byref(fx), i, C.byref(fy), j)


def gram(cov, x, y=None, x_meas=False, y_meas=False, threads=0):
    p, n = _prog(cov)
    fx, kx = _feat(cov, x, x_meas)
    if y is None:
        out = np.zeros((fx.n, fx.n), order="F")
        if threads > 1:
            lib().orc_gram_sym_pooled(p, n, C.byref(fx), _ptr(out), fx.n, threads)
        else:
            lib().orc_gram_sym(p, n, C.byref(fx), _ptr(out), fx.n)
        return out
    fy, ky = _feat(cov, y, y_meas)
    out = np.zeros((fx.n, fy.n), order="F")
    if threads > 1:
        lib().orc_gram_cross_pooled(p, n, C.byref(fx), C.byref(fy), _ptr(out), fx.n, threads)
    else:
        lib().orc_gram_cross(p, n, C.byref(fx), C.byref(fy), _ptr(out), fx.n)
    return out


def ldlt(A):
    A = np.array(A, dtype=np.float64, order="F")
    n = A.shape[0]
    tr = np.zeros(n, dtype=np.int64)
    ok = lib().orc_ldlt(_ptr(A), n, n, _ptr(tr))
    return A, tr, bool(ok)


def ldlt_solve(packed, tr, B):
    B = np.array(B, dtype=np.float64, order="F")
    B2 = np.asfortranarray(B.reshape(B.shape[0], -1, order="F"))
    n = packed.shape[0]
    lib().orc_ldlt_solve(_ptr(packed), n, n, _ptr(tr), _ptr(B2), B2.shape[1], n)
    return B2.reshape(B.shape, order="F")


def ldlt_sqrt_solve(packed, tr, B):
    """SerializableLDLT::sqrt_solve (serializable_ldlt.hpp:99-109): D^-1/2 L^-1 P B."""
    B = np.array(B, dtype=np.float64, order="F")
    B2 = np.asfortranarray(B.reshape(B.shape[0], -1, order="F"))
    n = packed.shape[0]
    lib().orc_ldlt_sqrt_solve(_ptr(packed), n, n, _ptr(tr), _ptr(B2), B2.shape[1], n)
    return B2.reshape(B.shape, order="F")


def ldlt_logdet(packed):
    return lib().orc_ldlt_logdet(_ptr(packed), packed.shape[0], packed.shape[0])


def llt(A):
    A = np.array(A, dtype=np.float64, order="F")
    n = A.shape[0]
    info = lib().orc_llt(_ptr(A), n, n)
    return A, int(info)


def llt_blocked(A, threads):
    """strong_llt.c: the blocked, pthread-parallel LL^T of bench.py's `strong_cpu` context row (NOT the reference's
    algorithm); the lower triangle of the result is the factor, the upper one is scratch."""
    A = np.array(A, dtype=np.float64, order="F")
    n = A.shape[0]
    info = lib().orc_llt_blocked(_ptr(A), n, n, int(threads))
    if info:
        raise FloatingPointError(f"pivot {info - 1} is not positive")
    return A


def llt_solve(L, B):
    B = np.array(B, dtype=np.float64, order="F")
    B2 = np.asfortranarray(B.reshape(B.shape[0], -1, order="F"))
    n = L.shape[0]
    lib().orc_llt_solve(_ptr(L), n, n, _ptr(B2), B2.shape[1], n)
    return B2.reshape(B.shape, order="F")


def llt_logdet(L):
    return lib().orc_llt_logdet(_ptr(L), L.shape[0], L.shape[0])


def nll_dense(dev, cov):
    dev = np.ascontiguousarray(dev, dtype=np.float64)
    cov = np.array(cov, dtype=np.float64, order="F")
    return lib().orc_nll_dense(_ptr(dev), _ptr(cov), dev.shape[0], cov.shape[0])


class OracleFit:
    """Fit<GPFit<SerializableLDLT, F>> restated on the CPU (gp.hpp:43-77)."""

    def __init__(self, cov, x, y, y_var=None, threads=0, use_llt=False, mean=None):
        """mean: a mean function (object with nodes() or a postfix node list, see mean_program): removed from
        the targets before the fit (_fit_impl, gp.hpp:291-292) and added back to every predicted mean
        (_predict_impl, gp.hpp:322,346,364)."""
        self.cov = cov
        self.mean = mean
        self._p, self._n = _prog(cov)
        fx, self._keep = _feat(cov, x, False)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if mean is not None:
            y = remove_mean(mean, cov, x, y)
        yv = None if y_var is None else np.ascontiguousarray(y_var, dtype=np.float64)
        st = C.c_int(0)
        self.h = lib().orc_fit_create(self._p, self._n, C.byref(fx), _ptr(y), _ptr(yv), threads,
                                      1 if use_llt else 0, C.byref(st))
        self.status = st.value
        self.n = int(fx.n)
        if not self.h:
            raise FloatingPointError(f"oracle fit failed with status {self.status}")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_fit_destroy(self.h)
            self.h = None

    def update(self, x, y, y_var=None):
        """FitModel::update -> _update_impl (gp.hpp:384-414): a new OracleFit whose solver is the BlockSymmetric of this
        fit (which it keeps alive), Ai_B and the pivoted LDL^T of the Schur complement.  y: raw targets (the mean
        function, if any, is removed here like ModelBase::update does)."""
        fx, keep = _feat(self.cov, x, False)
        y = np.ascontiguousarray(y, dtype=np.float64)
        if self.mean is not None:
            y = remove_mean(self.mean, self.cov, x, y)
        yv = None if y_var is None else np.ascontiguousarray(y_var, dtype=np.float64)
        new = OracleFit.__new__(OracleFit)
        new.cov, new.mean, new._p, new._n = self.cov, self.mean, self._p, self._n
        new._base = self  # the C side keeps a pointer to this fit
        new._keep = keep
        new.h = lib().orc_fit_update(self.h, self._p, self._n, C.byref(fx), _ptr(y), _ptr(yv))
        new.n = self.n + int(fx.n)
        new.status = 0
        return new

    @property
    def information(self):
        out = np.zeros(self.n)
        lib().orc_fit_information(self.h, _ptr(out))
        return out

    @property
    def log_determinant(self):
        return lib().orc_fit_logdet(self.h)

    def solve(self, B):
        B = np.array(B, dtype=np.float64, order="F")
        B2 = np.asfortranarray(B.reshape(B.shape[0], -1, order="F"))
        lib().orc_fit_solve(self.h, _ptr(B2), B2.shape[1])
        return B2.reshape(B.shape, order="F")

    def inverse_diagonal(self):
        out = np.zeros(self.n)
        lib().orc_fit_inverse_diagonal(self.h, _ptr(out))
        return out

    def loo_marginal(self, y):
        y = np.ascontiguousarray(y, dtype=np.float64)
        mean, var = np.zeros(self.n), np.zeros(self.n)
        lib().orc_fit_loo_marginal(self.h, _ptr(y), _ptr(mean), _ptr(var))
        return mean, var

    @staticmethod
    def _groups(groups):
        offsets = np.zeros(len(groups) + 1, dtype=np.int64)
        offsets[1:] = np.cumsum([len(g) for g in groups])
        indices = np.ascontiguousarray(np.concatenate([np.asarray(g, dtype=np.int64) for g in groups])
                                       if len(groups) else np.zeros(0, dtype=np.int64))
        return offsets, indices

    def inverse_blocks(self, groups):
        """SerializableLDLT::inverse_blocks: list of (K^-1)[I_g, I_g]."""
        offsets, indices = self._groups(groups)
        out = np.zeros(int(sum(len(g) ** 2 for g in groups)))
        lib().orc_fit_inverse_blocks(self.h, len(groups), _ptr(offsets), _ptr(indices), _ptr(out))
        blocks, pos = [], 0
        for g in groups:
            m = len(g)
            blocks.append(out[pos:pos + m * m].reshape(m, m, order="F").copy())
            pos += m * m
        return blocks

    def held_out(self, y, groups, joint=False):
        """held_out_predictions: per group (mean, variance[, joint covariance])."""
        y = np.ascontiguousarray(y, dtype=np.float64)
        offsets, indices = self._groups(groups)
        total = int(offsets[-1])
        mean, var = np.zeros(total), np.zeros(total)
        jb = np.zeros(int(sum(len(g) ** 2 for g in groups))) if joint else None
        ok = lib().orc_fit_held_out(self.h, _ptr(y), len(groups), _ptr(offsets), _ptr(indices), _ptr(mean),
                                    _ptr(var), _ptr(jb))
        assert ok
        out, pos = [], 0
        for gi, g in enumerate(groups):
            m, o = len(g), int(offsets[gi])
            item = [mean[o:o + m].copy(), var[o:o + m].copy()]
            if joint:
                item.append(jb[pos:pos + m * m].reshape(m, m, order="F").copy())
                pos += m * m
            out.append(tuple(item))
        return out

    def predict_mean(self, xs, xs_meas=False):
        f, keep = _feat(self.cov, xs, xs_meas)
        mean = np.zeros(f.n)
        lib().orc_predict_mean(self.h, self._p, self._n, C.byref(f), _ptr(mean))
        return self._add_mean(xs, mean)

    def predict_marginal(self, xs, xs_meas=False):
        f, keep = _feat(self.cov, xs, xs_meas)
        mean = np.zeros(f.n)
        var = np.zeros(f.n)
        lib().orc_predict_marginal(self.h, self._p, self._n, C.byref(f), _ptr(mean), _ptr(var))
        return self._add_mean(xs, mean), var

    def predict_joint(self, xs, xs_meas=False):
        f, keep = _feat(self.cov, xs, xs_meas)
        mean = np.zeros(f.n)
        cov = np.zeros((f.n, f.n), order="F")
        lib().orc_predict_joint(self.h, self._p, self._n, C.byref(f), _ptr(mean), _ptr(cov))
        return self._add_mean(xs, mean), cov

    def _add_mean(self, xs, mean):
        return mean if self.mean is None else add_mean(self.mean, self.cov, xs, mean)


def nll(cov, x, y, mean=None):
    """-GaussianProcessBase::log_likelihood(dataset) without priors (gp.hpp:442-451): the mean function is
    removed from y, the covariance is covariance_function_(as_measurements(x)) alone (no target variance)."""
    p, n = _prog(cov)
    fx, keep = _feat(cov, x, False)
    y = np.array(y, dtype=np.float64)
    if mean is not None:
        y = remove_mean(mean, cov, x, y)
    return lib().orc_nll(p, n, C.byref(fx), _ptr(y))


def nll_with_variance(cov, x, y, y_var):
    """negative_log_likelihood(y, k(x, x) + diag(y_var)) on measurement-wrapped features: the checker of the
    C-ABI's agp_nll when a variance is passed (no reference entry point adds it)."""
    p, n = _prog(cov)
    fx, keep = _feat(cov, x, False)
    y = np.ascontiguousarray(y, dtype=np.float64)
    yv = None if y_var is None else np.ascontiguousarray(y_var, dtype=np.float64)
    return lib().orc_nll_with_variance(p, n, C.byref(fx), _ptr(y), _ptr(yv))


# ---- mean functions (mean_function.hpp; LinearMean polynomials.hpp:92-106) ----
_MEAN_NODE = np.dtype([("op", np.int32), ("pad", np.int32), ("params", np.float64, 2)])


def mean_program(nodes):
    """nodes: postfix list of ("zero",) | ("linear", slope, offset) | ("constant", v) | ("sum",) | ("product",)"""
    ops = {"zero": 0, "linear": 1, "constant": 2, "sum": 10, "product": 11}
    arr = np.zeros(len(nodes), dtype=_MEAN_NODE)
    for i, nd in enumerate(nodes):
        arr[i]["op"] = ops[nd[0]]
        for j, v in enumerate(nd[1:]):
            arr[i]["params"][j] = float(v)
    return arr


def _mean_nodes(mean):
    return mean if isinstance(mean, (list, tuple)) and mean and isinstance(mean[0], tuple) else mean.nodes()


def mean_vector(mean, cov, x):
    arr = mean_program(_mean_nodes(mean))
    fx, keep = _feat(cov, x, False)
    out = np.zeros(fx.n)
    lib().orc_mean_vector(_ptr(arr), len(arr), C.byref(fx), _ptr(out))
    return out


def remove_mean(mean, cov, x, y):
    arr = mean_program(_mean_nodes(mean))
    fx, keep = _feat(cov, x, False)
    out = np.array(y, dtype=np.float64)
    lib().orc_mean_apply(_ptr(arr), len(arr), C.byref(fx), -1.0, _ptr(out))
    return out


def add_mean(mean, cov, x, y):
    arr = mean_program(_mean_nodes(mean))
    fx, keep = _feat(cov, x, False)
    out = np.array(y, dtype=np.float64)
    lib().orc_mean_apply(_ptr(arr), len(arr), C.byref(fx), 1.0, _ptr(out))
    return out


class OracleSparseFit:
    """Fit<SparseGPFit> of SparseGaussianProcessRegression::_fit_impl (models/sparse_gp.hpp:354-381) with the
    DenseQRImplementation, restated on the CPU.  group_keys[i] = grouper(features[i]); y are the RAW
    target means (the reference copies y before it removes the mean function, :664-668)."""

    def __init__(self, cov, x, group_keys, y, y_var, u, measurement_nugget=1e-8, inducing_nugget=1e-8):
        self.cov = cov
        self._p, self._n = _prog(cov)
        fx, self._kx = _feat(cov, x, False)
        fu, self._ku = _feat(cov, u, False)
        keys = np.ascontiguousarray(group_keys, dtype=np.int64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        yv = None if y_var is None else np.ascontiguousarray(y_var, dtype=np.float64)
        self.m = int(fu.n)
        self.h = lib().orc_sparse_fit_create(self._p, self._n, C.byref(fx), _ptr(keys), _ptr(y), _ptr(yv),
                                             C.byref(fu), measurement_nugget, inducing_nugget)

    def update(self, x, group_keys, y, y_var, measurement_nugget=1e-8, inducing_nugget=1e-8):
        """_update_impl (:322-371): a new OracleSparseFit with the further observations folded in."""
        fx, kx = _feat(self.cov, x, False)
        keys = np.ascontiguousarray(group_keys, dtype=np.int64)
        y = np.ascontiguousarray(y, dtype=np.float64)
        yv = None if y_var is None else np.ascontiguousarray(y_var, dtype=np.float64)
        new = OracleSparseFit.__new__(OracleSparseFit)
        new.cov, new._p, new._n, new.m = self.cov, self._p, self._n, self.m
        new.h = lib().orc_sparse_fit_update(self.h, self._p, self._n, C.byref(fx), _ptr(keys), _ptr(y), _ptr(yv),
                                            measurement_nugget, inducing_nugget)
        return new

    @classmethod
    def from_prediction(cls, cov, z, mean, covariance):
        """fit_from_prediction (sparse_gp.hpp:406-461): the fit on the inducing points z that reproduces a joint
        prediction (mean, covariance) made AT z."""
        new = cls.__new__(cls)
        new.cov = cov
        new._p, new._n = _prog(cov)
        fz, new._ku = _feat(cov, z, False)
        new.m = int(fz.n)
        mean = np.ascontiguousarray(mean, dtype=np.float64)
        covariance = np.asfortranarray(covariance, dtype=np.float64)
        assert mean.shape == (new.m,) and covariance.shape == (new.m, new.m)
        new.h = lib().orc_sparse_fit_from_prediction(new._p, new._n, C.byref(fz), _ptr(mean), _ptr(covariance))
        return new

    def rebase(self, z):
        """rebase_inducing_points (sparse_gp.hpp:714-725): fit_from_prediction(z, predict(z).joint())."""
        mean, _, covariance = self.predict(z, joint=True)
        return OracleSparseFit.from_prediction(self.cov, z, mean, covariance)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_sparse_fit_destroy(self.h)
            self.h = None

    @property
    def information(self):
        out = np.zeros(self.m)
        lib().orc_sparse_fit_information(self.h, _ptr(out))
        return out

    @property
    def numerical_rank(self):
        return int(lib().orc_sparse_fit_rank(self.h))

    @property
    def nll(self):
        return lib().orc_sparse_fit_nll(self.h)

    def predict(self, xs, xs_meas=False, joint=False):
        f, keep = _feat(self.cov, xs, xs_meas)
        mean, var = np.zeros(f.n), np.zeros(f.n)
        cov = np.zeros((f.n, f.n), order="F") if joint else None
        lib().orc_sparse_predict(self.h, self._p, self._n, C.byref(f), _ptr(mean), _ptr(var), _ptr(cov))
        return (mean, var, cov) if joint else (mean, var)
