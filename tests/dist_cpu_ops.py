"""TEST-ONLY numpy implementation of the block arithmetic (`agp_shard_ops_callbacks`, csrc/shard_internal.h) that the
library's C++ sharded-fit schedule (albatross_amd/csrc/shard_sched.hip) is written against, so that the schedule -
ownership, broadcasts, all-gathers, look-ahead, both substitutions - runs on CPU-only machines over gloo with world
sizes > 1.  The product uses HipShardOps (the HIP kernels) instead; nothing under albatross_amd/ imports this file."""
import ctypes as C

import numpy as np
import scipy.linalg

from albatross_amd import _capi as capi


def _mat(ptr, ld, rows, cols):
    """column-major view of rows x cols doubles at `ptr` with leading dimension ld"""
    rows, cols, ld = int(rows), int(cols), int(ld)
    if rows <= 0 or cols <= 0:
        return np.zeros((max(rows, 0), max(cols, 0)))
    flat = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(ld * (cols - 1) + rows,))
    return np.lib.stride_tricks.as_strided(flat, shape=(rows, cols), strides=(8, 8 * ld))


def _vec(ptr, n):
    return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(int(n),))


class NumpyShardOps:
    def __init__(self):
        self.calls = {"factor_diag": 0, "trsm_rows": 0, "gemm": 0}

        def factor_diag(user, D, ld, w, img, zblk, logsum):
            self.calls["factor_diag"] += 1
            A = _mat(D, ld, w, w)
            L = np.tril(A).copy()
            bad = 0
            # unblocked LL^T so that the FIRST non-positive pivot is known (np.linalg.cholesky only raises)
            for j in range(w):
                d = L[j, j] - L[j, :j] @ L[j, :j]
                if not d > 0. and bad == 0:
                    bad = j + 1
                L[j, j] = np.sqrt(d) if d > 0. else np.nan
                L[j + 1:, j] = (L[j + 1:, j] - L[j + 1:, :j] @ L[j, :j]) / L[j, j]
            A[np.tril_indices(w)] = L[np.tril_indices(w)]
            z = _vec(zblk, w)
            z[:] = scipy.linalg.solve_triangular(L, z, lower=True, check_finite=False) if bad == 0 else np.nan
            logsum[0] = float(np.sum(np.log(np.diag(L)))) if bad == 0 else float("nan")
            return bad

        def trsm_rows(user, X, ld, nrows, w, Lkk, img, z, yrows):
            self.calls["trsm_rows"] += 1
            Xv = _mat(X, ld, nrows, w)
            L = np.tril(_mat(Lkk, w, w, w))
            Xv[:] = scipy.linalg.solve_triangular(L, Xv.T, lower=True, check_finite=False).T
            y = _vec(yrows, nrows)
            y -= Xv @ _vec(z, w)

        def gemm(user, Cp, ldc, P, ldp, Q, ldq, M, N, K, tri):
            self.calls["gemm"] += 1
            Cv = _mat(Cp, ldc, M, N)
            upd = _mat(P, ldp, M, K) @ _mat(Q, ldq, N, K).T
            if tri:  # only the entries on / below the diagonal are required: leave the rest untouched on purpose
                upd = np.tril(upd)
            Cv -= upd

        def copy2d(user, dst, ldd, src, lds, rows, cols):
            _mat(dst, ldd, rows, cols)[:] = _mat(src, lds, rows, cols)

        def invert_diag(user, D, ld, w, img, W):
            L = np.tril(_mat(D, ld, w, w))
            _mat(W, w, w, w)[:] = scipy.linalg.solve_triangular(L, np.eye(w), lower=True, check_finite=False)

        def colvec_dot(user, W, ld, m, n, v, alpha, beta, base, out):
            r = alpha * (_mat(W, ld, m, n).T @ _vec(v, m))
            if base:
                r = r + beta * _vec(base, n)
            _vec(out, n)[:] = r

        def axpby(user, n, a, x, b, y, out):
            _vec(out, n)[:] = a * _vec(x, n) + b * _vec(y, n)

        def fill_zero(user, p, count):
            _vec(p, count)[:] = 0.

        self._fns = (capi.FACTOR_DIAG_FN(factor_diag), capi.TRSM_ROWS_FN(trsm_rows), capi.GEMM_FN(gemm),
                     capi.COPY2D_FN(copy2d), capi.INVERT_DIAG_FN(invert_diag), capi.COLVEC_DOT_FN(colvec_dot),
                     capi.AXPBY_FN(axpby), capi.FILL_ZERO_FN(fill_zero))
        self.struct = capi.ShardOpsCallbacks(None, *self._fns)


def sharded_factor_numpy(K_lower_full, y, block, comm):
    """Run the library's schedule on this rank's rows of the dense symmetric matrix K (only its lower triangle is
    used).  comm: albatross_amd.distributed.Communicator or None.  Returns (status, information, log_det, bad_pivot,
    ops.calls)."""
    from albatross_amd.distributed import ShardLayout
    lib = capi.load_debug()  # the schedule over caller-supplied block arithmetic is a TEST entry point (csrc/debug_api.hip)
    n = K_lower_full.shape[0]
    world = 1 if comm is None else comm.world
    rank = 0 if comm is None else comm.rank
    lay = ShardLayout(n, world, block)
    rows = lay.global_rows(rank)
    n_loc = len(rows)
    assert n_loc == lay.local_rows(rank)
    ld = max(n_loc, 1) + 3  # an odd padding on purpose
    A = np.full((ld, n), np.nan, order="F")
    for l, g in enumerate(rows):  # the staircase: row g holds columns 0 .. end of its own diagonal block
        end = min(n, (g // block + 1) * block)
        A[l, :g + 1] = K_lower_full[g, :g + 1]
        A[l, g + 1:end] = K_lower_full[g + 1:end, g]  # the upper part of the diagonal block may be touched (symmetric)
    yl = np.ascontiguousarray(y[rows], dtype=np.float64) if n_loc else np.zeros(1)
    work = np.full(lib.agp_debug_shard_work_doubles(n, block, world, rank), np.nan)
    info = np.full(n, np.nan)
    logdet, bad = C.c_double(), C.c_int64(-1)
    ops = NumpyShardOps()
    st = lib.agp_debug_shard_factor_custom(C.byref(ops.struct), None if comm is None else comm._h, n, block,
                                     C.c_void_p(A.ctypes.data), ld, C.c_void_p(yl.ctypes.data), C.c_void_p(work.ctypes.data),
                                     C.c_void_p(info.ctypes.data), C.byref(logdet), C.byref(bad))
    return st, info, logdet.value, bad.value, ops.calls
