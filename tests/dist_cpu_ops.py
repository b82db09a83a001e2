"""TEST-ONLY numpy / oracle implementation of the block interface that
albatross_amd/distributed.py sequences, so that the sharded-fit schedule
(ownership, panel broadcasts, running y, back substitution) can run on CPU
tensors over gloo.  The product's HipBlockOps calls the HIP library instead."""
import numpy as np
import scipy.linalg
import torch

from oracle import oracle_py as orc


class NumpyBlockOps:
    def empty(self, count):
        return torch.full((int(count),), float("nan"), dtype=torch.float64)

    def from_host(self, array):
        return torch.from_numpy(np.array(array, dtype=np.float64))

    def to_host(self, tensor):
        return tensor.detach().clone().numpy()

    def sync(self):
        pass

    @staticmethod
    def _mat(t, offset, ld, rows, cols):
        return np.lib.stride_tricks.as_strided(t.numpy()[offset:], shape=(rows, cols), strides=(8, 8 * ld))

    def pack_panel(self, col, lda, m, width, buf, ldp):
        if m - width > 0:
            self._mat(buf, 0, ldp, m - width, width)[:] = self._mat(col, width, lda, m - width, width)

    def gram_block(self, cov, rows_fs, cols_fs, out, ld, diag_add, diag_offset):
        K = orc.gram(cov, rows_fs, cols_fs, x_meas=True, y_meas=True)
        if diag_add is not None:
            w = K.shape[1]
            K[np.arange(w), np.arange(w)] += diag_add.numpy()[diag_offset:diag_offset + w]
        view = self._mat(out, 0, ld, K.shape[0], K.shape[1])
        view[:] = np.tril(K) + np.triu(np.full(K.shape, np.nan), 1)  # the strict upper part is never read
        return int(np.isnan(np.tril(K)).any())

    def panel_factor(self, col, m, lda, width, img, y):
        A = self._mat(col, 0, lda, m, width)
        D = np.tril(A[:width]) + np.tril(A[:width], -1).T
        L, info = orc.llt(D)
        if info:
            return info - 1, 0.
        L = np.tril(L)
        A[:width] = L + np.triu(np.full((width, width), np.nan), 1)
        if m > width:
            A[width:] = scipy.linalg.solve_triangular(L, A[width:].T, lower=True).T
        yv = y.numpy()
        yv[:width] = scipy.linalg.solve_triangular(L, yv[:width], lower=True)
        if m > width:
            yv[width:m] -= A[width:] @ yv[:width]
        return -1, float(np.log(np.diag(L)).sum())

    def update(self, col, ldc, buf, row_offset, ldp, M, N, K):
        P = self._mat(buf, row_offset, ldp, M, K)
        Cm = self._mat(col, 0, ldc, M, N)
        upd = P @ P[:N].T
        mask = np.tril(np.ones((M, N), dtype=bool))
        Cm[mask] -= upd[mask]

    def back_diag(self, col, lda, width, img, z):
        L = np.tril(self._mat(col, 0, lda, width, width))
        zv = z.numpy()
        zv[:width] = scipy.linalg.solve_triangular(L.T, zv[:width], lower=False)

    def back_update(self, col, row_offset, lda, nrows, ncols, x, z):
        Lr = self._mat(col, row_offset, lda, nrows, ncols)
        z.numpy()[:ncols] -= Lr.T @ x.numpy()[:nrows]
