"""GPU tests of the pivoted L D L^T (Eigen::LDLT as SerializableLDLT wraps it): bit-exact against the
oracle's restatement of the unblocked algorithm, semi-definite inputs, the solve with D^+ and the
properties tests/test_serializable_ldlt.cc:34-85 pins."""
import numpy as np
import pytest

import albatross_amd as ab
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def spd(n, seed):
    rng = np.random.default_rng(seed)
    G = rng.standard_normal((n, n + 3))
    return G @ G.T / n + np.eye(n)


@pytest.mark.parametrize("n", [1, 2, 7, 63, 64, 65, 97, 200, 513, 1100])
def test_factor_is_bit_identical_to_oracle(ctx, n):
    sc = np.random.default_rng(n).uniform(0.1, 10., n)
    A = spd(n, n) * sc[:, None] * sc[None, :]  # widely varying diagonal: plenty of pivoting
    f = ab.PivotedLDLT(A, ctx)
    packed, tr, ok = orc.ldlt(A)
    assert f.success == ok
    assert np.array_equal(f.transpositions(), tr)
    got = f.matrix_ldlt()
    low = np.tril_indices(n)
    assert np.array_equal(got[low], packed[low])          # L (strict lower) and D (diagonal): same bits
    assert np.array_equal(f.vector_d(), np.diag(packed))
    B = np.random.default_rng(n + 1).standard_normal((n, 5))
    X = orc.ldlt_solve(packed, tr, B)
    assert np.abs(f.solve(B) - X).max() <= 1e-12 * max(1., np.abs(X).max())
    assert abs(f.log_determinant - orc.ldlt_logdet(packed)) <= 1e-12 * n
    # sqrt_solve (serializable_ldlt.hpp:99-109; tests/test_serializable_ldlt.cc pins S^T S = B^T A^-1 B at 1e-14)
    S, So = f.sqrt_solve(B), orc.ldlt_sqrt_solve(packed, tr, B)
    assert np.abs(S - So).max() <= 1e-13 * max(1., np.abs(So).max())
    assert np.abs(S.T @ S - B.T @ X).max() <= 1e-12 * max(1., np.abs(B.T @ X).max())


def test_semi_definite_matrix(ctx):
    """A rank-deficient covariance (tests/test_gp.cc:20-33 'unobservable' style): LL^T refuses, the pivoted
    factor goes through and its solve is the reference's (pseudo-inverse on the exactly-zero pivots)."""
    rng = np.random.default_rng(5)
    n, r = 120, 37
    G = rng.standard_normal((n, r))
    A = G @ G.T
    with pytest.raises(ab.NotPositiveDefiniteError):
        ab.DenseFactor(A, ctx)
    f = ab.PivotedLDLT(A, ctx)
    packed, tr, ok = orc.ldlt(A)
    assert np.array_equal(f.transpositions(), tr) and np.array_equal(f.vector_d(), np.diag(packed))
    assert np.array_equal(np.tril(f.matrix_ldlt()), np.tril(packed))
    # the trailing pivots are rounding noise (1e-13 .. 1e-16, not exact zeros): like Eigen's, the solve keeps
    # them, so only the residual of a consistent system is meaningful, not x itself
    b = A @ rng.standard_normal(n)
    assert np.abs(A @ f.solve(b) - b).max() <= 1e-6 * np.abs(b).max()
    # exactly singular: duplicated rows / columns give exact zero pivots and D^+ drops them
    A2 = np.kron(np.array([[1., 1.], [1., 1.]]), spd(20, 3))
    f2 = ab.PivotedLDLT(A2, ctx)
    p2, t2, ok2 = orc.ldlt(A2)
    assert f2.success == ok2 and np.array_equal(f2.vector_d(), np.diag(p2))
    assert np.sum(f2.vector_d() == 0.) >= 1
    b2 = A2 @ np.ones(40)
    assert np.abs(f2.solve(b2) - orc.ldlt_solve(p2, t2, b2)).max() <= 1e-10
    assert np.abs(A2 @ f2.solve(b2) - b2).max() <= 1e-8 * np.abs(b2).max()


def test_serializable_ldlt_properties(ctx):
    """tests/test_serializable_ldlt.cc:40-85 on the pivoted device factor: solve(A) = I, log determinant,
    row-major input, and the zero matrix (k == 0 zero pivot branch)."""
    n = 90
    A = spd(n, 11)
    f = ab.PivotedLDLT(np.ascontiguousarray(A), ctx)  # C-ordered input: uplo = 1 path
    assert np.abs(f.solve(A) - np.eye(n)).max() <= 1e-10
    assert abs(f.log_determinant - np.linalg.slogdet(A)[1]) <= 1e-9 * n
    z = ab.PivotedLDLT(np.zeros((6, 6)), ctx)
    assert z.success and np.array_equal(z.transpositions(), np.arange(6)) and np.all(z.solve(np.ones(6)) == 0.)
