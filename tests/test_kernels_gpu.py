"""GPU tests of the building blocks, through the debug entry points of the
C-ABI library: MFMA lane map, the fp64 MFMA update kernel, the blocked LL^T."""
import ctypes as C

import numpy as np
import pytest

from albatross_amd import _capi as capi
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def _p(a):
    return C.c_void_p(a.ctypes.data)


@pytest.fixture(scope="module")
def dbg(ctx):
    lib = capi.load_debug()
    lib.agp_debug_mfma_tile.restype = C.c_int
    lib.agp_debug_mfma_tile.argtypes = [C.c_void_p] * 4
    lib.agp_debug_gemm.restype = C.c_int
    lib.agp_debug_gemm.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int, C.c_void_p,
                                   C.c_int64, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int]
    lib.agp_debug_factor.restype = C.c_int
    lib.agp_debug_factor.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p,
                                     C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.agp_debug_trailing_update.restype = C.c_int
    lib.agp_debug_trailing_update.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64,
                                              C.c_int64, C.c_int]
    return lib


def test_mfma_f64_lane_map(ctx, dbg):
    """v_mfma_f64_16x16x4_f64 operand / result lane maps assumed in mfma_f64.h,
    checked with exact integer data and an asymmetric B."""
    rng = np.random.default_rng(0)
    A = rng.integers(-8, 9, (16, 4)).astype(np.float64)
    B = rng.integers(-8, 9, (4, 16)).astype(np.float64)
    D = np.zeros((16, 16))
    assert dbg.agp_debug_mfma_tile(ctx._h, _p(A), _p(B), _p(D)) == 0
    assert np.array_equal(D, A @ B)


def test_mfma_peak_is_sane(ctx, dbg):
    out = C.c_double()
    dbg.agp_debug_mfma_f64_peak.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_double)]
    assert dbg.agp_debug_mfma_f64_peak(ctx._h, 4000, C.byref(out)) == 0
    tf = out.value
    print(f"measured fp64 MFMA issue-loop rate: {tf:.1f} TFLOP/s")
    assert 20. < tf < 400.


@pytest.mark.parametrize("M,N,K,tri,akm,bkm", [
    (128, 128, 16, 0, 0, 0), (256, 128, 128, 0, 0, 0), (384, 384, 512, 1, 0, 0),
    (200, 130, 70, 0, 0, 0), (333, 333, 129, 1, 0, 0), (257, 100, 128, 0, 0, 1),
    (140, 90, 50, 0, 1, 1), (300, 64, 128, 0, 1, 0), (130, 130, 1000, 1, 1, 1),
    # few 64-tiles and K >= 256: the 32 x 32-tile kernel (the sharded fit's next-block-column and diagonal-block updates)
    (1536, 512, 512, 0, 0, 0), (512, 512, 512, 1, 0, 0), (100, 70, 300, 0, 0, 0), (333, 333, 257, 1, 0, 0), (33, 31, 256, 0, 0, 0),
    # >= 512 tiles of 128 x 128: the large-tile kernel (smaller launches use 64 x 64 tiles)
    (4500, 4500, 48, 1, 0, 0), (3000, 2900, 40, 0, 0, 0), (70, 70, 300, 1, 0, 0),
    # skinny second dimension with a transposed second operand: the 64 x 64-tile kernel with the k-major loader
    (300, 8, 128, 0, 0, 1), (1000, 64, 512, 0, 0, 1), (130, 1, 70, 0, 0, 1), (4000, 33, 128, 0, 0, 1), (64, 64, 16, 0, 0, 1),
    (300, 8, 128, 0, 1, 1), (1000, 64, 512, 0, 1, 1), (131, 3, 70, 0, 1, 1), (4000, 33, 130, 0, 1, 1),
])
def test_gemm_nt_sub(ctx, dbg, M, N, K, tri, akm, bkm):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    A = rng.standard_normal((M, K))
    B = rng.standard_normal((N, K))
    C0 = rng.standard_normal((M, N))
    ldc = M + 3
    Cd = np.zeros((ldc, N), order="F")
    Cd[:M] = C0
    Ah = np.asfortranarray(A.T if akm else A)   # kmajor: K x M column-major
    Bh = np.asfortranarray(B.T if bkm else B)
    st = dbg.agp_debug_gemm(ctx._h, _p(Cd), ldc, _p(Ah), Ah.shape[0], akm, _p(Bh), Bh.shape[0], bkm, M, N, K, tri)
    assert st == 0
    want = C0 - A @ B.T
    got = Cd[:M]
    scale = np.abs(A) @ np.abs(B.T) + np.abs(C0)
    if tri:
        # tiles strictly above the diagonal are skipped: compare the lower triangle only
        mask = np.tril(np.ones((M, N), dtype=bool))
        err = (np.abs(got - want) / scale)[mask].max()
    else:
        err = (np.abs(got - want) / scale).max()
    assert err < 8 * np.finfo(float).eps * np.sqrt(K)
    assert np.array_equal(Cd[M:], np.zeros((ldc - M, N)))  # padding rows untouched


@pytest.mark.parametrize("M,N,K,tri,akm,bkm", [
    # >= 512 tiles of 128 x 128 AND at least 8 chunks of K: the interior tiles add C into their accumulators during the K
    # loop and only store behind it (gemm_nt_sub_tile_cpf), the edge tiles keep the read-modify-write epilogue; every
    # operand layout; K = 136 is not a whole number of chunks and keeps all tiles on the old path
    (3000, 2900, 160, 0, 0, 0), (3000, 2900, 160, 0, 0, 1), (3000, 2900, 160, 0, 1, 0), (3000, 2900, 160, 0, 1, 1),
    (4500, 4500, 144, 1, 0, 0), (4480, 4480, 256, 1, 1, 1), (3000, 2900, 136, 0, 0, 1),
])
def test_gemm_nt_sub_prefetching_tiles(ctx, dbg, M, N, K, tri, akm, bkm):
    test_gemm_nt_sub(ctx, dbg, M, N, K, tri, akm, bkm)


@pytest.mark.parametrize("variant", [0, 3, 13, 14, 15])
@pytest.mark.parametrize("M,K", [(5900, 128), (6016, 512), (700, 512), (1418, 96)])
def test_trailing_update_large_tiles(ctx, dbg, M, K, variant):
    """Bulk updates of more than two rounds of 128 x 128 tiles: full rounds of prefetching tiles, the tiles of the partial
    round as 64 x 64 quadrants at the head of the same launch, ragged edge tiles (M = 5900) on the old path.  Variant 0:
    fp64 MFMA; 3: fp32 products of operands rounded while staged; 13: fp32 products of an fp32 copy of the panel
    (launch_convert_panel_f32) - the two must agree bit for bit; 14 (round 5): fp32-accurate products on the BF16 pipe from
    three bf16 planes of the panel (gemm_bf16x3.hip: hi + mid + lo, six partial products) - the same 2e-7 bar as the fp32
    kernel (measured: both ~3e-8 of the scale), interior tiles with C prefetched and ragged / shallow ones without; 15
    (round 6): products from two fp16 planes of power-of-two-scaled rows (gemm_f16x2.hip: h1 + h2, three partial products,
    scales from the rows' squared norms - the bound a Cholesky factor's rows obey), same bar."""
    rng = np.random.default_rng(M + K)
    ldc, ldp = M + 8, M + 10
    Cm = np.asfortranarray(rng.standard_normal((ldc, M)))
    P = np.asfortranarray(rng.standard_normal((ldp, K)))
    want = Cm[:M] - P[:M] @ P[:M].T
    got = Cm.copy(order="F")
    assert dbg.agp_debug_trailing_update(ctx._h, _p(got), ldc, _p(P), ldp, M, K, variant) == 0
    low = np.tril_indices(M)
    scale = np.abs(P[:M]).sum(axis=1).max() ** 2
    tol = 1e-14 if variant == 0 else 2e-7
    assert np.abs(got[:M][low] - want[low]).max() <= tol * scale
    assert np.array_equal(got[M:], Cm[M:])  # padding rows untouched
    if variant == 13:
        ref = Cm.copy(order="F")
        assert dbg.agp_debug_trailing_update(ctx._h, _p(ref), ldc, _p(P), ldp, M, K, 3) == 0
        assert np.array_equal(ref[:M][low], got[:M][low])


@pytest.mark.parametrize("variant", [14, 15])
def test_trailing_update_split_planes_badly_scaled_rows(ctx, dbg, variant):
    """Rows of the panel between 1e-9 and 1e+9 times each other (a covariance matrix whose diagonal spans 36 orders of
    magnitude): the error of every entry stays below 2e-7 of ITS OWN row and column scale, sum_k |P[i, k]| sum_k |P[j, k]|.
    bf16 x 3 (14) has fp32's exponent range; fp16 x 2 (15) gets there through the power-of-two row scales - without them
    the large rows would overflow fp16 and the small ones vanish."""
    M, K = 1418, 256
    rng = np.random.default_rng(variant)
    ldc, ldp = M + 8, M + 10
    rowscale = 10. ** rng.uniform(-9., 9., size=ldp)
    P = np.asfortranarray(rng.standard_normal((ldp, K)) * rowscale[:, None])
    # (entries of very different size inside a row too: every fourth column 1e-5 of the others)
    P[:, ::4] *= 1e-5
    Cm = np.asfortranarray(rng.standard_normal((ldc, M)) * np.outer(np.r_[rowscale[:M], np.ones(ldc - M)], rowscale[:M]))
    want = Cm[:M] - P[:M] @ P[:M].T
    got = Cm.copy(order="F")
    assert dbg.agp_debug_trailing_update(ctx._h, _p(got), ldc, _p(P), ldp, M, K, variant) == 0
    rs = np.abs(P[:M]).sum(axis=1)
    rel = np.abs(got[:M] - want) / np.outer(rs, rs)
    assert rel[np.tril_indices(M)].max() <= 2e-7
    assert np.array_equal(got[M:], Cm[M:])


@pytest.mark.parametrize("head_cols", [1, 4, 8])
@pytest.mark.parametrize("M,K", [(9216, 512), (6016, 512), (5900, 128), (1418, 512), (700, 96), (384, 512)])
def test_trailing_update_merged(ctx, dbg, M, K, head_cols):
    """The merged update of factor_lower (round 6; csrc/gemm.hip: GemmArgs::head_cols): the whole trailing matrix in one
    launch on the bulk stream whose first workgroups are the 128 x 128 tiles of the next block column (head_cols tile columns),
    each counting itself when its tile is in memory; the gate kernel on the chain stream ends at the full count.  Same
    tile bodies as the plain bulk launch: within 1e-14 of the scale everywhere; the debug entry point also checks the
    count and that no hand-over timed out.  Sizes: full rounds + remainder quadrants behind the head (9216, 6016), ragged
    last tile row (5900, 1418, 700), a head wider than the matrix (384: 3 tile columns)."""
    rng = np.random.default_rng(M + K + head_cols)
    ldc, ldp = M + 8, M + 10
    Cm = np.asfortranarray(rng.standard_normal((ldc, M)))
    P = np.asfortranarray(rng.standard_normal((ldp, K)))
    want = Cm[:M] - P[:M] @ P[:M].T
    got = Cm.copy(order="F")
    assert dbg.agp_debug_trailing_update(ctx._h, _p(got), ldc, _p(P), ldp, M, K, 20 + head_cols) == 0
    low = np.tril_indices(M)
    scale = np.abs(P[:M]).sum(axis=1).max() ** 2
    assert np.abs(got[:M][low] - want[low]).max() <= 1e-14 * scale
    assert np.array_equal(got[M:], Cm[M:])  # padding rows untouched


@pytest.mark.parametrize("variant", [0, 2, 4, 5])
@pytest.mark.parametrize("M,K", [(256, 16), (640, 128), (1000, 256), (1418, 512), (130, 32), (4300, 64)])
def test_trailing_update_variants(ctx, dbg, M, K, variant):
    """Bulk update C -= P P^T on the lower tiles: MFMA kernel (0), the DPP-broadcast VALU kernel (2), every tile as
    four 64 x 64 workgroups (4), full rounds of 128-tiles + a 64-tile tail (5)."""
    rng = np.random.default_rng(M + K)
    ldc, ldp = M + 8 - (M % 2), M + 10 - (M % 2)
    Cm = np.asfortranarray(rng.standard_normal((ldc, M)))
    P = np.asfortranarray(rng.standard_normal((ldp, K)))
    want = Cm[:M] - P[:M] @ P[:M].T
    got = Cm.copy(order="F")
    assert dbg.agp_debug_trailing_update(ctx._h, _p(got), ldc, _p(P), ldp, M, K, variant) == 0
    low = np.tril_indices(M)
    scale = np.abs(P[:M]).sum(axis=1).max() ** 2
    assert np.abs(got[:M][low] - want[low]).max() <= 1e-14 * scale
    assert np.array_equal(got[M:], Cm[M:])  # padding rows untouched


@pytest.mark.parametrize("n,ncols", [(1024, 8192), (1536, 12289), (1024, 9000)])
def test_forward_solve_wide(ctx, dbg, n, ncols):
    """X = L^-1 B out of place for a right-hand side much wider than L (csrc/solve.hip: forward_solve_wide - inverted 512 x 512
    diagonal blocks, one deep product per block row, in-place triangular products bottom-up): the sparse GP's m x n solves.
    Ragged column counts exercise the edge tiles of the out-of-place kernel."""
    import ctypes as C
    from scipy.linalg import solve_triangular
    rng = np.random.default_rng(n + ncols)
    G = rng.standard_normal((n, n))
    K = np.asfortranarray(G @ G.T / n + np.eye(n))
    B = np.asfortranarray(rng.standard_normal((n, ncols)))
    X = np.zeros((n, ncols), order="F")
    fn = dbg.agp_debug_forward_solve_wide
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    assert fn(ctx._h, _p(K), n, _p(B), ncols, _p(X)) == 0
    want = solve_triangular(np.linalg.cholesky(K), B, lower=True)
    assert np.abs(X - want).max() <= 1e-11 * np.abs(want).max()


@pytest.mark.parametrize("n", [16, 100, 128, 129, 300, 512, 640, 1000, 1537])
def test_factor_matches_oracle_llt(ctx, dbg, n):
    rng = np.random.default_rng(n)
    G = rng.standard_normal((n, n + 5))
    A = G @ G.T / n + np.eye(n)
    y = rng.standard_normal(n)
    lda = n + (n % 2) + 2
    Ad = np.full((lda, n), np.nan, order="F")
    Ad[:n] = np.tril(A) + np.triu(np.full((n, n), 7.5), 1)  # upper triangle must be ignored
    yd = y.copy()
    logdet = C.c_double()
    bad = C.c_int64()
    assert dbg.agp_debug_factor(ctx._h, _p(Ad), n, lda, _p(yd), C.byref(logdet), C.byref(bad)) == 0
    assert bad.value == -1
    L = np.tril(Ad[:n])
    Lo, info = orc.llt(A)
    assert info == 0
    Lo = np.tril(Lo)
    assert np.abs(L - Lo).max() <= 1e-12 * np.abs(Lo).max()
    assert np.abs(L @ L.T - A).max() <= 1e-13 * np.abs(A).max() * n
    z = np.linalg.solve(Lo, y)
    assert np.abs(yd - z).max() <= 1e-11 * np.abs(z).max()
    assert abs(logdet.value - orc.llt_logdet(Lo)) <= 1e-11 * max(1., abs(orc.llt_logdet(Lo)))


def test_factor_reports_first_bad_pivot(ctx, dbg):
    n = 300
    rng = np.random.default_rng(3)
    G = rng.standard_normal((n, n))
    A = G @ G.T / n + np.eye(n)
    A[200, 200] = -1.0  # leading 200x200 block stays PD; pivot 200 goes non-positive
    Ad = np.asfortranarray(A.copy())
    bad = C.c_int64()
    logdet = C.c_double()
    assert dbg.agp_debug_factor(ctx._h, _p(Ad), n, n, None, C.byref(logdet), C.byref(bad)) == 0
    assert bad.value == 200


@pytest.mark.parametrize("n", [129, 700, 1500, 2304, 3400])
def test_panel_step_kernel(make_ctx, dbg, n, monkeypatch):
    """AGP_STEP_BELOW: the chain-bound tail as ONE launch per panel (chol.hip: panel_phase step_mode - the update-ahead panel
    kernel plus trailing-update workgroups in the same launch) against the two-launch tail (AGP_STEP_BELOW=0, and with
    AGP_PANEL_FUSED=0 the round-2 POTRF / TRSM launches) and numpy; sizes with a partial last panel, a partial last
    64-row tile, and a tail that starts in the middle of the matrix.  The switches are read per context."""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    A = np.asfortranarray(B @ B.T + n * np.eye(n))
    y = rng.standard_normal(n)
    out = {}
    for mode in ("0", "2048", "8192", "unfused"):  # 8192: every panel a step launch
        monkeypatch.setenv("AGP_STEP_BELOW", "0" if mode == "unfused" else mode)
        monkeypatch.setenv("AGP_PANEL_FUSED", "0" if mode == "unfused" else "1")
        ctx = make_ctx()
        Ad, yd = A.copy(order="F"), y.copy()
        logdet, bad = C.c_double(), C.c_int64()
        assert dbg.agp_debug_factor(ctx._h, _p(Ad), n, n, _p(yd), C.byref(logdet), C.byref(bad)) == 0
        assert bad.value == -1
        out[mode] = (np.tril(Ad), yd, logdet.value)
    L = np.linalg.cholesky(A)
    for mode in out:
        assert np.abs(out[mode][0] - L).max() <= 1e-11 * np.abs(L).max()
        assert np.abs(out[mode][1] - np.linalg.solve(L, y)).max() <= 1e-10
    for mode in ("2048", "8192", "unfused"):
        assert abs(out["0"][2] - out[mode][2]) <= 1e-10 * abs(out["0"][2])


@pytest.mark.parametrize("n", [1, 31, 32, 33, 500, 4096, 4097, 9000])
def test_symv_lower(ctx, dbg, n):
    """launch_symv_lower (the K p of the mixed-precision fit's conjugate gradients): only the lower triangle is read
    (the upper one holds NaN here), every stored entry used twice; against numpy."""
    rng = np.random.default_rng(n)
    B = rng.standard_normal((n, n))
    K = B + B.T
    ld = n + (n % 2) + 2
    Kd = np.full((ld, n), np.nan, order="F")
    Kd[:n] = np.tril(K) + np.triu(np.full((n, n), np.nan), 1)
    p, base = rng.standard_normal(n), rng.standard_normal(n)
    out = np.empty(n)
    dbg.agp_debug_symv_lower.restype = C.c_int
    dbg.agp_debug_symv_lower.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_double, C.c_double,
                                         C.c_void_p, C.c_void_p]
    assert dbg.agp_debug_symv_lower(ctx._h, _p(Kd), n, ld, _p(p), -1.0, 0.5, _p(base), _p(out)) == 0
    want = -(K @ p) + 0.5 * base
    assert np.abs(out - want).max() <= 1e-12 * max(1., np.abs(K).sum(axis=1).max() * np.abs(p).max())
    assert dbg.agp_debug_symv_lower(ctx._h, _p(Kd), n, ld, _p(p), 1.0, 0.0, None, _p(out)) == 0
    assert np.abs(out - K @ p).max() <= 1e-12 * max(1., np.abs(K).sum(axis=1).max() * np.abs(p).max())

