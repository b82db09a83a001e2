"""GPU robustness checks: device memory does not grow with repeated use of the whole C-ABI."""
import os
import subprocess
import sys

import numpy as np
import pytest

import albatross_amd as ab
from conftest import synthetic_3d

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_no_device_memory_growth():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "leak_check.py")], capture_output=True, text=True,
                         timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert out.stdout.strip().endswith("ok")


def test_two_contexts_in_two_threads():
    """"One context per host thread; calls on distinct contexts are concurrent-safe" (include/albatross_amd.h):
    two threads fit / predict concurrently on their own contexts and get the single-thread answers."""
    import threading

    import numpy as np

    import albatross_amd as ab

    rng = np.random.default_rng(3)
    n = 900
    x = rng.uniform(0., 10., (n, 3))
    ys = [np.sin(x).sum(axis=1) + 0.1 * rng.standard_normal(n) for _ in range(2)]
    xs = rng.uniform(0., 10., (64, 3))
    covs = [ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.2), ab.SquaredExponential(1.5, 0.8) + ab.IndependentNoise(0.1)]

    def work(i, ctx, out):
        model = ab.gp_from_covariance(covs[i], context=ctx)
        res = []
        for _ in range(6):
            fm = model.fit(ab.RegressionDataset(x, ys[i]))
            res.append((fm.get_fit().information, fm.predict(xs).joint().covariance, model.log_likelihood(ab.RegressionDataset(x, ys[i]))))
        out[i] = res

    ref = {}
    ctx0 = ab.Context(0)
    for i in range(2):
        work(i, ctx0, ref)
    got = {}
    ctxs = [ab.Context(0), ab.Context(0)]
    threads = [threading.Thread(target=work, args=(i, ctxs[i], got)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        for (a, b, c), (ra, rb, rc) in zip(got[i], ref[i]):
            assert np.array_equal(a, ra) and np.array_equal(b, rb) and c == rc  # same kernels, same order: bit-identical


@pytest.mark.parametrize("n,reps", [(2048, 25), (4096, 25), (8192, 12), (16384, 6)])
def test_repeated_fits_are_bitwise_identical(ctx, n, reps):
    """The factorisation runs on two streams with event hand-offs, split bulk updates and (N % 512 == 0) the
    blocked backward substitution: a missing dependency would show up as run-to-run differences.  Every repeat of
    the same fit must reproduce the information vector and the log-determinant bit for bit."""
    x, y = synthetic_3d(n, 1234 + n)
    model = ab.gp_from_covariance(ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1), context=ctx)
    ds = ab.RegressionDataset(x, y)
    first = None
    for _ in range(reps):
        fm = model.fit(ds)
        got = (fm.get_fit().information.copy(), fm.get_fit().log_determinant)
        del fm
        if first is None:
            first = got
        else:
            assert np.array_equal(got[0], first[0]) and got[1] == first[1]


@pytest.mark.parametrize("n,reps", [(129, 300), (512, 300), (1280, 150), (1920, 100), (2047, 100)])
def test_repeated_small_fits_are_bitwise_identical(ctx, n, reps):
    """The polling kernels of a small fit - the step launches' hand-overs and the one-launch back substitution, whose
    workgroups wait for each other's values inside ONE launch - must neither time out nor depend on timing: every repeat
    of the same fit reproduces the information vector bit for bit (scripts/stress_small_fits.py runs thousands)."""
    x, y = synthetic_3d(n, 77 + n)
    model = ab.gp_from_covariance(ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1), context=ctx)
    ds = ab.RegressionDataset(x, y)
    first = None
    for i in range(reps):
        fm = model.fit(ds)
        if i % 10 == 0:
            got = (fm.get_fit().information.copy(), fm.get_fit().log_determinant)
            if first is None:
                first = got
            else:
                assert np.array_equal(got[0], first[0]) and got[1] == first[1], (n, i)
        del fm


def test_merged_bulk_launches_on_several_contexts_at_once(ctx):
    """The merged bulk launches of large fits (round 6, csrc/chol.hip: factor_lower - the next block column's tiles ride at
    the head of the bulk launch, a one-wave gate kernel on the panel stream waits for their count) make one stream of a
    context wait for a LAUNCH ON ANOTHER STREAM of the same context.  Three contexts on three host threads put twelve and
    more streams on the runtime's few hardware queues: every fit must still be the fit (bit-identical to the one computed
    alone, same as AGP_MERGE_ABOVE=0 to rounding) - should two streams of one context ever share a hardware queue, the gate's
    deadline turns the fit into an error and agp_fit_create repeats it without merged launches instead of hanging."""
    import threading
    n = 10240  # (merged launches while more than 8704 trailing rows remain: the first two outer steps)
    x, y = synthetic_3d(n, 91)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    ds = ab.RegressionDataset(x, y)
    want = np.array(ab.gp_from_covariance(cov, context=ctx).fit(ds).get_fit().information)
    K = ctx.gram(cov, ab.Measurement(x))
    assert np.abs(K @ want - y).max() <= 1e-9 * np.abs(K).sum(axis=1).max() * np.abs(want).max()
    del K
    results, errors = {}, []

    def work(tag):
        try:
            c = ab.Context(0)
            for _ in range(3):
                results[tag] = np.array(ab.gp_from_covariance(cov, context=c).fit(ds).get_fit().information)
            c.close()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)
    ts = [threading.Thread(target=work, args=(i,)) for i in range(3)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errors, errors
    for i in range(3):
        assert np.array_equal(results[i], want)


def test_merged_bulk_launches_match_the_separate_update(make_ctx, monkeypatch):
    """AGP_MERGE_ABOVE=0 (U1 as a launch of its own on the panel stream, as in rounds 1-5) against the default: the same
    factorisation to rounding (the 64 x 64 tiles of the separate launch start their accumulators from C, the 128 x 128 head
    tiles add C in eight parts), both against the residual of the normal equations."""
    n = 12288
    x, y = synthetic_3d(n, 92)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    ds = ab.RegressionDataset(x, y)
    c1 = make_ctx()
    a1 = np.array(ab.gp_from_covariance(cov, context=c1).fit(ds).get_fit().information)
    ld1 = ab.gp_from_covariance(cov, context=c1).fit(ds).get_fit().log_determinant
    monkeypatch.setenv("AGP_MERGE_ABOVE", "0")
    c0 = make_ctx()
    f0 = ab.gp_from_covariance(cov, context=c0).fit(ds).get_fit()
    a0, ld0 = np.array(f0.information), f0.log_determinant
    assert np.abs(a1 - a0).max() <= 1e-9 * np.abs(a0).max()
    assert abs(ld1 - ld0) <= 1e-9 * abs(ld0)
    K = c1.gram(cov, ab.Measurement(x))
    for a in (a0, a1):
        assert np.abs(K @ a - y).max() <= 1e-9 * np.abs(K).sum(axis=1).max() * np.abs(a).max()
