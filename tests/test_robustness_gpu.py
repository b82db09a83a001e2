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
