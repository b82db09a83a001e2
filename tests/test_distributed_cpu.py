"""CPU tests (-m "not gpu") of the multi-GPU path: the sharded-fit schedule of
albatross_amd/distributed.py run over gloo with world_size 1, 2 and 3, on a
numpy implementation of the block interface, checked against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import albatross_amd as ab
from albatross_amd.distributed import ShardedGaussianProcessFit, ShardLayout
from oracle import oracle_py as orc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_layout_is_a_partition():
    for n, world, block in [(1000, 3, 128), (16384, 8, 512), (130, 4, 128), (512, 2, 512)]:
        lay = ShardLayout(n, world, block)
        cols = sorted(c for r in range(world) for c in lay.owned(r))
        assert cols == list(range(lay.n_blocks))
        assert sum(lay.width(c) for c in cols) == n
        assert all(0 <= lay.owner(c) < world for c in cols)
    lay = ShardLayout(16384, 8, 512)
    per_rank = [lay.local_elements(r) for r in range(8)]
    assert max(per_rank) / min(per_rank) < 1.05  # snake-cyclic assignment balances the triangle
    with pytest.raises(ValueError):
        ShardLayout(100, 2, 100)


def _problem(n):
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10., (n, 3))
    x[5] = x[2]  # duplicate point: off-diagonal noise
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    yvar = rng.uniform(0., 0.05, n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    return cov, x, y, yvar


def _worker(rank, world, port, n, block, out):
    from dist_cpu_ops import NumpyBlockOps
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        cov, x, y, yvar = _problem(n)
        fit = ShardedGaussianProcessFit(NumpyBlockOps(), cov, block=block)
        res = fit.fit(x, y, yvar)
        res2 = fit.fit(x, 2. * y, yvar)  # storage re-use across fits
        bad = None
        try:
            xs = np.random.default_rng(99).uniform(0., 10., (n, 3))  # no early duplicate
            xs[n // 2 + 3] = xs[1]  # singular at a pivot inside a later block column
            ShardedGaussianProcessFit(NumpyBlockOps(), ab.SquaredExponential(1., 1.), block=block).fit(xs, y)
        except ab.NotPositiveDefiniteError as e:
            bad = str(e)
        out[rank] = (res.information, res.log_determinant, res2.information, bad)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,block,lookahead", [(2, 700, 128, "1"), (3, 1000, 256, "1"), (2, 512, 512, "1"),
                                                     (4, 300, 128, "1"), (3, 700, 128, "0")])
def test_sharded_fit_over_gloo(world, n, block, lookahead, monkeypatch):
    # AGP_SHARDED_LOOKAHEAD: "1" = panel c + 1 is factored and broadcast while panel c is applied (default),
    # "0" = the synchronous schedule; the spawned ranks inherit the environment
    monkeypatch.setenv("AGP_SHARDED_LOOKAHEAD", lookahead)
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n, block, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(120)
            assert p.exitcode == 0
        cov, x, y, yvar = _problem(n)
        ofit = orc.OracleFit(cov, x, y, yvar)
        for r in range(world):
            info, logdet, info2, bad = out[r]
            assert np.abs(info - ofit.information).max() <= 1e-9 * np.abs(ofit.information).max()
            assert abs(logdet - ofit.log_determinant) <= 1e-9 * abs(ofit.log_determinant)
            assert np.abs(info2 - 2. * ofit.information).max() <= 1e-9 * np.abs(ofit.information).max() * 2
            assert bad is not None and f"pivot {n // 2 + 3}" in bad
        # every rank ends with the same answer, bit for bit
        assert all(np.array_equal(out[0][0], out[r][0]) for r in range(world))


def test_sharded_fit_single_process():
    from dist_cpu_ops import NumpyBlockOps
    cov, x, y, yvar = _problem(333)
    res = ShardedGaussianProcessFit(NumpyBlockOps(), cov, block=128).fit(x, y, yvar)
    ofit = orc.OracleFit(cov, x, y, yvar)
    assert np.abs(res.information - ofit.information).max() <= 1e-9 * np.abs(ofit.information).max()
    with pytest.raises(ab.NanInputError):
        xn = x.copy()
        xn[7, 1] = np.nan
        ShardedGaussianProcessFit(NumpyBlockOps(), cov, block=128).fit(xn, y, yvar)
