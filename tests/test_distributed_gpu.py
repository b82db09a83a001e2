"""GPU tests of the sharded-fit path on one rank: the same schedule as the
gloo tests, but with HipBlockOps (block-level C-ABI on CUDA tensors), with and
without an initialised single-rank RCCL process group."""
import os

import numpy as np
import pytest

import albatross_amd as ab
from albatross_amd.distributed import HipBlockOps, ShardedGaussianProcessFit
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu


def problem(n, dim=3):
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10., (n, dim))
    x[5] = x[2]
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    yvar = rng.uniform(0., 0.05, n)
    return x, y, yvar


@pytest.mark.parametrize("n,block", [(100, 128), (700, 128), (1500, 512), (2048, 512), (1000, 256)])
def test_sharded_fit_one_rank_matches_oracle(ctx, n, block):
    x, y, yvar = problem(n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    fit = ShardedGaussianProcessFit(HipBlockOps(ctx, "cuda:0"), cov, block=block)
    res = fit.fit(x, y, yvar)
    ofit = orc.OracleFit(cov, x, y, yvar)
    assert np.abs(res.information - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
    assert abs(res.log_determinant - ofit.log_determinant) <= 1e-6 * n
    # same answer as the single-GPU entry point
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    assert np.abs(res.information - fm.get_fit().information).max() <= 1e-9 * np.abs(ofit.information).max()


def test_sharded_fit_errors(ctx):
    x, y, yvar = problem(600)
    ops = HipBlockOps(ctx, "cuda:0")
    with pytest.raises(ab.NotPositiveDefiniteError, match="pivot 5"):
        ShardedGaussianProcessFit(ops, ab.SquaredExponential(1., 1.), block=128).fit(x, y)
    xn = x.copy()
    xn[300, 0] = np.nan
    with pytest.raises(ab.NanInputError):
        ShardedGaussianProcessFit(ops, ab.Matern52(2., 1.) + ab.IndependentNoise(0.1), block=128).fit(xn, y)


def test_sharded_fit_with_rccl_group_of_one(ctx):
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        x, y, yvar = problem(900)
        cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
        # force_collectives: every panel / solution broadcast and the NaN all-reduce really go
        # through RCCL (self-broadcast) on buffers written by the HIP library's kernels
        fit = ShardedGaussianProcessFit(HipBlockOps(ctx, "cuda:0"), cov, block=256, force_collectives=True)
        assert fit.active and fit.world == 1
        res = fit.fit(x, y, yvar)
        ofit = orc.OracleFit(cov, x, y, yvar)
        assert np.abs(res.information - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
    finally:
        dist.destroy_process_group()
