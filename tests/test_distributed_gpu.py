"""GPU tests of the sharded fit (include/albatross_amd.h: agp_comm_* / agp_sharded_fit_*): the library's schedule with the
HIP block operations
  - on one rank without a transport (the launch sequence of the single-GPU factorisation),
  - on one rank with an RCCL communicator of size one and AGP_SHARD_FORCE_COMM=1: every broadcast, all-gather and
    all-reduce of the multi-rank schedule goes through RCCL on buffers the kernels wrote,
  - on TWO processes sharing this box's one GPU, collectives over gloo through the callback transport (RCCL refuses two
    ranks per device): the real multi-rank data flow with the real kernels,
and the replicated factor's predictions; all against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest

import albatross_amd as ab
from albatross_amd.distributed import Communicator, ShardedGaussianProcessFit
from oracle import oracle_py as orc

pytestmark = pytest.mark.gpu
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def problem(n, dim=3):
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10., (n, dim))
    x[5] = x[2]
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    yvar = rng.uniform(0., 0.05, n)
    return x, y, yvar


@pytest.mark.parametrize("n,block", [(100, 128), (700, 128), (1500, 512), (2048, 512), (1000, 256), (3000, 512)])
def test_sharded_fit_one_rank_matches_oracle(make_ctx, n, block, monkeypatch):
    monkeypatch.setenv("AGP_SHARD_BLOCK", str(block))
    ctx = make_ctx()  # (the switches are read when the context is created)
    x, y, yvar = problem(n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    res = ShardedGaussianProcessFit(ctx, cov).fit(x, y, yvar)
    ofit = orc.OracleFit(cov, x, y, yvar)
    assert np.abs(res.information - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
    assert abs(res.log_determinant - ofit.log_determinant) <= 1e-6 * n
    # same answer as the single-GPU entry point
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar)))
    assert np.abs(res.information - fm.get_fit().information).max() <= 1e-9 * np.abs(ofit.information).max()


@pytest.mark.parametrize("n,block,forced", [(900, 128, False), (1700, 256, True), (2048, 512, True), (1300, 512, False)])
def test_sharded_predict_marginal_without_replication(make_ctx, n, block, forced, monkeypatch):
    """agp_sharded_predict_marginal: the distributed forward substitution on one rank - without a transport (the local
    matrix is the whole factor) and with the multi-rank schedule forced on through an RCCL group of one (every
    broadcast / all-reduce a real RCCL call) - against the oracle."""
    monkeypatch.setenv("AGP_SHARD_BLOCK", str(block))
    comm = None
    if forced:
        monkeypatch.setenv("AGP_SHARD_FORCE_COMM", "1")
    else:
        monkeypatch.delenv("AGP_SHARD_FORCE_COMM", raising=False)
    ctx = make_ctx()  # (the switches are read when the context is created)
    if forced:
        comm = Communicator.rccl(ctx, 1, 0, Communicator.unique_id())
    try:
        x, y, yvar = problem(n)
        cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
        sharded = ShardedGaussianProcessFit(ctx, cov, comm)
        sharded.fit(x, y, yvar)
        xs = np.random.default_rng(3).uniform(0., 10., (70, 3))
        mean, var = sharded.predict_marginal(xs)
        ofit = orc.OracleFit(cov, x, y, yvar)
        om, ov = ofit.predict_marginal(xs)
        assert np.abs(mean - om).max() <= 1e-8 * np.abs(om).max()
        assert np.abs(var - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-9
        # the joint prediction from the same distributed substitution (gp.hpp:103-113)
        jm, jc = sharded.predict_joint(xs)
        ojm, ojc = ofit.predict_joint(xs)
        assert np.abs(jm - ojm).max() <= 1e-8 * np.abs(ojm).max()
        assert np.abs(jc - ojc).max() <= 1e-8 * np.abs(ojc).max() + 1e-9 and np.array_equal(jc, jc.T)
    finally:
        if comm is not None:
            comm.close()


@pytest.mark.parametrize("n,block", [(700, 128), (1000, 256), (1500, 512)])
def test_one_rank_replicate_any_block(make_ctx, n, block, monkeypatch):
    """agp_sharded_fit_replicate on the one-rank, no-transport path: the tile images are laid out per 128-block by the
    single-GPU factorisation whatever AGP_SHARD_BLOCK says (it used to read them with the 512-block stride)."""
    monkeypatch.setenv("AGP_SHARD_BLOCK", str(block))
    monkeypatch.delenv("AGP_SHARD_FORCE_COMM", raising=False)
    ctx = make_ctx()
    x, y, yvar = problem(n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    sharded = ShardedGaussianProcessFit(ctx, cov)
    sharded.fit(x, y, yvar)
    fm = sharded.replicate(ab.gp_from_covariance(cov, context=ctx))
    ofit = orc.OracleFit(cov, x, y, yvar)
    xs = np.random.default_rng(3).uniform(0., 10., (40, 3))
    om, ov = ofit.predict_marginal(xs)
    marg = fm.predict(xs).marginal()
    assert np.abs(marg.mean - om).max() <= 1e-8 * np.abs(om).max()
    assert np.abs(marg.covariance - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-9
    rhs = np.random.default_rng(4).standard_normal(n)
    assert np.abs(fm.get_fit().solve(rhs) - ofit.solve(rhs)).max() <= 1e-8 * np.abs(ofit.solve(rhs)).max()


def test_sharded_fit_config3_size_one_rank(ctx):
    """N = 16384 through the sharded entry point on one rank: same result as agp_fit_create."""
    from conftest import synthetic_3d
    x, y = synthetic_3d(16384, 44)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
    res = ShardedGaussianProcessFit(ctx, cov).fit(x, y)
    fm = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y))
    info = fm.get_fit().information
    assert np.abs(res.information - info).max() <= 1e-9 * np.abs(info).max()
    assert abs(res.log_determinant - fm.get_fit().log_determinant) <= 1e-9 * abs(res.log_determinant)


def test_sharded_fit_errors(make_ctx, monkeypatch):
    monkeypatch.setenv("AGP_SHARD_BLOCK", "128")
    ctx = make_ctx()
    x, y, yvar = problem(600)
    with pytest.raises(ab.NotPositiveDefiniteError, match="pivot 5"):
        ShardedGaussianProcessFit(ctx, ab.SquaredExponential(1., 1.)).fit(x, y)
    xs = np.random.default_rng(99).uniform(0., 10., (600, 3))
    xs[303] = xs[1]  # singular at a pivot inside a later row block
    with pytest.raises(ab.NotPositiveDefiniteError, match="pivot 303"):
        ShardedGaussianProcessFit(ctx, ab.SquaredExponential(1., 1.)).fit(xs, y)
    xn = x.copy()
    xn[300, 0] = np.nan
    with pytest.raises(ab.NanInputError):
        ShardedGaussianProcessFit(ctx, ab.Matern52(2., 1.) + ab.IndependentNoise(0.1)).fit(xn, y)


@pytest.mark.parametrize("n,block", [(900, 256), (1700, 128), (2048, 512)])
def test_sharded_fit_through_rccl_group_of_one(make_ctx, n, block, monkeypatch):
    """RCCL itself (the ROCm installation's librccl, bound by the library at run time): communicator of size one, the
    multi-rank schedule forced on, so that ncclBroadcast / ncclAllGather / ncclAllReduce run on the library's streams"""
    monkeypatch.setenv("AGP_SHARD_BLOCK", str(block))
    monkeypatch.setenv("AGP_SHARD_FORCE_COMM", "1")
    ctx = make_ctx()
    comm = Communicator.rccl(ctx, 1, 0, Communicator.unique_id())
    try:
        assert comm.world == 1 and comm.rank == 0
        assert comm.all_reduce([3.5, -1.0], "sum").tolist() == [3.5, -1.0]
        comm.barrier()
        x, y, yvar = problem(n)
        cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
        sharded = ShardedGaussianProcessFit(ctx, cov, comm)
        res = sharded.fit(x, y, yvar)
        ofit = orc.OracleFit(cov, x, y, yvar)
        assert np.abs(res.information - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
        assert abs(res.log_determinant - ofit.log_determinant) <= 1e-6 * n
        # replicated factor (all-gather of the row blocks through RCCL) -> ordinary predictions
        model = ab.gp_from_covariance(cov, context=ctx)
        fm = sharded.replicate(model)
        xs = np.random.default_rng(3).uniform(0., 10., (50, 3))
        om, ov = ofit.predict_marginal(xs)
        marg = fm.predict(xs).marginal()
        assert np.abs(marg.mean - om).max() <= 1e-8 * np.abs(om).max()
        assert np.abs(marg.covariance - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-9
        L = fm.get_fit().factor()
        K = orc.gram(cov, x, x_meas=True) + np.diag(yvar)
        assert np.abs(L @ L.T - K).max() <= 1e-11 * np.abs(K).max()
    finally:
        comm.close()


def test_sharded_fit_config3_size_through_rccl(make_ctx, monkeypatch):
    """The multi-rank schedule at BASELINE config 3's size (N = 16384, 512-row blocks, 32 block columns) with every
    broadcast / all-gather / all-reduce a real RCCL call (communicator of size one): same answer as agp_fit_create on
    the dataset bench.py times."""
    from bench import make_dataset
    monkeypatch.setenv("AGP_SHARD_FORCE_COMM", "1")
    monkeypatch.delenv("AGP_SHARD_BLOCK", raising=False)
    ctx = make_ctx()
    comm = Communicator.rccl(ctx, 1, 0, Communicator.unique_id())
    try:
        x, y = make_dataset(16384, 44)
        cov = ab.SquaredExponential(1.0, 1.0) + ab.IndependentNoise(0.1)
        res = ShardedGaussianProcessFit(ctx, cov, comm).fit(x, y)
        fit = ab.gp_from_covariance(cov, context=ctx).fit(ab.RegressionDataset(x, y)).get_fit()
        info = fit.information
        assert np.abs(res.information - info).max() <= 1e-9 * np.abs(info).max()
        assert abs(res.log_determinant - fit.log_determinant) <= 1e-9 * abs(fit.log_determinant)
    finally:
        comm.close()


def _rccl_worker(rank, world, port, n, block, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AGP_SHARD_BLOCK"] = str(block)
    os.environ["AGP_COMM_TIMEOUT_S"] = "30"  # a deadlock fails fast
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    torch.cuda.init()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = ab.Context(rank)  # one rank per GPU
        comm = Communicator.from_torch(ctx, transport="rccl")
        x, y, yvar = problem(n)
        cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
        sharded = ShardedGaussianProcessFit(ctx, cov, comm)
        res = sharded.fit(x, y, yvar)
        fm = sharded.replicate(ab.gp_from_covariance(cov, context=ctx))
        xs = np.random.default_rng(3).uniform(0., 10., (64, 3))
        mine = slice(rank * 64 // world, (rank + 1) * 64 // world)
        marg = fm.predict(xs[mine]).marginal()
        out[rank] = (res.information, res.log_determinant, marg.mean, marg.covariance, None)
        comm.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("n,block", [(1500, 128), (4096, 256), (6000, 512)])
def test_sharded_fit_rccl_one_rank_per_gpu(n, block):
    """The real thing: one rank per GPU, RCCL broadcast / all-gather / all-reduce between them.  Needs >= 2 GPUs (the pool's
    boxes have one: skipped there, run by whoever has a node)."""
    ngpu = ab._capi.load().agp_device_count()
    if ngpu < 2:
        pytest.skip("needs at least two GPUs")
    world = min(ngpu, 4)
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    with mpc.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [mpc.Process(target=_rccl_worker, args=(r, world, port, n, block, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        _check_against_oracle(out, world, n, expect_bad=False)


def _worker(rank, world, port, n, block, out, transport="callbacks", env=None):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ["AGP_SHARD_BLOCK"] = str(block)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.update(env or {})
    import torch
    import torch.distributed as dist
    torch.cuda.init()  # torch's HIP runtime first (tests/conftest.py)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = ab.Context(0)  # every rank on THIS box's one GPU
        if transport == "ipc_small":  # 64 Ki doubles per mailbox slot: panels, stacks and tile images travel in pieces
            comm = Communicator.ipc(ctx, mailbox_doubles=1 << 16)
        else:
            comm = Communicator.from_torch(ctx, transport=transport)
        x, y, yvar = problem(n)
        cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
        sharded = ShardedGaussianProcessFit(ctx, cov, comm)
        res = sharded.fit(x, y, yvar)
        if transport != "callbacks":
            assert sharded.stage(2) == (0. if (env or {}).get("AGP_SHARD_HOST_PACING") == "1" else 1.)  # the pacing that ran
            res2 = sharded.fit(x, 2. * y, yvar)  # the flags' sequence numbers carry on from fit to fit
            assert np.abs(res2.information - 2. * res.information).max() <= 1e-12 * np.abs(res.information).max()
            res = sharded.fit(x, y, yvar)
        fm = sharded.replicate(ab.gp_from_covariance(cov, context=ctx))
        xs = np.random.default_rng(3).uniform(0., 10., (64, 3))
        mine = slice(rank * 64 // world, (rank + 1) * 64 // world)  # this rank's share of the test points
        marg = fm.predict(xs[mine]).marginal()
        dmean, dvar = sharded.predict_marginal(xs)  # all 64 points, from the sharded factor itself (collective)
        assert np.abs(dmean[mine] - marg.mean).max() <= 1e-9 * np.abs(marg.mean).max()
        assert np.abs(dvar[mine] - marg.covariance).max() <= 1e-9 * np.abs(marg.covariance).max() + 1e-10
        # ... and the joint prediction (every rank its own rows of V, one all-reduce of the 64 x 64 product) against the
        # replicated factor's
        jmean, jcov = sharded.predict_joint(xs)
        rj = fm.predict(xs).joint()
        assert np.abs(jmean - rj.mean).max() <= 1e-9 * np.abs(rj.mean).max()
        assert np.abs(jcov - rj.covariance).max() <= 1e-9 * np.abs(rj.covariance).max() + 1e-10
        assert np.abs(np.diag(jcov) - dvar).max() <= 1e-9 * np.abs(dvar).max() + 1e-10
        bad = None
        try:
            xb = np.random.default_rng(99).uniform(0., 10., (n, 3))
            xb[n // 2 + 3] = xb[1]
            ShardedGaussianProcessFit(ctx, ab.SquaredExponential(1., 1.), comm).fit(xb, y)
        except ab.NotPositiveDefiniteError as e:
            bad = str(e)
        out[rank] = (res.information, res.log_determinant, marg.mean, marg.covariance, bad)
        comm.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


# (4, 2300, 128): 18 block columns over 4 ranks in snake order - every rank is the root of several broadcasts (roots != 0)
# (8, 1400, 128) / (8, 1100, 256): the world size of the node (8 processes sharing this box's one GPU; the gloo-callback transport is
# host-synchronous - (8, 4200, 128), (8, 2100, 256) and (4, 4200, 128) took 285 s of the suite here; those sizes run on the
# device-asynchronous ipc transport below in 20-27 s each)
@pytest.mark.parametrize("world,n,block", [(2, 1500, 128), (3, 2100, 256), (2, 2048, 512), (4, 2300, 128), (8, 1400, 128),
                                           (8, 1100, 256)])
def test_sharded_fit_two_processes_one_gpu(world, n, block):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    with mpc.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [mpc.Process(target=_worker, args=(r, world, port, n, block, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        _check_against_oracle(out, world, n, expect_bad=True)


# The asynchronous schedule with asynchronous collectives on ONE GPU (csrc/shard_ipc.hip): every broadcast / all-gather /
# all-reduce is a few kernels on the collectives' queue writing into the peers' hipIpc mailboxes - the host never waits
# between the steps, exactly as over RCCL.  Device pacing (flags + gate kernels, the default) and host pacing.
@pytest.mark.parametrize("world,n,block,transport,pacing", [
    (2, 1500, 128, "ipc", "device"), (2, 2048, 512, "ipc", "host"), (3, 2100, 256, "ipc_small", "device"),
    (4, 4200, 128, "ipc", "device"), (4, 2100, 128, "ipc_small", "host"), (8, 4200, 128, "ipc", "device"),
    (8, 2100, 256, "ipc_small", "device")])
def test_sharded_fit_async_transport_processes_one_gpu(world, n, block, transport, pacing):
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    env = {"AGP_COMM_TIMEOUT_S": "60"}
    if pacing == "host":
        env["AGP_SHARD_HOST_PACING"] = "1"
    with mpc.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [mpc.Process(target=_worker, args=(r, world, port, n, block, out, transport, env)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(420)
            assert p.exitcode == 0
        _check_against_oracle(out, world, n, expect_bad=True)


def _check_against_oracle(out, world, n, expect_bad):
    x, y, yvar = problem(n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    ofit = orc.OracleFit(cov, x, y, yvar)
    xs = np.random.default_rng(3).uniform(0., 10., (64, 3))
    om, ov = ofit.predict_marginal(xs)
    mean = np.concatenate([out[r][2] for r in range(world)])
    var = np.concatenate([out[r][3] for r in range(world)])
    for r in range(world):
        info, logdet, _, _, bad = out[r]
        assert np.abs(info - ofit.information).max() <= 1e-8 * np.abs(ofit.information).max()
        assert abs(logdet - ofit.log_determinant) <= 1e-6 * n
        if expect_bad:
            assert bad is not None and f"pivot {n // 2 + 3}" in bad
    assert np.abs(mean - om).max() <= 1e-8 * np.abs(om).max()
    assert np.abs(var - ov).max() <= 1e-8 * np.abs(ov).max() + 1e-9
    assert all(np.array_equal(out[0][0], out[r][0]) for r in range(world))


# ---- sparse GP (PITC) with its observations split by group over the ranks (BASELINE configs[4]) ----
def _pitc(n, m, seed):
    rng = np.random.default_rng(seed)
    x = np.sort(rng.uniform(0., n / 16., n))
    y = np.sin(x) + 0.1 * np.cos(10. * x) + 0.1 * rng.standard_normal(n)
    yvar = rng.uniform(0.01, 0.02, n)
    cov = ab.SquaredExponential(1.0, 1.0) + ab.measurement_only(ab.IndependentNoise(0.1))
    return x, y, yvar, cov, np.linspace(x.min(), x.max(), m)


def _sparse_model(ctx, cov, x, u, gs):
    sorted_x = np.sort(x)

    def grouper(f):
        r = np.searchsorted(sorted_x, np.asarray(f, dtype=np.float64).reshape(-1)) // gs
        return r if np.ndim(f) else int(r[0])
    grouper.vectorized = True
    model = ab.sparse_gp_from_covariance(cov, grouper, ab.FixedInducingPoints(u), "pitc", context=ctx)
    model.set_param("inducing_nugget", 1e-6)
    return model


def _sparse_worker(rank, world, port, n, m, gs, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = ab.Context(0)
        comm = Communicator.from_torch(ctx, transport="callbacks")
        x, y, yvar, cov, u = _pitc(n, m, 7)
        model = _sparse_model(ctx, cov, x, u, gs)
        groups = np.arange(n) // gs
        mine = (groups % world) == rank  # whole groups, dealt round-robin
        ds = ab.RegressionDataset(x[mine], ab.MarginalDistribution(y[mine], yvar[mine]))
        fm = model.fit(ds, comm=comm)
        xs = np.linspace(x.min(), x.max(), 40)
        marg = fm.predict(xs).marginal()
        out[rank] = (fm.get_fit().information, fm.get_fit().nll, marg.mean, marg.covariance, model.log_likelihood(ds, comm=comm))
        comm.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n,m,gs", [(2, 4096, 96, 256), (3, 3000, 64, 250)])
def test_sparse_fit_sharded_by_group(ctx, world, n, m, gs):
    """agp_sparse_fit_create_sharded: every rank its own groups, the m x m sums all-reduced; == the one-process fit of
    all observations (and through it the oracle, tests/test_sparse_gp_gpu.py)."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    with mpc.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [mpc.Process(target=_sparse_worker, args=(r, world, port, n, m, gs, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(300)
            assert p.exitcode == 0
        x, y, yvar, cov, u = _pitc(n, m, 7)
        model = _sparse_model(ctx, cov, x, u, gs)
        ds = ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar))
        ref = model.fit(ds)
        v = ref.get_fit().information
        xs = np.linspace(x.min(), x.max(), 40)
        rm = ref.predict(xs).marginal()
        keys = np.arange(n) // gs
        ofit = orc.OracleSparseFit(cov, x, keys, y, yvar, u, model.get_params()["measurement_nugget"], 1e-6)
        for r in range(world):
            info, nll, mean, var, ll = out[r]
            assert np.abs(info - v).max() <= 1e-8 * np.abs(v).max()
            assert np.abs(info - ofit.information).max() <= 1e-7 * np.abs(v).max()
            assert abs(nll - ref.get_fit().nll) <= 1e-8 * n and abs(nll - ofit.nll) <= 1e-8 * n and abs(ll + nll) <= 1e-9 * n
            assert np.abs(mean - rm.mean).max() <= 1e-8 and np.abs(var - rm.covariance).max() <= 1e-8
        assert all(np.array_equal(out[0][0], out[r][0]) for r in range(world))


def _sparse_error_worker(rank, world, port, case, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    torch.cuda.init()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ctx = ab.Context(0)
        comm = Communicator.from_torch(ctx, transport="callbacks")
        n, m, gs = 2048, 48, 256
        x, y, yvar, cov, u = _pitc(n, m, 11)
        if case == "inducing" and rank == 1:
            u = u + 1e-3  # a data-dependent inducing-point strategy run on rank-local features would do this
        model = _sparse_model(ctx, cov, x, u, gs)
        mine = ((np.arange(n) // gs) % world) == rank
        xm, ym, vm = x[mine], y[mine].copy(), yvar[mine].copy()
        if case == "nan" and rank == 1:
            vm[5] = np.nan  # NaN in ONE rank's own blocks of A
        if case == "not_pd" and rank == 0:
            vm[:] = -5.0  # ONE rank's blocks of A are not positive definite
        try:
            model.fit(ab.RegressionDataset(xm, ab.MarginalDistribution(ym, vm)), comm=comm)
            out[rank] = "ok"
        except ab.AlbatrossAmdError as e:
            out[rank] = f"{type(e).__name__}: {e}"
        # the communicator is still usable: every rank left the failed fit at the same point
        out[rank] += " | " + str(comm.all_reduce([1.0], "sum")[0])
        comm.close()
        ctx.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("case", ["nan", "not_pd", "inducing"])
def test_sparse_fit_sharded_rank_local_failure_is_agreed(case):
    """A failure only ONE rank can see (NaN / a non-positive-definite block among its own groups, different inducing
    points) must end the collective fit on EVERY rank with an error - not leave the peers inside an all-reduce."""
    import torch.multiprocessing as mp
    mpc = mp.get_context("spawn")
    world = 2
    with mpc.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [mpc.Process(target=_sparse_error_worker, args=(r, world, port, case, out)) for r in range(world)]
        for p in procs:
            p.start()
        for p in procs:
            p.join(240)
            assert p.exitcode == 0
        for r in range(world):
            assert not out[r].startswith("ok"), out[r]
            assert out[r].endswith("| 2.0"), out[r]


def test_sparse_fit_sharded_through_rccl_group_of_one(ctx):
    comm = Communicator.rccl(ctx, 1, 0, Communicator.unique_id())
    try:
        x, y, yvar, cov, u = _pitc(2048, 64, 3)
        model = _sparse_model(ctx, cov, x, u, 256)
        ds = ab.RegressionDataset(x, ab.MarginalDistribution(y, yvar))
        a, b = model.fit(ds, comm=comm), model.fit(ds)
        assert np.array_equal(a.get_fit().information, b.get_fit().information) and a.get_fit().nll == b.get_fit().nll
    finally:
        comm.close()


def _run_bench(extra_env, *args):
    import json
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, BENCH_SINGLE_DEVICE="1", **extra_env)
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--n", "2048",
                        *args], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith('{"metric"')]
    return p.returncode, (json.loads(lines[-1]) if lines else None), p.stderr


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher: the two ranks are child processes; on this one-GPU box they share GPU 0
    and the collectives are staged over gloo (BENCH_SINGLE_DEVICE test mode - the transport string says so)."""
    rc, line, err = _run_bench({})
    assert rc == 0 and line is not None, err[-2000:]
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["n_ranks"] == 2
    assert "callbacks" in line["config"]["transport"] and "not RCCL" in line["config"]["transport"]
    assert line["self_check"]["ok"] and line["self_check"]["max_rel_residual"] < 1e-8
    assert "sharded_fallback" not in line
    # predict pts/sec of the N-GPU job: test points partitioned over the ranks + the distributed marginal prediction
    pr = line["predict"]
    assert pr["m"] == 2 * pr["m_per_gpu"] and pr["scaling"] == "weak" and pr["marginal_pts_per_sec"] > 0
    assert pr["sharded_factor"]["marginal_pts_per_sec"] > 0


def test_bench_falls_back_to_replicas_when_the_sharded_fit_fails_its_check():
    """one rank's self-check of the sharded fit "fails" (test hook): every rank agrees over gloo to measure independent fits
    instead, and the line says so - weak scaling, FALLBACK in the parallelism string, the reason recorded."""
    rc, line, err = _run_bench({"BENCH_TEST_SHARDED_FAILURE": "1"})
    assert rc == 5 and line is not None, err[-2000:]  # the labelled line is printed, but the run is not green
    assert line["scaling"] == "weak" and line["config"]["parallelism"].startswith("FALLBACK")
    rc, line, err = _run_bench({"BENCH_TEST_SHARDED_FAILURE": "1"}, "--allow-fallback")
    assert rc == 0 and line is not None, err[-2000:]
    assert line["scaling"] == "weak" and line["config"]["parallelism"].startswith("FALLBACK")
    assert "failed" in line["sharded_fallback"] and line["self_check"]["ok"]
    assert line["predict"]["m"] == 2 * line["predict"]["m_per_gpu"] and "sharded_factor" not in line["predict"]
    rc, line, err = _run_bench({"BENCH_TEST_SHARDED_FAILURE": "1"}, "--no-fallback")
    assert rc != 0 and line is None
