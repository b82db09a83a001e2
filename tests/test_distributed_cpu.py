"""CPU tests (-m "not gpu") of the multi-GPU path: the library's C++ sharded-fit schedule
(albatross_amd/csrc/shard_sched.hip: row-block-cyclic LL^T with look-ahead, both substitutions) driven through
`agp_debug_shard_factor_custom` (libalbatross_amd_debug.so) with numpy block operations (tests/dist_cpu_ops.py) and gloo collectives
(`Communicator.torch_callbacks`), world sizes 1-8, checked against the oracle."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.distributed as dist
import torch.multiprocessing as mp

import albatross_amd as ab
from albatross_amd import _capi as capi
from albatross_amd.distributed import Communicator, ShardLayout
from oracle import oracle_py as orc

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def test_layout_is_a_partition():
    for n, world, block in [(1000, 3, 128), (16384, 8, 512), (130, 4, 128), (512, 2, 512), (65536, 8, 512), (700, 5, 128)]:
        lay = ShardLayout(n, world, block)
        rows = np.concatenate([lay.global_rows(r) for r in range(world)])
        assert np.array_equal(np.sort(rows), np.arange(n))
        for r in range(world):
            g = lay.global_rows(r)
            assert len(g) == lay.local_rows(r)
            assert np.all(np.diff(g) > 0)  # local order = increasing global order
            lib = capi.load()
            for l in (0, len(g) // 2, len(g) - 1):
                if len(g):
                    assert lib.agp_shard_global_row(n, block, world, r, l) == g[l]
        assert all(0 <= lay.owner(b) < world for b in range(lay.n_blocks))
    # snake order: the update work of a row block grows like b^2; no rank may carry much more than its share
    lay = ShardLayout(16384, 8, 512)
    work = [sum((b + 1) ** 2 for b in range(lay.n_blocks) if lay.owner(b) == r) for r in range(8)]
    assert max(work) / min(work) < 1.25
    lay = ShardLayout(65536, 8, 512)
    work = [sum((b + 1) ** 2 for b in range(lay.n_blocks) if lay.owner(b) == r) for r in range(8)]
    assert max(work) / min(work) < 1.05


def _problem(n):
    rng = np.random.default_rng(n)
    x = rng.uniform(0., 10., (n, 3))
    x[5] = x[2]  # duplicate point: off-diagonal noise
    y = np.sin(x).sum(axis=1) + 0.1 * np.cos(10. * x[:, 0])
    yvar = rng.uniform(0., 0.05, n)
    cov = ab.Matern52(2.0, 1.0) + ab.IndependentNoise(0.1)
    return cov, x, y, yvar


def _covariance(cov, x, yvar):
    K = orc.gram(cov, x, x_meas=True)  # as_measurements(features), gp.hpp:288-290
    K[np.diag_indices_from(K)] += yvar  # gp.hpp:64-65
    return K


def _worker(rank, world, port, n, block, out):
    from dist_cpu_ops import sharded_factor_numpy
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        comm = Communicator.torch_callbacks()
        assert comm.world == world and comm.rank == rank
        # control plane helpers
        assert comm.all_reduce([float(rank), 1.0], "sum").tolist() == [world * (world - 1) / 2., float(world)]
        assert comm.all_reduce([float(rank)], "max")[0] == world - 1
        comm.barrier()
        cov, x, y, yvar = _problem(n)
        K = _covariance(cov, x, yvar)
        st, info, logdet, bad, calls = sharded_factor_numpy(K, y, block, comm)
        st2, info2, _, _, _ = sharded_factor_numpy(K, 2. * y, block, comm)
        # singular at a pivot inside a later block
        xs = np.random.default_rng(99).uniform(0., 10., (n, 3))
        xs[n // 2 + 3] = xs[1]
        Kb = orc.gram(ab.SquaredExponential(1., 1.), xs)
        st3, _, _, bad3, _ = sharded_factor_numpy(Kb, y, block, comm)
        out[rank] = (st, info, logdet, bad, st2, info2, st3, bad3, calls)
        comm.close()
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world,n,block", [(2, 700, 128), (3, 1000, 256), (2, 512, 512), (4, 300, 128), (3, 1300, 128),
                                           (4, 2100, 128), (2, 1024, 512),
                                           # world = 8, the size of the node the schedule is written for: 33 / 9 block columns,
                                           # every rank is the root of several broadcasts; (8, 700, 128): fewer blocks than
                                           # ranks, two ranks own nothing
                                           (8, 4200, 128), (8, 2100, 256), (8, 700, 128)])
def test_sharded_schedule_over_gloo(world, n, block):
    ctx = mp.get_context("spawn")
    with ctx.Manager() as mgr:
        out = mgr.dict()
        port = _free_port()
        procs = [ctx.Process(target=_worker, args=(r, world, port, n, block, out)) for r in range(world)]
        threads = os.environ.get("OMP_NUM_THREADS")
        os.environ["OMP_NUM_THREADS"] = "1"  # inherited by the ranks: `world` BLAS pools on this machine's few cores crawl
        try:
            for p in procs:
                p.start()
        finally:
            if threads is None:
                os.environ.pop("OMP_NUM_THREADS", None)
            else:
                os.environ["OMP_NUM_THREADS"] = threads
        for p in procs:
            p.join(180)
            assert p.exitcode == 0
        cov, x, y, yvar = _problem(n)
        ofit = orc.OracleFit(cov, x, y, yvar)
        nb = (n + block - 1) // block
        total_diag = 0
        for r in range(world):
            st, info, logdet, bad, st2, info2, st3, bad3, calls = out[r]
            assert st == capi.AGP_OK and bad == -1
            assert np.abs(info - ofit.information).max() <= 1e-9 * np.abs(ofit.information).max()
            assert abs(logdet - ofit.log_determinant) <= 1e-9 * abs(ofit.log_determinant)
            assert st2 == capi.AGP_OK
            assert np.abs(info2 - 2. * ofit.information).max() <= 2e-9 * np.abs(ofit.information).max()
            # the not-positive-definite exit is taken by every rank with the same pivot
            assert st3 == capi.AGP_ERR_NOT_POSITIVE_DEFINITE and bad3 == n // 2 + 3
            total_diag += calls["factor_diag"]
        assert total_diag == nb  # every diagonal block factored exactly once, by its owner
        # every rank ends with the same answer, bit for bit
        assert all(np.array_equal(out[0][1], out[r][1]) for r in range(world))


@pytest.mark.parametrize("n,block", [(333, 128), (512, 512), (900, 256), (100, 128)])
def test_sharded_schedule_single_process(n, block):
    from dist_cpu_ops import sharded_factor_numpy
    cov, x, y, yvar = _problem(n)
    st, info, logdet, bad, _ = sharded_factor_numpy(_covariance(cov, x, yvar), y, block, None)
    ofit = orc.OracleFit(cov, x, y, yvar)
    assert st == capi.AGP_OK
    assert np.abs(info - ofit.information).max() <= 1e-9 * np.abs(ofit.information).max()
    assert abs(logdet - ofit.log_determinant) <= 1e-9 * abs(ofit.log_determinant)


def test_single_rank_through_the_multi_rank_schedule(monkeypatch):
    """AGP_SHARD_FORCE_COMM=1: one rank runs the pack / all-gather / re-ordering / broadcast path of the schedule with
    its own (trivial) collectives - what the GPU box does to drive RCCL with a communicator of size one."""
    from dist_cpu_ops import sharded_factor_numpy
    monkeypatch.setenv("AGP_SHARD_FORCE_COMM", "1")
    seen = {"broadcast": 0, "all_gather": 0, "all_reduce": 0}

    def broadcast(buf, root):
        assert root == 0
        seen["broadcast"] += 1

    def all_gather(send, recv):
        recv[:] = send
        seen["all_gather"] += 1

    def all_reduce(buf, op):
        seen["all_reduce"] += 1

    comm = Communicator.callbacks(1, 0, broadcast, all_gather, all_reduce)
    n, block = 700, 128
    cov, x, y, yvar = _problem(n)
    st, info, logdet, bad, _ = sharded_factor_numpy(_covariance(cov, x, yvar), y, block, comm)
    ofit = orc.OracleFit(cov, x, y, yvar)
    assert st == capi.AGP_OK
    assert np.abs(info - ofit.information).max() <= 1e-9 * np.abs(ofit.information).max()
    nb = (n + block - 1) // block
    nsb = (nb + 3) // 4  # super-blocks of the back substitution (csrc/shard.h: SHARD_SUPER)
    # per block column one broadcast and (but for the last) one all-gather; the back substitution exchanges ONE all-reduce
    # per super-block but the last; log-determinant and bad pivot travel in one more all-gather
    assert seen["broadcast"] == nb and seen["all_gather"] == nb - 1 + 1 and seen["all_reduce"] == nsb - 1
    comm.close()
