"""One GP fit sharded over the GPUs of a node (north_star: "the N x N Gram and its Cholesky
shard row-block across the 8 GPUs of one node with RCCL over xGMI for the panel broadcasts").

The schedule, the block arithmetic and the RCCL calls all live in the HIP library (albatross_amd/csrc/shard*.hip,
C-ABI `agp_comm_*` / `agp_sharded_fit_*` in include/albatross_amd.h); this module is the thin host mirror:

    comm = Communicator.from_torch(ctx)            # one rank per GPU; the 128-byte RCCL id travels over the
                                                   # process group that is already there (gloo or nccl)
    sharded = ShardedGaussianProcessFit(ctx, cov, comm)
    result = sharded.fit(x, y)                     # every rank: full information vector + log-determinant
    fit_model = sharded.replicate(model)           # every rank: an ordinary FitModel; predict YOUR share of the
                                                   # test points (gp.hpp:82-113 is independent per test point)

Layout (ShardLayout mirrors csrc/shard.h ShardPlan): row blocks of 512 dealt to the ranks in snake order; per block
column a broadcast of the factored diagonal block, a panel solve on every rank's own rows, an all-gather of the panel
and the MFMA updates of the own rows, with one block column of look-ahead (see the header).
"""
import ctypes as C

import numpy as np

from . import _capi as capi

BLOCK = 512


class ShardLayout:
    """Row-block-cyclic ownership arithmetic, evaluated by the library (agp_shard_*)."""

    def __init__(self, n, world, block=BLOCK):
        self.n, self.world, self.block = int(n), int(world), int(block)
        self.n_blocks = (self.n + self.block - 1) // self.block
        self._lib = capi.load()

    def owner(self, b):
        return self._lib.agp_shard_owner(b, self.world)

    def width(self, b):
        return min(self.block, self.n - b * self.block)

    def local_rows(self, rank):
        return self._lib.agp_shard_local_rows(self.n, self.block, self.world, rank)

    def global_rows(self, rank):
        """global row index of every local row of `rank`, in local order"""
        blocks = [b for b in range(self.n_blocks) if self.owner(b) == rank]
        if not blocks:
            return np.zeros(0, dtype=np.int64)
        return np.concatenate([np.arange(b * self.block, b * self.block + self.width(b), dtype=np.int64) for b in blocks])



class Communicator:
    """agp_comm: the transport of the sharded fit.  RCCL (`rccl`, `from_torch`) or caller-supplied collectives on host
    arrays (`callbacks`, `torch_callbacks`: tests, and one-GPU boxes where RCCL refuses two ranks per device)."""

    def __init__(self, handle, keepalive=None):
        self._lib = capi.load()
        self._h = handle
        self._keep = keepalive
        self._ctx = None

    # ---- constructors ----
    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(capi.COMM_ID_BYTES)
        st = capi.load().agp_comm_unique_id(buf)
        if st != capi.AGP_OK:
            raise RuntimeError("agp_comm_unique_id failed: librccl is not available")
        return buf.raw

    @classmethod
    def rccl(cls, ctx, world, rank, unique_id):
        h = C.c_void_p()
        idbuf = C.create_string_buffer(bytes(unique_id), capi.COMM_ID_BYTES)
        ctx._check(ctx._lib.agp_comm_create(ctx._h, world, rank, idbuf, C.byref(h)), "agp_comm_create")
        c = cls(h)
        c._ctx = ctx  # the RCCL transport dereferences its context until it is destroyed: keep it alive, and see close()
        return c

    @staticmethod
    def _callback_struct(world, broadcast, all_gather, all_reduce):
        def view(ptr, count):
            return np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_double)), shape=(int(count),))

        def guard(fn):
            def wrapped(*a):
                try:
                    fn(*a)
                    return 0
                except Exception:  # noqa: BLE001 - an exception must not unwind through the C caller
                    import traceback
                    traceback.print_exc()
                    return 1
            return wrapped

        b = capi.BROADCAST_FN(guard(lambda user, buf, count, root: broadcast(view(buf, count), int(root))))
        g = capi.ALL_GATHER_FN(guard(lambda user, send, recv, count: all_gather(view(send, count), view(recv, count * world))))
        r = capi.ALL_REDUCE_FN(guard(lambda user, buf, count, op: all_reduce(view(buf, count), int(op))))
        cbs = capi.CommCallbacks(None, b, g, r)
        return cbs, (b, g, r, cbs)

    @classmethod
    def callbacks(cls, world, rank, broadcast, all_gather, all_reduce):
        """broadcast(buf: ndarray, root), all_gather(send: ndarray, recv: ndarray), all_reduce(buf: ndarray, op: 0 sum |
        1 max): Python callables working IN PLACE on float64 arrays"""
        cbs, keep = cls._callback_struct(world, broadcast, all_gather, all_reduce)
        h = C.c_void_p()
        st = capi.load().agp_comm_create_callbacks(world, rank, C.byref(cbs), C.byref(h))
        if st != capi.AGP_OK:
            raise RuntimeError(f"agp_comm_create_callbacks failed ({st})")
        return cls(h, keepalive=keep)

    @staticmethod
    def _torch_collectives(group=None):
        import torch
        import torch.distributed as dist
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        src = (lambda r: dist.get_global_rank(group, r)) if group is not None else (lambda r: r)

        def broadcast(buf, root):
            t = torch.from_numpy(buf)
            dist.broadcast(t, src=src(root), group=group)

        def all_gather(send, recv):
            dist.all_gather_into_tensor(torch.from_numpy(recv), torch.from_numpy(send.copy()), group=group)

        def all_reduce(buf, op):
            dist.all_reduce(torch.from_numpy(buf), op=dist.ReduceOp.MAX if op == 1 else dist.ReduceOp.SUM, group=group)

        return world, rank, broadcast, all_gather, all_reduce

    @classmethod
    def torch_callbacks(cls, group=None):
        """collectives of an initialised torch.distributed process group on CPU tensors (gloo)"""
        return cls.callbacks(*cls._torch_collectives(group))

    @classmethod
    def ipc(cls, ctx, group=None, mailbox_doubles=0):
        """agp_comm_create_ipc: the device-asynchronous transport between processes sharing ONE GPU (peer stores through
        hipIpc mailboxes + stream-ordered flags; csrc/shard_ipc.hip).  The torch group (gloo) only carries the IPC
        handles and the control plane.  Collective."""
        world, rank, broadcast, all_gather, all_reduce = cls._torch_collectives(group)
        cbs, keep = cls._callback_struct(world, broadcast, all_gather, all_reduce)
        h = C.c_void_p()
        ctx._check(ctx._lib.agp_comm_create_ipc(ctx._h, world, rank, C.byref(cbs), int(mailbox_doubles), C.byref(h)), "agp_comm_create_ipc")
        c = cls(h, keepalive=keep)
        c._ctx = ctx
        return c

    @classmethod
    def from_torch(cls, ctx, group=None, transport="rccl"):
        """One communicator over the ranks of an initialised torch.distributed group.  transport="rccl": the library's
        own RCCL communicator (the unique id is broadcast over the torch group, whatever its backend);
        "callbacks": the torch group's collectives on host arrays; "ipc": device-asynchronous
        peer stores between processes on one GPU (tests)."""
        import torch.distributed as dist
        if transport == "callbacks":
            return cls.torch_callbacks(group)
        if transport == "ipc":
            return cls.ipc(ctx, group)
        world, rank = dist.get_world_size(group), dist.get_rank(group)
        box = [cls.unique_id() if rank == 0 else None]
        dist.broadcast_object_list(box, src=(dist.get_global_rank(group, 0) if group is not None else 0), group=group)
        return cls.rccl(ctx, world, rank, box[0])

    # ---- use ----
    @property
    def world(self):
        return self._lib.agp_comm_size(self._h)

    @property
    def rank(self):
        return self._lib.agp_comm_rank(self._h)

    def all_reduce(self, values, op="sum"):
        a = np.ascontiguousarray(values, dtype=np.float64).copy()
        st = self._lib.agp_comm_all_reduce_host(self._h, C.c_void_p(a.ctypes.data), a.size, 1 if op == "max" else 0)
        if st != capi.AGP_OK:
            raise RuntimeError(f"agp_comm_all_reduce_host failed ({st})")
        return a

    def barrier(self):
        if self._lib.agp_comm_barrier(self._h) != capi.AGP_OK:
            raise RuntimeError("agp_comm_barrier failed")

    def close(self):
        if getattr(self, "_h", None):
            ctx = getattr(self, "_ctx", None)
            if ctx is None or ctx._h:  # a context that is already closed took its streams with it: nothing left to destroy safely
                self._lib.agp_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ShardedFitResult:
    def __init__(self, information, log_determinant, layout):
        self.information = information
        self.log_determinant = log_determinant
        self.layout = layout


class ShardedGaussianProcessFit:
    """`Fit<GPFit<...>>` (models/gp.hpp:61-69) of ONE dataset over all ranks of a Communicator (None: one rank).  Every
    rank passes the same (full) features and targets and receives the full information vector and log-determinant."""

    def __init__(self, ctx, cov, comm=None):
        self.ctx, self.cov, self.comm = ctx, cov, comm
        self._h = None
        self._features = None

    def _release(self):
        if self._h is not None and self.ctx._h:
            self.ctx._lib.agp_sharded_fit_destroy(self._h)
        self._h = None

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def fit(self, features, targets_mean, targets_variance=None, features_struct=None, device_targets=None):
        """features_struct / device_targets: a ready `_capi.Features` view and target pointer (bench.py: inputs already
        resident in HBM); otherwise host arrays."""
        from .gp import NotPositiveDefiniteError
        ctx, lib = self.ctx, self.ctx._lib
        self._release()
        if features_struct is not None:
            s, n = features_struct, int(features_struct.n)
            yp = C.c_void_p(device_targets)
            vp = None
            keep = None
        else:
            fs = self.cov.features(features)
            s, n = fs.as_struct(), fs.n
            y = np.ascontiguousarray(targets_mean, dtype=np.float64)
            if y.shape[0] != n:
                raise ValueError("features and targets differ in size")
            yv = None if targets_variance is None else np.ascontiguousarray(targets_variance, dtype=np.float64)
            yp = C.c_void_p(y.ctypes.data)
            vp = None if yv is None else C.c_void_p(yv.ctypes.data)
            keep = (fs, y, yv)
        info = np.empty(n)
        logdet = C.c_double()
        h = C.c_void_p()
        st = lib.agp_sharded_fit_create(ctx._h, None if self.comm is None else self.comm._h, ctx.kernel(self.cov), C.byref(s), yp,
                                        vp, C.byref(h), C.c_void_p(info.ctypes.data), C.byref(logdet))
        del keep
        if st == capi.AGP_ERR_NOT_POSITIVE_DEFINITE:
            pivot = lib.agp_sharded_fit_failed_pivot(h) if h else -1
            if h:
                lib.agp_sharded_fit_destroy(h)
            raise NotPositiveDefiniteError(st, f"sharded fit (pivot {pivot})")
        ctx._check(st, "agp_sharded_fit_create")
        self._h = h
        self._features = features
        world = 1 if self.comm is None else self.comm.world
        return ShardedFitResult(info, logdet.value, ShardLayout(n, world))

    def stage(self, index):
        v = C.c_double()
        self.ctx._check(self.ctx._lib.agp_sharded_fit_stage(self._h, index, C.byref(v)), "agp_sharded_fit_stage")
        return v.value

    def predict_marginal(self, features):
        """gp_marginal_prediction (gp.hpp:87-101) straight from the sharded factor, WITHOUT replicating it
        (agp_sharded_predict_marginal: distributed forward substitution, one broadcast per block row).  Collective:
        every rank passes the same test features and receives (mean, variance) of all of them.  ZeroMean models."""
        fs = self.cov.features(features)
        s = fs.as_struct()
        mean, var = np.empty(fs.n), np.empty(fs.n)
        self.ctx._check(self.ctx._lib.agp_sharded_predict_marginal(self.ctx._h, self.ctx.kernel(self.cov), self._h, C.byref(s),
                                                                   C.c_void_p(mean.ctypes.data), C.c_void_p(var.ctypes.data), capi.HOST),
                        "agp_sharded_predict_marginal")
        return mean, var

    def predict_joint(self, features):
        """gp_joint_prediction (gp.hpp:103-113) from the sharded factor, not replicated (agp_sharded_predict_joint): every
        rank multiplies its own rows of V = L^-1 K*, one all-reduce of the m x m product.  Collective: every rank passes the
        same test features and receives (mean, covariance)."""
        fs = self.cov.features(features)
        s = fs.as_struct()
        mean, cov = np.empty(fs.n), np.empty((fs.n, fs.n), order="F")
        self.ctx._check(self.ctx._lib.agp_sharded_predict_joint(self.ctx._h, self.ctx.kernel(self.cov), self._h, C.byref(s),
                                                                C.c_void_p(mean.ctypes.data), C.c_void_p(cov.ctypes.data), capi.HOST),
                        "agp_sharded_predict_joint")
        return mean, cov

    def replicate(self, model):
        """All-gather the factor: every rank gets an ordinary FitModel of `model` (a GaussianProcessRegression with this
        covariance function) and predicts its own share of the test points."""
        from .gp import FitModel, GPFit
        h = C.c_void_p()
        self.ctx._check(self.ctx._lib.agp_sharded_fit_replicate(self.ctx._h, self._h, C.byref(h)), "agp_sharded_fit_replicate")
        n = int(self.ctx._lib.agp_fit_size(h))
        return FitModel(model, GPFit(self.ctx, h, n, self._features))
