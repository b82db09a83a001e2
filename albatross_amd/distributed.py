"""One GP fit sharded over the GPUs of a node (north_star:
"the N x N Gram and its Cholesky shard ... across the 8 GPUs of one node with
RCCL over xGMI for the panel broadcasts").

Layout: the lower triangle of K is cut into block columns of `block` (512)
columns; block column c (rows c*block .. n) is dealt to the ranks in snake
order 0..G-1, G-1..0, ... (block-column-cyclic: balances the shrinking
trailing matrix).  Every rank
builds the Gram entries of its own block columns locally (no communication),
then for every block column c, in order:

    owner(c):  panel factorisation of its (already fully updated) block column
               — POTRF / TRSM / inner updates, with the fused forward
               substitution on y — and ONE broadcast of the sub-diagonal panel
               (+ the running y and two status words) to all ranks;
    all ranks: C' -= P P'^T on every block column c' > c they own (fp64 MFMA
               update kernel).

The information vector follows by a right-looking back substitution with one
small broadcast (the solved 512 entries) per block column.  The panel
broadcast is the path's only real exchange step; everything else is local.

All arithmetic is done by the HIP library through the block-level C-ABI
(`agp_blk_*`, include/albatross_amd.h); this module only sequences launches and
`torch.distributed` collectives (backend "nccl" = RCCL on GPUs).  The
sequencing is backend-agnostic: tests/test_distributed_cpu.py runs it on CPU
tensors over gloo with a numpy implementation of the same block interface.
"""
import ctypes as C
import os

import numpy as np
import torch  # noqa: F401  (imported BEFORE the HIP library is loaded, see _init_torch_first)

from . import _capi as capi


def _init_torch_first():
    """The torch wheel carries its own HIP runtime (ROCm 7.0, soname
    libamdhip64.so) next to the system one this library links (ROCm 7.2,
    libamdhip64.so.7).  Both can live in one process, but only if torch's is
    initialised first; so any process that shares buffers between torch and
    this library initialises torch's CUDA context before the first agp_* call."""
    if torch.cuda.is_available():
        torch.cuda.init()


_init_torch_first()

IMG_DOUBLES = 36 * 16 * 16  # tile image of one 128 x 128 diagonal block
NB = 128


def _round_ld(rows):
    ld = (max(rows, 1) + 7) // 8 * 8
    if ld % 256 == 0:
        ld += 8
    return ld


class ShardLayout:
    """Block-column-cyclic ownership arithmetic (pure host logic)."""

    def __init__(self, n, world, block=512):
        if block % NB != 0 or block <= 0:
            raise ValueError("block must be a positive multiple of 128")
        self.n, self.world, self.block = int(n), int(world), int(block)
        self.n_blocks = (self.n + self.block - 1) // self.block

    def owner(self, c):
        # boustrophedon ("snake") cyclic order 0..G-1, G-1..0, ...: block columns
        # shrink with c, and the snake gives every rank the same number of rows
        # per pair of rounds (plain c mod G leaves rank 0 with ~1.5x the work of
        # rank G-1 at 32 block columns over 8 ranks)
        r, rnd = c % self.world, c // self.world
        return r if rnd % 2 == 0 else self.world - 1 - r

    def start(self, c):
        return c * self.block

    def width(self, c):
        return min(self.block, self.n - c * self.block)

    def rows(self, c):  # rows of block column c that are stored: c*block .. n
        return self.n - c * self.block

    def owned(self, rank):
        return [c for c in range(self.n_blocks) if self.owner(c) == rank]

    def local_elements(self, rank):
        return sum(_round_ld(self.rows(c)) * self.width(c) for c in self.owned(rank))


class HipBlockOps:
    """Block interface on one GPU: torch CUDA tensors for storage, the C-ABI
    (`agp_blk_*`) for every arithmetic step.  No CPU fallback."""

    def __init__(self, ctx, device):
        import torch
        self.torch = torch
        self.ctx = ctx
        self.lib = ctx._lib
        self.device = torch.device(device)

    # ---- storage / plumbing ----
    def empty(self, count):
        return self.torch.empty(int(count), dtype=self.torch.float64, device=self.device)

    def from_host(self, array):
        return self.torch.from_numpy(np.ascontiguousarray(array, dtype=np.float64)).to(self.device)

    def to_host(self, tensor):
        return tensor.detach().cpu().numpy()

    def _p(self, tensor, offset=0):
        return C.c_void_p(tensor.data_ptr() + 8 * int(offset))

    def sync(self):
        """drain the library's streams and torch's current stream"""
        self.ctx.synchronize()
        self.torch.cuda.current_stream(self.device).synchronize()

    def pack_panel(self, col, lda, m, width, buf, ldp):
        """buf[ldp x width] <- rows width..m of the block column (strided device copy)"""
        if m - width <= 0:
            return
        t = self.torch
        src = t.as_strided(col, (m - width, width), (1, lda), storage_offset=col.storage_offset() + width)
        dst = t.as_strided(buf, (m - width, width), (1, ldp), storage_offset=buf.storage_offset())
        dst.copy_(src)

    # ---- arithmetic: C-ABI ----
    def _check(self, st, what):
        self.ctx._check(st, what)

    def gram_block(self, cov, rows_fs, cols_fs, out, ld, diag_add, diag_offset):
        nan = C.c_int(0)
        rs, cs = rows_fs.as_struct(), cols_fs.as_struct()
        self._check(self.lib.agp_blk_gram(self.ctx._h, self.ctx.kernel(cov), C.byref(rs), C.byref(cs), self._p(out), ld,
                                          None if diag_add is None else self._p(diag_add, diag_offset), C.byref(nan)),
                    "agp_blk_gram")
        return nan.value

    def panel_factor(self, col, m, lda, width, img, y):
        bad, logsum = C.c_int64(-1), C.c_double(0.)
        self._check(self.lib.agp_blk_panel_factor(self.ctx._h, self._p(col), m, lda, width, self._p(img), self._p(y),
                                                  C.byref(bad), C.byref(logsum)), "agp_blk_panel_factor")
        return bad.value, logsum.value

    def update(self, col, ldc, buf, row_offset, ldp, M, N, K):
        self._check(self.lib.agp_blk_update(self.ctx._h, self._p(col), ldc, self._p(buf, row_offset), ldp,
                                            self._p(buf, row_offset), ldp, M, N, K, 1), "agp_blk_update")

    def back_diag(self, col, lda, width, img, z):
        self._check(self.lib.agp_blk_back_diag(self.ctx._h, self._p(col), lda, width, self._p(img), self._p(z)),
                    "agp_blk_back_diag")

    def back_update(self, col, row_offset, lda, nrows, ncols, x, z):
        self._check(self.lib.agp_blk_back_update(self.ctx._h, self._p(col, row_offset), lda, nrows, ncols, self._p(x),
                                                 self._p(z)), "agp_blk_back_update")


class ShardedFitResult:
    def __init__(self, information, log_determinant, layout):
        self.information = information
        self.log_determinant = log_determinant
        self.layout = layout


class ShardedGaussianProcessFit:
    """`Fit<GPFit<...>>` (models/gp.hpp:61-69) of ONE dataset over all ranks of
    a process group.  Every rank passes the same (full) features and targets and
    receives the full information vector and log-determinant."""

    def __init__(self, ops, cov, block=512, group=None, force_collectives=False):
        # force_collectives: issue the broadcasts / all-reduce even on a single
        # rank (used by the GPU test to exercise RCCL on buffers the HIP library wrote)
        self.force_collectives = force_collectives
        self.ops = ops
        self.cov = cov
        self.block = block
        self.group = group
        import torch.distributed as dist
        self.dist = dist
        self.active = dist.is_available() and dist.is_initialized()
        self.rank = dist.get_rank(group) if self.active else 0
        self.world = dist.get_world_size(group) if self.active else 1
        self._store = None

    # ---- collectives (no-ops on a single rank) ----
    def _src(self, owner):
        return self.dist.get_global_rank(self.group, owner) if (self.active and self.group is not None) else owner

    def _broadcast(self, tensor, owner):
        if self.active and (self.world > 1 or self.force_collectives):
            self.ops.sync()
            self.dist.broadcast(tensor, src=self._src(owner), group=self.group)
            self.ops.sync()

    def _broadcast_start(self, tensor, owner):
        """asynchronous broadcast (the caller keeps working on OTHER buffers); returns a handle for _broadcast_finish"""
        if self.active and (self.world > 1 or self.force_collectives):
            self.ops.sync()
            return self.dist.broadcast(tensor, src=self._src(owner), group=self.group, async_op=True)
        return None

    def _broadcast_finish(self, work):
        if work is not None:
            work.wait()
            self.ops.sync()

    def _all_max(self, value):
        if not (self.active and (self.world > 1 or self.force_collectives)):
            return value
        t = self.ops.from_host(np.array([float(value)]))
        self.ops.sync()
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX, group=self.group)
        self.ops.sync()
        return float(self.ops.to_host(t)[0])

    def _allocate(self, lay):
        """block-column storage, tile images and the panel buffer (re-used across fits of one size)"""
        key = (lay.n, lay.world, lay.block)
        if self._store is not None and self._store["key"] == key:
            return self._store
        ops = self.ops
        st = {"key": key, "cols": {}, "img": {}, "ld": {}}
        for c in lay.owned(self.rank):
            st["ld"][c] = _round_ld(lay.rows(c))
            st["cols"][c] = ops.empty(st["ld"][c] * lay.width(c))
            st["img"][c] = ops.empty(((lay.width(c) + NB - 1) // NB) * IMG_DOUBLES)
        st["ldp"] = _round_ld(max(lay.n - lay.block, 1))
        st["buf"] = ops.empty(st["ldp"] * lay.block + lay.n + 8)
        st["buf2"] = ops.empty(st["ldp"] * lay.block + lay.n + 8)  # look-ahead: panel c + 1 travels while panel c is applied
        st["y"] = ops.empty(lay.n + 8)
        st["z"] = ops.empty(lay.n + 8)
        self._store = st
        return st

    def fit(self, features, targets_mean, targets_variance=None):
        from .covariance import FeatureSet
        from .gp import NanInputError, NotPositiveDefiniteError
        ops, cov = self.ops, self.cov
        fs = cov.features(features)
        n = fs.n
        lay = ShardLayout(n, self.world, self.block)
        st = self._allocate(lay)
        y_host = np.ascontiguousarray(targets_mean, dtype=np.float64)
        if y_host.shape[0] != n:
            raise ValueError("features and targets differ in size")
        yvar = None if targets_variance is None else ops.from_host(targets_variance)

        def slice_fs(lo, hi):  # measurement-wrapped slice (as_measurements, gp.hpp:288)
            return FeatureSet(fs.coords[lo:hi], None if fs.scales is None else list(fs.scales[lo:hi].T),
                              None if fs.eq_id is None else fs.eq_id[lo:hi], True)

        # ---- 1. local Gram of the owned block columns (no communication) ----
        nan = 0
        for c in lay.owned(self.rank):
            s0, w = lay.start(c), lay.width(c)
            nan |= ops.gram_block(cov, slice_fs(s0, n), slice_fs(s0, s0 + w), st["cols"][c], st["ld"][c], yvar, s0)
        if self._all_max(nan) > 0:
            raise NanInputError(capi.AGP_ERR_NAN_INPUT, "sharded fit")

        # ---- 2. right-looking LL^T, one panel broadcast per block column, ONE block column of look-ahead:
        # as soon as panel c has arrived, the owner of block column c + 1 applies it to THAT column only, factors
        # panel c + 1 and starts broadcasting it; everybody applies panel c to the rest of their columns while
        # panel c + 1 travels.  The serial part of a step is one narrow update + one panel phase, the broadcast
        # overlaps the bulk of the updates.  (AGP_SHARDED_LOOKAHEAD=0: the synchronous schedule.) ----
        y_cur = st["y"]
        y_cur[:n].copy_(ops.from_host(y_host))
        ops.sync()
        z_local = st["z"]         # z entries of the owned block columns, at their global index
        bufs, ldp = [st["buf"], st["buf2"]], st["ldp"]
        lookahead = os.environ.get("AGP_SHARDED_LOOKAHEAD", "1") != "0"

        def used_of(c):
            mp_ = lay.rows(c) - lay.width(c)
            return (ldp * lay.width(c) + mp_ + 2) if mp_ > 0 else 2

        def factor_and_pack(c, log_sum_before):
            """owner of c: panel phase on the (fully updated) block column, panel + running y + status into its buffer"""
            s0, w, m = lay.start(c), lay.width(c), lay.rows(c)
            mp_, buf = m - w, bufs[c % 2]
            ycol = y_cur[:m]   # y_cur[i] belongs to training row start(c) + i
            bad, lsum = ops.panel_factor(st["cols"][c], m, st["ld"][c], w, st["img"][c], ycol)
            ops.sync()
            z_local[s0:s0 + w].copy_(ycol[:w])
            if mp_ > 0:
                ops.pack_panel(st["cols"][c], st["ld"][c], m, w, buf, ldp)
                buf[ldp * w:ldp * w + mp_].copy_(ycol[w:m])
            tail = used_of(c) - 2
            status = np.array([float(s0 + bad) if bad >= 0 else -1., log_sum_before + lsum])
            buf[tail:tail + 2].copy_(ops.from_host(status))

        def apply_panel(c, c2):
            """block column c2 (> c) -= panel c"""
            r0 = lay.start(c2) - (lay.start(c) + lay.width(c))  # first panel row that meets block column c2
            ops.update(st["cols"][c2], st["ld"][c2], bufs[c % 2], r0, ldp, lay.rows(c2), lay.width(c2), lay.width(c))

        log_sum = 0.
        if self.rank == lay.owner(0):
            factor_and_pack(0, 0.)
        self._broadcast(bufs[0][:used_of(0)], lay.owner(0))
        for c in range(lay.n_blocks):
            w, m = lay.width(c), lay.rows(c)
            mp = m - w  # rows of the sub-diagonal panel
            buf = bufs[c % 2]
            tail = used_of(c) - 2
            status = ops.to_host(buf[tail:tail + 2])
            if status[0] >= 0:
                raise NotPositiveDefiniteError(capi.AGP_ERR_NOT_POSITIVE_DEFINITE, f"sharded fit (pivot {int(status[0])})")
            log_sum = float(status[1])
            if mp <= 0:
                break
            # the running y travels with the panel: rows start(c + 1) ..
            y_cur[:mp].copy_(buf[ldp * w:ldp * w + mp])
            nxt = c + 1
            mine = [c2 for c2 in lay.owned(self.rank) if c2 > c]
            work = None
            if lookahead:
                if self.rank == lay.owner(nxt):
                    apply_panel(c, nxt)
                    ops.sync()
                    factor_and_pack(nxt, log_sum)
                    mine = [c2 for c2 in mine if c2 != nxt]
                work = self._broadcast_start(bufs[nxt % 2][:used_of(nxt)], lay.owner(nxt))
            for c2 in mine:
                apply_panel(c, c2)
            ops.sync()
            if lookahead:
                self._broadcast_finish(work)
            else:
                if self.rank == lay.owner(nxt):
                    factor_and_pack(nxt, log_sum)
                self._broadcast(bufs[nxt % 2][:used_of(nxt)], lay.owner(nxt))

        # ---- 3. information = L^-T z, right-looking, one small broadcast per block ----
        xbuf = ops.empty(lay.block + 8)
        alpha = ops.empty(n)
        for c in range(lay.n_blocks - 1, -1, -1):
            owner, s0, w = lay.owner(c), lay.start(c), lay.width(c)
            if self.rank == owner:
                xbuf[:w].copy_(z_local[s0:s0 + w])
                ops.sync()
                ops.back_diag(st["cols"][c], st["ld"][c], w, st["img"][c], xbuf)
                ops.sync()
            self._broadcast(xbuf[:w], owner)
            alpha[s0:s0 + w].copy_(xbuf[:w])
            ops.sync()
            for c2 in lay.owned(self.rank):
                if c2 >= c:
                    continue
                # rows of block c inside block column c2 start at local row start(c) - start(c2)
                ops.back_update(st["cols"][c2], s0 - lay.start(c2), st["ld"][c2], w, lay.width(c2), xbuf,
                                z_local[lay.start(c2):lay.start(c2) + lay.width(c2)])
            ops.sync()
        return ShardedFitResult(ops.to_host(alpha), 2. * log_sum, lay)
