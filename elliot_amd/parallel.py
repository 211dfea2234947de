"""Multi-GPU execution -- new design, the reference is single-device.

Two ways to cut the BPR-MF step over G ranks live here: by USER (ShardedBprmfByUser: user rows sharded, item table
replicated, all-reduce of the 51 MB item gradient; what bench.py uses) and by ITEM (north_star's formulation, below).

Item shards:

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the MI355X node, "gloo" in the CPU
tests).  Rank r owns the item range [lo_r, hi_r): its rows of Gi / Bi and their optimiser state; the user table is
replicated.

  top-k     every rank scores each user block against its item slice (global ids through `item_offset`), the
            [Ub, k] partial lists are ALL-GATHERED (8*Ub*k bytes per rank) and merged with the single-GPU ordering
            rule -> identical lists to one GPU.
  training  every rank draws its own triplets with positive AND negative inside its shard (shard-local negative
            sampling, north_star), computes loss / item-row gradients locally.  User-row gradients are exchanged in one
            of two ways (pick_exchange): "rows" -- one gradient row per triplet, the (user id, row) pairs are
            ALL-GATHERED, every rank reduces them by user in the same order and applies the optimiser to its replica of
            the user table; "dense" -- REDUCE-SCATTER of the dense gradient table, optimiser on the rank's own user rows,
            ALL-GATHER of the updated rows (ShardedBprmfDense; the choice when B per rank exceeds 2U/G).  G ranks x B triplets are
            mathematically one step on the concatenated batch (sum-loss, gradients add).  Deviation from the reference
            for G > 1 (documented): the triplet distribution is the shard-restricted one, not custom_sampler.py:31-42's.

The numeric work is behind a small backend object so that the collective logic can be exercised on CPU with gloo:
`HipBackend` (product; kernels of libelliot_hip.so) -- tests inject a NumPy backend built from oracle/.
"""
import numpy as np
import torch

from . import ops


def item_range(n_items, rank, world):
    return (n_items * rank) // world, (n_items * (rank + 1)) // world


def shard_csr(indptr, indices, lo, hi):
    """Train CSR restricted to item columns [lo, hi) with LOCAL column ids (torch tensors, any device)."""
    keep = (indices >= lo) & (indices < hi)
    csum = torch.zeros(indices.shape[0] + 1, dtype=torch.int64, device=indices.device)
    csum[1:] = torch.cumsum(keep.to(torch.int64), 0)
    new_indptr = csum[indptr]
    new_indices = (indices[keep] - lo).to(torch.int32)
    return new_indptr.contiguous(), new_indices.contiguous()


class _Collectives:
    """all_gather along dim 0 with equal shapes on every rank (None / world 1 = identity)."""

    def __init__(self, group_world=None):
        import torch.distributed as dist
        self.dist = dist if (dist.is_available() and dist.is_initialized()) else None
        self.world = self.dist.get_world_size() if self.dist else 1
        self.rank = self.dist.get_rank() if self.dist else 0
        import os
        # EL_FORCE_COLLECTIVES=1: call the backend even with one rank (API check of the RCCL path on a 1-GPU box)
        self.always = self.dist is not None and os.environ.get("EL_FORCE_COLLECTIVES") == "1"

    def all_gather(self, t):
        if self.world == 1 and not self.always:
            return t
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t.contiguous())
        return out

    def all_reduce_sum(self, t, async_op=False):
        """In-place sum over the ranks.  async_op: returns the work handle (None when nothing is in flight); kernels the
        caller enqueues before `wait()` overlap the collective (RCCL runs it on its own stream)."""
        if self.world > 1 or self.always:
            work = self.dist.all_reduce(t, async_op=async_op)
            return work if async_op else t
        return None if async_op else t

    def reduce_scatter_rows(self, out, full):
        """out [n, ...] = rank-th row block of the element-wise sum of `full` [world * n, ...] over the ranks."""
        n = out.shape[0]
        if self.world == 1 and not self.always:
            out.copy_(full[:n])
            return out
        if self.dist.get_backend() == "gloo":                      # gloo has no reduce-scatter: all-reduce a copy, slice
            tmp = full.clone()
            self.dist.all_reduce(tmp)
            out.copy_(tmp[self.rank * n:(self.rank + 1) * n])
            return out
        self.dist.reduce_scatter_tensor(out, full)                  # RCCL reduce-scatter over xGMI
        return out

    def all_gather_rows_into(self, full, part, async_op=False):
        """full [world * n, ...] = concatenation of every rank's `part` [n, ...] (part may be full's own row block: the
        in-place form).  async_op: returns the work handle (None when nothing is in flight); the caller `wait()`s
        before it touches `full` again, kernels enqueued meanwhile overlap the transfer."""
        if self.world == 1 and not self.always:
            if full.data_ptr() != part.data_ptr():
                full[:part.shape[0]].copy_(part)
            return None if async_op else full
        work = self.dist.all_gather_into_tensor(full, part.contiguous(), async_op=async_op)
        return work if async_op else full


class _AbiWork:
    """Handle of a collective issued on the communication stream: wait() makes the CURRENT stream wait for it."""

    def __init__(self, event):
        self.event = event

    def wait(self):
        torch.cuda.current_stream().wait_event(self.event)


class RcclAbiCollectives:
    """The same four collectives as _Collectives, issued through the C ABI of libelliot_hip.so (el_comm_init /
    el_allreduce_rows / el_reduce_scatter_rows / el_allgather_rows / el_allgather_topk: RCCL called directly) instead
    of torch.distributed -- what a C host of the library runs.  torch is only the buffer holder here.  One communicator per
    rank; the 128-byte RCCL id travels out of band (`exchange_id`: a TCPStore next to MASTER_PORT, or any callable).
    Collectives run on a side stream ordered after the caller's stream; async_op=True returns a handle whose wait() orders
    the caller's stream after the collective (kernels enqueued in between overlap it)."""

    def __init__(self, ctx, rank, world, unique_id=None, exchange_id=None):
        import ctypes as C
        self.ctx, self.rank, self.world = ctx, int(rank), int(world)
        self.always = True
        self.dist = None
        if unique_id is None:
            unique_id = (exchange_id or self._store_exchange)(self._make_id() if self.rank == 0 else None)
        self._id = C.create_string_buffer(bytes(unique_id), 128)
        h = C.c_void_p()
        torch.cuda.set_device(ctx.device)
        ops.check(ctx.lib.el_comm_init(ctx.handle, self._id, self.rank, self.world, C.byref(h)), "el_comm_init")
        self._h = h
        self.stream = torch.cuda.Stream(device=ctx.device)

    def _make_id(self):
        import ctypes as C
        buf = C.create_string_buffer(128)
        ops.check(self.ctx.lib.el_comm_unique_id(buf), "el_comm_unique_id")
        return buf.raw

    def _store_exchange(self, mine):
        """Rank 0 publishes the id in a TCPStore on MASTER_ADDR : MASTER_PORT + 17 (host sockets: control plane only)."""
        import os
        from datetime import timedelta
        if self.world == 1:
            return mine
        from torch.distributed import TCPStore
        store = TCPStore(os.environ.get("MASTER_ADDR", "127.0.0.1"), int(os.environ.get("MASTER_PORT", "29500")) + 17, self.world,
                         self.rank == 0, timeout=timedelta(seconds=120))
        if self.rank == 0:
            store.set("el_comm_id", mine)
            return mine
        return store.get("el_comm_id")

    def close(self):
        if getattr(self, "_h", None) is not None:
            torch.cuda.synchronize(self.ctx.device)
            self.ctx.lib.el_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _run(self, fn, async_op, tensors=()):
        """tensors: every buffer the collective reads or writes on the SIDE stream.  They are (a) recorded on that stream, so that the
        caching allocator does not hand a block freed on the caller's stream (a `.contiguous()` temporary, an output the caller
        drops early) to the next allocation while the collective still uses it, and (b) kept alive by the returned handle."""
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        self.stream.wait_event(ready)                              # inputs written on the caller's stream are complete
        for t in tensors:
            t.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            fn(self.stream.cuda_stream)
            done = torch.cuda.Event()
            done.record(self.stream)
        work = _AbiWork(done)
        work.keep = tuple(tensors)
        if async_op:
            return work
        work.wait()
        return None

    def all_reduce_sum(self, t, async_op=False):
        if t.dtype != torch.float32:
            # the fp64 loss scalar: gather the ranks' values (el_allgather_rows moves bytes) and add them here
            parts = self.all_gather(t.reshape(1, -1)).reshape(self.world, -1)
            t.copy_(parts.sum(0).reshape(t.shape))
            return None if async_op else t
        assert t.is_contiguous()
        w = self._run(lambda s: ops.check(self.ctx.lib.el_allreduce_rows(self.ctx.handle, self._h, s, t.data_ptr(), t.numel()),
                                          "el_allreduce_rows"), async_op, (t,))
        return w if async_op else t

    def all_gather(self, t):
        t = t.contiguous()
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self._run(lambda s: ops.check(self.ctx.lib.el_allgather_rows(self.ctx.handle, self._h, s, t.data_ptr(), out.data_ptr(),
                                                                     t.numel() * t.element_size()), "el_allgather_rows"), False, (t, out))
        return out

    def all_gather_topk(self, idx, val):
        """[n, k] partial lists of every rank -> ([world, n, k] ids, [world, n, k] scores) in one RCCL group."""
        n, k = idx.shape
        gi = torch.empty((self.world, n, k), dtype=torch.int32, device=idx.device)
        gv = torch.empty((self.world, n, k), dtype=torch.float32, device=idx.device)
        self._run(lambda s: ops.check(self.ctx.lib.el_allgather_topk(self.ctx.handle, self._h, s, idx.data_ptr(), val.data_ptr(), int(n),
                                                                     int(k), gi.data_ptr(), gv.data_ptr()), "el_allgather_topk"), False,
                  (idx, val, gi, gv))
        return gi, gv

    def reduce_scatter_rows(self, out, full):
        assert out.is_contiguous() and full.is_contiguous() and full.numel() == self.world * out.numel()
        self._run(lambda s: ops.check(self.ctx.lib.el_reduce_scatter_rows(self.ctx.handle, self._h, s, full.data_ptr(), out.data_ptr(),
                                                                          out.numel()), "el_reduce_scatter_rows"), False, (out, full))
        return out

    def all_gather_rows_into(self, full, part, async_op=False):
        part = part if part.is_contiguous() else part.contiguous()
        w = self._run(lambda s: ops.check(self.ctx.lib.el_allgather_rows(self.ctx.handle, self._h, s, part.data_ptr(), full.data_ptr(),
                                                                         part.numel() * part.element_size()), "el_allgather_rows"), async_op,
                      (part, full))
        return w if async_op else full


def make_collectives(kind, ctx=None, rank=0, world=1):
    """kind "torch": torch.distributed (RCCL through the process group); "abi": RCCL through the library's own C ABI."""
    if kind == "abi":
        return RcclAbiCollectives(ctx, rank, world)
    return _Collectives()


# ------------------------------------------------------------------------------------------------------
# top-k
# ------------------------------------------------------------------------------------------------------
def sharded_topk(ctx, coll, Gu, Gi_shard, Bi_shard, item_lo, u_start, u_stop, k, excl=None, algo="auto", items_unchanged=False):
    """Full-catalogue top-k of users [u_start, u_stop) with the item table sharded over `coll.world` ranks."""
    pi, pv = ops.score_topk(ctx, Gu, Gi_shard, Bi_shard, u_start, u_stop, k, excl=excl, item_offset=item_lo, algo=algo,
                            items_unchanged=items_unchanged)
    if coll.world == 1 and not coll.always:
        return pi, pv
    n = u_stop - u_start
    if hasattr(coll, "all_gather_topk"):                          # C ABI: both lists in one RCCL group (el_allgather_topk)
        gi, gv = coll.all_gather_topk(pi, pv)
    else:
        gi = coll.all_gather(pi).reshape(coll.world, n, k)
        gv = coll.all_gather(pv).reshape(coll.world, n, k)
    return ops.topk_merge(ctx, gi, gv)


def gather_item_table(coll, Gi_shard, Bi_shard, n_items):
    """Every rank's item shard -> the whole table on every rank (one RCCL all-gather per evaluation: I F 4 bytes, 51 MB at
    C2).  With the table whole, full-catalogue top-k shards by USER: independent units, no collective on the data path."""
    if coll.world == 1 and not coll.always:
        return Gi_shard, Bi_shard
    rows = (n_items + coll.world - 1) // coll.world                      # item_range() shards differ by at most one row
    F = Gi_shard.shape[1]
    pad = torch.zeros((rows, F + 1), dtype=torch.float32, device=Gi_shard.device)
    pad[:Gi_shard.shape[0], :F] = Gi_shard
    pad[:Bi_shard.shape[0], F] = Bi_shard
    allp = coll.all_gather(pad).reshape(coll.world, rows, F + 1)
    parts = []
    for r in range(coll.world):
        lo, hi = item_range(n_items, r, coll.world)
        parts.append(allp[r, :hi - lo])
    full = torch.cat(parts)
    return full[:, :F].contiguous(), full[:, F].contiguous()


# ------------------------------------------------------------------------------------------------------
# training
# ------------------------------------------------------------------------------------------------------
class HipBackend:
    """Product backend: device state + libelliot_hip.so kernels."""

    def __init__(self, ctx, Gu, Gi_shard, Bi_shard, optimizer="adam_tf_dense"):
        if optimizer not in ("adam", "adam_tf_dense", "sgd"):
            raise ValueError("item-sharded training supports the dense optimisers (adam_tf_dense, sgd)")
        self.ctx = ctx
        self.state = ops.BprmfDeviceState(ctx, Gu, Gi_shard, Bi_shard, optimizer="sgd_dense" if optimizer == "sgd" else optimizer, deferred=False,
                                          compact_user_grads=False)     # el_rows_segment_sum fills the dense accumulator
        self._ws = None
        self._ws2 = None
        self._dU = None

    def shard_grads(self, u, i, j, l_w, l_b):
        import ctypes as C
        st, ctx = self.state, self.ctx
        B = u.numel()
        need = int(ctx.lib.el_bprmf_ws_bytes(int(B), int(st.U), int(st.I), int(st.F)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=ctx.device)
        if self._dU is None or self._dU.shape[0] != B:
            self._dU = torch.empty((B, st.F), dtype=torch.float32, device=ctx.device)
        ops.check(ctx.lib.el_bprmf_shard_grads(ctx.handle, ctx.stream(), C.byref(st._c), ops._ptr(u, torch.int32),
                                               ops._ptr(i, torch.int32), ops._ptr(j, torch.int32), int(B), float(l_w),
                                               float(l_b), int(st.step + 1), ops._ptr(self._dU, torch.float32),
                                               ops._ptr(st.loss, torch.float64), C.c_void_p(self._ws.data_ptr()),
                                               self._ws.numel()), "el_bprmf_shard_grads")
        return self._dU

    def reduce_user_rows(self, ids, rows):
        import ctypes as C
        st, ctx = self.state, self.ctx
        n = ids.numel()
        need = int(ctx.lib.el_rows_segment_sum_ws_bytes(int(n), int(st.U)))
        if self._ws2 is None or self._ws2.numel() < need:
            self._ws2 = torch.empty(need, dtype=torch.uint8, device=ctx.device)
        ops.check(ctx.lib.el_rows_segment_sum(ctx.handle, ctx.stream(), ops._ptr(ids, torch.int32),
                                              ops._ptr(rows, torch.float32), int(n), int(st.F), int(st.U),
                                              ops._ptr(st.gGu, torch.float32), C.c_void_p(self._ws2.data_ptr()),
                                              self._ws2.numel()), "el_rows_segment_sum")

    def apply(self, lr):
        import ctypes as C
        st, ctx = self.state, self.ctx
        st.step += 1
        ops.check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(st._c), float(lr), int(st.opt), int(st.step),
                                         float(ops.adam_lr_t(lr, st.step))), "el_bprmf_apply")

    def local_loss_tensor(self):
        return self.state.loss


def user_shard_rows(n_users, world):
    """Rows per rank of the user table in "dense" mode (the table is padded to world * rows)."""
    return (n_users + world - 1) // world


class HipDenseBackend:
    """Product backend of the "dense" mode: full (padded) user table replica for the forward pass, dense gradient
    accumulator gGu [world * Us, F], Adam state only for the rank's own user rows [rank * Us, (rank + 1) * Us)."""

    def __init__(self, ctx, Gu, Gi_shard, Bi_shard, rank, world, optimizer="adam_tf_dense"):
        import ctypes as C
        from ._lib import BprmfState
        if optimizer not in ("adam", "adam_tf_dense", "sgd"):
            raise ValueError("item-sharded training supports the dense optimisers (adam_tf_dense, sgd)")
        self.ctx, self.rank, self.world = ctx, rank, world
        dev = ctx.device
        U, F = Gu.shape
        self.U, self.F = int(U), int(F)
        self.Us = user_shard_rows(self.U, world)
        if self.Us * world == self.U and isinstance(Gu, torch.Tensor):
            Gu_pad = Gu                                             # nothing to pad (a 51 GB table at configs[4]: no second copy)
        else:
            Gu_pad = torch.zeros((self.Us * world, F), dtype=torch.float32, device=dev)
            Gu_pad[:U].copy_(Gu if isinstance(Gu, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(Gu)))
        # state of the gradient pass: whole (padded) user table, local item shard; no optimiser slots needed for Gu
        self.state = ops.BprmfDeviceState(ctx, Gu_pad, Gi_shard, Bi_shard, optimizer="sgd_dense", deferred=False)
        del Gu_pad
        st = self.state
        self.opt = ops.OPTIMIZERS["sgd_dense" if optimizer == "sgd" else optimizer]
        adam = self.opt == ops.EL_OPT_ADAM_TF_DENSE
        lo = rank * self.Us
        self.Gu_own = st.Gu[lo:lo + self.Us]                        # view: the rows this rank updates
        self.g_own = torch.zeros((self.Us, F), dtype=torch.float32, device=dev)     # reduce-scatter output
        z = torch.zeros_like
        self.mGu = z(self.g_own) if adam else None
        self.vGu = z(self.g_own) if adam else None
        self.mGi, self.vGi = (z(st.Gi), z(st.Gi)) if adam else (None, None)
        self.mBi, self.vBi = (z(st.Bi), z(st.Bi)) if adam else (None, None)
        p = lambda t: t.data_ptr() if t is not None else None
        self._apply_c = BprmfState(Gu=p(self.Gu_own), Gi=p(st.Gi), Bi=p(st.Bi), gGu=p(self.g_own), gGi=p(st.gGi), gBi=p(st.gBi),
                                   mGu=p(self.mGu), vGu=p(self.vGu), mGi=p(self.mGi), vGi=p(self.vGi), mBi=p(self.mBi),
                                   vBi=p(self.vBi), tGu=None, tGi=None, tBi=None, U=self.Us, I=st.I, F=self.F)
        self._ws = None
        self._C = C

    def grads(self, u, i, j, l_w, l_b):
        C = self._C
        st, ctx = self.state, self.ctx
        B = u.numel()
        need = int(ctx.lib.el_bprmf_ws_bytes(int(B), int(st.U), int(st.I), int(st.F)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=ctx.device)
        ops.check(ctx.lib.el_bprmf_grads(ctx.handle, ctx.stream(), C.byref(st._c), ops._ptr(u, torch.int32),
                                         ops._ptr(i, torch.int32), ops._ptr(j, torch.int32), int(B), float(l_w), float(l_b),
                                         int(st.step + 1), ops._ptr(st.loss, torch.float64), C.c_void_p(self._ws.data_ptr()),
                                         self._ws.numel()), "el_bprmf_grads")
        return st.gGu

    def apply_own(self, lr):
        C = self._C
        st, ctx = self.state, self.ctx
        st.step += 1
        ops.check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(self._apply_c), float(lr), int(self.opt),
                                         int(st.step), float(ops.adam_lr_t(lr, st.step))), "el_bprmf_apply")

    def local_loss_tensor(self):
        return self.state.loss


class ShardedBprmfDense:
    """BPRMF_batch train step over item shards, "dense" exchange: the all-reduce of user-row gradients split around a
    sharded optimiser.  Per step and rank: local gradients into the dense accumulator (el_bprmf_grads) -> RCCL
    REDUCE-SCATTER of gGu (each rank receives the summed gradient of its U/G user rows) -> optimiser on the owned user
    rows and the local item shard (el_bprmf_apply) -> RCCL ALL-GATHER of the updated user rows into every replica.
    Traffic per rank 2 (G-1)/G U F 4 bytes, independent of the batch -- versus (G-1) B F 4 bytes for the row exchange of
    ShardedBprmf -- and the dense TF-Adam pass over the user table shrinks by G.  Same mathematics as one rank with the
    concatenated batch (the reduction order of the fp32 sums is RCCL's)."""

    def __init__(self, backend, coll=None):
        self.backend = backend
        self.coll = coll or _Collectives()
        self._pending = None

    def finish(self):
        """The all-gather of a step is left in flight so that the caller's next sampling overlaps it; call this before
        anything else reads the user table (train_step and pop_loss do it themselves)."""
        if self._pending is not None:
            self._pending.wait()
            self._pending = None

    def train_step(self, u, i_local, j_local, lr, l_w, l_b):
        be, coll = self.backend, self.coll
        self.finish()
        gfull = be.grads(u, i_local, j_local, l_w, l_b)
        coll.reduce_scatter_rows(be.g_own, gfull)
        gfull.zero_()                                               # accumulators are zero on entry of every step
        be.apply_own(lr)
        self._pending = coll.all_gather_rows_into(be.state.Gu, be.Gu_own, async_op=True)

    def pop_loss(self):
        self.finish()
        t = self.backend.local_loss_tensor()
        tot = self.coll.all_reduce_sum(t.clone())
        t.zero_()
        return float(tot.item())


def user_range(n_users, rank, world):
    return (n_users * rank) // world, (n_users * (rank + 1)) // world


class HipUserShardBackend:
    """Product backend of the user-sharded step: the rank's user rows (+ their optimiser state) and a full replica of
    the item table (+ identical optimiser state on every rank)."""

    def __init__(self, ctx, Gu_shard, Gi, Bi, optimizer="adam_tf_dense"):
        if optimizer not in ("adam", "adam_tf_dense", "sgd"):
            raise ValueError("sharded training supports the dense optimisers (adam_tf_dense, sgd)")
        self.ctx = ctx
        # (the step runs as grads + all-reduce + apply here: the every-row two-pass form -- no second user table, no deferred
        # decay, no fused item side: the item gradients have to exist as a table for the all-reduce anyway)
        self.state = ops.BprmfDeviceState(ctx, Gu_shard, Gi, Bi, optimizer="sgd_dense" if optimizer == "sgd" else optimizer, deferred=False,
                                          fused_user_step=False)
        self._ws = None

    def _workspace(self, B):
        st, ctx = self.state, self.ctx
        need = int(ctx.lib.el_bprmf_ws_bytes(int(B), int(st.U), int(st.I), int(st.F)))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=ctx.device)

    def presort(self, u_local, i, j):
        """Order a batch for grads(..., presorted=True): reads only the triplets, so it can run under the previous step's
        collective.  One workspace: the previous batch's segment kernels are done (stream order) before this overwrites it."""
        import ctypes as C
        st, ctx = self.state, self.ctx
        self._workspace(u_local.numel())
        ops.check(ctx.lib.el_bprmf_presort(ctx.handle, ctx.stream(), ops._ptr(u_local, torch.int32), ops._ptr(i, torch.int32),
                                           ops._ptr(j, torch.int32), int(u_local.numel()), int(st.U), int(st.I),
                                           C.c_void_p(self._ws.data_ptr()), self._ws.numel()), "el_bprmf_presort")

    def grads(self, u_local, i, j, l_w, l_b, presorted=False):
        import ctypes as C
        st, ctx = self.state, self.ctx
        B = u_local.numel()
        self._workspace(B)
        st.ensure_rows(B)                                       # (compact user-gradient rows: one slot per sorted position)
        fn = ctx.lib.el_bprmf_grads_presorted if presorted else ctx.lib.el_bprmf_grads
        ops.check(fn(ctx.handle, ctx.stream(), C.byref(st._c), ops._ptr(u_local, torch.int32), ops._ptr(i, torch.int32),
                     ops._ptr(j, torch.int32), int(B), float(l_w), float(l_b), int(st.step + 1), ops._ptr(st.loss, torch.float64),
                     C.c_void_p(self._ws.data_ptr()), self._ws.numel()), "el_bprmf_grads")

    def item_grads(self):
        return [self.state.item_grad_flat]                    # gGi rows + gBi in one buffer: one collective per step

    # -- row-sparse exchange of the item gradients (ShardedBprmfByUser, item_exchange="rows") --------------------------------
    def touched_item_rows(self, i, j):
        """The distinct items of this rank's batch (ascending) and their rows of the gradient accumulators the segment kernels just
        filled: (ids int32 [n], dGi rows [n, F], dBi [n]).  Every non-zero row of gGi / gBi is in the list."""
        st = self.state
        ids = torch.unique(torch.cat([i, j]).to(torch.int64))
        return ids.to(torch.int32), st.gGi.index_select(0, ids), st.gBi.index_select(0, ids)

    def set_item_grads(self, ids_all, rows_all, bias_all):
        """gGi / gBi <- the sum, per item, of the gathered (id, row) records of every rank -- the rows in the gathered order, which is
        the same on every rank (el_rows_segment_sum: stable sort by id, one lane group walks a segment): identical replicas.  Every row
        a rank's own batch touched is among the records, so every stale local row is overwritten with the global sum."""
        import ctypes as C
        st, ctx = self.state, self.ctx
        n = int(ids_all.numel())
        need = int(ctx.lib.el_rows_segment_sum_ws_bytes(n, int(st.I)))
        if getattr(self, "_ws_rows", None) is None or self._ws_rows.numel() < need:
            self._ws_rows = torch.empty(need, dtype=torch.uint8, device=ctx.device)
        for rows, F, out in ((rows_all, st.F, st.gGi), (bias_all, 1, st.gBi)):
            ops.check(ctx.lib.el_rows_segment_sum(ctx.handle, ctx.stream(), ops._ptr(ids_all, torch.int32), ops._ptr(rows, torch.float32), n,
                                                  int(F), int(st.I), ops._ptr(out, torch.float32), C.c_void_p(self._ws_rows.data_ptr()),
                                                  self._ws_rows.numel()), "el_rows_segment_sum")

    def _apply(self, c_state, lr):
        import ctypes as C
        st, ctx = self.state, self.ctx
        ops.check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(c_state), float(lr), int(st.opt), int(st.step),
                                         float(ops.adam_lr_t(lr, st.step))), "el_bprmf_apply")

    def apply(self, lr):
        self.state.step += 1
        self._apply(self.state._c, lr)

    def begin_step(self):
        self.state.step += 1

    def apply_users(self, lr):
        """Optimiser on the rank's own user rows only (needs no remote data: overlaps the item-gradient all-reduce)."""
        if not hasattr(self, "_c_users"):
            clone = lambda c: type(c).from_buffer_copy(c)
            self._c_users, self._c_items = clone(self.state._c), clone(self.state._c)
            self._c_users.I = 0                                 # zero-length item passes
            self._c_items.U = 0
            self.state._c_clones = (self._c_users, self._c_items)
        self._apply(self._c_users, lr)

    def apply_items(self, lr):
        self._apply(self._c_items, lr)

    def local_loss_tensor(self):
        return self.state.loss


class ShardedBprmfByUser:
    """BPRMF_batch train step with the USER table sharded and the item table replicated -- the cheap way round when
    U >> I (every BASELINE configuration: U / I = 10).  Rank r owns the user rows [ulo_r, uhi_r) with their optimiser
    state and draws its B triplets for ITS users (u uniform in the shard, i in pos(u), j anywhere in the catalogue: over
    the ranks that is exactly the reference's sampling distribution, custom_sampler.py:31-42 -- no shard-restricted
    negatives).  User-row gradients never leave the rank; the item-side gradients gGi [I,F], gBi [I] are ALL-REDUCED
    (RCCL, I F 4 bytes = 51 MB at C2, versus 2 (G-1)/G U F 4 = 0.9 GB for the user-gradient exchange of the item-sharded
    forms) and every rank applies the same Adam step to its item replica.  G ranks x B triplets are one
    reference-semantics step on the concatenated batch.  Full-catalogue top-k then needs no collective at all: each rank
    scores the users it owns."""

    def __init__(self, backend, coll=None, item_exchange="dense"):
        """item_exchange: how the item-side gradients meet.  "dense" -- one all-reduce of gGi [I, F] + gBi [I] (I (F + 1) 4 bytes whatever
        the batch).  "rows" -- every rank contributes only the rows its batch touched: all-gather of (item id, dGi row, dBi) records,
        padded to the longest list, then a segment sum per item in the gathered order on every rank (the same order everywhere:
        identical replicas, and a deterministic sum -- the all-reduce's order is RCCL's).  G n (F + 2) 4 bytes for lists of n rows:
        it wins while G^2 n < 2 (G - 1) I, i.e. for batches that touch a small part of the catalogue (pick_item_exchange)."""
        if item_exchange not in ("dense", "rows"):
            raise ValueError("item_exchange must be 'dense' or 'rows'")
        self.backend = backend
        self.coll = coll or _Collectives()
        self.item_exchange = item_exchange
        self.last_rows = None                                   # rows mode: (local list length, padded length) of the last step

    def _exchange_rows(self, i, j):
        """All-gather of the ranks' touched item rows.  One host synchronisation per step: the list lengths."""
        be, coll = self.backend, self.coll
        ids, rows, bias = be.touched_item_rows(i, j)
        n = int(ids.shape[0])
        if coll.world == 1 and not coll.always:
            self.last_rows = (n, n)
            return ids, rows, bias
        cnt = coll.all_gather(torch.tensor([n], dtype=torch.int64, device=ids.device))
        n_max = int(cnt.max().item())
        self.last_rows = (n, n_max)
        if n < n_max:                                            # padding records: a touched id of this rank with a zero row (adds nothing)
            pad = n_max - n
            fill = ids[:1] if n else torch.zeros(1, dtype=ids.dtype, device=ids.device)
            ids = torch.cat([ids, fill.expand(pad)])
            rows = torch.cat([rows, torch.zeros((pad, rows.shape[1]), dtype=rows.dtype, device=rows.device)])
            bias = torch.cat([bias, torch.zeros(pad, dtype=bias.dtype, device=bias.device)])
        return coll.all_gather(ids.contiguous()), coll.all_gather(rows.contiguous()), coll.all_gather(bias.contiguous())

    def train_step(self, u_local, i, j, lr, l_w, l_b, overlap=None, presorted=False):
        """overlap: optional callable enqueuing model-independent work (drawing AND ordering the NEXT batch: backend.presort)
        while the collective is in flight -- it is called after the all-reduce was issued and before its result is waited for.
        presorted: this batch was ordered by backend.presort already."""
        be, coll = self.backend, self.coll
        if presorted:
            be.grads(u_local, i, j, l_w, l_b, presorted=True)
        else:
            be.grads(u_local, i, j, l_w, l_b)
        if self.item_exchange == "rows":
            gathered = self._exchange_rows(i, j)
            be.begin_step()
            be.apply_users(lr)
            if overlap is not None:
                overlap()
            be.set_item_grads(*gathered)
            be.apply_items(lr)
            return
        if not hasattr(be, "apply_users"):                      # test backends: plain order
            for g in be.item_grads():
                coll.all_reduce_sum(g)
            be.apply(lr)
            if overlap is not None:
                overlap()
            return
        works = [coll.all_reduce_sum(g, async_op=True) for g in be.item_grads()]
        be.begin_step()
        be.apply_users(lr)                                      # the rank's own rows: runs under the all-reduce
        if overlap is not None:
            overlap()                                           # ... and so does whatever does not read the model
        for w in works:
            if w is not None:
                w.wait()
        be.apply_items(lr)

    def pop_loss(self):
        t = self.backend.local_loss_tensor()
        tot = self.coll.all_reduce_sum(t.clone())
        t.zero_()
        return float(tot.item())


def expected_touched_items(n_items, batch_per_rank):
    """Expected distinct items of one rank's batch when both of a triplet's items are drawn uniformly (an upper bound: the
    positives of a real catalogue are popularity-skewed and collide more)."""
    import math
    return n_items * (1.0 - math.exp(-2.0 * batch_per_rank / max(1, n_items)))


def item_exchange_bytes(n_items, F, batch_per_rank, world, rows=None):
    """Bytes a rank puts on the wire per step for the item-gradient exchange of the user-sharded step: the dense all-reduce
    (2 (G - 1) / G of the table) and the all-gather of touched rows ((G - 1) lists of `rows` records of F + 2 words arrive)."""
    n = expected_touched_items(n_items, batch_per_rank) if rows is None else rows
    return {"dense": 2.0 * (world - 1) / world * n_items * (F + 1) * 4, "rows": (world - 1) * n * (F + 2) * 4.0, "touched": n}


def pick_item_exchange(n_items, F, batch_per_rank, world):
    """ "rows" while the gathered lists are smaller than what the all-reduce moves; "dense" otherwise (every configuration of
    BASELINE.json at B = 2^20 per rank: a batch touches 24 % (C5) to 88 % (C4) of the catalogue, eight of them all of it)."""
    if world <= 1:
        return "dense"
    b = item_exchange_bytes(n_items, F, batch_per_rank, world)
    return "rows" if b["rows"] < 0.8 * b["dense"] else "dense"


def pick_exchange(n_users, batch_per_rank, world):
    """Bytes moved per rank and step decide: rows = (G-1) B F 4 (all-gather of per-triplet rows), dense = 2 (G-1)/G U F 4."""
    if world <= 1:
        return "rows"
    return "dense" if 2.0 * n_users / world < batch_per_rank else "rows"


class ShardedBprmf:
    """BPRMF_batch train step over item shards (see module docstring)."""

    def __init__(self, backend, coll=None):
        self.backend = backend
        self.coll = coll or _Collectives()

    def train_step(self, u, i_local, j_local, lr, l_w, l_b):
        be, coll = self.backend, self.coll
        dU = be.shard_grads(u, i_local, j_local, l_w, l_b)          # local: loss, dGi/dBi, per-triplet dGu rows
        ids = coll.all_gather(u)                                    # RCCL all-gather over xGMI
        rows = coll.all_gather(dU)
        be.reduce_user_rows(ids, rows)                              # same order on every rank -> identical replicas
        be.apply(lr)

    def pop_loss(self):
        """Global batch loss (sum over ranks), like the single-GPU accumulator."""
        t = self.backend.local_loss_tensor()
        tot = self.coll.all_reduce_sum(t.clone())
        t.zero_()
        return float(tot.item())


# ------------------------------------------------------------------------------------------------------
# NeuMF / GMF: data parallel over samples, item tables sharded, RCCL all-reduce of the shared gradients
# ------------------------------------------------------------------------------------------------------
class ShardedNmf:
    """One NeuMF / GMF step over G ranks, data parallel over samples with one kind of embedding table sharded:

      shard="user" (default)  rank r owns the rows [ulo_r, uhi_r) of both USER tables (+ Adam state) and draws its samples
                              for its own users (any item: the reference's sampling distribution); item tables, Dense
                              layers and head are replicated.  All-reduced per step: 2 I F 4 bytes + the MLP.
      shard="item"            north_star's formulation: item rows sharded, samples restricted to the shard's items, user
                              tables replicated (2 U F 4 bytes all-reduced per step -- 10x more when U = 10 I).

    Per step: el_nmf_grads with the BinaryCrossentropy mean over the GLOBAL batch (sum of all ranks' n) -> RCCL all-reduce
    (sum) of the gradients of the replicated variables -> el_nmf_apply (Keras Adam) on every rank: G ranks x n samples are
    exactly one reference-semantics step on the concatenated batch, and the replicas stay identical because every rank
    applies the same reduced gradients.  `backend` = ops.NmfDeviceState built from weights whose sharded tables hold only
    the local rows (ids of that kind passed in are shard-local), or a stand-in with the same four methods."""

    def __init__(self, backend, coll=None, shard="user"):
        if shard not in ("user", "item"):
            raise ValueError("shard must be 'user' or 'item'")
        self.backend = backend
        self.coll = coll or _Collectives()
        self.shard = shard
        if hasattr(backend, "set_deferred"):
            # the replicated tables take gradients of OTHER ranks' samples through the all-reduce: which rows moved is not known
            # from the local batch, so the per-row deferred decay of ops.NmfDeviceState is off and every step streams every row
            backend.set_deferred(False)

    def train_step(self, u, i, label, lr, n_global=None):
        be, coll = self.backend, self.coll
        if n_global is None:
            n_global = coll.world * int(u.shape[0])
        be.grads(u, i, label, n_global)
        for g in be.replicated_grads(self.shard):
            coll.all_reduce_sum(g)
        be.apply(lr)

    def pop_loss(self):
        t = self.backend.loss
        tot = self.coll.all_reduce_sum(t.clone())
        t.zero_()
        return float(tot.item())


# ------------------------------------------------------------------------------------------------------
# point-wise factor models (MF, PMF, FunkSVD, LogisticMF): user rows sharded, item side replicated
# ------------------------------------------------------------------------------------------------------
class ShardedPwmf:
    """One step of a point-wise factor model over G ranks, the same scheme as ShardedBprmfByUser: rank r owns the user rows
    [ulo_r, uhi_r) of Gu (and Bu) with their optimiser slots and draws its (u, i, label) samples for ITS users (any item: the
    reference's sampling distribution over the ranks, pointwise_pos_neg_sampler.py:32-46); Gi / Bi are replicated.
    Per step: el_pwmf_grads with the batch MEAN over the GLOBAL batch -> ONE asynchronous all-reduce (sum) per item-side
    accumulator, under which the rank's own user rows take their optimiser step -> the same item-side step on every rank.
    G ranks x n samples are one reference-semantics step on the concatenated batch.  LogisticMF's alternating sides: a
    `side="users"` step needs no collective at all.  `backend` = ops.PwmfDeviceState on (local user rows, full item tables;
    user ids passed in are shard-local) or a stand-in with grads / apply / item_grads / loss."""

    def __init__(self, backend, coll=None):
        self.backend = backend
        self.coll = coll or _Collectives()

    def train_step(self, u_local, i, label, lr, n_global=None, side="both"):
        be, coll = self.backend, self.coll
        if n_global is None:
            n_global = coll.world * int(u_local.shape[0])
        be.grads(u_local, i, label, n_global=n_global, side=side)
        works = [coll.all_reduce_sum(g, async_op=True) for g in be.item_grads()] if side != "users" else []
        if side != "items":
            be.apply(lr, side="users", advance=True)             # own rows: runs under the all-reduce
        for w in works:
            if w is not None:
                w.wait()
        if side != "users":
            be.apply(lr, side="items", advance=(side == "items"))

    def pop_loss(self):
        t = self.backend.loss
        tot = self.coll.all_reduce_sum(t.clone())
        t.zero_()
        return float(tot.item())


# ------------------------------------------------------------------------------------------------------
# CML: user rows sharded, item side replicated -- and a real exchange step
# ------------------------------------------------------------------------------------------------------
class ShardedCml:
    """One CML step over G ranks.  The reference's [B,B] hinge (CML_model.py:60-66,78) couples every triplet's distance
    difference D_a with every triplet's bias difference E_b of the batch, so a data-parallel step has something to exchange:
    el_cml_forward -> ALL-GATHER of the ranks' D and E vectors (2 x B floats per rank over RCCL) -> el_cml_grads of the rank's
    triplets against the gathered vectors -> all-reduce of the item-side gradients, the rank's own user rows stepping under it
    -> the item-side step.  G ranks x B triplets are one reference-semantics step on the concatenated batch of G B triplets
    (its hinge has (G B)^2 terms).  `backend` = ops.CmlDeviceState on (local user rows, full item tables) or a stand-in."""

    def __init__(self, backend, coll=None):
        self.backend = backend
        self.coll = coll or _Collectives()

    def train_step(self, u_local, i, j, lr, l_w, l_b, margin):
        be, coll = self.backend, self.coll
        D, E = be.forward_de(u_local, i, j, l_w, l_b)
        D_all, E_all = coll.all_gather(D), coll.all_gather(E)
        be.grads_de(u_local, i, j, l_w, l_b, margin, D, E, D_all, E_all)
        works = [coll.all_reduce_sum(g, async_op=True) for g in be.item_grads()]
        be.begin_step()
        be.apply_users(lr)
        for w in works:
            if w is not None:
                w.wait()
        be.apply_items(lr)

    def pop_loss(self):
        t = self.backend.loss
        tot = self.coll.all_reduce_sum(t.clone())
        t.zero_()
        return float(tot.item())


# ------------------------------------------------------------------------------------------------------
# Mult-VAE / Mult-DAE: data parallel over the user rows of a batch, weights replicated
# ------------------------------------------------------------------------------------------------------
class ShardedVae:
    """One Mult-VAE / Mult-DAE step over G ranks: the users of a batch are independent samples, so the batch is
    split; every rank runs el_vae_grads on its rows with both batch means taken over the GLOBAL batch, the gradients of the
    ten (replicated) dense variables are all-reduced (sum) -- 4 (2 I H + ...) bytes, 130 MB at the ML-20M shape: the bandwidth-
    heavy form the survey names; it is here for completeness, the named configuration is single-GPU -- and every rank applies
    the same Adam step.  `backend` = ops.VaeDeviceState or a stand-in with grads / apply / dense_grads / loss."""

    def __init__(self, backend, coll=None):
        self.backend = backend
        self.coll = coll or _Collectives()

    def train_step(self, train_csr, rows, lr, anneal, eps=None, dropout_rate=0.0, dropout_seed=42, n_global=None):
        be, coll = self.backend, self.coll
        if n_global is None:
            n_global = coll.world * int(rows.shape[0])
        be.grads(train_csr, rows, anneal, eps=eps, dropout_rate=dropout_rate, dropout_seed=dropout_seed, n_global=n_global)
        for g in be.dense_grads():
            coll.all_reduce_sum(g)
        be.apply(lr)

    def pop_loss(self):
        t = self.backend.loss
        tot = self.coll.all_reduce_sum(t.clone())
        t.zero_()
        return float(tot.item())
