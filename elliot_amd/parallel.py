"""Item-sharded multi-GPU execution -- new design, the reference is single-device.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI on the MI355X node, "gloo" in the CPU
tests).  Rank r owns the item range [lo_r, hi_r): its rows of Gi / Bi and their optimiser state; the user table is
replicated.

  top-k     every rank scores each user block against its item slice (global ids through `item_offset`), the
            [Ub, k] partial lists are ALL-GATHERED (8*Ub*k bytes per rank) and merged with the single-GPU ordering
            rule -> identical lists to one GPU.
  training  every rank draws its own triplets with positive AND negative inside its shard (shard-local negative
            sampling, north_star), computes loss / item-row gradients locally and one user-gradient row per triplet;
            the (user id, row) pairs are ALL-GATHERED, every rank reduces them by user in the same order and applies
            the optimiser to its replica of the user table and to its item shard.  G ranks x B triplets are
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

    def all_gather(self, t):
        if self.world == 1:
            return t
        out = torch.empty((self.world * t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        self.dist.all_gather_into_tensor(out, t.contiguous())
        return out

    def all_reduce_sum(self, t):
        if self.world > 1:
            self.dist.all_reduce(t)
        return t


# ------------------------------------------------------------------------------------------------------
# top-k
# ------------------------------------------------------------------------------------------------------
def sharded_topk(ctx, coll, Gu, Gi_shard, Bi_shard, item_lo, u_start, u_stop, k, excl=None, algo="auto"):
    """Full-catalogue top-k of users [u_start, u_stop) with the item table sharded over `coll.world` ranks."""
    pi, pv = ops.score_topk(ctx, Gu, Gi_shard, Bi_shard, u_start, u_stop, k, excl=excl, item_offset=item_lo, algo=algo)
    if coll.world == 1:
        return pi, pv
    n = u_stop - u_start
    gi = coll.all_gather(pi).reshape(coll.world, n, k)
    gv = coll.all_gather(pv).reshape(coll.world, n, k)
    return ops.topk_merge(ctx, gi, gv)


# ------------------------------------------------------------------------------------------------------
# training
# ------------------------------------------------------------------------------------------------------
class HipBackend:
    """Product backend: device state + libelliot_hip.so kernels."""

    def __init__(self, ctx, Gu, Gi_shard, Bi_shard, optimizer="adam_tf_dense"):
        if optimizer not in ("adam", "adam_tf_dense", "sgd"):
            raise ValueError("item-sharded training supports the dense optimisers (adam_tf_dense, sgd)")
        self.ctx = ctx
        self.state = ops.BprmfDeviceState(ctx, Gu, Gi_shard, Bi_shard, optimizer="sgd_dense" if optimizer == "sgd" else optimizer)
        self._ws = None
        self._ws2 = None
        self._dU = None

    def shard_grads(self, u, i, j, l_w, l_b):
        import ctypes as C
        st, ctx = self.state, self.ctx
        B = u.numel()
        need = int(ctx.lib.el_bprmf_ws_bytes(int(B), int(st.U), int(st.I)))
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
