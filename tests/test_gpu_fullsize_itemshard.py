"""north_star's own partitioning at full size: ONE rank's item shard of BASELINE configs[4] (BPRMF d = 256, 50 M users x 5 M items
over 8 GPUs: 50 M user rows REPLICATED x 625 K item rows; U F = 1.28e10 > 2^32, a 51 GB user table, a 51 GB dense user-gradient
table for the reduce-scatter) and of the metric's target shape (10 M x 1 M x 128: 10 M x 125 K per rank), through the product's own
sharded trainers -- parallel.ShardedBprmfDense (reduce-scatter of the dense gradient table / optimiser on the owned rows /
all-gather of the rows), parallel.ShardedBprmf (all-gather of per-triplet rows) and parallel.sharded_topk (item-offset lists +
el_topk_merge) -- with RCCL called through the C ABI.

The dev box has ONE GPU, so rank 3 of 8 is run alone: `_VirtualRank` presents the world-8 shapes to the trainer and forwards each
collective to a ONE-rank RCCL communicator on the rank's own slice (the other seven ranks' contributions are zero gradients /
untouched rows).  What that covers: every kernel at the real per-rank shape (64-bit row offsets, the 51 GB accumulators, shard-local
item ids + item_offset), the RCCL entry points, the trainer's call sequence.  What it cannot: bytes on a wire between two GPUs.

Properties (the oracle cannot run at this size; these can):
  training  batch loss == independent fp64 evaluation of BPRMF_batch_model.py:65-75 on the same triplets (1e-4); PRE-optimiser
            gradients of sampled user rows (owned and not owned) and item rows (random + hottest) == oracle/bprmf_batch.py on
            exactly the triplets that touch them; after two steps the OWNED user rows and the shard's item rows == the oracle's
            Keras Adam, rows of other ranks' user shards have not moved; accumulators zero on exit
  top-k     item-offset lists of the shard == the C oracle's fma chain on a sample of users (global ids); the shard cut in two
            + el_topk_merge == the unsplit list, through el_allgather_topk
If 50 M x 256 does not fit next to its gradient table on this box the c5 case falls back to the largest U that does and says so.
"""
import numpy as np
import pytest
import torch

from elliot_amd import ops, parallel
from elliot_amd.synthetic import zipf_csr_device
from oracle import bprmf_batch as ob
from oracle import cref
from tests.gpu_util import cpu

pytestmark = pytest.mark.gpu

WORLD, RANK, B, K = 8, 3, 1 << 20, 10
LR, L_W, L_B = 0.001, 0.1, 0.001
SHAPES = {"c4": dict(U=10_000_000, I=1_000_000, F=128, mean_log=3.0), "c5": dict(U=50_000_000, I=5_000_000, F=256, mean_log=1.6)}


class _VirtualRank:
    """World-`world` collectives seen from rank `rank` when every other rank contributes nothing: the rank's own slice goes
    through the inner (one-rank, RCCL) communicator."""

    def __init__(self, inner, rank, world):
        self.inner, self.rank, self.world, self.always = inner, rank, world, True

    def reduce_scatter_rows(self, out, full):
        n = out.shape[0]
        return self.inner.reduce_scatter_rows(out, full[self.rank * n:(self.rank + 1) * n])

    def all_gather_rows_into(self, full, part, async_op=False):
        n = part.shape[0]
        return self.inner.all_gather_rows_into(full[self.rank * n:(self.rank + 1) * n], part, async_op=async_op)

    def all_reduce_sum(self, t, async_op=False):
        return self.inner.all_reduce_sum(t, async_op=async_op)

    def all_gather(self, t):
        return self.inner.all_gather(t)                          # (the other ranks' rows: none)

    def all_gather_topk(self, idx, val):
        return self.inner.all_gather_topk(idx, val)


@pytest.fixture(scope="module", params=["c4", "c5"])
def shard(ctx, request):
    cfg = dict(SHAPES[request.param])
    dev = ctx.device
    U, I, F = cfg["U"], cfg["I"], cfg["F"]
    lo, hi = parallel.item_range(I, RANK, WORLD)
    torch.cuda.empty_cache()
    free = torch.cuda.mem_get_info(dev)[0]
    # the gradient pass keeps the user table + its dense gradient table (2 x U F 4 bytes), the caller's copy lives until the state has
    # cloned it (3 x at the peak) + owner-side slots; what does not fit is scaled down and the test says so
    note = None
    while 3.4 * U * F * 4 + (24 << 30) > free:
        U = U * 3 // 4 // WORLD * WORLD
        note = f"{request.param}: U reduced to {U} users (this box has {free / 2**30:.0f} GiB free)"
    if note:
        print(note)
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=cfg["mean_log"], sigma_log=1.0, dmin=5, dmax=2000, seed=808)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)            # global item ids: the exclusion mask of the top-k
    sip, six = parallel.shard_csr(indptr, indices, lo, hi)
    pos_shard = ops.DeviceCSR.from_tensors(sip, six, hi - lo)       # local ids: the rank's sampler (positive AND negative in the shard)
    g = torch.Generator(device=dev)
    g.manual_seed(9)
    Gu = torch.empty((U, F), device=dev)
    for s in range(0, U, 1 << 22):                                   # (in slices: torch.rand of 1.28e10 elements would need a second table)
        Gu[s:s + (1 << 22)] = (torch.rand((min(1 << 22, U - s), F), generator=g, device=dev) * 2 - 1) * 0.05
    Gi = (torch.rand((hi - lo, F), generator=g, device=dev) * 2 - 1) * 0.05
    Bi = (torch.rand(hi - lo, generator=g, device=dev) - 0.5) * 0.02
    inner = parallel.RcclAbiCollectives(ctx, 0, 1)
    coll = _VirtualRank(inner, RANK, WORLD)
    be = parallel.HipDenseBackend(ctx, Gu, Gi, Bi, RANK, WORLD, optimizer="adam_tf_dense")
    del Gu
    torch.cuda.empty_cache()
    out = {"name": request.param, "U": U, "I": I, "F": F, "lo": lo, "hi": hi, "pos": pos, "pos_shard": pos_shard, "be": be, "coll": coll,
           "g": g, "note": note}
    yield out
    inner.close()
    del be, out
    torch.cuda.empty_cache()


def _loss64(Gu, Gi, Bi, u, i, j, F):
    ref = 0.0
    step = 1 << (18 if F <= 128 else 17)
    for s in range(0, u.numel(), step):
        sl = slice(s, s + step)
        gu, gi, gj = Gu[u[sl].long()].double(), Gi[i[sl].long()].double(), Gi[j[sl].long()].double()
        bi, bj = Bi[i[sl].long()].double(), Bi[j[sl].long()].double()
        d = (bi + (gu * gi).sum(1)) - (bj + (gu * gj).sum(1))
        ref += float(torch.nn.functional.softplus(-d.clamp(-80.0, 1e8)).sum()
                     + L_W * 0.5 * ((gu * gu).sum() + (gi * gi).sum() + (gj * gj).sum())
                     + L_B * 0.5 * (bi * bi).sum() + (L_B / 10) * 0.5 * (bj * bj).sum())
    return ref


def _oracle_rows(U, Ish, Gu0, Gi0, Bi0, u, i, j, su, si):
    """oracle/bprmf_batch.py gradients (fp32 and fp64) of the sampled user rows / item rows from exactly the triplets that touch
    them, re-indexed into small tables (as tests/test_gpu_fullsize_c5.py does)."""
    dev = u.device
    mu = torch.zeros(U, dtype=torch.bool, device=dev)
    mu[su] = True
    mi = torch.zeros(Ish, dtype=torch.bool, device=dev)
    mi[si] = True
    sel = torch.nonzero(mu[u.long()] | mi[i.long()] | mi[j.long()]).flatten()
    uu, ii, jj = u[sel].long(), i[sel].long(), j[sel].long()
    users = torch.unique(torch.cat([uu, su]))
    items = torch.unique(torch.cat([ii, jj, si]))
    ru = torch.full((U,), -1, dtype=torch.int64, device=dev)
    ru[users] = torch.arange(users.numel(), device=dev)
    ri = torch.full((Ish,), -1, dtype=torch.int64, device=dev)
    ri[items] = torch.arange(items.numel(), device=dev)
    gu_s, gi_s, bi_s = cpu(Gu0[users]), cpu(Gi0[items]), cpu(Bi0[items])
    a = (cpu(ru[uu]), cpu(ri[ii]), cpu(ri[jj]))
    g32 = ob.gradients(gu_s, gi_s, bi_s, *a, L_W, L_B)
    g64 = ob.gradients(gu_s, gi_s, bi_s, *a, L_W, L_B, dtype=np.float64)
    pu, pi = cpu(ru[su]), cpu(ri[si])
    return {"n": int(sel.numel()), "gGu": (g32[1][pu], g64[1][pu]), "gGi": (g32[2][pi], g64[2][pi]), "gBi": (g32[0][pi], g64[0][pi])}


def test_dense_exchange_step_at_the_item_shard_shape(ctx, shard):
    be, coll, pos_shard, g = shard["be"], shard["coll"], shard["pos_shard"], shard["g"]
    U, F, Ish, dev = shard["U"], shard["F"], shard["hi"] - shard["lo"], ctx.device
    st = be.state
    assert st.Gu.shape == (U, F) and st.gGu.shape == (U, F) and be.Us * WORLD == U
    assert shard["name"] != "c5" or shard["note"] or U * F > (1 << 32)          # the 64-bit offsets are really exercised
    trainer = parallel.ShardedBprmfDense(be, coll)
    own_lo, own_hi = RANK * be.Us, (RANK + 1) * be.Us
    t0 = ops.bpr_sample(ctx, pos_shard, B, seed=42, first_sample=0, item_lo=0, item_hi=Ish)
    hot = torch.argsort(torch.bincount(torch.cat([t0[1], t0[2]]).long(), minlength=Ish), descending=True)[:3]
    su_own = own_lo + torch.randperm(be.Us, generator=g, device=dev)[:768]
    su_far = torch.randperm(own_lo, generator=g, device=dev)[:256]                # rows of ranks 0..2: gradients yes, updates no
    su = torch.cat([su_own, su_far])
    si = torch.unique(torch.cat([torch.randperm(Ish, generator=g, device=dev)[:1024], hot]))
    th = {"Gu": cpu(st.Gu[su_own]), "Gi": cpu(st.Gi[si]), "Bi": cpu(st.Bi[si])}
    far0 = cpu(st.Gu[su_far])
    m = {k: np.zeros_like(x) for k, x in th.items()}
    v = {k: np.zeros_like(x) for k, x in th.items()}
    for step in range(2):
        u, i, j = t0 if step == 0 else ops.bpr_sample(ctx, pos_shard, B, seed=42, first_sample=step * B, item_lo=0, item_hi=Ish)
        assert int(i.max()) < Ish and int(j.max()) < Ish and int(u.max()) < U
        trainer.finish()
        exp = _oracle_rows(U, Ish, st.Gu, st.Gi, st.Bi, u, i, j, su, si)
        assert exp["n"] > 3000
        ref = _loss64(st.Gu, st.Gi, st.Bi, u, i, j, F)
        if step == 0:
            # the trainer's own sequence, opened up after the gradient pass to look at the accumulators
            gfull = be.grads(u, i, j, L_W, L_B)
            got = {"gGu": cpu(gfull[su]), "gGi": cpu(st.gGi[si]), "gBi": cpu(st.gBi[si])}
            for name in ("gGu", "gGi", "gBi"):
                e32, e64 = exp[name]
                scale = float(np.abs(e64).max())
                err = float(np.abs(got[name] - e64).max())
                ref_err = float(np.abs(e32.astype(np.float64) - e64).max())
                assert err <= max(2e-5 * scale, 4 * ref_err), (name, err, ref_err, scale)
            coll.reduce_scatter_rows(be.g_own, gfull)
            assert torch.equal(be.g_own, gfull[own_lo:own_hi])
            gfull.zero_()
            be.apply_own(LR)
            trainer._pending = coll.all_gather_rows_into(st.Gu, be.Gu_own, async_op=True)
        else:
            trainer.train_step(u, i, j, LR, L_W, L_B)
        loss = trainer.pop_loss()
        assert abs(loss - ref) <= 1e-4 * abs(ref), (step, loss, ref)
        n_own = su_own.numel()
        for name, gname, rows in (("Gu", "gGu", slice(0, n_own)), ("Gi", "gGi", slice(None)), ("Bi", "gBi", slice(None))):
            ob.adam_tf_sparse_apply(th[name], m[name], v[name], exp[gname][0][rows].astype(np.float32), LR, step + 1)
    trainer.finish()
    assert not bool(st.gGi.any()) and not bool(st.gGu[own_lo:own_lo + (1 << 20)].any()) and not bool(st.gGu[:1 << 20].any())
    for name, rows in (("Gu", su_own), ("Gi", si), ("Bi", si)):
        gotw = cpu(getattr(st, name)[rows])
        err = np.abs(gotw - th[name])
        assert float((err > 2e-5).mean()) <= 2e-4 and float(err.max()) < 3 * LR, (name, float(err.max()), float((err > 2e-5).mean()))
    assert np.array_equal(cpu(st.Gu[su_far]), far0), "a user row owned by another rank moved"
    # Adam slots exist only for the owned rows
    assert be.mGu.shape == (be.Us, F) and float(be.mGu.abs().max()) > 0


def test_item_offset_topk_and_merge_at_the_item_shard_shape(ctx, shard):
    be, coll, pos = shard["be"], shard["coll"], shard["pos"]
    lo, hi, U, F, dev = shard["lo"], shard["hi"], shard["U"], shard["F"], ctx.device
    st = be.state
    Ub = 16384
    s0 = 7 * Ub
    # the trainer's path: partial lists of the shard (global ids), all-gathered (one rank here) and merged
    pi, pv = parallel.sharded_topk(ctx, coll, st.Gu, st.Gi, st.Bi, lo, s0, s0 + Ub, K, excl=pos)
    torch.cuda.synchronize()
    assert int(pi.min()) >= lo and int(pi.max()) < hi
    assert bool((pv[:, :-1] >= pv[:, 1:]).all())
    # the C oracle on a sample: scores of the shard's items only, exclusions re-based to local ids
    n = 4
    ip = cpu(pos.indptr[s0:s0 + n + 1])
    ix = cpu(pos.indices[int(ip[0]):int(ip[-1])]).astype(np.int64)
    rows = np.repeat(np.arange(n), np.diff(ip))
    keep = (ix >= lo) & (ix < hi)
    lip = np.concatenate([[0], np.cumsum(np.bincount(rows[keep], minlength=n))]).astype(np.int64)
    ei, ev = cref.score_topk_f32(cpu(st.Gu[s0:s0 + n]), cpu(st.Gi), cpu(st.Bi), 0, n, K, excl=(lip, (ix[keep] - lo).astype(np.int32)))
    assert np.array_equal(cpu(pi[:n]) - lo, ei) and np.array_equal(cpu(pv[:n]), ev)
    # the shard cut in two, each half scored with its own offset, el_allgather_topk + el_topk_merge: the unsplit lists
    mid = (hi - lo) // 2
    a_i, a_v = ops.score_topk(ctx, st.Gu, st.Gi[:mid], st.Bi[:mid], s0, s0 + Ub, K, excl=pos, item_offset=lo)
    b_i, b_v = ops.score_topk(ctx, st.Gu, st.Gi[mid:], st.Bi[mid:], s0, s0 + Ub, K, excl=pos, item_offset=lo + mid)
    ga_i, ga_v = coll.all_gather_topk(a_i, a_v)
    gb_i, gb_v = coll.all_gather_topk(b_i, b_v)
    mi, mv = ops.topk_merge(ctx, torch.cat([ga_i, gb_i]), torch.cat([ga_v, gb_v]))
    assert torch.equal(mi, pi) and torch.equal(mv.view(torch.int32), pv.view(torch.int32))


def test_rows_exchange_step_at_the_c4_item_shard_shape(ctx, shard):
    """parallel.ShardedBprmf (what pick_exchange chooses at 10 M users / 8 ranks / B = 2^20: all-gather of per-triplet rows) on the
    same shard: the replicated user table with its full Adam state, one step against the oracle on sampled rows."""
    if shard["name"] != "c4":
        pytest.skip("the rows exchange keeps Adam slots for every replicated user row: 154 GB at 50 M x 256 -- the dense exchange is that shape's")
    be0, coll, pos_shard, g = shard["be"], shard["coll"], shard["pos_shard"], shard["g"]
    U, F, Ish, dev = shard["U"], shard["F"], shard["hi"] - shard["lo"], ctx.device
    assert parallel.pick_exchange(U, B, WORLD) == "rows"
    be = parallel.HipBackend(ctx, be0.state.Gu, be0.state.Gi, be0.state.Bi, optimizer="adam_tf_dense")
    trainer = parallel.ShardedBprmf(be, coll)
    st = be.state
    u, i, j = ops.bpr_sample(ctx, pos_shard, B, seed=43, first_sample=0, item_lo=0, item_hi=Ish)
    su = torch.randperm(U, generator=g, device=dev)[:1024]
    si = torch.randperm(Ish, generator=g, device=dev)[:1024]
    exp = _oracle_rows(U, Ish, st.Gu, st.Gi, st.Bi, u, i, j, su, si)
    th = {"Gu": cpu(st.Gu[su]), "Gi": cpu(st.Gi[si]), "Bi": cpu(st.Bi[si])}
    ref = _loss64(st.Gu, st.Gi, st.Bi, u, i, j, F)
    trainer.train_step(u, i, j, LR, L_W, L_B)
    loss = trainer.pop_loss()
    assert abs(loss - ref) <= 1e-4 * abs(ref), (loss, ref)
    for name, gname, rows in (("Gu", "gGu", su), ("Gi", "gGi", si), ("Bi", "gBi", si)):
        mm, vv = np.zeros_like(th[name]), np.zeros_like(th[name])
        ob.adam_tf_sparse_apply(th[name], mm, vv, exp[gname][0].astype(np.float32), LR, 1)
        err = np.abs(cpu(getattr(st, name)[rows]) - th[name])
        assert float((err > 2e-5).mean()) <= 2e-4 and float(err.max()) < 3 * LR, (name, float(err.max()))
    del be, trainer
    torch.cuda.empty_cache()
