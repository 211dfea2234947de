"""N > 1 path on CPU: world_size-2 gloo runs of the item-sharded collectives (elliot_amd/parallel.py) with a NumPy
backend standing in for the HIP kernels (the oracle as checker).  Verifies the design claims:
  * sharded top-k + all-gather + merge == single-shard top-k
  * G ranks x B triplets == ONE step of the reference semantics on the concatenated batch (user replicas identical)
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from elliot_amd import parallel
from oracle import bprmf_batch as ob
from oracle import cref


class NumpyBackend:
    def __init__(self, Gu, Gi, Bi, lr_opt="adam_tf_dense"):
        self.Gu, self.Gi, self.Bi = Gu.copy(), Gi.copy(), Bi.copy()
        z = np.zeros_like
        self.g = [z(self.Bi), z(self.Gu), z(self.Gi)]
        self.m = [z(self.Bi), z(self.Gu), z(self.Gi)]
        self.v = [z(self.Bi), z(self.Gu), z(self.Gi)]
        self.loss = torch.zeros(1, dtype=torch.float64)
        self.t = 0

    def shard_grads(self, u, i, j, l_w, l_b):
        u, i, j = (x.numpy().astype(np.int64) for x in (u, i, j))
        self.loss += float(ob.forward_loss(self.Gu, self.Gi, self.Bi, u, i, j, l_w, l_b))
        dBi, _, dGi = ob.gradients(self.Gu, self.Gi, self.Bi, u, i, j, l_w, l_b)
        self.g[0] += dBi
        self.g[2] += dGi
        gu, gi, gj = self.Gu[u], self.Gi[i], self.Gi[j]
        d = (self.Bi[i] + (gu * gi).sum(1)) - (self.Bi[j] + (gu * gj).sum(1))
        s = np.where(d >= -80.0, -1.0 / (1.0 + np.exp(d.astype(np.float64))), 0.0).astype(np.float32)
        return torch.from_numpy((s[:, None] * (gi - gj) + np.float32(l_w) * gu).astype(np.float32))

    def reduce_user_rows(self, ids, rows):
        np.add.at(self.g[1], ids.numpy().astype(np.int64), rows.numpy())

    def apply(self, lr):
        self.t += 1
        for th, g, m, v in zip((self.Bi, self.Gu, self.Gi), self.g, self.m, self.v):
            ob.adam_tf_sparse_apply(th, m, v, g, lr, self.t)
            g[:] = 0

    def local_loss_tensor(self):
        return self.loss


class NumpyDenseBackend:
    """Stand-in for parallel.HipDenseBackend: padded user table replica, dense gradient table, optimiser state for the
    rank's own user rows only."""

    def __init__(self, Gu, Gi, Bi, rank, world):
        U, F = Gu.shape
        self.U, self.rank, self.world = U, rank, world
        self.Us = parallel.user_shard_rows(U, world)
        Gu_pad = np.zeros((self.Us * world, F), np.float32)
        Gu_pad[:U] = Gu
        z = np.zeros_like
        self.state = type("S", (), {})()
        self.state.Gu = torch.from_numpy(Gu_pad)                   # torch views: the collectives work on these buffers
        self.Gi, self.Bi = Gi.copy(), Bi.copy()
        self.gGu = torch.zeros_like(self.state.Gu)
        self.gGi, self.gBi = z(self.Gi), z(self.Bi)
        lo = rank * self.Us
        self.Gu_own = self.state.Gu[lo:lo + self.Us]
        self.g_own = torch.zeros((self.Us, F), dtype=torch.float32)
        self.m = [z(self.Bi), np.zeros((self.Us, F), np.float32), z(self.Gi)]
        self.v = [z(self.Bi), np.zeros((self.Us, F), np.float32), z(self.Gi)]
        self.loss = torch.zeros(1, dtype=torch.float64)
        self.t = 0

    def grads(self, u, i, j, l_w, l_b):
        u, i, j = (x.numpy().astype(np.int64) for x in (u, i, j))
        Gu = self.state.Gu.numpy()
        self.loss += float(ob.forward_loss(Gu, self.Gi, self.Bi, u, i, j, l_w, l_b))
        dBi, dGu, dGi = ob.gradients(Gu, self.Gi, self.Bi, u, i, j, l_w, l_b)
        self.gBi += dBi
        self.gGi += dGi
        self.gGu += torch.from_numpy(dGu)
        return self.gGu

    def apply_own(self, lr):
        self.t += 1
        own = self.Gu_own.numpy()                                  # shares memory with the padded table
        for th, g, m, v in zip((self.Bi, own, self.Gi), (self.gBi, self.g_own.numpy(), self.gGi), self.m, self.v):
            ob.adam_tf_sparse_apply(th, m, v, g, lr, self.t)
            g[:] = 0

    def local_loss_tensor(self):
        return self.loss


def _dense_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(1)
        U, I, F, B = 61, 80, 8, 96                                  # U not divisible by the world size: padded rows
        Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
        Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
        Bi = rs.normal(scale=0.01, size=I).astype(np.float32)
        lo, hi = parallel.item_range(I, rank, world)
        coll = parallel._Collectives()
        be = NumpyDenseBackend(Gu, Gi[lo:hi], Bi[lo:hi], rank, world)
        tr = parallel.ShardedBprmfDense(be, coll)
        ref = ob.BPRMFBatchOracle(Gu, Gi, Bi, 0.01, 0.1, 0.001)
        for step in range(3):
            batches = []
            for r in range(world):
                brs = np.random.RandomState(200 + 10 * step + r)
                l, h = parallel.item_range(I, r, world)
                batches.append((brs.randint(0, U, B), brs.randint(l, h, B), brs.randint(l, h, B)))
            u, i, j = batches[rank]
            tr.train_step(torch.from_numpy(u.astype(np.int32)), torch.from_numpy((i - lo).astype(np.int32)),
                          torch.from_numpy((j - lo).astype(np.int32)), 0.01, 0.1, 0.001)
            loss = tr.pop_loss()
            cu, ci, cj = (np.concatenate([b[x] for b in batches]) for x in range(3))
            ref_loss = ref.train_step((cu, ci, cj))
            assert abs(loss - ref_loss) < 1e-4 * abs(ref_loss), (loss, ref_loss)
            assert np.abs(be.state.Gu.numpy()[:U] - ref.Gu).max() < 2e-6
            assert np.abs(be.state.Gu.numpy()[U:]).max() == 0.0                      # padding rows never move
            assert np.abs(be.Gi - ref.Gi[lo:hi]).max() < 2e-6 and np.abs(be.Bi - ref.Bi[lo:hi]).max() < 2e-6
            assert float(be.gGu.abs().max()) == 0.0                                  # accumulators zero on exit
        t = be.state.Gu.clone()
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_item_sharded_training_dense_exchange_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_dense_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_pick_exchange():
    assert parallel.pick_exchange(1_000_000, 1 << 20, 1) == "rows"
    assert parallel.pick_exchange(1_000_000, 1 << 20, 8) == "dense"       # 8 x 1M triplets touch every user row anyway
    assert parallel.pick_exchange(10_000_000, 65536, 8) == "rows"          # few triplets, huge table: ship the rows
    assert parallel.user_shard_rows(61, 2) == 31


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(0)
        U, I, F, B = 60, 80, 8, 96
        Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
        Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
        Bi = rs.normal(scale=0.01, size=I).astype(np.float32)
        lo, hi = parallel.item_range(I, rank, world)
        coll = parallel._Collectives()
        be = NumpyBackend(Gu, Gi[lo:hi], Bi[lo:hi])
        tr = parallel.ShardedBprmf(be, coll)
        ref = ob.BPRMFBatchOracle(Gu, Gi, Bi, 0.01, 0.1, 0.001)
        losses = []
        for step in range(3):
            batches = []
            for r in range(world):                                    # every rank can rebuild every rank's batch
                brs = np.random.RandomState(100 + 10 * step + r)
                l, h = parallel.item_range(I, r, world)
                batches.append((brs.randint(0, U, B), brs.randint(l, h, B), brs.randint(l, h, B)))
            u, i, j = batches[rank]
            tr.train_step(torch.from_numpy(u.astype(np.int32)), torch.from_numpy((i - lo).astype(np.int32)),
                          torch.from_numpy((j - lo).astype(np.int32)), 0.01, 0.1, 0.001)
            losses.append(tr.pop_loss())
            cu, ci, cj = (np.concatenate([b[x] for b in batches]) for x in range(3))
            ref_loss = ref.train_step((cu, ci, cj))
            assert abs(losses[-1] - ref_loss) < 1e-4 * abs(ref_loss), (losses[-1], ref_loss)
            assert np.abs(be.Gu - ref.Gu).max() < 2e-6
            assert np.abs(be.Gi - ref.Gi[lo:hi]).max() < 2e-6 and np.abs(be.Bi - ref.Bi[lo:hi]).max() < 2e-6
        # replicas of the user table are bit-identical across ranks
        t = torch.from_numpy(be.Gu.copy())
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert all(torch.equal(gathered[0], g) for g in gathered)

        # ---- sharded top-k: local oracle top-k on the shard, all-gather, merge by (score desc, index asc)
        k = 7
        pi, pv = cref.score_topk_f32(ref.Gu, ref.Gi[lo:hi], ref.Bi[lo:hi], 0, U, k, item_offset=lo)
        gi = coll.all_gather(torch.from_numpy(pi)).reshape(world, U, k).numpy()
        gv = coll.all_gather(torch.from_numpy(pv)).reshape(world, U, k).numpy()
        full_i, full_v = cref.score_topk_f32(ref.Gu, ref.Gi, ref.Bi, 0, U, k)
        for uu in range(U):
            ci_, cv_ = gi[:, uu].reshape(-1), gv[:, uu].reshape(-1)
            order = np.lexsort((ci_, -cv_))[:k]
            assert np.array_equal(ci_[order], full_i[uu]) and np.array_equal(cv_[order], full_v[uu])
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_item_sharded_training_and_topk_world2_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    assert dict(out) == {0: 1, 1: 1}


def test_shard_csr_and_item_range():
    indptr = torch.tensor([0, 3, 3, 6], dtype=torch.int64)
    indices = torch.tensor([1, 5, 9, 0, 4, 8], dtype=torch.int32)
    ip, ix = parallel.shard_csr(indptr, indices, 4, 9)
    assert ip.tolist() == [0, 1, 1, 3] and ix.tolist() == [1, 0, 4]
    assert [parallel.item_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]


def _gather_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        I, F = 11, 3                                                  # uneven shards: 5 + 6 rows
        Gi = torch.arange(I * F, dtype=torch.float32).reshape(I, F)
        Bi = torch.arange(I, dtype=torch.float32) * 10
        lo, hi = parallel.item_range(I, rank, world)
        g, b = parallel.gather_item_table(parallel._Collectives(), Gi[lo:hi].contiguous(), Bi[lo:hi].contiguous(), I)
        assert torch.equal(g, Gi) and torch.equal(b, Bi)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_gather_item_table_world2_gloo():
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_gather_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


class NumpyNmfBackend:
    """Stand-in for ops.NmfDeviceState in parallel.ShardedNmf: local item shard, replicated user tables / MLP / head."""

    def __init__(self, w, lr):
        from oracle import neumf as on
        self.on = on
        self.orc = on.NeuMFOracle(w, lr)
        self.loss = torch.zeros(1, dtype=torch.float64)
        self.g = None

    def grads(self, u, i, label, n_global):
        on, w = self.on, self.orc.w
        u, i, y = u.numpy().astype(np.int64), i.numpy().astype(np.int64), label.numpy()
        c = on.forward(w, u, i)
        scale = np.float32(len(y) / n_global)
        self.loss += on.bce(c["p"], y) * float(scale)
        g = on.gradients(w, c, u, i, y)
        self.g = {k: ([torch.from_numpy((x * scale).astype(np.float32)) for x in v] if isinstance(v, list)
                      else torch.from_numpy((v * scale).astype(np.float32))) for k, v in g.items()}

    def replicated_grads(self, shard="user"):
        out = [self.g[k] for k in (("Imf", "Imlp") if shard == "user" else ("Umf", "Umlp")) if k in self.g]
        out += list(self.g.get("W", [])) + list(self.g.get("b", [])) + [self.g["hw"]]
        if "hb" in self.g:
            out.append(self.g["hb"])
        return out

    def apply(self, lr):
        o = self.orc
        o.t += 1
        from oracle.bprmf_batch import adam_tf_sparse_apply
        npg = lambda t: t.numpy()
        for k in ("Umf", "Imf", "Umlp", "Imlp"):
            if k in o.w:
                adam_tf_sparse_apply(o.w[k], o.m[k], o.v[k], npg(self.g[k]), o.lr, o.t)
        if "W" in o.w:
            for l in range(len(o.w["W"])):
                o._dense(o.w["W"][l], o.m["W"][l], o.v["W"][l], npg(self.g["W"][l]))
                o._dense(o.w["b"][l], o.m["b"][l], o.v["b"][l], npg(self.g["b"][l]))
        o._dense(o.w["hw"], o.m["hw"], o.v["hw"], npg(self.g["hw"]))
        if "hb" in o.w:
            o._dense(o.w["hb"], o.m["hb"], o.v["hb"], npg(self.g["hb"]))


def _nmf_worker(rank, world, port, out, shard="item"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import neumf as on
        U, I, F, n = 41, 30, 8, 64
        w = on.init_neumf(U, I, F, seed=3)
        by_user = shard == "user"
        rng = (lambda r: parallel.user_range(U, r, world)) if by_user else (lambda r: parallel.item_range(I, r, world))
        lo, hi = rng(rank)
        wl = {k: ([x.copy() for x in v] if isinstance(v, list) else v.copy()) for k, v in w.items()}
        for k in (("Umf", "Umlp") if by_user else ("Imf", "Imlp")):
            wl[k] = w[k][lo:hi].copy()
        be = NumpyNmfBackend(wl, 0.01)
        tr = parallel.ShardedNmf(be, parallel._Collectives(), shard=shard)
        ref = on.NeuMFOracle(w, 0.01)
        for step in range(3):
            batches = []
            for r in range(world):
                brs = np.random.RandomState(300 + 10 * step + r)
                l, h = rng(r)
                us = brs.randint(l, h, n) if by_user else brs.randint(0, U, n)
                its = brs.randint(0, I, n) if by_user else brs.randint(l, h, n)
                batches.append((us, its, brs.randint(0, 2, n).astype(np.float32)))
            u, i, y = batches[rank]
            ul, il = (u - lo, i) if by_user else (u, i - lo)
            tr.train_step(torch.from_numpy(ul.astype(np.int32)), torch.from_numpy(il.astype(np.int32)), torch.from_numpy(y), 0.01)
            loss = tr.pop_loss()
            cu, ci, cy = (np.concatenate([b[x] for b in batches]) for x in range(3))
            ref_loss = ref.train_step(cu, ci, cy)
            assert abs(loss - ref_loss) < 1e-5 * max(1.0, abs(ref_loss)), (loss, ref_loss)
            o = be.orc.w
            su, si = (slice(lo, hi), slice(None)) if by_user else (slice(None), slice(lo, hi))
            assert np.abs(o["Umf"] - ref.w["Umf"][su]).max() < 3e-6 and np.abs(o["Umlp"] - ref.w["Umlp"][su]).max() < 3e-6
            assert np.abs(o["Imf"] - ref.w["Imf"][si]).max() < 3e-6 and np.abs(o["Imlp"] - ref.w["Imlp"][si]).max() < 3e-6
            for a, b in zip(o["W"] + o["b"] + [o["hw"]], ref.w["W"] + ref.w["b"] + [ref.w["hw"]]):
                assert np.abs(a - b).max() < 3e-6
        t = torch.from_numpy(be.orc.w["Imlp" if by_user else "Umlp"].copy())       # a replicated table: identical everywhere
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(240)
@pytest.mark.parametrize("shard", ["user", "item"])
def test_neumf_sharded_step_world2_gloo(shard):
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_nmf_worker, args=(2, port, out, shard), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


class NumpyUserShardBackend:
    """Stand-in for parallel.HipUserShardBackend: local user rows, full item replica."""

    def __init__(self, Gu_shard, Gi, Bi):
        self.Gu, self.Gi, self.Bi = Gu_shard.copy(), Gi.copy(), Bi.copy()
        z = np.zeros_like
        self.gGu = z(self.Gu)
        self.gGi, self.gBi = torch.zeros(Gi.shape, dtype=torch.float32), torch.zeros(Bi.shape, dtype=torch.float32)
        self.m = [z(self.Bi), z(self.Gu), z(self.Gi)]
        self.v = [z(self.Bi), z(self.Gu), z(self.Gi)]
        self.loss = torch.zeros(1, dtype=torch.float64)
        self.t = 0

    def grads(self, u, i, j, l_w, l_b):
        u, i, j = (x.numpy().astype(np.int64) for x in (u, i, j))
        self.loss += float(ob.forward_loss(self.Gu, self.Gi, self.Bi, u, i, j, l_w, l_b))
        dBi, dGu, dGi = ob.gradients(self.Gu, self.Gi, self.Bi, u, i, j, l_w, l_b)
        self.gGu += dGu
        self.gGi += torch.from_numpy(dGi.astype(np.float32))
        self.gBi += torch.from_numpy(dBi.astype(np.float32))

    def item_grads(self):
        return [self.gGi, self.gBi]

    def touched_item_rows(self, i, j):
        ids = torch.unique(torch.cat([i, j]).to(torch.int64))
        return ids.to(torch.int32), self.gGi.index_select(0, ids), self.gBi.index_select(0, ids)

    def set_item_grads(self, ids_all, rows_all, bias_all):
        ids = ids_all.numpy().astype(np.int64)
        touched = np.unique(ids)
        g = np.zeros((self.gGi.shape[0], self.gGi.shape[1] + 1), np.float32)
        np.add.at(g, ids, np.concatenate([rows_all.numpy(), bias_all.numpy()[:, None]], axis=1))     # (gathered order)
        self.gGi[touched] = torch.from_numpy(g[touched, :-1])
        self.gBi[touched] = torch.from_numpy(g[touched, -1])

    def _apply(self, which, lr):
        for n, (th, g, m, v) in enumerate(zip((self.Bi, self.Gu, self.Gi), (self.gBi.numpy(), self.gGu, self.gGi.numpy()), self.m, self.v)):
            if n in which:
                ob.adam_tf_sparse_apply(th, m, v, g, lr, self.t)
                g[:] = 0

    def apply(self, lr):
        self.t += 1
        self._apply((0, 1, 2), lr)

    # the split form ShardedBprmfByUser overlaps with the asynchronous all-reduce of the item gradients
    def begin_step(self):
        self.t += 1
        self.order = []

    def apply_users(self, lr):
        self.order.append("users")
        self._apply((1,), lr)

    def apply_items(self, lr):
        self.order.append("items")
        self._apply((0, 2), lr)

    def local_loss_tensor(self):
        return self.loss


def _user_shard_worker(rank, world, port, out, item_exchange="dense"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rs = np.random.RandomState(2)
        U, I, F, B = 61, 80, 8, 96
        Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
        Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
        Bi = rs.normal(scale=0.01, size=I).astype(np.float32)
        ulo, uhi = parallel.user_range(U, rank, world)
        be = NumpyUserShardBackend(Gu[ulo:uhi], Gi, Bi)
        tr = parallel.ShardedBprmfByUser(be, parallel._Collectives(), item_exchange=item_exchange)
        ref = ob.BPRMFBatchOracle(Gu, Gi, Bi, 0.01, 0.1, 0.001)
        for step in range(3):
            batches = []
            for r in range(world):
                brs = np.random.RandomState(400 + 10 * step + r)
                l, h = parallel.user_range(U, r, world)
                # items: whole catalogue; rows mode: rank 1 draws from a narrower range, so the ranks' lists differ in length (padding)
                hi_i = I if (item_exchange == "dense" or r == 0) else I // 3
                batches.append((brs.randint(l, h, B), brs.randint(0, hi_i, B), brs.randint(0, hi_i, B)))
            u, i, j = batches[rank]
            tr.train_step(torch.from_numpy((u - ulo).astype(np.int32)), torch.from_numpy(i.astype(np.int32)),
                          torch.from_numpy(j.astype(np.int32)), 0.01, 0.1, 0.001)
            loss = tr.pop_loss()
            assert be.order == ["users", "items"]            # own rows under the collective, the replica after it
            if item_exchange == "rows":
                n, n_max = tr.last_rows
                assert n <= n_max and (rank == 0 or n < n_max)                # rank 1's list is the shorter one: padded records in play
            cu, ci, cj = (np.concatenate([b[x] for b in batches]) for x in range(3))
            ref_loss = ref.train_step((cu, ci, cj))
            assert abs(loss - ref_loss) < 1e-4 * abs(ref_loss), (loss, ref_loss)
            assert np.abs(be.Gu - ref.Gu[ulo:uhi]).max() < 2e-6
            assert np.abs(be.Gi - ref.Gi).max() < 2e-6 and np.abs(be.Bi - ref.Bi).max() < 2e-6
        t = torch.from_numpy(be.Gi.copy())                           # item replicas bit-identical across ranks
        gathered = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(gathered, t)
        assert all(torch.equal(gathered[0], g) for g in gathered)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("item_exchange", ["dense", "rows"])
def test_user_sharded_training_world2_gloo(item_exchange):
    """User shards over two gloo ranks == one reference-semantics step on the concatenated batch -- with the item gradients meeting in
    one dense all-reduce, and with the row-sparse exchange (all-gather of each rank's touched (item id, gradient row) records, lists of
    unequal length, segment sum in the gathered order on every rank)."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_user_shard_worker, args=(2, port, out, item_exchange), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}
    assert parallel.pick_item_exchange(5_000_000, 256, 1 << 20, 8) == "dense" and parallel.pick_item_exchange(5_000_000, 256, 1 << 14, 8) == "rows"
    assert [parallel.user_range(10, r, 3) for r in range(3)] == [(0, 3), (3, 6), (6, 10)]


# ------------------------------------------------------------------------------------------ point-wise models, user shards
class NumpyPwmfBackend:
    """Stand-in for ops.PwmfDeviceState on (local user rows, full item tables): oracle/pointwise_mf.py arithmetic."""

    def __init__(self, weights, kind, lr, optimizer, alpha=0.0, l_w=0.0):
        from oracle import pointwise_mf as pw
        self.pw = pw
        self.o = pw.PointwiseOracle(weights, kind, lr, optimizer=optimizer, alpha=alpha, l_w=l_w)
        self.g = {}
        self.gt = {k: torch.zeros(v.shape, dtype=torch.float32) for k, v in self.o.w.items() if k in ("Gi", "Bi")}
        self.loss = torch.zeros(1, dtype=torch.float64)
        self.log = []

    def grads(self, u, i, y, n_global=None, side="both"):
        u, i, y = u.numpy().astype(np.int64), i.numpy().astype(np.int64), y.numpy()
        n = len(y)
        loss, g = self.pw.loss_and_grads(self.o.w, self.o.kind, u, i, y, self.o.alpha, self.o.l_w)
        scale = 1.0 if self.o.kind == "logistic" else n / float(n_global)       # batch MEAN over the global batch
        self.loss += loss * scale
        self.g = {k: (v * np.float32(scale)).astype(np.float32) for k, v in g.items()}
        for k in self.gt:
            self.gt[k].copy_(torch.from_numpy(self.g[k]))

    def item_grads(self):
        return [self.gt[k] for k in ("Gi", "Bi") if k in self.gt]

    def apply(self, lr, side="both", advance=True):
        o = self.o
        if advance:
            o.t += 1
        self.log.append(side)
        for k in {"users": ("Gu", "Bu"), "items": ("Gi", "Bi")}[side]:
            if k not in o.w:
                continue
            g = self.gt[k].numpy() if k in self.gt else self.g[k]
            if o.optimizer == "adam":
                ob.adam_tf_sparse_apply(o.w[k], o.m[k], o.v[k], g.astype(np.float32), o.lr, o.t)
            else:
                self.pw.adagrad_apply(o.w[k], o.m[k], g.astype(np.float32), o.lr)


def _pwmf_worker(rank, world, port, out, model):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import pointwise_mf as pw
        kind, bias, opt = {"FunkSVD": ("mse", True, "adam"), "PMF": ("mse_sigmoid", False, "adam"),
                           "LogisticMF": ("logistic", True, "adagrad")}[model]
        rs = np.random.RandomState(3)
        U, I, F, n, lr = 41, 30, 6, 64, 0.01
        w = {"Gu": rs.normal(scale=0.3, size=(U, F)).astype(np.float32), "Gi": rs.normal(scale=0.3, size=(I, F)).astype(np.float32)}
        if bias:
            w["Bu"], w["Bi"] = rs.normal(scale=0.1, size=U).astype(np.float32), rs.normal(scale=0.1, size=I).astype(np.float32)
        ulo, uhi = parallel.user_range(U, rank, world)
        local = {k: (v[ulo:uhi] if k in ("Gu", "Bu") else v) for k, v in w.items()}
        be = NumpyPwmfBackend(local, kind, lr, opt, alpha=0.5, l_w=0.02)
        tr = parallel.ShardedPwmf(be, parallel._Collectives())
        ref = pw.PointwiseOracle(w, kind, lr, optimizer=opt, alpha=0.5, l_w=0.02)
        sides = ("items", "users") if kind == "logistic" else ("both",)
        for step in range(4):
            batches = []
            for r in range(world):
                brs = np.random.RandomState(900 + 10 * step + r)
                l, h = parallel.user_range(U, r, world)
                batches.append((brs.randint(l, h, n), brs.randint(0, I, n), brs.randint(0, 2, n).astype(np.float32)))
            u, i, y = batches[rank]
            side = sides[step % len(sides)]
            tr.train_step(torch.from_numpy((u - ulo).astype(np.int32)), torch.from_numpy(i.astype(np.int32)), torch.from_numpy(y), lr,
                          side=side)
            loss = tr.pop_loss()
            cu, ci, cy = (np.concatenate([b[x] for b in batches]) for x in range(3))
            ref_loss = ref.train_step((cu, ci, cy), side=side)
            assert abs(loss - ref_loss) <= 1e-5 * max(abs(ref_loss), 1e-6), (step, loss, ref_loss)
            assert np.abs(be.o.w["Gu"] - ref.w["Gu"][ulo:uhi]).max() < 2e-6
            assert np.abs(be.o.w["Gi"] - ref.w["Gi"]).max() < 2e-6
            if bias:
                assert np.abs(be.o.w["Bu"] - ref.w["Bu"][ulo:uhi]).max() < 2e-6 and np.abs(be.o.w["Bi"] - ref.w["Bi"]).max() < 2e-6
        assert be.log == (["items", "users"] * 2 if kind == "logistic" else ["users", "items"] * 4)     # own rows first, under the collective
        out[rank] = 1
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("model", ["FunkSVD", "PMF", "LogisticMF"])
def test_pointwise_models_user_sharded_world2_gloo(model):
    """SURVEY 8e for the N3 siblings: G ranks x n samples = one reference-semantics step on the concatenated batch."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_pwmf_worker, args=(2, port, out, model), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


# ------------------------------------------------------------------------------------------ CML: the all-gather of D, E
class NumpyCmlBackend:
    """Stand-in for ops.CmlDeviceState on (local user rows, full item tables), oracle/cml.py's separable form."""

    def __init__(self, Gu_local, Gi, Bi, lr):
        from oracle import cml as oc
        self.oc, self.lr = oc, lr
        self.w = {"Gu": Gu_local.copy(), "Gi": Gi.copy(), "Bi": Bi.copy()}
        self.m = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.v = {k: np.zeros_like(v) for k, v in self.w.items()}
        self.flat = torch.zeros(Gi.size + Bi.size, dtype=torch.float32)
        self.loss = torch.zeros(1, dtype=torch.float64)
        self.t = 0

    def forward_de(self, u, i, j, l_w, l_b):
        u, i, j = (x.numpy().astype(np.int64) for x in (u, i, j))
        D, E = self.oc.distances(self.w["Gu"], self.w["Gi"], self.w["Bi"], u, i, j)
        self.loss += self.oc.regulariser(self.w["Gu"], self.w["Gi"], self.w["Bi"], u, i, j, l_w, l_b)
        return torch.from_numpy(D.astype(np.float32)), torch.from_numpy(E.astype(np.float32))

    def grads_de(self, u, i, j, l_w, l_b, margin, D, E, D_all, E_all):
        u, i, j = (x.numpy().astype(np.int64) for x in (u, i, j))
        cD, cE, hinge = self.oc.coefficients(D.numpy(), E.numpy(), D_all.numpy(), E_all.numpy(), margin)
        self.loss += hinge
        self.gGu, dGi, dBi = self.oc.row_gradients(self.w["Gu"], self.w["Gi"], self.w["Bi"], u, i, j, cD, cE, l_w, l_b)
        self.flat.copy_(torch.from_numpy(np.concatenate([dGi.reshape(-1), dBi]).astype(np.float32)))

    def item_grads(self):
        return [self.flat]

    def begin_step(self):
        self.t += 1

    def apply_users(self, lr):
        ob.adam_tf_sparse_apply(self.w["Gu"], self.m["Gu"], self.v["Gu"], self.gGu.astype(np.float32), lr, self.t)

    def apply_items(self, lr):
        g = self.flat.numpy()
        nI = self.w["Gi"].size
        ob.adam_tf_sparse_apply(self.w["Gi"], self.m["Gi"], self.v["Gi"], g[:nI].reshape(self.w["Gi"].shape), lr, self.t)
        ob.adam_tf_sparse_apply(self.w["Bi"], self.m["Bi"], self.v["Bi"], g[nI:].copy(), lr, self.t)


def _cml_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import cml as oc
        rs = np.random.RandomState(8)
        U, I, F, B, lr, l_w, l_b, margin = 37, 28, 5, 48, 0.01, 0.01, 0.02, 0.5
        Gu, Gi = rs.normal(scale=0.4, size=(U, F)).astype(np.float32), rs.normal(scale=0.4, size=(I, F)).astype(np.float32)
        Bi = rs.normal(scale=0.2, size=I).astype(np.float32)
        ulo, uhi = parallel.user_range(U, rank, world)
        be = NumpyCmlBackend(Gu[ulo:uhi], Gi, Bi, lr)
        tr = parallel.ShardedCml(be, parallel._Collectives())
        ref = oc.CMLOracle(Gu, Gi, Bi, lr, l_w, l_b, margin)
        for step in range(3):
            batches = []
            for r in range(world):
                brs = np.random.RandomState(700 + 10 * step + r)
                l, h = parallel.user_range(U, r, world)
                batches.append((brs.randint(l, h, B), brs.randint(0, I, B), brs.randint(0, I, B)))
            u, i, j = batches[rank]
            tr.train_step(torch.from_numpy((u - ulo).astype(np.int32)), torch.from_numpy(i.astype(np.int32)),
                          torch.from_numpy(j.astype(np.int32)), lr, l_w, l_b, margin)
            loss = tr.pop_loss()
            cu, ci, cj = (np.concatenate([b[x] for b in batches]) for x in range(3))
            ref_loss = ref.train_step((cu, ci, cj))                   # the [2B, 2B] hinge of the concatenated batch
            assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss), (step, loss, ref_loss)
            assert np.abs(be.w["Gu"] - ref.Gu[ulo:uhi]).max() < 1e-5 and np.abs(be.w["Gi"] - ref.Gi).max() < 1e-5
            assert np.abs(be.w["Bi"] - ref.Bi).max() < 1e-5
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_cml_user_sharded_world2_gloo_with_the_all_gather_of_distances():
    """CML's hinge couples the whole batch: the ranks all-gather D and E (a real exchange step), then G ranks x B triplets are
    one reference-semantics step on the concatenated batch."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_cml_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}


# ------------------------------------------------------------------------------------------ Mult-VAE, data parallel
class NumpyVaeBackend:
    """Stand-in for ops.VaeDeviceState: oracle/multi_vae.py forward / gradients on the rank's rows of a dense batch."""

    def __init__(self, weights, lr):
        from oracle import multi_vae as ov
        self.ov = ov
        self.o = ov.MultiVAEOracle(weights, lr)
        self.gt = {k: torch.zeros(v.shape, dtype=torch.float32) for k, v in self.o.w.items()}
        self.loss = torch.zeros(1, dtype=torch.float64)

    def grads(self, X, rows, anneal, eps=None, dropout_rate=0.0, dropout_seed=42, n_global=None):
        x = X[rows.numpy()]
        c = self.ov.forward(self.o.w, x, eps.numpy())
        scale = len(x) / float(n_global)                       # both losses are batch MEANS: local mean x n / n_global
        self.loss += self.ov.loss_from(c, anneal) * scale
        g = self.ov.gradients(self.o.w, c, np.float32(anneal))
        for k in self.gt:
            self.gt[k].copy_(torch.from_numpy((g[k] * np.float32(scale)).astype(np.float32)))

    def dense_grads(self):
        return [self.gt[k] for k in self.ov.NAMES]

    def apply(self, lr):
        o, f = self.o, np.float32
        o.t += 1
        a = self.ov.adam_lr_t(o.lr, o.t)
        for k in self.ov.NAMES:
            gg = self.gt[k].numpy()
            o.m[k] += (gg - o.m[k]) * f(1 - self.ov.BETA1)
            o.v[k] += (gg * gg - o.v[k]) * f(1 - self.ov.BETA2)
            o.w[k] -= (o.m[k] * a) / (np.sqrt(o.v[k]) + f(self.ov.EPS))


def _vae_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import multi_vae as ov
        rs = np.random.RandomState(4)
        U, I, H, L, n, lr = 40, 50, 12, 5, 8, 0.005
        X = (rs.rand(U, I) < 0.15).astype(np.float32)
        X[X.sum(1) == 0, 0] = 1
        w0 = ov.init_weights(I, H, L, 2)
        be = NumpyVaeBackend(w0, lr)
        tr = parallel.ShardedVae(be, parallel._Collectives())
        ref = ov.MultiVAEOracle(w0, lr)
        for step in range(3):
            brs = np.random.RandomState(50 + step)
            rows_all = brs.permutation(U)[:world * n]
            eps_all = brs.normal(size=(world * n, L)).astype(np.float32)
            rows, eps = rows_all[rank * n:(rank + 1) * n], eps_all[rank * n:(rank + 1) * n]
            tr.train_step(X, torch.from_numpy(rows), lr, 0.1, eps=torch.from_numpy(eps))
            loss = tr.pop_loss()
            ref_loss = ref.train_step(X[rows_all], eps_all, 0.1)
            assert abs(loss - ref_loss) <= 1e-5 * abs(ref_loss), (step, loss, ref_loss)
            for k in ov.NAMES:
                assert np.abs(be.o.w[k] - ref.w[k]).max() < 2e-6, (step, k)
        out[rank] = 1
    finally:
        dist.destroy_process_group()


def test_multivae_data_parallel_world2_gloo():
    """the rows of a batch split over the ranks, gradients of the replicated weights all-reduced: G ranks x n
    rows = one reference-semantics step on the batch of G n rows."""
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_vae_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: 1, 1: 1}
