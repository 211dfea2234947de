"""GPU parity: Collaborative Metric Learning (el_cml_*) against oracle/cml.py -- the reference's [B,B] broadcast hinge
evaluated by sorting, the direct-formula scoring, and the plugin end to end."""
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from elliot_amd import ops
from oracle import cml as oc
from tests.gpu_util import cpu, random_excl

pytestmark = pytest.mark.gpu


def tables(rs, U, I, F, scale=0.3):
    return (rs.normal(scale=scale, size=(U, F)).astype(np.float32), rs.normal(scale=scale, size=(I, F)).astype(np.float32),
            rs.normal(scale=0.2, size=I).astype(np.float32))


def dev(ctx, a):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=np.int32)).to(ctx.device)


@pytest.mark.parametrize("F,margin,scale", [(8, 0.5, 0.3), (10, 0.5, 0.3), (64, 0.2, 0.1), (200, 1.5, 0.05), (16, 0.5, 4.0)])
def test_train_steps_match_oracle(ctx, F, margin, scale):
    """scale 4.0: distances of order 100 put many pairs outside the -80 clip (constant terms, zero gradient)."""
    rs = np.random.RandomState(F)
    U, I, B, lr, l_w, l_b = 150, 120, 700, 0.01, 0.01, 0.02
    Gu, Gi, Bi = tables(rs, U, I, F, scale)
    st = ops.CmlDeviceState(ctx, Gu, Gi, Bi)
    orc = oc.CMLOracle(Gu, Gi, Bi, lr, l_w, l_b, margin)
    for s in range(4):
        n = B if s != 2 else 33
        u, i, j = rs.randint(0, U, n), rs.randint(0, min(I, 40), n), rs.randint(0, I, n)
        st.train_step(dev(ctx, u), dev(ctx, i), dev(ctx, j), lr, l_w, l_b, margin)
        got, exp = st.pop_loss(), orc.train_step((u, i, j))
        assert abs(got - exp) <= 1e-4 * abs(exp), (s, got, exp)
        for name, ref in (("Gu", orc.Gu), ("Gi", orc.Gi), ("Bi", orc.Bi)):
            err = np.abs(cpu(getattr(st, name)) - ref)
            # a pair sitting within rounding of the hinge flips one unit of a coefficient; Adam turns that into <= lr
            assert (err > 5e-5).mean() < 5e-3 and err.max() < 5 * lr, (s, name, float(err.max()), float((err > 5e-5).mean()))
        assert not cpu(st.gGu).any() and not cpu(st.gGi).any() and not cpu(st.gBi).any()


def test_large_batch_counts(ctx):
    """B = 20000: 4e8 pairs -- the counts come from sorted searches, the oracle loops over row blocks of the [B,B] matrix."""
    rs = np.random.RandomState(1)
    U, I, F, B, margin = 3000, 800, 32, 20000, 0.5
    Gu, Gi, Bi = tables(rs, U, I, F, 0.2)
    st = ops.CmlDeviceState(ctx, Gu, Gi, Bi)
    u, i, j = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
    st.train_step(dev(ctx, u), dev(ctx, i), dev(ctx, j), 0.001, 0.0, 0.0, margin)
    got = st.pop_loss()
    D = (np.sum((Gu[u] - Gi[j]) ** 2, -1) - np.sum((Gu[u] - Gi[i]) ** 2, -1)).astype(np.float64)
    E = (Bi[i] - Bi[j]).astype(np.float64)
    exp = 0.0
    for a0 in range(0, B, 2000):
        diff = np.clip(D[a0:a0 + 2000, None] + E[None, :], -80, 1e8)
        exp += np.maximum(margin - diff, 0).sum()
    assert abs(got - exp) <= 2e-5 * exp, (got, exp)


def test_recommend_values_are_the_direct_formula(ctx):
    rs = np.random.RandomState(4)
    U, I, F, k = 300, 1100, 32, 10
    Gu, Gi, Bi = tables(rs, U, I, F, 0.3)
    st = ops.CmlDeviceState(ctx, Gu, Gi, Bi)
    indptr, indices = random_excl(rs, U, I, 0, 30)
    excl = ops.DeviceCSR(indptr, indices, I, ctx.device)
    idx, val = (cpu(t) for t in st.recommend(0, U, k, excl=excl))
    scores = oc.CMLOracle(Gu, Gi, Bi, 0, 0, 0, 0).predict(0, U).astype(np.float64)
    for r in range(U):
        scores[r, indices[indptr[r]:indptr[r + 1]]] = -np.inf
    order = np.lexsort((np.arange(I)[None, :].repeat(U, 0), -scores), axis=1)[:, :k]
    exp = np.take_along_axis(scores, order, 1)
    assert np.abs(val - exp).max() < 5e-6
    assert np.abs(np.take_along_axis(scores, idx.astype(np.int64), 1) - exp).max() < 5e-6
    assert (idx == order).mean() > 0.999
    assert (np.diff(val, axis=1) <= 0).all()
    # few admissible items: -inf padding survives the re-scoring
    cptr, cidx = random_excl(rs, U, I, 2, 6)
    cand = ops.DeviceCSR(cptr, cidx, I, ctx.device)
    idx, val = (cpu(t) for t in st.recommend(0, U, k, cand=cand))
    for r in range(U):
        n = cptr[r + 1] - cptr[r]
        assert np.isin(idx[r, :n], cidx[cptr[r]:cptr[r + 1]]).all() and np.isinf(val[r, n:]).all() and np.isfinite(val[r, :n]).all()


def test_cml_plugin_end_to_end(ctx, tmp_path):
    from elliot_amd.dataset.samplers import custom_sampler
    from elliot_amd.recommender import CML
    from tests.test_gpu_plugin import make_data
    data, cfg = make_data(tmp_path)
    U, I, F, B, epochs, lr = data.num_users, data.num_items, 16, 512, 2, 0.01
    rs = np.random.RandomState(0)
    w0 = (rs.uniform(-0.05, 0.05, (U, F)).astype(np.float32), rs.uniform(-0.05, 0.05, (I, F)).astype(np.float32),
          rs.uniform(-0.05, 0.05, I).astype(np.float32))
    params = SimpleNamespace(meta=SimpleNamespace(verbose=False), epochs=epochs, batch_size=B, seed=42, factors=F, lr=lr, l_w=0.001,
                             l_b=0.001, margin=0.5)
    model = CML(data=data, config=cfg, params=params, init_weights=w0)
    assert model.name == "CML_seed=42_e=2_bs=512_factors=16_lr=0$01_l_w=0$001_l_b=0$001_margin=0$5"
    model.train()
    orc = oc.CMLOracle(*w0, lr, 0.001, 0.001, 0.5)
    sampler = custom_sampler.Sampler(data.sp_i_train, ctx=ctx)
    losses = []
    for it in range(epochs):
        tot = 0.0
        for u, i, j in sampler.step(data.transactions, B):
            tot += orc.train_step((u.cpu().numpy(), i.cpu().numpy(), j.cpu().numpy()))
        losses.append(tot / (it + 1))
    for got, exp in zip(model._losses, losses):
        assert abs(got - exp) <= 2e-4 * abs(exp), (model._losses, losses)
    assert (np.abs(cpu(model._model.state.Gi) - orc.Gi) > 1e-4).mean() < 1e-2
    res = model.get_results()
    assert set(res.keys()) == {10, 5} and 0.0 <= res[10]["test_results"]["nDCG"] <= 1.0


def test_sorted_gradient_path_matches_oracle(ctx):
    """B >= 2048: row gradients through the sorted segment kernels of the BPR path (coefficients given), hot items included."""
    rs = np.random.RandomState(11)
    U, I, F, B, lr, l_w, l_b, margin = 900, 500, 32, 4096, 0.01, 0.01, 0.02, 0.5
    Gu, Gi, Bi = tables(rs, U, I, F, 0.3)
    st = ops.CmlDeviceState(ctx, Gu, Gi, Bi)
    orc = oc.CMLOracle(Gu, Gi, Bi, lr, l_w, l_b, margin)
    for s in range(3):
        u, j = rs.randint(0, U, B), rs.randint(0, I, B)
        i = np.where(rs.rand(B) < 0.3, 5, rs.randint(0, I, B))          # item 5 owns ~30 % of the positives: chunk-crossing segment
        st.train_step(dev(ctx, u), dev(ctx, i), dev(ctx, j), lr, l_w, l_b, margin)
        got, exp = st.pop_loss(), orc.train_step((u, i, j))
        assert abs(got - exp) <= 1e-4 * abs(exp), (s, got, exp)
        for name, ref in (("Gu", orc.Gu), ("Gi", orc.Gi), ("Bi", orc.Bi)):
            err = np.abs(cpu(getattr(st, name)) - ref)
            assert (err > 5e-5).mean() < 5e-3 and err.max() < 5 * lr, (s, name, float(err.max()), float((err > 5e-5).mean()))
        assert not cpu(st.gGu).any() and not cpu(st.gGi).any() and not cpu(st.gBi).any()


def test_tiny_shapes_fuzz(ctx):
    rs = np.random.RandomState(98)
    for trial in range(25):
        U, I, F, n = rs.randint(1, 5), rs.randint(2, 6), rs.randint(1, 7), rs.randint(1, 9)
        Gu, Gi, Bi = tables(rs, U, I, F, 0.5)
        st = ops.CmlDeviceState(ctx, Gu, Gi, Bi)
        orc = oc.CMLOracle(Gu, Gi, Bi, 0.01, 0.01, 0.02, 0.5)
        u, i, j = rs.randint(0, U, n), rs.randint(0, I, n), rs.randint(0, I, n)
        st.train_step(dev(ctx, u), dev(ctx, i), dev(ctx, j), 0.01, 0.01, 0.02, 0.5)
        got, exp = st.pop_loss(), orc.train_step((u, i, j))
        assert abs(got - exp) <= 1e-4 * max(abs(exp), 1e-3), (trial, U, I, F, n, got, exp)
        for name, ref in (("Gu", orc.Gu), ("Gi", orc.Gi), ("Bi", orc.Bi)):
            assert np.abs(cpu(getattr(st, name)) - ref).max() < 0.05 + 1e-6, (trial, name)     # <= 5 lr: a hinge sitting on the boundary
        k = min(I, 3)
        idx, val = (cpu(t) for t in st.recommend(0, U, k))
        best = -np.sort(-oc.CMLOracle(cpu(st.Gu), cpu(st.Gi), cpu(st.Bi), 0, 0, 0, 0).predict(0, U).astype(np.float64), axis=1)[:, :k]
        assert np.abs(val - best).max() < 1e-5, (trial, U, I, F)


def test_user_sharded_hip_path_equals_concatenated_batch(ctx):
    """parallel.ShardedCml's kernel sequence with two virtual ranks on one GPU: forward -> gather of D, E -> grads against the
    gathered vectors -> sum of the item-side accumulators -> split apply.  Equals the oracle's step on the concatenated batch
    (whose hinge has (2B)^2 terms); item replicas stay bit-identical."""
    from elliot_amd import parallel
    rs = np.random.RandomState(31)
    U, I, F, B, lr, l_w, l_b, margin, G = 400, 300, 32, 2500, 0.01, 0.01, 0.02, 0.5, 2
    Gu, Gi, Bi = tables(rs, U, I, F, 0.3)
    rng = [parallel.user_range(U, r, G) for r in range(G)]
    sts = [ops.CmlDeviceState(ctx, Gu[lo:hi], Gi, Bi) for lo, hi in rng]
    orc = oc.CMLOracle(Gu, Gi, Bi, lr, l_w, l_b, margin)
    for step in range(3):
        batches = [(rs.randint(lo, hi, B), rs.randint(0, 40, B), rs.randint(0, I, B)) for lo, hi in rng]
        dv = [[dev(ctx, u - lo), dev(ctx, i), dev(ctx, j)] for (lo, hi), (u, i, j) in zip(rng, batches)]
        de = [st.forward_de(*t, l_w, l_b) for st, t in zip(sts, dv)]
        D_all, E_all = torch.cat([d for d, _ in de]), torch.cat([e for _, e in de])          # the all-gather
        for st, t, (D, E) in zip(sts, dv, de):
            st.grads_de(*t, l_w, l_b, margin, D, E, D_all, E_all)
        tot = sts[0].item_grad_flat + sts[1].item_grad_flat                                  # the all-reduce
        loss = 0.0
        for st in sts:
            st.item_grad_flat.copy_(tot)
            st.begin_step()
            st.apply_users(lr)
            st.apply_items(lr)
            loss += st.pop_loss()
        cu, ci, cj = (np.concatenate([b[x] for b in batches]) for x in range(3))
        exp = orc.train_step((cu, ci, cj))
        assert abs(loss - exp) <= 1e-4 * abs(exp), (step, loss, exp)
        assert torch.equal(sts[0].Gi, sts[1].Gi) and torch.equal(sts[0].Bi, sts[1].Bi)
        assert (np.abs(cpu(sts[0].Gi) - orc.Gi) > 5e-5).mean() < 5e-3 and (np.abs(cpu(sts[0].Bi) - orc.Bi) > 5e-5).mean() < 5e-3
        for st, (lo, hi) in zip(sts, rng):
            assert (np.abs(cpu(st.Gu) - orc.Gu[lo:hi]) > 5e-5).mean() < 5e-3
            assert not bool(st.gGu.any()) and not bool(st.item_grad_flat.any())
