"""GPU parity: point-wise factor models (el_pwmf_*: MF, FunkSVD, PMF, LogisticMF) against oracle/pointwise_mf.py."""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from oracle import pointwise_mf as pw
from tests.gpu_util import cpu, random_excl

pytestmark = pytest.mark.gpu

MODELS = {  # name -> (kind, biases, optimizer)
    "MF": ("mse", False, "adam"), "FunkSVD": ("mse", True, "adam"), "PMF": ("mse_sigmoid", False, "adam"),
    "LogisticMF": ("logistic", True, "adagrad"),
}


def weights(rs, U, I, F, bias, scale=0.3):
    w = {"Gu": rs.normal(scale=scale, size=(U, F)).astype(np.float32), "Gi": rs.normal(scale=scale, size=(I, F)).astype(np.float32)}
    if bias:
        w["Bu"] = rs.normal(scale=0.1, size=U).astype(np.float32)
        w["Bi"] = rs.normal(scale=0.1, size=I).astype(np.float32)
    return w


def make(ctx, w, model, alpha=0.5, l_w=0.02, lr=0.01):
    kind, _, opt = MODELS[model]
    st = ops.PwmfDeviceState(ctx, w["Gu"], w["Gi"], w.get("Bu"), w.get("Bi"), kind=kind, optimizer=opt, alpha=alpha, l_w=l_w)
    return st, pw.PointwiseOracle(w, kind, lr, optimizer=opt, alpha=alpha, l_w=l_w)


def dev(ctx, a, dtype):
    return torch.from_numpy(np.ascontiguousarray(a, dtype=dtype)).to(ctx.device)


def step_both(ctx, st, orc, u, i, y, lr, side="both"):
    st.train_step(dev(ctx, u, np.int32), dev(ctx, i, np.int32), dev(ctx, y, np.float32), lr, side=side)
    return st.pop_loss(), orc.train_step((u, i, y), side=side)


def assert_state(st, orc, lr, tag):
    got = st.weights()
    for k, v in orc.w.items():
        err = np.abs(got[k].reshape(v.shape) - v)
        assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (tag, k, float(err.max()), float((err > 2e-5).mean()))
    for g in ("gGu", "gGi", "gBu", "gBi"):                           # accumulators are handed back clean
        t = getattr(st, g)
        assert t is None or float(t.abs().max()) == 0.0, (tag, g)


@pytest.mark.parametrize("model", list(MODELS))
@pytest.mark.parametrize("F", [8, 10, 64, 200])
def test_train_steps_match_oracle(ctx, model, F):
    rs = np.random.RandomState(F + len(model))
    U, I, lr = 300, 180, 0.01
    w = weights(rs, U, I, F, MODELS[model][1])
    st, orc = make(ctx, w, model, lr=lr)
    sides = ("items", "users") if model == "LogisticMF" else ("both",)
    for s in range(6):
        n = 900 if s != 3 else 37
        u = rs.randint(0, U, n)
        i = np.minimum(rs.zipf(1.3, n) - 1, I - 1)                   # a few very hot items: long segments
        y = rs.randint(0, 2, n).astype(np.float32)
        got, exp = step_both(ctx, st, orc, u, i, y, lr, side=sides[s % len(sides)])
        assert abs(got - exp) <= 1e-4 * max(abs(exp), 1e-3), (model, F, s, got, exp)
        assert_state(st, orc, lr, (model, F, s))


def test_segments_across_chunks_large_batch(ctx):
    """B large enough for multi-position chunks on both sides; one item owns a third of the batch (atomic flushes)."""
    rs = np.random.RandomState(1)
    U, I, F, n, lr = 5000, 800, 32, 300000, 0.005
    for model in ("FunkSVD", "LogisticMF"):
        st, orc = make(ctx, weights(rs, U, I, F, True, scale=0.05), model, lr=lr, l_w=0.001)
        u = rs.randint(0, U, n)
        i = np.where(rs.rand(n) < 0.33, 7, rs.randint(0, I, n))
        y = rs.randint(0, 2, n).astype(np.float32)
        got, exp = step_both(ctx, st, orc, u, i, y, lr)
        assert abs(got - exp) <= 2e-4 * max(abs(exp), 1e-3), (model, got, exp)
        gw = st.weights()
        for k, v in orc.w.items():
            err = np.abs(gw[k].reshape(v.shape) - v)
            assert err.max() < 5 * lr and (err > 1e-4).mean() < 1e-3, (model, k, float(err.max()))


def test_run_to_run_reproducibility(ctx):
    """No floating-point atomics in the segment sums (round 6): a row whose segment lies in one chunk is summed in sorted (= batch) order
    with plain stores, the pieces of a row cut by chunk boundaries go to partial slots and are added in a fixed order
    (el_segcombine.h: k_seg_combine for a few pieces, k_seg_combine_long for the hot items of a Zipf catalogue).  Eight steps twice from
    the same state: every table and bias bit-identical, for a model with the L2 term inside the segments (LogisticMF) and one without."""
    rs = np.random.RandomState(4)
    U, I, F, n = 200000, 1500, 16, 50000
    w = weights(rs, U, I, F, True)
    steps = [(rs.randint(0, U, n), np.minimum(rs.zipf(1.2, n) - 1, I - 1), rs.randint(0, 2, n).astype(np.float32)) for _ in range(8)]
    assert max(np.bincount(s[1]).max() for s in steps) > 5000            # the hottest item is cut into hundreds of pieces
    for model in ("FunkSVD", "LogisticMF"):
        outs = []
        for _ in range(2):
            st, _ = make(ctx, w, model, l_w=0.001)
            for u, i, y in steps:
                st.train_step(dev(ctx, u, np.int32), dev(ctx, i, np.int32), dev(ctx, y, np.float32), 0.01)
            outs.append(st.weights())
        for k in outs[0]:
            assert np.array_equal(outs[0][k], outs[1][k]), (model, k, int((outs[0][k] != outs[1][k]).sum()))


@pytest.mark.parametrize("model", list(MODELS))
def test_forward_pairs(ctx, model):
    rs = np.random.RandomState(2)
    U, I, F = 150, 170, 24
    w = weights(rs, U, I, F, MODELS[model][1])
    st, orc = make(ctx, w, model)
    u, i = rs.randint(0, U, 4000), rs.randint(0, I, 4000)
    got = cpu(st.forward(dev(ctx, u, np.int32), dev(ctx, i, np.int32)))
    exp = pw.predict(orc.w, orc.kind, u, i, dtype=np.float64)
    assert np.abs(got - exp).max() < 2e-6


def expected_topk(scores, indptr, indices, k):
    s = scores.copy()
    for r in range(s.shape[0]):
        s[r, indices[indptr[r]:indptr[r + 1]]] = -np.inf
    order = np.lexsort((np.arange(s.shape[1])[None, :].repeat(s.shape[0], 0), -s), axis=1)[:, :k]
    return order, np.take_along_axis(s, order, 1)


@pytest.mark.parametrize("model", list(MODELS))
def test_recommend_matches_oracle_scores(ctx, model):
    rs = np.random.RandomState(6)
    U, I, F, k = 400, 900, 32, 10
    w = weights(rs, U, I, F, MODELS[model][1])
    st, orc = make(ctx, w, model)
    indptr, indices = random_excl(rs, U, I, 0, 30)
    excl = ops.DeviceCSR(indptr, indices, I, ctx.device)
    idx, val = (cpu(t) for t in st.recommend(0, U, k, excl=excl))
    scores = orc.predict_all(0, U).astype(np.float64)
    eidx, eval_ = expected_topk(scores, indptr, indices, k)
    assert np.abs(val - eval_).max() < 2e-6                          # same score profile ...
    picked = np.take_along_axis(scores, idx.astype(np.int64), 1)
    assert np.abs(picked - eval_).max() < 2e-6                       # ... reached by items that really score that much
    assert (idx == eidx).mean() > 0.999                              # and the same items except fp32 near-ties
    for r in range(U):
        assert not np.isin(idx[r], indices[indptr[r]:indptr[r + 1]]).any()


def full_list(ctx, st, U, I, excl):
    """Every item of every user through the same kernels (k = I), linked: the device's own score table."""
    idx, val = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, 0, U, I, excl=excl)
    st._link(val, I, 0)
    return cpu(idx), cpu(val)


@pytest.mark.parametrize("model,scale", [("PMF", 0.3), ("PMF", 3.0), ("PMF", 40.0), ("FunkSVD", 0.3), ("FunkSVD", 1e-4)])
def test_recommend_tie_rule_after_the_link(ctx, model, scale):
    """Where the link collapses distinct raw scores into one float the order must be index-ascending and the cut must
    not lose a lower index (tf.nn.top_k on the linked scores).  scale 3: sigmoid saturates to 1.0 for many items (long
    collapsed runs -> the list is regrown); scale 40: everything is 0.0 or 1.0 (dense path); tiny scale + biases ~0.1:
    the bias swallows the low bits of the dot product."""
    rs = np.random.RandomState(8)
    dense = scale >= 40
    U, I, F, k = 64, (5000 if dense else 1200), 16, 10               # the list can grow to 4032 entries, then dense
    w = weights(rs, U, I, F, MODELS[model][1], scale=scale)
    st, orc = make(ctx, w, model)
    indptr, indices = random_excl(rs, U, I, 0, 20)
    excl = ops.DeviceCSR(indptr, indices, I, ctx.device)
    idx, val = (cpu(t) for t in st.recommend(0, U, k, excl=excl))
    if not dense:
        fidx, fval = full_list(ctx, st, U, I, excl)
        order = np.lexsort((fidx, -fval.astype(np.float64)), axis=1)[:, :k]
        eidx, eval_ = np.take_along_axis(fidx, order, 1), np.take_along_axis(fval, order, 1)
        assert np.array_equal(idx, eidx) and np.array_equal(val, eval_)
    else:                                                            # GEMM summation order: values to 1e-6, same tie rule
        _, eval_ = expected_topk(orc.predict_all(0, U).astype(np.float64), indptr, indices, k)
        assert np.abs(val - eval_).max() < 1e-6
        for r in range(U):
            same = val[r, :-1] == val[r, 1:]
            assert (idx[r, :-1][same] < idx[r, 1:][same]).all()
            assert not np.isin(idx[r], indices[indptr[r]:indptr[r + 1]]).any()
    assert (np.diff(val, axis=1) <= 0).all()


def test_recommend_candidate_protocol(ctx):
    rs = np.random.RandomState(10)
    U, I, F, k = 50, 300, 8, 5
    st, orc = make(ctx, weights(rs, U, I, F, True), "FunkSVD")
    cptr, cidx = random_excl(rs, U, I, 3, 40)
    cand = ops.DeviceCSR(cptr, cidx, I, ctx.device)
    idx, val = (cpu(t) for t in st.recommend(0, U, k, cand=cand))
    scores = orc.predict_all(0, U)
    for r in range(U):
        c = cidx[cptr[r]:cptr[r + 1]]
        n = min(k, len(c))
        assert np.isin(idx[r, :n], c).all()
        best = np.sort(scores[r, c])[::-1][:n]
        assert np.abs(val[r, :n] - best).max() < 2e-6


def test_rejects_bad_arguments(ctx):
    rs = np.random.RandomState(0)
    w = weights(rs, 10, 10, 4, False)
    st, _ = make(ctx, w, "MF")
    u = dev(ctx, np.zeros(4), np.int32)
    y = dev(ctx, np.zeros(4), np.float32)
    with pytest.raises(Exception, match="workspace|null|step"):
        ops.check(ctx.lib.el_pwmf_train_step(ctx.handle, ctx.stream(), ops.C.byref(st._c), u.data_ptr(), u.data_ptr(), y.data_ptr(),
                                             4, 0, 0, 1, 0.01, st.loss.data_ptr(), None, 0), "el_pwmf_train_step")
    with pytest.raises(ValueError):
        ops.PwmfDeviceState(ctx, w["Gu"], w["Gi"], Bu=np.zeros(10, np.float32))


def test_tiny_shapes_fuzz(ctx):
    """Degenerate sizes (one user, two items, one factor, one sample, F not a multiple of 4): every kernel path still agrees with
    the oracle."""
    rs = np.random.RandomState(99)
    for trial in range(40):
        U, I, F, n = rs.randint(1, 5), rs.randint(2, 6), rs.randint(1, 7), rs.randint(1, 9)
        model = list(MODELS)[trial % 4]
        w = weights(rs, U, I, F, MODELS[model][1])
        st, orc = make(ctx, w, model, lr=0.01)
        for s in range(2):
            u, i = rs.randint(0, U, n), rs.randint(0, I, n)
            y = rs.randint(0, 2, n).astype(np.float32)
            side = ("items", "users")[s] if model == "LogisticMF" else "both"
            got, exp = step_both(ctx, st, orc, u, i, y, 0.01, side=side)
            assert abs(got - exp) <= 1e-4 * max(abs(exp), 1e-3), (trial, model, U, I, F, n, got, exp)
        assert_state(st, orc, 0.01, (trial, model, U, I, F, n))
        k = min(I, 3)
        idx, val = (cpu(t) for t in st.recommend(0, U, k))
        scores = orc.predict_all(0, U).astype(np.float64)
        best = -np.sort(-scores, axis=1)[:, :k]
        assert np.abs(val - best).max() < 1e-5, (trial, model, U, I, F)


class _PairSum:
    """Two virtual ranks on one GPU: `all_reduce_sum` adds the partner's tensor (the k-th call of rank 0 pairs with the k-th of rank 1)."""

    def __init__(self):
        self.world, self.always, self.pending = 2, False, []

    def all_reduce_sum(self, t, async_op=False):
        self.pending.append(t)
        return None if async_op else t


@pytest.mark.parametrize("model", ["FunkSVD", "PMF", "LogisticMF"])
def test_user_sharded_hip_path_equals_concatenated_batch(ctx, model):
    """parallel.ShardedPwmf with two virtual ranks: local user rows + item replicas on el_pwmf_grads / el_pwmf_apply, the
    all-reduce emulated by adding the two ranks' accumulators between grads and the item-side apply."""
    from elliot_amd import parallel
    rs = np.random.RandomState(21)
    U, I, F, n, lr, G = 501, 260, 32, 3000, 0.01, 2
    w = weights(rs, U, I, F, MODELS[model][1])
    kind, _, opt = MODELS[model]
    rng = [parallel.user_range(U, r, G) for r in range(G)]
    sts = [ops.PwmfDeviceState(ctx, w["Gu"][lo:hi], w["Gi"], None if "Bu" not in w else w["Bu"][lo:hi], w.get("Bi"), kind=kind,
                               optimizer=opt, alpha=0.5, l_w=0.02) for lo, hi in rng]
    orc = pw.PointwiseOracle(w, kind, lr, optimizer=opt, alpha=0.5, l_w=0.02)
    sides = ("items", "users") if kind == "logistic" else ("both",)
    for step in range(4):
        side = sides[step % len(sides)]
        batches = [(rs.randint(lo, hi, n), np.minimum(rs.zipf(1.4, n) - 1, I - 1), rs.randint(0, 2, n).astype(np.float32)) for lo, hi in rng]
        for st, (lo, hi), (u, i, y) in zip(sts, rng, batches):
            st.grads(dev(ctx, u - lo, np.int32), dev(ctx, i, np.int32), dev(ctx, y, np.float32), n_global=G * n, side=side)
        if side != "users":
            for a, b in zip(sts[0].item_grads(), sts[1].item_grads()):            # the all-reduce
                tot = a + b
                a.copy_(tot)
                b.copy_(tot)
        loss = 0.0
        for st in sts:
            if side != "items":
                st.apply(lr, side="users", advance=True)
            if side != "users":
                st.apply(lr, side="items", advance=(side == "items"))
            loss += st.pop_loss()
        cu, ci, cy = (np.concatenate([b[x] for b in batches]) for x in range(3))
        exp = orc.train_step((cu, ci, cy), side=side)
        assert abs(loss - exp) <= 1e-4 * max(abs(exp), 1e-3), (model, step, loss, exp)
        assert torch.equal(sts[0].Gi, sts[1].Gi) and (sts[0].Bi is None or torch.equal(sts[0].Bi, sts[1].Bi))   # replicas stay identical
        assert (np.abs(cpu(sts[0].Gi) - orc.w["Gi"]) > 2e-5).mean() < 2e-3
        for st, (lo, hi) in zip(sts, rng):
            assert (np.abs(cpu(st.Gu) - orc.w["Gu"][lo:hi]) > 2e-5).mean() < 2e-3
            for g in ("gGu", "gGi", "gBu", "gBi"):
                t = getattr(st, g)
                assert t is None or not bool(t.any()), (model, step, g)
        assert sts[0].step == sts[1].step == step + 1


@pytest.mark.parametrize("model,B", [("FunkSVD", 512), ("PMF", 4096), ("LogisticMF", 700)])
def test_train_loop_equals_per_batch_calls(ctx, model, B):
    """el_pwmf_train_loop = pointwise sampler + train_step per batch (same Philox offsets, same kernels, short last batch)."""
    from elliot_amd.synthetic import zipf_csr
    rs = np.random.RandomState(12)
    U, I, F = 2500, 700, 16
    indptr, indices = zipf_csr(U, I, mean_log=2.0, sigma_log=0.6, dmin=1, dmax=60, seed=3)
    pos = ops.DeviceCSR(indptr, indices, I, ctx.device)
    w = weights(rs, U, I, F, MODELS[model][1])
    a, _ = make(ctx, w, model)
    b, _ = make(ctx, w, model)
    events, lr = 4 * B + B // 3, 0.01
    sides = ("items", "users") if model == "LogisticMF" else ("both",)
    first = 500
    for side in sides:
        for start in range(0, events, B):
            n = min(B, events - start)
            u, i, y = ops.pointwise_sample(ctx, pos, n, seed=9, first_sample=first + start)
            a.train_step(u, i, y, lr, side=side)
        assert b.train_loop(pos, events, B, 9, first, lr, side=side) == 5
        first += events
    assert a.step == b.step
    la, lb = a.pop_loss(), b.pop_loss()
    assert abs(la - lb) <= 1e-8 * abs(la), (la, lb)
    wa, wb = a.weights(), b.weights()
    for k in ("Gu", "Gi") + (("Bu", "Bi") if MODELS[model][1] else ()):
        assert np.abs(wa[k] - wb[k]).max() < 1e-6, k


@pytest.mark.parametrize("kk", [2, 12, 64, 65, 200, 1000, 4032])
def test_topk_rerank_any_length_equals_a_stable_two_key_sort(ctx, kk):
    """el_topk_rerank: (value desc, index asc) for lists of up to 4096 entries -- ties on purpose (values drawn from a few
    levels), rows of different content."""
    rs = np.random.RandomState(kk)
    n = 37
    idx = np.stack([rs.permutation(10 * kk + 50)[:kk] for _ in range(n)]).astype(np.int32)
    val = rs.choice(np.array([-1.5, 0.0, 0.25, 0.25000003, 3.0], np.float32), size=(n, kk)).astype(np.float32)
    d = ctx.device
    oi, ov = ops.topk_rerank(ctx, torch.from_numpy(idx.copy()).to(d), torch.from_numpy(val.copy()).to(d))
    for r in range(n):
        order = np.lexsort((idx[r], -val[r].astype(np.float64)))
        assert np.array_equal(cpu(oi[r]), idx[r][order]) and np.array_equal(cpu(ov[r]), val[r][order]), (kk, r)
