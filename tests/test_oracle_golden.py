"""The CPU oracle against fixtures produced by the reference's own code (oracle/gen_golden.py)."""
import numpy as np

from oracle import cref, sampler as osampler, sgd as osgd, topk as otopk


def test_ref_sampler_replays_reference_stream(golden):
    g = golden("sampler_ref.npz")
    lp, li = g["lists_indptr"], g["lists_items"]
    lists = [li[lp[u]:lp[u + 1]].tolist() for u in range(int(g["n_users"]))]
    s = osampler.RefSampler(lists, int(g["n_items"]), seed=42)
    n = g["u"].shape[0]
    bu, bi, bj = [], [], []
    for b in s.step(n, 512):
        assert b[0].shape[1] == 1
        bu.append(b[0])
        bi.append(b[1])
        bj.append(b[2])
    assert np.array_equal(np.concatenate(bu).reshape(-1), g["u"])
    assert np.array_equal(np.concatenate(bi).reshape(-1), g["i"])
    assert np.array_equal(np.concatenate(bj).reshape(-1), g["j"])


def test_sampler_invariants(golden):
    g = golden("sampler_ref.npz")
    ip, ix = g["indptr"], g["indices"]
    for u, i, j in zip(g["u"][:2000], g["i"][:2000], g["j"][:2000]):
        row = ix[ip[u]:ip[u + 1]]
        assert i in row and j not in row


def test_sgd_oracle_matches_reference_trace(golden):
    g = golden("bprmf_sgd_trace.npz")
    P, Q, b = g["P0"].copy(), g["Q0"].copy(), g["b0"].copy()
    hp = {k: float(g[k]) for k in ("lr", "reg_bias", "reg_user", "reg_pos", "reg_neg")}
    osgd.train_sequential(P, Q, b, g["u"], g["i"], g["j"], **hp)
    assert np.abs(P - g["P1"]).max() < 1e-13
    assert np.abs(Q - g["Q1"]).max() < 1e-13
    assert np.abs(b - g["b1"]).max() < 1e-13


def test_sgd_init_matches_reference(golden):
    g = golden("bprmf_sgd_trace.npz")
    P, Q, b = osgd.initialize(g["P0"].shape[0], g["Q0"].shape[0], g["P0"].shape[1], 42)
    assert np.array_equal(P, g["P0"]) and np.array_equal(Q, g["Q0"]) and np.array_equal(b, g["b0"])


def test_f64_topk_oracle_matches_reference_get_user_predictions(golden):
    t = golden("bprmf_sgd_trace.npz")
    s = golden("sampler_ref.npz")
    k = golden("bprmf_sgd_topk.npz")
    kk = int(k["k"])
    excl = (s["indptr"], s["indices"])
    for r, u in enumerate(k["users"]):
        oi, ov = cref.score_topk_f64(t["P1"], t["Q1"], t["b1"], int(u), int(u) + 1, kk, excl=excl)
        assert np.array_equal(oi[0], k["idx"][r]), (u, oi[0], k["idx"][r])
        assert np.allclose(ov[0], k["val"][r], rtol=0, atol=1e-12)


def test_c_topk_equals_numpy_topk_restatement():
    rs = np.random.RandomState(3)
    U, I, F, k = 37, 211, 24, 10
    Gu = rs.normal(size=(U, F)).astype(np.float32)
    Gi = rs.normal(size=(I, F)).astype(np.float32)
    Gi[50] = Gi[20]  # exact ties -> lower index first
    Gi[51] = Gi[20]
    Bi = rs.normal(size=I).astype(np.float32)
    Bi[50] = Bi[51] = Bi[20]
    indptr = np.arange(0, 5 * U + 1, 5, dtype=np.int64)
    indices = np.concatenate([np.sort(rs.choice(I, 5, replace=False)) for _ in range(U)]).astype(np.int32)
    scores = cref.scores_f32(Gu, Gi, Bi, 0, U)
    # fp32 fma chain vs fp64 maths
    ref64 = Bi.astype(np.float64) + Gu.astype(np.float64) @ Gi.astype(np.float64).T
    assert np.abs(scores - ref64).max() < 1e-4
    mask = otopk.dense_mask_from_excl(indptr, indices, 0, U, I)
    v, idx = otopk.get_top_k(scores, mask, k)
    oi, ov = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=(indptr, indices))
    assert np.array_equal(oi, idx) and np.array_equal(ov, v)
    # candidate protocol
    cptr = np.arange(0, 12 * U + 1, 12, dtype=np.int64)
    cidx = np.concatenate([np.sort(rs.choice(I, 12, replace=False)) for _ in range(U)]).astype(np.int32)
    cmask = otopk.dense_mask_from_cand(cptr, cidx, 0, U, I)
    v, idx = otopk.get_top_k(scores, cmask, k)
    oi, ov = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, cand=(cptr, cidx))
    assert np.array_equal(oi, idx) and np.array_equal(ov, v)
    # fewer than k candidates: -inf padding with the lowest masked indices
    v, idx = otopk.get_top_k(scores, cmask, 20)
    oi, ov = cref.score_topk_f32(Gu, Gi, Bi, 0, U, 20, cand=(cptr, cidx))
    assert np.array_equal(oi, idx) and np.array_equal(ov, v)
    assert np.isneginf(ov[:, 12:]).all()


def test_c_topk_shards_merge_to_global():
    rs = np.random.RandomState(5)
    U, I, F, k = 9, 300, 16, 7
    Gu = rs.normal(size=(U, F)).astype(np.float32)
    Gi = rs.normal(size=(I, F)).astype(np.float32)
    full_i, full_v = cref.score_topk_f32(Gu, Gi, None, 0, U, k)
    parts = []
    for lo, hi in ((0, 100), (100, 230), (230, 300)):
        parts.append(cref.score_topk_f32(Gu, Gi[lo:hi], None, 0, U, k, item_offset=lo))
    ci = np.concatenate([p[0] for p in parts], axis=1)
    cv = np.concatenate([p[1] for p in parts], axis=1)
    for u in range(U):
        order = np.lexsort((ci[u], -cv[u]))[:k]
        assert np.array_equal(ci[u][order], full_i[u]) and np.array_equal(cv[u][order], full_v[u])


def _small_itd():
    from elliot_amd.synthetic import small_dataset
    return small_dataset(200, 150, seed=0)           # the data set oracle/gen_golden.py ran the reference on


def test_pointwise_sampler_stream_equals_the_reference(golden):
    """oracle restatement AND the product's host replay (`sampler: replay` of the point-wise plugins) against the stream the
    reference's own pointwise_pos_neg_sampler.Sampler.step produced (interleaved np.random / random draws, both seeded 42)."""
    from elliot_amd.dataset.samplers.pointwise_pos_neg_sampler import replay_stream
    g = golden("pointwise_sampler_ref.npz")
    indptr, indices, itd = _small_itd()
    n = g["u"].shape[0]
    lists = [list(set(itd[u])) for u in itd]
    ora = osampler.RefPointwiseSampler(lists, int(indices.max()) + 1)
    parts = list(ora.step(n, 512))
    for k, name in enumerate(("u", "i", "b")):
        assert np.array_equal(np.concatenate([p[k] for p in parts]), g[name].astype(np.int64)), name
    # the product: two calls continue one stream (epoch after epoch), like consecutive Sampler.step batches
    u1, i1, b1, st = replay_stream(itd, 2500)
    u2, i2, b2, _ = replay_stream(itd, n - 2500, st)
    assert np.array_equal(np.concatenate([u1, u2]), g["u"]) and np.array_equal(np.concatenate([i1, i2]), g["i"])
    assert np.array_equal(np.concatenate([b1, b2]), g["b"])
    rows = [set(indices[indptr[u]:indptr[u + 1]].tolist()) for u in range(len(itd))]
    assert all((int(i) in rows[int(u)]) == bool(b) for u, i, b in zip(g["u"], g["i"], g["b"]))


def test_neumf_epoch_sampler_equals_the_reference(golden):
    """elliot_amd's NeuMF epoch sampler (host bookkeeping of neural/NeuMF/custom_sampler.py:27-48: positives + m negatives per
    positive, set-deduplicated, random.sample shuffle) reproduces the reference's epoch sample for sample, m = 0 and m = 2."""
    from elliot_amd.recommender.neural.NeuMF.custom_sampler import Sampler
    g = golden("neumf_sampler_ref.npz")
    _, _, itd = _small_itd()
    for m in (0, 2):
        s = Sampler(itd, m)._epoch_python()                     # (constructor seeds np.random / random with 42, like the reference)
        assert np.array_equal(s[:, 0], g[f"u_m{m}"]) and np.array_equal(s[:, 1], g[f"i_m{m}"]) and np.array_equal(s[:, 2], g[f"b_m{m}"]), m
