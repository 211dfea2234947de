"""GPU parity of the fused NeuMF / GMF full-catalogue scoring (el_nmf_score_topk, el_gmf_item_image) against the
C oracle's pinned-order logits (oracle/c/el_oracle.c: orc_nmf_logits) + its masked top-k: index lists AND logit bits, at toy
shapes (every padding / mask / candidate-list edge) and at d = 128 against a 100 000-item catalogue."""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from oracle import cref
from oracle import neumf as on
from tests.gpu_util import assert_topk_equal, cpu, random_excl

pytestmark = pytest.mark.gpu


def _weights(U, I, F, seed, units=None, scale=4.0):
    """GlorotUniform draws are tiny at these sizes (logits ~ 1e-3, everything ties after the sigmoid): scaled up, with biases."""
    w = on.init_neumf(U, I, F, seed, units=units)
    rs = np.random.RandomState(seed + 100)
    for k in ("Umf", "Imf", "Umlp", "Imlp"):
        w[k] = (w[k] * scale).astype(np.float32)
    w["W"] = [(x * 1.5).astype(np.float32) for x in w["W"]]
    w["b"] = [rs.normal(scale=0.05, size=b.shape).astype(np.float32) for b in w["b"]]
    w["hb"] = np.array([0.07], np.float32)
    return w


def _oracle_topk(w, u0, u1, k, excl=None, cand=None, i0=0, i1=None):
    L = cref.nmf_logits(w, np.arange(u0, u1), i0, i1)
    ex = None if excl is None else (excl[0], excl[1])
    ca = None if cand is None else (cand[0], cand[1])
    # orc_topk_rows_f32 indexes CSR rows by ABSOLUTE user id
    return cref.topk_rows_f32(L, u0, k, excl=ex, cand=ca, item_offset=i0)


@pytest.mark.parametrize("F,units,k", [(32, None, 10), (16, None, 7), (8, [40, 24, 12], 5), (64, None, 20), (12, [48, 20, 10], 12),
                                       (9, [36, 18, 9], 10)])
def test_nmf_score_topk_matches_oracle_bit_for_bit(ctx, F, units, k):
    U, I = 70, 2500
    w = _weights(U, I, F, seed=F, units=units)
    st = ops.NmfDeviceState(ctx, w, max_batch=1024)
    assert st.fused_supported(k)
    rs = np.random.RandomState(1)
    ip, ix = random_excl(rs, U, I, 0, 40)
    excl = ops.DeviceCSR(ip, ix, I, ctx.device)
    for u0, u1 in ((0, U), (13, 41)):
        idx, val = st.score_topk_logits(u0, u1, k, excl=excl)
        ei, ev = _oracle_topk(w, u0, u1, k, excl=(ip, ix))
        assert_topk_equal(f"nmf_score_F{F}_k{k}", cpu(idx), cpu(val), ei, ev)
    # no mask at all
    idx, val = st.score_topk_logits(0, 9, k)
    ei, ev = _oracle_topk(w, 0, 9, k)
    assert_topk_equal(f"nmf_score_nomask_F{F}", cpu(idx), cpu(val), ei, ev)


def test_nmf_score_branches_mlp_only_and_no_head_bias(ctx):
    U, I, F, k = 40, 1800, 16, 10
    w = _weights(U, I, F, seed=5)
    mlp_only = {kk: v for kk, v in w.items() if kk not in ("Umf", "Imf")}
    mlp_only["hw"] = w["hw"][F:].copy()
    st = ops.NmfDeviceState(ctx, mlp_only, max_batch=512)
    idx, val = st.score_topk_logits(0, U, k)
    ei, ev = _oracle_topk(mlp_only, 0, U, k)
    assert_topk_equal("nmf_score_mlp_only", cpu(idx), cpu(val), ei, ev)
    nob = {kk: v for kk, v in w.items() if kk != "hb"}
    st = ops.NmfDeviceState(ctx, nob, max_batch=512)
    idx, val = st.score_topk_logits(0, U, k)
    ei, ev = _oracle_topk(nob, 0, U, k)
    assert_topk_equal("nmf_score_no_hb", cpu(idx), cpu(val), ei, ev)


def test_nmf_score_short_lists_candidates_and_item_shards(ctx):
    U, I, F, k = 30, 700, 16, 10
    w = _weights(U, I, F, seed=11)
    st = ops.NmfDeviceState(ctx, w, max_batch=512)
    rs = np.random.RandomState(2)
    # users whose train rows cover almost the whole catalogue: fewer than k unmasked items -> -inf padding with the lowest masked ids
    rows = [np.sort(rs.choice(I, I - rs.randint(0, 2 * k), replace=False)) if u % 3 == 0 else np.sort(rs.choice(I, 5, replace=False))
            for u in range(U)]
    ip = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    ix = np.concatenate(rows).astype(np.int32)
    excl = ops.DeviceCSR(ip, ix, I, ctx.device)
    idx, val = st.score_topk_logits(0, U, k, excl=excl)
    ei, ev = _oracle_topk(w, 0, U, k, excl=(ip, ix))
    assert_topk_equal("nmf_score_short", cpu(idx), cpu(val), ei, ev)
    # candidate protocol (negative sampling: mask = in the candidate row), incl. rows with fewer than k candidates and an empty one
    crow = [np.sort(rs.choice(I, rs.randint(0, 4) if u % 5 == 0 else rs.randint(20, 120), replace=False)) for u in range(U)]
    crow[3] = np.zeros(0, np.int64)
    cp = np.concatenate([[0], np.cumsum([len(r) for r in crow])]).astype(np.int64)
    cx = np.concatenate(crow).astype(np.int32)
    cand = ops.DeviceCSR(cp, cx, I, ctx.device)
    idx, val = st.score_topk_logits(0, U, k, cand=cand)
    ei, ev = _oracle_topk(w, 0, U, k, cand=(cp, cx))
    assert_topk_equal("nmf_score_cand", cpu(idx), cpu(val), ei, ev)
    # item shards (global ids through item_offset) + merge == one shard
    parts_i, parts_v = [], []
    for lo, hi in ((0, 260), (260, I)):
        pi, pv = st.score_topk_logits(0, U, k, excl=excl, item_offset=lo, I_local=hi - lo)
        e2i, e2v = _oracle_topk(w, 0, U, k, excl=(ip, ix), i0=lo, i1=hi)
        assert_topk_equal(f"nmf_score_shard_{lo}", cpu(pi), cpu(pv), e2i, e2v)
        parts_i.append(pi)
        parts_v.append(pv)
    mi, mv = ops.topk_merge(ctx, torch.stack(parts_i), torch.stack(parts_v))
    fi, fv = _oracle_topk(w, 0, U, k, excl=(ip, ix))
    assert_topk_equal("nmf_score_merged", cpu(mi), cpu(mv), fi, fv)


def test_nmf_score_large_k_and_item_image_reuse(ctx):
    U, I, F = 24, 5000, 32
    w = _weights(U, I, F, seed=21)
    st = ops.NmfDeviceState(ctx, w, max_batch=512)
    for k in (100, 300):
        idx, val = st.score_topk_logits(0, U, k)
        ei, ev = _oracle_topk(w, 0, U, k)
        assert_topk_equal(f"nmf_score_k{k}", cpu(idx), cpu(val), ei, ev)
    # EL_TOPK_ITEMS_UNCHANGED is verified on the device: the same call with the flag gives the same lists; after an in-place
    # update of the item MLP table (same address) the kept projection would be stale -- the hash notices and it is rebuilt
    i1, v1 = st.score_topk_logits(0, U, 10)
    i2, v2 = st.score_topk_logits(0, U, 10, items_unchanged=True)
    assert torch.equal(i1, i2) and torch.equal(v1.view(torch.int32), v2.view(torch.int32))
    st.tab[3].mul_(-1.0)
    w2 = dict(w)
    w2["Imlp"] = -w["Imlp"]
    i3, v3 = st.score_topk_logits(0, U, 10, items_unchanged=True)
    ei, ev = _oracle_topk(w2, 0, U, 10)
    assert_topk_equal("nmf_score_after_inplace_update", cpu(i3), cpu(v3), ei, ev)


def test_nmf_score_d128_against_100k_items(ctx):
    """BASELINE configs[3] model (d = 128, tower 512-256-128) against a 100 000-item catalogue: a handful of users bit-exact against
    the oracle's pinned chain (1.6e10 fma per user on the host), 64 more against the device's own pair scoring (el_nmf_forward:
    another summation order, so values to 2e-6 and the sets compared where the k / k+1 gap exceeds that)."""
    U, I, F, k = 80, 100_000, 128, 10
    w = _weights(U, I, F, seed=7, scale=12.0)
    st = ops.NmfDeviceState(ctx, w, max_batch=1 << 20)
    rs = np.random.RandomState(3)
    ip, ix = random_excl(rs, U, I, 5, 60)
    excl = ops.DeviceCSR(ip, ix, I, ctx.device)
    idx, val = st.score_topk_logits(0, U, k + 1, excl=excl)
    n = 3
    ei, ev = _oracle_topk(w, 0, n, k + 1, excl=(ip, ix))
    assert_topk_equal("nmf_score_d128", cpu(idx[:n]), cpu(val[:n]), ei, ev)
    # pair scoring of the listed items: probabilities of el_nmf_forward == sigmoid(fused logits) to fp32 round-off
    users = torch.arange(U, dtype=torch.int32, device=ctx.device).repeat_interleave(k + 1)
    p_pairs = st.forward(users, idx.reshape(-1).contiguous()).reshape(U, k + 1)
    p_fused = torch.sigmoid(val.double()).float()
    assert float((p_pairs - p_fused).abs().max()) < 2e-6
    # ordering, range, exclusions
    v = cpu(val)
    assert (v[:, :-1] >= v[:, 1:]).all()
    got = cpu(idx)
    assert got.min() >= 0 and got.max() < I
    for u in range(U):
        assert not set(got[u].tolist()) & set(ix[ip[u]:ip[u + 1]].tolist())


def test_recommend_links_and_reranks_like_topk_on_probabilities(ctx):
    """NmfDeviceState.recommend = get_recs + get_top_k of the reference: sigmoid of the logits, ties of the PROBABILITY broken by
    item index.  Checked against the device's own probabilities of every pair (el_nmf_forward + el_dense_topk, the reference's
    route) on weights whose logits are large enough to collapse after the sigmoid (saturation: p == 1.0f for many items)."""
    U, I, F, k = 50, 1500, 16, 10
    for scale, wscale in ((4.0, 1.0), (30.0, 6.0)):           # ordinary; saturated (many probabilities round to exactly 1)
        w = _weights(U, I, F, seed=31, scale=scale)
        w["hw"] = (w["hw"] * wscale).astype(np.float32)
        st = ops.NmfDeviceState(ctx, w, max_batch=1 << 17)
        rs = np.random.RandomState(4)
        ip, ix = random_excl(rs, U, I, 0, 30)
        excl = ops.DeviceCSR(ip, ix, I, ctx.device)
        idx, val = st.recommend(0, U, k, excl=excl)
        ri, rv = st._pairs_topk(0, U, k + 1, excl, None)
        # the two routes evaluate the logit in different summation orders: values agree to fp32 round-off everywhere; a row whose
        # LIST differs must contain a near-tie (a probability gap inside that noise) among the pair route's first k + 1 entries
        assert float((val - rv[:, :k]).abs().max()) < 2e-6
        diff_rows = np.nonzero(~cpu((idx == ri[:, :k]).all(1)))[0]
        full = cpu(rv)
        for u in diff_rows:
            assert (np.abs(np.diff(full[u])) < 4e-6).any(), (scale, int(u), full[u].tolist())
        if scale < 10:
            assert len(diff_rows) <= U // 4
        # internal consistency of the fused list: ordered by (probability desc, index asc)
        v, ii = cpu(val), cpu(idx)
        assert (v[:, :-1] >= v[:, 1:]).all()
        tie = v[:, :-1] == v[:, 1:]
        assert (ii[:, :-1][tie] < ii[:, 1:][tie]).all()


def test_gmf_recommend_through_the_dot_product_kernels(ctx):
    """GMF: sigmoid(sum_f h_f u_f i_f) ranked by the fused dot-product top-k kernels on the item image Imf * h: lists equal the C
    oracle's fma chain on (Umf, Imf * h) (bit-exact raw scores), values = sigmoid of those."""
    U, I, F, k = 300, 4000, 64, 10
    w = on.init_gmf(U, I, F, 5)
    w = {kk: (v * 6).astype(np.float32) for kk, v in w.items()}
    st = ops.NmfDeviceState(ctx, w, max_batch=4096)
    rs = np.random.RandomState(6)
    ip, ix = random_excl(rs, U, I, 0, 25)
    excl = ops.DeviceCSR(ip, ix, I, ctx.device)
    idx, val = st.recommend(0, U, k, excl=excl)
    img = (w["Imf"] * w["hw"][None, :]).astype(np.float32)
    ei, ev = cref.score_topk_f32(w["Umf"], img, None, 0, U, k, excl=(ip, ix))
    p = 1.0 / (1.0 + np.exp(-ev.astype(np.float64)))
    distinct = (np.diff(p.astype(np.float32), axis=1) != 0).all(1)           # rows without probability ties: the lists must agree
    assert distinct.mean() > 0.9
    assert np.array_equal(cpu(idx)[distinct], ei[distinct])
    assert np.abs(cpu(val).astype(np.float64) - p).max() < 1e-6
    # and the pair route agrees on the values
    users = torch.arange(U, dtype=torch.int32, device=ctx.device).repeat_interleave(k)
    pp = st.forward(users, idx.reshape(-1).contiguous()).reshape(U, k)
    assert float((pp - val).abs().max()) < 2e-6


def test_unsupported_tower_takes_the_pair_route(ctx):
    U, I, F = 20, 300, 8
    w = on.init_neumf(U, I, F, 3, units=[16, 8])                    # two Dense layers: not the fused kernel's shape
    st = ops.NmfDeviceState(ctx, w, max_batch=8192)
    assert not st.fused_supported(10)
    idx, val = st.recommend(0, U, 5)
    ri, rv = st._pairs_topk(0, U, 5, None, None)
    assert torch.equal(idx, ri) and torch.equal(val, rv)
    with pytest.raises(Exception):
        st.score_topk_logits(0, U, 5)


# ---------------------------------------------------------------------------------------------------- screened route (EL_NMF_SCREEN)
@pytest.mark.parametrize("F,units,k,I", [(128, None, 10, 30000), (64, None, 20, 9000), (32, [128, 64, 32], 10, 6000), (16, [72, 40, 16], 50, 5000),
                                         (128, None, 100, 12000), (24, [300, 128, 40], 10, 7000)])
def test_screened_route_returns_the_unscreened_lists_and_logit_bits(ctx, F, units, k, I, monkeypatch, lib_option):
    """EL_NMF_SCREEN: layers 2-3 on the half-precision matrix instruction with a per-pair error bound, a per-user threshold from the
    lower bounds, the fp32 kernel on the pairs whose upper bound reaches it.  The answer is the fp32 kernel's: index lists and logit
    bits equal the unscreened call's -- with an exclusion CSR, without, on an item shard with its offset, with the item image kept
    from the previous call.  (the option nmf_screen_maxfrac = 1: the candidate route runs whatever share of the pairs survives -- with
    these scaled-up weights and biases the bound keeps most of them, which exercises the candidate regions at every fill.)"""
    lib_option("nmf_screen_maxfrac", 1.0)
    U = 48
    w = _weights(U, I, F, seed=F + k, units=units)
    st = ops.NmfDeviceState(ctx, w, max_batch=1024)
    rs = np.random.RandomState(3)
    ip, ix = random_excl(rs, U, I, 0, 60)
    excl = ops.DeviceCSR(ip, ix, I, ctx.device)
    for kwargs in ({"excl": excl}, {}, {"excl": excl, "item_offset": 1000, "I_local": I - 1500}):
        ref_i, ref_v = st.score_topk_logits(0, U, k, screen=False, **kwargs)
        got_i, got_v = st.score_topk_logits(0, U, k, screen=True, **kwargs)
        pairs, fell_back = st.screen_stats()
        n_items = kwargs.get("I_local", I)
        assert fell_back == (n_items < 4096)                       # (a shard this small is not worth two passes: exact kernel alone)
        assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32)), (kwargs.keys(), int((got_i != ref_i).sum()))
        assert k * U <= pairs <= U * n_items, (pairs, U * n_items)
        again_i, again_v = st.score_topk_logits(0, U, k, screen=True, items_unchanged=True, **kwargs)
        assert torch.equal(again_i, ref_i) and torch.equal(again_v.view(torch.int32), ref_v.view(torch.int32))
    # a sub-range of the users
    ref_i, ref_v = st.score_topk_logits(7, 29, k, excl=excl, screen=False)
    got_i, got_v = st.score_topk_logits(7, 29, k, excl=excl, screen=True)
    assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32))


def test_screened_route_on_glorot_weights_filters_and_matches_the_oracle(ctx):
    """On the reference's initialisation (GlorotUniform everywhere, zero biases: neural_matrix_factorization_model.py:44-70 -- the
    weights of bench.py's neumf leg; as many users as items, so that both embedding tables draw from the same range) the bound leaves
    well under 2 % of 200 000 items per user (DESIGN quotes 0.14 % at 1 M items); lists and logits equal the unscreened call's, and
    the oracle's (orc_nmf_logits + masked top-k) on a sample of users."""
    U, I, F, k, nu = 200_000, 200_000, 128, 10, 16
    w = on.init_neumf(U, I, F, 77)
    st = ops.NmfDeviceState(ctx, w, max_batch=1024)
    rs = np.random.RandomState(5)
    ip, ix = random_excl(rs, nu, I, 0, 200)
    ip = np.concatenate([ip, np.full(U - nu, ip[-1], np.int64)])
    excl = ops.DeviceCSR(ip, ix, I, ctx.device)
    ref_i, ref_v = st.score_topk_logits(0, nu, k, excl=excl, screen=False)
    idx, val = st.score_topk_logits(0, nu, k, excl=excl, screen=True)
    pairs, fell_back = st.screen_stats()
    assert not fell_back and k * nu <= pairs <= 0.02 * nu * I, (pairs, fell_back)
    assert torch.equal(idx, ref_i) and torch.equal(val.view(torch.int32), ref_v.view(torch.int32))
    ei, ev = _oracle_topk(w, 0, 2, k, excl=(ip, ix))
    assert_topk_equal("nmf_screen_d128", cpu(idx[:2]), cpu(val[:2]), ei, ev)


def test_screened_route_falls_back_when_too_many_pairs_survive_or_a_row_is_short(ctx, monkeypatch, lib_option):
    """More surviving pairs than the option nmf_screen_maxfrac of the block, or a user with fewer than k unmasked items, sends the call
    through the unscreened route: same answer, `fell_back` set; the default policy of score_topk_logits then leaves the next calls
    unscreened."""
    U, I, F, k = 12, 8000, 32, 10
    w = _weights(U, I, F, seed=9)
    st = ops.NmfDeviceState(ctx, w, max_batch=512)
    ref_i, ref_v = st.score_topk_logits(0, U, k, screen=False)
    lib_option("nmf_screen_maxfrac", 0.0001)           # 9.6 pairs: fewer than the k * U the lists themselves need
    got_i, got_v = st.score_topk_logits(0, U, k, screen=True)
    pairs, fell_back = st.screen_stats()
    assert fell_back and pairs == U * I                           # the exact kernel scored every pair
    assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32))
    got_i, got_v = st.score_topk_logits(0, U, k)                   # default policy: screens, falls back, ...
    assert st.screen_stats()[1] and st._screen_skip == 15
    assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32))
    got_i, got_v = st.score_topk_logits(0, U, k)                   # ... and does not try again at once
    assert st.screen_stats() == (U * I, False) and st._screen_skip == 14
    assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32))
    lib_option("nmf_screen_maxfrac", 1.0)
    # user 3 keeps only 4 unmasked items
    keep = np.array([5, 77, 4000, 7999])
    rows = [np.zeros(0, np.int32)] * U
    rows[3] = np.setdiff1d(np.arange(I), keep).astype(np.int32)
    ip = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    excl = ops.DeviceCSR(ip, np.concatenate(rows).astype(np.int32), I, ctx.device)
    st3 = ops.NmfDeviceState(ctx, w, max_batch=512)
    ref_i, ref_v = st3.score_topk_logits(0, U, k, excl=excl, screen=False)
    got_i, got_v = st3.score_topk_logits(0, U, k, excl=excl, screen=True)
    assert st3.screen_stats()[1]
    assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32))


def test_screened_route_survives_activations_past_the_half_range(ctx, monkeypatch, lib_option):
    """Weights large enough for layer-1 activations beyond 65504: the half-precision pass overflows (inf, NaN), those pairs carry no
    bound and go to the exact kernel; the lists are still the unscreened call's."""
    lib_option("nmf_screen_maxfrac", 1.0)
    U, I, F, k = 8, 6000, 32, 10
    w = _weights(U, I, F, seed=4)
    w["Umlp"] = (w["Umlp"] * 3e4).astype(np.float32)               # PU_u ~ 1e5 on some units
    st = ops.NmfDeviceState(ctx, w, max_batch=512)
    ref_i, ref_v = st.score_topk_logits(0, U, k, screen=False)
    got_i, got_v = st.score_topk_logits(0, U, k, screen=True)
    assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32))


def test_screened_route_rebuilds_its_item_image_after_an_unscreened_call_rebuilt_the_projection(ctx):
    """The half-precision image of the item projection follows the projection even when the call that rebuilt the projection did
    not screen: screened call (weights v1) -> the item table changes in place -> UNSCREENED call without the unchanged claim
    (rebuilds the projection) -> screened call WITH the claim (projection kept): its bounds must come from the new image, i.e. the
    lists are the fp32 kernel's on the new weights."""
    U, I, F, k, nu = 60_000, 60_000, 64, 10, 32
    w = on.init_neumf(U, I, F, 21)
    st = ops.NmfDeviceState(ctx, w, max_batch=1024)
    st.score_topk_logits(0, nu, k, screen=True)
    assert not st.screen_stats()[1]
    rs = torch.Generator(device=ctx.device)
    rs.manual_seed(4)
    st.tab[3].copy_((torch.rand(st.tab[3].shape, generator=rs, device=ctx.device) * 2 - 1) * 0.02)     # new Imlp, same address
    ref_i, ref_v = st.score_topk_logits(0, nu, k, screen=False)                                          # rebuilds the projection
    got_i, got_v = st.score_topk_logits(0, nu, k, screen=True, items_unchanged=True)
    pairs, fell_back = st.screen_stats()
    assert not fell_back and pairs < 0.2 * nu * I, (pairs, fell_back)
    assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32))


@pytest.mark.parametrize("between", ["nothing", "unscreened_call"])
def test_screened_route_keeps_its_item_image_across_user_blocks_of_other_sizes(ctx, between, monkeypatch, lib_option):
    """(the option nmf_screen_maxfrac = 1: whatever share survives, the screened route is taken.)
    The last, shorter user block of an evaluation reuses the workspace with EL_TOPK_ITEMS_UNCHANGED: the half-precision item image
    and its residual norms must be found where the full block built them (they sit in front of every region sized by the user range,
    k or the split), and an unscreened call in between -- whose user-side regions lie over the image -- forces their rebuild.  Lists
    and logit bits equal the unscreened call's either way, for a shorter block, another k, and a longer block again."""
    lib_option("nmf_screen_maxfrac", 1.0)
    U, I, F = 300, 40_000, 64
    w = on.init_neumf(U, I, F, 31)
    st = ops.NmfDeviceState(ctx, w, max_batch=1024)
    rs = np.random.RandomState(8)
    ip, ix = random_excl(rs, U, I, 0, 40)
    excl = ops.DeviceCSR(ip, ix, I, ctx.device)
    ref = ops.NmfDeviceState(ctx, w, max_batch=1024)                    # its own workspace: never screened
    st.score_topk_logits(0, 256, 10, excl=excl, screen=True)            # the full block: builds the projection and its image
    assert not st.screen_stats()[1]
    ws_ptr = st._score_ws.data_ptr()
    for (a, b, k) in ((256, 300, 10), (256, 263, 50), (10, 11, 10), (0, 256, 10), (40, 296, 20)):
        if between == "unscreened_call":
            st.score_topk_logits(a, b, k, excl=excl, screen=False, items_unchanged=True)
        got_i, got_v = st.score_topk_logits(a, b, k, excl=excl, screen=True, items_unchanged=True)
        pairs, fell_back = st.screen_stats()
        assert st._score_ws.data_ptr() == ws_ptr or k > 10            # (same workspace unless a larger k outgrew it)
        ws_ptr = st._score_ws.data_ptr()
        ref_i, ref_v = ref.score_topk_logits(a, b, k, excl=excl, screen=False)
        # the bounds come from the image: a fresh state that builds its own image for this very call must be left with the same pairs
        fresh = ops.NmfDeviceState(ctx, w, max_batch=1024)
        fresh.score_topk_logits(a, b, k, excl=excl, screen=True)
        assert (pairs, fell_back) == fresh.screen_stats(), (a, b, k, pairs, fell_back, fresh.screen_stats())
        assert torch.equal(got_i, ref_i) and torch.equal(got_v.view(torch.int32), ref_v.view(torch.int32)), (a, b, k, between)
