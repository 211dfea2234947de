"""GPU parity: fused scoring + masked top-k (el_score_topk / el_topk_merge / el_dense_topk / f64)
against the C oracle on the same seeded inputs.  Bit-exact: index lists AND score bits."""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from oracle import cref
from tests.gpu_util import assert_topk_equal, cpu, random_excl

pytestmark = pytest.mark.gpu


def make(rs, U, I, F, bias=True, ties=True, scale=1.0):
    Gu = (rs.normal(size=(U, F)) * scale).astype(np.float32)
    Gi = (rs.normal(size=(I, F)) * scale).astype(np.float32)
    Bi = rs.normal(size=I).astype(np.float32) if bias else None
    if ties and I > 60:
        for dst, src in ((50, 20), (51, 20), (I - 1, 3)):   # exact duplicate items -> exact score ties
            Gi[dst] = Gi[src]
            if Bi is not None:
                Bi[dst] = Bi[src]
    return Gu, Gi, Bi


def run_gpu(ctx, Gu, Gi, Bi, u0, u1, k, excl=None, cand=None, item_offset=0, algo="auto"):
    d = ctx.device
    tGu = torch.from_numpy(Gu).to(d)
    tGi = torch.from_numpy(Gi).to(d)
    tBi = None if Bi is None else torch.from_numpy(Bi).to(d)
    n_items_global = Gi.shape[0] + item_offset
    e = None if excl is None else ops.DeviceCSR(excl[0], excl[1], n_items_global, d)
    c = None if cand is None else ops.DeviceCSR(cand[0], cand[1], n_items_global, d)
    idx, val = ops.score_topk(ctx, tGu, tGi, tBi, u0, u1, k, excl=e, cand=c, item_offset=item_offset, algo=algo)
    torch.cuda.synchronize()
    return cpu(idx), cpu(val)


@pytest.mark.parametrize("algo", ["simple", "mfma", "screen"])
@pytest.mark.parametrize("F,k", [(64, 10), (128, 10), (10, 10), (128, 1), (32, 14), (200, 10), (256, 10), (64, 32), (128, 40), (12, 5)])
def test_topk_matches_oracle_bitexact(ctx, algo, F, k):
    if algo == "screen" and (F > 256 or k > 128):
        pytest.skip("screened kernel: F <= 256, k <= 128")
    rs = np.random.RandomState(100 + F + k)
    U, I = 300, 1000 + F        # ragged last user block (300 = 2*128 + 44) and ragged last item tile
    Gu, Gi, Bi = make(rs, U, I, F)
    excl = random_excl(rs, U, I, 0, 30)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo=algo)
    assert_topk_equal(f"topk_{algo}_F{F}_k{k}", gi, gv, ei, ev)


@pytest.mark.parametrize("algo", ["simple", "mfma", "screen"])
def test_topk_no_bias_no_mask_and_user_subrange(ctx, algo):
    rs = np.random.RandomState(7)
    U, I, F, k = 500, 777, 64, 10
    Gu, Gi, _ = make(rs, U, I, F, bias=False)
    ei, ev = cref.score_topk_f32(Gu, Gi, None, 130, 401, k)
    gi, gv = run_gpu(ctx, Gu, Gi, None, 130, 401, k, algo=algo)
    assert_topk_equal(f"topk_{algo}_nobias_subrange", gi, gv, ei, ev)


@pytest.mark.parametrize("algo", ["simple", "mfma"])
def test_topk_large_k_needs_wave_kernel_or_errors(ctx, algo):
    rs = np.random.RandomState(8)
    U, I, F, k = 70, 900, 48, 100
    Gu, Gi, Bi = make(rs, U, I, F)
    excl = random_excl(rs, U, I, 0, 20)
    if algo == "mfma":
        with pytest.raises(Exception):
            run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="mfma")
        return
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="auto")
    assert_topk_equal("topk_k100", gi, gv, ei, ev)


@pytest.mark.parametrize("algo", ["simple", "mfma", "screen"])
def test_topk_heavy_exclusions_trained_like(ctx, algo):
    """Excluded (train) items get the HIGHEST scores, as after training: they must never surface."""
    rs = np.random.RandomState(9)
    U, I, F, k = 260, 2000, 64, 10
    Gu, Gi, Bi = make(rs, U, I, F, ties=False)
    excl = random_excl(rs, U, I, 20, 120)
    ip, ix = excl
    for u in range(U):                       # make each user's train items score very high
        cols = ix[ip[u]:ip[u + 1]]
        Gi[cols[:3]] += 0.5 * Gu[u]
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo=algo)
    assert_topk_equal(f"topk_{algo}_heavy_excl", gi, gv, ei, ev)
    for u in range(U):
        assert not set(gi[u]) & set(ix[ip[u]:ip[u + 1]])


@pytest.mark.parametrize("algo", ["simple", "mfma", "screen"])
def test_topk_fewer_than_k_candidates_pads_with_neg_inf(ctx, algo):
    rs = np.random.RandomState(10)
    U, I, F, k = 40, 64, 16, 10
    Gu, Gi, Bi = make(rs, U, I, F, ties=False)
    rows = [np.sort(rs.choice(I, I - rs.randint(0, 6), replace=False)) for _ in range(U)]  # 0..5 unmasked
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=(indptr, indices))
    assert np.isneginf(ev).any()
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=(indptr, indices), algo=algo)
    assert_topk_equal(f"topk_{algo}_pad", gi, gv, ei, ev)


def test_topk_candidate_protocol(ctx):
    """negative-sampling protocol: mask = candidate rows (recommender_utils_mixin.py:102-107)."""
    rs = np.random.RandomState(11)
    U, I, F = 90, 700, 64
    Gu, Gi, Bi = make(rs, U, I, F)
    cand = random_excl(rs, U, I, 5, 120)
    for k in (10, 50):
        ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, cand=cand)
        gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, cand=cand)
        assert_topk_equal(f"topk_cand_k{k}", gi, gv, ei, ev)


@pytest.mark.parametrize("algo", ["simple", "mfma", "screen"])
def test_item_shards_merge_equals_single_shard(ctx, algo):
    rs = np.random.RandomState(12)
    U, I, F, k = 200, 1500, 64, 10
    Gu, Gi, Bi = make(rs, U, I, F)
    excl = random_excl(rs, U, I, 0, 40)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    bounds = [0, 400, 401, 1100, 1500]
    pi, pv = [], []
    for lo, hi in zip(bounds[:-1], bounds[1:]):
        gi, gv = run_gpu(ctx, Gu, np.ascontiguousarray(Gi[lo:hi]), np.ascontiguousarray(Bi[lo:hi]), 0, U, k,
                         excl=excl, item_offset=lo, algo=algo)
        oi, ov = cref.score_topk_f32(Gu, Gi[lo:hi], Bi[lo:hi], 0, U, k, excl=excl, item_offset=lo)
        assert_topk_equal(f"topk_{algo}_shard_{lo}", gi, gv, oi, ov)
        pi.append(gi)
        pv.append(gv)
    d = ctx.device
    mi, mv = ops.topk_merge(ctx, torch.from_numpy(np.stack(pi)).to(d), torch.from_numpy(np.stack(pv)).to(d))
    torch.cuda.synchronize()
    assert_topk_equal(f"topk_{algo}_merged", cpu(mi), cpu(mv), ei, ev)


def test_empty_user_range_and_arg_errors(ctx):
    rs = np.random.RandomState(13)
    Gu, Gi, Bi = make(rs, 10, 100, 8)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 5, 5, 3)
    assert gi.shape == (0, 3)
    with pytest.raises(Exception):
        run_gpu(ctx, Gu, Gi, Bi, 0, 10, 0)


def test_dense_topk(ctx):
    rs = np.random.RandomState(14)
    U, I, k = 150, 1234, 20
    preds = rs.normal(size=(U, I)).astype(np.float32)
    preds[:, 10] = preds[:, 5]
    excl = random_excl(rs, U + 7, I, 0, 50)
    ei, ev = cref.topk_rows_f32(preds, 7, k, excl=excl)      # rows are users 7..7+U
    d = ctx.device
    e = ops.DeviceCSR(excl[0], excl[1], I, d)
    gi, gv = ops.dense_topk(ctx, torch.from_numpy(preds).to(d), 7, 7 + U, k, excl=e)
    torch.cuda.synchronize()
    assert_topk_equal("dense_topk", cpu(gi), cpu(gv), ei, ev)


def test_topk_f64_matches_oracle_and_reference_fixture(ctx, golden):
    t, s, kf = golden("bprmf_sgd_trace.npz"), golden("sampler_ref.npz"), golden("bprmf_sgd_topk.npz")
    d = ctx.device
    P, Q, b = (torch.from_numpy(t[n]).to(d) for n in ("P1", "Q1", "b1"))
    U, I = P.shape[0], Q.shape[0]
    excl = ops.DeviceCSR(s["indptr"], s["indices"], I, d)
    k = int(kf["k"])
    gi, gv = ops.score_topk_f64(ctx, P, Q, b, 0, U, k, excl=excl)
    torch.cuda.synchronize()
    ei, ev = cref.score_topk_f64(t["P1"], t["Q1"], t["b1"], 0, U, k, excl=(s["indptr"], s["indices"]))
    assert_topk_equal("topk_f64", cpu(gi), cpu(gv), ei, ev)
    # and the reference's own MFModel.get_user_predictions outputs (BPRMF_model.py:70-85)
    for r, u in enumerate(kf["users"]):
        assert np.array_equal(cpu(gi)[u], kf["idx"][r])
        assert np.allclose(cpu(gv)[u], kf["val"][r], rtol=0, atol=1e-12)


def test_scores_within_fp32_roundoff_of_fp64_matmul(ctx):
    """The pinned fma chain vs the mathematically exact score: |err| <= ~1e-6 * sum|a*b| (MFMA f32)."""
    rs = np.random.RandomState(15)
    U, I, F, k = 128, 512, 128, 10
    Gu, Gi, Bi = make(rs, U, I, F, ties=False)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, algo="mfma")
    ref = Bi.astype(np.float64) + Gu.astype(np.float64) @ Gi.astype(np.float64).T
    exact = np.take_along_axis(ref, gi.astype(np.int64), axis=1)
    assert np.abs(gv - exact).max() < 5e-5


def test_screened_topk_pathological_ties_fall_back_exactly(ctx):
    """All items identical for half of the users (every score ties -> the 2E window holds the whole catalogue): the
    overflow flag must route those users through the exact wave kernel; results stay bit-identical to the oracle."""
    rs = np.random.RandomState(21)
    U, I, F, k = 600, 900, 64, 10
    Gu, Gi, Bi = make(rs, U, I, F, ties=False)
    Gi[100:800] = Gi[100]                    # 700 identical items
    Bi[100:800] = Bi[100]
    Gu[::2] *= 40.0                          # large norms -> wide windows as well
    excl = random_excl(rs, U, I, 0, 30)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="screen")
    assert_topk_equal("topk_screen_ties", gi, gv, ei, ev)


def test_screened_topk_large_random_block(ctx):
    rs = np.random.RandomState(22)
    U, I, F, k = 1100, 5000, 128, 10
    Gu = rs.uniform(-0.01, 0.01, size=(U, F)).astype(np.float32)
    Gi = rs.uniform(-0.03, 0.03, size=(I, F)).astype(np.float32)
    Gi[rs.randint(0, I, 50)] *= 6.0          # a few large-norm (popular) items
    Bi = rs.normal(scale=0.001, size=I).astype(np.float32)
    excl = random_excl(rs, U, I, 5, 60)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="screen")
    assert_topk_equal("topk_screen_large", gi, gv, ei, ev)


@pytest.mark.parametrize("case", ["bf16_exact", "near_power_of_two", "one_huge_item", "mixed_scales", "sparse_rows", "tiny", "tiny_users",
                                  "tiny_rows_among_normal"])
def test_screened_topk_residual_bound_adversarial_inputs(ctx, case):
    """The screening bound is ||du|| max||i~|| + ||u|| max||di|| (+ fp32 accumulation), du / di the MEASURED bf16 residuals.  Inputs
    that push on it: operands that ARE bf16 numbers (residuals 0: the window collapses to the accumulation term, and thousands of
    exact ties appear), values just above a power of two (the largest relative bf16 error), one item with a norm 1000x the
    others, per-row scales over 6 orders of magnitude, rows with a handful of non-zeros, values near the fp32 denormal range.
    Indices and score bits must still be the oracle's."""
    rs = np.random.RandomState(sum(map(ord, case)))
    U, I, F, k = 700, 9000, 128, 10
    Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
    Bi = rs.normal(scale=0.01, size=I).astype(np.float32)

    def to_bf16(x):
        b = x.view(np.uint32).astype(np.uint64)
        return (((b + 0x7fff + ((b >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)
    if case == "bf16_exact":
        Gu, Gi = to_bf16(Gu), to_bf16(np.round(Gi * 8) / 8)            # coarse item grid: many exactly equal scores
    elif case == "near_power_of_two":
        Gu = (np.sign(Gu) * (1.0 + 2.0 ** -9 + rs.uniform(0, 2.0 ** -12, size=Gu.shape))).astype(np.float32) * 0.125
        Gi = (np.sign(Gi) * (1.0 + 2.0 ** -9 + rs.uniform(0, 2.0 ** -12, size=Gi.shape))).astype(np.float32) * 0.25
    elif case == "one_huge_item":
        Gi[1234] *= 1000.0
        Gi[77] *= 300.0
    elif case == "mixed_scales":
        Gu *= (10.0 ** rs.uniform(-3, 3, size=(U, 1))).astype(np.float32)
        Gi *= (10.0 ** rs.uniform(-2, 1, size=(I, 1))).astype(np.float32)
    elif case == "sparse_rows":
        Gu *= (rs.uniform(size=Gu.shape) < 0.05)
        Gi *= (rs.uniform(size=Gi.shape) < 0.1)
    elif case == "tiny":
        Gu *= np.float32(1e-18)
        Gi *= np.float32(1e-18)
        Bi *= np.float32(1e-30)
    elif case == "tiny_users":
        # |u_f| ~ 1e-22: v * v underflows to zero in fp32 -- a residual norm summed from plain squares reads 0 and the bound collapses
        # below the true bf16 error (advisor, round 2); the norms are taken relative to the row maximum instead
        Gu *= np.float32(1e-21)
        Bi[:] = 0
    elif case == "tiny_rows_among_normal":
        Gu[::3] *= np.float32(1e-21)
        Gi[::5] *= np.float32(1e-21)
        Bi[:] = 0
    excl = random_excl(rs, U, I, 0, 50)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="screen")
    assert_topk_equal("topk_screen_" + case, gi, gv, ei, ev)


@pytest.mark.parametrize("k", [10, 50])
def test_screened_topk_all_users_fall_back(ctx, k):
    """Every item identical -> every score of a user ties -> every user is flagged: exercises all fallback tiers (dense
    scores for the first 64 users, item-split kernels for the next 512, the plain list kernel for the rest; fp32 MFMA
    kernels for k <= 40, wave kernels above)."""
    rs = np.random.RandomState(23)
    U, I, F = 700, 3000, 32
    Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
    Gi = np.repeat(rs.normal(scale=0.1, size=(1, F)).astype(np.float32), I, axis=0)
    Bi = np.zeros(I, np.float32)
    excl = random_excl(rs, U, I, 0, 40)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="screen")
    assert_topk_equal("topk_screen_all_fallback", gi, gv, ei, ev)


def test_screened_topk_nonfinite_inputs_fall_back(ctx):
    rs = np.random.RandomState(24)
    U, I, F, k = 130, 1000, 64, 10
    Gu, Gi, Bi = make(rs, U, I, F, ties=False)
    Gi[17, 3] = np.nan
    Gi[400, 0] = np.inf
    excl = random_excl(rs, U, I, 0, 20)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="screen")
    assert_topk_equal("topk_screen_nonfinite", gi, gv, ei, ev)


@pytest.mark.parametrize("F,k,I", [(64, 100, 20000), (128, 50, 9000), (32, 128, 6000), (128, 20, 3000)])
def test_screened_topk_large_k_guess_and_verify(ctx, F, k, I):
    """k > 12: the threshold is a guess from a strided pass 1, verified per user by k_screen_final (exact fallback when
    the guess was too high).  Results must still be bit-identical."""
    rs = np.random.RandomState(F + k)
    U = 400
    Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
    Bi = rs.normal(scale=0.01, size=I).astype(np.float32)
    excl = random_excl(rs, U, I, 0, 200)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
    gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo="screen")
    assert_topk_equal(f"topk_screen_k{k}", gi, gv, ei, ev)


def test_topk_fuzz_all_algorithms_against_oracle(ctx):
    """Random shapes and degenerate corners (tiny catalogues, k >= #unmasked, rows masked completely, F = 1, huge or tiny
    scores, constant columns): every eligible algorithm must return the oracle's lists bit for bit."""
    rs = np.random.RandomState(2024)
    corner = [dict(U=3, I=1, F=1, k=1), dict(U=70, I=5, F=3, k=10), dict(U=65, I=64, F=16, k=64), dict(U=33, I=63, F=7, k=5),
              dict(U=129, I=65, F=130, k=12), dict(U=200, I=12288, F=24, k=9), dict(U=90, I=12287, F=40, k=13)]
    cases = corner + [dict(U=int(rs.randint(1, 400)), I=int(rs.choice([rs.randint(1, 300), rs.randint(300, 6000)])),
                           F=int(rs.choice([1, 2, 5, 8, 31, 32, 33, 64, 65, 100, 128, 129, 200, 256, 300])),
                           k=int(rs.choice([1, 2, 5, 10, 11, 20, 37, 64, 100, 128]))) for _ in range(28)]
    for n, c in enumerate(cases):
        U, I, F, k = c["U"], c["I"], c["F"], c["k"]
        scale = float(rs.choice([1e-3, 1.0, 50.0]))
        Gu = (rs.normal(size=(U, F)) * scale).astype(np.float32)
        Gi = (rs.normal(size=(I, F)) * scale).astype(np.float32)
        if n % 3 == 0 and I > 4:
            Gi[rs.randint(0, I, size=max(1, I // 3))] = Gi[0]            # many exact ties
        Bi = None if n % 4 == 0 else (rs.normal(size=I) * scale).astype(np.float32)
        hi = int(rs.choice([0, 3, min(I, 40), I]))
        excl = random_excl(rs, U, I, 0, hi)
        if n % 5 == 0 and U > 2:                                          # a user with everything masked
            rows = [excl[1][excl[0][u]:excl[0][u + 1]] for u in range(U)]
            rows[1] = np.arange(I, dtype=np.int32)
            ip = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
            excl = (ip, np.concatenate(rows).astype(np.int32))
        ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=excl)
        algos = ["simple"]
        if F <= 256 and k <= 40:
            algos.append("mfma")
        if F <= 256 and k <= 128:
            algos += ["screen", "auto"]
        for algo in algos:
            gi, gv = run_gpu(ctx, Gu, Gi, Bi, 0, U, k, excl=excl, algo=algo)
            assert_topk_equal(f"fuzz_{n}_{algo}_U{U}_I{I}_F{F}_k{k}", gi, gv, ei, ev)


def test_items_unchanged_flag_reuses_only_a_matching_image(ctx):
    """EL_TOPK_ITEMS_UNCHANGED keeps the item-side bf16 image of the previous call; results are identical, and a claim that
    does not match the previous call (other table, other shape) is ignored."""
    rs = np.random.RandomState(3)
    U, I, F, k = 700, 2500, 64, 10
    Gu = torch.from_numpy(rs.normal(size=(U, F)).astype(np.float32)).to(ctx.device)
    Gi = torch.from_numpy(rs.normal(size=(I, F)).astype(np.float32)).to(ctx.device)
    Bi = torch.from_numpy(rs.normal(size=I).astype(np.float32)).to(ctx.device)
    a = ops.score_topk(ctx, Gu, Gi, Bi, 0, 350, k, algo="screen")
    b = ops.score_topk(ctx, Gu, Gi, Bi, 350, 700, k, algo="screen", items_unchanged=True)
    full = ops.score_topk(ctx, Gu, Gi, Bi, 0, 700, k, algo="mfma")
    assert torch.equal(torch.cat([a[0], b[0]]), full[0]) and torch.equal(torch.cat([a[1], b[1]]), full[1])
    Gi2 = Gi * 1.5 + 0.1                                             # another table: the (false) claim must not be believed
    c = ops.score_topk(ctx, Gu, Gi2, Bi, 0, 350, k, algo="screen", items_unchanged=True)
    d = ops.score_topk(ctx, Gu, Gi2, Bi, 0, 350, k, algo="mfma")
    assert torch.equal(c[0], d[0]) and torch.equal(c[1], d[1])
    e = ops.score_topk(ctx, Gu, Gi2[:2000].contiguous(), Bi[:2000].contiguous(), 0, 350, k, algo="screen", items_unchanged=True)
    f = ops.score_topk(ctx, Gu, Gi2[:2000].contiguous(), Bi[:2000].contiguous(), 0, 350, k, algo="mfma")
    assert torch.equal(e[0], f[0]) and torch.equal(e[1], f[1])


def test_items_unchanged_claim_is_verified_after_in_place_updates(ctx):
    """The C-ABI hazard: Gi / Bi are updated IN PLACE (same pointers, same shapes -- what an optimiser does) and the caller still
    passes EL_TOPK_ITEMS_UNCHANGED.  The library hashes the tables on the device and rebuilds the image: lists and score bits
    equal a fresh exact evaluation after every mutation, whether the whole table, a single element or only a bias changed, and
    the flag keeps working (bit-identical results) while nothing changes."""
    rs = np.random.RandomState(4)
    U, I, F, k = 600, 3000, 128, 10
    Gu = torch.from_numpy(rs.normal(size=(U, F)).astype(np.float32)).to(ctx.device)
    Gi = torch.from_numpy(rs.normal(size=(I, F)).astype(np.float32)).to(ctx.device)
    Bi = torch.from_numpy(rs.normal(size=I).astype(np.float32)).to(ctx.device)
    ptr = (Gi.data_ptr(), Bi.data_ptr())

    def check(what):
        got = ops.score_topk(ctx, Gu, Gi, Bi, 0, U, k, algo="screen", items_unchanged=True)        # the claim, true or not
        exp = ops.score_topk(ctx, Gu, Gi, Bi, 0, U, k, algo="mfma")
        assert torch.equal(got[0], exp[0]) and torch.equal(got[1].view(torch.int32), exp[1].view(torch.int32)), what
        assert (Gi.data_ptr(), Bi.data_ptr()) == ptr

    ops.score_topk(ctx, Gu, Gi, Bi, 0, U, k, algo="screen")
    check("unchanged")
    Gi.mul_(-0.7).add_(0.05)                                   # an optimiser step: every element moves, in place
    check("whole table rewritten in place")
    check("unchanged again")
    best = int(ops.score_topk(ctx, Gu, Gi, Bi, 0, 1, 1, algo="mfma")[0][0, 0])
    Gi[best, 5] -= 40.0                                         # ONE element: user 0's best item drops out
    check("one element")
    Bi[(best + 1) % I] += 300.0                                 # only a bias: that item now tops every list
    check("one bias")
    top = ops.score_topk(ctx, Gu, Gi, Bi, 0, U, k, algo="screen", items_unchanged=True)[0][:, 0]
    assert bool((top == (best + 1) % I).all())


def test_fragile_user_report_matches_numpy(ctx):
    """el_topk_fragile: users whose rank-k / k+1 gap is inside F 2^-23 |u| max|i| -- counted exactly like a NumPy
    evaluation of the same bound on the oracle's k+1 lists; a planted exact tie is always fragile."""
    rs = np.random.RandomState(11)
    U, I, F, k = 700, 900, 64, 10
    Gu = rs.normal(size=(U, F)).astype(np.float32)
    Gi = rs.normal(size=(I, F)).astype(np.float32) * 0.01      # small item norms: many gaps inside the bound
    Bi = np.zeros(I, np.float32)
    Gi[5] = Gi[17]                                             # exact tie for every user
    indptr, indices = random_excl(rs, U, I, 0, 8)
    pos = ops.DeviceCSR(indptr, indices, I, ctx.device)
    d = ctx.device
    rep, flags = ops.fragile_users(ctx, torch.from_numpy(Gu).to(d), torch.from_numpy(Gi).to(d), torch.from_numpy(Bi).to(d), 0, U, k,
                                   excl=pos, flags=True)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k + 1, excl=(indptr, indices))
    nu = np.sqrt((Gu.astype(np.float64) ** 2).sum(1))
    ni = np.sqrt((Gi.astype(np.float64) ** 2).sum(1))
    bound = F * 2.0 ** -23 * nu * np.maximum(ni[ei[:, k - 1]], ni[ei[:, k]])
    exp = (ev[:, k - 1].astype(np.float64) - ev[:, k].astype(np.float64)) < bound
    got = cpu(flags).astype(bool)
    assert np.array_equal(got, exp)
    assert rep["fragile"] == int(exp.sum()) and rep["users"] == U and rep["short_lists"] == 0
    tie_rows = np.where((ei[:, k - 1] == 5) & (ei[:, k] == 17))[0]
    assert all(got[r] for r in tie_rows)
    assert 0 < rep["fragile"] < U
