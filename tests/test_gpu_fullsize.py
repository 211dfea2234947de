"""Full BASELINE size (configs[1]: 1 M users x 100 K items, d = 128, ~78 M interactions, B = 1 M triplets) through
size-independent properties -- the oracle cannot be run at this size, the properties can:

  top-k     two independent kernels (bf16-screened / fp32 MFMA) agree bit for bit on a whole 131 072-user block; the
            lists are ordered (score desc, item asc), exclude every train item, and their scores are the exact fp32
            fma chain (C oracle on a sample of users); item shards + merge == one shard; the call is idempotent
  sampler   every triplet is valid (i in pos(u), j not in pos(u)), users ~ uniform
  training  the batch loss equals an independent fp64 evaluation of BPRMF_batch_model.py:65-75 on the same triplets
            (1e-4, the north_star tolerance); at Adam step 1 an untouched row does not move and a touched entry moves
            by at most lr (|m/(sqrt(v)+eps)| <= 1 with m = (1-b1) g, v = (1-b2) g^2)
  metrics   sums over two different block partitions of the users agree (checksum of checksums)
"""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr_device
from oracle import cref
from tests.gpu_util import cpu

pytestmark = pytest.mark.gpu

U, I, F, B, K, UB = 1_000_000, 100_000, 128, 1 << 20, 10, 131072


@pytest.fixture(scope="module")
def world(ctx):
    dev = ctx.device
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=1234)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * (6.0 / (U + F)) ** 0.5
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * (6.0 / (I + F)) ** 0.5
    Bi = (torch.rand(I, generator=g, device=dev) - 0.5) * 0.01
    st = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer="adam_tf_dense")
    return {"pos": pos, "st": st}


def _row_members(pos, users, items):
    """items[r, c] in row users[r] of the CSR?  (device, vectorised binary search)"""
    lo = pos.indptr[users.long()][:, None].expand_as(items).clone()
    hi = pos.indptr[users.long() + 1][:, None].expand_as(items).clone()
    it = items.to(torch.int32)
    for _ in range(12):                               # rows have <= 2000 entries
        mid = (lo + hi) // 2
        v = pos.indices[mid.clamp(max=pos.indices.numel() - 1)]
        go = (v < it) & (lo < hi)
        lo = torch.where(go, mid + 1, lo)
        hi = torch.where(go | (lo >= hi), hi, mid)
    found = (lo < pos.indptr[users.long() + 1][:, None]) & (pos.indices[lo.clamp(max=pos.indices.numel() - 1)] == it)
    return found


def test_fullsize_sampler_and_train_step_properties(ctx, world):
    pos, st = world["pos"], world["st"]
    u, i, j = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=0)
    assert int(u.min()) >= 0 and int(u.max()) < U and int(j.min()) >= 0 and int(j.max()) < I
    assert bool(_row_members(pos, u, i[:, None]).all()), "a positive is not a train item of its user"
    assert not bool(_row_members(pos, u, j[:, None]).any()), "a negative is a train item of its user"
    cnt = torch.bincount(u.long(), minlength=U).double()
    assert abs(float(cnt.mean()) - B / U) < 1e-9 and float(cnt.max()) < 25         # uniform over users (custom_sampler.py:32)

    lr, l_w, l_b = 0.001, 0.1, 0.001
    Gu0, Gi0, Bi0 = st.Gu.clone(), st.Gi.clone(), st.Bi.clone()
    st.train_step(u, i, j, lr, l_w, l_b)
    loss = st.pop_loss()
    gu, gi, gj = Gu0[u.long()].double(), Gi0[i.long()].double(), Gi0[j.long()].double()
    bi, bj = Bi0[i.long()].double(), Bi0[j.long()].double()
    d = (bi + (gu * gi).sum(1)) - (bj + (gu * gj).sum(1))
    ref = (torch.nn.functional.softplus(-d.clamp(-80.0, 1e8)).sum()
           + l_w * 0.5 * ((gu * gu).sum() + (gi * gi).sum() + (gj * gj).sum())
           + l_b * 0.5 * (bi * bi).sum() + (l_b / 10) * 0.5 * (bj * bj).sum())
    assert abs(loss - float(ref)) <= 1e-4 * abs(float(ref)), (loss, float(ref))
    # Adam step 1
    touched_u = torch.zeros(U, dtype=torch.bool, device=ctx.device)
    touched_u[u.long()] = True
    du = (st.Gu - Gu0).abs()
    assert float(du[~touched_u].max()) == 0.0
    assert float(du.max()) <= lr * (1 + 1e-3)
    assert float(du[touched_u].max()) > 0.5 * lr
    touched_i = torch.zeros(I, dtype=torch.bool, device=ctx.device)
    touched_i[i.long()] = True
    touched_i[j.long()] = True
    di = (st.Gi - Gi0).abs()
    assert float(di.max()) <= lr * (1 + 1e-3)
    if bool((~touched_i).any()):
        assert float(di[~touched_i].max()) == 0.0
    assert (st.gGu is None or not bool(st.gGu.any())) and not bool(st.gGi.any())   # accumulators zero on exit
    world["trained"] = True


def _oracle_rows(Gu0, Gi0, Bi0, u, i, j, su, si, l_w, l_b):
    """Oracle gradients (oracle/bprmf_batch.py, fp32 and fp64) of the sampled user rows `su` / item rows `si` (device int64
    tensors): every triplet of the batch that touches a sampled row is pulled out and re-indexed into small tables that hold
    just the rows those triplets reference -- for the sampled rows the sub-batch gradient IS the full-batch gradient."""
    from oracle import bprmf_batch as ob
    dev = u.device
    mu = torch.zeros(U, dtype=torch.bool, device=dev)
    mu[su] = True
    mi = torch.zeros(I, dtype=torch.bool, device=dev)
    mi[si] = True
    sel = torch.nonzero(mu[u.long()] | mi[i.long()] | mi[j.long()]).flatten()
    uu, ii, jj = u[sel].long(), i[sel].long(), j[sel].long()
    users = torch.unique(torch.cat([uu, su]))
    items = torch.unique(torch.cat([ii, jj, si]))
    ru = torch.full((U,), -1, dtype=torch.int64, device=dev)
    ru[users] = torch.arange(users.numel(), device=dev)
    ri = torch.full((I,), -1, dtype=torch.int64, device=dev)
    ri[items] = torch.arange(items.numel(), device=dev)
    gu_s, gi_s, bi_s = cpu(Gu0[users]), cpu(Gi0[items]), cpu(Bi0[items])
    a = (cpu(ru[uu]), cpu(ri[ii]), cpu(ri[jj]))
    g32 = ob.gradients(gu_s, gi_s, bi_s, *a, l_w, l_b)                      # (dBi, dGu, dGi)
    g64 = ob.gradients(gu_s, gi_s, bi_s, *a, l_w, l_b, dtype=np.float64)
    pu, pi = cpu(ru[su]), cpu(ri[si])
    return {"n": int(sel.numel()),
            "gGu": (g32[1][pu], g64[1][pu]), "gGi": (g32[2][pi], g64[2][pi]), "gBi": (g32[0][pi], g64[0][pi])}


def test_fullsize_gradients_and_weights_on_sampled_rows(ctx):
    """configs[1] training parity beyond the loss: three steps at B = 2^20 on the 1M x 100K x 128 tables; at every step the
    PRE-optimiser gradients gGu / gGi / gBi of 2048 random user rows, 2048 random item rows and the 8 HOTTEST item rows
    (tens of thousands of occurrences each: chunk-crossing segments, the 32-bit U + item sort keys, hot-row combines)
    against oracle/bprmf_batch.py evaluated on exactly the triplets that touch those rows; after the three steps the
    weights of the same rows against the oracle's Keras-Adam recurrence fed with the oracle's gradients.
    Tolerances: gradients 2e-5 of the tensor's largest sampled entry or 4x the oracle's own fp32-vs-fp64 distance (fp32
    summation order is the only freedom); weights: at most 2e-4 of the entries off by more than 2e-5 (Adam's m/(sqrt(v)+eps)
    flips where a sum cancels to ~0), none by more than 3 lr."""
    from oracle import bprmf_batch as ob
    dev = ctx.device
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=1234)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
    g = torch.Generator(device=dev)
    g.manual_seed(43)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.05
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.05
    Bi = (torch.rand(I, generator=g, device=dev) - 0.5) * 0.02
    st = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer="adam_tf_dense")
    assert st.compact                                        # what bench.py runs at this size
    del Gu, Gi, Bi
    lr, l_w, l_b = 0.001, 0.1, 0.001
    t0 = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=0)
    hot = torch.argsort(torch.bincount(torch.cat([t0[1], t0[2]]).long(), minlength=I), descending=True)[:8]
    su = torch.randperm(U, generator=g, device=dev)[:2048]
    si = torch.unique(torch.cat([torch.randperm(I, generator=g, device=dev)[:2048], hot]))
    th = {"Gu": cpu(st.Gu[su]), "Gi": cpu(st.Gi[si]), "Bi": cpu(st.Bi[si])}
    m = {k: np.zeros_like(x) for k, x in th.items()}
    v = {k: np.zeros_like(x) for k, x in th.items()}
    for step in range(3):
        u, i, j = t0 if step == 0 else ops.bpr_sample(ctx, pos, B, seed=42, first_sample=step * B)
        exp = _oracle_rows(st.Gu, st.Gi, st.Bi, u, i, j, su, si, l_w, l_b)
        assert exp["n"] > 100_000                            # the hot rows really are hot
        st.grads(u, i, j, l_w, l_b)
        got = {"gGu": cpu(st.user_grad_dense()[su]), "gGi": cpu(st.gGi[si]), "gBi": cpu(st.gBi[si])}
        for name in ("gGu", "gGi", "gBi"):
            e32, e64 = exp[name]
            scale = float(np.abs(e64).max())
            err = float(np.abs(got[name] - e64).max())
            ref_err = float(np.abs(e32.astype(np.float64) - e64).max())
            assert err <= max(2e-5 * scale, 4 * ref_err), (step, name, err, ref_err, scale)
        st.apply(lr)
        for name, gname in (("Gu", "gGu"), ("Gi", "gGi"), ("Bi", "gBi")):
            ob.adam_tf_sparse_apply(th[name], m[name], v[name], exp[gname][0].astype(np.float32), lr, step + 1)
    st.pop_loss()
    for name, rows in (("Gu", su), ("Gi", si), ("Bi", si)):
        gotw = cpu(getattr(st, name)[rows])
        err = np.abs(gotw - th[name])
        assert float((err > 2e-5).mean()) <= 2e-4 and float(err.max()) < 3 * lr, (name, float(err.max()), float((err > 2e-5).mean()))


def test_fullsize_topk_properties(ctx, world):
    pos, st = world["pos"], world["st"]
    s0 = 3 * UB
    i_scr, v_scr = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="screen")
    i_mf, v_mf = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="mfma")
    torch.cuda.synchronize()
    assert torch.equal(i_scr, i_mf), "screened and fp32 MFMA kernels disagree on the index lists"
    assert torch.equal(v_scr.view(torch.int32), v_mf.view(torch.int32)), "screened and fp32 MFMA kernels disagree on the score bits"
    # ordering (score desc, item asc), range, exclusions
    assert bool((v_scr[:, :-1] >= v_scr[:, 1:]).all())
    tie = v_scr[:, :-1] == v_scr[:, 1:]
    assert bool((i_scr[:, :-1][tie] < i_scr[:, 1:][tie]).all())
    assert int(i_scr.min()) >= 0 and int(i_scr.max()) < I
    users = torch.arange(s0, s0 + UB, device=ctx.device, dtype=torch.int32)
    assert not bool(_row_members(pos, users, i_scr).any()), "a train item was recommended"
    # scores are the exact chain: C oracle on a sample of users (full catalogue, the oracle's own top-k)
    sample = np.arange(s0, s0 + 96)
    ip = cpu(pos.indptr[s0:s0 + 97])
    ix = cpu(pos.indices[int(ip[0]):int(ip[-1])])
    ei, ev = cref.score_topk_f32(cpu(st.Gu[s0:s0 + 96]), cpu(st.Gi), cpu(st.Bi), 0, 96, K, excl=(ip - ip[0], ix))
    assert np.array_equal(cpu(i_scr[:96]), ei) and np.array_equal(cpu(v_scr[:96]), ev), sample[:3]
    # idempotence
    i2, v2 = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="screen")
    assert torch.equal(i2, i_scr) and torch.equal(v2.view(torch.int32), v_scr.view(torch.int32))
    # item shards + merge == one shard (a quarter of the block is enough)
    n = UB // 4
    parts_i, parts_v = [], []
    for lo, hi in ((0, 37_000), (37_000, I)):
        pi, pv = ops.score_topk(ctx, st.Gu, st.Gi[lo:hi].contiguous(), st.Bi[lo:hi].contiguous(), s0, s0 + n, K, excl=pos,
                                item_offset=lo, algo="auto")
        parts_i.append(pi)
        parts_v.append(pv)
    mi, mv = ops.topk_merge(ctx, torch.stack(parts_i), torch.stack(parts_v))
    assert torch.equal(mi, i_scr[:n]) and torch.equal(mv.view(torch.int32), v_scr[:n].view(torch.int32))
    world["idx"] = i_scr
    world["s0"] = s0


def test_c4_catalogue_topk_screened_vs_fp32_and_oracle(ctx):
    """north_star's target catalogue (BASELINE configs[3]/C4: 1 M items, d = 128): a 16 384-user block, screened kernel == fp32
    MFMA kernel on indices and score bits, == the C oracle on a sample of users; k = 10 and k = 100."""
    dev = ctx.device
    U4, I4, UB4 = 65536, 1_000_000, 16384
    indptr, indices = zipf_csr_device(U4, I4, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=99)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I4)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    Gu = (torch.rand((U4, F), generator=g, device=dev) * 2 - 1) * (6.0 / (U4 + F)) ** 0.5
    Gi = (torch.rand((I4, F), generator=g, device=dev) * 2 - 1) * (6.0 / (I4 + F)) ** 0.5
    Bi = (torch.rand(I4, generator=g, device=dev) - 0.5) * 0.01
    s0 = 2 * UB4
    for k, other, UB4 in ((10, "mfma", UB4), (100, "simple", 2048)):          # (the fp32 MFMA kernel keeps k <= 40 candidates per lane)
        i_scr, v_scr = ops.score_topk(ctx, Gu, Gi, Bi, s0, s0 + UB4, k, excl=pos, algo="screen")
        i_mf, v_mf = ops.score_topk(ctx, Gu, Gi, Bi, s0, s0 + UB4, k, excl=pos, algo=other)
        assert torch.equal(i_scr, i_mf) and torch.equal(v_scr.view(torch.int32), v_mf.view(torch.int32)), k
        users = torch.arange(s0, s0 + UB4, device=dev, dtype=torch.int32)
        assert not bool(_row_members(pos, users, i_scr).any())
        n = 6
        ip = cpu(pos.indptr[s0:s0 + n + 1])
        ix = cpu(pos.indices[int(ip[0]):int(ip[-1])])
        ei, ev = cref.score_topk_f32(cpu(Gu[s0:s0 + n]), cpu(Gi), cpu(Bi), 0, n, k, excl=(ip - ip[0], ix))
        assert np.array_equal(cpu(i_scr[:n]), ei) and np.array_equal(cpu(v_scr[:n]), ev), k


def test_fullsize_metrics_checksum(ctx, world):
    if "idx" not in world:
        pytest.skip("needs the top-k block of the previous test")
    idx, s0 = world["idx"], world["s0"]
    tip, tix = zipf_csr_device(U, I, ctx.device, mean_log=2.0, sigma_log=0.7, dmin=1, dmax=200, seed=99)
    held = ops.DeviceTestSet.from_tensors(tip, tix, None)
    whole = cpu(ops.rec_metrics(ctx, idx, held, 0.0, K, u_start=s0))
    acc = torch.zeros(8, dtype=torch.float64, device=ctx.device)
    for a in range(0, UB, 30_000):                                           # ragged partition
        b = min(UB, a + 30_000)
        ops.rec_metrics(ctx, idx[a:b].contiguous(), held, 0.0, K, u_start=s0 + a, sums=acc)
    torch.cuda.synchronize()
    assert np.allclose(whole, cpu(acc), rtol=1e-12, atol=0)
    assert whole[7] > 0.5 * UB and np.all(whole[:7] >= 0) and np.all(whole[:7] <= whole[7])


def test_fullsize_cml_pair_counts_and_clean_state(ctx, world):
    """CML at the full size (B = 2^20: 1.1e12 (distance, bias) pairs -- only the sorted evaluation can do that).  The two
    coefficient vectors count the SAME set of active pairs from both sides, so their sums agree exactly; the sorted copies are
    sorted; the gradient accumulators are handed back clean; an independent fp64 evaluation of the separable sum agrees."""
    pos = world["pos"]
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    Gu = ((torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
    Gi = ((torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
    Bi = ((torch.rand(I, generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
    st = ops.CmlDeviceState(ctx, Gu, Gi, Bi)
    u, i, j = ops.bpr_sample(ctx, pos, B, seed=9, first_sample=0)
    D = ((st.Gu[u.long()] - st.Gi[j.long()]) ** 2).sum(1).double() - ((st.Gu[u.long()] - st.Gi[i.long()]) ** 2).sum(1).double()
    E = (st.Bi[i.long()] - st.Bi[j.long()]).double()
    margin = 0.5
    st.train_step(u, i, j, 0.001, 0.0, 0.0, margin)
    loss = st.pop_loss()
    al = lambda n: (n + 255) // 256 * 256
    ws = st._cml_ws
    D_, E_, cD_, cE_, Ds_, Es_ = (ws[k * al(B * 4):k * al(B * 4) + B * 4].view(torch.float32) for k in range(6))   # el_cml.hip carve()
    arr = [D_, E_, Ds_, Es_, cD_, cE_]
    assert bool((arr[2][1:] >= arr[2][:-1]).all()) and bool((arr[3][1:] >= arr[3][:-1]).all())
    assert float(arr[4].double().sum()) == float(arr[5].double().sum()) < 0
    assert float((arr[0].double() - D).abs().max()) < 1e-5 and float((arr[1].double() - E).abs().max()) < 1e-7
    # the separable sum in fp64: sum_a [n_a (margin - D_a)] - sum_b [m_b E_b]   (no pair is near the -80 clip at this scale)
    Es, Ds = torch.sort(E).values, torch.sort(D).values
    n_a = torch.searchsorted(Es, margin - D, right=True)
    m_b = torch.searchsorted(Ds, margin - E, right=True)
    exp = float((n_a.double() * (margin - D)).sum() - (m_b.double() * E).sum())
    assert abs(loss - exp) <= 1e-6 * exp, (loss, exp)
    assert not bool(st.gGu.any()) and not bool(st.gGi.any()) and not bool(st.gBi.any())


def test_fullsize_pointwise_step_properties(ctx, world):
    """FunkSVD at the full size: the batch loss equals an independent fp64 evaluation on the same samples; accumulators are
    clean on exit; at Adam step 1 no entry moves by more than lr and an untouched row not at all."""
    pos = world["pos"]
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(8)
    Gu = ((torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
    Gi = ((torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.05).cpu().numpy()
    st = ops.PwmfDeviceState(ctx, Gu, Gi, np.zeros(U, np.float32), np.zeros(I, np.float32), kind="mse", optimizer="adam")
    u, i, y = ops.pointwise_sample(ctx, pos, B, seed=5, first_sample=0)
    before = st.Gu.clone()
    x = (st.Gu[u.long()].double() * st.Gi[i.long()].double()).sum(1)
    exp = float(((y.double() - x) ** 2).mean())
    lr = 0.001
    st.train_step(u, i, y, lr)
    loss = st.pop_loss()
    assert abs(loss - exp) <= 1e-5 * exp, (loss, exp)
    for name in ("gGu", "gGi", "gBu", "gBi"):
        assert not bool(getattr(st, name).any()), name
    moved = (st.Gu - before).abs()
    touched = torch.zeros(U, dtype=torch.bool, device=dev)
    touched[u.long()] = True
    assert float(moved.max()) <= lr * 1.0001
    assert not bool(moved[~touched].any())
    assert float(moved[touched].max()) > 0            # (a batch-MEAN loss: |g| ~ 1e-7 sits below Adam's epsilon, moves are small)
