"""BASELINE configs[4] at its PER-GPU shape under user sharding at N = 8: 6.25 M users x 5 M items, d = 256 (the whole
configuration is 50 M x 5 M x 256 on 8 GPUs; a rank owns U/8 user rows and a replica of the item table).  F = 256 takes code
paths nothing else at full size touches: the 64-lane row groups of the segment kernels (16 B per lane), `launch_mfma<256, 2,
..., 4, 1>` of the fp32 MFMA top-k kernel and the one-column-block (UB = 1) geometry of the screened passes.

Same size-independent properties as tests/test_gpu_fullsize.py (the oracle cannot run at this size, the properties can):
  training  batch loss == independent fp64 evaluation of BPRMF_batch_model.py:65-75 on the same triplets (1e-4, the north_star
            tolerance); Adam step 1: untouched rows do not move, touched entries move by at most lr; PRE-optimiser gradients
            and post-Adam weights of sampled rows (random users, random items, the hottest items) == oracle/bprmf_batch.py on
            exactly the triplets that touch them
  top-k     screened (bf16 search / fp32 answer) == fused fp32 MFMA kernel on index lists and score bits for a whole block
            against the 5 M-item catalogue, == the C oracle's fma chain on a sample of users; ordered, in range, no train item
"""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr_device
from oracle import cref
from tests.gpu_util import cpu
from tests.test_gpu_fullsize import _row_members

pytestmark = pytest.mark.gpu

U, I, F, B, K, UB = 6_250_000, 5_000_000, 256, 1 << 20, 10, 16384
LR, L_W, L_B = 0.001, 0.1, 0.001


@pytest.fixture(scope="module")
def c5(ctx):
    dev = ctx.device
    # ~33 interactions per user (2e8 in all): the CSR is the exclusion mask of the top-k and the sampler's positives
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.0, sigma_log=1.0, dmin=5, dmax=2000, seed=2345)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
    g = torch.Generator(device=dev)
    g.manual_seed(44)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.05
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.05
    Bi = (torch.rand(I, generator=g, device=dev) - 0.5) * 0.02
    st = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer="adam_tf_dense")
    assert st.compact                                            # the form bench.py's c5_per_gpu leg runs
    del Gu, Gi, Bi
    torch.cuda.empty_cache()
    return {"pos": pos, "st": st, "g": g}


def _oracle_rows(Gu0, Gi0, Bi0, u, i, j, su, si):
    """oracle/bprmf_batch.py gradients (fp32 and fp64) of the sampled user rows `su` / item rows `si`: the triplets that touch a
    sampled row, re-indexed into small tables holding just the rows they reference (for the sampled rows the sub-batch gradient
    IS the full-batch gradient)."""
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
    g32 = ob.gradients(gu_s, gi_s, bi_s, *a, L_W, L_B)                      # (dBi, dGu, dGi)
    g64 = ob.gradients(gu_s, gi_s, bi_s, *a, L_W, L_B, dtype=np.float64)
    pu, pi = cpu(ru[su]), cpu(ri[si])
    return {"n": int(sel.numel()),
            "gGu": (g32[1][pu], g64[1][pu]), "gGi": (g32[2][pi], g64[2][pi]), "gBi": (g32[0][pi], g64[0][pi])}


def test_c5_gradients_weights_loss_and_adam_properties(ctx, c5):
    from oracle import bprmf_batch as ob
    pos, st, g = c5["pos"], c5["st"], c5["g"]
    dev = ctx.device
    t0 = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=0)
    u, i, j = t0
    assert int(u.min()) >= 0 and int(u.max()) < U and int(j.min()) >= 0 and int(j.max()) < I
    assert bool(_row_members(pos, u, i[:, None]).all()), "a positive is not a train item of its user"
    assert not bool(_row_members(pos, u, j[:, None]).any()), "a negative is a train item of its user"

    hot = torch.argsort(torch.bincount(torch.cat([t0[1], t0[2]]).long(), minlength=I), descending=True)[:3]
    su = torch.randperm(U, generator=g, device=dev)[:1024]
    si = torch.unique(torch.cat([torch.randperm(I, generator=g, device=dev)[:1024], hot]))
    th = {"Gu": cpu(st.Gu[su]), "Gi": cpu(st.Gi[si]), "Bi": cpu(st.Bi[si])}
    m = {k: np.zeros_like(x) for k, x in th.items()}
    v = {k: np.zeros_like(x) for k, x in th.items()}
    for step in range(2):
        if step:
            u, i, j = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=step * B)
        exp = _oracle_rows(st.Gu, st.Gi, st.Bi, u, i, j, su, si)
        assert exp["n"] > 10_000                                  # the hot rows really are hot (chunk-crossing segments)
        # independent fp64 evaluation of the batch loss (BPRMF_batch_model.py:65-75), in slices (3 x B x 256 doubles = 6 GB whole)
        ref = 0.0
        for s in range(0, B, 1 << 18):
            sl = slice(s, s + (1 << 18))
            gu, gi, gj = st.Gu[u[sl].long()].double(), st.Gi[i[sl].long()].double(), st.Gi[j[sl].long()].double()
            bi, bj = st.Bi[i[sl].long()].double(), st.Bi[j[sl].long()].double()
            d = (bi + (gu * gi).sum(1)) - (bj + (gu * gj).sum(1))
            ref += float(torch.nn.functional.softplus(-d.clamp(-80.0, 1e8)).sum()
                         + L_W * 0.5 * ((gu * gu).sum() + (gi * gi).sum() + (gj * gj).sum())
                         + L_B * 0.5 * (bi * bi).sum() + (L_B / 10) * 0.5 * (bj * bj).sum())
            del gu, gi, gj
        if step == 0:
            Gu0 = st.Gu.clone()
        st.grads(u, i, j, L_W, L_B)
        loss = st.pop_loss()
        assert abs(loss - ref) <= 1e-4 * abs(ref), (step, loss, ref)
        got = {"gGu": cpu(st.user_grad_dense()[su]), "gGi": cpu(st.gGi[si]), "gBi": cpu(st.gBi[si])}
        for name in ("gGu", "gGi", "gBi"):
            e32, e64 = exp[name]
            scale = float(np.abs(e64).max())
            err = float(np.abs(got[name] - e64).max())
            ref_err = float(np.abs(e32.astype(np.float64) - e64).max())
            assert err <= max(2e-5 * scale, 4 * ref_err), (step, name, err, ref_err, scale)
        st.apply(LR)
        if step == 0:
            # Adam step 1: an untouched row does not move, a touched entry moves by at most lr
            touched = torch.zeros(U, dtype=torch.bool, device=dev)
            touched[u.long()] = True
            du = (st.Gu - Gu0).abs().amax(1)
            del Gu0
            assert float(du[~touched].max()) == 0.0
            assert float(du.max()) <= LR * (1 + 1e-3) and float(du[touched].max()) > 0.5 * LR
            del du, touched
        for name, gname in (("Gu", "gGu"), ("Gi", "gGi"), ("Bi", "gBi")):
            ob.adam_tf_sparse_apply(th[name], m[name], v[name], exp[gname][0].astype(np.float32), LR, step + 1)
    assert (st.gGu is None or not bool(st.gGu.any())) and not bool(st.gGi.any())       # accumulators zero on exit
    for name, rows in (("Gu", su), ("Gi", si), ("Bi", si)):
        gotw = cpu(getattr(st, name)[rows])
        err = np.abs(gotw - th[name])
        assert float((err > 2e-5).mean()) <= 2e-4 and float(err.max()) < 3 * LR, (name, float(err.max()), float((err > 2e-5).mean()))


def test_c5_topk_screened_vs_fp32_mfma_vs_oracle(ctx, c5):
    pos, st = c5["pos"], c5["st"]
    s0 = 5 * UB
    i_scr, v_scr = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="screen")
    i_mf, v_mf = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="mfma")
    torch.cuda.synchronize()
    assert torch.equal(i_scr, i_mf), "screened and fp32 MFMA kernels disagree on the index lists (F = 256, I = 5 M)"
    assert torch.equal(v_scr.view(torch.int32), v_mf.view(torch.int32)), "screened and fp32 MFMA kernels disagree on the score bits"
    assert bool((v_scr[:, :-1] >= v_scr[:, 1:]).all())
    tie = v_scr[:, :-1] == v_scr[:, 1:]
    assert bool((i_scr[:, :-1][tie] < i_scr[:, 1:][tie]).all())
    assert int(i_scr.min()) >= 0 and int(i_scr.max()) < I
    users = torch.arange(s0, s0 + UB, device=ctx.device, dtype=torch.int32)
    assert not bool(_row_members(pos, users, i_scr).any()), "a train item was recommended"
    n = 3                                                          # 1.3e9 fma per user on one host core
    ip = cpu(pos.indptr[s0:s0 + n + 1])
    ix = cpu(pos.indices[int(ip[0]):int(ip[-1])])
    ei, ev = cref.score_topk_f32(cpu(st.Gu[s0:s0 + n]), cpu(st.Gi), cpu(st.Bi), 0, n, K, excl=(ip - ip[0], ix))
    assert np.array_equal(cpu(i_scr[:n]), ei) and np.array_equal(cpu(v_scr[:n]), ev)
    # idempotence with the item image kept from the previous call
    i2, v2 = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="screen", items_unchanged=True)
    assert torch.equal(i2, i_scr) and torch.equal(v2.view(torch.int32), v_scr.view(torch.int32))
