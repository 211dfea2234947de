"""BASELINE configs[3] (NeuMF d = 128, tower 512-256-128, 10 M users x 1 M items over 8 GPUs) at BOTH per-GPU shapes, batch 262 144:
  "user"  1.25 M users x 1 M items   -- user tables sharded, item tables replicated (the shape of bench.py's `neumf` leg)
  "item"  10 M users x 125 K items   -- north_star's / configs[3]'s wording: item tables sharded, the two 10 M x 128 user tables
                                        replicated on every rank (5 GB each + gradient table + Adam slots: 41 GB of user-side state)
The NumPy oracle cannot run at this size; these properties can:

  optimiser   the deferred decay of the four embedding tables (el_nmf_state.row_last) against Keras' every-row Adam
              on a SAMPLE of rows: a host shadow moves the sampled rows at every step (el_adam_elem in NumPy fp32) with the
              device's own gradient rows -- rows inside the batch, rows that wait several steps, the hottest items (thousands
              of duplicate samples summed), rows never touched -- and theta, m, v must come out BIT-identical after the sync
  loss        the batch loss == an independent fp64 evaluation of neural_matrix_factorization_model.py:75-106 on the same samples
              with the tables' current rows (1e-4, the north_star tolerance)
  scoring     el_nmf_score_topk of a few users against the whole 1 M-item catalogue == el_nmf_forward on the same pairs
              (logits of the k winners; both read the synced tables)
"""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr_device
from tests.gpu_util import cpu

pytestmark = pytest.mark.gpu

F, B, LR = 128, 262_144, 0.001
UNITS = (4 * F, 2 * F, F)
SHAPES = {"user": (1_250_000, 1_000_000), "item": (10_000_000, 125_000)}


def _adam_np(th, m, v, g, lr_t):
    f = np.float32
    b1, b2, eps = f(0.9), f(0.999), f(1e-7)
    m[...] = m * b1 + g * (f(1) - b1)
    v[...] = v * b2 + (g * g) * (f(1) - b2)
    th[...] = th - (f(lr_t) * m) / (np.sqrt(v) + eps)


@pytest.fixture(scope="module", params=["user", "item"])
def c3(ctx, request):
    dev = ctx.device
    U, I = SHAPES[request.param]
    torch.cuda.empty_cache()
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=77)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
    g = torch.Generator(device=dev)
    g.manual_seed(5)
    glorot = lambda r, c: (torch.rand((r, c), generator=g, device=dev) * 2 - 1) * (6.0 / (r + c)) ** 0.5
    w = {"Umf": glorot(U, F), "Imf": glorot(I, F), "Umlp": glorot(U, F), "Imlp": glorot(I, F), "W": [], "b": []}
    kin = 2 * F
    for n in UNITS:
        w["W"].append(glorot(kin, n))
        w["b"].append(torch.zeros(n, device=dev))
        kin = n
    w["hw"] = glorot(F + UNITS[-1], 1)[:, 0].contiguous()
    w["hb"] = torch.zeros(1, device=dev)
    st = ops.NmfDeviceState(ctx, w, max_batch=B)
    assert st.deferred                                           # the form bench.py's neumf leg runs
    del w
    torch.cuda.empty_cache()
    yield {"pos": pos, "st": st, "U": U, "I": I}
    del st, pos
    torch.cuda.empty_cache()


def _loss64(st, u, i, y):
    """BinaryCrossentropy of the network on (u, i, y) in fp64 from the device's CURRENT table rows (torch on the device: data
    plumbing of the test, not the product path)."""
    d = torch.float64
    ul, il = u.long(), i.long()
    mf = st.tab[0][ul].to(d) * st.tab[1][il].to(d)
    x = torch.cat([st.tab[2][ul].to(d), st.tab[3][il].to(d)], dim=1)
    for W, b in zip(st.W, st.b):
        x = torch.relu(x @ W.to(d) + b.to(d))
    logit = torch.cat([mf, x], dim=1) @ st.hw.to(d) + st.hb.to(d)
    p = torch.sigmoid(logit).clamp(1e-7, 1 - 1e-7)                # K.binary_crossentropy: clip, then epsilon inside the logarithms
    return float(-(y.to(d) * torch.log(p + 1e-7) + (1 - y.to(d)) * torch.log(1 - p + 1e-7)).mean())


def test_deferred_decay_at_the_configs3_shard_shape_on_sampled_rows(ctx, c3):
    st, pos, dev, U, I = c3["st"], c3["pos"], ctx.device, c3["U"], c3["I"]
    rs = np.random.RandomState(3)
    hot_items = torch.topk(torch.bincount(pos.indices.long(), minlength=I).float(), 64).indices
    rows = {0: torch.from_numpy(np.unique(rs.randint(0, U, 2048))).to(dev),
            1: torch.unique(torch.cat([torch.from_numpy(rs.randint(0, I, 2048)).to(dev), hot_items]))}
    tabs = [(0, 0), (1, 1), (2, 0), (3, 1)]                      # (table, side): Umf, Imf, Umlp, Imlp
    sh = {t: [cpu(st.tab[t][rows[s]]), cpu(st.mtab[t][rows[s]]), cpu(st.vtab[t][rows[s]])] for t, s in tabs}
    touched = {0: np.zeros(len(rows[0]), bool), 1: np.zeros(len(rows[1]), bool)}
    for step in range(1, 7):
        u, i, y = ops.pointwise_sample(ctx, pos, B, seed=11, first_sample=(step - 1) * B)
        if step == 1:
            st.sync()
            exp = _loss64(st, u, i, y)                           # tables are current before the first step
        st.grads(u, i, y)
        if step == 1:
            got = st.pop_loss()
            assert abs(got - exp) <= 1e-4 * abs(exp), (got, exp)
        g = {t: cpu(st.gtab[t][rows[s]]) for t, s in tabs}       # the gradient rows step `step` applies
        st.apply(LR)
        lr_t = np.float32(ops.adam_lr_t(LR, step))
        for t, s in tabs:
            _adam_np(*sh[t], g[t], lr_t)
            touched[s] |= g[t].any(axis=1)
        if step in (3, 6):
            st.sync()
            for t, s in tabs:
                for got, exp, what in zip((st.tab[t], st.mtab[t], st.vtab[t]), sh[t], "tmv"):
                    got = cpu(got[rows[s]])
                    assert np.array_equal(got, exp), (step, t, what, int((got != exp).any(axis=1).sum()), float(np.abs(got - exp).max()))
    st.pop_loss()
    # the sample saw every case: rows in a batch, rows that only ever decayed, hot rows
    hot_pos = int(torch.searchsorted(rows[1], hot_items[:1]).item())
    assert touched[0].any() and (~touched[0]).any() and touched[1][hot_pos]
    # accumulators clean, nothing pending
    assert not bool(st.gtab[1][hot_items].any())


def test_scoring_reads_the_synced_tables_at_full_size(ctx, c3):
    st, pos, dev, I = c3["st"], c3["pos"], ctx.device, c3["I"]
    k, nu = 10, 8
    for step in range(2):                                        # leave row updates pending
        u, i, y = ops.pointwise_sample(ctx, pos, B, seed=12, first_sample=step * B)
        st.train_step(u, i, y, LR)
    idx, val = st.score_topk_logits(0, nu, k, excl=pos)          # syncs inside the library
    uu = torch.arange(nu, dtype=torch.int32, device=dev).repeat_interleave(k)
    p = st.forward(uu, idx.reshape(-1).contiguous())
    logit = torch.log(p.double() / (1 - p.double())).reshape(nu, k)
    assert float((logit - val.double()).abs().max()) < 1e-3 * max(1.0, float(val.abs().max()))
    assert bool((val[:, :-1] >= val[:, 1:]).all()) and int(idx.min()) >= 0 and int(idx.max()) < I
    st.pop_loss()
