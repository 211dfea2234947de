"""GPU parity: NeuMF / GMF train step and pair scoring (el_nmf_*) and the point-wise sampler against the oracle."""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr
from oracle import neumf as on
from tests.gpu_util import cpu

pytestmark = pytest.mark.gpu


def run_steps(ctx, w0, U, I, steps=5, B=700, lr=0.002, seed=0):
    rs = np.random.RandomState(seed)
    st = ops.NmfDeviceState(ctx, w0, max_batch=B)
    orc = on.NeuMFOracle(w0, lr)
    d = ctx.device
    for s in range(steps):
        n = B if s != 2 else 33
        u = rs.randint(0, U, n).astype(np.int32)
        i = rs.randint(0, min(I, 30), n).astype(np.int32)       # hot items -> duplicate rows
        y = rs.randint(0, 2, n).astype(np.float32)
        st.train_step(torch.from_numpy(u).to(d), torch.from_numpy(i).to(d), torch.from_numpy(y).to(d), lr)
        got = st.pop_loss()
        exp = orc.train_step(u, i, y)
        assert abs(got - exp) <= 1e-4 * max(abs(exp), 1e-3), (s, got, exp)
        gw = st.weights()
        for k, v in orc.w.items():
            pairs = zip(gw[k], v) if isinstance(v, list) else [(gw[k], v)]
            for a, b in pairs:
                err = np.abs(a - b)
                assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (s, k, float(err.max()), float((err > 2e-5).mean()))
    return st, orc


@pytest.mark.parametrize("F", [8, 32, 10])
def test_neumf_train_matches_oracle(ctx, F):
    U, I = 200, 150
    st, orc = run_steps(ctx, on.init_neumf(U, I, F, 3), U, I)
    # pair scoring == oracle predict on the device's weights
    rs = np.random.RandomState(9)
    u = rs.randint(0, U, 500).astype(np.int32)
    i = rs.randint(0, I, 500).astype(np.int32)
    d = ctx.device
    p = cpu(st.forward(torch.from_numpy(u).to(d), torch.from_numpy(i).to(d)))
    ref = on.forward(st.weights(), u.astype(np.int64), i.astype(np.int64), dtype=np.float64)["p"]
    assert np.abs(p - ref).max() < 1e-5


def test_gmf_train_matches_oracle(ctx):
    U, I = 300, 120
    run_steps(ctx, on.init_gmf(U, I, 16, 5), U, I)


def test_neumf_branches(ctx):
    U, I, F = 100, 90, 8
    w = on.init_neumf(U, I, F, 2)
    mlp_only = {k: v for k, v in w.items() if k not in ("Umf", "Imf")}
    mlp_only["hw"] = w["hw"][F:].copy()
    run_steps(ctx, mlp_only, U, I, steps=3)
    mf_only = {"Umf": w["Umf"], "Imf": w["Imf"], "hw": w["hw"][:F].copy(), "hb": w["hb"]}
    run_steps(ctx, mf_only, U, I, steps=3)


def test_pointwise_sampler_invariants(ctx):
    U, I = 3000, 2000
    indptr, indices = zipf_csr(U, I, mean_log=2.5, sigma_log=0.7, dmin=1, dmax=200, seed=4)
    pos = ops.DeviceCSR(indptr, indices, I, ctx.device)
    n = 200000
    u, i, y = (cpu(t) for t in ops.pointwise_sample(ctx, pos, n, seed=42))
    assert abs(y.mean() - 0.5) < 0.01                               # fair coin (pointwise_pos_neg_sampler.py:39)
    rows = [set(indices[indptr[x]:indptr[x + 1]].tolist()) for x in range(U)]
    for uu, ii, yy in zip(u[:20000], i[:20000], y[:20000]):
        assert (ii in rows[uu]) == (yy == 1.0)
    cnt = np.bincount(u, minlength=U)
    assert cnt.std() < 1.25 * np.sqrt(n / U)
    u2, i2, y2 = (cpu(t) for t in ops.pointwise_sample(ctx, pos, 100, seed=42, first_sample=500))
    assert np.array_equal(u2, u[500:600]) and np.array_equal(i2, i[500:600]) and np.array_equal(y2, y[500:600])


def test_neumf_item_sharded_hip_path_equals_concatenated_batch(ctx):
    """Two virtual ranks on one GPU (the all-reduce is emulated with torch adds): el_nmf_grads with the global-batch
    mean + reduced gradients of the replicated variables + el_nmf_apply == ONE step on the concatenated batch."""
    from elliot_amd import parallel
    U, I, F, n, G, lr = 120, 90, 16, 400, 2, 0.002
    w = on.init_neumf(U, I, F, 11)
    d = ctx.device
    sts, rng = [], []
    for r in range(G):
        lo, hi = parallel.item_range(I, r, G)
        rng.append((lo, hi))
        wl = {k: ([x.copy() for x in v] if isinstance(v, list) else v.copy()) for k, v in w.items()}
        wl["Imf"], wl["Imlp"] = w["Imf"][lo:hi].copy(), w["Imlp"][lo:hi].copy()
        sts.append(ops.NmfDeviceState(ctx, wl, max_batch=n))
    orc = on.NeuMFOracle(w, lr)
    rs = np.random.RandomState(5)
    for step in range(3):
        batches = [(rs.randint(0, U, n), rs.randint(lo, hi, n), rs.randint(0, 2, n).astype(np.float32)) for lo, hi in rng]
        for st, (lo, hi), (u, i, y) in zip(sts, rng, batches):
            st.grads(torch.from_numpy(u.astype(np.int32)).to(d), torch.from_numpy((i - lo).astype(np.int32)).to(d),
                     torch.from_numpy(y).to(d), n_global=G * n)
        lists = [st.replicated_grads("item") for st in sts]
        for gs in zip(*lists):                                   # the all-reduce
            tot = gs[0] + gs[1]
            for g in gs:
                g.copy_(tot)
        loss = 0.0
        for st in sts:
            st.apply(lr)
            loss += st.pop_loss()
        cu, ci, cy = (np.concatenate([b[x] for b in batches]) for x in range(3))
        exp = orc.train_step(cu, ci, cy)
        assert abs(loss - exp) <= 1e-4 * max(abs(exp), 1e-3), (step, loss, exp)
        w0, w1 = sts[0].weights(), sts[1].weights()
        for k in ("Umf", "Umlp", "hw"):
            assert np.array_equal(w0[k], w1[k]), k                 # replicas identical
            err = np.abs(w0[k] - orc.w[k])
            assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (step, k, float(err.max()))
        for st, (lo, hi), wr in zip(sts, rng, (w0, w1)):
            for k in ("Imf", "Imlp"):
                err = np.abs(wr[k] - orc.w[k][lo:hi])
                assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (step, k, float(err.max()))


@pytest.mark.parametrize("rate", [0.25, 0.6])
def test_neumf_dropout_matches_oracle_with_the_same_masks(ctx, rate):
    """Dropout(rate) in front of every Dense, training only: the device's Philox masks are restated in oracle.dropout_masks."""
    rs = np.random.RandomState(6)
    U, I, F, B, lr = 150, 120, 8, 500, 0.002
    w0 = on.init_neumf(U, I, F, 3)
    st = ops.NmfDeviceState(ctx, w0, max_batch=B, dropout=rate, dropout_seed=1234)
    orc = on.NeuMFOracle(w0, lr)
    d = ctx.device
    for s in range(4):
        n = B if s != 2 else 77
        u = rs.randint(0, U, n).astype(np.int32)
        i = rs.randint(0, I, n).astype(np.int32)
        y = rs.randint(0, 2, n).astype(np.float32)
        st.train_step(torch.from_numpy(u).to(d), torch.from_numpy(i).to(d), torch.from_numpy(y).to(d), lr)
        masks = on.dropout_masks(n, [2 * F, 4 * F, 2 * F], rate, 1234, s + 1)
        got, exp = st.pop_loss(), orc.train_step(u, i, y, masks=masks)
        assert abs(got - exp) <= 1e-4 * max(abs(exp), 1e-3), (s, got, exp)
        gw = st.weights()
        for k, v in orc.w.items():
            pairs = zip(gw[k], v) if isinstance(v, list) else [(gw[k], v)]
            for a, b in pairs:
                err = np.abs(a - b)
                assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (s, k, float(err.max()))
    # scoring never drops
    u, i = rs.randint(0, U, 300).astype(np.int32), rs.randint(0, I, 300).astype(np.int32)
    p = cpu(st.forward(torch.from_numpy(u).to(d), torch.from_numpy(i).to(d)))
    ref = on.forward(st.weights(), u.astype(np.int64), i.astype(np.int64), dtype=np.float64)["p"]
    assert np.abs(p - ref).max() < 1e-5


def test_pointwise_sampler_records_give_the_same_samples(ctx):
    indptr, indices = zipf_csr(4000, 900, mean_log=2.2, sigma_log=1.2, dmin=1, dmax=850, seed=3)
    pos = ops.DeviceCSR(indptr, indices, 900, ctx.device)
    a = ops.pointwise_sample(ctx, pos, 150000, seed=5, first_sample=777, use_meta=False)
    b = ops.pointwise_sample(ctx, pos, 150000, seed=5, first_sample=777, use_meta=True)
    for x, y in zip(a, b):
        assert torch.equal(x, y)


def _relu_branch_audit(st, w, n):
    """The device's ReLU branch pattern of the batch it just evaluated (mask[l] = act[l] > 0: the rule of k_relu_bwd_colsum and of
    the head kernel) against fp64 pre-activations computed FROM THE DEVICE'S OWN INPUT of each layer -- so that a difference is that
    layer's product alone, not error carried up from the layers below.  Returns (masks, flips): flips lists, per unit evaluation
    where the two patterns differ, (layer, |z64|, bound) with bound = the worst-case fp32 round-off of a K-term chain,
    2 K 2^-24 sum_k |x_k w_k| (x2: the split GEMM is within twice the fp32 instruction's own error, tests/test_gpu_dense.py)."""
    masks, flips = [], []
    x = cpu(st.X0[:n]).astype(np.float64)
    for l in range(len(st.units)):
        W, b = np.asarray(w["W"][l], np.float64), np.asarray(w["b"][l], np.float64)
        z = x @ W + b
        act = cpu(st.act[l][:n])
        m = act > 0
        masks.append(m)
        for r, c in zip(*np.nonzero(m != (z > 0))):
            bound = 2.0 * W.shape[0] * 2.0 ** -24 * float(np.abs(x[r]) @ np.abs(W[:, c]) + abs(b[c]))
            flips.append((l, float(abs(z[r, c])), bound))
        # (and where the branch is 'on', the device's activation is the fp64 one to fp32 round-off)
        assert float(np.abs(act - np.maximum(z, 0)).max()) <= 1e-5 * max(1.0, float(np.abs(z).max())), l
        x = act.astype(np.float64)
    return masks, flips


@pytest.mark.parametrize("split", ["0", "1"])
def test_neumf_step_at_d128_matches_oracle(ctx, lib_option, split):
    """BASELINE configs[3] model shape: d = 128, tower (512, 256, 128) (neural_matrix_factorization.py:71-72), batch 65 536 -- the
    GEMM shapes of the full-size run, on the fp32 matrix instruction (option gemm_split = 0) and on the three-way bf16 split (=1, the
    default for these shapes); users x items kept small so that the NumPy oracle finishes in seconds.  Ten batches.

    The ReLU derivative is a step function: of the 58 M unit evaluations of a batch a few have a pre-activation within fp32
    round-off of 0 and take the other branch than an fp64 evaluation would; such a unit moves one sample's contribution in every
    gradient entry below it.  So the comparison is made under the DEVICE's branch pattern, after an audit of that pattern
    (_relu_branch_audit): every unit where it differs from fp64 must have |z| inside the worst-case fp32 round-off of its own dot
    product, and there must be only a few.  Then, with no further allowance:
      loss: 1e-4 relative (north_star's tolerance);
      every gradient tensor: <= 1e-3 of its entries further from the fp64 gradient than max(2e-5 of the tensor's largest entry,
        64x the fp32 oracle's own rms distance from fp64), none further than 5 % of the largest entry;
      weights after Keras Adam (first batch): <= 2e-3 of the entries off by more than 2e-5, none by more than 5 lr.
    Match: neural_matrix_factorization_model.py:96-106."""
    lib_option("gemm_split", int(split))
    U, I, F, B, lr = 20000, 8000, 128, 65536, 0.001
    w0 = on.init_neumf(U, I, F, 3)
    st = ops.NmfDeviceState(ctx, w0, max_batch=B)
    orc = on.NeuMFOracle(w0, lr)
    d = ctx.device
    names = ["Umf", "Imf", "Umlp", "Imlp"]
    own_rel = {}
    total_flips = 0
    for seed in range(10):
        rs = np.random.RandomState(100 + seed)
        u = rs.randint(0, U, B).astype(np.int32)
        i = (rs.zipf(1.2, B) % I).astype(np.int32)                 # popular items: long duplicate-row sums
        y = rs.randint(0, 2, B).astype(np.float32)
        u64, i64 = u.astype(np.int64), i.astype(np.int64)
        # every batch starts from the DEVICE's weights on both sides
        for k, v in st.weights().items():
            orc.w[k] = [np.array(x, np.float32, copy=True) for x in v] if isinstance(v, list) else np.array(v, np.float32, copy=True)
        st.grads(torch.from_numpy(u).to(d), torch.from_numpy(i).to(d), torch.from_numpy(y).to(d))
        got = st.pop_loss()
        masks, flips = _relu_branch_audit(st, orc.w, B)
        for (l, z, bound) in flips:
            assert z <= bound, (seed, l, z, bound)                  # the other branch only where the sign of z is undecidable in fp32
        assert len(flips) <= 24, (seed, [(l, z) for l, z, _ in flips])
        total_flips += len(flips)
        c64 = on.forward(orc.w, u64, i64, dtype=np.float64)
        exp = float(on.bce(c64["p"], y.astype(np.float64)))
        assert abs(got - exp) <= 1e-4 * abs(exp), (seed, got, exp)
        g64 = on.gradients(orc.w, c64, u64, i64, y.astype(np.float64), relu_masks=masks)
        if seed == 0:
            # what fp32 summation costs the ORACLE itself under the same branch pattern (rms of its fp32 - fp64 gradients, relative to
            # the tensor's largest entry; measured on the first batch, the batches are draws of one distribution): the dense-layer
            # gradients sum 65 536 signed terms that cancel 60-fold; NumPy's BLAS adds them in blocks, the MFMA chain strictly in k order
            c32 = on.forward(orc.w, u64, i64)
            g32 = on.gradients(orc.w, c32, u64, i64, y, relu_masks=masks)
        got_g = {n: cpu(t) for n, t in zip(names, st.gtab)}
        got_g.update({"hw": cpu(st.ghw), "hb": cpu(st.ghb)})

        def close(a, b32, b64, what):
            b64 = np.asarray(b64, np.float64).reshape(a.shape)
            scale = float(np.abs(b64).max())
            if seed == 0:
                own_rel[what] = float(np.sqrt(np.mean((np.asarray(b32, np.float64).reshape(a.shape) - b64) ** 2))) / scale
            tol = max(2e-5, 64 * own_rel[what]) * scale
            err = np.abs(a - b64)
            assert float((err > tol).mean()) <= 1e-3 and float(err.max()) <= 0.05 * scale, \
                (split, seed, what, float((err > tol).mean()), float(err.max()), tol, scale, len(flips))

        for k in names + ["hw", "hb"]:
            close(got_g[k], g32[k] if seed == 0 else None, g64[k], k)
        for l in range(3):
            close(cpu(st.gW[l]), g32["W"][l] if seed == 0 else None, g64["W"][l], ("W", l))
            close(cpu(st.gb[l]), g32["b"][l] if seed == 0 else None, g64["b"][l], ("b", l))
        st.apply(lr)
        if seed == 0:
            orc.train_step(u, i, y)
            gw = st.weights()
            for k, v in orc.w.items():
                pairs = zip(gw[k], v) if isinstance(v, list) else [(gw[k], v)]
                for a, b in pairs:
                    err = np.abs(a - b)
                    assert int((err > 2e-5).sum()) <= max(2, int(2e-3 * err.size)) and err.max() < 5 * lr, (k, float(err.max()), int((err > 2e-5).sum()))
    print(f"gemm_split={split}: {total_flips} ReLU units took the other branch than fp64 over 10 batches x 58.7 M unit evaluations")


def test_pointwise_replay_sampler_emits_the_reference_stream(ctx, golden):
    """`sampler: replay` of the point-wise models: device batches == the stream of the reference's own Sampler.step."""
    from elliot_amd.dataset.samplers import pointwise_pos_neg_sampler as pps
    from elliot_amd.synthetic import small_dataset
    g = golden("pointwise_sampler_ref.npz")
    _, _, itd = small_dataset(200, 150, seed=0)
    s = pps.Sampler(itd, ctx=ctx, replay=True)
    assert not s.philox
    got = [tuple(cpu(t) for t in b) for b in s.step(3000, 512)] + [tuple(cpu(t) for t in b) for b in s.step(3000, 700)]
    u, i, y = (np.concatenate([b[k] for b in got]) for k in range(3))
    assert u.dtype == np.int32 and y.dtype == np.float32
    assert np.array_equal(u, g["u"]) and np.array_equal(i, g["i"]) and np.array_equal(y, g["b"].astype(np.float32))


def _adam_np(th, m, v, g, lr_t):
    """el_adam_elem (csrc/el_common.h) in NumPy fp32: every operation rounded on its own, the order of the kernel."""
    f = np.float32
    b1, b2, eps = f(0.9), f(0.999), f(1e-7)
    m[...] = m * b1 + g * (f(1) - b1)
    v[...] = v * b2 + (g * g) * (f(1) - b2)
    th[...] = th - (f(lr_t) * m) / (np.sqrt(v) + eps)


@pytest.mark.parametrize("model,F,hist", [("neumf", 128, None), ("neumf", 40, 4), ("gmf", 64, 3), ("neumf", 300, 5), ("gmf", 33, None)])
def test_deferred_decay_replays_the_every_row_adam_bit_for_bit(ctx, model, F, hist, monkeypatch):
    """The embedding tables under the deferred decay (el_nmf_state.row_last) against an EAGER shadow on the host that moves every
    row at every step with the device's own gradient rows: theta, m, v of all four tables bit-identical whenever the tables are
    read -- after long stretches without a read (rows untouched for 9 steps are replayed at once), after reads in the middle
    (forward / weights / scoring sync), with duplicate rows in a batch, rows that are never touched, and with a 3..5-step lr
    history that fills up and restarts.  F = 300 takes the chunked row loop, 40 the sub-wave one, 33 (odd: rows not 8-byte aligned) the scalar one."""
    if hist:
        monkeypatch.setattr(ops.NmfDeviceState, "_LR_HIST", hist)
    U, I, B, lr = 900, 700, 256, 0.01
    w0 = on.init_neumf(U, I, F, 11, units=[64, 32, 16]) if model == "neumf" else on.init_gmf(U, I, F, 11)
    st = ops.NmfDeviceState(ctx, w0, max_batch=B, deferred=True)
    assert st.deferred
    names = [n for n in ("Umf", "Imf", "Umlp", "Imlp") if n in w0]
    tix = {"Umf": 0, "Imf": 1, "Umlp": 2, "Imlp": 3}
    sh = {n: [np.array(w0[n], np.float32, copy=True), np.zeros_like(w0[n], dtype=np.float32), np.zeros_like(w0[n], dtype=np.float32)] for n in names}
    rs = np.random.RandomState(3)
    d = ctx.device

    def check(tag):
        st.sync()
        for n in names:
            t = tix[n]
            for got, exp, what in zip((st.tab[t], st.mtab[t], st.vtab[t]), sh[n], "tmv"):
                got = cpu(got)
                assert np.array_equal(got, exp), (tag, n, what, int((got != exp).sum()), float(np.abs(got - exp).max()))

    for step in range(1, 25):
        n = B if step % 5 else 17
        u = rs.randint(0, U // 3 if step % 2 else U, n).astype(np.int32)             # a third of the users: the rest wait
        i = (rs.zipf(1.3, n) % (I - 50)).astype(np.int32)                             # hot items; the last 50 never appear
        y = rs.randint(0, 2, n).astype(np.float32)
        st.grads(torch.from_numpy(u).to(d), torch.from_numpy(i).to(d), torch.from_numpy(y).to(d))
        g = {nm: cpu(st.gtab[tix[nm]]) for nm in names}                               # the gradient rows the apply will consume
        st.apply(lr)
        for nm in names:
            _adam_np(*sh[nm], g[nm], np.float32(ops.adam_lr_t(lr, step)))
            assert not cpu(st.gtab[tix[nm]]).any()                                    # accumulators zero again
        if step in (1, 2, 11, 12, 21):
            check(step)
        if step == 15:                                                                # a read through the library syncs by itself
            uu = torch.arange(0, 50, dtype=torch.int32, device=d)
            p = cpu(st.forward(uu, uu))
            ww = st.weights()
            ref = on.forward({**ww, **{k: sh[k][0] for k in names}}, np.arange(50), np.arange(50), dtype=np.float64)["p"]
            assert np.abs(p - ref).max() < 1e-5
    check("end")
    # switching the feature off leaves an eager state that carries on from the same numbers
    st.set_deferred(False)
    u = torch.from_numpy(rs.randint(0, U, B).astype(np.int32)).to(d)
    i = torch.from_numpy(rs.randint(0, I, B).astype(np.int32)).to(d)
    st.grads(u, i, torch.ones(B, device=d))
    g = {nm: cpu(st.gtab[tix[nm]]) for nm in names}
    st.apply(lr)
    for nm in names:
        _adam_np(*sh[nm], g[nm], np.float32(ops.adam_lr_t(lr, 25)))
    for nm in names:
        assert np.array_equal(cpu(st.tab[tix[nm]]), sh[nm][0]), nm


def _flat_weights(w):
    out = {}
    for k, v in w.items():
        if isinstance(v, list):
            for j, a in enumerate(v):
                out[f"{k}{j}"] = np.asarray(a)
        else:
            out[k] = np.asarray(v)
    return out


@pytest.mark.parametrize("model,F,split", [("neumf", 64, 1), ("neumf", 64, 0), ("gmf", 32, 1), ("neumf", 33, 1), ("neumf", 128, 1)])
def test_the_step_is_deterministic_and_its_forms_agree_bit_for_bit(ctx, lib_option, model, F, split):
    """The NeuMF / GMF step without float atomics: the embedding rows of a batch are walked as sorted segments
    (k_nmf_seg_fwd / k_nmf_seg_bwd; hot items of the Zipf batch take the long-segment kernel), the Dense biases', the head's and the
    loss' reductions over the batch add workgroup partials in a fixed order.  50 steps twice from the same state: every variable and the
    loss history bit-identical; the fused step (sums + Adam in one pass, deferred decay), el_nmf_grads + el_nmf_apply (gradient rows through
    gtab) and the eager every-row form give the same bits again."""
    lib_option("gemm_split", split)
    U, I, B, lr, steps = 6000, 2500, 4096, 0.003, 50
    units = [4 * F, 2 * F, F] if F != 33 else [64, 32, 16]
    w0 = on.init_neumf(U, I, F, 21, units=units) if model == "neumf" else on.init_gmf(U, I, F, 21)
    d = ctx.device
    rs = np.random.RandomState(8)
    batches = []
    for s in range(steps):
        n = B if s % 7 else 1000
        u = rs.randint(0, U, n).astype(np.int32)
        i = (rs.zipf(1.25, n) % I).astype(np.int32)                     # a few items hold hundreds of samples each
        y = rs.randint(0, 2, n).astype(np.float32)
        batches.append(tuple(torch.from_numpy(a).to(d) for a in (u, i, y)))
    assert max(np.bincount(cpu(b[1])).max() for b in batches) > 200      # the long-segment kernel runs

    side = torch.cuda.Stream(device=d)

    def run(form):
        st = ops.NmfDeviceState(ctx, w0, max_batch=B, deferred=(form != "eager"))
        losses = []
        for u, i, y in batches:
            if form == "two_pass":
                st.grads(u, i, y)
                st.apply(lr)
            elif form == "presort":                                   # keys ordered ahead of the step on another stream (el_nmf_presort)
                side.wait_stream(torch.cuda.current_stream())
                with torch.cuda.stream(side):
                    st.presort(u, i)
                torch.cuda.current_stream().wait_stream(side)
                st.train_step(u, i, y, lr)
            else:
                st.train_step(u, i, y, lr)
            losses.append(st.pop_loss())
        w = _flat_weights(st.weights())
        for t, nm in enumerate(("Umf", "Imf", "Umlp", "Imlp")):
            if st.mtab[t] is not None:
                w["m_" + nm], w["v_" + nm] = cpu(st.mtab[t]), cpu(st.vtab[t])
        return w, losses

    a, la = run("fused")
    assert np.isfinite(la).all()
    for form in ("fused", "two_pass", "eager", "presort"):
        b, lb = run(form)
        assert la == lb, (form, [k for k in range(steps) if la[k] != lb[k]][:5])
        for k in a:
            assert np.array_equal(a[k], b[k]), (form, k, int((a[k] != b[k]).sum()), float(np.abs(a[k] - b[k]).max()))


@pytest.mark.parametrize("model,F", [("neumf", 32), ("gmf", 64), ("neumf", 128)])
def test_the_four_samples_per_wave_head_agrees_with_the_one_sample_form(ctx, lib_option, model, F):
    """k_nmf_head4 (16 lanes per sample, option nmf_head4 = 1, the default where the rows are 16-byte aligned) against k_nmf_head: the
    two order the feature sums differently, so probabilities, the loss and every gradient agree at fp32 rounding level, not bit for bit;
    a batch that is no multiple of 16 and one smaller than a workgroup's share are in."""
    U, I = 3000, 2000
    units = [4 * F, 2 * F, F]
    w0 = on.init_neumf(U, I, F, 5, units=units) if model == "neumf" else on.init_gmf(U, I, F, 5)
    d = ctx.device
    rs = np.random.RandomState(2)
    out = {}
    for h4 in (1, 0):
        lib_option("nmf_head4", h4)
        st = ops.NmfDeviceState(ctx, w0, max_batch=5000, deferred=False)
        res = []
        for n in (5000, 4099, 7):
            u = torch.from_numpy(rs.randint(0, U, n).astype(np.int32)).to(d)
            i = torch.from_numpy(rs.randint(0, I, n).astype(np.int32)).to(d)
            y = torch.from_numpy(rs.randint(0, 2, n).astype(np.float32)).to(d)
            p = cpu(st.forward(u, i))
            st.grads(u, i, y)
            res.append((p, st.pop_loss(), cpu(st.ghw), cpu(st.ghb) if st.head_bias else None, cpu(st.dlogit[:n]),
                        [cpu(g) for g in st.gW], cpu(st.gtab[0]) if st.gtab[0] is not None else None))
            st.apply(0.001)
        out[h4] = res
        rs = np.random.RandomState(2)
    for a, b in zip(out[1], out[0]):
        assert np.abs(a[0] - b[0]).max() < 2e-6
        assert abs(a[1] - b[1]) <= 2e-6 * max(abs(b[1]), 1e-3)
        for x, y in ((a[2], b[2]), (a[3], b[3]), (a[4], b[4]), (a[6], b[6])):
            if x is not None:
                sc = max(float(np.abs(y).max()), 1e-12)
                assert np.abs(x - y).max() <= 2e-5 * sc, float(np.abs(x - y).max() / sc)
        for x, y in zip(a[5], b[5]):
            sc = max(float(np.abs(y).max()), 1e-12)
            assert np.abs(x - y).max() <= 5e-5 * sc


@pytest.mark.parametrize("model,F", [("neumf", 64), ("gmf", 32), ("neumf", 33)])
def test_series_replay_of_the_embedding_rows_stays_at_rounding_level(ctx, model, F):
    """el_nmf_state.replay_series: waiting embedding rows brought forward in closed form (four row-level sums over the lr_t history, O(1) per
    element) against the step-by-step replay -- which is bit-identical to the eager every-row Adam (test above).  40 steps in which most
    users wait several steps between two of their batches and 10 % of the items never appear: losses agree to 1e-5, the embedding tables
    and their Adam slots to 1e-4 of the table's scale on all but 1e-4 of the elements (median below 1e-6): fp32 rounding, amplified by
    Adam like any other reordering -- not a different trajectory."""
    U, I, B, lr, steps = 5000, 3000, 1024, 0.003, 40
    units = [4 * F, 2 * F, F] if F != 33 else [64, 32, 16]
    w0 = on.init_neumf(U, I, F, 4, units=units) if model == "neumf" else on.init_gmf(U, I, F, 4)
    d = ctx.device
    rs = np.random.RandomState(12)
    batches = []
    for s in range(steps):
        u = rs.randint(0, U // 2 if s % 3 else U, B).astype(np.int32)
        i = (rs.zipf(1.3, B) % (I - I // 10)).astype(np.int32)
        y = rs.randint(0, 2, B).astype(np.float32)
        batches.append(tuple(torch.from_numpy(a).to(d) for a in (u, i, y)))
    res = {}
    for mode in ("exact", "series"):
        st = ops.NmfDeviceState(ctx, w0, max_batch=B, deferred=True, replay=mode)
        losses = []
        for k, (u, i, y) in enumerate(batches):
            st.train_step(u, i, y, lr)
            losses.append(st.pop_loss())
            if k == 17:
                st.sync()                                            # a flush in the middle (k_nmf_flush_rows takes the same mode)
        w = _flat_weights(st.weights())
        for t, nm in enumerate(("Umf", "Imf", "Umlp", "Imlp")):
            if st.mtab[t] is not None:
                w["m_" + nm], w["v_" + nm] = cpu(st.mtab[t]), cpu(st.vtab[t])
        res[mode] = (w, losses)
    (wa, la), (wb, lb) = res["exact"], res["series"]
    assert np.allclose(la, lb, rtol=1e-5, atol=1e-7), float(np.abs(np.array(la) - np.array(lb)).max())
    moved = False
    for k in wa:
        sc = max(float(np.abs(wa[k]).max()), 1e-12)
        rel = np.abs(wa[k] - wb[k]) / sc
        if rel.size >= 1000:
            assert (rel > 1e-4).mean() < 1e-4 and np.median(rel) < 1e-6, (k, float(rel.max()), float((rel > 1e-4).mean()), float(np.median(rel)))
        else:
            assert rel.max() < 1e-4, (k, float(rel.max()))            # (biases, the head: a handful of elements)
        moved = moved or bool((wa[k] != wb[k]).any())
    assert moved                                                     # (the two modes are different arithmetic: identical bits would mean the switch is dead)


def test_a_pending_presort_is_left_alone_by_a_step_on_other_arrays(ctx):
    """el_nmf_presort bookkeeping: a batch ordered ahead stays pending while a step runs on OTHER arrays (that step sorts for itself
    into the other sort set), is consumed by the step that names the same arrays, and is forgotten when the activation buffers -- and
    with them the workspace -- are re-allocated for a larger batch.  Every variant ends on the same bits as plain steps."""
    U, I, F, B, lr = 4000, 2500, 32, 2048, 0.003
    w0 = on.init_neumf(U, I, F, 3, units=[64, 32, 16])
    d = ctx.device
    rs = np.random.RandomState(6)
    mk = lambda n: tuple(torch.from_numpy(a).to(d) for a in (rs.randint(0, U, n).astype(np.int32), (rs.zipf(1.3, n) % I).astype(np.int32),
                                                             rs.randint(0, 2, n).astype(np.float32)))
    b1, b2, b3, big = mk(B), mk(B), mk(B), mk(3 * B)

    ref = ops.NmfDeviceState(ctx, w0, max_batch=B)
    for b in (b1, b2, b3):
        ref.train_step(*b, lr)
    ref.ensure_batch(3 * B)
    ref.train_step(*big, lr)
    want = _flat_weights(ref.weights())

    st = ops.NmfDeviceState(ctx, w0, max_batch=B)
    st.presort(b2[0], b2[1])                 # b2 ordered ahead ...
    st.train_step(*b1, lr)                   # ... a step on other arrays sorts for itself
    st.train_step(*b2, lr)                   # consumes the pending order
    st.presort(b3[0], b3[1])
    st.presort(b3[0], b3[1])                 # ordering the same batch twice is harmless
    st.train_step(*b3, lr)
    st.presort(b1[0], b1[1])                 # pending when the buffers grow: forgotten with the old workspace
    st.ensure_batch(3 * B)
    st.train_step(*big, lr)
    got = _flat_weights(st.weights())
    for k in want:
        assert np.array_equal(want[k], got[k]), (k, int((want[k] != got[k]).sum()))
