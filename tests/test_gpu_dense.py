"""GPU parity: fp32 MFMA GEMM (el_gemm_f32) and the Mult-VAE train step / predict against the NumPy oracle."""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from elliot_amd import ops
from oracle import multi_vae as ov
from oracle.sampler import philox4x32_10
from tests.gpu_util import cpu

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("M,N,K", [(300, 513, 77), (128, 128, 32), (512, 600, 5000), (5, 1030, 600), (600, 260, 512)])
def test_gemm_matches_fp64(ctx, tA, tB, M, N, K):
    rs = np.random.RandomState(M + N + K)
    A = rs.normal(size=(K, M) if tA else (M, K)).astype(np.float32)
    Bm = rs.normal(size=(N, K) if tB else (K, N)).astype(np.float32)
    bias = rs.normal(size=N).astype(np.float32)
    d = ctx.device
    ref = (A.T if tA else A).astype(np.float64) @ (Bm.T if tB else Bm).astype(np.float64)
    got = cpu(ops.gemm(ctx, torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d), tA, tB))
    tol = 3e-6 * np.sqrt(K) * 4 + 1e-6 * K * 0.02
    assert np.abs(got - ref).max() < tol, (np.abs(got - ref).max(), tol)
    got = cpu(ops.gemm(ctx, torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d), tA, tB,
                       bias=torch.from_numpy(bias).to(d), act="tanh"))
    assert np.abs(got - np.tanh(ref + bias)).max() < tol
    got = cpu(ops.gemm(ctx, torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d), tA, tB,
                       bias=torch.from_numpy(bias).to(d), act="relu"))
    assert np.abs(got - np.maximum(ref + bias, 0)).max() < tol


@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_three_way_split_is_fp32_grade(ctx, tA, tB, lib_option):
    """k_gemm_b3 (fp32 operands as three bf16 planes, six products on the bf16 matrix instruction) against the fp32 matrix
    instruction on the same operands -- entries spread over 12 orders of magnitude, sums that cancel: its error against fp64 stays
    within 2x the fp32 instruction's own (both are a few ulp of sum |a b|), edge tiles and a split K included."""
    rs = np.random.RandomState(7)
    for M, N, K in ((520, 644, 4096), (136, 260, 40000)):
        A = (rs.normal(size=(K, M) if tA else (M, K)) * 10.0 ** rs.uniform(-6, 6, size=(K, M) if tA else (M, K))).astype(np.float32)
        Bm = (rs.normal(size=(N, K) if tB else (K, N)) * 10.0 ** rs.uniform(-6, 6, size=(N, K) if tB else (K, N))).astype(np.float32)
        d = ctx.device
        a64, b64 = (A.T if tA else A).astype(np.float64), (Bm.T if tB else Bm).astype(np.float64)
        ref, mag = a64 @ b64, np.abs(a64) @ np.abs(b64)
        lib_option("gemm_split", 1)
        got3 = cpu(ops.gemm(ctx, torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d), tA, tB))
        lib_option("gemm_split", 0)
        got1 = cpu(ops.gemm(ctx, torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d), tA, tB))
        e3, e1 = np.abs(got3 - ref) / mag, np.abs(got1 - ref) / mag
        assert e3.max() <= max(2.0 * e1.max(), 4 * 2.0 ** -24), (M, N, K, float(e3.max()), float(e1.max()))
        assert np.sqrt((e3 ** 2).mean()) <= 2.0 * np.sqrt((e1 ** 2).mean()) + 2.0 ** -26, (float(np.sqrt((e3 ** 2).mean())), float(np.sqrt((e1 ** 2).mean())))


@pytest.mark.parametrize("tA,tB", [(False, False), (False, True), (True, False), (True, True)])
def test_gemm_xcd_aware_order_equals_the_plain_grid_bit_for_bit(ctx, tA, tB, lib_option):
    """k_gemm_b3 with its tiles dealt to the XCDs in groups that share an operand strip (option gemm_xcd, the default) against the plain
    3-D grid: the same tiles, the same products in the same order -- identical bits: whole tiles, edge tiles, a K that is no multiple of
    the k tile, bias + activation in the epilogue; the split-K form (partials in the workspace + k_gemm_reduce) against fp64."""
    rs = np.random.RandomState(11)
    d = ctx.device
    for M, N, K in ((512, 26744, 600), (600, 1304, 1500), (260, 520, 8004), (128, 1028, 8000), (8200, 4100, 260), (33000, 1028, 128)):
        A = rs.normal(size=(K, M) if tA else (M, K)).astype(np.float32)
        Bm = rs.normal(size=(N, K) if tB else (K, N)).astype(np.float32)
        bias = rs.normal(size=N).astype(np.float32)
        At, Bt, bt = torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d), torch.from_numpy(bias).to(d)
        outs = {}
        for x in (1, 0):
            lib_option("gemm_xcd", x)
            outs[x] = (ops.gemm(ctx, At, Bt, tA, tB, ws=False).clone(), ops.gemm(ctx, At, Bt, tA, tB, bias=bt, act="relu", ws=False).clone())
        assert torch.equal(outs[1][0].view(torch.int32), outs[0][0].view(torch.int32)), (M, N, K)
        assert torch.equal(outs[1][1].view(torch.int32), outs[0][1].view(torch.int32)), (M, N, K)
        r64 = (A.T if tA else A).astype(np.float64) @ (Bm.T if tB else Bm).astype(np.float64)
        assert np.abs(cpu(outs[1][0]) - r64).max() < 3e-6 * np.sqrt(K) * 6 + 1e-6 * K * 0.02
    lib_option("gemm_xcd", 1)
    M, N, K = 512, 600, 26744                                  # long K, few tiles: split K
    A = rs.normal(size=(K, M) if tA else (M, K)).astype(np.float32)
    Bm = rs.normal(size=(N, K) if tB else (K, N)).astype(np.float32)
    ref = (A.T if tA else A).astype(np.float64) @ (Bm.T if tB else Bm).astype(np.float64)
    got = cpu(ops.gemm(ctx, torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d), tA, tB))
    assert np.abs(got - ref).max() < 3e-6 * np.sqrt(K) * 6 + 1e-6 * K * 0.02


def test_gemm_unaligned_leading_dims(ctx):
    rs = np.random.RandomState(5)
    A = rs.normal(size=(70, 33)).astype(np.float32)       # lda = 33: scalar load path
    Bm = rs.normal(size=(33, 45)).astype(np.float32)
    d = ctx.device
    got = cpu(ops.gemm(ctx, torch.from_numpy(A).to(d), torch.from_numpy(Bm).to(d)))
    assert np.abs(got - A.astype(np.float64) @ Bm.astype(np.float64)).max() < 1e-4


def drop_scale_matrix(users, I, rate, seed, step):
    out = np.ones((len(users), I), np.float32)
    if rate <= 0:
        return out
    for r, u in enumerate(users):
        for i in range(I):
            x = philox4x32_10(int(u), i, step, 0, seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)[0]
            uni = np.float32(x >> 8) * np.float32(1.0 / 16777216.0)
            out[r, i] = 0.0 if uni < np.float32(rate) else np.float32(1.0) / (np.float32(1.0) - np.float32(rate))
    return out


@pytest.mark.parametrize("rate", [0.0, 0.3])
def test_vae_train_steps_and_predict_match_oracle(ctx, rate):
    rs = np.random.RandomState(11)
    U, I, H, L, B = 300, 700, 64, 16, 128
    X = (rs.rand(U, I) < 0.04).astype(np.float32)
    X[np.arange(U), rs.randint(0, I, U)] = 1.0               # every user has >= 1 item
    m = sp.csr_matrix(X)
    m.sort_indices()
    w0 = ov.init_weights(I, H, L, 42)
    for k in ("b1", "bm", "bv", "b3", "b4"):
        w0[k] = rs.normal(scale=0.01, size=w0[k].shape).astype(np.float32)
    lr = 0.001
    st = ops.VaeDeviceState(ctx, w0, max_batch=B)
    orc = ov.MultiVAEOracle(w0, lr)
    csr = ops.DeviceCSR(m.indptr, m.indices, I, ctx.device)
    d = ctx.device
    for s in range(5):
        rows = rs.permutation(U)[:B if s != 3 else 37].astype(np.int32)
        eps = rs.normal(size=(len(rows), L)).astype(np.float32)
        anneal = min(0.2, s / 10.0)
        st.train_step(csr, torch.from_numpy(rows).to(d), lr, anneal, eps=torch.from_numpy(eps).to(d),
                      dropout_rate=rate, dropout_seed=42)
        got = st.pop_loss()
        exp = orc.train_step(X[rows], eps, anneal, drop_scale_matrix(rows, I, rate, 42, s + 1) if rate > 0 else None)
        assert abs(got - exp) <= 1e-4 * abs(exp), (s, got, exp)
        gw = st.weights()
        for k in ov.NAMES:
            err = np.abs(gw[k] - orc.w[k])
            assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (s, k, float(err.max()), float((err > 2e-5).mean()))
    rows = np.arange(40, 40 + 64, dtype=np.int32)
    eps = rs.normal(size=(64, L)).astype(np.float32)
    pred = cpu(st.predict(csr, torch.from_numpy(rows).to(d), eps=torch.from_numpy(eps).to(d)))
    w_dev = st.weights()
    ref = ov.log_softmax(ov.forward(w_dev, X[rows], eps, dtype=np.float64)["logits"])
    assert np.abs(pred - ref).max() < 1e-4
    assert np.abs(np.exp(pred).sum(1) - 1).max() < 1e-4


@pytest.mark.parametrize("rate", [0.0, 0.3])
def test_dae_train_steps_and_predict_match_oracle(ctx, rate):
    """Mult-DAE mode of the same kernels (el_vae_state.dae = 1): tanh latent layer, no sampling, no KL."""
    from oracle import multi_dae as od
    rs = np.random.RandomState(12)
    U, I, H, L, B = 300, 700, 64, 16, 128
    X = (rs.rand(U, I) < 0.04).astype(np.float32)
    X[np.arange(U), rs.randint(0, I, U)] = 1.0
    m = sp.csr_matrix(X)
    m.sort_indices()
    w0 = od.init_weights(I, H, L, 42)
    for k in ("b1", "bm", "b3", "b4"):
        w0[k] = rs.normal(scale=0.01, size=w0[k].shape).astype(np.float32)
    lr = 0.001
    st = ops.VaeDeviceState(ctx, w0, max_batch=B)
    assert st.dae
    orc = od.MultiDAEOracle(w0, lr)
    csr = ops.DeviceCSR(m.indptr, m.indices, I, ctx.device)
    d = ctx.device
    for s in range(5):
        rows = rs.permutation(U)[:B if s != 3 else 37].astype(np.int32)
        st.train_step(csr, torch.from_numpy(rows).to(d), lr, 0.0, eps=None, dropout_rate=rate, dropout_seed=42)
        got = st.pop_loss()
        exp = orc.train_step(X[rows], drop_scale_matrix(rows, I, rate, 42, s + 1) if rate > 0 else None)
        assert abs(got - exp) <= 1e-4 * abs(exp), (s, got, exp)
        gw = st.weights()
        for k in od.NAMES:
            err = np.abs(gw[k] - orc.w[k])
            assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (s, k, float(err.max()), float((err > 2e-5).mean()))
    rows = np.arange(40, 40 + 64, dtype=np.int32)
    pred = cpu(st.predict(csr, torch.from_numpy(rows).to(d)))
    ref = od.log_softmax(od.forward(st.weights(), X[rows], dtype=np.float64)["logits"])
    assert np.abs(pred - ref).max() < 1e-4


@pytest.mark.parametrize("dae", [False, True])
def test_data_parallel_hip_path_equals_one_batch(ctx, dae):
    """parallel.ShardedVae's kernel sequence with two virtual ranks: el_vae_grads on half of the rows each (batch means over the
    WHOLE batch), the gradient buffers added (the all-reduce), el_vae_apply on both: equals the oracle's step on the whole
    batch, replicas stay bit-identical."""
    from oracle import multi_dae as od
    rs = np.random.RandomState(17)
    U, I, H, L, n, lr, G = 260, 500, 32, 8, 48, 0.001, 2
    X = (rs.rand(U, I) < 0.05).astype(np.float32)
    X[np.arange(U), rs.randint(0, I, U)] = 1.0
    m = sp.csr_matrix(X)
    m.sort_indices()
    csr = ops.DeviceCSR(m.indptr, m.indices, I, ctx.device)
    w0 = od.init_weights(I, H, L, 5) if dae else ov.init_weights(I, H, L, 5)
    sts = [ops.VaeDeviceState(ctx, w0, max_batch=n) for _ in range(G)]
    orc = od.MultiDAEOracle(w0, lr) if dae else ov.MultiVAEOracle(w0, lr)
    d = ctx.device
    for s in range(3):
        rows_all = rs.permutation(U)[:G * n].astype(np.int32)
        eps_all = rs.normal(size=(G * n, L)).astype(np.float32)
        for r, st in enumerate(sts):
            st.grads(csr, torch.from_numpy(rows_all[r * n:(r + 1) * n]).to(d), 0.1,
                     eps=None if dae else torch.from_numpy(eps_all[r * n:(r + 1) * n]).to(d), n_global=G * n)
        for a, b in zip(sts[0].dense_grads(), sts[1].dense_grads()):
            tot = a + b
            a.copy_(tot)
            b.copy_(tot)
        loss = 0.0
        for st in sts:
            st.apply(lr)
            loss += st.pop_loss()
        exp = orc.train_step(X[rows_all]) if dae else orc.train_step(X[rows_all], eps_all, 0.1)
        assert abs(loss - exp) <= 1e-4 * abs(exp), (s, loss, exp)
        for a, b in zip(sts[0].w, sts[1].w):
            assert torch.equal(a, b)
        gw = sts[0].weights()
        for k in (od.NAMES if dae else ov.NAMES):
            err = np.abs(gw[k] - orc.w[k])
            assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (s, k, float(err.max()))


def test_vae_step_at_the_ml20m_shape_matches_oracle(ctx):
    """BASELINE configs[2] shape: I = 26 744 items, hidden 600, latent 200, batch 512 (multi_vae.py:57-61,68-69) -- the shapes
    whose GEMMs run the stream-K schedule (512 x 26744 x 600 logits, its two transposed products, the K = 26744 split).
    One el_vae_grads + el_vae_apply per step, two steps, against oracle/multi_vae.py on the densified batch:
      loss        1e-4 relative (north_star)
      gradients   every one of the ten tensors within 2e-5 of its own largest entry (fp32 summation order is the freedom; the
                  products sum up to 26 744 terms)
      weights     after Adam: at most 2e-3 of the entries off by more than 2e-5 (m / (sqrt(v) + eps) flips sign where a
                  gradient cancels to ~0), none by more than 5 lr."""
    rs = np.random.RandomState(21)
    U, I, H, L, B = 1536, 26744, 600, 200, 512
    from elliot_amd.synthetic import zipf_csr
    indptr, indices = zipf_csr(U, I, mean_log=4.5, sigma_log=1.0, dmin=20, dmax=3000, seed=5)
    csr = ops.DeviceCSR(indptr, indices, I, ctx.device)
    w0 = ov.init_weights(I, H, L, 42)
    for k in ("b1", "bm", "bv", "b3", "b4"):
        w0[k] = rs.normal(scale=0.01, size=w0[k].shape).astype(np.float32)
    lr = 0.001
    st = ops.VaeDeviceState(ctx, w0, max_batch=B)
    orc = ov.MultiVAEOracle(w0, lr)
    d = ctx.device
    L2 = st.L
    for s in range(2):
        rows = rs.permutation(U)[:B].astype(np.int32)
        X = np.zeros((B, I), np.float32)
        for r, u in enumerate(rows):
            X[r, indices[indptr[u]:indptr[u + 1]]] = 1.0
        eps = rs.normal(size=(B, L)).astype(np.float32)
        anneal = 0.1
        st.grads(csr, torch.from_numpy(rows).to(d), anneal, eps=torch.from_numpy(eps).to(d))
        got_loss = st.pop_loss()
        c = ov.forward(orc.w, X, eps)
        exp_loss = float(ov.loss_from(c, anneal))
        assert abs(got_loss - exp_loss) <= 1e-4 * abs(exp_loss), (s, got_loss, exp_loss)
        g = ov.gradients(orc.w, c, np.float32(anneal))
        dev_g = {n: cpu(t) for n, t in zip(st.ORDER, st.g)}
        got_g = {"W1": dev_g["W1"], "b1": dev_g["b1"], "Wm": dev_g["Wmv"][:, :L2], "Wv": dev_g["Wmv"][:, L2:], "bm": dev_g["bmv"][:L2],
                 "bv": dev_g["bmv"][L2:], "W3": dev_g["W3"], "b3": dev_g["b3"], "W4": dev_g["W4"], "b4": dev_g["b4"]}
        for k in ov.NAMES:
            scale = float(np.abs(g[k]).max())
            err = float(np.abs(got_g[k] - g[k]).max())
            assert err <= 2e-5 * scale, (s, k, err, scale)
        st.apply(lr)
        assert abs(orc.train_step(X, eps, anneal) - exp_loss) < 1e-9 * abs(exp_loss)
        gw = st.weights()
        for k in ov.NAMES:
            err = np.abs(gw[k] - orc.w[k])
            assert (err > 2e-5).mean() < 2e-3 and err.max() < 5 * lr, (s, k, float(err.max()), float((err > 2e-5).mean()))


@pytest.mark.parametrize("split", [1, 0])
def test_the_vae_step_leaves_the_same_bits_on_every_run(ctx, lib_option, split):
    """Every variable of the Mult-VAE step without float atomics: the first layer's gradient is built from sorted (item, row) lists, the
    split-K products add their partials in a fixed order, and the bias gradients -- the last atomics of the step's variables -- are
    column sums in a fixed order (el_colsum_finish).  30 steps twice from the same state, with the backward pass on two streams: all ten
    variables and their Adam slots bit-identical.  (The scalar loss still collects its per-workgroup shares with a double-precision
    atomic: it may differ in its last bits.)"""
    lib_option("gemm_split", split)
    rs = np.random.RandomState(5)
    U, I, H, L, B = 2000, 5000, 128, 32, 512
    X = (rs.rand(U, I) < 0.01).astype(np.float32)
    X[np.arange(U), rs.randint(0, I, U)] = 1.0
    m = sp.csr_matrix(X)
    m.sort_indices()
    w0 = ov.init_weights(I, H, L, 7)
    d = ctx.device
    csr = ops.DeviceCSR(m.indptr, m.indices, I, d)
    steps = [(torch.from_numpy(rs.permutation(U)[:B].astype(np.int32)).to(d), torch.from_numpy(rs.normal(size=(B, L)).astype(np.float32)).to(d))
             for _ in range(30)]

    def run():
        st = ops.VaeDeviceState(ctx, w0, max_batch=B)
        losses = []
        for s, (rows, eps) in enumerate(steps):
            st.train_step(csr, rows, 0.001, min(0.2, s / 20.0), eps=eps, dropout_rate=0.3, dropout_seed=9)
            losses.append(st.pop_loss())
        torch.cuda.synchronize()
        return st.weights(), [cpu(t) for t in list(st.m) + list(st.v)], losses

    wa, sa, la = run()
    wb, sb, lb = run()
    for k in ov.NAMES:
        assert np.array_equal(wa[k], wb[k]), (k, int((wa[k] != wb[k]).sum()))
    for a, b in zip(sa, sb):
        assert np.array_equal(a, b)
    assert np.allclose(la, lb, rtol=1e-12, atol=0)
