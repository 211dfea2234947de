"""GPU parity of the graph-based BPR head (LightGCN) and of MF2020, through the C ABI:
  el_spmm_csr_f32 / el_lightgcn_propagate   against SciPy's CSR product and oracle/lightgcn.py (rows of 0 .. 3 000 neighbours: empty rows,
                                            one-chunk rows, rows cut into several 512-entry chunks whose partials are added in order)
  LightGcnDeviceState.train_step            against the oracle's train step AND against the fixture the reference's LightGCN_model.py
                                            produced on the tensorflow stand-in (tests/golden/tfshim_lightgcn.npz)
  el_mf2020_train                           against the reference's own MFModel trace (tests/golden/mf2020_ref.npz) and against the oracle
                                            on a longer run with repeated users / items inside and across the chunks
"""
import numpy as np
import pytest
import scipy.sparse as sp
import torch

from elliot_amd import ops
from oracle import lightgcn as ol
from oracle import mf2020 as om
from tests.gpu_util import cpu

pytestmark = pytest.mark.gpu


def _graph(ctx, R, U, I, width):
    ip, ix, v = ops.normalized_bipartite_laplacian(R.indptr, R.indices, U, I)
    return ops.GraphCSR(ctx, ip, ix, v, U, width), sp.csr_matrix((v, ix, ip), shape=(U + I, U + I))


@pytest.mark.parametrize("F", [8, 64, 128, 256])
def test_spmm_equals_scipy_with_rows_of_every_length_and_is_deterministic(ctx, F):
    rs = np.random.RandomState(F)
    U, I = 900, 40                                             # 40 items x up to 900 neighbours; one item connected to all users
    dense = (rs.rand(U, I) < 0.15).astype(np.float32)
    dense[:, 0] = 1.0                                          # a 900-neighbour row: two chunks
    dense[5, :] = 0.0                                          # an isolated user (empty row) ...
    dense[:, 7] = 0.0                                          # ... and an isolated item
    dense[5, 0] = 0.0
    R = sp.csr_matrix(dense)
    g, L = _graph(ctx, R, U, I, F)
    assert int(g.multi_cnt.max()) >= 2 and int((g.indptr[1:] == g.indptr[:-1]).sum()) >= 2
    X0 = torch.from_numpy(rs.normal(size=(U, F)).astype(np.float32)).to(ctx.device)
    X1 = torch.from_numpy(rs.normal(size=(I, F)).astype(np.float32)).to(ctx.device)
    Y0, Y1 = g.spmm(X0, X1)
    ref = L.astype(np.float64) @ np.concatenate([cpu(X0), cpu(X1)]).astype(np.float64)
    got = np.concatenate([cpu(Y0), cpu(Y1)])
    mag = abs(L).astype(np.float64) @ np.abs(np.concatenate([cpu(X0), cpu(X1)])).astype(np.float64)
    assert (np.abs(got - ref) <= 4e-7 * mag + 1e-12).all()
    assert not got[5].any() and not got[U + 7].any()
    Z0, Z1 = g.spmm(X0, X1)                                    # fixed summation order: the same bits again
    assert torch.equal(Y0.view(torch.int32), Z0.view(torch.int32)) and torch.equal(Y1.view(torch.int32), Z1.view(torch.int32))


@pytest.mark.parametrize("n_layers", [0, 1, 2, 3, 4])
def test_lightgcn_propagate_equals_the_oracle(ctx, n_layers):
    rs = np.random.RandomState(n_layers)
    U, I, F = 1500, 300, 64
    R = sp.random(U, I, density=0.03, format="csr", random_state=rs, dtype=np.float32)
    R.data[:] = 1.0
    R = sp.csr_matrix(sp.hstack([sp.csr_matrix(np.ones((U, 1), np.float32)), R[:, 1:]]))     # item 0: 1 500 neighbours
    g, L = _graph(ctx, R, U, I, F)
    Gu = rs.normal(scale=0.3, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.3, size=(I, F)).astype(np.float32)
    st = ops.LightGcnDeviceState(ctx, Gu, Gi, g, n_layers=n_layers)
    st.propagate()
    eu, ei = ol.propagate(Gu, Gi, L, n_layers)
    assert np.abs(cpu(st.Gu) - eu).max() < 2e-6 and np.abs(cpu(st.Gi) - ei).max() < 2e-6


def test_lightgcn_train_steps_equal_the_reference_model_file_and_the_oracle(ctx, golden):
    gold = golden("tfshim_lightgcn.npz")
    U, I = int(gold["U"]), int(gold["I"])
    lr, l_w = float(gold["lr"]), float(gold["l_w"])
    R = sp.csr_matrix((np.ones(len(gold["R_indices"]), np.float32), gold["R_indices"], gold["R_indptr"]), shape=(U, I))
    dev = ctx.device
    for L in (1, 2):
        g, _ = _graph(ctx, R, U, I, int(gold["F"]))
        st = ops.LightGcnDeviceState(ctx, gold[f"L{L}_Gu0"], gold[f"L{L}_Gi0"], g, n_layers=L)
        for step in range(3):
            u, i, j = (torch.from_numpy(gold[f"L{L}_{x}{step}"].astype(np.int32)).to(dev) for x in "uij")
            st.train_step(u, i, j, lr, l_w)
            loss = st.pop_loss()
            assert abs(loss - float(gold[f"L{L}_loss{step}"])) <= 1e-4 * abs(loss), (L, step)       # north_star's tolerance on the loss
            assert np.abs(cpu(st.Gu) - gold[f"L{L}_Gu{step + 1}"]).max() < 5e-6, (L, step)
            assert np.abs(cpu(st.Gi) - gold[f"L{L}_Gi{step + 1}"]).max() < 5e-6, (L, step)
        idx, val = ops.score_topk(ctx, st.Gu, st.Gi, None, 3, 11, 5)
        preds = gold[f"L{L}_preds"]                                                                # after the third step's tables
        order = np.argsort(-preds, axis=1, kind="stable")[:, :5]
        assert np.array_equal(cpu(idx), order)
    # a larger graph, more steps, B >= 2048 (the sorted segment kernels), against the oracle
    rs = np.random.RandomState(5)
    U, I, F, B = 3000, 800, 64, 4096
    R = sp.random(U, I, density=0.02, format="csr", random_state=rs, dtype=np.float32)
    R.data[:] = 1.0
    g, Lm = _graph(ctx, R, U, I, F)
    Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
    st = ops.LightGcnDeviceState(ctx, Gu, Gi, g, n_layers=2)
    orc = ol.LightGCNOracle(Gu, Gi, Lm, 0.001, 0.1, 2)
    for step in range(4):
        u, i, j = rs.randint(0, U, B), rs.randint(0, I, B), rs.randint(0, I, B)
        st.train_step(*(torch.from_numpy(x.astype(np.int32)).to(dev) for x in (u, i, j)), 0.001, 0.1)
        loss, exp = st.pop_loss(), orc.train_step((u, i, j))
        assert abs(loss - exp) <= 1e-4 * abs(exp), (step, loss, exp)
    for name in ("Gu", "Gi"):
        err = np.abs(cpu(getattr(st, name)) - getattr(orc, name))
        assert (err > 2e-5).mean() < 2e-4 and err.max() < 3 * 0.001, (name, err.max())
    assert not bool(st.bpr._Bi.any())


def test_mf2020_equals_the_reference_trace_and_the_oracle(ctx, golden):
    g = golden("mf2020_ref.npz")
    U, I, F = int(g["U"]), int(g["I"]), int(g["F"])
    lr, reg, B = float(g["lr"]), float(g["reg"]), int(g["batch"])
    st = ops.Mf2020DeviceState(ctx, g["P0"], g["Q0"], lr=lr, reg=reg)
    ep = torch.from_numpy(g["epoch"]).to(ctx.device)
    for k in range(int(g["n_batches"])):
        st.train(ep[k * B:(k + 1) * B])
        loss = st.pop_loss()
        assert abs(loss - g["losses"][k]) <= 1e-11 * abs(g["losses"][k]), (k, loss, g["losses"][k])
    for name in ("P", "Q", "bu", "bi"):
        assert np.abs(cpu(getattr(st, name)) - g[name]).max() < 1e-12, name
    assert abs(float(st.gb.item()) - float(g["gb"])) < 1e-12
    assert np.abs(cpu(st.predictions()) - g["preds"]).max() < 1e-11
    # longer, wider, hot rows: 5 000 samples over 300 users x 40 items (every chunk of 128 samples repeats rows), F = 64 and F = 200
    rs = np.random.RandomState(1)
    for F2 in (64, 200):
        P, Q, bu, bi, gb = om.initialize(300, 40, F2, 7)
        st = ops.Mf2020DeviceState(ctx, P, Q, lr=0.05, reg=0.002)
        samples = np.stack([rs.randint(0, 300, 5000), rs.randint(0, 40, 5000), rs.randint(0, 2, 5000)], axis=1).astype(np.int32)
        samples[:700, 0] = 3                                     # one user for 700 samples in a row
        st.train(torch.from_numpy(samples).to(ctx.device))
        exp_loss, gb = om.train_step(P, Q, bu, bi, gb, samples, 0.05, 0.002)
        assert abs(st.pop_loss() - exp_loss) <= 1e-10 * abs(exp_loss)
        assert np.abs(cpu(st.P) - P).max() < 1e-11 and np.abs(cpu(st.Q) - Q).max() < 1e-11
        assert np.abs(cpu(st.bu) - bu).max() < 1e-11 and np.abs(cpu(st.bi) - bi).max() < 1e-11 and abs(float(st.gb.item()) - gb) < 1e-11


def test_ngcf_train_steps_equal_the_reference_model_file(ctx, golden):
    """NgcfDeviceState against the fixture NGCF_model.py produced on the tensorflow stand-in (tests/golden/tfshim_ngcf.npz): three steps,
    two propagation layers -- loss, the full-width tables and every GraphLayers parameter."""
    g = golden("tfshim_ngcf.npz")
    U, I, F = int(g["U"]), int(g["I"]), int(g["F"])
    lr, l_w = float(g["lr"]), float(g["l_w"])
    R = sp.csr_matrix((np.ones(len(g["R_indices"]), np.float32), g["R_indices"], g["R_indptr"]), shape=(U, I))
    ws = [int(x) for x in g["weight_size"]]
    graph, _ = _graph(ctx, R, U, I, max([F] + ws))
    layers = [{"W1": g[f"W_1_{k}_0"], "b1": g[f"b_1_{k}_0"], "W2": g[f"W_2_{k}_0"], "b2": g[f"b_2_{k}_0"]} for k in range(len(ws))]
    st = ops.NgcfDeviceState(ctx, g["Gu0"], g["Gi0"], graph, layers, F, message_dropout=[0.0] * len(ws))
    for step in range(3):
        u, i, j = (torch.from_numpy(g[f"{x}{step}"].astype(np.int32)).to(ctx.device) for x in "uij")
        st.train_step(u, i, j, lr, l_w)
        loss = st.pop_loss()
        assert abs(loss - float(g[f"loss{step}"])) <= 1e-4 * abs(loss), (step, loss, float(g[f"loss{step}"]))
        assert np.abs(cpu(st.Gu) - g[f"Gu{step + 1}"]).max() < 1e-5 and np.abs(cpu(st.Gi) - g[f"Gi{step + 1}"]).max() < 1e-5, step
        for k in range(len(ws)):
            for mine, name in (("W1", "W_1"), ("b1", "b_1"), ("W2", "W_2"), ("b2", "b_2")):
                assert np.abs(cpu(st.layers[k][mine]) - g[f"{name}_{k}_{step + 1}"]).max() < 2e-6, (step, k, name)
    # message dropout: a counter-based mask -- kept entries scaled by 1 / (1 - rate), the expected share dropped, rows stay unit length
    st2 = ops.NgcfDeviceState(ctx, g["Gu0"], g["Gi0"], graph, layers, F, message_dropout=[0.5, 0.0])
    st2.propagate()
    blk = cpu(st2.Gu)[:, F:F + ws[0]]
    assert 0.25 < (blk == 0).mean() < 0.75
    nz = np.linalg.norm(blk, axis=1)
    assert np.all((np.abs(nz - 1) < 1e-5) | (nz == 0))
