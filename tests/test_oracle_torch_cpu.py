"""oracle/torch_cpu.py (the N-thread CPU baseline of bench.py) computes what the NumPy / C oracles compute."""
import numpy as np
import torch

from oracle import bprmf_batch as ob
from oracle import cref
from oracle import torch_cpu as tc


def test_train_step_equals_numpy_oracle():
    rs = np.random.RandomState(3)
    U, I, F, B = 400, 250, 32, 3000
    Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
    Bi = rs.normal(scale=0.01, size=I).astype(np.float32)
    a = tc.BprmfBatchTorchCpu(Gu, Gi, Bi, 0.01, 0.1, 0.001)
    b = ob.BPRMFBatchOracle(Gu, Gi, Bi, 0.01, 0.1, 0.001)
    for _ in range(4):
        u, i, j = rs.randint(0, U, B), rs.randint(0, 30, B), rs.randint(0, I, B)
        la = a.train_step(torch.from_numpy(u), torch.from_numpy(i), torch.from_numpy(j))
        lb = b.train_step((u, i, j))
        assert abs(la - lb) <= 1e-5 * abs(lb)
    for x, y in ((a.Gu, b.Gu), (a.Gi, b.Gi), (a.Bi, b.Bi)):
        err = np.abs(x.numpy() - y)
        assert (err > 2e-5).mean() <= 1e-3 and err.max() < 0.05


def test_predict_topk_equals_c_oracle_sets():
    rs = np.random.RandomState(4)
    U, I, F, k = 60, 500, 16, 10
    Gu = rs.normal(size=(U, F)).astype(np.float32)
    Gi = rs.normal(size=(I, F)).astype(np.float32)
    Bi = rs.normal(size=I).astype(np.float32)
    rows = [np.sort(rs.choice(I, rs.randint(0, 20), replace=False)) for _ in range(U)]
    indptr = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.int64)
    indices = np.concatenate(rows).astype(np.int32)
    idx, val = tc.predict_topk(torch.from_numpy(Gu), torch.from_numpy(Gi), torch.from_numpy(Bi), torch.from_numpy(indptr),
                               torch.from_numpy(indices.astype(np.int64)), k)
    ei, ev = cref.score_topk_f32(Gu, Gi, Bi, 0, U, k, excl=(indptr, indices))
    assert np.array_equal(idx.numpy(), ei)               # (well-separated random scores: BLAS order vs fma chain cannot flip a rank)
    assert np.allclose(val.numpy(), ev, rtol=1e-5, atol=1e-5)
    assert tc.use_all_cores() >= 1
