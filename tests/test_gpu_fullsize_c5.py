"""BASELINE configs[4] at its PER-GPU shape under user sharding at N = 8: 6.25 M users x 5 M items, d = 256 (the whole
configuration is 50 M x 5 M x 256 on 8 GPUs; a rank owns U/8 user rows and a replica of the item table).  F = 256 takes code
paths nothing else at full size touches: the 64-lane row groups of the segment kernels (16 B per lane), `launch_mfma<256, 2,
..., 4, 1>` of the fp32 MFMA top-k kernel and the one-column-block (UB = 1) geometry of the screened passes.

Same size-independent properties as tests/test_gpu_fullsize.py (the oracle cannot run at this size, the properties can):
  training  the path bench.py's c5_per_gpu leg runs (cover batches, then 12 pipelined train_step_presorted steps with deferred user and
            item rows, sync()): every batch loss == independent fp64 evaluation of BPRMF_batch_model.py:65-75 (1e-4, the north_star
            tolerance); weights of sampled rows (random users, random items, the hottest items) == oracle/bprmf_batch.py's every-row
            Keras Adam fed with the oracle's gradients of exactly the triplets that touch them; == the every-row two-pass form
            (grads() + apply()) of the library, whose PRE-optimiser gradient rows are checked against the oracle's as well
  top-k     screened (bf16 search / fp32 answer) == fused fp32 MFMA kernel on index lists and score bits for a whole block
            against the 5 M-item catalogue, == the C oracle's fma chain on a sample of users; ordered, in range, no train item
"""
import numpy as np
import pytest
import torch

from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr_device
from oracle import cref
from tests.fullsize_common import bench_path_vs_two_pass
from tests.gpu_util import cpu
from tests.test_gpu_fullsize import _row_members

pytestmark = pytest.mark.gpu

U, I, F, B, K, UB = 6_250_000, 5_000_000, 256, 1 << 20, 10, 16384
STEPS = 12
LR, L_W, L_B = 0.001, 0.1, 0.001


@pytest.fixture(scope="module", params=["series", "exact"])
def c5(ctx, request):
    dev = ctx.device
    # ~33 interactions per user (2e8 in all): the CSR is the exclusion mask of the top-k and the sampler's positives
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.0, sigma_log=1.0, dmin=5, dmax=2000, seed=2345)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
    g = torch.Generator(device=dev)
    g.manual_seed(44)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.05
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.05
    Bi = (torch.rand(I, generator=g, device=dev) - 0.5) * 0.02
    # the path bench.py's c5_per_gpu leg runs (cover batches, pipelined presorted steps, deferred user AND item rows), checked against
    # the fp64 loss, the oracle's every-row Adam on sampled rows and the every-row two-pass form: tests/fullsize_common.py
    rec = bench_path_vs_two_pass(ctx, pos, indptr, indices, Gu, Gi, Bi, B, STEPS, LR, L_W, L_B, replay=request.param)
    del Gu, Gi, Bi
    torch.cuda.empty_cache()
    rec["pos"], rec["st"] = pos, rec["A"]
    return rec


def test_c5_bench_path_losses_oracle_rows_and_two_pass_form(c5):
    assert c5["n_cover"] == -(-max(U, I) // B)
    assert c5["deferred"] and c5["item_deferred"]                     # 4 B <= U and 2 B <= I: user rows AND item rows wait for their replays
    assert all(c5["pending_before_sync"])
    assert min(c5["touch"]) > 10_000                                  # the hot rows really are hot (chunk-crossing segments)
    print("oracle_err", c5["oracle_err"])
    print("vs_two_pass", c5["vs_two_pass"], "exact" if c5["exact"] else "re-association accuracy")


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
