"""The configuration BASELINE.json's metric is quoted on -- BPRMF d = 128 on 10 M users x 1 M items, B = 2^20, one GPU (bench.py's
headline leg) -- through the path the bench runs, with an answer checked at every stage (tests/fullsize_common.py):

  training  cover batches, then 24 steps of train_step_presorted fed by PrefetchSampler on the side stream (deferred user decay with
            ~10-step replay gaps in steady state, fused item side, row offsets past 2^31 elements, 2^20-triplet segments cut by chunk
            boundaries, the 64-row flush walk over 10 M rows), sync(); against an fp64 evaluation of every batch loss, against
            oracle/bprmf_batch.py's every-row Keras Adam on 1 024 user rows + 1 024 item rows + the 3 hottest items, and against the
            every-row two-pass form (grads() + apply()) of the same library on the sampled rows and over the whole tables --
            in both replay modes of the waiting rows: "series" (closed form; what bench.py's `value` runs) and "exact" (step by step)
  top-k     on the TRAINED tables: screened (algo="auto") == fp32 MFMA kernel on index lists and score bits for a 16 384-user block,
            == the C oracle's fma chain on a sample of users

Reference: BPRMF_batch_model.py:58-88.
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

U, I, F, B, K, UB = 10_000_000, 1_000_000, 128, 1 << 20, 10, 16384
STEPS = 24
LR, L_W, L_B = 0.001, 0.1, 0.001                                    # BPRMF_batch.py:66-71 defaults (what bench.py trains with)


@pytest.fixture(scope="module", params=["series", "exact"])
def c4(ctx, request):
    dev = ctx.device
    free, _ = torch.cuda.mem_get_info()
    if free < (80 << 30):
        pytest.skip("needs ~60 GB of HBM (two 10 M x 128 states + the positives' CSR)")
    # bench.py's headline data: same generator, same parameters, same seed
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=4321)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    lim_u, lim_i = (6.0 / (U + F)) ** 0.5, (6.0 / (I + F)) ** 0.5     # GlorotUniform (BPRMF_batch_model.py:39-42), as bench.py
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * lim_u
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * lim_i
    Bi = torch.zeros(I, device=dev)
    rec = bench_path_vs_two_pass(ctx, pos, indptr, indices, Gu, Gi, Bi, B, STEPS, LR, L_W, L_B, replay=request.param)
    del Gu, Gi, Bi
    torch.cuda.empty_cache()
    rec["pos"] = pos
    return rec


def test_c4_bench_path_losses_oracle_rows_and_two_pass_form(c4):
    assert c4["n_cover"] == -(-U // B)                                # 10 cover steps, as bench.py
    assert c4["deferred"] and not c4["item_deferred"]                 # 4 B <= U: user rows wait; 2 B > I: item rows replayed every step
    assert c4["pending_before_sync"][0]                               # the 24 steps really ran without a flush in between
    assert len(c4["loss_A"]) == c4["n_cover"] + STEPS
    assert min(c4["touch"]) > 2000                                    # the sampled rows (3 hottest items among them) are hit every step
    print("oracle_err", c4["oracle_err"])
    print("vs_two_pass", c4["vs_two_pass"], "exact" if c4["exact"] else "re-association accuracy")


def test_c4_topk_on_the_trained_tables(ctx, c4):
    pos, st = c4["pos"], c4["A"]
    s0 = 7 * UB
    i_a, v_a = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="auto")
    i_m, v_m = ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s0, s0 + UB, K, excl=pos, algo="mfma")
    torch.cuda.synchronize()
    assert torch.equal(i_a, i_m), "screened and fp32 MFMA kernels disagree on the index lists (trained 10M x 1M x 128 tables)"
    assert torch.equal(v_a.view(torch.int32), v_m.view(torch.int32)), "screened and fp32 MFMA kernels disagree on the score bits"
    assert bool((v_a[:, :-1] >= v_a[:, 1:]).all())
    tie = v_a[:, :-1] == v_a[:, 1:]
    assert bool((i_a[:, :-1][tie] < i_a[:, 1:][tie]).all())
    users = torch.arange(s0, s0 + UB, device=ctx.device, dtype=torch.int32)
    assert not bool(_row_members(pos, users, i_a).any()), "a train item was recommended"
    print("distinct top-1 items over the block:", int(torch.unique(i_a[:, 0]).numel()))
    n = 24                                                             # 1.3e8 fma per user on one host core
    ip = cpu(pos.indptr[s0:s0 + n + 1])
    ix = cpu(pos.indices[int(ip[0]):int(ip[-1])])
    ei, ev = cref.score_topk_f32(cpu(st.Gu[s0:s0 + n]), cpu(st.Gi), cpu(st.Bi), 0, n, K, excl=(ip - ip[0], ix))
    assert np.array_equal(cpu(i_a[:n]), ei) and np.array_equal(cpu(v_a[:n]), ev)
