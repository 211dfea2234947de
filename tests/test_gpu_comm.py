"""el_comm_* (RCCL behind the C ABI) on the one GPU of the dev box: a one-rank communicator must behave as the identity
collectives, asynchronously on the caller's stream order, and the sharded trainers must run through it unchanged.
(N > 1 ranks need N GPUs: the driver's SCALE run; the call pattern is the one torch.distributed's RCCL backend issues.)"""
import numpy as np
import pytest
import torch

from elliot_amd import ops, parallel
from oracle import bprmf_batch as ob
from tests.gpu_util import cpu

pytestmark = pytest.mark.gpu


def test_one_rank_communicator_collectives(ctx):
    coll = parallel.RcclAbiCollectives(ctx, 0, 1)
    d = ctx.device
    g = torch.Generator(device=d)
    g.manual_seed(0)
    a = torch.randn(1 << 20, device=d, generator=g)
    ref = a.clone()
    w = coll.all_reduce_sum(a, async_op=True)
    w.wait()
    assert torch.equal(a, ref)                                   # sum over one rank
    part = torch.randn((1000, 64), device=d, generator=g)
    assert torch.equal(coll.all_gather(part), part)
    full = torch.zeros((1000, 64), device=d)
    coll.all_gather_rows_into(full, part)
    assert torch.equal(full, part)
    own = torch.zeros(5000, device=d)
    coll.reduce_scatter_rows(own, a[:5000].contiguous())
    assert torch.equal(own, a[:5000])
    idx = torch.randint(0, 1000, (300, 10), device=d, dtype=torch.int32)
    val = torch.randn((300, 10), device=d, generator=g)
    gi, gv = coll.all_gather_topk(idx, val)
    assert gi.shape == (1, 300, 10) and torch.equal(gi[0], idx) and torch.equal(gv[0], val)
    coll.close()


def test_user_sharded_step_through_the_abi_collectives(ctx):
    rs = np.random.RandomState(2)
    U, I, F, B = 400, 300, 64, 4096
    Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32)
    Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32)
    Bi = np.zeros(I, np.float32)
    coll = parallel.RcclAbiCollectives(ctx, 0, 1)
    be = parallel.HipUserShardBackend(ctx, Gu, Gi, Bi)
    tr = parallel.ShardedBprmfByUser(be, coll)
    orc = ob.BPRMFBatchOracle(Gu, Gi, Bi, 0.01, 0.1, 0.001)
    d = ctx.device
    for _ in range(3):
        u, i, j = (rs.randint(0, n, B).astype(np.int32) for n in (U, 40, I))
        tr.train_step(*(torch.from_numpy(x).to(d) for x in (u, i, j)), 0.01, 0.1, 0.001)
        orc.train_step((u, i, j))
    for name in ("Gu", "Gi", "Bi"):
        assert (np.abs(cpu(getattr(be.state, name)) - getattr(orc, name)) > 2e-5).mean() <= 2e-4
    coll.close()
