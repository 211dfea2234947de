import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())
os.environ["EL_NMF_SCREEN_MAXFRAC"] = "1.0"
from elliot_amd import ops
from tests.test_gpu_nmf_score import _weights
from tests.gpu_util import random_excl
ctx = ops.Context(0)
F, units, k, I, U = 16, [72, 40, 16], 50, 5000, 48
w = _weights(U, I, F, seed=F + k, units=units)
st = ops.NmfDeviceState(ctx, w, max_batch=1024)
rs = np.random.RandomState(3)
ip, ix = random_excl(rs, U, I, 0, 60)
excl = ops.DeviceCSR(ip, ix, I, ctx.device)
for rep in range(3):
    for name, kw in (("excl", {"excl": excl}), ("none", {})):
        ri, rv = st.score_topk_logits(0, U, k, screen=False, **kw)
        gi, gv = st.score_topk_logits(0, U, k, screen=True, **kw)
        bad = (gi != ri)
        rows = torch.nonzero(bad.any(1)).flatten().tolist()
        print(rep, name, "mismatch entries", int(bad.sum()), "rows", rows[:10], st.screen_stats())
        for r in rows[:2]:
            c = torch.nonzero(bad[r]).flatten()[0].item()
            print("  row", r, "first col", c, "ref", ri[r, c:c+4].tolist(), rv[r, c:c+4].tolist(), "got", gi[r, c:c+4].tolist(), gv[r, c:c+4].tolist())
            missing = set(ri[r].tolist()) - set(gi[r].tolist())
            print("  missing items", sorted(missing)[:10])
