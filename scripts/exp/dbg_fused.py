import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from elliot_amd import ops
from elliot_amd.synthetic import zipf_csr
ctx = ops.get_context(0)
F, U, I, B = 128, 30000, 1500, 8192
rs = np.random.RandomState(F + U)
indptr, indices = zipf_csr(U, I, mean_log=2.0, sigma_log=0.9, dmin=1, dmax=150, seed=F)
pos = ops.DeviceCSR(indptr, indices, I, ctx.device)
Gu = rs.normal(scale=0.1, size=(U, F)).astype(np.float32); Gi = rs.normal(scale=0.1, size=(I, F)).astype(np.float32); Bi = rs.normal(scale=0.01, size=I).astype(np.float32)
lr, l_w, l_b = 0.01, 0.1, 0.001
a = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer="adam_tf_dense", compact_user_grads=True, fused_user_step=False)
b = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer="adam_tf_dense", compact_user_grads=True)
for s in range(3):
    t = ops.bpr_sample(ctx, pos, B, seed=7, first_sample=s * B)
    a.train_step(t[0], t[1], t[2], lr, l_w, l_b)
    b.train_step(t[0], t[1], t[2], lr, l_w, l_b)
    print("loss", a.pop_loss(), b.pop_loss())
    for name in ("Gu", "mGu", "vGu", "Gi", "Bi"):
        x, y = getattr(a, name), getattr(b, name)
        d = (x - y).abs()
        nz = torch.nonzero(d.reshape(x.shape[0], -1).amax(1) > 0).flatten()
        print(s, name, "max", float(d.max()), "rows differing", int(nz.numel()), nz[:8].tolist())
        if name == "mGu" and nz.numel():
            r = int(nz[0]); cnt = int((t[0] == r).sum())
            print("   row", r, "triplets of that user in the batch:", cnt, "a", x[r, :4].tolist(), "b", y[r, :4].tolist())
