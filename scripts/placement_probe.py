#!/usr/bin/env python
"""One process of the placement probe (scripts/placement_probe.sh): the headline BPR state (10M x 1M x 128, bench.py's data and
initial tables), cover batches, then --steps sequential train steps with per-kernel events; prints ONE JSON line: per-kernel mean
ms + the virtual addresses of every table the segment kernels touch (the physical placement is not visible from user space)."""
import argparse
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd import ops  # noqa: E402
from elliot_amd.pipeline import cover_batches  # noqa: E402
from elliot_amd.synthetic import zipf_csr_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--users", type=int, default=10_000_000)
ap.add_argument("--items", type=int, default=1_000_000)
ap.add_argument("--pad-mib", type=int, default=0, help="allocate (and keep) this much before the tables: shifts where they land")
ap.add_argument("--tag", default="")
a = ap.parse_args()
ctx = ops.get_context(0)
dev = ctx.device
U, I, F, B = a.users, a.items, 128, 1 << 20
pad = torch.empty(a.pad_mib << 20, dtype=torch.uint8, device=dev) if a.pad_mib else None
indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=4321)
pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
g = torch.Generator(device=dev)
g.manual_seed(42)
Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * (6.0 / (U + F)) ** 0.5
Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * (6.0 / (I + F)) ** 0.5
st = ops.BprmfDeviceState(ctx, Gu, Gi, torch.zeros(I, device=dev), optimizer="adam_tf_dense", replay="series")
del Gu, Gi
lr, l_w, l_b = 0.001, 0.1, 0.001
cover_batches(st, indptr, indices, U, 0, U, I, B, lr, l_w, l_b)
for s in range(3):
    t = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=s * B)
    st.train_step(*t, lr, l_w, l_b)
torch.cuda.synchronize()
series = []
rep = {}
for blk in range(max(1, a.steps // 10)):                     # per-kernel means of consecutive 10-step blocks: does a process FLIP?
    ctx.timing(True)
    for s in range(10):
        t = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=(3 + blk * 10 + s) * B)
        st.train_step(*t, lr, l_w, l_b)
    if blk == max(1, a.steps // 10) - 1:
        st.sync()
    torch.cuda.synchronize()
    ctx.timing(False)
    r = ctx.timing_report()
    series.append({n[6:]: round(v[1] / v[0], 3) for n, v in r.items() if n in ("k_bpr_item_seg", "k_bpr_user_seg", "k_bpr_sample")})
    for n, v in r.items():
        c = rep.get(n, (0, 0.0))
        rep[n] = (c[0] + v[0], c[1] + v[1])
tabs = {"Gu": st._Gu, "mGu": st.mGu, "vGu": st.vGu, "Gu_old": st.Gu_old, "Gi": st._Gi, "mGi": st.mGi, "vGi": st.vGi, "gGi": st.gGi,
        "Gi_last": st.Gi_last, "Gu_last": st.Gu_last, "ws": st._ws}
out = {"tag": a.tag, "ms": {n: round(v[1] / v[0], 4) for n, v in rep.items() if n.startswith("k_bpr")},
       "va": {n: hex(x.data_ptr()) for n, x in tabs.items() if x is not None},
       "va_mod_1g_mib": {n: (x.data_ptr() % (1 << 30)) >> 20 for n, x in tabs.items() if x is not None},
       "layout_gap": getattr(st, "layout_gap", None), "series": series}
print(json.dumps(out), flush=True)
