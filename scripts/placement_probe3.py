#!/usr/bin/env python
"""Does the slow / fast mode of the BPR segment kernels (profiles/r06_placement_probe.md) belong to the PROCESS or to an ALLOCATION?
One process, several trials: the headline state is built, timed (10 steps, per-kernel events) and freed again (memory returned to the
driver: torch.cuda.empty_cache()); trials 0..2 keep the positives' CSR and rebuild only the tables, trials 3..5 rebuild everything.
Prints one JSON line per trial."""
import argparse
import gc
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd import ops  # noqa: E402
from elliot_amd.pipeline import cover_batches  # noqa: E402
from elliot_amd.synthetic import zipf_csr_device  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--trials", type=int, default=6)
ap.add_argument("--tag", default="")
ap.add_argument("--gaps", default="", help="comma list of byte distances: item theta / m / v from one allocation that far apart (one per trial)")
a = ap.parse_args()
ctx = ops.get_context(0)
dev = ctx.device
U, I, F, B = 10_000_000, 1_000_000, 128, 1 << 20
lr, l_w, l_b = 0.001, 0.1, 0.001
csr = None
gaps = [int(x) for x in a.gaps.split(",")] if a.gaps else None
for trial in range(a.trials if gaps is None else len(gaps)):
    if gaps is not None:
        os.environ["EL_ITEM_GAP"] = str(gaps[trial])
    if csr is None or trial >= 3:
        csr = None
        gc.collect()
        torch.cuda.empty_cache()
        indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=4321)
        csr = (indptr, indices, ops.DeviceCSR.from_tensors(indptr, indices, I))
    indptr, indices, pos = csr
    pad = torch.empty(((trial * 733) % 2048) << 20, dtype=torch.uint8, device=dev) if trial else None     # shifts what the tables get
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * (6.0 / (U + F)) ** 0.5
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * (6.0 / (I + F)) ** 0.5
    st = ops.BprmfDeviceState(ctx, Gu, Gi, torch.zeros(I, device=dev), optimizer="adam_tf_dense", replay="series")
    del Gu, Gi
    cover_batches(st, indptr, indices, U, 0, U, I, B, lr, l_w, l_b)
    for s in range(3):
        st.train_step(*ops.bpr_sample(ctx, pos, B, seed=42, first_sample=s * B), lr, l_w, l_b)
    torch.cuda.synchronize()
    t_begin = __import__("time").time()
    ctx.timing(True)
    for s in range(10):
        st.train_step(*ops.bpr_sample(ctx, pos, B, seed=42, first_sample=(3 + s) * B), lr, l_w, l_b)
    torch.cuda.synchronize()
    ctx.timing(False)
    r = ctx.timing_report()
    # a streaming probe on the SAME item arrays: Keras' dense Adam pass over theta / g / m / v (what ops.tune_table_layout times)
    import ctypes as C
    from elliot_amd._lib import BprmfState, EL_OPT_ADAM_TF_DENSE, check
    dm = [torch.zeros((64, F), dtype=torch.float32, device=dev) for _ in range(4)] + [torch.zeros(64, dtype=torch.float32, device=dev) for _ in range(4)]
    c = BprmfState(Gu=st._Gi.data_ptr(), gGu=st.gGi.data_ptr(), mGu=st.mGi.data_ptr(), vGu=st.vGi.data_ptr(),
                   Gi=dm[0].data_ptr(), gGi=dm[1].data_ptr(), mGi=dm[2].data_ptr(), vGi=dm[3].data_ptr(),
                   Bi=dm[4].data_ptr(), gBi=dm[5].data_ptr(), mBi=dm[6].data_ptr(), vBi=dm[7].data_ptr(), tGu=None, tGi=None, tBi=None, U=I, I=64, F=F)
    st.sync()
    for it in range(2):
        check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(c), 0.001, EL_OPT_ADAM_TF_DENSE, it + 1, 0.001), "apply")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(8):
        check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(c), 0.001, EL_OPT_ADAM_TF_DENSE, it + 3, 0.001), "apply")
    e1.record()
    e1.synchronize()
    adam_ms = e0.elapsed_time(e1) / 8
    print(json.dumps({"tag": a.tag, "trial": trial, "adam_probe_ms": round(adam_ms, 4), "t0": round(t_begin, 3), "t1": round(__import__("time").time(), 3), "rebuilt": "all" if trial >= 3 or trial == 0 else "tables", "item_placement": getattr(st, "item_placement", None),
                      "ms": {n[6:]: round(v[1] / v[0], 3) for n, v in r.items() if n in ("k_bpr_item_seg", "k_bpr_user_seg", "k_bpr_sample", "k_bpr_flush_items")},
                      "va_Gi": hex(st._Gi.data_ptr()), "va_Gu": hex(st._Gu.data_ptr())}), flush=True)
    del st, pad
    gc.collect()
    torch.cuda.empty_cache()
