#!/usr/bin/env python
"""Keras' dense Adam pass over theta / g / m / v of a 1M x 128 table (the probe that tracks the slow / fast mode of the item kernels,
placement_probe3.py) as a function of the DISTANCE between the arrays when they are carved from one allocation: --count 3 = theta, m, v in
the block and g elsewhere, 4 = all four in the block.  One JSON line per pass over the candidate distances."""
import argparse
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd import ops  # noqa: E402
from elliot_amd._lib import BprmfState, EL_OPT_ADAM_TF_DENSE, check  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--count", type=int, default=3)
ap.add_argument("--passes", type=int, default=2)
ap.add_argument("--step-kib", type=int, default=256)
ap.add_argument("--max-mib", type=float, default=8.0)
a = ap.parse_args()
ctx = ops.get_context(0)
dev = ctx.device
I, F = 1_000_000, 128
dm = [torch.zeros((64, F), dtype=torch.float32, device=dev) for _ in range(4)] + [torch.zeros(64, dtype=torch.float32, device=dev) for _ in range(4)]
gsep = torch.zeros((I, F), dtype=torch.float32, device=dev)


def probe(th, g, m, v):
    c = BprmfState(Gu=th.data_ptr(), gGu=g.data_ptr(), mGu=m.data_ptr(), vGu=v.data_ptr(),
                   Gi=dm[0].data_ptr(), gGi=dm[1].data_ptr(), mGi=dm[2].data_ptr(), vGi=dm[3].data_ptr(),
                   Bi=dm[4].data_ptr(), gBi=dm[5].data_ptr(), mBi=dm[6].data_ptr(), vBi=dm[7].data_ptr(), tGu=None, tGi=None, tBi=None, U=I, I=64, F=F)
    for it in range(2):
        check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(c), 0.001, EL_OPT_ADAM_TF_DENSE, it + 1, 0.001), "apply")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for it in range(8):
        check(ctx.lib.el_bprmf_apply(ctx.handle, ctx.stream(), C.byref(c), 0.001, EL_OPT_ADAM_TF_DENSE, it + 3, 0.001), "apply")
    e1.record()
    e1.synchronize()
    return e0.elapsed_time(e1) / 8


for p in range(a.passes):
    res = {}
    kib = 0
    while kib <= a.max_mib * 1024:
        tabs, big = ops._strided_tables(I, F, a.count, kib << 10, dev)
        ms = probe(tabs[0], gsep, tabs[1], tabs[2]) if a.count == 3 else probe(*tabs)
        res[kib] = round(ms, 3)
        del tabs, big
        kib += a.step_kib
    sep = [torch.zeros((I, F), dtype=torch.float32, device=dev) for _ in range(3)]
    print(json.dumps({"pass": p, "count": a.count, "separate_allocations": round(probe(sep[0], gsep, sep[1], sep[2]), 3), "by_gap_kib": res}), flush=True)
    del sep
    torch.cuda.empty_cache()
