#!/usr/bin/env python
"""k_bpr_sample alone (profiles/r06_placement_probe.md, part 3): the positives' CSR and the sampler records are built ONCE; per trial
only the three 4 MB OUTPUT arrays are re-allocated (behind a pad of varying size) and 20 launches are timed; every fourth trial reuses
the arrays of the trial before (same addresses: does the time repeat?)."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd import ops  # noqa: E402
from elliot_amd.synthetic import zipf_csr_device  # noqa: E402

ctx = ops.get_context(0)
dev = ctx.device
U, I, B = 10_000_000, 1_000_000, 1 << 20
indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=4321)
pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
out = None
for trial in range(24):
    if trial % 4 != 3 or out is None:
        out = None
        torch.cuda.empty_cache()
        pad = torch.empty(((trial * 311) % 1500 + 1) << 20, dtype=torch.uint8, device=dev)
        out = tuple(torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3))
    for s in range(3):
        ops.bpr_sample(ctx, pos, B, seed=42, first_sample=s * B, out=out)
    torch.cuda.synchronize()
    ctx.timing(True)
    for s in range(20):
        ops.bpr_sample(ctx, pos, B, seed=42, first_sample=(3 + s) * B, out=out)
    torch.cuda.synchronize()
    ctx.timing(False)
    r = ctx.timing_report()
    print(json.dumps({"trial": trial, "reused": trial % 4 == 3, "sample_ms": round(r["k_bpr_sample"][1] / r["k_bpr_sample"][0], 4),
                      "va_out": [hex(t.data_ptr()) for t in out]}), flush=True)
    del pad
