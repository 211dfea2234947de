#!/usr/bin/env python
"""el_gemm_f32 against torch fp64 matmul on the device, model-sized shapes (quick correctness probe).  usage: gemm_check.py [M,N,K,tA,tB ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd import ops  # noqa: E402

ctx = ops.get_context(0)
dev = ctx.device
g = torch.Generator(device=dev)
g.manual_seed(0)
shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]] or [
    (65536, 256, 512, 0, 1), (65536, 512, 256, 0, 1), (65536, 256, 128, 0, 1), (256, 512, 65536, 1, 0), (65536, 512, 256, 0, 0),
    (65536, 128, 256, 0, 0), (512, 26744, 600, 0, 0), (600, 26744, 512, 1, 0), (512, 600, 26744, 0, 1), (4096, 4096, 4096, 0, 0),
    (1000, 1000, 1000, 1, 1), (131072, 512, 256, 0, 1)]
for (M, N, K, tA, tB) in shapes:
    A = torch.randn((K, M) if tA else (M, K), device=dev, generator=g)
    B = torch.randn((N, K) if tB else (K, N), device=dev, generator=g)
    bias = torch.randn(N, device=dev, generator=g)
    C = ops.gemm(ctx, A, B, bool(tA), bool(tB), bias=bias, act="relu")
    ref = torch.relu((A.t() if tA else A).double() @ (B.t() if tB else B).double() + bias.double())
    err = (C.double() - ref).abs()
    bad = int((err > 1e-3 * (K ** 0.5)).sum())
    r, c = divmod(int(err.argmax()), N)
    print(f"M={M} N={N} K={K} tA={tA} tB={tB}: max err {float(err.max()):.3e} at ({r},{c}), entries off {bad}")
