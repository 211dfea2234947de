#!/usr/bin/env python
"""Micro-benchmarks of single hot kernels (for rocprofv3 PMC passes and A/B of kernel variants).
usage: python scripts/mb.py topk|adam|train [--users N --items N --factors F --k K --iters N --algo ...]"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from elliot_amd import ops  # noqa: E402
from elliot_amd.synthetic import zipf_csr_device  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("what", choices=["topk", "train", "vae", "gemm", "pwmf", "nmf", "nmfscore"])
    ap.add_argument("--score-users", type=int, default=128, help="nmfscore: users per el_nmf_score_topk call")
    ap.add_argument("--users", type=int, default=131072)
    ap.add_argument("--items", type=int, default=100000)
    ap.add_argument("--factors", type=int, default=128)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--iters", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--algo", default="mfma")
    ap.add_argument("--opt", default="adam_tf_dense")
    ap.add_argument("--no-excl", action="store_true")
    ap.add_argument("--train-steps", type=int, default=0, help="topk: BPR steps (B = --batch, TF-dense Adam) on the tables before scoring -- "
                    "the candidate windows of the screened kernels depend on the norms a trained model has")
    ap.add_argument("--sweep", default="", help="topk: comma list of stride:kA settings (options screen_stride / screen_ka) to time after the default")
    ap.add_argument("--model", default="FunkSVD", choices=["MF", "PMF", "FunkSVD", "LogisticMF", "NeuMF", "GMF"])
    ap.add_argument("--shape", action="append", default=None, help="gemm: M,N,K,tA,tB (repeatable; default: the model shapes)")
    a = ap.parse_args()
    ctx = ops.get_context(0)
    dev = ctx.device
    g = torch.Generator(device=dev)
    g.manual_seed(1)
    if a.what == "gemm":
        ctx.timing(True)
        B2 = 262144
        shapes = [tuple(int(x) for x in sh.split(",")) for sh in a.shape] if a.shape else None
        for (M, N, K, tA, tB) in shapes or [(512, 26744, 600, 0, 0), (600, 26744, 512, 1, 0), (512, 600, 26744, 0, 1),      # Mult-VAE, ML-20M shape
                                  (512, 400, 600, 0, 0), (512, 600, 200, 0, 0), (600, 400, 512, 1, 0), (512, 600, 400, 0, 1),
                                  (B2, 512, 256, 0, 0), (B2, 256, 512, 0, 0), (B2, 128, 256, 0, 0),                # NeuMF tower fwd
                                  (B2, 256, 512, 0, 1), (B2, 512, 256, 0, 1), (256, 512, B2, 1, 0), (512, 256, B2, 1, 0),
                                  (4096, 4096, 4096, 0, 0), (8192, 8192, 1024, 0, 1)]:
            A = torch.randn((K, M) if tA else (M, K), device=dev)
            Bm = torch.randn((N, K) if tB else (K, N), device=dev)
            out = torch.empty((M, N), device=dev)
            for _ in range(3):                                   # warm-up: code load, clocks
                ops.gemm(ctx, A, Bm, bool(tA), bool(tB), out=out)
            torch.cuda.synchronize()
            ctx.timing_report()
            for _ in range(a.iters if a.iters > 3 else 10):
                ops.gemm(ctx, A, Bm, bool(tA), bool(tB), out=out)
            torch.cuda.synchronize()
            rep = ctx.timing_report()
            ms = sum(v[1] for v in rep.values()) / (a.iters if a.iters > 3 else 10)
            n_it = a.iters if a.iters > 3 else 10
            parts = " ".join(f"{k.replace('k_gemm_', '')}={v[1] / n_it:.3f}" for k, v in rep.items())
            print(f"gemm M={M} N={N} K={K} tA={tA} tB={tB}: {ms:.3f} ms  {2.0 * M * N * K / ms / 1e9:.1f} TFLOP/s  [{parts}]")
        return
    if a.what == "vae":
        import numpy as np
        U, I, H, L, B = 138493, 26744, 600, 200, 512
        ip, ix = zipf_csr_device(U, I, dev, mean_log=4.5, sigma_log=1.0, dmin=20, dmax=3000, seed=5)
        csr = ops.DeviceCSR.from_tensors(ip, ix, I)
        print("interactions", csr.nnz)
        rs = np.random.RandomState(1)
        gl = lambda a, b: (rs.standard_normal((a, b)) * np.sqrt(2.0 / (a + b))).astype(np.float32)
        z = lambda n: np.zeros(n, np.float32)
        st = ops.VaeDeviceState(ctx, {"W1": gl(I, H), "b1": z(H), "Wm": gl(H, L), "bm": z(L), "Wv": gl(H, L), "bv": z(L),
                                      "W3": gl(L, H), "b3": z(H), "W4": gl(H, I), "b4": z(I)}, max_batch=B)
        rows_all = torch.randperm(U, device=dev, generator=g).to(torch.int32)
        ctx.timing(True)
        steps = a.iters
        for it in range(steps + 2):
            if it == 2:
                torch.cuda.synchronize(); ctx.timing_report(); t0 = time.perf_counter()
            rows = rows_all[it * B:(it + 1) * B].contiguous()
            eps = torch.randn((B, L), device=dev)
            st.train_step(csr, rows, 0.001, 0.1, eps=eps)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        rep = ctx.timing_report()
        tot = 0
        for n, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
            print(f"{n}: {ms / steps:.4f} ms/step ({c // steps} launches/step)")
            tot += ms / steps
        flops = B * (2.0 * (H * 2 * L + L * H + H * I) * 3)   # dense GEMM flops fwd + 2x bwd (first layer is sparse)
        print(f"kernel total {tot:.3f} ms/step, wall {dt / steps * 1e3:.3f} ms/step -> {B / (dt / steps):.0f} users/s, "
              f"{flops / (tot * 1e-3) / 1e12:.1f} TFLOP/s on the dense part")
        return
    U, I, F = a.users, a.items, a.factors
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.01
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.01
    Bi = torch.zeros(I, device=dev)
    ip, ix = zipf_csr_device(U, I, dev, mean_log=3.9, seed=5)
    pos = ops.DeviceCSR.from_tensors(ip, ix, I)
    if a.what in ("nmf", "nmfscore"):
        import numpy as np
        rs = np.random.RandomState(3)
        gu = lambda a, b: rs.uniform(-np.sqrt(6.0 / (a + b)), np.sqrt(6.0 / (a + b)), size=(a, b)).astype(np.float32)
        w = {"Umf": gu(U, F), "Imf": gu(I, F)}
        if a.model != "GMF":
            units = [4 * F, 2 * F, F]
            w.update({"Umlp": gu(U, F), "Imlp": gu(I, F), "W": [], "b": []})
            kin = 2 * F
            for n_out in units:
                w["W"].append(gu(kin, n_out))
                w["b"].append(np.zeros(n_out, np.float32))
                kin = n_out
            w["hw"], w["hb"] = gu(F + units[-1], 1)[:, 0].copy(), np.zeros(1, np.float32)
        else:
            w["hw"] = gu(F, 1)[:, 0].copy()
        st = ops.NmfDeviceState(ctx, w, max_batch=a.batch if a.what == "nmf" else 4096)
        if a.what == "nmfscore":
            nu = a.score_users
            st.recommend(0, nu, a.k, excl=pos)
            torch.cuda.synchronize()
            ctx.timing(True)
            t0 = time.perf_counter()
            for it in range(a.iters):
                st.recommend((it + 1) * nu, (it + 2) * nu, a.k, excl=pos, items_unchanged=True)
            torch.cuda.synchronize()
            print(f"wall (with per-kernel timing events): {(time.perf_counter() - t0) * 1e3 / a.iters:.3f} ms/call of {nu} users")
            for n, (c, ms) in sorted(ctx.timing_report().items(), key=lambda kv: -kv[1][1]):
                print(f"{n}: {ms / a.iters:.4f} ms/call ({c // a.iters} launches/call)")
            pairs, fb = st.screen_stats()
            print(f"screen: exact pairs of the last call {pairs} = {pairs / max(nu * I, 1):.5f} of the block, fell back: {fb}")
            return
        ctx.timing(True)
        for it in range(a.iters + 2):
            if it == 2:
                torch.cuda.synchronize(); ctx.timing_report(); t0 = time.perf_counter()
            u, i, y = ops.pointwise_sample(ctx, pos, a.batch, seed=3, first_sample=it * a.batch)
            st.train_step(u, i, y, 0.001)
        st.sync()                                        # deferred decay: the postponed row updates belong to the timed steps
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        rep, tot = ctx.timing_report(), 0
        for n, (c, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
            print(f"{n}: {ms / a.iters:.4f} ms/step ({c // a.iters} launches/step)")
            tot += ms / a.iters
        units = [4 * F, 2 * F, F]
        flops = a.batch * 2.0 * (2 * F * units[0] + units[0] * units[1] + units[1] * units[2]) * 3
        print(f"NeuMF F={F} B={a.batch}: kernels {tot:.3f} ms/step, wall {dt * 1e3:.3f} ms/step -> {a.batch / dt / 1e6:.1f} M samples/s, "
              f"MLP {flops / (tot * 1e-3) / 1e12:.1f} TFLOP/s over the step")
        return
    if a.what == "pwmf":
        kind, bias, opt = {"MF": ("mse", False, "adam"), "PMF": ("mse_sigmoid", False, "adam"), "FunkSVD": ("mse", True, "adam"),
                           "LogisticMF": ("logistic", True, "adagrad")}[a.model]
        st = ops.PwmfDeviceState(ctx, Gu.cpu().numpy(), Gi.cpu().numpy(), torch.zeros(U).numpy() if bias else None,
                                 torch.zeros(I).numpy() if bias else None, kind=kind, optimizer=opt, alpha=0.5, l_w=0.01)
        ctx.timing(True)
        for it in range(a.iters + 2):
            if it == 2:
                torch.cuda.synchronize(); ctx.timing_report(); t0 = time.perf_counter()
            u, i, y = ops.pointwise_sample(ctx, pos, a.batch, seed=3, first_sample=it * a.batch)
            st.train_step(u, i, y, 0.001, side=("items" if it % 2 else "users") if kind == "logistic" else "both")
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        rep, tot = ctx.timing_report(), 0
        for n, (c, ms) in rep.items():
            print(f"{n}: {ms / a.iters:.4f} ms/step ({c} launches)")
            tot += ms / a.iters
        print(f"{a.model}: kernels {tot:.4f} ms/step, wall {dt * 1e3:.4f} ms/step -> {a.batch / dt / 1e6:.1f} M samples/s")
        idx, val = st.recommend(0, min(U, 131072), a.k, excl=pos)
        torch.cuda.synchronize(); ctx.timing_report()
        t0 = time.perf_counter()
        for _ in range(a.iters):
            st.recommend(0, min(U, 131072), a.k, excl=pos)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.iters
        print(f"{a.model}: recommend {min(U, 131072)} users in {dt * 1e3:.3f} ms -> {min(U, 131072) / dt / 1e6:.2f} M users/s")
        return
    if a.what == "topk":
        if a.train_steps > 0:
            lim_u, lim_i = (6.0 / (1_000_000 + F)) ** 0.5, (6.0 / (I + F)) ** 0.5       # GlorotUniform of the bench's 1M x I tables
            Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * lim_u
            Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * lim_i
            stt = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer="adam_tf_dense")
            for it in range(a.train_steps):
                u, i, j = ops.bpr_sample(ctx, pos, a.batch, seed=42, first_sample=it * a.batch)
                stt.train_step(u, i, j, 0.001, 0.1, 0.001)
            Gu, Gi, Bi = stt.Gu, stt.Gi, stt.Bi
        settings = [("default", None, None)] + [(x, x.split(":")[0], x.split(":")[1]) for x in a.sweep.split(",") if x]
        for name, sd, ka in settings:
            for opt, val in (("screen_stride", sd), ("screen_ka", ka)):
                ctx.set_option(opt, 0 if val is None else int(val))
            ctx.timing(False)
            ops.score_topk(ctx, Gu, Gi, Bi, 0, U, a.k, excl=None if a.no_excl else pos, algo=a.algo)   # warm-up: code load, workspace
            torch.cuda.synchronize()
            ctx.timing(True)
            t0 = time.perf_counter()
            for _ in range(a.iters):
                ops.score_topk(ctx, Gu, Gi, Bi, 0, U, a.k, excl=None if a.no_excl else pos, algo=a.algo, items_unchanged=True)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) / a.iters * 1e3
            rep = ctx.timing_report()
            tot = sum(ms / a.iters for _, (c, ms) in rep.items())
            print(f"[{name}] kernels {tot:.3f} ms/call, wall (with events) {wall:.3f} ms -> {U / tot * 1e3 / 1e6:.2f} M users/s")
            for n, (c, ms) in rep.items():
                per = ms / a.iters
                if per > 0.02:
                    print(f"    {n}: {per:.3f} ms/call  {2.0 * U * I * F / per / 1e9:.1f} TFLOP/s")
        return
    st = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer=a.opt)
    algo = a.algo if a.algo in ("auto", "atomic", "sorted") else "auto"
    warm = 3
    for it in range(warm + a.iters):
        if it == warm:
            st.sync()
            torch.cuda.synchronize()
            ctx.timing(True)
            ctx.timing_report()
            t0 = time.perf_counter()
        u, i, j = ops.bpr_sample(ctx, pos, a.batch, seed=3, first_sample=it * a.batch)
        st.train_step(u, i, j, 0.001, 0.1, 0.001, algo=algo)
    st.sync()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / a.iters * 1e3
    rep = ctx.timing_report()
    tot = 0
    for n, (c, ms) in rep.items():
        print(f"{n}: {ms / a.iters:.4f} ms/step ({c} launches)")
        tot += ms / a.iters
    print(f"total {tot:.4f} ms/step (wall with events {wall:.4f}) -> {a.batch / tot * 1e3 / 1e6:.1f} M pairs/s; "
          f"deferred users={st.deferred} item_fused={st.item_fused} item_deferred={st.item_deferred}")

if __name__ == "__main__":
    main()
