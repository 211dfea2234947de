#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: BPR-MF positive-pairs/sec + full-catalog top-k users/sec.

One JSON line on rank 0.  N=1 workload = BASELINE.json configs[1]: BPRMF d=128 on synthetic
1M users x 100K items (SURVEY.md 8d "S-1M"), inputs resident in HBM before the timed region.

  step (train) : sample B triplets on the device -> gather -> BPR loss -> Adam (TF-dense semantics, the
                 reference's BPRMF_batch_model.train_step) for one batch of B = --batch triplets
  step (top-k) : fused score + masked top-k for one block of --topk-block users against the full catalogue

`value` is the training throughput (pairs/s); the top-k leg is reported under "topk".  Both legs carry a
roofline object for their dominant kernel (duration from hipEvents recorded inside the library on the
launch stream).  `cpu_baseline` times the CPU oracle (a port of the reference path; /root/reference is
absent on the GPU box) on a bounded sample, rank 0 / N=1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

from elliot_amd import ops, parallel  # noqa: E402
from elliot_amd.synthetic import zipf_csr_device  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--users", type=int, default=1_000_000)
    ap.add_argument("--items", type=int, default=100_000)
    ap.add_argument("--factors", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--topk-block", type=int, default=131072)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--opt", default="adam_tf_dense", choices=["adam_tf_dense", "adam_lazy", "sgd"])
    ap.add_argument("--train-algo", default="auto", choices=["auto", "atomic", "sorted"])
    ap.add_argument("--topk-algo", default="auto", choices=["auto", "screen", "mfma", "simple"])
    ap.add_argument("--shard", default="user", choices=["user", "item"],
                    help="N > 1: shard the USER table (item table replicated, all-reduce of item gradients) or the ITEM table "
                         "(north_star's formulation: user table replicated, user-gradient exchange per --exchange)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rows", "dense"],
                    help="N > 1: how user-row gradients travel (parallel.pick_exchange)")
    ap.add_argument("--topk-shard", default="user", choices=["user", "item"],
                    help="N > 1: users are independent units (no collective) / north_star's item shards + all-gather of partial top-k")
    ap.add_argument("--prefetch", action="store_true",
                    help="draw the triplets of step t+1 on a side stream during step t (measured: no gain, the step is HBM-bound)")
    ap.add_argument("--force-sharded", action="store_true", help="run the N > 1 code path even with one rank (API check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-topk-users", type=int, default=640)
    return ap.parse_args()


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if os.environ.get("EL_BENCH_SHARED_GPU") == "1":
            # development check of the N > 1 control flow on a ONE-GPU box: every rank on cuda:0, gloo instead of RCCL
            # (RCCL refuses two ranks on one device).  The numbers of such a run mean nothing.
            local = 0
            torch.cuda.set_device(0)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    elif args.force_sharded:
        # one rank, but through RCCL and the N > 1 code path: an API check of the collectives on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ["EL_FORCE_COLLECTIVES"] = "1"
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
    return world, rank, local


class PrefetchSampler:
    """BPR triplets of step t+1 are drawn on a side stream while step t trains: the sampler (custom_sampler.py:31-46) does not
    depend on the model, so a training loop can always run it one batch ahead.  Two triplet buffers, events both ways."""

    def __init__(self, ctx, pos, B, seed, enabled=True):
        dev = ctx.device
        self.ctx, self.pos, self.B, self.seed, self.enabled = ctx, pos, B, seed, enabled
        self.bufs = [tuple(torch.empty(B, dtype=torch.int32, device=dev) for _ in range(3)) for _ in range(2)]
        self.side = torch.cuda.Stream(device=dev)
        self.ready = [torch.cuda.Event(), torch.cuda.Event()]
        self.free = [None, None]
        self.cur, self.ctr = 0, 0
        if enabled:
            self._issue(0)

    def _draw(self, b):
        ops.bpr_sample(self.ctx, self.pos, self.B, seed=self.seed, first_sample=self.ctr, out=self.bufs[b])
        self.ctr += self.B

    def _issue(self, b):
        self.side.wait_stream(torch.cuda.current_stream())           # (first use / anything the caller queued before)
        with torch.cuda.stream(self.side):
            if self.free[b] is not None:
                self.side.wait_event(self.free[b])                    # the training step that read this buffer is done
            self._draw(b)
            self.ready[b].record(self.side)

    def next(self):
        """Triplets of this step (valid on the current stream)."""
        b = self.cur
        if not self.enabled:
            self._draw(b)
            return self.bufs[b], b
        torch.cuda.current_stream().wait_event(self.ready[b])
        self.cur ^= 1
        self._issue(self.cur)                                         # next batch: overlaps this step's kernels
        return self.bufs[b], b

    def release(self, b):
        if self.enabled:
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            self.free[b] = ev


def barrier(world):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    torch.cuda.synchronize()


def max_over_ranks(x, world, device):
    if world == 1:
        return x
    import torch.distributed as dist
    t = torch.tensor([x], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def cpu_baseline(args, host):
    """CPU oracle ("port") timed on this host: a bounded sample of the same workload."""
    from oracle import bprmf_batch as ob
    from oracle import cref
    out = {"kind": "port", "cores": 1}
    # --- train: one TF-semantics step (NumPy fp32) at the GPU workload's shapes, smaller batch
    Bc = min(args.batch, 1 << 16)
    rs = np.random.RandomState(0)
    u = rs.randint(0, args.users, Bc)
    i = rs.randint(0, args.items, Bc)
    j = rs.randint(0, args.items, Bc)
    orc = ob.BPRMFBatchOracle(host["Gu"], host["Gi"], host["Bi"], 0.001, 0.1, 0.001, optimizer=args.opt)
    t0 = time.perf_counter()
    nsteps = 0
    while time.perf_counter() - t0 < 10.0:                 # ~10 s of CPU work
        orc.train_step((u, i, j))
        nsteps += 1
    dt = time.perf_counter() - t0
    out["value"] = Bc * nsteps / dt
    out["unit"] = "pairs/s"
    out["sample"] = (f"oracle/bprmf_batch.py train_step ({args.opt}), {nsteps} steps, B={Bc}, U={args.users}, "
                     f"I={args.items}, F={args.factors}, NumPy fp32 single thread; {dt:.2f}s")
    # --- top-k: C oracle (fmaf chain + selection) on a few users against the full catalogue
    nu = args.cpu_topk_users
    t0 = time.perf_counter()
    cref.score_topk_f32(host["Gu"][:nu], host["Gi"], host["Bi"], 0, nu, args.k,
                        excl=(host["indptr"][:nu + 1], host["indices"][:int(host["indptr"][nu])]))
    dt = time.perf_counter() - t0
    out["topk"] = {"value": nu / dt, "unit": "users/s", "cores": 1, "kind": "port",
                   "sample": f"oracle/c/el_oracle.c orc_score_topk_f32, {nu} users x {args.items} items, F={args.factors}, "
                             f"k={args.k}; {dt:.2f}s"}
    return out


def main():
    args = parse()
    world, rank, local = dist_setup(args)
    if world != args.gpus and not (world == 1 and args.gpus == 1):
        if rank == 0:
            print(f"warning: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
    ctx = ops.get_context(local)
    dev = ctx.device
    torch.cuda.set_device(dev)
    U, I, F, B, k = args.users, args.items, args.factors, args.batch, args.k

    # ---------------- synthetic inputs, resident in HBM -------------------------------------------
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=1234)
    pos = ops.DeviceCSR.from_tensors(indptr, indices, I)
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    lim_u, lim_i = (6.0 / (U + F)) ** 0.5, (6.0 / (I + F)) ** 0.5           # GlorotUniform (BPRMF_batch_model.py:39-42)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * lim_u
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * lim_i
    Bi = torch.zeros(I, device=dev)

    # item shard of this rank (north_star: tables shard by item; N=1 -> the whole catalogue)
    lo, hi = parallel.item_range(I, rank, world)
    lr, l_w, l_b = 0.001, 0.1, 0.001                                          # BPRMF_batch.py:66-71 defaults
    sample_ctr = [0]
    finish_train = None
    exchange_used = [None]
    coll = parallel._Collectives()
    if world == 1 and not args.force_sharded:
        st = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer=args.opt)
        pos_train = pos

        sampler = PrefetchSampler(ctx, pos, B, 42, enabled=args.prefetch)

        def train_step():
            t, b = sampler.next()
            st.train_step(t[0], t[1], t[2], lr, l_w, l_b, algo=args.train_algo)
            sampler.release(b)

        pop_loss = st.pop_loss
    elif args.shard == "user":
        # USER shards: the rank owns the user rows [ulo, uhi) and a replica of the item table; B triplets per rank for its
        # own users (items: the whole catalogue, the reference's sampling distribution), all-reduce of the item gradients
        ulo, uhi = parallel.user_range(U, rank, world)
        uip = (indptr[ulo:uhi + 1] - indptr[ulo]).contiguous()
        uix = indices[int(indptr[ulo]):int(indptr[uhi])].contiguous()
        pos_train = ops.DeviceCSR.from_tensors(uip, uix, I)
        be = parallel.HipUserShardBackend(ctx, Gu[ulo:uhi], Gi, Bi, optimizer=args.opt)
        trainer = parallel.ShardedBprmfByUser(be, coll)
        st = be.state
        exchange_used[0] = "user"
        # the triplets of step t+1 are drawn AND sorted while step t's all-reduce is in flight (neither reads the model):
        # queued after the collective was issued, before its result is waited for
        drawn = [0]

        def draw():
            t = ops.bpr_sample(ctx, pos_train, B, seed=42 + rank, first_sample=drawn[0])
            drawn[0] += B
            be.presort(*t)                                           # ... and ordered (prep + radix sort read only the triplets)
            return t

        nxt = [draw()]

        def train_step():
            t = nxt[0]
            trainer.train_step(t[0], t[1], t[2], lr, l_w, l_b, overlap=lambda: nxt.__setitem__(0, draw()), presorted=True)

        pop_loss = trainer.pop_loss
    else:
        # item-sharded training: B triplets PER RANK with positive and negative inside the rank's shard, all-gather of
        # the per-triplet user-gradient rows, identical user-table replicas (elliot_amd/parallel.py)
        sip, six = parallel.shard_csr(indptr, indices, lo, hi)
        pos_train = ops.DeviceCSR.from_tensors(sip, six, hi - lo)
        exchange = args.exchange if args.exchange != "auto" else parallel.pick_exchange(U, B, world)
        exchange_used[0] = exchange
        if exchange == "dense":
            # reduce-scatter of the dense user-gradient table, optimiser on the owned user rows, all-gather of the rows
            be = parallel.HipDenseBackend(ctx, Gu, Gi[lo:hi].contiguous(), Bi[lo:hi].contiguous(), rank, world, optimizer=args.opt)
            trainer = parallel.ShardedBprmfDense(be, coll)
        else:
            be = parallel.HipBackend(ctx, Gu, Gi[lo:hi].contiguous(), Bi[lo:hi].contiguous(), optimizer=args.opt)
            trainer = parallel.ShardedBprmf(be, coll)
        st = be.state

        sampler = PrefetchSampler(ctx, pos_train, B, 42 + rank, enabled=args.prefetch)

        def train_step():
            t, b = sampler.next()
            trainer.train_step(t[0], t[1], t[2], lr, l_w, l_b)
            sampler.release(b)

        pop_loss = trainer.pop_loss
        finish_train = getattr(trainer, "finish", None)
    del Gu, Gi, Bi

    Ub = min(args.topk_block, U)
    n_blocks = max(1, U // Ub)
    blk = [0]

    sharded = world > 1 or args.force_sharded
    user_sharded = sharded and args.shard == "user"
    topk_by_user = sharded and args.topk_shard == "user" and not user_sharded
    full_items = {}

    def prepare_topk():
        """After training, before the evaluation: with user-sharded top-k the item table (sharded for training) is
        all-gathered ONCE per evaluation (I F 4 bytes); the per-block work then has no collective."""
        if topk_by_user:
            if finish_train:
                finish_train()
            full_items["Gi"], full_items["Bi"] = parallel.gather_item_table(coll, st.Gi, st.Bi, I)

    if user_sharded:
        # the rank owns its users' rows and a replica of the item table: top-k of ITS users, no collective at all
        n_local = st.U
        Ub = min(Ub, n_local)
        n_blocks = max(1, n_local // Ub)

        def topk_step():
            s = (blk[0] % n_blocks) * Ub
            blk[0] += 1
            ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s, s + Ub, k, excl=pos_train, algo=args.topk_algo,
                           items_unchanged=s > 0)                  # block 0 of every pass over the users derives the item image
    elif topk_by_user:
        # users are independent units: each rank scores ITS blocks of users against the whole catalogue
        def topk_step():
            s = ((blk[0] * world + rank) % n_blocks) * Ub
            blk[0] += 1
            ops.score_topk(ctx, st.Gu, full_items["Gi"], full_items["Bi"], s, s + Ub, k, excl=pos, algo=args.topk_algo,
                           items_unchanged=blk[0] > 1 and (blk[0] - 1) % n_blocks != 0)
    else:
        def topk_step():
            s = (blk[0] % n_blocks) * Ub
            blk[0] += 1
            parallel.sharded_topk(ctx, coll, st.Gu, st.Gi, st.Bi, lo, s, s + Ub, k, excl=pos, algo=args.topk_algo,
                                  items_unchanged=s > 0)           # block 0 of every pass over the users derives the item image

    def timed(fn, warmup, steps, finish=None):
        for _ in range(warmup):
            fn()
        if finish:
            finish()
        barrier(world)
        ctx.timing(True)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        if finish:
            finish()                                                 # a collective still in flight belongs to the timed work
        barrier(world)
        dt = time.perf_counter() - t0
        ctx.timing(False)
        rep = ctx.timing_report()
        return max_over_ranks(dt, world, dev), rep

    K, W = args.steps, args.warmup
    dt_train, rep_train = timed(train_step, W, K, finish_train)
    loss = pop_loss()
    prepare_topk()
    dt_topk, rep_topk = timed(topk_step, W, K)

    # ---- accuracy metrics from the index tensor: one block of users, synthetic held-out set -------------
    dt_met = None
    if world == 1:
        tip, tix = zipf_csr_device(U, I, dev, mean_log=2.0, sigma_log=0.7, dmin=1, dmax=200, seed=99)
        held = ops.DeviceTestSet.from_tensors(tip, tix, None)
        idx_blk, _ = parallel.sharded_topk(ctx, coll, st.Gu, st.Gi, st.Bi, lo, 0, Ub, k, excl=pos, algo=args.topk_algo)
        msum = torch.zeros(8, dtype=torch.float64, device=dev)

        def metrics_step():
            ops.rec_metrics(ctx, idx_blk, held, 0.0, k, u_start=0, sums=msum)

        dt_met, rep_met = timed(metrics_step, W, K)

    if rank != 0:
        return
    # ---------------- metrics ---------------------------------------------------------------------
    pairs_per_s = world * B * K / dt_train            # B triplets per rank and step
    users_per_s = (world if (topk_by_user or user_sharded) else 1) * Ub * K / dt_topk

    # HBM bytes per launch from the rocprofv3 PMC passes (profiles/r01_pmc_traffic.md), valid for the default workload only
    traffic = {}
    try:
        tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        c = tj["config"]
        if (c["users"], c["items"], c["factors"], c["batch"], c["topk_block"]) == (U, I, F, B, Ub) and world == 1:
            traffic = tj["bytes_per_launch"]
    except Exception:
        pass

    def dominant(rep):
        name = max(rep, key=lambda n: rep[n][1])
        return name, rep[name][1] / rep[name][0] * 1e-3   # seconds per launch

    # train roofline: algorithmic bytes of the dominant kernel (DESIGN.md "algorithmic bytes")
    rows_u, rows_i = int(st.Gu.shape[0]), int(st.Gi.shape[0])  # what this rank's optimiser pass actually covers
    if exchange_used[0] == "dense":
        rows_u = be.Us
    alg = {
        "k_adam_dense_Gu": 24.0 * rows_u * F,                 # theta, m, v read + write
        "k_adam_dense_Gi": 24.0 * rows_i * F,
        "k_bprmf_fwd_bwd": B * (24.0 * F + 28.0),             # 3 rows read + 3 gradient rows written (+ idx, bias)
        "k_bpr_user_seg": B * (16.0 * F + 28.0),              # gamma_u, gamma_i, gamma_j read + dGu row written
        "k_bpr_item_seg": 2.0 * B * (8.0 * F + 12.0),         # gamma_u(b) read + dGi row written, per occurrence
        "k_bpr_triplet_rows": B * (16.0 * F + 28.0),          # 3 rows read, dGu row written
        "k_rows_segsum": world * B * 8.0 * F,                 # gathered row read + reduced row written
        "rocprim_radix_sort_pairs": 3.0 * B * 16.0,
        "k_rows_apply": B * (72.0 * F + 60.0) - B * (24.0 * F + 28.0) if args.opt == "adam_lazy" else B * (24.0 * F),
        "k_bpr_sample": B * 48.0,
    }
    dn, dsec = dominant(rep_train)
    achieved = alg.get(dn, 0.0) / dsec / 1e9
    roof_train = {"kernel": dn, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get(dn),
                  "kernels_ms_per_step": {n: v[1] / K for n, v in rep_train.items()}}
    # top-k roofline: the dominant kernel is one full user x item scoring GEMM (2*U*I*F flop per launch): the fp32 MFMA
    # kernel, or one of the two bf16 passes of the screened kernel (the other pass repeats the same flops; results are
    # re-scored in fp32 and bit-identical)
    tn, tsec = dominant(rep_topk)
    flops = 2.0 * Ub * (I if (topk_by_user or user_sharded) else (hi - lo)) * F
    ach_t = flops / tsec / 1e12
    screened = tn.startswith("k_screen")
    peak_t = MFMA_BF16_PEAK_TFLOPS if screened else MFMA_F32_PEAK_TFLOPS
    roof_topk = {"kernel": tn, "bound": "mfma", "achieved": ach_t, "peak": peak_t, "unit": "TFLOP/s",
                 "frac": ach_t / peak_t, "traffic": traffic.get(tn),
                 "dtype": "bf16 MFMA screen + f32 exact re-score" if screened else "f32",
                 "effective_TFLOPs": flops / (dt_topk / K) / 1e12,
                 "kernels_ms_per_step": {n: v[1] / K for n, v in rep_topk.items()}}

    line = {
        "metric": "BPR-MF positive-pairs/sec + full-catalog top-k users/sec",
        "value": pairs_per_s, "unit": "pairs/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": dt_train / K * 1e3, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"workload": "BPRMF d=128, synthetic 1M users x 100K items (BASELINE configs[1])" if (U, I, F) == (1_000_000, 100_000, 128)
                   else f"BPRMF d={F}, synthetic {U} users x {I} items",
                   "users": U, "items": I, "factors": F, "interactions": int(pos.nnz), "batch": B,
                   "batch_per_gpu": B, "optimizer": args.opt, "topk_block": Ub, "k": k,
                   "parallelism": "single" if world == 1 else
                   (f"user-shard x{world}: train = {B} triplets/rank for the rank's own users (item table replicated) + "
                    f"all-reduce of the item gradients ({I * (F + 1) * 4 / 1e6:.0f} MB)") if exchange_used[0] == "user" else
                   f"item-shard x{world}: train = {B} triplets/rank + "
                   + ("reduce-scatter of the dense user-gradient table, optimiser on U/G user rows, all-gather of the rows"
                      if exchange_used[0] == "dense" else "all-gather of user-gradient rows")
                   + " (weak); top-k: see topk.sharding"},
        "loss_per_pair_last": loss / (B * world * (K + W)),
        "roofline": roof_train,
        "topk": {"value": users_per_s, "unit": "users/s", "ms_per_step": dt_topk / K * 1e3,
                 "scaling": "weak" if (topk_by_user or user_sharded or world == 1) else "strong",
                 "sharding": ("single" if not sharded else
                              f"by user: each rank scores {Ub}-user blocks of the users it owns against its replica of the item table, no collective"
                              if user_sharded else
                              f"by user: {Ub} users per rank and step vs the whole catalogue (item table all-gathered once per evaluation)"
                              if topk_by_user else f"by item: all users vs I/{world} items per rank + all-gather/merge of partial lists"),
                 "roofline": roof_topk},
    }
    if dt_met is not None:
        line["metrics"] = {"value": Ub * K / dt_met, "unit": "users/s", "ms_per_step": dt_met / K * 1e3,
                           "what": f"nDCG/Precision/Recall/HR/MAP/MRR/F1@{k} from the [users, k] index tensor (el_rec_metrics), "
                                   f"{int(held.nnz)} held-out interactions",
                           "kernels_ms_per_step": {n: v[1] / K for n, v in rep_met.items()}}
    if world == 1 and not args.no_cpu_baseline:
        host = {"Gu": st.Gu.cpu().numpy(), "Gi": st.Gi.cpu().numpy(), "Bi": st.Bi.cpu().numpy(),
                "indptr": pos.indptr.cpu().numpy(), "indices": pos.indices.cpu().numpy()}
        line["cpu_baseline"] = cpu_baseline(args, host)
    print(json.dumps(line), flush=True)


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
