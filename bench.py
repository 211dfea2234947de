#!/usr/bin/env python
"""bench.py -- BASELINE.json metric: BPR-MF positive-pairs/sec + full-catalog top-k users/sec @1/2/4/8 GPU.

One JSON line on rank 0.

  python bench.py [--gpus N] [--steps K] [--warmup W]

N = 1  workload = the shape BASELINE.json's metric / north_star quote their target on: BPRMF d=128 on synthetic 10M users x 1M items
       (SURVEY.md 8d "S-10M"; it fits one MI355X: tables + Adam slots 17 GB, positives' CSR 3.3 GB), inputs resident in HBM before
       the timed region.
         step (train) : sample B triplets on the device -> gather -> BPR loss -> Adam (TF-dense semantics, the reference's
                        BPRMF_batch_model.train_step) for one batch of B = --batch triplets; software-pipelined: the sampler,
                        prep and radix sort of step t+1 (they never read the model) run on a side stream under step t
         step (top-k) : fused score + masked top-k for one block of --topk-block users against the full catalogue
       `value` is the training throughput (pairs/s); the top-k leg is reported under "topk" and as topk_users_per_s /
       topk_ms_per_block / topk_frac at the top level.  Secondary legs follow (parity-test configurations of BASELINE.json, here
       with their own rooflines): "c2" = the same two steps at BASELINE configs[1] (1 M users x 100 K items: the headline of rounds
       1-3), with the batch sweep and the plugin end-to-end leg on its data; "c5_per_gpu" = configs[4]'s per-GPU shape; "vae" =
       Mult-VAE at the ML-20M shape (configs[2]), "neumf" = NeuMF d=128 at the per-GPU shape of configs[3] under user sharding.
N > 1  the SAME 10M x 1M x 128 model partitioned over the ranks (what north_star's "at 1/2/4/8 MI355X" names), B triplets per
       rank and step.  One process per GPU.  `python bench.py --gpus N` launches itself under torch.distributed.run when WORLD_SIZE is not
       set (the driver's own torchrun launch is honoured as is).  Primary leg: USER shards (the rank's user rows + a replica
       of the item table, all-reduce of the item gradients, collective-free top-k).  Second leg "item_shard": north_star's
       partitioning (item rows sharded, user-gradient exchange, all-gather + merge of partial top-k).  Each leg reports the
       bytes and the time of its collectives.

Both BPR legs carry a roofline object for their dominant kernel (duration from hipEvents recorded inside the library on the
launch stream, around that kernel only inside the timed region; the per-kernel breakdown is a separate, untimed pass).  `cpu_baseline` times the CPU restatements (oracle/: ports of the reference path; /root/reference is absent on
the GPU box) on a bounded sample, rank 0 / N=1 only.
"""
import argparse
import hashlib
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
MFMA_F32_PEAK_TFLOPS = 157.3   # v_mfma_f32_32x32x2_f32 dense peak
MFMA_BF16_PEAK_TFLOPS = 2500.0  # v_mfma_f32_32x32x16_bf16 dense peak (MI355X_MICROARCH.md)
# k_gemm_b3 (the Dense layers' default): fp32 operands as three bf16 planes, six bf16 products per fp32 product -- the ceiling of an
# fp32-grade product on the bf16 matrix instruction
MFMA_B3_PEAK_TFLOPS = MFMA_BF16_PEAK_TFLOPS / 6.0


def gemm_roofline(rep, flops, steps):
    """Roofline entry of the dense-layer GEMM launches of one step: the three-way-split kernel where it ran (its flop count is the fp32
    product's, its peak the bf16 instruction's / 6), the fp32 matrix instruction otherwise (option gemm_split = 0, small shapes)."""
    gms = sum(v[1] for n, v in rep.items() if n.startswith("k_gemm")) / steps
    b3 = sum(v[1] for n, v in rep.items() if n == "k_gemm_b3") / steps
    ach = flops / (gms * 1e-3) / 1e12 if gms > 0 else 0.0
    split = b3 > 0.5 * gms
    peak = MFMA_B3_PEAK_TFLOPS if split else MFMA_F32_PEAK_TFLOPS
    return gms, {"kernel": "k_gemm_b3" if split else "k_gemm_f32", "bound": "mfma", "achieved": ach, "peak": peak, "unit": "TFLOP/s",
                 "frac": ach / peak, "frac_of_f32_instruction_peak": ach / MFMA_F32_PEAK_TFLOPS,
                 "dtype": "f32 operands and results; products on v_mfma_f32_32x32x16_bf16 as three bf16 planes per operand, six per fp32 "
                          "product, fp32 accumulation (error at fp32 rounding level: tests/test_gpu_dense.py)" if split else "f32"}
MFMA_BF16_RANDOM_TFLOPS = 1800.0  # what a bare stream of that instruction sustains on random operands (power-limited; measured:
#                                   scripts/exp/mfma_rate.hip, profiles/r02_mfma_ceiling.md: 1.77-1.85 PFLOP/s, 2.48 on zeros)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--repeats", type=int, default=3, help="timed repeats of K steps per leg; the median repeat is the leg's time")
    ap.add_argument("--users", type=int, default=10_000_000, help="headline shape: north_star's target (10M users x 1M items x 128)")
    ap.add_argument("--items", type=int, default=1_000_000)
    ap.add_argument("--factors", type=int, default=128)
    ap.add_argument("--batch", type=int, default=1 << 20)
    ap.add_argument("--topk-block", type=int, default=131072)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--opt", default="adam_tf_dense", choices=["adam_tf_dense", "adam_lazy", "sgd"])
    ap.add_argument("--train-algo", default="auto", choices=["auto", "atomic", "sorted"])
    ap.add_argument("--replay", default="series", choices=["series", "exact"],
                    help="N = 1, deferred decay: a waiting row's gradient-free Adam steps in closed form (el_bprmf_state.replay_series: O(1) per "
                         "element; inside the parity tolerances, tests/test_gpu_fullsize_c4.py) or step by step (Keras' bits).  `value` is this "
                         "mode's; the other mode's training throughput rides under legs.replay_other")
    ap.add_argument("--topk-algo", default="auto", choices=["auto", "screen", "mfma", "simple"])
    ap.add_argument("--shard", default="user", choices=["user", "item"],
                    help="N > 1, primary leg: shard the USER table (item table replicated, all-reduce of item gradients) or the "
                         "ITEM table (north_star's formulation: user table replicated, user-gradient exchange per --exchange)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "rows", "dense"],
                    help="N > 1, item shards: how user-row gradients travel (parallel.pick_exchange)")
    ap.add_argument("--item-exchange", default="auto", choices=["auto", "dense", "rows"],
                    help="N > 1, user shards: the item gradients meet in one dense all-reduce of [I, F + 1], or as an all-gather of each rank's "
                         "touched (item id, gradient row) records (parallel.pick_item_exchange: rows only while the lists are smaller)")
    ap.add_argument("--topk-shard", default=None, choices=["user", "item"],
                    help="N > 1: users are independent units (no collective) / north_star's item shards + all-gather of partial "
                         "top-k (default: user for --shard user, item for the item-shard leg)")
    ap.add_argument("--legs", default="auto",
                    help="comma list of bpr,item_shard,c2,sweep,plugin,c5,vae,neumf,graph,metrics (auto: N=1 -> all but item_shard; N>1 -> bpr,item_shard)")
    ap.add_argument("--c2-shape", default="1000000,100000", help="users,items of the c2 leg (BASELINE configs[1]; sweep and plugin legs run on its data)")
    ap.add_argument("--c5-shape", default="6250000,5000000,256",
                    help="users,items,factors of the c5 leg (BASELINE configs[4] = 50M x 5M x 256 on 8 GPUs: the per-GPU shape under user sharding)")
    ap.add_argument("--comm", default="torch", choices=["torch", "abi"],
                    help="N > 1: collectives through torch.distributed (RCCL process group) or through the library's own C ABI "
                         "(el_comm_init / el_allreduce_rows / el_allgather_topk: RCCL called directly)")
    ap.add_argument("--prefetch", action="store_true",
                    help="item-shard leg: draw the triplets of step t+1 on a side stream during step t")
    ap.add_argument("--no-pipeline", action="store_true",
                    help="N = 1: do NOT draw and order (prep + radix sort) the batch of step t+1 on a side stream while step t's segment "
                         "kernels and optimiser pass run (default: pipelined, 1.47 -> 1.41 ms per step; same kernels, same results)")
    ap.add_argument("--force-sharded", action="store_true", help="run the N > 1 code path even with one rank (API check)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--neumf-prefetch", action="store_true", help="NeuMF leg: sampler + el_nmf_presort of the next batch on a side stream (measured slower: off)")
    ap.add_argument("--trained-epochs", type=float, default=1.0, help="N = 1: epochs of further training before the top-k block is timed a second "
                                                                     "time on trained tables (0: skip)")
    ap.add_argument("--neumf-trained-steps", type=int, default=3000, help="neumf leg: further training steps before its scoring step is timed a "
                                                                        "second time (0: skip)")
    ap.add_argument("--legs-file", default=None, help="where the full per-leg report goes (default: bench_legs.json beside bench.py; "
                                                      "the stdout line is the compact summary)")
    ap.add_argument("--cpu-topk-users", type=int, default=640)
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="CPU time budget of each cpu_baseline measurement")
    ap.add_argument("--vae-shape", default="138493,26744,600,200,512", help="users,items,hidden,latent,batch of the vae leg")
    ap.add_argument("--neumf-shape", default="1250000,1000000,128,262144", help="users,items,factors,batch of the neumf leg")
    ap.add_argument("--graph-shape", default="1000000,100000,64,2", help="users,items,factors,n_layers of the graph leg (LightGCN; MF2020 beside it)")
    ap.add_argument("--neumf-topk-users", type=int, default=128, help="users per full-catalogue scoring step of the neumf leg (0: skip)")
    return ap.parse_args()


# ---------------------------------------------------------------------------------------------------------------------
# launch
# ---------------------------------------------------------------------------------------------------------------------
def self_launch(args):
    """`python bench.py --gpus N` with N > 1 outside a torchrun environment: start N ranks (one per GPU, RCCL) and relay
    rank 0's JSON line.  Refuses -- loudly, non-zero -- when the node has fewer than N GPUs instead of silently measuring one."""
    shared = os.environ.get("EL_BENCH_SHARED_GPU") == "1"
    have = torch.cuda.device_count()
    if have < args.gpus and not shared:
        print(f"bench.py: --gpus {args.gpus} requested but this node exposes {have} GPU(s); refusing to report a {args.gpus}-GPU "
              f"number from fewer devices (EL_BENCH_SHARED_GPU=1 runs the control flow on one GPU over gloo for development)",
              file=sys.stderr)
        return 2
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC for RCCL (see the environment notes)
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def dist_setup(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    backend = None
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
        backend = dist.get_backend()
        world = dist.get_world_size()                       # what the process group observed, not what the flag claims
    elif args.force_sharded:
        # one rank, but through RCCL and the N > 1 code path: an API check of the collectives on a 1-GPU box
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        os.environ["EL_FORCE_COLLECTIVES"] = "1"
        torch.cuda.set_device(local)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", local))
        backend = dist.get_backend()
    return world, rank, local, backend


from elliot_amd.pipeline import PrefetchPointwise, PrefetchSampler, cover_batches, cover_triplets  # noqa: E402,F401  (the step pipeline the tests drive too)


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


REPEATS = 3                     # timed repeats of K steps per leg; the median is reported


class _Report(dict):
    """{kernel: (launches, total_ms)} of the breakdown pass; .live = the same for the kernel bracketed inside the timed region;
    .repeats_ms = ms per step of every timed repeat (the leg reports their median)."""
    live = {}
    repeats_ms = []
    calls = 0


def timed(ctx, world, fn, warmup, steps, finish=None, events_in_timed_region=True, fn_breakdown=None, repeats=None):
    """W untimed calls, then R repeats of EXACTLY K calls, each repeat between barrier + synchronize on both sides and taken as
    the max over ranks; the leg's time is the MEDIAN repeat (box-to-box and run-to-run spread of a 30 ms region is +-3 %; all
    repeats are reported next to it: rep.repeats_ms).
    A hipEvent between two kernels costs their back-to-back overlap (measured: events around all 8 launches of the 1.5 ms
    training step = +4 % wall).  So the timed region carries events on ONE kernel -- the dominant one, whose live duration the
    roofline is computed from (report.live) -- and the per-kernel breakdown comes from a separate pass of K steps with events
    on every launch, taken first: it also names the dominant kernel.  fn_breakdown: the step to run in that pass when `fn`
    overlaps kernels on several streams (per-kernel elapsed times of concurrent kernels mean nothing; the breakdown is then
    of the same kernels run back to back).  Legs of 40+ short launches per step (Mult-VAE, NeuMF:
    events_in_timed_region=False) are timed without any event."""
    repeats = REPEATS if repeats is None else repeats
    for _ in range(warmup):
        fn()
    if finish:
        finish()
    barrier(world)
    ctx.timing(True)                                             # breakdown pass (untimed)
    for _ in range(steps):
        (fn_breakdown or fn)()
    if finish:
        finish()
    barrier(world)
    ctx.timing(False)
    rep = _Report(ctx.timing_report())
    only = max(rep, key=lambda n: rep[n][1]) if (rep and events_in_timed_region) else None
    if fn_breakdown is not None:
        fn()                                                     # (re-prime the look-ahead of the pipelined step)
        if finish:
            finish()
    if only:
        ctx.timing(True, only=only)
    dts = []
    for _ in range(max(1, repeats)):
        barrier(world)
        t0 = time.perf_counter()
        for _ in range(steps):
            fn()
        if finish:
            finish()                                             # a collective still in flight belongs to the timed work
        barrier(world)
        dts.append(max_over_ranks(time.perf_counter() - t0, world, ctx.device))
    ctx.timing(False)
    rep.live = ctx.timing_report()                               # (launch counts and times over all repeats: the mean is unaffected)
    rep.repeats_ms = [d / steps * 1e3 for d in dts]
    rep.calls = warmup + steps + (1 if fn_breakdown is not None else 0) + len(dts) * steps      # calls of a step function in all
    return sorted(dts)[len(dts) // 2], rep


XGMI_LINK_GBS = 153.0            # one xGMI link, one direction (7 links per GPU on an 8-GPU MI355X node: point to point, no switch)


def expected_collective(op, nbytes, world):
    """Wire model of one collective on the node's xGMI mesh, printed beside the measured time so that the first real N-GPU run
    checks itself.  nbytes = the full buffer (all_reduce: the buffer; reduce_scatter: its input; all_gather: its output).  A rank
    puts frac * nbytes on the wire (2 (G-1)/G for all_reduce, (G-1)/G otherwise).  Two bounds: "direct" = every peer reached over
    its own link at once (reduce-scatter / all-gather by direct exchange: min(G-1, 7) links busy), "ring" = one link per direction
    (what a ring schedule is bound by on point-to-point links).  Link rate 153 GB/s at 100 %; RCCL typically delivers 60-75 % of it."""
    if world < 2:
        return None
    frac = {"all_reduce": 2.0 * (world - 1) / world, "reduce_scatter": (world - 1.0) / world, "all_gather": (world - 1.0) / world}[op]
    wire = frac * nbytes
    links = min(world - 1, 7)
    return {"bytes_on_wire_per_rank": wire, "direct_all_links_ms": wire / (links * XGMI_LINK_GBS * 1e9) * 1e3,
            "ring_one_link_ms": wire / (XGMI_LINK_GBS * 1e9) * 1e3, "link_GBs": XGMI_LINK_GBS, "links_used_direct": links}


def time_collective(world, dev, fn, reps=5):
    """Wall time of one collective call (barrier-bracketed mean of `reps`, max over ranks), outside every timed region."""
    fn()
    barrier(world)
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    barrier(world)
    return max_over_ranks((time.perf_counter() - t0) / reps, world, dev) * 1e3


def dominant(rep):
    """(name, seconds per launch) of the kernel with the largest share of the breakdown pass; the duration is the one
    measured INSIDE the timed region when that kernel was bracketed there."""
    name = max(rep, key=lambda n: rep[n][1])
    cnt, ms = getattr(rep, "live", {}).get(name, rep[name])
    return name, ms / cnt * 1e-3


def source_hash():
    """Hash of the kernel sources: PMC traffic figures are only quoted for the code they were collected on."""
    h = hashlib.sha256()
    csrc = os.path.join(REPO, "elliot_amd", "csrc")
    for f in sorted(os.listdir(csrc)):
        if f.endswith((".hip", ".h")):
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(fh.read())
    return h.hexdigest()[:16]


def load_traffic(U, I, F, B, Ub, world, leg=None):
    """HBM bytes per launch from the rocprofv3 PMC passes (scripts/collect_traffic.sh -> profiles/traffic.json, one entry per bench
    leg).  Not measured in this run: quoted only when the file was collected on THIS workload and THESE kernel sources, otherwise
    dropped (null).  leg: the entry to look in (default: whichever BPR workload has this shape)."""
    try:
        tj = json.load(open(os.path.join(REPO, "profiles", "traffic.json")))
        if tj.get("source_hash") != source_hash():
            return {}, f"profiles/traffic.json is stale (collected on kernel sources {tj.get('source_hash')}, commit {tj.get('commit')})"
        if world != 1:
            return {}, "profiles/traffic.json holds single-GPU workloads"
        for name, w in tj.get("workloads", {}).items():
            c = w.get("config", {})
            if (leg is not None and name == leg) or (leg is None and "users" in c and
                                                     (c["users"], c["items"], c["factors"], c["batch"], c["topk_block"]) == (U, I, F, B, Ub)):
                return w["bytes_per_launch"], (f"rocprofv3 PMC passes at commit {tj.get('commit')} (scripts/collect_traffic.sh, workload "
                                               f"'{name}'), same kernel sources")
        return {}, "profiles/traffic.json has no entry for this workload"
    except Exception as ex:  # noqa: BLE001
        return {}, f"no traffic file ({type(ex).__name__})"


# ---------------------------------------------------------------------------------------------------------------------
# CPU baselines
# ---------------------------------------------------------------------------------------------------------------------
def cpu_baseline(args, host):
    """The reference path on this host's cores, bounded samples of the same workload.
    (1) "value": N-thread torch-CPU restatement of BPRMF_batch_model.train_step / predict + get_top_k (the reference's TF eager
        path is BLAS + Eigen thread pools; TF 2.3.2 cannot be installed here) -- oracle/torch_cpu.py;
    (2) "port_1core": the NumPy / plain-C oracles the parity tests check against, one core."""
    from oracle import bprmf_batch as ob
    from oracle import cref
    from oracle import torch_cpu as tc
    cores = tc.use_all_cores()                               # affinity mask / cgroup quota, not os.cpu_count()
    budget = args.cpu_seconds
    out = {"kind": "port", "cores": cores, "host_cpu_count": os.cpu_count(), "unit": "pairs/s"}
    rs = np.random.RandomState(0)
    B = args.batch
    # the N-thread restatement keeps three copies of the tables (theta, m, v) + temporaries of the same size: the whole shape when
    # the host has the memory for it, else the first users of it (a bounded sample of the same workload; the dense Adam pass -- the
    # part of a step that does not depend on B -- then covers fewer rows, which flatters the CPU)
    Uc = args.users
    need = 6.5 * (args.users + args.items) * args.factors * 4
    try:
        import psutil
        avail = psutil.virtual_memory().available
    except Exception:  # noqa: BLE001
        avail = 0
    if avail < need + (8 << 30):
        Uc = max(1, min(args.users, int((max(avail, 16 << 30) - (8 << 30)) / (6.5 * args.factors * 4)) - args.items))
    Gu_h = host["Gu"][:Uc]
    u = torch.from_numpy(rs.randint(0, Uc, B))
    i = torch.from_numpy(rs.randint(0, args.items, B))
    j = torch.from_numpy(rs.randint(0, args.items, B))
    m = tc.BprmfBatchTorchCpu(Gu_h, host["Gi"], host["Bi"], 0.001, 0.1, 0.001)
    m.train_step(u, i, j)                                   # warm-up (page faults of the optimiser slots)
    t0 = time.perf_counter()
    n = 0
    while n < 2 or time.perf_counter() - t0 < budget:
        m.train_step(u, i, j)
        n += 1
    dt = time.perf_counter() - t0
    out["value"] = B * n / dt
    out["sample"] = (f"oracle/torch_cpu.py BprmfBatchTorchCpu.train_step (restatement of BPRMF_batch_model.py:58-80 with Keras dense "
                     f"Adam; TF 2.3.2 not installable), {n} steps of B={B} on U={Uc}" + (f" (the first {Uc} of {args.users} users: host memory)" if Uc < args.users else "")
                     + f", I={args.items}, F={args.factors}, fp32, {cores} torch threads; {dt:.2f}s")
    del m
    # top-k: matmul + where + top_k on blocks of users (the [Ub, I] score block is materialised, as the reference does: 4 GB per block)
    Ub = 4096 if args.items <= 250_000 else 1024
    Gu, Gi, Bi = (torch.from_numpy(host[k]) for k in ("Gu", "Gi", "Bi"))
    indptr, indices = torch.from_numpy(host["indptr"]), torch.from_numpy(host["indices"]).to(torch.int64)
    t0 = time.perf_counter()
    nb = 0
    while nb < 1 or (time.perf_counter() - t0 < budget and (nb + 1) * Ub <= args.users):
        s = nb * Ub
        e = min(s + Ub, args.users)
        ip = indptr[s:e + 1] - indptr[s]
        tc.predict_topk(Gu[s:e], Gi, Bi, ip, indices[int(indptr[s]):int(indptr[e])], args.k)
        nb += 1
    dt = time.perf_counter() - t0
    nu_done = min(nb * Ub, args.users)
    out["topk"] = {"value": nu_done / dt, "unit": "users/s", "cores": cores, "kind": "port",
                   "sample": f"oracle/torch_cpu.py predict_topk (addmm + masked fill + torch.topk; BPRMF_batch_model.py:83-88), "
                             f"{nu_done} users x {args.items} items in blocks of {Ub}, F={args.factors}, k={args.k}, {cores} threads; {dt:.2f}s"}
    del Gu, indices
    # ---- the parity oracles themselves, one core, on the first users of the shape (a NumPy Adam pass over 11 M rows takes ~20 s)
    p1 = {"cores": 1, "kind": "port"}
    Bc = min(args.batch, 1 << 16)
    U1 = min(args.users, 1_000_000)
    un, inn, jn = (rs.randint(0, U1, Bc), i[:Bc].numpy(), j[:Bc].numpy())
    orc = ob.BPRMFBatchOracle(host["Gu"][:U1], host["Gi"], host["Bi"], 0.001, 0.1, 0.001, optimizer=args.opt)
    t0 = time.perf_counter()
    n = 0
    while n < 1 or time.perf_counter() - t0 < budget * 0.6:
        orc.train_step((un, inn, jn))
        n += 1
    dt = time.perf_counter() - t0
    p1["value"], p1["unit"] = Bc * n / dt, "pairs/s"
    p1["sample"] = (f"oracle/bprmf_batch.py train_step ({args.opt}), {n} steps, B={Bc}, U={U1}" + (f" (the first {U1} of {args.users} users)" if U1 < args.users else "")
                    + f", I={args.items}, NumPy fp32 single thread; {dt:.2f}s")
    del orc
    nu = args.cpu_topk_users if args.items <= 250_000 else max(32, args.cpu_topk_users // 8)
    t0 = time.perf_counter()
    cref.score_topk_f32(host["Gu"][:nu], host["Gi"], host["Bi"], 0, nu, args.k,
                        excl=(host["indptr"][:nu + 1], host["indices"][:int(host["indptr"][nu])]))
    dt = time.perf_counter() - t0
    p1["topk"] = {"value": nu / dt, "unit": "users/s",
                  "sample": f"oracle/c/el_oracle.c orc_score_topk_f32 (fmaf chain, the bit-exact checker), {nu} users x {args.items} items; {dt:.2f}s"}
    out["port_1core"] = p1
    return out


def single_gpu_trainer(args, ctx, data, Gu, Gi, Bi, replay):
    """The single-GPU training loop of a BPR leg: state, (pipelined) step, cover batches -- everything up to the timed region."""
    from elliot_amd import ops
    dev = ctx.device
    U, I, B = args.users, args.items, args.batch
    indptr, indices, pos = data["indptr"], data["indices"], data["pos"]
    lr, l_w, l_b = 0.001, 0.1, 0.001                                          # BPRMF_batch.py:66-71 defaults
    finish_train = breakdown_step = None
    cover_steps = 0
    st = ops.BprmfDeviceState(ctx, Gu, Gi, Bi, optimizer=args.opt, replay=replay)
    # Software pipeline of the step: drawing AND ordering a batch (sampler, prep, radix sort) reads only the positives' CSR and
    # the triplets, never the model -- so the batch of step t+1 is prepared on a side stream while step t's segment kernels and
    # optimiser pass run (latency-bound work under bandwidth-bound work: 1.47 -> 1.41 ms per step).  Same kernels, same
    # triplets, same results as the sequential step (the loss of the last step is identical to the last digit).
    pipelined = (not args.no_pipeline) and args.train_algo in ("auto", "sorted") and B >= 2048 and args.opt == "adam_tf_dense"
    # (el_bprmf_apply is the dense optimiser pass; the lazy / row-wise optimisers keep the one-call step)
    sampler = PrefetchSampler(ctx, pos, B, 42, enabled=pipelined, presort_state=st if pipelined else None)
    hi_prio = torch.cuda.Stream(device=dev, priority=-1) if pipelined else None     # the step's own kernels outrank the look-ahead
    if hi_prio is not None:
        hi_prio.wait_stream(torch.cuda.current_stream())

    def train_step():
        if pipelined:
            with torch.cuda.stream(hi_prio):
                t, b = sampler.next()
                st.train_step_presorted(t[0], t[1], t[2], lr, l_w, l_b, sampler.ws[b])
                sampler.release(b)
            return
        t, b = sampler.next()
        st.train_step(t[0], t[1], t[2], lr, l_w, l_b, algo=args.train_algo)
        sampler.release(b)

    if pipelined:
        seq_drawn = [1 << 40]                                    # (its own part of the Philox stream)

        def train_step_sequential():
            """The same step with nothing in flight beside it: what the per-kernel breakdown pass runs."""
            torch.cuda.current_stream().wait_stream(hi_prio)
            torch.cuda.current_stream().wait_stream(sampler.side)
            t = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=seq_drawn[0])
            seq_drawn[0] += B
            st.train_step(t[0], t[1], t[2], lr, l_w, l_b, algo=args.train_algo)
            hi_prio.wait_stream(torch.cuda.current_stream())
            sampler.side.wait_stream(torch.cuda.current_stream())
        breakdown_step = train_step_sequential
    # Steady state of the deferred decay before anything is timed: a row that never had a gradient (m = v = 0) is a fixed point
    # of the gradient-free Adam step and the replay kernels skip it, so a short run from fresh tables would replay far fewer
    # element-steps than a long one.  "Cover" batches give EVERY user row and EVERY item row a gradient once (users in order,
    # one of their positives each; negatives in item order): ceil(max(U, I) / B) extra untimed steps, the same train_step.
    if args.opt == "adam_tf_dense" and (4 * B <= U or 2 * B <= I):
        cover_batches(st, indptr, indices, U if 4 * B <= U else 0, I if 2 * B <= I else 0, U, I, B, lr, l_w, l_b, args.train_algo)
        cover_steps = -(-max(U if 4 * B <= U else 0, I if 2 * B <= I else 0) // B)
    # deferred decay of the user table (the state turns it on at its first batch when 4 B <= U): the timed region ends with
    # the replay of every postponed row update -- each (element, step) update of Keras' every-row Adam is inside the timed
    # region.  A no-op in the every-row form.
    def finish_train():
        if hi_prio is not None:
            with torch.cuda.stream(hi_prio):
                st.sync()
        else:
            st.sync()
    return {"st": st, "train_step": train_step, "finish_train": finish_train, "breakdown_step": breakdown_step, "pipelined": pipelined,
            "cover_steps": cover_steps, "sampler": sampler}


def replay_other_leg(args, ctx, data, replay):
    """The headline training step once more with the OTHER replay mode of the deferred decay (train only: same data, same initial
    tables, same cover batches, same pipelined timed region ending with the flush of every pending row)."""
    dev = ctx.device
    U, I, F, B = args.users, args.items, args.factors, args.batch
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    lim_u, lim_i = (6.0 / (U + F)) ** 0.5, (6.0 / (I + F)) ** 0.5
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * lim_u
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * lim_i
    Bi = torch.zeros(I, device=dev)
    tr = single_gpu_trainer(args, ctx, data, Gu, Gi, Bi, replay)
    del Gu, Gi, Bi
    if not tr["st"].deferred and not tr["st"]._deferred_auto:
        return None
    dt, rep = timed(ctx, 1, tr["train_step"], args.warmup, args.steps, tr["finish_train"], fn_breakdown=tr["breakdown_step"])
    loss = tr["st"].pop_loss()
    K = args.steps
    return {"replay": replay, "value": B * K / dt, "unit": "pairs/s", "ms_per_step": dt / K * 1e3, "repeats_ms_per_step": rep.repeats_ms,
            "loss_per_pair_last": loss / (B * K * max(1, len(rep.repeats_ms))) if loss else None,
            "kernels_ms_per_step": {n: v[1] / K for n, v in rep.items()},
            "what": ("waiting rows brought forward step by step: the same fp32 operations in the same order as Keras' every-row pass (bit-identical "
                     "tables, tests/test_gpu_bpr.py)" if replay == "exact" else
                     "waiting rows brought forward in closed form (el_bprmf_state.replay_series)")}


# ---------------------------------------------------------------------------------------------------------------------
# BPR-MF leg (train + top-k), any sharding
# ---------------------------------------------------------------------------------------------------------------------
def bpr_leg(args, ctx, world, rank, data, shard, topk_shard, with_metrics=False, keep_host=False):
    from elliot_amd import ops, parallel
    dev = ctx.device
    U, I, F, B, k = args.users, args.items, args.factors, args.batch, args.k
    K, W = args.steps, args.warmup
    indptr, indices, pos = data["indptr"], data["indices"], data["pos"]
    g = torch.Generator(device=dev)
    g.manual_seed(42)
    lim_u, lim_i = (6.0 / (U + F)) ** 0.5, (6.0 / (I + F)) ** 0.5           # GlorotUniform (BPRMF_batch_model.py:39-42)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * lim_u
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * lim_i
    Bi = torch.zeros(I, device=dev)

    lo, hi = parallel.item_range(I, rank, world) if shard == "item" else (0, I)
    lr, l_w, l_b = 0.001, 0.1, 0.001                                          # BPRMF_batch.py:66-71 defaults
    finish_train = None
    breakdown_step = None
    pipelined = False
    exchange = None
    coll = data.get("coll") or parallel._Collectives()
    sharded = world > 1 or args.force_sharded
    collectives = []                                                          # (what, op, bytes per rank and call, callable)
    cover_steps = 0
    if not sharded:
        tr = single_gpu_trainer(args, ctx, data, Gu, Gi, Bi, args.replay)
        st, pos_train, train_step, finish_train, breakdown_step = tr["st"], pos, tr["train_step"], tr["finish_train"], tr["breakdown_step"]
        pop_loss, pipelined, cover_steps = st.pop_loss, tr["pipelined"], tr["cover_steps"]
    elif shard == "user":
        # USER shards: the rank owns the user rows [ulo, uhi) and a replica of the item table; B triplets per rank for its
        # own users (items: the whole catalogue, the reference's sampling distribution), all-reduce of the item gradients
        ulo, uhi = parallel.user_range(U, rank, world)
        uip = (indptr[ulo:uhi + 1] - indptr[ulo]).contiguous()
        uix = indices[int(indptr[ulo]):int(indptr[uhi])].contiguous()
        pos_train = ops.DeviceCSR.from_tensors(uip, uix, I)
        be = parallel.HipUserShardBackend(ctx, Gu[ulo:uhi], Gi, Bi, optimizer=args.opt)
        item_ex = args.item_exchange if args.item_exchange != "auto" else parallel.pick_item_exchange(I, F, B, world)
        trainer = parallel.ShardedBprmfByUser(be, coll, item_exchange=item_ex)
        st = be.state
        exchange = "user"
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
        scratch = torch.zeros_like(st.item_grad_flat)
        ixb = parallel.item_exchange_bytes(I, F, B, world)
        n_touch = int(ixb["touched"])
        rows_scratch = torch.zeros((max(1, n_touch), F + 2), dtype=torch.float32, device=dev)
        # both forms are timed and priced (expected_ms: the wire model); the step itself runs `item_ex`
        collectives.append(("item gradients gGi [I,F] + gBi [I], dense" + (" (in use)" if item_ex == "dense" else ""), "all_reduce",
                            scratch.numel() * 4, lambda: coll.all_reduce_sum(scratch)))
        collectives.append((f"item gradients as touched rows: ~{n_touch} (id, dGi row, dBi) records per rank" + (" (in use)" if item_ex == "rows" else ""),
                            "all_gather", world * rows_scratch.numel() * 4, lambda: coll.all_gather(rows_scratch)))
    else:
        # item-sharded training: B triplets PER RANK with positive and negative inside the rank's shard, exchange of the
        # user-row gradients, identical user-table replicas (elliot_amd/parallel.py)
        sip, six = parallel.shard_csr(indptr, indices, lo, hi)
        pos_train = ops.DeviceCSR.from_tensors(sip, six, hi - lo)
        exchange = args.exchange if args.exchange != "auto" else parallel.pick_exchange(U, B, world)
        if exchange == "dense":
            # reduce-scatter of the dense user-gradient table, optimiser on the owned user rows, all-gather of the rows
            be = parallel.HipDenseBackend(ctx, Gu, Gi[lo:hi].contiguous(), Bi[lo:hi].contiguous(), rank, world, optimizer=args.opt)
            trainer = parallel.ShardedBprmfDense(be, coll)
            full = torch.zeros_like(be.state.gGu)
            own = torch.zeros_like(be.g_own)
            collectives.append(("dense user-gradient table gGu [U,F]", "reduce_scatter", full.numel() * 4,
                                lambda: coll.reduce_scatter_rows(own, full)))
            collectives.append(("updated user rows Gu [U/G,F] -> every replica", "all_gather", full.numel() * 4,
                                lambda: coll.all_gather_rows_into(full, own)))
        else:
            be = parallel.HipBackend(ctx, Gu, Gi[lo:hi].contiguous(), Bi[lo:hi].contiguous(), optimizer=args.opt)
            trainer = parallel.ShardedBprmf(be, coll)
            rows = torch.zeros((B, F), dtype=torch.float32, device=dev)
            collectives.append(("per-triplet user-gradient rows [B,F] (+ ids)", "all_gather", world * B * (F + 1) * 4,
                                lambda: coll.all_gather(rows)))
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
    user_sharded = sharded and shard == "user"
    topk_by_user = sharded and topk_shard == "user" and not user_sharded
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
        if sharded:
            part = torch.zeros((Ub, k), dtype=torch.float32, device=dev)
            collectives.append(("partial top-k lists [Ub,k] (ids + scores) per block", "all_gather", 2 * world * Ub * k * 4,
                                lambda: (coll.all_gather(part), coll.all_gather(part))))

    dt_train, rep_train = timed(ctx, world, train_step, W, K, finish_train, fn_breakdown=breakdown_step)
    loss = pop_loss()
    prepare_topk()
    Kt = int(getattr(args, "topk_steps", 0) or K)            # (a leg whose top-k block takes 0.3 s times fewer of them than training steps)
    dt_topk, rep_topk = timed(ctx, world, topk_step, min(W, Kt), Kt)
    single = world == 1 and not args.force_sharded
    steps_so_far = int(rep_train.calls) + cover_steps
    scr0 = ops.topk_screen_stats(ctx) if single else None     # (before any other scoring call: the diagnostics are the last call's)

    # ---- the fp32-only MFMA kernel beside the screened one: its arithmetic IS the reference's fp32 matmul form, the screened
    #      route returns the same bits after its exact re-score (tests/test_gpu_topk.py)
    f32_entry = None
    if world == 1 and not args.force_sharded and args.topk_algo == "auto" and F <= 256 and k <= 40:
        def topk_f32_step():
            s = (blk[0] % n_blocks) * Ub
            blk[0] += 1
            ops.score_topk(ctx, st.Gu, st.Gi, st.Bi, s, s + Ub, k, excl=pos, algo="mfma")
        K32 = max(2, min(K, 4))
        dt32, rep32 = timed(ctx, world, topk_f32_step, 1, K32)
        n32 = dominant(rep32)[0]
        s32 = rep32[n32][1] / K32 * 1e-3                       # all launches of the kernel in one block (it may split the items)
        f32_entry = {"value": Ub * K32 / dt32, "unit": "users/s", "ms_per_step": dt32 / K32 * 1e3, "steps": K32,
                     "repeats_ms_per_step": rep32.repeats_ms,
                     "roofline": {"kernel": n32, "bound": "mfma", "achieved": 2.0 * Ub * I * F / s32 / 1e12, "peak": MFMA_F32_PEAK_TFLOPS,
                                  "unit": "TFLOP/s", "frac": 2.0 * Ub * I * F / s32 / 1e12 / MFMA_F32_PEAK_TFLOPS, "dtype": "f32",
                                  "kernels_ms_per_step": {n: v[1] / K32 for n, v in rep32.items()}}}

    # ---- the same block on TRAINED tables.  The screened route's work depends on the model: a trained BPR model puts its large-norm
    #      (popular) items at the top of every list, exactly where the bf16 bound is widest.  After one more EPOCH
    #      of training (`transactions` = nnz triplets, BPRMF_batch.py:95-109) the block is timed again, with the records the bf16 pass keeps
    #      per user and the users that fall back to the exact kernels
    trained = None
    if single and args.trained_epochs > 0:
        n_more = max(1, int(args.trained_epochs * int(pos.nnz)) // B)
        for _ in range(n_more):
            train_step()
        if finish_train:
            finish_train()
        pop_loss()
        Kt2 = max(2, min(Kt, 5))
        dt_tr, rep_tr = timed(ctx, world, topk_step, 1, Kt2)
        scr1 = ops.topk_screen_stats(ctx)
        tn2, tsec2 = dominant(rep_tr)
        trained = {"train_steps_before": steps_so_far + n_more, "value": Ub * Kt2 / dt_tr, "unit": "users/s", "ms_per_step": dt_tr / Kt2 * 1e3,
                   "steps": Kt2, "records_per_user": scr1["records_per_user"], "fallback_users": scr1["fallback_users"],
                   "kernel": tn2, "kernel_frac": (2.0 * Ub * I * F / tsec2 / 1e12) / (MFMA_BF16_PEAK_TFLOPS if tn2.startswith("k_screen") else MFMA_F32_PEAK_TFLOPS),
                   "kernels_ms_per_step": {n: v[1] / Kt2 for n, v in rep_tr.items()},
                   "what": f"the same top-k block after {n_more} more training steps ({args.trained_epochs:g} epoch(s) of {int(pos.nnz)} triplets)"}

    # ---- accuracy metrics from the index tensor: one block of users, synthetic held-out set -------------
    met = None
    if with_metrics and world == 1:
        from elliot_amd.synthetic import zipf_csr_device
        tip, tix = zipf_csr_device(U, I, dev, mean_log=2.0, sigma_log=0.7, dmin=1, dmax=200, seed=99)
        held = ops.DeviceTestSet.from_tensors(tip, tix, None)
        idx_blk, _ = parallel.sharded_topk(ctx, coll, st.Gu, st.Gi, st.Bi, lo, 0, Ub, k, excl=pos, algo=args.topk_algo)
        msum = torch.zeros(8, dtype=torch.float64, device=dev)
        dt_met, rep_met = timed(ctx, world, lambda: ops.rec_metrics(ctx, idx_blk, held, 0.0, k, u_start=0, sums=msum), W, K)
        met = {"value": Ub * K / dt_met, "unit": "users/s", "ms_per_step": dt_met / K * 1e3,
               "what": f"nDCG/Precision/Recall/HR/MAP/MRR/F1@{k} from the [users, k] index tensor (el_rec_metrics), "
                       f"{int(held.nnz)} held-out interactions",
               "kernels_ms_per_step": {n: v[1] / K for n, v in rep_met.items()}}
        del held, tip, tix

    # ---- the leg's collectives, timed on their own (bytes per rank and call) --------------------------------------------
    coll_rep = []
    if sharded:
        for what, op, nbytes, fn in collectives:
            ms = time_collective(world, dev, fn)
            coll_rep.append({"what": what, "op": op, "bytes": int(nbytes), "ms": ms,
                             "algbw_GBs": nbytes / (ms * 1e-3) / 1e9 if ms > 0 else None,
                             "expected_ms": expected_collective(op, nbytes, world)})

    # ---- fragile users of the last block: rank-k / k+1 gap inside the fp32 re-association bound -----------
    fragile = None
    if world == 1 and not args.force_sharded and hasattr(ops, "fragile_users"):
        fragile = ops.fragile_users(ctx, st.Gu, st.Gi, st.Bi, 0, Ub, k, excl=pos)

    pairs_per_s = world * B * K / dt_train            # B triplets per rank and step
    users_per_s = (world if (topk_by_user or user_sharded) else 1) * Ub * Kt / dt_topk
    traffic, traffic_note = load_traffic(U, I, F, B, Ub, world if not args.force_sharded else 0)

    # train roofline: algorithmic bytes of the dominant kernel (DESIGN.md "algorithmic bytes")
    rows_u, rows_i = int(st.Gu.shape[0]), int(st.Gi.shape[0])  # what this rank's optimiser pass actually covers
    if exchange == "dense":
        rows_u = be.Us
    alg = {
        "k_adam_dense_Gu": 24.0 * rows_u * F,                 # theta, m, v read + write
        "k_adam_rows_Gu": 24.0 * rows_u * F,                  # the same pass reading compact gradient rows
        "k_bpr_user_adam": 24.0 * rows_u * F + B * (8.0 * F + 28.0),   # user segments fused with the Adam pass: theta, m, v of every
        #                                                        user row read + written once, gamma_i / gamma_j gathered per triplet
        "k_adam_dense_Gi": 24.0 * rows_i * F,
        "k_adam_dense3": 24.0 * ((rows_u + rows_i) * F + rows_i),   # small models: the three dense passes share one launch
        "k_bprmf_fwd_bwd": B * (24.0 * F + 28.0),             # 3 rows read + 3 gradient rows written (+ idx, bias)
        "k_bpr_user_seg": B * (16.0 * F + 28.0),              # gamma_u, gamma_i, gamma_j read + dGu row written
        "k_bpr_item_seg": 2.0 * B * (8.0 * F + 12.0),         # gamma_u(b) read + dGi row written, per occurrence
        "k_bpr_triplet_rows": B * (16.0 * F + 28.0),          # 3 rows read, dGu row written
        "k_rows_segsum": world * B * 8.0 * F,                 # gathered row read + reduced row written
        "rocprim_radix_sort_pairs": 3.0 * B * 16.0,
        "k_rows_apply": B * (72.0 * F + 60.0) - B * (24.0 * F + 28.0) if args.opt == "adam_lazy" else B * (24.0 * F),
        "k_bpr_sample": B * 48.0,
    }
    deferred = bool(getattr(st, "deferred", False))
    rows_touched = None
    item_fused = bool(getattr(st, "item_fused", False)) and not sharded
    item_deferred = bool(getattr(st, "item_deferred", False)) and item_fused
    items_touched = None
    if item_fused:
        # fused item side: the segments gather gamma_u per occurrence and move theta, m, v (+ bias) of the batch's distinct items in
        # place; counted on one drawn batch
        tb = ops.bpr_sample(ctx, pos_train, B, seed=4242, first_sample=0)
        items_touched = int(torch.unique(torch.cat([tb[1], tb[2]])).numel())
        alg["k_bpr_item_seg"] = 2.0 * B * (4.0 * F + 12.0) + 24.0 * items_touched * (F + 1)
        alg["k_bpr_flush_items"] = 24.0 * (rows_i - items_touched) * F      # every-step form: the rows the batch left alone, one step each
        alg["k_bpr_catchup_items"] = 24.0 * items_touched * F
        del tb
    if deferred:
        # the user-side kernel reads / writes only the rows of the batch's distinct users (+ one pre-update row each for the item
        # segments); counted on one drawn batch
        rows_touched = int(torch.unique(ops.bpr_sample(ctx, pos_train, B, seed=4242, first_sample=0)[0]).numel())
        alg["k_bpr_user_seg"] = 28.0 * rows_touched * F + B * (8.0 * F + 32.0)    # theta, m, v of the batch's rows read + written, the
        #                                                                        pre-update row written, gamma_i / gamma_j gathered
        alg["k_bpr_catchup"] = 24.0 * rows_touched * F     # the rows it brings up to date (its bound is the replay arithmetic, see valu)
        alg["k_bpr_flush_users"] = 24.0 * rows_u * F
    dn, dsec = dominant(rep_train)
    valu_bound = ("k_bpr_catchup", "k_bpr_flush_users") + (("k_bpr_catchup_items", "k_bpr_flush_items") if item_deferred else ())
    if (deferred or item_deferred) and dn in valu_bound:
        # the replay kernels are bound by the fp32 sqrt / division rate of the VALUs, not by HBM (their figures go under
        # `roofline.valu`): the HBM roofline of the leg is that of its largest bandwidth-bound kernel
        hb = {n: v for n, v in rep_train.items() if n not in valu_bound}
        dn = max(hb, key=lambda n: hb[n][1])
        cnt_, ms_ = getattr(rep_train, "live", {}).get(dn, rep_train[dn])
        if cnt_ == 0:
            cnt_, ms_ = rep_train[dn]
        dsec = ms_ / cnt_ * 1e-3
    step_bytes = 24.0 * (rows_u + rows_i) * F + B * (24.0 * F + 28.0)     # dense-Adam surcharge + per-triplet bytes
    achieved = alg.get(dn, step_bytes) / dsec / 1e9                       # (a kernel without an entry: priced at the whole step's bytes)
    roof_train = {"kernel": dn, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": achieved / HBM_PEAK_GBS, "traffic": traffic.get(dn), "traffic_source": traffic_note,
                  "step_GBs": step_bytes / (dt_train / K) / 1e9,
                  "kernels_ms_per_step": {n: v[1] / K for n, v in rep_train.items()}}
    if item_fused:
        roof_train.update({"item_side": "fused: item segments + Keras Adam on the batch's item rows in place (k_bpr_item_seg, k_bpr_item_split); "
                                        + ("rows outside the batch wait for their gradient-free updates (deferred, replayed when next needed)"
                                           if item_deferred else "rows outside the batch replayed at the end of every step (k_bpr_flush_items)"),
                           "item_rows_per_step": items_touched})
    if deferred:
        item_rows_moved = (items_touched if item_deferred else rows_i) if item_fused else rows_i
        moved = 24.0 * (rows_touched + item_rows_moved) * F + 4.0 * rows_touched * F + B * (24.0 * F + 32.0) + 24.0 * rows_u * F / K
        roof_train.update({
            "replay": args.replay,
            "deferred_decay": f"user rows without triplets in a batch are not moved by that step: their gradient-free Adam updates are taken "
                              + ("in closed form (four row-level sums over the lr_t history, O(1) per element: el_bprmf_state.replay_series; inside "
                                 "the oracle tolerances, tests/test_gpu_fullsize_c4.py)" if args.replay == "series" else "step by step, bit for bit,")
                              + f" when next needed; the {K} timed steps end with the catch-up of every pending row (k_bpr_flush_users)",
            "user_rows_per_step": rows_touched,
            "valu": {"what": ("k_bpr_user_seg and k_bpr_flush_users bring waiting rows forward in closed form: one v_sqrt + one v_rcp + ~10 fp32 "
                              "operations per ELEMENT and 10 scalar operations per ROW and missed step" if args.replay == "series" else
                              "k_bpr_user_seg (a row's missed steps when its segment starts) and k_bpr_flush_users replay the postponed updates in "
                              "registers: one correctly rounded fp32 sqrt and division per element and step, on packed fp32 instructions -- U F "
                              "element-steps per optimiser step in the steady state, whatever B is; the user-segment kernel's HBM fraction "
                              "above is that of a kernel that also carries this arithmetic"),
                     "element_steps_per_step": float(rows_u) * F + (float(rows_i) * F if item_deferred else 0.0),
                     "steady_state": "every row was given a gradient once before the timed region (cover batches): none sits at the m = v = 0 "
                                     "fixed point the replay kernels skip",
                     "flush_ms_per_step": rep_train.get("k_bpr_flush_users", (0, 0.0))[1] / K},
            "step_GBs_note": "step_GBs_moved = the bytes this form has to move (batch rows + 1/K of the final replay) / step time; SURVEY 8d's "
                             "every-row byte count over the same time is kept outside the roofline object as `survey8d_equivalent_GBs` (a work "
                             "equivalent, not a bandwidth: the deferred decay replays those rows in registers instead of moving them)",
            "step_GBs_moved": moved / (dt_train / K) / 1e9})
        roof_train["survey8d_equivalent_GBs"] = roof_train.pop("step_GBs")
    if pipelined:
        # the timed steps overlap the NEXT batch's sampler / prep / sort with this step's kernels: `achieved` is the dominant
        # kernel's duration inside that timed region (it shares HBM with the look-ahead work), the breakdown above and the
        # figure below are the same kernels run back to back
        b2b = alg.get(dn, step_bytes) / (rep_train[dn][1] / rep_train[dn][0] * 1e-3) / 1e9
        roof_train.update({"pipelined": "sampler + prep + sort of step t+1 on a side stream under step t",
                           "achieved_back_to_back": b2b, "frac_back_to_back": b2b / HBM_PEAK_GBS})
    # top-k roofline: the dominant kernel is one full user x item scoring GEMM (2*U*I*F flop per launch): the fp32 MFMA
    # kernel, or one of the two bf16 passes of the screened kernel (the other pass repeats the same flops; results are
    # re-scored in fp32 and bit-identical)
    tn, tsec = dominant(rep_topk)
    flops = 2.0 * Ub * (I if (topk_by_user or user_sharded or not sharded) else (hi - lo)) * F
    ach_t = flops / tsec / 1e12
    screened = tn.startswith("k_screen")
    peak_t = MFMA_BF16_PEAK_TFLOPS if screened else MFMA_F32_PEAK_TFLOPS
    roof_topk = {"kernel": tn, "bound": "mfma", "achieved": ach_t, "peak": peak_t, "unit": "TFLOP/s",
                 "frac": ach_t / peak_t, "traffic": traffic.get(tn), "traffic_source": traffic_note,
                 "dtype": "bf16 MFMA screen + f32 exact re-score" if screened else "f32",
                 **({"measured_ceiling_random_operands": MFMA_BF16_RANDOM_TFLOPS, "frac_of_measured_ceiling": ach_t / MFMA_BF16_RANDOM_TFLOPS}
                    if screened else {}),
                 "effective_TFLOPs": flops / (dt_topk / Kt) / 1e12,
                 "kernels_ms_per_step": {n: v[1] / Kt for n, v in rep_topk.items()}}
    if not sharded:
        par = "single"
    elif exchange == "user":
        par = (f"user-shard x{world}: train = {B} triplets/rank for the rank's own users (item table replicated) + "
               + (f"all-reduce of the item gradients ({I * (F + 1) * 4 / 1e6:.0f} MB)" if item_ex == "dense" else
                  f"all-gather of the ranks' touched item-gradient rows (~{n_touch} records of {F + 2} words per rank)"))
    else:
        par = (f"item-shard x{world}: train = {B} triplets/rank + "
               + ("reduce-scatter of the dense user-gradient table, optimiser on U/G user rows, all-gather of the rows"
                  if exchange == "dense" else "all-gather of user-gradient rows") + " (weak)")
    topk_sharding = ("single" if not sharded else
                     f"by user: each rank scores {Ub}-user blocks of the users it owns against its replica of the item table, no collective"
                     if user_sharded else
                     f"by user: {Ub} users per rank and step vs the whole catalogue (item table all-gathered once per evaluation)"
                     if topk_by_user else f"by item: all users vs I/{world} items per rank + all-gather/merge of partial lists")
    res = {
        "value": pairs_per_s, "unit": "pairs/s", "ms_per_step": dt_train / K * 1e3, "repeats_ms_per_step": rep_train.repeats_ms,
        "scaling": "weak",
        "parallelism": par, "loss_per_pair_last": loss / (B * world * max(rep_train.calls, 1)), "roofline": roof_train,
        "topk": {"value": users_per_s, "unit": "users/s", "ms_per_step": dt_topk / Kt * 1e3, "repeats_ms_per_step": rep_topk.repeats_ms, "steps": Kt,
                 "scaling": "weak" if (topk_by_user or user_sharded or not sharded) else "strong",
                 "sharding": topk_sharding, "roofline": roof_topk},
        "interactions": int(pos.nnz), "topk_block": Ub,
    }
    if scr0 is not None and scr0["users"]:
        res["topk"]["screen"] = {"train_steps_before": steps_so_far, "records_per_user": scr0["records_per_user"], "fallback_users": scr0["fallback_users"]}
    if trained is not None:
        res["topk"]["trained"] = trained
    if fragile is not None:
        res["topk"]["fragile_users"] = fragile
    if f32_entry is not None:
        res["topk"]["fp32_mfma"] = f32_entry
    if coll_rep:
        res["collectives"] = coll_rep
    if met is not None:
        res["metrics"] = met
    if keep_host:
        res["_host"] = {"Gu": st.Gu.cpu().numpy(), "Gi": st.Gi.cpu().numpy(), "Bi": st.Bi.cpu().numpy(),
                        "indptr": pos.indptr.cpu().numpy(), "indices": pos.indices.cpu().numpy()}
    return res


# ---------------------------------------------------------------------------------------------------------------------
# batch-size sweep of the headline (B in {4 096, 65 536, 1 048 576}; the reference's default batch_size is 512)
# ---------------------------------------------------------------------------------------------------------------------
def sweep_leg(args, ctx, data):
    """pairs/s of the training step at B in {4096, 65536, 2^20} for the reference's optimiser semantics (adam_tf_dense: Keras'
    sparse apply moves EVERY row each step, 24 (U + I) F bytes whatever B is) and for the touched-rows-only Adam (adam_lazy: a
    documented deviation, what a user who does not need TF's semantics would run).  Each point is one el_bprmf_train_loop call
    (sampler + step per batch launched back to back inside the library, the plugin's fused-epoch path) of `steps` batches,
    median of --repeats, after cover batches that gave every user row a gradient (steady state of the deferred decay)."""
    from elliot_amd import ops
    dev = ctx.device
    U, I, F = args.users, args.items, args.factors
    pos = data["pos"]
    lr, l_w, l_b = 0.001, 0.1, 0.001
    out = []
    for opt in ("adam_tf_dense", "adam_lazy"):
        g = torch.Generator(device=dev)
        g.manual_seed(42)
        lim_u, lim_i = (6.0 / (U + F)) ** 0.5, (6.0 / (I + F)) ** 0.5
        st = ops.BprmfDeviceState(ctx, (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * lim_u,
                                  (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * lim_i, torch.zeros(I, device=dev), optimizer=opt)
        drawn = 0
        if opt == "adam_tf_dense":
            # steady state of the deferred points: every user row gets a gradient once before anything is timed (round-3 sweeps
            # started from rows at the m = v = 0 fixed point and climbed 0.170 -> 0.212 ms at B = 4 096 as the rows woke up)
            cover_batches(st, data["indptr"], data["indices"], U, 0, U, I, min(U, 1 << 20), lr, l_w, l_b)
        for B in (4096, 65536, 1 << 20):
            steps = args.steps * (1 if B >= (1 << 20) else 4)
            st.train_loop(pos, args.warmup * B, B, 42, drawn, lr, l_w, l_b)
            drawn += args.warmup * B
            dts = []
            for _ in range(REPEATS):
                barrier(1)
                t0 = time.perf_counter()
                st.train_loop(pos, steps * B, B, 42, drawn, lr, l_w, l_b)
                st.sync()                                    # deferred decay: the pending row updates belong to these steps
                barrier(1)
                dts.append(time.perf_counter() - t0)
                drawn += steps * B
            dt = sorted(dts)[len(dts) // 2]
            dense_bytes = 24.0 * (U + I) * F if opt == "adam_tf_dense" else 0.0
            step_bytes = dense_bytes + B * ((24.0 if opt == "adam_tf_dense" else 72.0) * F + 28.0)
            out.append({"optimizer": opt, "batch": B, "steps": steps, "value": B * steps / dt, "unit": "pairs/s",
                        "ms_per_step": dt / steps * 1e3, "repeats_ms_per_step": [d / steps * 1e3 for d in dts],
                        "step_GBs_algorithmic": step_bytes / (dt / steps) / 1e9,
                        **({"deferred_decay": True} if getattr(st, "deferred", False) else {})})
        st.pop_loss()
        del st
        torch.cuda.empty_cache()
    return {"what": "el_bprmf_train_loop (sampler + step per batch, launched inside the library), U x I x F of the headline leg; "
                    "adam_tf_dense = the reference's Keras semantics (every row of the tables moves each step), adam_lazy = touched rows only",
            "points": out}


# ---------------------------------------------------------------------------------------------------------------------
# Mult-VAE leg (BASELINE configs[2]) and NeuMF leg (configs[3], per-GPU shape)
# ---------------------------------------------------------------------------------------------------------------------
def vae_leg(args, ctx):
    """multi_vae_model.py:125-142 train_step at the ML-20M shape: users/s + MFMA roofline of the fp32 GEMM kernel."""
    from elliot_amd import ops
    from elliot_amd.synthetic import zipf_csr_device
    dev = ctx.device
    U, I, H, L, B = (int(x) for x in args.vae_shape.split(","))
    K, W = args.steps, args.warmup
    ip, ix = zipf_csr_device(U, I, dev, mean_log=4.5, sigma_log=1.0, dmin=20, dmax=3000, seed=5)
    csr = ops.DeviceCSR.from_tensors(ip, ix, I)
    g = torch.Generator(device=dev)
    g.manual_seed(7)
    gn = lambda a, b: torch.randn((a, b), generator=g, device=dev) * (2.0 / (a + b)) ** 0.5       # GlorotNormal scale
    z = lambda n: torch.zeros(n, device=dev)
    st = ops.VaeDeviceState(ctx, {"W1": gn(I, H), "b1": z(H), "Wm": gn(H, L), "bm": z(L), "Wv": gn(H, L), "bv": z(L),
                                  "W3": gn(L, H), "b3": z(H), "W4": gn(H, I), "b4": z(I)}, max_batch=B)
    perm = torch.randperm(U, device=dev, generator=g).to(torch.int32)
    nb = U // B
    eps = torch.randn((B, L), device=dev, generator=g)
    it = [0]

    def step():
        b = it[0] % nb
        it[0] += 1
        st.train_step(csr, perm[b * B:(b + 1) * B], 0.001, min(0.2, it[0] / 200000.0), eps=eps)

    def step_one_stream():
        # the per-kernel breakdown: the same kernels back to back on one stream (the step proper runs the weight-gradient products, the
        # column sums and the index of the sparse first-layer gradient on the library's second stream: elapsed times of kernels that
        # overlap mean nothing)
        with ctx.option("vae_side", 0):
            step()

    dt, rep = timed(ctx, 1, step, W, K, events_in_timed_region=False, fn_breakdown=step_one_stream)
    loss = st.pop_loss()
    ms = dt / K * 1e3
    nnz_row = float(csr.nnz) / U
    # flops executed by the dense GEMM launches of one step (the first layer and its weight gradient run on the CSR rows):
    #   fwd  h->mv (H x 2L), z->h2 (L x H), h2->logits (H x I);  bwd: two products each
    gemm_flops = B * 2.0 * (H * 2 * L + L * H + H * I) * 3
    alg_flops = B * (2400.0 * I + 720000.0) * 2.5                      # dense-input formulation, fwd x 2.5
    vtraffic, vnote = (load_traffic(0, 0, 0, 0, 0, 1, leg="vae") if args.vae_shape == "138493,26744,600,200,512" else ({}, "non-default shape"))
    gms, groof = gemm_roofline(rep, gemm_flops, K)
    return {"value": B * K / dt, "unit": "users/s", "ms_per_step": ms,
            "workload": f"MultiVAE {U} users x {I} items (ML-20M shape, BASELINE configs[2]), hidden {H}, latent {L}, batch {B}, "
                        f"{int(csr.nnz)} interactions ({nnz_row:.0f}/user), Adam, anneal schedule of multi_vae.py:105-108",
            "loss_mean": loss / max(rep.calls, 1), "repeats_ms_per_step": rep.repeats_ms,
            "roofline": {**groof, "traffic": vtraffic.get("k_gemm_per_step", vtraffic.get("k_gemm_f32_per_step")),
                         "traffic_unit": "HBM bytes of the GEMM launches of ONE step", "traffic_source": vnote,
                         "flops_per_step_gemm": gemm_flops, "gemm_ms_per_step": gms,
                         "step_TFLOPs_dense_gemm": gemm_flops / (ms * 1e-3) / 1e12,
                         "step_TFLOPs_survey8d": alg_flops / (ms * 1e-3) / 1e12,
                         "kernels_ms_per_step": {n: v[1] / K for n, v in rep.items()}}}


def graph_leg(args, ctx):
    """SURVEY 8f N3 siblings with kernels of their own: LightGCN (one propagation of the normalised adjacency -- the CSR x dense gather
    kernel, HBM-bound -- + the bias-free BPR head) at BASELINE configs[1]'s users x items, F = 64 (the reference's default width,
    LightGCN.py:72), n_layers = 2; and MF2020's strictly sequential fp64 SGD (samples/s of one chain)."""
    from elliot_amd import ops
    from elliot_amd.synthetic import zipf_csr_device
    dev = ctx.device
    U, I, F, L = (int(x) for x in args.graph_shape.split(","))
    K, W = args.steps, args.warmup
    B = args.batch
    ip, ix = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=1234)
    pos = ops.DeviceCSR.from_tensors(ip, ix, I)
    lip, lix, lv = ops.normalized_bipartite_laplacian_device(ip, ix, U, I)
    graph = ops.GraphCSR(ctx, lip, lix, lv, U, F)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    Gu = (torch.rand((U, F), generator=g, device=dev) * 2 - 1) * 0.05      # (the reference starts at zero and stays there: injected tables)
    Gi = (torch.rand((I, F), generator=g, device=dev) * 2 - 1) * 0.05
    st = ops.LightGcnDeviceState(ctx, Gu, Gi, graph, n_layers=L)
    drawn = [0]

    def step():
        t = ops.bpr_sample(ctx, pos, B, seed=42, first_sample=drawn[0])
        drawn[0] += B
        st.train_step(t[0], t[1], t[2], 0.0005, 0.1)

    dt, rep = timed(ctx, 1, step, W, K)
    loss = st.pop_loss()
    nnz, N = graph.nnz, U + I
    spmm_bytes = nnz * (8.0 + 4.0 * F) + N * 4.0 * F * 2                   # per layer: index + value + one gathered row per non-zero; a row
    #                                                                         written (and the layer-combination operand read) per node
    cnt, ms = getattr(rep, "live", {}).get("k_spmm_csr", rep.get("k_spmm_csr", (0, 0.0)))
    if not cnt:
        cnt, ms = rep.get("k_spmm_csr", (1, 0.0))
    sec = ms / max(cnt, 1) * 1e-3
    out = {"lightgcn": {"value": B * K / dt, "unit": "pairs/s", "ms_per_step": dt / K * 1e3, "repeats_ms_per_step": rep.repeats_ms,
                        "loss_per_pair_last": loss / (B * max(rep.calls, 1)),
                        "workload": f"LightGCN {U} users x {I} items, F={F}, n_layers={L}, {nnz} non-zeros of the normalised adjacency, B={B}",
                        "roofline": {"kernel": "k_spmm_csr", "bound": "hbm", "achieved": spmm_bytes / sec / 1e9 if sec > 0 else None,
                                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": spmm_bytes / sec / 1e9 / HBM_PEAK_GBS if sec > 0 else None,
                                     "traffic": None, "bytes_per_launch": spmm_bytes,
                                     "hbm_compulsory_bytes_per_launch": nnz * 8.0 + N * 4.0 * F * 3,
                                     "note": "bytes_per_launch counts one gathered operand row per non-zero (the algorithmic unit of a CSR x dense "
                                             "product, as the BPR kernels count a gathered row per triplet); the operand itself "
                                             f"({N * 4.0 * F / 1e6:.0f} MB) is about the size of the 256 MiB Infinity Cache, so most of those rows are served "
                                             "on-die: `achieved` is a gather rate, the HBM-side traffic is near hbm_compulsory_bytes_per_launch "
                                             "(index + value per non-zero, operand read once, result written, combination operand read)",
                                     "kernels_ms_per_step": {n: v[1] / K for n, v in rep.items()}}}}
    del st, graph, Gu, Gi
    torch.cuda.empty_cache()
    # MF2020: one chain of point-wise SGD updates (MF_model.py:80-113), F = 64, uniform samples over the same users x items
    Um, Im = min(U, 1_000_000), min(I, 100_000)
    n = 400_000
    P = torch.randn((Um, 64), generator=g, device=dev, dtype=torch.float64) * 0.1
    Q = torch.randn((Im, 64), generator=g, device=dev, dtype=torch.float64) * 0.1
    mf = ops.Mf2020DeviceState(ctx, P, Q, lr=0.05, reg=0.0025)
    smp = torch.stack([torch.randint(0, Um, (n,), generator=g, device=dev), torch.randint(0, Im, (n,), generator=g, device=dev),
                       torch.randint(0, 2, (n,), generator=g, device=dev)], 1).to(torch.int32)
    mf.train(smp[:20000])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    mf.train(smp)
    torch.cuda.synchronize()
    dtm = time.perf_counter() - t0
    out["mf2020"] = {"value": n / dtm, "unit": "samples/s", "us_per_sample": dtm / n * 1e6, "loss_per_sample": mf.pop_loss() / (n + 20000),
                     "workload": f"MF2020 sequential fp64 SGD, {n} samples over {Um} users x {Im} items, F=64 (one chain: every sample updates the global bias)",
                     "note": "bound by the chain's latency (two LDS row reads, one 64-lane fp64 reduction, one fp64 exp and one division per sample on one wave; the loss terms are computed off the chain), not by a roofline"}
    return out


def neumf_leg(args, ctx):
    """neural_matrix_factorization_model.py:96-106 train_step, d=128, tower (4F, 2F, F): samples/s + MFMA roofline."""
    from elliot_amd import ops
    from elliot_amd.synthetic import zipf_csr_device
    dev = ctx.device
    U, I, F, B = (int(x) for x in args.neumf_shape.split(","))
    K, W = args.steps, args.warmup
    ip, ix = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=11)
    pos = ops.DeviceCSR.from_tensors(ip, ix, I)
    g = torch.Generator(device=dev)
    g.manual_seed(3)
    gu = lambda a, b: (torch.rand((a, b), generator=g, device=dev) * 2 - 1) * (6.0 / (a + b)) ** 0.5    # GlorotUniform
    units = [4 * F, 2 * F, F]
    w = {"Umf": gu(U, F), "Imf": gu(I, F), "Umlp": gu(U, F), "Imlp": gu(I, F), "W": [], "b": []}
    kin = 2 * F
    for n_out in units:
        w["W"].append(gu(kin, n_out))
        w["b"].append(torch.zeros(n_out, device=dev))
        kin = n_out
    w["hw"], w["hb"] = gu(F + units[-1], 1)[:, 0].contiguous(), torch.zeros(1, device=dev)
    st = ops.NmfDeviceState(ctx, w, max_batch=B, replay=getattr(args, "replay", "series"))
    del w
    it = [0]

    # the sampler never reads the model (pointwise_pos_neg_sampler.py:26-50): batch t+1 COULD be drawn, and its (embedding row, sample)
    # keys ordered (el_nmf_presort), on a side stream under step t as in the BPR leg (elliot_amd/pipeline.py: --neumf-prefetch).
    # Measured and OFF: the sort's dozen small kernels land between the tower's products, whose grids are sized to the machine, and
    # cost them a second round of workgroups -- 5.13 -> 6.5 ms per step (scripts/exp/nmf_ab.sh); the step draws and sorts in line
    pipe = PrefetchPointwise(ctx, pos, B, seed=3, enabled=bool(getattr(args, "neumf_prefetch", False)), presort_state=st)

    def step():
        (u, i, y), b = pipe.next()
        it[0] += 1
        st.train_step(u, i, y, 0.001)
        pipe.release(b)

    # the timed region ends with st.sync(): under the deferred decay (el_nmf_state.row_last) the postponed every-row updates of
    # the K steps are replayed there -- every (element, step) update of Keras' Adam is inside the timed region
    def step_one_stream():
        # the per-kernel breakdown: the same kernels on one stream (the step proper runs the tower's weight-gradient products on the library's
        # second stream beside the embedding kernels: elapsed times of overlapping kernels mean nothing)
        with ctx.option("nmf_side", 0):
            step()

    dt, rep = timed(ctx, 1, step, W, K, finish=st.sync, events_in_timed_region=False, fn_breakdown=step_one_stream)
    loss = st.pop_loss()
    ms = dt / K * 1e3
    # ---- full-catalogue scoring + top-k: el_nmf_score_topk on a block of users against the leg's whole catalogue.
    # 20 F^2 flop per (user, item) pair after the separable first layer (2 (H1 H2 + H2 H3)): inherently ~10^4 x the work of a dot
    # product, so the block is small and the step count its own
    nu = int(args.neumf_topk_users)
    ntraffic, nnote = (load_traffic(0, 0, 0, 0, 0, 1, leg="neumf") if args.neumf_shape == "1250000,1000000,128,262144" else ({}, "non-default shape"))
    tk = None
    if nu > 0 and st.fused_supported(args.k + 2):
        ks, ws_ = max(1, min(K, 2)), 1
        blk = [0]

        def topk_step():
            s0 = (blk[0] * nu) % max(U - nu, 1)
            blk[0] += 1
            st.recommend(s0, s0 + nu, args.k, excl=pos, items_unchanged=blk[0] > 1)

        dt_k, rep_k = timed(ctx, 1, topk_step, ws_, ks)
        pairs, fell_back = st.screen_stats()
        pair_flops = 2.0 * (units[0] * units[1] + units[1] * units[2])

        def kernel_tflops(name):
            c, ms_k = rep_k.live.get(name, rep_k.get(name, (1, 0.0)))
            sec = ms_k / max(c, 1) * 1e-3
            return pair_flops * nu * I / sec / 1e12 if sec > 0 else 0.0

        # the same block through the fp32 kernel alone (screen=False): what the screened route is measured against
        def exact_step():
            st.score_topk_logits(0, nu, args.k + 2, excl=pos, items_unchanged=True, screen=False)

        dt_x, rep_x = timed(ctx, 1, exact_step, 0, 1)
        xc, xms = rep_x.live.get("k_nmf_score", rep_x.get("k_nmf_score", (1, 0.0)))
        ach_x = pair_flops * nu * I / (xms / max(xc, 1) * 1e-3) / 1e12 if xms > 0 else 0.0
        screened = "k_nmf_screen" in rep_k and not fell_back
        ach_k = kernel_tflops("k_nmf_screen") if screened else kernel_tflops("k_nmf_score")
        peak_k = MFMA_BF16_PEAK_TFLOPS if screened else MFMA_F32_PEAK_TFLOPS     # (f16 and bf16 share the dense rate)
        tk = {"value": nu * ks / dt_k, "unit": "users/s", "ms_per_step": dt_k / ks * 1e3, "repeats_ms_per_step": rep_k.repeats_ms,
              "steps": ks, "users_per_step": nu,
              "what": f"NeuMF get_recs + get_top_k (neural_matrix_factorization_model.py:119-148) of {nu} users x {I} items, k={args.k}: "
                      f"el_nmf_score_topk (layer 1 separable; EL_NMF_SCREEN: layers 2-3 on the half-precision matrix instruction with a "
                      f"per-pair error bound, then layers 2-3 + head on fp32 MFMA for the pairs that can still reach the list; selection "
                      f"fused; lists and logit bits are the fp32 kernel's) + sigmoid link + re-rank; the reference's route materialises "
                      f"{nu} x {I} x {4 * F} activations",
              "screen": {"used": bool(screened), "exact_pairs": pairs, "exact_pairs_frac": pairs / float(nu * I), "fell_back": bool(fell_back)},
              "unscreened": {"ms_per_step": dt_x * 1e3, "users_per_s": nu / dt_x, "k_nmf_score_TFLOPs": ach_x,
                             "k_nmf_score_frac_of_f32_mfma_peak": ach_x / MFMA_F32_PEAK_TFLOPS},
              "roofline": {"kernel": "k_nmf_screen" if screened else "k_nmf_score", "bound": "mfma", "achieved": ach_k, "peak": peak_k,
                           "unit": "TFLOP/s", "frac": ach_k / peak_k,
                           "traffic": ntraffic.get("k_nmf_screen" if screened else "k_nmf_score") if nu == 128 else None,
                           "traffic_source": nnote, "dtype": "f16" if screened else "f32",
                           "flops_per_pair": pair_flops, "flops_per_pair_reference_form": 2.0 * (2 * F * units[0] + units[0] * units[1] + units[1] * units[2]),
                           "kernels_ms_per_step": {n: v[1] / ks for n, v in rep_k.items()}}}
        # ---- the same scoring step on a TRAINED network: the share of pairs the bound leaves to the exact kernel grows as the weights
        #      move away from their initialisation (0.003 % -> 31 % over 3 000 steps); the line reports both ends
        n_tr = int(getattr(args, "neumf_trained_steps", 0))
        if n_tr > 0:
            for _ in range(n_tr):
                step()
            st.sync()
            st.pop_loss()
            st._screen_skip = 0                                          # (the default policy gets a fresh try on the trained weights)
            blk[0] = 0
            dt_t, rep_t = timed(ctx, 1, topk_step, 1, ks)
            pairs_t, fb_t = st.screen_stats()
            scr_t = "k_nmf_screen" in rep_t and not fb_t
            c_t, ms_t = rep_t.live.get("k_nmf_screen", rep_t.get("k_nmf_screen", (1, 0.0)))
            tk["trained"] = {"train_steps_before": int(rep.calls) + n_tr, "value": nu * ks / dt_t, "unit": "users/s", "ms_per_step": dt_t / ks * 1e3,
                             "screen": {"used": bool(scr_t), "exact_pairs": pairs_t, "exact_pairs_frac": pairs_t / float(nu * I), "fell_back": bool(fb_t)},
                             "k_nmf_screen_TFLOPs": (pair_flops * nu * I / (ms_t / max(c_t, 1) * 1e-3) / 1e12) if ms_t > 0 else None,
                             "kernels_ms_per_step": {n: v[1] / ks for n, v in rep_t.items()}}
            tk["screen"]["train_steps_before"] = int(rep.calls)
    mlp_flops = B * (36.0 * F * F + 4.0 * F) * 3                       # fwd 36 F^2 + 4 F per sample, x3 fwd + bwd
    gms, groof = gemm_roofline(rep, mlp_flops, K)
    emb_bytes = 24.0 * 2 * (U + I) * F                                 # Keras Adam moves every row of the 4 embedding tables
    ams = sum(v[1] for n, v in rep.items() if n.startswith("k_adam_dense")) / K
    # the embedding side: sort of the batch's (row, sample) keys + the two segment passes (+ the long-segment / tail kernels) + the flush
    seg_names = ("k_nmf_keys", "rocprim_radix_sort_pairs", "k_nmf_seg_fwd", "k_nmf_seg_fwd_tail", "k_nmf_seg_bwd", "k_nmf_seg_bwd_long", "k_nmf_flush_rows")
    rows_ms = sum(v[1] for n, v in rep.items() if n in seg_names) / K
    gemm_names = ("k_gemm_b3", "k_gemm_f32", "k_gemm_reduce")
    non_gemm_ms = sum(v[1] for n, v in rep.items() if n not in gemm_names and n != "k_pw_sample") / K
    return {"value": B * K / dt, "unit": "samples/s", "ms_per_step": ms,
            "workload": f"NeuMF d={F} (GMF + MLP {units}), {U} users x {I} items = the per-GPU shape of BASELINE configs[3] (10M x 1M over "
                        f"8 GPUs) under user sharding, batch {B}, point-wise sampler on the device, Adam (Keras semantics: every row of "
                        f"the four embedding tables moves at every step"
                        + (f"; rows without a gradient are brought forward when next needed ({'in closed form: el_nmf_state.replay_series' if st.replay == 'series' else 'replayed bit for bit'}), "
                           f"the {K}-step region ends with the replay of all of them)" if st.deferred else ", one dense pass per table and step)"),
            "replay": st.replay,
            "loss_mean": loss / max(rep.calls, 1), "repeats_ms_per_step": rep.repeats_ms,
            **({"topk": tk} if tk is not None else {}),
            "roofline": {**groof, "traffic": ntraffic.get("k_gemm_per_step", ntraffic.get("k_gemm_f32_per_step")),
                         "traffic_unit": "HBM bytes of the GEMM launches of ONE step", "traffic_source": nnote,
                         "flops_per_step_mlp": mlp_flops, "gemm_ms_per_step": gms,
                         "step_TFLOPs_mlp": mlp_flops / (ms * 1e-3) / 1e12,
                         "adam_tables_GBs": emb_bytes / (ams * 1e-3) / 1e9 if ams > 0 else None,
                         "embedding_rows_ms_per_step": rows_ms if st.deferred else None,
                         "non_gemm_ms_per_step": non_gemm_ms,
                         "embedding_step_equivalent_GBs": emb_bytes / (rows_ms * 1e-3) / 1e9 if (st.deferred and rows_ms > 0) else None,
                         "kernels_ms_per_step": {n: v[1] / K for n, v in rep.items()}}}


# ---------------------------------------------------------------------------------------------------------------------
# plugin level: external.BPRMF_batch through RecMixin.train() + evaluate() at the headline shape
# ---------------------------------------------------------------------------------------------------------------------
def plugin_e2e_leg(args, ctx, data):
    """One epoch of `external.BPRMF_batch` driven exactly as ModelCoordinator.single drives a model (model_coordinator.py:62-65:
    cls(data=, config=, params=) -> train()), loaded the way elliot/run.py:67-75 loads an external model: RecMixin.train() =
    the sampler's `transactions` triplets in batches of `batch_size` through the model's train_step / train_epoch, then
    RecMixin.evaluate() = top-k of EVERY user under the train mask + nDCG / Recall on the device (recommender_utils_mixin.py:
    29-61).  The data set is the headline leg's interactions split 80/20 per interaction (every user keeps a train item).
    Timed: the constructor (tables drawn on the device, masks and sampler records shipped), train() -- of which evaluate() is
    timed inside."""
    import importlib.util
    import tempfile
    from types import SimpleNamespace
    from elliot_amd.dataset.dataset import DataSet, default_config
    path = os.path.join(REPO, "elliot_amd", "external", "__init__.py")
    spec = importlib.util.spec_from_file_location("external", path, submodule_search_locations=[os.path.dirname(path)])
    external = importlib.util.module_from_spec(spec)
    sys.modules[spec.name] = external
    spec.loader.exec_module(external)
    cls = getattr(sys.modules["external"], "BPRMF_batch")
    U, I, F = args.users, args.items, args.factors
    t_data = time.perf_counter()
    indptr, indices = data["indptr"].cpu().numpy(), data["indices"].cpu().numpy()
    users = np.repeat(np.arange(U, dtype=np.int64), np.diff(indptr))
    rs = np.random.RandomState(7)
    held = rs.random_sample(indices.shape[0]) < 0.2
    held[indptr[:-1]] = False                                         # every user keeps at least one train interaction
    ones = np.ones(indices.shape[0], dtype=np.float32)
    out = tempfile.mkdtemp(prefix="el_bench_")
    cfg = default_config(top_k=args.k, cutoffs=[args.k], simple_metrics=["nDCG", "Recall"], out_dir=out)
    for pth in (cfg.path_output_rec_result, cfg.path_output_rec_weight):
        os.makedirs(pth, exist_ok=True)
    tr = ~held
    ds = DataSet(cfg, (users[tr], indices[tr].astype(np.int64), ones[tr]), (users[held], indices[held].astype(np.int64), ones[held]),
                 public_users=np.arange(U, dtype=np.int64), public_items=np.arange(I, dtype=np.int64))
    t_data = time.perf_counter() - t_data
    params = SimpleNamespace(meta=SimpleNamespace(save_recs=False, verbose=False, save_weights=False), epochs=2, seed=42,
                             factors=F, lr=0.001, l_w=0.1, l_b=0.001, batch_size=args.batch)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    m = cls(data=ds, config=cfg, params=params)
    torch.cuda.synchronize()
    t_init = time.perf_counter() - t0
    evals = []
    ev = m.evaluate

    def timed_eval(*a, **k):
        torch.cuda.synchronize()
        t = time.perf_counter()
        ev(*a, **k)
        torch.cuda.synchronize()
        evals.append(time.perf_counter() - t)
    m.evaluate = timed_eval
    t0 = time.perf_counter()
    m.train()
    torch.cuda.synchronize()
    t_train = (time.perf_counter() - t0 - sum(evals)) / 2
    spent = {"eval": evals[-1]}                                       # the second evaluate(): steady state (the first one also ships the
    #                                                                  held-out CSR / masks and loads the kernels)
    res = m.get_results()[args.k]["test_results"]
    del sys.modules["external"]
    steps = -(-ds.transactions // args.batch)
    return {"what": f"external.BPRMF_batch (elliot/run.py:67-75 loading) -> RecMixin.train(), 2 epochs: an epoch = {ds.transactions} triplets in "
                    f"{steps} steps of batch_size {args.batch} + evaluate(): top-{args.k} of all {U} users under the train mask, "
                    f"nDCG / Recall on the device against {int(held.sum())} held-out interactions",
            "train_epoch_s": t_train, "train_pairs_per_s": ds.transactions / t_train, "evaluate_s": spent["eval"],
            "evaluate_first_call_s": evals[0],
            "evaluate_users_per_s": U / spent["eval"] if spent["eval"] > 0 else None,
            "constructor_s": t_init, "dataset_build_host_s": t_data,
            "constructor_note": "tables drawn in HBM (GlorotUniform distribution), train CSR / sampler records shipped once",
            "nDCG": res.get("nDCG"), "Recall": res.get("Recall")}


# ---------------------------------------------------------------------------------------------------------------------
# output: ONE compact JSON line on stdout (the driver's consumer reads a bounded tail), the full per-leg objects in a side file
# ---------------------------------------------------------------------------------------------------------------------
LINE_LIMIT = 6144          # bytes; tests/test_bench_line.py holds the line to it


def _r(x, sig=6):
    """Floats to `sig` significant digits (ints, strings, None unchanged); non-finite floats become null (strict JSON)."""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None
        return float(f"{x:.{sig}g}")
    if isinstance(x, dict):
        return {k: _r(v, sig) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_r(v, sig) for v in x]
    return x


def _roof(r, extra=()):
    """The contract's roofline object (+ named extras), nothing else."""
    if not r:
        return None
    keys = ("kernel", "bound", "achieved", "peak", "unit", "frac", "traffic") + tuple(extra)
    return {k: r.get(k) for k in keys if k in r or k in ("traffic",)}


def _topk_summary(t):
    if not t:
        return None
    out = {"users_per_s": t.get("value"), "ms_per_block": t.get("ms_per_step"), "roofline": _roof(t.get("roofline"), ("effective_TFLOPs",))}
    if "fragile_users" in t:
        out["fragile_users"] = t["fragile_users"].get("fragile")
    if "fp32_mfma" in t:
        out["fp32_kernel_users_per_s"] = t["fp32_mfma"].get("value")
        out["fp32_kernel_frac"] = (t["fp32_mfma"].get("roofline") or {}).get("frac")
    if "screen" in t and "records_per_user" in t["screen"]:
        out["records_per_user"] = t["screen"]["records_per_user"]
        out["train_steps_before"] = t["screen"].get("train_steps_before")
    if "trained" in t:
        tr = t["trained"]
        out["trained"] = {k: tr.get(k) for k in ("train_steps_before", "value", "ms_per_step", "records_per_user", "fallback_users", "kernel_frac")}
    return out


def _bpr_summary(leg):
    if not leg:
        return None
    out = {"pairs_per_s": leg.get("value"), "ms_per_step": leg.get("ms_per_step"),
           "roofline": _roof(leg.get("roofline"), ("step_GBs_moved", "frac_back_to_back")), "topk": _topk_summary(leg.get("topk"))}
    if "metrics" in leg:
        out["metrics_users_per_s"] = leg["metrics"].get("value")
    if "collectives" in leg:
        out["collectives"] = [{"op": c["op"], "MB": c["bytes"] / 1e6, "ms": c["ms"], "expected_ms": c.get("expected_ms")} for c in leg["collectives"]]
    if "parallelism" in leg:
        out["parallelism"] = leg["parallelism"][:160]
    return out


def compact_line(full):
    """The stdout line: the contract's fields + one-number summaries of the secondary legs.  Everything else stays in `full`."""
    line = {k: full[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                 "vs_baseline", "dtype", "data", "topk_users_per_s", "topk_ms_per_block", "topk_frac") if k in full}
    cfg = full.get("config", {})
    line["config"] = {k: cfg[k] for k in ("workload", "users", "items", "factors", "interactions", "batch", "batch_per_gpu", "optimizer", "replay",
                                          "topk_block", "k", "parallelism", "world_size_observed", "backend", "collectives_through")
                      if cfg.get(k) is not None}
    if "workload" in line["config"]:
        line["config"]["workload"] = line["config"]["workload"][:120]
    if "parallelism" in line["config"]:
        line["config"]["parallelism"] = line["config"]["parallelism"][:200]
    line["loss_per_pair_last"] = full.get("loss_per_pair_last")
    line["roofline"] = _roof(full.get("roofline"), ("step_GBs_moved", "frac_back_to_back"))
    line["topk"] = _topk_summary(full.get("topk"))
    if "collectives" in full:
        line["collectives"] = [{"op": c["op"], "MB": c["bytes"] / 1e6, "ms": c["ms"], "expected_ms": c.get("expected_ms")}
                               for c in full["collectives"]]
    if "item_shard" in full:
        line["item_shard"] = _bpr_summary(full["item_shard"])
    legs = {}
    if "replay_other" in full:
        ro = full["replay_other"]
        legs["replay_" + ro["replay"]] = {"pairs_per_s": ro["value"], "ms_per_step": ro["ms_per_step"]}
    if "c2" in full:
        legs["c2"] = dict(_bpr_summary(full["c2"]), workload="BPRMF d=128, 1M users x 100K items (BASELINE configs[1])")
    if "c5_per_gpu" in full:
        legs["c5_per_gpu"] = dict(_bpr_summary(full["c5_per_gpu"]), workload="BPRMF d=256, 6.25M users x 5M items (configs[4] per GPU)")
    if "batch_sweep" in full:
        legs["batch_sweep"] = [{"opt": p["optimizer"], "B": p["batch"], "pairs_per_s": p["value"]} for p in full["batch_sweep"].get("points", [])]
    if "plugin_e2e" in full:
        pe = full["plugin_e2e"]
        legs["plugin_e2e"] = {k: pe.get(k) for k in ("train_pairs_per_s", "evaluate_users_per_s", "nDCG")}
    if "vae" in full:
        v = full["vae"]
        legs["vae"] = {"users_per_s": v.get("value"), "ms_per_step": v.get("ms_per_step"),
                       "roofline": _roof(v.get("roofline"), ("gemm_ms_per_step",)), "workload": "MultiVAE ML-20M shape (configs[2]), B=512"}
    if "neumf" in full:
        n = full["neumf"]
        nt = n.get("topk") or {}
        legs["neumf"] = {"samples_per_s": n.get("value"), "ms_per_step": n.get("ms_per_step"), "replay": n.get("replay"),
                         "roofline": _roof(n.get("roofline"), ("gemm_ms_per_step",)),
                         "topk": {"users_per_s": nt.get("value"), "ms_per_step": nt.get("ms_per_step"),
                                  "roofline": _roof(nt.get("roofline")), "survivor_frac": (nt.get("screen") or {}).get("exact_pairs_frac"),
                                  "train_steps_before": (nt.get("screen") or {}).get("train_steps_before"),
                                  **({"trained": {"train_steps_before": nt["trained"].get("train_steps_before"),
                                                  "users_per_s": nt["trained"].get("value"), "ms_per_step": nt["trained"].get("ms_per_step"),
                                                  "survivor_frac": nt["trained"]["screen"].get("exact_pairs_frac"),
                                                  "fell_back": nt["trained"]["screen"].get("fell_back")}} if "trained" in nt else {})} if nt else None,
                         "workload": "NeuMF d=128, 1.25M users x 1M items (configs[3] per GPU), B=262144"}
    if "graph" in full:
        gl, gm = full["graph"]["lightgcn"], full["graph"]["mf2020"]
        legs["lightgcn"] = {"pairs_per_s": gl.get("value"), "ms_per_step": gl.get("ms_per_step"), "roofline": _roof(gl.get("roofline")),
                            "workload": "LightGCN 1M users x 100K items, F=64, 2 layers"}
        legs["mf2020"] = {"samples_per_s": gm.get("value"), "us_per_sample": gm.get("us_per_sample")}
    if legs:
        line["legs"] = legs
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {"value": cb.get("value"), "unit": cb.get("unit"), "cores": cb.get("cores"), "kind": cb.get("kind"),
                                "host_cpu_count": cb.get("host_cpu_count"), "sample": (cb.get("sample") or "")[:160],
                                "topk": {k: (cb.get("topk") or {}).get(k) for k in ("value", "unit", "cores", "kind")},
                                **({"reference": cb["reference"]} if "reference" in cb else {})}
    line["legs_file"] = full.get("legs_file")
    # every non-null `traffic` in the line comes from the same place: said once
    def _any_traffic(o):
        if isinstance(o, dict):
            return any((k == "traffic" and v is not None) or _any_traffic(v) for k, v in o.items())
        if isinstance(o, list):
            return any(_any_traffic(v) for v in o)
        return False
    if _any_traffic(line):
        line["traffic_source"] = "builder's rocprofv3 PMC passes (profiles/traffic.json, same kernel sources), not this run"
    return _r(line)


def emit(full, args):
    """Full objects -> bench_legs.json (repo root; + gpurun_out/ when present) and stderr; compact line -> stdout (last line)."""
    if getattr(args, "legs_file", None):
        paths = [args.legs_file]
    else:
        paths = [os.path.join(REPO, "bench_legs.json" if full.get("n_gpus", 1) == 1 else f"bench_legs_n{full['n_gpus']}.json")]
        if os.path.isdir(os.path.join(REPO, "gpurun_out")):
            paths.append(os.path.join(REPO, "gpurun_out", os.path.basename(paths[0])))
    full["legs_file"] = os.path.basename(paths[0])
    blob = json.dumps(_r(full, 9))
    for p in paths:
        try:
            with open(p, "w") as fh:
                fh.write(blob + "\n")
        except OSError as ex:
            print(f"bench.py: could not write {p}: {ex}", file=sys.stderr)
    print("bench.py full per-leg report: " + blob, file=sys.stderr, flush=True)
    line = json.dumps(compact_line(full), allow_nan=False, separators=(",", ":"))
    if len(line) > LINE_LIMIT:
        # never grow past what the consumer reads: drop the secondary legs' summaries first, then the collectives
        c = compact_line(full)
        for key in ("legs", "item_shard", "collectives"):
            c.pop(key, None)
            line = json.dumps(c, allow_nan=False, separators=(",", ":"))
            if len(line) <= LINE_LIMIT:
                break
    # the line is the LAST thing on stdout: C-level buffers first (RCCL prints its version banner through stdio, which a pipe holds
    # back until exit -- behind the Python-level print of the line)
    sys.stdout.flush()
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    print(line, flush=True)


def main():
    global REPEATS
    args = parse()
    REPEATS = max(1, args.repeats)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        sys.exit(self_launch(args))
    world, rank, local, backend = dist_setup(args)
    if world != args.gpus:
        # never report an N-GPU line from a different number of ranks
        if rank == 0:
            print(f"bench.py: --gpus {args.gpus} but the process group has {world} rank(s)", file=sys.stderr)
        sys.exit(2)
    from elliot_amd import ops
    from elliot_amd.synthetic import zipf_csr_device
    ctx = ops.get_context(local)
    dev = ctx.device
    torch.cuda.set_device(dev)
    U, I, F, B, k = args.users, args.items, args.factors, args.batch, args.k
    sharded = world > 1 or args.force_sharded
    legs = args.legs.split(",") if args.legs != "auto" else (["bpr", "item_shard"] if world > 1 else
                                                            ["bpr", "c2", "metrics", "sweep", "plugin", "c5", "vae", "neumf", "graph"])

    # ---------------- synthetic inputs, resident in HBM -------------------------------------------
    # headline: north_star's target shape (every rank builds the same CSR and keeps its part)
    indptr, indices = zipf_csr_device(U, I, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=1234 if (U, I) != (10_000_000, 1_000_000) else 4321)
    data = {"indptr": indptr, "indices": indices, "pos": ops.DeviceCSR.from_tensors(indptr, indices, I)}
    if args.comm == "abi" and sharded:
        from elliot_amd import parallel
        data["coll"] = parallel.RcclAbiCollectives(ctx, rank, world)          # RCCL through el_comm_* (one communicator for all legs)

    want_cpu = world == 1 and not args.no_cpu_baseline
    topk_shard = args.topk_shard or ("user" if args.shard == "user" else "item")
    main_leg = bpr_leg(args, ctx, world, rank, data, args.shard, topk_shard, with_metrics="metrics" in legs and "c2" not in legs,
                       keep_host=want_cpu and rank == 0)
    other = None
    if world == 1 and not args.force_sharded and args.opt == "adam_tf_dense" and "bpr" in legs:
        torch.cuda.empty_cache()
        other = replay_other_leg(args, ctx, data, "exact" if args.replay == "series" else "series")
    second = None
    if "item_shard" in legs and sharded and args.shard == "user":
        torch.cuda.empty_cache()
        second = bpr_leg(args, ctx, world, rank, data, "item", args.topk_shard or "item")
    del data, indptr, indices
    torch.cuda.empty_cache()
    sweep = plugin = vae = neumf = c2 = c5 = graph = None
    if world == 1 and not args.force_sharded:
        if "c2" in legs or "sweep" in legs or "plugin" in legs:
            # BASELINE configs[1] (1 M users x 100 K items, d = 128: the headline of rounds 1-3): the same two steps, the every-row fused
            # kernels (B = U: a batch touches most user rows), + the batch sweep and the plugin end-to-end leg on its data
            a2 = argparse.Namespace(**vars(args))
            a2.users, a2.items = (int(x) for x in args.c2_shape.split(","))
            ip2, ix2 = zipf_csr_device(a2.users, a2.items, dev, mean_log=3.9, sigma_log=1.0, dmin=5, dmax=2000, seed=1234)
            d2 = {"indptr": ip2, "indices": ix2, "pos": ops.DeviceCSR.from_tensors(ip2, ix2, a2.items)}
            if "c2" in legs:
                c2 = bpr_leg(a2, ctx, world, rank, d2, "user", "user", with_metrics="metrics" in legs)
                c2["workload"] = f"BPRMF d={F}, synthetic {a2.users} users x {a2.items} items (BASELINE configs[1])"
            if "sweep" in legs:
                sweep = sweep_leg(a2, ctx, d2)
            if "plugin" in legs:
                plugin = plugin_e2e_leg(a2, ctx, d2)
            del d2, ip2, ix2
            torch.cuda.empty_cache()
        if "c5" in legs:
            # BASELINE configs[4] (BPRMF d=256, 50 M users x 5 M items on 8 GPUs) at its PER-GPU shape under user sharding: the rank's
            # 6.25 M user rows + a replica of the 5 M-item table (~ 50 GB of tables and optimiser state); fewer top-k blocks per repeat --
            # a block against 5 M items takes ~0.3 s (the training steps keep K: their timed region ends with the replay of every pending
            # row, which K amortises)
            a5 = argparse.Namespace(**vars(args))
            a5.users, a5.items, a5.factors = (int(x) for x in args.c5_shape.split(","))
            a5.topk_steps = max(2, min(args.steps, 5))
            ip5, ix5 = zipf_csr_device(a5.users, a5.items, dev, mean_log=3.0, sigma_log=1.0, dmin=5, dmax=2000, seed=5432)
            d5 = {"indptr": ip5, "indices": ix5, "pos": ops.DeviceCSR.from_tensors(ip5, ix5, a5.items)}
            c5 = bpr_leg(a5, ctx, world, rank, d5, "user", "user")
            c5["workload"] = (f"BPRMF d={a5.factors}, synthetic {a5.users} users x {a5.items} items = the per-GPU shape of BASELINE configs[4] "
                              f"(50M x 5M x 256 over 8 GPUs) under user sharding; {a5.steps} training steps / {a5.topk_steps} top-k blocks per repeat")
            del d5, ip5, ix5
            torch.cuda.empty_cache()
        if "vae" in legs:
            vae = vae_leg(args, ctx)
            torch.cuda.empty_cache()
        if "neumf" in legs:
            neumf = neumf_leg(args, ctx)
            torch.cuda.empty_cache()
        if "graph" in legs:
            graph = graph_leg(args, ctx)
            torch.cuda.empty_cache()
    # every rank empties its C-level stdout (RCCL's banner) before rank 0 prints the line, so that the line is the last one
    try:
        import ctypes
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)
    except Exception:  # noqa: BLE001
        pass
    barrier(world)
    if rank != 0:
        return

    host = main_leg.pop("_host", None)
    tk = main_leg["topk"]
    line = {
        "metric": "BPR-MF positive-pairs/sec + full-catalog top-k users/sec",
        "value": main_leg["value"], "unit": "pairs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": main_leg["ms_per_step"], "repeats_ms_per_step": main_leg["repeats_ms_per_step"], "repeats": REPEATS,
        "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        # the second half of the metric, at the top level (the full object is "topk")
        "topk_users_per_s": tk["value"], "topk_ms_per_block": tk["ms_per_step"], "topk_frac": tk["roofline"]["frac"],
        "config": {"workload": ("BPRMF d=128, synthetic 10M users x 1M items (the shape BASELINE.json's metric / north_star quote the target on; "
                                "fits one MI355X)" if (U, I, F) == (10_000_000, 1_000_000, 128)
                                else "BPRMF d=128, synthetic 1M users x 100K items (BASELINE configs[1])" if (U, I, F) == (1_000_000, 100_000, 128)
                                else f"BPRMF d={F}, synthetic {U} users x {I} items"),
                   "users": U, "items": I, "factors": F, "interactions": main_leg["interactions"], "batch": B,
                   "batch_per_gpu": B, "optimizer": args.opt, "replay": args.replay if (not sharded and args.opt == "adam_tf_dense") else None,
                   "topk_block": main_leg["topk_block"], "k": k,
                   "parallelism": main_leg["parallelism"] + ("; top-k: see topk.sharding" if sharded else ""),
                   "scaling_note": ("one model of this shape partitioned over the ranks (user rows sharded, item table replicated), B triplets "
                                    "per rank and step: the batch grows with N, the tables do not") if sharded else None,
                   "world_size_observed": world, "backend": backend, "collectives_through": args.comm if sharded else None},
        "loss_per_pair_last": main_leg["loss_per_pair_last"],
        "roofline": main_leg["roofline"],
        "topk": main_leg["topk"],
    }
    for key in ("collectives", "metrics"):
        if key in main_leg:
            line[key] = main_leg[key]
    if other is not None:
        line["replay_other"] = other
    if second is not None:
        line["item_shard"] = {kk: second[kk] for kk in ("value", "unit", "ms_per_step", "scaling", "parallelism", "loss_per_pair_last",
                                                       "roofline", "topk", "collectives") if kk in second}
    if c2 is not None:
        line["c2"] = {kk: c2[kk] for kk in ("workload", "value", "unit", "ms_per_step", "repeats_ms_per_step", "interactions", "loss_per_pair_last",
                                            "roofline", "topk", "metrics") if kk in c2}
    if c5 is not None:
        line["c5_per_gpu"] = {kk: c5[kk] for kk in ("workload", "value", "unit", "ms_per_step", "repeats_ms_per_step", "interactions",
                                                    "loss_per_pair_last", "roofline", "topk") if kk in c5}
    if sweep is not None:
        line["batch_sweep"] = sweep
    if plugin is not None:
        line["plugin_e2e"] = plugin
    if vae is not None:
        line["vae"] = vae
    if neumf is not None:
        line["neumf"] = neumf
    if graph is not None:
        line["graph"] = graph
    if want_cpu and host is not None:
        line["cpu_baseline"] = cpu_baseline(args, host)
    emit(line, args)


if __name__ == "__main__":
    try:
        main()
    finally:
        import torch.distributed as _dist
        if _dist.is_available() and _dist.is_initialized():
            _dist.destroy_process_group()
