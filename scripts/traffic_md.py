#!/usr/bin/env python
"""gpurun_out/traffic/summary.json (scripts/collect_traffic.sh) -> the markdown table under profiles/.
usage: python scripts/traffic_md.py > profiles/r02_pmc_traffic.md"""
import json
import os

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(os.path.join(R, "gpurun_out", "traffic", "summary.json")))
c = s["config"]
U, I, F, B = c["users"], c["items"], c["factors"], c["batch"]
alg = {"k_adam_rows_Gu": 24 * U * F, "k_adam_dense_Gi": 24 * I * F, "k_bpr_user_seg": B * (16 * F + 28), "k_bpr_item_seg": 2 * B * (8 * F + 12),
       "k_bpr_sample": 48 * B}
print("# HBM traffic per launch from rocprofv3 PMC (round 2, same tree as `profiles/traffic.json`)\n")
print("Collected by `scripts/collect_traffic.sh` (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, `--kernel-trace` only) on the bench")
print(f"workload: train step B = {B:,} triplets, U = {U // 10**6}M, I = {I // 1000}K, F = {F} (`scripts/mb.py train`), one top-k block of "
      f"{c['topk_block']:,} users")
print("(`scripts/mb.py topk --algo screen`).  Counters are KiB, per-dispatch averages; reads are doubled (gfx950 FETCH_SIZE correction,")
print(f"MI355X_MICROARCH.md HBM section), WRITE_SIZE is used as is.  Kernel-source hash `{s['source_hash']}`.\n")
print("| kernel | FETCH_SIZE MiB | x2 MiB | WRITE_SIZE MiB | HBM bytes per launch | algorithmic bytes per launch | ratio |")
print("|---|---|---|---|---|---|---|")
for k in sorted(s["kernels"]):
    d = s["kernels"][k]
    f = d.get("FETCH_SIZE", {}).get("KiB_per_dispatch", 0.0) / 1024
    w = d.get("WRITE_SIZE", {}).get("KiB_per_dispatch", 0.0) / 1024
    tot = (2 * f + w) * 1024 * 1024
    a = alg.get(k)
    print(f"| `{k[:60]}` | {f:.1f} | {2 * f:.1f} | {w:.1f} | {tot / 1e6:.0f} MB | " + (f"{a / 1e6:.0f} MB | {tot / a:.2f} |" if a else "-- | -- |"))
print("\nAlgorithmic bytes: SURVEY 8d figures x the units of one launch (DESIGN.md 6): dense Adam 24 B / parameter, user segments")
print("`16 F + 28` B / triplet, item segments `8 F + 12` B / occurrence (2 per triplet), sampler 48 B / triplet.  Ratios below 1 are L2 /")
print("Infinity-Cache hits on re-used rows (hot items, the user rows a triplet shares with its neighbours); the sampler's 8x is its")
print("per-user record + CSR probes (cache lines, not bytes, are what a random access costs).  The rocPRIM rows average the step's")
print("3 M-pair sort together with the 10^8-element sorts torch runs while the synthetic data set is generated in the same process.")
