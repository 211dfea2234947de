#!/usr/bin/env python
"""gpurun_out/traffic/summary.json (scripts/collect_traffic.sh) -> the markdown table under profiles/.
usage: python scripts/traffic_md.py [round tag] > profiles/r03_pmc_traffic.md"""
import json
import os
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
s = json.load(open(os.path.join(R, "gpurun_out", "traffic", "summary.json")))
tag = sys.argv[1] if len(sys.argv) > 1 else "round 4"
print(f"# HBM traffic per launch from rocprofv3 PMC ({tag}, same tree as `profiles/traffic.json`)\n")
print("Collected by `scripts/collect_traffic.sh` (separate `--pmc FETCH_SIZE` and `--pmc WRITE_SIZE` passes, `--kernel-trace` only) on the workload")
print("of every bench leg (`scripts/mb.py train | topk | vae | nmf | nmfscore`).  Counters are KiB, per-dispatch averages; reads are doubled (gfx950")
print(f"FETCH_SIZE correction, MI355X_MICROARCH.md HBM section), WRITE_SIZE is used as is.  Kernel-source hash `{s['source_hash']}`.\n")
for leg, w in s["workloads"].items():
    c = w["config"]
    alg = {}
    if "users" in c:
        U, I, F, B = c["users"], c["items"], c["factors"], c["batch"]
        alg = {"k_adam_rows_Gu": 24 * U * F, "k_adam_dense_Gi": 24 * I * F, "k_bpr_user_seg": B * (16 * F + 28), "k_bpr_item_seg": 2 * B * (8 * F + 12),
               "k_bpr_sample": 48 * B, "k_bpr_user_adam": 24 * U * F + B * (8 * F + 28)}
        import math
        # fused item side (el_bprmf_state.Gi_last): the segments gather gamma_u per occurrence and move theta, m, v of the batch's distinct
        # items in place -- expected distinct items: B uniform negatives + the positives' popularity (taken as ~0.7 I (1 - exp(-2B/I)))
        rows_i = I * (1.0 - math.exp(-2.0 * B / I)) * 0.85
        alg["k_bpr_item_seg"] = 2 * B * (4 * F + 12) + 24 * rows_i * (F + 1)
        alg["k_bpr_flush_items"] = 24 * max(I - rows_i, 0) * F
        alg["k_bpr_catchup_items"] = 24 * rows_i * F
        if 4 * B <= U:
            # deferred decay of the user table (the state turns it on when 4 B <= U): the user side moves the rows of the batch's
            # distinct users only -- expected U (1 - exp(-B / U)) of them for uniformly drawn users
            rows = U * (1.0 - math.exp(-B / U))
            alg["k_bpr_user_seg"] = 28 * rows * F + B * (8 * F + 32)      # theta, m, v read + written, old row written, gamma_i / gamma_j
            alg["k_bpr_catchup"] = 24 * rows * F                           # upper bound: every one of those rows replayed and rewritten
            alg["k_bpr_flush_users"] = 24 * U * F
        print(f"## {leg}: BPRMF {U:,} users x {I:,} items, d = {F}, B = {B:,}, top-k block {c['topk_block']:,}\n")
    else:
        print(f"## {leg}: {c}\n")
    print("| kernel | dispatches | FETCH_SIZE MiB | x2 MiB | WRITE_SIZE MiB | HBM bytes per launch | algorithmic bytes per launch | ratio |")
    print("|---|---|---|---|---|---|---|---|")
    for k in sorted(w["kernels"]):
        d = w["kernels"][k]
        f = d.get("FETCH_SIZE", {}).get("KiB_per_dispatch", 0.0) / 1024
        wr = d.get("WRITE_SIZE", {}).get("KiB_per_dispatch", 0.0) / 1024
        n = d.get("FETCH_SIZE", {}).get("dispatches", 0)
        tot = (2 * f + wr) * 1024 * 1024
        a = alg.get(k)
        print(f"| `{k[:60]}` | {n} | {f:.1f} | {2 * f:.1f} | {wr:.1f} | {tot / 1e6:.0f} MB | " + (f"{a / 1e6:.0f} MB | {tot / a:.2f} |" if a else "-- | -- |"))
    print()
print("Algorithmic bytes: SURVEY 8d figures x the units of one launch (DESIGN.md 6): dense Adam 24 B / parameter; the fused user-side kernel")
print("`k_bpr_user_adam` = 24 B / parameter of the user table + `8 F + 28` B / triplet (gamma_i, gamma_j gathers); user segments `16 F + 28` B /")
print("triplet (with the deferred decay of the user table, workloads with 4 B <= U: theta, m, v of the batch's distinct users read + written and the")
print("pre-update row written, `28 F` B / row, + `8 F + 32` B / triplet; `k_bpr_catchup` at most 24 F B / row -- rows whose m = v = 0 are skipped),")
print("item segments `8 F + 12` B / occurrence (2 per triplet), sampler 48 B / triplet.  Ratios below 1 are L2 / Infinity-Cache hits on")
print("re-used rows (hot items).  Round 4: the item segments carry the Adam step of their rows (`2 B (4 F + 12)` of gathers + `24 (F + 1)` B per distinct item of")
print("the batch, the count estimated as 0.85 I (1 - exp(-2 B / I))); `k_bpr_flush_items` = the rows the batch left alone, one step each; the user segments of")
print("the deferred form also replay a row's postponed steps (the `k_bpr_catchup` launch of round 3 is gone).  The rocPRIM rows average the step's 3 M-pair sort together with the 10^8-element sorts torch runs while the")
print("synthetic data set is generated in the same process; `k_gemm_f32` rows of the vae / neumf legs average launches of many shapes.")
