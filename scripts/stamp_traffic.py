#!/usr/bin/env python
"""gpurun_out/traffic/summary.json (scripts/collect_traffic.sh, run on the GPU box) -> profiles/traffic.json, stamped with the
commit and the kernel-source hash it was collected on.  bench.py quotes `roofline.traffic` from this file only while the hash of
elliot_amd/csrc still matches, and says so in `roofline.traffic_source`.
HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (KiB -> bytes): FETCH_SIZE counts 64 B per 128-B request of wide coalesced
reads on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE is used as is.  One entry per bench leg (c2 = headline, c4, c5, vae,
neumf); for the GEMM kernel of the vae / neumf legs (launches of many shapes) the figure is per STEP (all its launches of a step)."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import bench  # noqa: E402


def per_launch(kernels, cfg, leg):
    out = {}
    for k, c in kernels.items():
        f, w = c.get("FETCH_SIZE"), c.get("WRITE_SIZE")
        if not f or not w:
            continue
        if leg in ("vae", "neumf") and k in ("k_gemm_f32", "k_gemm_b3", "k_gemm_reduce", "k_adam_dense", "k_adam_apply_dense"):
            steps = float(cfg.get("steps_profiled", 1))
            out[k + "_per_step"] = (2.0 * f["KiB_total"] + w["KiB_total"]) * 1024.0 / steps
        else:
            out[k] = (2.0 * f["KiB_per_dispatch"] + w["KiB_per_dispatch"]) * 1024.0
    if leg in ("vae", "neumf"):                          # every GEMM launch of a step: the split kernel, the fp32 one (small shapes), the split-K sums
        g = [out[k] for k in ("k_gemm_f32_per_step", "k_gemm_b3_per_step", "k_gemm_reduce_per_step") if k in out]
        if g:
            out["k_gemm_per_step"] = sum(g)
    if "k_adam_dense" in out and leg in ("c2", "c4", "c5"):
        # one kernel name, launches of different sizes per step (item table, item bias; the user table too in the dense form):
        # split the per-step total by element counts
        U, I, F = cfg["users"], cfg["items"], cfg["factors"]
        parts = {"k_adam_dense_Gi": I * F, "k_adam_dense_Bi": I}
        per_step = 2
        if "k_adam_rows_Gu" not in out:
            parts["k_adam_dense_Gu"] = U * F
            per_step = 3
        tot = out.pop("k_adam_dense") * per_step
        s = float(sum(parts.values()))
        for name, cnt in parts.items():
            out[name] = tot * cnt / s
    return out


def main():
    src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(REPO, "gpurun_out", "traffic", "summary.json")
    d = json.load(open(src))
    if d["source_hash"] != bench.source_hash():
        print(f"warning: collected on kernel sources {d['source_hash']}, the tree is at {bench.source_hash()}", file=sys.stderr)
    commit = subprocess.check_output(["git", "-C", REPO, "rev-parse", "--short", "HEAD"], text=True).strip()
    wl = {}
    for leg, w in d["workloads"].items():
        wl[leg] = {"config": w["config"], "bytes_per_launch": per_launch(w["kernels"], w["config"], leg)}
    res = {"note": "HBM bytes per launch (2*FETCH_SIZE + WRITE_SIZE, KiB->bytes) from rocprofv3 --pmc passes on the workload of every "
                   "bench leg (scripts/collect_traffic.sh); quoted by bench.py only while source_hash matches elliot_amd/csrc",
           "commit": commit, "source_hash": d["source_hash"], "workloads": wl}
    json.dump(res, open(os.path.join(REPO, "profiles", "traffic.json"), "w"), indent=1)
    for leg, w in wl.items():
        for k, v in sorted(w["bytes_per_launch"].items()):
            print(f"{leg:6s} {k:32s} {v / 1e6:10.1f} MB")


if __name__ == "__main__":
    main()
