#!/bin/bash
# HBM traffic per launch (rocprofv3 PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --kernel-trace only -- the guide's
# recipe) of the dominant kernels of EVERY bench leg: c2 = BASELINE configs[1] (headline), c4 = 10M x 1M x 128 on one GPU,
# c5 = configs[4] per-GPU shape (6.25M x 5M x 256), vae = configs[2], neumf = configs[3] per-GPU shape (train step + fused scoring).
# Writes gpurun_out/traffic/summary.json (+ the hash of the kernel sources it was collected on); copy into
# profiles/traffic.json with `python scripts/stamp_traffic.py` (adds the commit).   usage: collect_traffic.sh [legs...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic
mkdir -p $OUT
cd $R
LEGS=${@:-c2 c4 c5 vae neumf}
run() {   # run <tag> <mb.py args...>
  tag=$1; shift
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${tag}_$c -o p -- python scripts/mb.py "$@" > $OUT/${tag}_$c.log 2>&1
  done
}
for leg in $LEGS; do
  case $leg in
    c2) run c2_train train --users 1000000 --items 100000 --factors 128 --iters 3 --algo auto
        run c2_topk topk --users 131072 --items 100000 --factors 128 --iters 2 --algo screen --train-steps 20 ;;
    c4) run c4_train train --users 10000000 --items 1000000 --factors 128 --iters 3 --algo auto
        run c4_topk topk --users 131072 --items 1000000 --factors 128 --iters 2 --algo screen ;;
    c5) run c5_train train --users 6250000 --items 5000000 --factors 256 --iters 2 --algo auto
        run c5_topk topk --users 131072 --items 5000000 --factors 256 --iters 1 --algo screen ;;
    vae) run vae_step vae --iters 4 ;;
    neumf) run neumf_step nmf --users 1250000 --items 1000000 --factors 128 --batch 262144 --iters 3 --model NeuMF
           run neumf_topk nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 2 ;;
  esac
done
python - <<PY
import csv, glob, collections, json, os, re, sys
sys.path.insert(0, "$R")
import bench
def short(name):
    k = name.replace("void ", "")
    base = k.split("<")[0].split("(")[0]
    if base == "k_screen_pass":
        mode = re.findall(r"<\s*\d+\s*,\s*(\d+)", k)
        return "k_screen_pass" + (mode[0] if mode else "")
    if base == "k_adam_rows":
        return "k_adam_rows_Gu"
    if base.startswith("k_gemm_f32"):
        return "k_gemm_f32"
    if base.startswith("k_gemm_b3"):
        return "k_gemm_b3"
    return base
res = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    tag = os.path.basename(os.path.dirname(f))            # <leg>_<what>_<COUNTER>
    leg = tag.split("_")[0]
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_") or "radix" in k or "onesweep" in k:
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[leg][k][c] = {"KiB_per_dispatch": v / cnt[(k, c)], "dispatches": cnt[(k, c)], "KiB_total": v}
cfgs = {"c2": {"users": 1000000, "items": 100000, "factors": 128, "batch": 1 << 20, "topk_block": 131072},
        "c4": {"users": 10000000, "items": 1000000, "factors": 128, "batch": 1 << 20, "topk_block": 131072},
        "c5": {"users": 6250000, "items": 5000000, "factors": 256, "batch": 1 << 20, "topk_block": 131072},
        "vae": {"shape": "138493,26744,600,200,512", "steps_profiled": 6},
        "neumf": {"shape": "1250000,1000000,128,262144", "steps_profiled": 5, "topk_users": 128}}
out = {"source_hash": bench.source_hash(), "workloads": {leg: {"config": cfgs.get(leg, {}), "kernels": res[leg]} for leg in res}}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for leg in res:
    for k, d in sorted(res[leg].items()):
        print(leg, k[:40], {c: round(x["KiB_per_dispatch"] / 1024, 1) for c, x in d.items()}, "MiB/dispatch")
PY
