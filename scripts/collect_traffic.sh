#!/bin/bash
# HBM traffic per launch (rocprofv3 PMC: FETCH_SIZE and WRITE_SIZE in SEPARATE passes, --kernel-trace only -- the guide's
# recipe) of the training-step kernels and the screened top-k kernels on the bench workload (BASELINE configs[1]).
# Writes gpurun_out/traffic/summary.json (+ the hash of the kernel sources it was collected on); copy into
# profiles/traffic.json with `python scripts/stamp_traffic.py` (adds the commit).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/traffic
mkdir -p $OUT
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/train_$c -o p -- python scripts/mb.py train --users 1000000 --iters 3 --algo auto > $OUT/train_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/topk_$c -o p -- python scripts/mb.py topk --users 131072 --iters 2 --algo screen > $OUT/topk_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json, re, sys
sys.path.insert(0, "$R")
import bench
res = collections.defaultdict(dict)
def short(name):
    k = name.replace("void ", "")
    base = k.split("<")[0].split("(")[0]
    if base == "k_screen_pass":
        mode = re.findall(r"<\s*\d+\s*,\s*(\d+)", k)
        return "k_screen_pass" + (mode[0] if mode else "")
    if base == "k_adam_rows":
        return "k_adam_rows_Gu"
    return base
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        if k.startswith("k_") or "radix" in k or "onesweep" in k:
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[k][c] = {"KiB_per_dispatch": v / cnt[(k, c)], "dispatches": cnt[(k, c)]}
out = {"source_hash": bench.source_hash(), "kernels": res,
       "config": {"users": 1000000, "items": 100000, "factors": 128, "batch": 1 << 20, "topk_block": 131072}}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, d in sorted(res.items()):
    print(k[:40], {c: round(x["KiB_per_dispatch"] / 1024, 1) for c, x in d.items()}, "MiB/dispatch")
PY
