#!/bin/bash
# HBM traffic counters (FETCH_SIZE / WRITE_SIZE, separate passes) for the train-step kernels and the top-k kernel.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_train
mkdir -p $OUT
cd $R
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/train_$c -o p -- python scripts/mb.py train --users 1000000 --iters 3 --algo sorted > $OUT/train_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/topk_$c -o p -- python scripts/mb.py topk --users 131072 --iters 2 > $OUT/topk_$c.log 2>&1
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/*/*counter_collection.csv")):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_") or "radix" in k or "onesweep" in k:
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[k][c] = {"sum_KiB": v, "dispatches": cnt[(k, c)], "KiB_per_dispatch": v / cnt[(k, c)]}
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, d in res.items():
    print(k[:50], {c: round(x["KiB_per_dispatch"] / 1024, 1) for c, x in d.items()}, "MiB/dispatch")
PY
