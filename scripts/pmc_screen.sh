#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + MFMA busy counters of the screened top-k kernels on one bench block.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_screen
mkdir -p $OUT
cd $R
CMD="python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --iters 2 --algo screen"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- $CMD > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/g*/*counter_collection.csv")):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if k.startswith("k_"):
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[k][c] = v / cnt[(k, c)]
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, d in res.items():
    print(k[:40], {c: round(x, 1) for c, x in d.items()})
PY
