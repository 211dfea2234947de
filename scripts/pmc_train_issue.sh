#!/bin/bash
# Issue-side counters of the headline training step's kernels (10M x 1M x 128, steady state of the deferred decay): how much of
# k_bpr_user_seg / k_bpr_flush_users is VALU work (the replay of postponed Adam steps) and how much is waiting on memory.
#   usage (through gpurun): bash scripts/pmc_train_issue.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_train_issue}
mkdir -p $OUT
cd $R
CMD="python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --steps 6 --warmup 1 --repeats 1 --legs-file /tmp/legs_pmc.json"
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- $CMD > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python - <<PY
import csv, glob, collections, json
res = collections.defaultdict(dict); n = collections.defaultdict(dict)
for f in sorted(glob.glob("$OUT/g*/*counter_collection.csv")):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k.startswith("k_bpr"):
            agg[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
    for (k, c), v in agg.items():
        res[k][c] = v / cnt[(k, c)]; n[k][c] = cnt[(k, c)]
json.dump({"per_dispatch": res, "dispatches": n}, open("$OUT/summary.json", "w"), indent=1)
for k, d in sorted(res.items()):
    print(k)
    for c, x in sorted(d.items()):
        print("   ", c, round(x, 1), "x", n[k][c])
PY
rm -rf $OUT/g1 $OUT/g2 $OUT/g3 $OUT/g4
