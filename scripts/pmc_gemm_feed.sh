#!/bin/bash
# Where k_gemm_b3's operand tiles come from: FETCH_SIZE (fabric-side reads of the L2s) and the L2 hit rate on three shapes.
#   usage (through gpurun): bash scripts/pmc_gemm_feed.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/${1:-pmc_gemm_feed}
mkdir -p $OUT
cd $R
CMD="python scripts/mb.py gemm --iters 5 --shape 4096,4096,4096,0,0 --shape 256,512,262144,1,0 --shape 262144,512,256,0,0 --shape 512,26744,600,0,0"
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- $CMD > $OUT/g$i.log 2>&1 || echo "group $i failed: $grp"
done
python - <<PY
import csv, glob, collections, json
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob("$OUT/g*/*counter_collection.csv")):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "k_gemm_b3" not in k:
            continue
        key = (k.split("(")[0].replace("void ", ""), r.get("Grid_Size", r.get("Grid_Size_X", "")), r.get("Workgroup_Size", ""))
        rows[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for key, d in rows.items():
    out[" | ".join(map(str, key))] = {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": len(next(iter(d.values())))}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
for k, d in out.items():
    print(k, {c: round(x, 1) for c, x in d.items()})
PY
grep "^gemm" $OUT/g1.log
rm -rf $OUT/g1 $OUT/g2 $OUT/g3
