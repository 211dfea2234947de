#!/bin/bash
# PMC passes for the fused top-k kernel (separate runs per counter group; --kernel-trace only).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_topk
mkdir -p $OUT
cd $R
CMD="python scripts/mb.py topk --users 65536 --iters 2 $EXTRA"
python scripts/mb.py topk --users 131072 --iters 3 $EXTRA > $OUT/plain.txt 2>&1
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/g$i -o p -- $CMD > $OUT/g$i.log 2>&1
done
python - <<PY
import csv, glob, collections
for f in sorted(glob.glob("$OUT/g*/*counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
    for k, d in agg.items():
        if "topk" in k or "screen" in k:
            print(f, k, dict(d))
PY
cat $OUT/plain.txt | tail -3
