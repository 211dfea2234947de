#!/bin/bash
# PMC counters of the fp32 GEMM kernel (one group per pass, --kernel-trace only): MFMA busy, wave-state split, LDS bank
# conflicts, occupancy-related counts.  usage: bash scripts/pmc_gemm.sh "<shape args of scripts/mb.py gemm>" tag
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
TAG=${2:-pmc_gemm}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
CMD="python scripts/mb.py gemm --iters 4 $1"
i=0
for grp in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVE_CYCLES" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS" "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
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
        if k.startswith("k_gemm"):
            key = (k + " grid=" + r.get("Grid_Size", "?"), r["Counter_Name"])
            agg[key] += float(r["Counter_Value"]); cnt[key] += 1
    for (k, c), v in agg.items():
        res[k][c] = v / cnt[(k, c)]
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for k, d in res.items():
    print(k)
    print("   ", {c: round(x, 1) for c, x in d.items()})
    if "SQ_VALU_MFMA_BUSY_CYCLES" in d and "GRBM_GUI_ACTIVE" in d:
        print("    MFMA util = %.1f %%" % (100.0 * d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / (d["GRBM_GUI_ACTIVE"] / 8.0)))
    if "SQ_WAVE_CYCLES" in d and "SQ_WAIT_ANY" in d:
        w = d["SQ_WAVE_CYCLES"]
        print("    waves: parked (waitcnt/barrier) %.0f %%, issue-stalled (MFMA pipe / dependency) %.0f %%, issuing %.0f %%"
              % (100 * d["SQ_WAIT_ANY"] / w, 100 * d["SQ_WAIT_INST_ANY"] / w, 100 * d["SQ_ACTIVE_INST_ANY"] / w))
    if "SQ_LDS_BANK_CONFLICT" in d:
        print("    LDS: bank-conflict cycles %.0f of %.0f active (%.2f %%)" % (d["SQ_LDS_BANK_CONFLICT"], d["SQ_LDS_IDX_ACTIVE"],
              100.0 * d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)))
    if "FETCH_SIZE" in d:
        print("    HBM-side traffic per launch: %.1f MiB fetched (x2 correction applied), %.1f MiB written" % (2 * d["FETCH_SIZE"] / 1024, d.get("WRITE_SIZE", 0) / 1024))
    if "SQ_WAVES" in d:
        print("    waves launched %.0f" % d["SQ_WAVES"])
PY
