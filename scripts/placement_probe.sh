#!/bin/bash
# Process-to-process spread of the BPR segment kernels (k_bpr_item_seg 0.68 / 0.80 ms, k_bpr_user_seg 0.95 / 1.04 ms on one box, one
# binary): N fresh processes, per-kernel times + table addresses each; then the same process under rocprofv3 with the address-
# translation and L2 counters.   usage (through gpurun): bash scripts/placement_probe.sh [n_plain] [n_pmc]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/placement
mkdir -p $OUT
cd $R
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|TCC_EA0_RDREQ|TCC_HIT|TCC_MISS|TCC_REQ|TCC_TAG_STALL|TCP_TCC_READ_REQ_sum|TCP_PENDING|TCP_TA_TCP_STATE_READ" | sed 's/^[ \t]*//' | cut -c1-160 | sort -u > $OUT/counters_avail.txt
N=${1:-8}
for i in $(seq 1 $N); do
  pad=$(( (i % 4) * 1536 ))
  python scripts/placement_probe.py --tag plain$i --pad-mib $pad >> $OUT/plain.jsonl 2>> $OUT/plain.err
done
cat $OUT/plain.jsonl | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d['tag'], {k: v for k, v in d['ms'].items() if v > 0.05}, 'Gi', d['va']['Gi'], 'mGi', d['va']['mGi'], 'vGi', d['va']['vGi'], 'Gu', d['va']['Gu'])
"
M=${2:-4}
for i in $(seq 1 $M); do
  for grp in "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum"; do
    g=$(echo $grp | cut -c1-12 | tr ' ' '_')
    timeout 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $OUT/pmc${i}_$g -o p -- python scripts/placement_probe.py --steps 8 --tag pmc$i > $OUT/pmc${i}_$g.log 2>&1 || echo "pmc $i $g failed"
  done
done
python - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$OUT/pmc*_*/")):
    cc = glob.glob(d + "*counter_collection.csv"); kt = glob.glob(d + "*kernel_trace.csv")
    if not cc: continue
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(cc[0])):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
        if k in ("k_bpr_item_seg", "k_bpr_user_seg"):
            agg[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]
            if k in ("k_bpr_item_seg", "k_bpr_user_seg"):
                dur[k].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    print(d.split("/")[-2], {k: round(sum(v[-6:]) / len(v[-6:]), 4) for k, v in dur.items()},
          {f"{k}:{c}": round(sum(v[-6:]) / len(v[-6:])) for (k, c), v in sorted(agg.items())})
PY
rm -rf $OUT/pmc*_*/
