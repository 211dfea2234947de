#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_10; cd $R; mkdir -p gpurun_out/$T
for v in 0 1; do
  EL_SCREEN_PACE=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/$T/c5_pace$v -o p -- python scripts/mb.py topk --users 131072 --items 5000000 --factors 256 --iters 1 --algo screen > gpurun_out/$T/c5_pace$v.log 2>&1
done
python - <<PY
import csv, glob, collections
for v in (0,1):
    agg=collections.defaultdict(float); cnt=collections.Counter()
    for f in glob.glob("gpurun_out/$T/c5_pace%d/*/*counter_collection.csv"%v)+glob.glob("gpurun_out/$T/c5_pace%d/*counter_collection.csv"%v):
        for r in csv.DictReader(open(f)):
            k=r["Kernel_Name"].split("(")[0][:60]
            if "k_screen_pass" in k:
                import re
                m=re.findall(r"<\s*\d+\s*,\s*(\d+)", r["Kernel_Name"]); k="k_screen_pass"+(m[0] if m else "")
                agg[k]+=float(r["Counter_Value"]); cnt[k]+=1
    print("pace",v,{k:(round(2*agg[k]/cnt[k]/1048576,1),"GiB-equiv(2x FETCH KiB)/dispatch",cnt[k]) for k in agg})
PY
rm -rf gpurun_out/$T/c5_pace0 gpurun_out/$T/c5_pace1
