#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=r04y; mkdir -p gpurun_out/$T
export EL_NMF_SCREEN=1
for c in FETCH_SIZE "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum"; do
timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$T/pmc -o pmc --output-format csv -- python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 1 > gpurun_out/$T/pmc.log 2>&1
python - <<'PY' >> gpurun_out/$T/pmc_screen.txt 2>&1
import csv, glob, collections
for f in glob.glob("gpurun_out/r04y/pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_nmf_screen" in r["Kernel_Name"] or "k_nmf_score" in r["Kernel_Name"]:
            key = (r["Kernel_Name"][:24], r["Counter_Name"])
            acc[key] += float(r["Counter_Value"]); n[key] += 1
    for k, v in acc.items(): print(k, v / max(n[k], 1), n[k])
PY
rm -rf gpurun_out/$T/pmc
done
cat gpurun_out/$T/pmc_screen.txt; tail -2 gpurun_out/$T/pmc.log
