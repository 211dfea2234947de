#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-pmcs}; mkdir -p gpurun_out/$T
# (the screened route is the default of NmfDeviceState.score_topk_logits)
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/$T/pmc -o pmc --output-format csv -- python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 1 > gpurun_out/$T/pmc.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU -d $R/gpurun_out/$T/pmc2 -o pmc --output-format csv -- python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 1 >> gpurun_out/$T/pmc.log 2>&1
python - <<'PY' > gpurun_out/$T/pmc_screen.txt 2>&1
import csv, glob, collections, os
for f in sorted(glob.glob("gpurun_out/*/pmc*/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_nmf_screen" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in sorted(acc.items()): print(k, "%.4g" % (v / max(n[k], 1)), n[k])
PY
cat gpurun_out/$T/pmc_screen.txt; tail -2 gpurun_out/$T/pmc.log
rm -rf gpurun_out/$T/pmc gpurun_out/$T/pmc2
