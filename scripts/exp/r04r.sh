#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-r04r}; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest tests/test_gpu_nmf_score.py -m gpu -q --timeout 600 2>&1 | tail -8) > gpurun_out/$T/pytest.log
cat gpurun_out/$T/pytest.log
EL_NMF_SCREEN=1 timeout 300 python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 3 2>&1 | grep -v amdgpu.ids | tail -18 > gpurun_out/$T/log.txt
cat gpurun_out/$T/log.txt
export EL_NMF_SCREEN=1
timeout 600 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES -d $R/gpurun_out/$T/pmc -o pmc --output-format csv -- python scripts/mb.py nmfscore --users 1250000 --items 1000000 --factors 128 --score-users 128 --iters 1 > gpurun_out/$T/pmc.log 2>&1
python - <<'PY' > gpurun_out/$T/pmc_screen.txt 2>&1
import csv, glob, collections, os
T = os.environ.get("T", "")
for f in glob.glob("gpurun_out/*/pmc/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_nmf_screen" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in acc.items(): print(k, v / max(n[k], 1), n[k])
PY
cat gpurun_out/$T/pmc_screen.txt; tail -3 gpurun_out/$T/pmc.log
rm -rf gpurun_out/$T/pmc
