#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; cd $R; T=${1:-pmcg}; mkdir -p gpurun_out/$T
for c in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES"; do
timeout 600 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/$T/pmc -o pmc --output-format csv -- python scripts/mb.py gemm --shape 4096,4096,4096,0,0 --iters 4 > gpurun_out/$T/pmc.log 2>&1
python - <<'PY' >> gpurun_out/$T/pmc.txt 2>&1
import csv, glob, collections
for f in sorted(glob.glob("gpurun_out/*/pmc/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(float); n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if "k_gemm_b3" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
    for k, v in sorted(acc.items()): print(k, "%.4g" % (v / max(n[k], 1)), n[k])
PY
rm -rf gpurun_out/$T/pmc
done
cat gpurun_out/$T/pmc.txt
