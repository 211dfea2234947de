#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_12; cd $R; mkdir -p gpurun_out/$T
(timeout 600 python -m pytest -q -m gpu --timeout 600 tests/test_gpu_dense.py tests/test_gpu_plugin.py 2>&1 | tail -5) > gpurun_out/$T/pytest.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --legs vae --no-cpu-baseline --trained-epochs 0 --users 200000 --items 50000 --legs-file gpurun_out/$T/legs_$tag.json 2>/dev/null | tail -1 > /dev/null; }
run early0 EL_VAE_EARLY_ADAM=0
run early1 EL_VAE_EARLY_ADAM=1
run early0_b EL_VAE_EARLY_ADAM=0
run early1_b EL_VAE_EARLY_ADAM=1
tail -3 gpurun_out/$T/pytest.log
python - <<PY
import json
for t in ("early0","early1","early0_b","early1_b"):
    d=json.load(open("gpurun_out/$T/legs_%s.json"%t)); v=d["vae"]
    print(t, round(v["value"]), round(v["ms_per_step"],4), [round(x,4) for x in v["repeats_ms_per_step"]])
PY
