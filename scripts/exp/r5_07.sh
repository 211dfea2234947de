#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_07; cd $R; mkdir -p gpurun_out/$T
(timeout 1200 python -m pytest -q -m gpu --timeout 900 tests/test_gpu_bpr.py tests/test_gpu_topk.py tests/test_gpu_fullsize_c5.py tests/test_gpu_fullsize.py tests/test_gpu_neumf.py 2>&1 | tail -40) > gpurun_out/$T/pytest.log
timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --legs-file gpurun_out/$T/legs_bpr.json 2> /dev/null | tail -1 > gpurun_out/$T/line_bpr.json
for v in 0 1; do
  EL_SCREEN_PACE=$v timeout 600 python scripts/mb.py topk --users 131072 --items 5000000 --factors 256 --algo screen --iters 2 > gpurun_out/$T/topk_c5_pace$v.txt 2>&1
  EL_SCREEN_PACE=$v timeout 600 python scripts/mb.py topk --users 131072 --items 1000000 --factors 128 --algo screen --iters 3 > gpurun_out/$T/topk_c4_pace$v.txt 2>&1
done
tail -8 gpurun_out/$T/pytest.log
python - <<PY
import json
d=json.load(open("gpurun_out/$T/legs_bpr.json")); r=d["roofline"]
print("bpr", d["value"], d["ms_per_step"], r["frac"], {k:round(x,4) for k,x in r["kernels_ms_per_step"].items()})
print("topk", d["topk"]["value"], d["topk"]["ms_per_step"], d["topk"]["roofline"]["frac"])
PY
for f in topk_c5_pace0 topk_c5_pace1 topk_c4_pace0 topk_c4_pace1; do echo == $f; grep -v amdgpu.ids gpurun_out/$T/$f.txt | head -8; done
