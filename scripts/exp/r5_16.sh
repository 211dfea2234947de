#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_16; cd $R; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest -q -m gpu --timeout 600 tests/test_gpu_bpr.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c5.py tests/test_gpu_plugin.py 2>&1 | tail -12) > gpurun_out/$T/pytest.log
timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --topk-block 16384 --legs-file gpurun_out/$T/legs.json 2>/dev/null | tail -1 > /dev/null
tail -4 gpurun_out/$T/pytest.log
python - <<PY
import json
d=json.load(open("gpurun_out/$T/legs.json")); r=d["roofline"]["kernels_ms_per_step"]
print(round(d["ms_per_step"],4), round(d["value"]/1e6,1), {k:round(v,4) for k,v in r.items()})
PY
