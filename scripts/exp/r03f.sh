#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03f; mkdir -p $O
python -m pytest tests/test_gpu_bpr.py -x -q > $O/pytest_bpr.log 2>&1
tail -15 $O/pytest_bpr.log
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_plugin.py tests/test_gpu_cml.py -x -q > $O/pytest_more.log 2>&1
tail -3 $O/pytest_more.log
python bench.py --legs bpr --no-cpu-baseline > $O/bench_fused.json 2> $O/bench_fused.err
EL_FUSED_USER=0 python bench.py --legs bpr --no-cpu-baseline > $O/bench_unfused.json 2> $O/bench_unfused.err
python - <<PY
import json
for n in ("fused","unfused"):
    try:
        d=json.loads([l for l in open("$O/bench_%s.json"%n) if l.startswith("{")][0])
        print(n, d["ms_per_step"], d["value"], {k:round(v,3) for k,v in d["roofline"]["kernels_ms_per_step"].items()})
    except Exception as e: print(n, "ERR", e)
PY
