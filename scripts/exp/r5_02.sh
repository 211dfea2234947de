#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_02; cd $R; mkdir -p gpurun_out/$T
(timeout 900 python -m pytest -q -m gpu -x --timeout 600 tests/test_gpu_bench_contract.py::test_one_rank_sharded_path_prints_the_same_contract tests/test_gpu_nmf_score.py tests/test_gpu_bpr.py 2>&1 | tail -150) > gpurun_out/$T/pytest.log
for v in 0 2 4; do
  EL_BPR_USER_PRE=$v timeout 600 python bench.py --legs bpr --no-cpu-baseline --legs-file gpurun_out/$T/legs_pre$v.json 2> /dev/null | tail -1 > gpurun_out/$T/line_pre$v.json
done
tail -30 gpurun_out/$T/pytest.log
python - <<PY
import json
for v in (0,2,4):
    d=json.load(open("gpurun_out/$T/legs_pre%d.json"%v))
    r=d["roofline"]
    print(v, d["value"], d["ms_per_step"], r["frac"], r.get("frac_back_to_back"), {k:round(x,4) for k,x in r["kernels_ms_per_step"].items()})
PY
