#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_18; cd $R; mkdir -p gpurun_out/$T
(EL_BPR_FLUSH_ROWS=4 EL_BPR_USER_PRE=3 timeout 600 python -m pytest -q -m gpu --timeout 600 tests/test_gpu_bpr.py 2>&1 | tail -3) > gpurun_out/$T/pytest_34.log
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --topk-block 16384 --legs-file gpurun_out/$T/legs_$tag.json 2>/dev/null | tail -1 > /dev/null; }
run warm EL_BPR_USER_PRE=2
run p2n2 EL_BPR_USER_PRE=2 EL_BPR_FLUSH_ROWS=2
run p3n2 EL_BPR_USER_PRE=3 EL_BPR_FLUSH_ROWS=2
run p2n4 EL_BPR_USER_PRE=2 EL_BPR_FLUSH_ROWS=4
run p3n4 EL_BPR_USER_PRE=3 EL_BPR_FLUSH_ROWS=4
run p2n2b EL_BPR_USER_PRE=2 EL_BPR_FLUSH_ROWS=2
tail -2 gpurun_out/$T/pytest_34.log
python - <<PY
import json
for t in ("warm","p2n2","p3n2","p2n4","p3n4","p2n2b"):
    d=json.load(open("gpurun_out/$T/legs_%s.json"%t)); r=d["roofline"]["kernels_ms_per_step"]
    print(t, round(d["ms_per_step"],4), round(d["value"]/1e6,1), {k:round(v,4) for k,v in r.items() if k in ("k_bpr_item_seg","k_bpr_user_seg","k_bpr_flush_items","k_bpr_flush_users")})
PY
