#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_19; cd $R; mkdir -p gpurun_out/$T
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --topk-block 16384 --steps 10 --repeats 1 --legs-file gpurun_out/$T/legs_$tag.json 2>/dev/null | tail -1 > /dev/null; }
run none1 X=1
run g0a EL_ITEM_LAYOUT_GAP_MIB=0
run g05a EL_ITEM_LAYOUT_GAP_MIB=0.5
run g25a EL_ITEM_LAYOUT_GAP_MIB=2.5
run none2 X=1
run g0b EL_ITEM_LAYOUT_GAP_MIB=0
run g05b EL_ITEM_LAYOUT_GAP_MIB=0.5
run g25b EL_ITEM_LAYOUT_GAP_MIB=2.5
python - <<PY
import json
for t in ("none1","none2","g0a","g0b","g05a","g05b","g25a","g25b"):
    d=json.load(open("gpurun_out/$T/legs_%s.json"%t)); r=d["roofline"]["kernels_ms_per_step"]
    print(t, round(d["ms_per_step"],4), {k:round(v,4) for k,v in r.items() if k in ("k_bpr_item_seg","k_bpr_flush_items","k_bpr_user_seg")})
PY
