#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_14; cd $R; mkdir -p gpurun_out/$T
for c in 64 128 192 256 384; do
  EL_ICHUNK=$c timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --topk-block 16384 --legs-file gpurun_out/$T/legs_$c.json 2>/dev/null | tail -1 > /dev/null
done
python - <<PY
import json
for c in (64,128,192,256,384):
    d=json.load(open("gpurun_out/$T/legs_%d.json"%c)); r=d["roofline"]["kernels_ms_per_step"]
    print(c, round(d["ms_per_step"],4), round(d["value"]/1e6,1), {k:round(v,4) for k,v in r.items() if k in ("k_bpr_item_seg","k_bpr_item_split","k_bpr_user_seg","k_bpr_flush_items")})
PY
