#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_15; cd $R; mkdir -p gpurun_out/$T
for c in 16 32 48 64 96; do
  EL_ICHUNK=$c timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --topk-block 16384 --legs-file gpurun_out/$T/legs_c4_$c.json 2>/dev/null | tail -1 > /dev/null
done
for c in 16 32 64 256; do
  EL_ICHUNK=$c timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --topk-block 4096 --users 6250000 --items 5000000 --factors 256 --legs-file gpurun_out/$T/legs_c5_$c.json 2>/dev/null | tail -1 > /dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$T/legs_*.json")):
    d=json.load(open(f)); r=d["roofline"]["kernels_ms_per_step"]
    print(f.split("legs_")[1], round(d["ms_per_step"],4), round(d["value"]/1e6,1), {k:round(v,4) for k,v in r.items() if k in ("k_bpr_item_seg","k_bpr_item_split","k_bpr_user_seg","k_bpr_flush_items","k_bpr_catchup_items")})
PY
