#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_17; cd $R; mkdir -p gpurun_out/$T
for c in 16 8 32 16; do
  EL_UCHUNK=$c timeout 600 python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --topk-block 16384 --legs-file gpurun_out/$T/legs_${c}_$RANDOM.json 2>/dev/null | tail -1 > /dev/null
done
python - <<PY
import json, glob
for f in sorted(glob.glob("gpurun_out/$T/legs_*.json")):
    d=json.load(open(f)); r=d["roofline"]["kernels_ms_per_step"]
    print(f.split("legs_")[1], round(d["ms_per_step"],4), round(d["value"]/1e6,1), {k:round(v,4) for k,v in r.items() if k in ("k_bpr_item_seg","k_bpr_user_seg","k_bpr_flush_items","k_bpr_flush_users")})
PY
