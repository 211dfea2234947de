#!/bin/bash
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; T=r5_11; cd $R; mkdir -p gpurun_out/$T
run() { tag=$1; shift; env "$@" timeout 600 python bench.py --legs vae --no-cpu-baseline --trained-epochs 0 --users 200000 --items 50000 --legs-file gpurun_out/$T/legs_$tag.json 2>/dev/null | tail -1 > /dev/null; }
run side0 EL_VAE_SIDE=0
run side1_main EL_VAE_SIDE_INDEX=0
run side1_side EL_VAE_SIDE_INDEX=1
run side1_main_b EL_VAE_SIDE_INDEX=0
run side1_side_b EL_VAE_SIDE_INDEX=1
python - <<PY
import json
for t in ("side0","side1_main","side1_side","side1_main_b","side1_side_b"):
    d=json.load(open("gpurun_out/$T/legs_%s.json"%t)); v=d["vae"]
    print(t, round(v["value"]), round(v["ms_per_step"],4), [round(x,4) for x in v["repeats_ms_per_step"]])
PY
