#!/bin/bash
# A/B of the headline BPR training step: each argument is "name|ENV=.. ENV=.." ; prints ms/step + per-kernel breakdown per variant
# usage (on the GPU box): scripts/exp/bpr_ab.sh "base|" "w3|EL_LIB_PATH=elliot_amd/csrc/variants/libelliot_hip_w3.so" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%|*}; envs=${spec#*|}
  env $envs python bench.py --legs bpr --no-cpu-baseline --trained-epochs 0 --legs-file gpurun_out/ab_$name.json > gpurun_out/ab_$name.line 2> gpurun_out/ab_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/ab_$name.json"))
k=d["roofline"]["kernels_ms_per_step"]
print("$name", "ms/step", round(d["ms_per_step"],4), "reps", [round(x,3) for x in d["repeats_ms_per_step"]], {n: round(v,3) for n,v in k.items() if v>0.015})
o=d.get("replay_other")
if o: print("   other(", o["replay"], ")", round(o["ms_per_step"],4))
PY
done
