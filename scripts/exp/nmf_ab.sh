#!/bin/bash
# The NeuMF leg of the bench with its per-kernel breakdown.  usage (on the GPU box): scripts/exp/nmf_ab.sh "name|ENV=..[|extra bench args]" ...
mkdir -p gpurun_out
for spec in "$@"; do
  name=${spec%%|*}; rest=${spec#*|}; envs=${rest%%|*}; extra=""; [[ "$rest" == *"|"* ]] && extra=${rest#*|}
  env $envs python bench.py $extra --legs neumf --no-cpu-baseline --neumf-trained-steps 0 --legs-file gpurun_out/nmf_$name.json > gpurun_out/nmf_$name.line 2> gpurun_out/nmf_$name.err
  python - <<PY
import json
d=json.load(open("gpurun_out/nmf_$name.json"))
n=d.get("legs",{}).get("neumf") or d.get("neumf") or d
r=n["roofline"]
print("$name", "ms/step", round(n["ms_per_step"],4), "reps", [round(x,3) for x in n.get("repeats_ms_per_step",[])], "gemm", round(r["gemm_ms_per_step"],3), "rows", r.get("embedding_rows_ms_per_step"), "non_gemm", r.get("non_gemm_ms_per_step"))
print("   ", {k: round(v,3) for k,v in r["kernels_ms_per_step"].items() if v>0.01})
PY
done
