#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03ab; mkdir -p $O
for v in "8 6" "16 4" "16 5" "16 6" "4 8"; do
  set -- $v
  EL_SCREEN_STRIDE=$1 EL_SCREEN_KA=$2 timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 3 2> $O/bpr_$1_$2.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); t=d['topk']
print('stride $1 kA $2 topk', round(t['ms_per_step'],3), round(t['value']/1e6,2), {k:round(v,3) for k,v in t['roofline']['kernels_ms_per_step'].items() if v>0.02})"
done 2>&1 | tee $O/summary.log
