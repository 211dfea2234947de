#!/bin/bash
# kA (how deep the guessed threshold aims) under the adaptive band, on the bench's tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03z; mkdir -p $O
for ka in 6 5 4 7 3; do
  EL_SCREEN_KA=$ka timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 3 2> $O/bpr_$ka.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); t=d['topk']
print('kA $ka topk', round(t['ms_per_step'],3), round(t['value']/1e6,2), {k:round(v,3) for k,v in t['roofline']['kernels_ms_per_step'].items() if v>0.02})"
done 2>&1 | tee $O/summary.log
for ka in 6 4; do
  EL_SCREEN_KA=$ka timeout 300 python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo auto --train-steps 40 --iters 5 2>&1 | grep "\[default\]" | sed "s/^/trained40 kA=$ka /"
  EL_SCREEN_KA=$ka timeout 300 python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo auto --train-steps 0 --iters 5 2>&1 | grep "\[default\]" | sed "s/^/fresh kA=$ka /"
done 2>&1 | tee -a $O/summary.log
