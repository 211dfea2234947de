#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03x; mkdir -p $O
for b in "default" "1.5,0.9" "1.5,1.0" "1.5,1.25" "2,2"; do
  if [ "$b" = default ]; then unset EL_SCREEN_BAND; else export EL_SCREEN_BAND=$b; fi
  EL_SCREEN_PROF=1 timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 1 --steps 5 > /dev/null 2> $O/prof_$b.err
  timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 3 2> $O/bpr_$b.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); t=d['topk']
print('band $b topk', round(t['ms_per_step'],3), round(t['value']/1e6,2), {k:round(v,3) for k,v in t['roofline']['kernels_ms_per_step'].items() if v>0.02})"
  grep "flagged\|final per user" $O/prof_$b.err | tail -2 | cut -c1-230
done 2>&1 | tee $O/summary.log
