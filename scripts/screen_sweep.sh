#!/bin/bash
# sweep of the screened top-k's threshold policy (pass-1 stride / kA) on the bench workload: ms per 131072-user block + kernel split
cd $GRAFT_REPO_ROOT
for cfg in "default" "EL_SCREEN_STRIDE=8 EL_SCREEN_KA=6" "EL_SCREEN_STRIDE=8 EL_SCREEN_KA=5" "EL_SCREEN_STRIDE=6 EL_SCREEN_KA=6" "EL_SCREEN_STRIDE=6 EL_SCREEN_KA=7" "EL_SCREEN_STRIDE=4 EL_SCREEN_KA=7" "EL_SCREEN_STRIDE=4 EL_SCREEN_KA=6" "EL_SCREEN_STRIDE=12 EL_SCREEN_KA=5" "EL_SCREEN_STRIDE=16 EL_SCREEN_KA=4"; do
  env $( [ "$cfg" = default ] || echo $cfg ) python bench.py --legs bpr --no-cpu-baseline --steps 14 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); t=d['topk']; k=t['roofline']['kernels_ms_per_step']
print('$cfg'.ljust(36), 'ms/block %.3f  users/s %.2fM ' % (t['ms_per_step'], t['value']/1e6), {a.replace('k_screen_',''):round(b,3) for a,b in k.items() if b>0.02})"
done
