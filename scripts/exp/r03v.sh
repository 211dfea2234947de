#!/bin/bash
# adaptive band of the screened top-k threshold: default [1, 2] against the fixed band 2, trained and fresh tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03v; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_topk.py tests/test_gpu_fullsize.py tests/test_gpu_fullsize_c5.py -q -x > $O/pytest.log 2>&1
tail -4 $O/pytest.log
for ts in 40 0; do
for b in "default" "2,2"; do
  if [ "$b" = default ]; then unset EL_SCREEN_BAND; else export EL_SCREEN_BAND=$b; fi
  EL_SCREEN_PROF=1 timeout 300 python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo auto --train-steps $ts --iters 3 > $O/prof_${ts}_$b.log 2>&1
  timeout 300 python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo auto --train-steps $ts --iters 5 > $O/time_${ts}_$b.log 2>&1
  echo "== train_steps=$ts band=$b"; grep "final per user\|flagged" $O/prof_${ts}_$b.log | tail -2 | cut -c1-260; grep "\[default\]\|k_screen_pass2\|k_screen_final\|k_score_topk\|k_list" $O/time_${ts}_$b.log | cut -c1-120
done; done 2>&1 | tee $O/summary.log
unset EL_SCREEN_BAND
timeout 600 python bench.py --legs bpr --no-cpu-baseline --repeats 3 2> $O/bpr.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); t=d['topk']
print('bench topk', round(t['ms_per_step'],3), round(t['value']/1e6,2), {k:round(v,3) for k,v in t['roofline']['kernels_ms_per_step'].items()}, t.get('fragile_users'))" 2>&1 | tee $O/bench_topk.log
