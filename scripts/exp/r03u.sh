#!/bin/bash
# band of the screened top-k threshold (thr = T - band E): candidates, fallback users and block time, trained and fresh tables
cd $GRAFT_REPO_ROOT
O=gpurun_out/r03u; mkdir -p $O
for ts in 40 0; do
for b in 2.0 1.5 1.25 1.0 0.75; do
  EL_SCREEN_BAND=$b EL_SCREEN_PROF=1 timeout 300 python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo auto --train-steps $ts --iters 3 > $O/prof_${ts}_$b.log 2>&1
  EL_SCREEN_BAND=$b timeout 300 python scripts/mb.py topk --users 131072 --items 100000 --factors 128 --algo auto --train-steps $ts --iters 5 > $O/time_${ts}_$b.log 2>&1
  echo "== train_steps=$ts band=$b"; grep "final per user\|flagged" $O/prof_${ts}_$b.log | tail -2 | cut -c1-260; grep "\[default\]\|k_screen_pass\|k_screen_final\|k_screen_thr\|k_score_topk\|k_topk_wave\|k_list" $O/time_${ts}_$b.log | cut -c1-120
done; done 2>&1 | tee $O/summary.log
